#!/bin/bash
# interleaved same-box A/B of two TREES (this one against tools/ab/old_tree, an exported earlier commit with its own library):
# tools/r06_ab_tree.sh <rounds> [bench args]      (NEW_ENV="A=1 B=2": extra configurations of THIS tree, one per word)
# the other tree, here (hipcc cross-compiles):  mkdir -p tools/ab/old_tree && git archive <commit> | tar -x -C tools/ab/old_tree &&
#   (cd tools/ab/old_tree && rm -rf profiles && python audio_diffusion_pytorch_amd/build.py)   -- tools/ab/ is git-ignored but travels
# An environment switch flips a code PATH inside one binary; only this comparison sees what a change did to the other paths' code.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p "$O"
R=$1; shift
Q="--no-cpu-baseline --no-dp1 --no-extras --no-roofline --no-calibration"
run() { (cd $1 && env $2 python bench.py $Q "${@:3}" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"); }
for i in $(seq $R); do
  echo "round $i [old_tree] $(run "$ROOT/tools/ab/old_tree" "X=1" "$@") ms" | tee -a "$O/r06_ab_tree.txt"
  echo "round $i [repo] $(run "$ROOT" "X=1" "$@") ms" | tee -a "$O/r06_ab_tree.txt"
  for e in ${NEW_ENV:-}; do
    echo "round $i [repo $e] $(run "$ROOT" "$e" "$@") ms" | tee -a "$O/r06_ab_tree.txt"
  done
done
