#!/bin/bash
# interleaved A/B of environment-switched kernel variants on ONE box: tools/r06_ab_env.sh <rounds> "<ENV=.. ENV=..>" "<...>" ...
# (first configuration "" = defaults).  Prints the headline ms_per_step per configuration and round.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p "$O"; cd "$ROOT"
R=$1; shift
Q="--no-cpu-baseline --no-dp1 --no-extras --no-roofline --no-calibration"
for i in $(seq $R); do
  for cfg in "$@"; do
    ms=$(env $cfg python bench.py $Q ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $i [$cfg] $ms ms" | tee -a "$O/r06_ab_env.txt"
  done
done
