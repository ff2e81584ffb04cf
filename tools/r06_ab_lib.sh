#!/bin/bash
# interleaved A/B of library builds on ONE box: tools/r06_ab_lib.sh <rounds> <lib1> <lib2> ...  (paths under tools/ab/)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p "$O"; cd "$ROOT"
R=$1; shift
Q="--no-cpu-baseline --no-dp1 --no-extras --no-roofline --no-calibration"
cp audio_diffusion_pytorch_amd/libadp_hip.so /tmp/lib_keep.so
for i in $(seq $R); do
  for lib in "$@"; do
    cp tools/ab/$lib audio_diffusion_pytorch_amd/libadp_hip.so
    ms=$(python bench.py $Q 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "round $i [$lib] $ms ms" | tee -a "$O/r06_ab_lib.txt"
  done
done
cp /tmp/lib_keep.so audio_diffusion_pytorch_amd/libadp_hip.so
