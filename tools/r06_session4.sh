#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p "$O"; cd "$ROOT"
python -m pytest tests/test_kernels.py -m gpu -x -q -k "gn_silu_bwd" 2>&1 | tail -3
python -m pytest tests -m gpu -x -q > "$O/r06e_gputests.txt" 2>&1; tail -3 "$O/r06e_gputests.txt"
bash tools/r06_ab_env.sh 2 "ADP_GN_BWD_SLAB=0" "ADP_GN_BWD_SLAB=1"
