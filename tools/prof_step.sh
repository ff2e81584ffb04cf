#!/bin/bash
# rocprofv3 kernel statistics of 8 eager bench steps (run on the GPU box): tools/prof_step.sh <tag> [bench args]
TAG=${1:-x}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python $ROOT/bench.py --steps 7 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline --no-extras "$@" > $O/prof.log 2>&1
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_8steps.csv
rm -rf $O/prof
