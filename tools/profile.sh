#!/bin/bash
# rocprofv3 kernel-trace summary of the bench step (run on the GPU box through gpurun).
# usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- \
  python "$ROOT/bench.py" --steps 3 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline --no-extras "$@" > "$OUT/bench.log" 2>&1
echo "rc=$?" >> "$OUT/bench.log"
find "$OUT" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
# the full per-dispatch trace is large; keep only the stats
find "$OUT" -name '*kernel_trace.csv' -size +4M -delete
ls -la "$OUT" | head
head -40 "$OUT/kernel_stats.csv"
