#!/bin/bash
# HBM traffic per kernel from the rocprofv3 PMC counters (run on the GPU box through gpurun).
# Two separate passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC
# slots"), each with --kernel-trace only -- never combined with the hip/hsa/memcopy trace domains.
# usage: tools/pmc.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -o pmc -- \
    python "$ROOT/bench.py" --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline --no-extras "$@" > "$OUT/$C.log" 2>&1
  echo "rc=$?" >> "$OUT/$C.log"
  f=$(find "$OUT/$C" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && head -4 "$f" > "$OUT/$C.head.txt"; [ -n "$f" ] && python "$ROOT/tools/pmc_summary.py" --reduce "$f" "$OUT/$C.summary.csv"
  rm -rf "$OUT/$C"   # the per-dispatch rows are large; the per-kernel reduction is what is kept
done
python "$ROOT/tools/pmc_summary.py" --merge "$OUT/FETCH_SIZE.summary.csv" "$OUT/WRITE_SIZE.summary.csv" "$OUT/pmc_traffic.json"
head -c 1500 "$OUT/pmc_traffic.json"
