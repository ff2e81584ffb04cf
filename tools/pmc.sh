#!/bin/bash
# Per-kernel hardware counters of the bench step from rocprofv3 PMC passes (run on the GPU box through gpurun):
#   FETCH_SIZE, WRITE_SIZE                      -> HBM-side bytes per launch           (profiles/pmc_traffic.json: hbm_bytes_per_launch)
#   SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE  -> matrix-core busy fraction per kernel (same file: mfma_busy, clock_ghz)
# Separate passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"), each
# with --kernel-trace only -- never combined with the hip/hsa/memcopy trace domains.
# usage: tools/pmc.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" GRBM_GUI_ACTIVE; do
  N=$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$N" -o pmc -- \
    python "$ROOT/bench.py" --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline --no-extras "$@" > "$OUT/$N.log" 2>&1
  echo "rc=$?" >> "$OUT/$N.log"
  f=$(find "$OUT/$N" -name '*counter_collection.csv' | head -1)
  t=$(find "$OUT/$N" -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && head -4 "$f" > "$OUT/$N.head.txt"
  [ -n "$f" ] && python "$ROOT/tools/pmc_summary.py" --reduce "$f" "$OUT/$N.summary.csv" "$t"
  rm -rf "$OUT/$N"   # the per-dispatch rows are large; the per-kernel reduction is what is kept
done
python "$ROOT/tools/pmc_summary.py" --merge "$OUT/FETCH_SIZE.summary.csv" "$OUT/WRITE_SIZE.summary.csv" "$OUT/pmc_traffic.json" \
  "$OUT/SQ_VALU_MFMA_BUSY_CYCLES.summary.csv" "$OUT/GRBM_GUI_ACTIVE.summary.csv"
head -c 1500 "$OUT/pmc_traffic.json"
