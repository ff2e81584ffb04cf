#!/bin/bash
# SQ counter passes on one conv shape: tools/pmc_probe.sh <tag> <fwd|dgrad|wgrad> B C L
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/probe_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$ROOT/tools/mm_probe.py" "$@" > "$OUT/timing.txt" 2>&1
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/p$i" -o pmc -- python "$ROOT/tools/mm_probe.py" "$@" 3 > "$OUT/p$i.log" 2>&1
  f=$(find "$OUT/p$i" -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" >> "$OUT/counters.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "conv_mm" in k or "wgrad_mm" in k or "conv_stream" in k:
        acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:34s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
  else
    tail -5 "$OUT/p$i.log" >> "$OUT/counters.txt"
  fi
  rm -rf "$OUT/p$i"
done
cat "$OUT/timing.txt" "$OUT/counters.txt"
