#!/bin/bash
# Everything the round's evidence consists of, in one gpurun call: tools/final_profiles.sh <tag>
#   PMC passes (HBM traffic, matrix-core busy) FIRST, so that the bench line of the same session reads counters of this very code |
#   pytest -m gpu | bench line + per-shape table | rocprofv3 kernel stats of the bench step, the batch-1 step and the sampler |
#   kernel-work microbenches behind DESIGN.md's round-5 measurements
set -u
TAG=${1:-r05z}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p "$O"
cd "$ROOT"
QUIET="--no-cpu-baseline --no-roofline --no-extras --no-calibration --no-dp1"
bash tools/pmc.sh "$TAG" --no-calibration --no-dp1 > "$O/${TAG}_pmc.log" 2>&1
[ -f "$O/pmc_$TAG/pmc_traffic.json" ] && cp "$O/pmc_$TAG/pmc_traffic.json" "$ROOT/profiles/pmc_traffic.json"
python -m pytest tests -m gpu -x -q > "$O/${TAG}_gputests.txt" 2>&1; tail -3 "$O/${TAG}_gputests.txt"
ADP_BENCH_DETAIL=$O/${TAG}_bench_b4_per_shape.txt python bench.py > "$O/${TAG}_bench_n1.json" 2> "$O/${TAG}_bench.err"
python tools/detail.py "$O/${TAG}_per_shape_b1_fwd.txt" --batch 1 --fwd > /dev/null 2>&1
python tools/detail.py "$O/${TAG}_per_shape_b1.txt" --batch 1 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_b4" -o trace -- python "$ROOT/bench.py" --steps 3 --warmup 1 --graph 0 $QUIET > "$O/prof_b4.log" 2>&1
find "$O/prof_b4" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$O/${TAG}_bench_b4_kernel_stats.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_b1" -o trace -- python "$ROOT/bench.py" --batch 1 --steps 3 --warmup 1 --graph 0 $QUIET > "$O/prof_b1.log" 2>&1
find "$O/prof_b1" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$O/${TAG}_batch1_kernel_stats.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_s" -o trace -- python "$ROOT/tools/sample_bench.py" --steps 10 --graph 0 > "$O/prof_s.log" 2>&1
find "$O/prof_s" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$O/${TAG}_sampler_b1_kernel_stats.csv"
ADP_CFG_PROF_REPLAY=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c4" -o trace -- python "$ROOT/tools/cfg_prof.py" config4 3 > "$O/prof_c4.log" 2>&1
find "$O/prof_c4" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$O/${TAG}_config4_kernel_stats.csv"
rm -rf "$O/prof_b4" "$O/prof_b1" "$O/prof_s" "$O/prof_c4"
cd "$ROOT"
if [ "${ADP_FINAL_MICRO:-0}" = "1" ]; then  # kernel-work microbenches (unchanged code: taken once per round)
(timeout 120 python tools/mm4_micro.py 4) 2>&1 | grep -v "Warn\|amdgpu.ids" > "$O/${TAG}_mm4_micro.txt"
(timeout 200 python tools/dp_capture_probe.py thread_local 4; timeout 200 python tools/dp_capture_probe.py thread_local 1) 2>&1 | grep "^\[" > "$O/${TAG}_dp_capture_probe.txt"
fi
ls -la "$O" | grep "$TAG"
