#!/bin/bash
# Everything the round's evidence consists of, in one gpurun call: tools/final_profiles.sh <tag>
#   pytest -m gpu | bench line + per-shape table | rocprofv3 kernel stats of the bench step, the batch-1 step and the sampler |
#   PMC passes (HBM traffic, matrix-core busy)
set -u
TAG=${1:-r04z}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p "$O"
cd "$ROOT"
python -m pytest tests -m gpu -x -q > "$O/${TAG}_gputests.txt" 2>&1; tail -3 "$O/${TAG}_gputests.txt"
ADP_BENCH_DETAIL=$O/${TAG}_bench_b4_per_shape.txt python bench.py > "$O/${TAG}_bench_n1.json" 2> "$O/${TAG}_bench.err"
python tools/detail.py "$O/${TAG}_per_shape_b1_fwd.txt" --batch 1 --fwd > /dev/null 2>&1
python tools/detail.py "$O/${TAG}_per_shape_b1.txt" --batch 1 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_b4" -o trace -- python "$ROOT/bench.py" --steps 3 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline --no-extras > "$O/prof_b4.log" 2>&1
find "$O/prof_b4" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$O/${TAG}_bench_b4_kernel_stats.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_b1" -o trace -- python "$ROOT/bench.py" --batch 1 --steps 3 --warmup 1 --graph 0 --no-cpu-baseline --no-roofline --no-extras > "$O/prof_b1.log" 2>&1
find "$O/prof_b1" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$O/${TAG}_batch1_kernel_stats.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_s" -o trace -- python "$ROOT/tools/sample_bench.py" --steps 10 --graph 0 > "$O/prof_s.log" 2>&1
find "$O/prof_s" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$O/${TAG}_sampler_b1_kernel_stats.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_sg" -o trace -- python "$ROOT/tools/sample_bench.py" --steps 10 --graph 1 > "$O/prof_sg.log" 2>&1
find "$O/prof_sg" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$O/${TAG}_sampler_graph_kernel_stats.csv"
rm -rf "$O/prof_b4" "$O/prof_b1" "$O/prof_s" "$O/prof_sg"
# probes and microbenches behind DESIGN.md's round-4 measurements
cd "$ROOT"
(timeout 60 tools/probe/tile_probe 0 1 0 1; timeout 60 tools/probe/tile_probe 0 1 1 0; timeout 60 tools/probe/tile_probe 1 0 0 0) > "$O/${TAG}_tile_probe.txt" 2>&1
timeout 60 tools/probe/alu_probe > "$O/${TAG}_alu_probe.txt" 2>&1
timeout 60 tools/probe/launch_probe > "$O/${TAG}_launch_probe.txt" 2>&1
(timeout 300 python tools/small_bench.py 4; timeout 300 python tools/small_bench.py 1) > "$O/${TAG}_small_bench.txt" 2>&1
(timeout 300 python tools/tile_bench.py; TILE_C=8 TILE_L=262144 timeout 300 python tools/tile_bench.py) > "$O/${TAG}_tile_bench.txt" 2>&1
cd "$ROOT"
bash tools/pmc.sh "$TAG" > "$O/${TAG}_pmc.log" 2>&1
ls -la "$O" | grep "$TAG"
