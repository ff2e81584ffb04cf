#!/bin/bash
# Copies one tools/final_profiles.sh result set out of gpurun_out/ (scratch) into profiles/ (tracked) under the r04z names
# profiles/README.md lists: tools/collect_profiles.sh <tag>
T=${1:-r04z}
G=gpurun_out; P=profiles
for f in gputests.txt bench_n1.json bench_b4_per_shape.txt bench_b4_kernel_stats.csv batch1_kernel_stats.csv sampler_b1_kernel_stats.csv per_shape_b1.txt per_shape_b1_fwd.txt; do
  cp $G/${T}_$f $P/r04z_$f
done
cp $G/${T}_sampler_graph_kernel_stats.csv $P/r04_sampler_graph_kernel_stats.csv
for f in tile_probe alu_probe launch_probe small_bench tile_bench; do grep -v "amdgpu.ids" $G/${T}_$f.txt > $P/r04_$f.txt; done
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do cp $G/pmc_$T/$c.summary.csv $P/r04z_pmc_$c.csv; done
cp $G/pmc_$T/pmc_traffic.json $P/pmc_traffic.json
[ -f $G/r04a/kernel_stats_8steps.csv ] && cp $G/r04a/kernel_stats_8steps.csv $P/r04a_kernel_stats_8steps.csv
ls $P | grep r04
