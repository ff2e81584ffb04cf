#!/bin/bash
# Copies one tools/final_profiles.sh result set out of gpurun_out/ (scratch) into profiles/ (tracked): tools/collect_profiles.sh <tag>
T=${1:-r05z}
G=gpurun_out; P=profiles
for f in gputests.txt bench_n1.json bench_b4_per_shape.txt bench_b4_kernel_stats.csv batch1_kernel_stats.csv sampler_b1_kernel_stats.csv config4_kernel_stats.csv per_shape_b1.txt per_shape_b1_fwd.txt mm4_micro.txt dp_capture_probe.txt; do
  [ -f $G/${T}_$f ] && cp $G/${T}_$f $P/${T}_$f
done
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do [ -f $G/pmc_$T/$c.summary.csv ] && cp $G/pmc_$T/$c.summary.csv $P/${T}_pmc_$c.csv; done
[ -f $G/pmc_$T/pmc_traffic.json ] && cp $G/pmc_$T/pmc_traffic.json $P/pmc_traffic.json
ls $P | grep $T
