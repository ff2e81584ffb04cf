#!/bin/bash
# A/B of two builds of libadp_hip.so on ONE box: tools/ab_lib.sh <rounds> [bench.py args]
#   expects audio_diffusion_pytorch_amd/lib_old.so and lib_new.so (copies of the two builds; git-ignored, they travel with gpurun)
R=${1:-2}; shift
P=audio_diffusion_pytorch_amd
for i in $(seq $R); do
  for v in old new; do
    cp $P/lib_$v.so $P/libadp_hip.so
    python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', j['ms_per_step'], [(k, v.get('ms_per_step')) for k, v in j.items() if isinstance(v, dict) and 'ms_per_step' in v])"
  done
done
cp $P/lib_new.so $P/libadp_hip.so
