#!/bin/bash
# interleaved A/B of environment configurations over every bench leg: tools/r06_ab_legs.sh <rounds> "<ENV=..>" "<ENV=..>" ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p "$O"; cd "$ROOT"
R=$1; shift
for i in $(seq $R); do
  for cfg in "$@"; do
    env $cfg python bench.py --no-cpu-baseline --no-dp1 --no-roofline --no-calibration 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $i [$cfg] headline',d['ms_per_step'],' '.join('%s %s'%(k,d[k].get('ms_per_step')) for k in ('batch1','sampler','readme_attention','config4') if k in d))" | tee -a "$O/r06_ab_legs.txt"
  done
done
