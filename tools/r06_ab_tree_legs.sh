#!/bin/bash
# same-box comparison of every bench leg between tools/ab/old_tree and this tree: tools/r06_ab_tree_legs.sh <rounds>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p "$O"
for i in $(seq ${1:-1}); do
  for t in "$ROOT/tools/ab/old_tree" "$ROOT"; do
    (cd $t && python bench.py --no-cpu-baseline --no-dp1 --no-roofline --no-calibration 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $i [$(basename $t)] headline',d['ms_per_step'],' '.join('%s %s'%(k,d[k].get('ms_per_step')) for k in ('batch1','sampler','readme_attention','config4') if k in d), 'eager_api', d.get('eager_api',{}).get('batch4',{}).get('ms_per_step'))") | tee -a "$O/r06_ab_tree_legs.txt"
  done
done
