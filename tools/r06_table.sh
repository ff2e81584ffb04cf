#!/bin/bash
# kernel table of the headline step under an environment: tools/r06_table.sh "<ENV=.. ...>"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p "$O"; cd "$ROOT"
env $1 ADP_BENCH_DETAIL=$O/r06_detail.txt python bench.py --no-cpu-baseline --no-dp1 --no-extras --no-calibration 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',l['value'],'ms',l['ms_per_step'],'instrumented',l.get('instrumented_kernel_ms_per_step'))
for k,v in l.get('kernels',{}).items(): print('  %-80s n=%3d avg %7.1f us total %.3f ms exec %s'%(k[:80],v['launches'],v['avg_us'],v['total_ms'],v.get('frac_executed')))
"
grep "wgrad_mm" $O/r06_detail.txt | head -12
