#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_kernels.py -q -m gpu -k "attention" 2>&1 | tail -2 > gpurun_out/attn_tests.txt
for fk in 0 1 0 1; do
  ADP_ATTN_FEWKEYS=$fk python bench.py --no-cpu-baseline --no-dp1 --no-roofline --no-calibration 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
def find(o):
    if isinstance(o,dict):
        for k,v in o.items():
            if k=='config4': return v
            r=find(v)
            if r: return r
c=find(d)
print('fewkeys=$fk headline',d['ms_per_step'],'config4',c['ms_per_step'],{k:(v.get('avg_us_per_call')) for k,v in c['attention_kernels'].items() if k.startswith('adp_')})
" >> gpurun_out/attn_tests.txt
done
cat gpurun_out/attn_tests.txt
