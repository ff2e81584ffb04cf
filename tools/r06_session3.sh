#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p "$O"
cd "$ROOT"
for tag in ""; do
  (KTRACE_TAG=$tag timeout 300 python tools/ktrace.py run 1024 256 4) > "$O/r06c_ktrace_d7$tag.txt" 2>&1
  echo "=== ktrace tag '$tag'"; grep -v "Warn\|amdgpu.ids" "$O/r06c_ktrace_d7$tag.txt" | grep "traced\|over 64\|lifetime\|block 0\|per chunk" | head -24
done
python -m pytest tests/test_train_graph.py -m gpu -x -q > "$O/r06c_train_graph_tests.txt" 2>&1; tail -3 "$O/r06c_train_graph_tests.txt"
python -m pytest tests/test_kernels.py -m gpu -x -q -k "mm4 or wgrad or dynamic_range" > "$O/r06c_kernel_tests.txt" 2>&1; tail -3 "$O/r06c_kernel_tests.txt"
python bench.py --no-cpu-baseline --no-dp1 --no-extras > "$O/r06c_bench_n1.json" 2> "$O/r06c_bench.err"; tail -2 "$O/r06c_bench.err"
python - <<'PY'
import json,os
p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","r06c_bench_n1.json")
try:
    l=json.loads(open(p).read().strip().splitlines()[-1])
    print("value",l["value"],"ms",l["ms_per_step"],"windows",l.get("ms_per_step_windows"))
    for k,v in l.get("kernels",{}).items(): print("  %-90s n=%3d avg %7.1f us total %.3f ms frac %s exec %s"%(k[:90],v["launches"],v["avg_us"],v["total_ms"],v["frac"],v.get("frac_executed")))
except Exception as e: print("bench parse failed",e)
PY
