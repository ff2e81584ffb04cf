#!/bin/bash
# round 6, GPU session 2: in-kernel timelines of the F(4,3) kernels, README-loop leg after the AccumulateGrad-stream fix, probe sweeps
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p "$O"
cd "$ROOT"
(timeout 300 python tools/ktrace.py run 1024 256 4) > "$O/r06b_ktrace_d7.txt" 2>&1; cat "$O/r06b_ktrace_d7.txt" | grep -v Warn
(timeout 300 python tools/ktrace.py run 512 1024 4) > "$O/r06b_ktrace_d5.txt" 2>&1; grep -v Warn "$O/r06b_ktrace_d5.txt" | head -60
python -m pytest tests/test_train_graph.py -m gpu -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-dp1 > "$O/r06b_bench_n1.json" 2> "$O/r06b_bench.err"; tail -3 "$O/r06b_bench.err"
python - <<'PY'
import json,os
p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","r06b_bench_n1.json")
try:
    l=json.loads(open(p).read().strip().splitlines()[-1])
    print("value",l["value"],"ms",l["ms_per_step"],"windows",l.get("ms_per_step_windows"))
    print("eager_api",json.dumps(l.get("eager_api")))
    c=l.get("calibration",{})
    print("copy",c.get("copy_256MB_gbps"),c.get("copy_1GiB_gbps_by_variant"),c.get("copy_1GiB_best_variant"))
    print("mfma",c.get("mfma_f32_probe_tflops"),c.get("mfma_f32_probe_tflops_by_variant"))
    print("hbm",json.dumps({k:v for k,v in l.get("roofline_hbm_convblock",{}).items() if k in ("frac","replay_frac","frac_of_copy_ceiling","replay_frac_of_copy_ceiling","avg_us","replay_avg_us")}))
except Exception as e: print("bench parse failed",e)
PY
