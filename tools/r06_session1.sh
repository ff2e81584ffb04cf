#!/bin/bash
# round 6, GPU session 1: new tests, bench line (with the eager_api leg), graph-branch concurrency probe, timeline of the captured
# data-parallel step on a one-rank RCCL group
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p "$O"
cd "$ROOT"
python -m pytest tests/test_train_graph.py -m gpu -x -q > "$O/r06a_train_graph_tests.txt" 2>&1; tail -15 "$O/r06a_train_graph_tests.txt"
python -m pytest tests -m gpu -x -q > "$O/r06a_gputests.txt" 2>&1; tail -3 "$O/r06a_gputests.txt"
(timeout 120 python tools/graph_branch_probe.py) 2>&1 | grep "^\[" > "$O/r06a_graph_branch_probe.txt"; cat "$O/r06a_graph_branch_probe.txt"
python bench.py > "$O/r06a_bench_n1.json" 2> "$O/r06a_bench.err"; tail -3 "$O/r06a_bench.err"
python - <<'PY'
import json,os
p=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","r06a_bench_n1.json")
try:
    l=json.loads(open(p).read().strip().splitlines()[-1])
    print("value",l["value"],"ms",l["ms_per_step"],"windows",l.get("ms_per_step_windows"))
    print("eager_api",json.dumps(l.get("eager_api")))
    print("dp1",json.dumps({k:(v.get("hipgraph_replay") if isinstance(v,dict) else v) for k,v in l.get("dp1",{}).items() if k!="what"}))
    print("clocks",json.dumps(l.get("calibration",{}).get("clocks_under_load")))
    print("batch1",l.get("batch1"),"sampler",l.get("sampler",{}).get("ms_per_step"))
except Exception as e: print("bench parse failed",e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O/prof_dp" -o trace -- python "$ROOT/tools/dp_capture_probe.py" thread_local 4 > "$O/r06a_dp_trace.log" 2>&1
T=$(find "$O/prof_dp" -name '*kernel_trace.csv' | head -1)
python "$ROOT/tools/timeline_overlap.py" "$T" --last-fraction 0.35 --out "$O/r06a_dp_timeline.txt"; cat "$O/r06a_dp_timeline.txt"
grep "^\[" "$O/r06a_dp_trace.log"
# keep a compact excerpt of the trace (one replayed step) for the record
python - "$T" "$O/r06a_dp_trace_excerpt.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
tail=rows[-1500:]
with open(sys.argv[2],"w") as f:
    w=csv.writer(f); w.writerow(["start_ns","end_ns","queue","stream","kernel"])
    for r in tail: w.writerow([r["Start_Timestamp"],r["End_Timestamp"],r.get("Queue_Id",""),r.get("Stream_Id",""),r["Kernel_Name"][:70]])
PY
rm -rf "$O/prof_dp"
ls -la "$O" | grep r06a
