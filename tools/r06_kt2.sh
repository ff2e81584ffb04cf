#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p "$O"; cd "$ROOT"
(KTRACE_TAG=_st8 timeout 300 python tools/ktrace.py run 512 1024 4) > "$O/r06f_ktrace_d5_st8.txt" 2>&1
grep -v "Warn\|amdgpu.ids" "$O/r06f_ktrace_d5_st8.txt" | grep "traced\|over 64\|lifetime\|start of\|end of\|rror" | cut -c1-700
