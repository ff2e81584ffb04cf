#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p "$O"; cd "$ROOT"
(KTRACE_TAG=${1:-} timeout 300 python tools/ktrace.py run 1024 256 4) > "$O/r06d_ktrace_d7${1:-}.txt" 2>&1
grep -v "Warn\|amdgpu.ids" "$O/r06d_ktrace_d7${1:-}.txt" | grep "traced\|over 64\|lifetime\|arrival\|block 0 \|Error\|error" | cut -c1-300
