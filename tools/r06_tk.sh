#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/tk.txt
for warm in 0 1; do
  echo "== TILEK_WARM=$warm" >> gpurun_out/tk.txt
  TILEK_WARM=$warm python tools/tilek_micro.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/tk.txt
done
cat gpurun_out/tk.txt
