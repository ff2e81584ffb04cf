#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/tk.txt
cp audio_diffusion_pytorch_amd/libadp_hip.so /tmp/keep.so
for lib in keep "$@"; do
  [ $lib = keep ] && cp /tmp/keep.so audio_diffusion_pytorch_amd/libadp_hip.so || cp tools/ab/$lib audio_diffusion_pytorch_amd/libadp_hip.so
  echo "== $lib" >> gpurun_out/tk.txt
  python tools/tilek_micro.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/tk.txt
done
cp /tmp/keep.so audio_diffusion_pytorch_amd/libadp_hip.so
cat gpurun_out/tk.txt
