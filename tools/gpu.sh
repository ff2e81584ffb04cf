#!/bin/bash
# usage: tools/gpu.sh <timeout_s> '<command>'   -- rebuilds libadp_hip.so first so the snapshot is never stale
set -e
cd /root/repo
python audio_diffusion_pytorch_amd/build.py > /tmp/adp_build.log 2>&1 || { tail -30 /tmp/adp_build.log; exit 1; }
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
