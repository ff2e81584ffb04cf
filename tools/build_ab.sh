#!/bin/bash
# tools/build_ab.sh <git-rev>: builds libadp_hip.so of <git-rev> into tools/ab/lib_old.so and copies the working tree's
# library to tools/ab/lib_new.so, for tools/ab_lib.py (interleaved A/B of two builds on one GPU box).
set -e
cd /root/repo
REV=${1:-HEAD}
TMP=$(mktemp -d)
git archive "$REV" audio_diffusion_pytorch_amd include | tar -x -C "$TMP"
(cd "$TMP" && python audio_diffusion_pytorch_amd/build.py > /tmp/adp_build_old.log 2>&1) || { tail -20 /tmp/adp_build_old.log; exit 1; }
mkdir -p tools/ab
cp "$TMP/audio_diffusion_pytorch_amd/libadp_hip.so" tools/ab/lib_old.so
python audio_diffusion_pytorch_amd/build.py > /tmp/adp_build.log 2>&1
cp audio_diffusion_pytorch_amd/libadp_hip.so tools/ab/lib_new.so
rm -rf "$TMP"
ls -la tools/ab
