#!/bin/bash
# One probe of tools/dev_dilate.hip (dev build): the reference's loop, static / mesh turning / every pose twice, with the work list made
# from dilated costs.  -> gpurun_out/r05/tri_dilate.log
mkdir -p gpurun_out/r05
export RT355_LIB=tools/bin/librt355_dev.so RT355_TRI_GX=168
run() { echo "## $*" | tee -a gpurun_out/r05/tri_dilate.log; timeout -k 5 60 python -u tools/loop_breakdown.py 2>&1 | grep --line-buffered -v amdgpu | tee -a gpurun_out/r05/tri_dilate.log; }
RT355_TRI_DILATE=8 run "neighbours x 1" || exit 1
RT355_TRI_DILATE=5 run "neighbours x 0.625" || exit 1
run "no dilation" || exit 1
