#!/bin/bash
# The reference's loop (instances turning, every frame awaited) with and without trace_roles, and with other cost factors for its parts.
mkdir -p gpurun_out/r05
export RT355_LIB=tools/bin/librt355_dev.so
run() { echo "## $*" | tee -a gpurun_out/r05/tri_roles_loop.log; timeout -k 10 200 python -u tools/loop_breakdown.py 2>&1 | grep --line-buffered -v amdgpu | tee -a gpurun_out/r05/tri_roles_loop.log; }
RT355_TRI_CM4=16 RT355_TRI_CM16=32 RT355_TRI_CAP4=2048 RT355_TRI_DIV4=8 run "roles CM 16 / 32, up to 2048 tiles in parts (one in 8)" || exit 1
RT355_TRI_CM4=16 RT355_TRI_CM16=32 RT355_TRI_CAP4=4096 RT355_TRI_DIV4=4 run "roles CM 16 / 32, up to 4096 tiles in parts (one in 4)" || exit 1
RT355_TRI_CM4=20 RT355_TRI_CM16=40 RT355_TRI_CAP4=2048 RT355_TRI_DIV4=8 run "roles CM 20 / 40, up to 2048 (one in 8)" || exit 1
RT355_TRI_CM4=20 RT355_TRI_CM16=40 run "roles CM 20 / 40" || exit 1
RT355_TRI_ROLES=0 RT355_TRI_CAP4=2048 RT355_TRI_DIV4=8 run "ROLES=0, up to 2048 (one in 8)" || exit 1
