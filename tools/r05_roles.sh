#!/bin/bash
# A/B of trace_roles (a split tile's idle lanes walk the next reflection ray while the owners walk the shadow ray) against the tile
# kernel, awaited frames: dev build, the three triangle configurations; frame hashes must agree.  -> gpurun_out/r05/tri_roles.log
mkdir -p gpurun_out/r05
export RT355_LIB=tools/bin/librt355_dev.so
run() { timeout -k 10 300 python -u tools/tri_ab_probe.py v0 "$@" 2>&1 | grep --line-buffered -v amdgpu | tee -a gpurun_out/r05/tri_roles.log; }
echo "# $1" >> gpurun_out/r05/tri_roles.log
run REF TRI "roles" || exit 1
RT355_TRI_HALVES=1 run REF TRI "roles, halves for quarters" || exit 1
RT355_TRI_HALVES=1 RT355_TRI_CM2=12 run REF TRI "roles, halves for quarters" || exit 1
RT355_TRI_HALVES=1 RT355_TRI_CM2=20 run REF TRI "roles, halves for quarters" || exit 1
RT355_TRI_HALVES=1 RT355_TRI_MULT16=2 run REF TRI "roles, halves for quarters" || exit 1
RT355_TRI_HALVES=1 RT355_TRI_CAP16=128 run REF TRI "roles, halves for quarters" || exit 1
