#!/bin/bash
# Development builds of the library (build container): tools/bin/librt355_dev.so reads its launch knobs from the environment
# (RT355_TRI_*, RT355_BVH_TAIL / _BLOCKS ...).  It does not ship; tests and bench.py load compute_raytracer_amd/librt355.so.
set -e
cd "$(dirname "$0")/.."
CS=compute_raytracer_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-function -ffp-contract=off"
mkdir -p tools/bin /tmp/rtdev
make -s lib
/opt/rocm/bin/hipcc $FL -DRT355_DEV_EXPORTS -DRT355_BUILD_ID='"dev"' -c $CS/rt_api.hip -o /tmp/rtdev/rt_api.o &
/opt/rocm/bin/hipcc $FL -fno-slp-vectorize -DRT_BVH_DEV_ENV -c $CS/rt_bvh.hip -o /tmp/rtdev/rt_bvh.o &
/opt/rocm/bin/hipcc $FL -fno-slp-vectorize -DRT_TRI_DEV_ENV -c $CS/rt_triangles.hip -o /tmp/rtdev/rt_triangles.o &
wait
# tools/dev_dilate.hip (an experiment on the work list's costs) stands in front of the library's rt_launch_order_hist: that one is renamed
/opt/rocm/bin/hipcc $FL -c tools/dev_dilate.hip -o /tmp/rtdev/dev_dilate.o
/opt/rocm/lib/llvm/bin/llvm-objcopy --redefine-sym _Z20rt_launch_order_histPjS_S_jjPyS0_jS0_P12ihipStream_t=_Z25rt_launch_order_hist_origPjS_S_jjPyS0_jS0_P12ihipStream_t /tmp/rtdev/rt_triangles.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/bin/librt355_dev.so /tmp/rtdev/rt_api.o /tmp/rtdev/rt_bvh.o \
    $CS/rt_kernels.o /tmp/rtdev/rt_triangles.o /tmp/rtdev/dev_dilate.o $CS/rt_assemble.o $CS/rt_comm.o -L/opt/rocm/lib -lrccl
ls -la tools/bin/*.so
