#!/bin/bash
# development: the walk's suspension threshold (RT355_BVH_TAIL, dev library) for C3 in flight / one at a time / rank of eight, and C5
mkdir -p gpurun_out/r03; L=gpurun_out/r03/tail_sweep.log; export RT355_LIB=tools/bin/librt355_dev.so
for t in 16 20 24 28; do RT355_BVH_TAIL=$t python tools/knob_ab.py >> $L 2>&1; done
for t in 8 12 16 20; do RT355_BVH_TAIL=$t python tools/knob_ab.py serial >> $L 2>&1; done
for t in 12 16 20; do KNOB_WORLD=8 RT355_BVH_TAIL=$t python tools/knob_ab.py >> $L 2>&1; done
for t in 24 32 40; do KNOB_CONFIG=C5 KNOB_BATCH=8 RT355_BVH_TAIL=$t python tools/knob_ab.py >> $L 2>&1; done
grep -v amdgpu.ids $L
