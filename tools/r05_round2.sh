mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_triangles_gpu.py tests/test_alternating_gpu.py tests/test_ref_pin.py tests/test_c_abi_gpu.py tests/test_host_paths_gpu.py tests/test_bench_gpu.py tests/test_node_host.py -m gpu -x -q -rs > gpurun_out/r05/t4.log 2>&1; tail -4 gpurun_out/r05/t4.log
for cfg in C3 C5; do
export KNOB_CONFIG=$cfg KNOB_BATCH=$([ $cfg = C5 ] && echo 8 || echo 64) RT355_LIB=tools/bin/librt355_dev.so
KNOB_WORLD=8 timeout -k 10 120 python tools/knob_ab.py "$cfg rank 0 of 8, render only" 2>&1 | grep "in flight" | tee -a gpurun_out/r05/root_probe.log
RT355_DEV_ROOT_WORLD=8 KNOB_WORLD=8 timeout -k 10 120 python tools/knob_ab.py "$cfg rank 0 of 8 + de-interleave of the whole frame (the root)" 2>&1 | grep "in flight" | tee -a gpurun_out/r05/root_probe.log
KNOB_WORLD=8 timeout -k 10 120 python tools/knob_ab.py serial "$cfg rank 0 of 8, render only" 2>&1 | grep "serial" | tee -a gpurun_out/r05/root_probe.log
RT355_DEV_ROOT_WORLD=8 KNOB_WORLD=8 timeout -k 10 120 python tools/knob_ab.py serial "$cfg rank 0 of 8 + de-interleave (the root)" 2>&1 | grep "serial" | tee -a gpurun_out/r05/root_probe.log
KNOB_WORLD=1 timeout -k 10 120 python tools/knob_ab.py "$cfg whole frame" 2>&1 | grep "in flight" | tee -a gpurun_out/r05/root_probe.log
done
