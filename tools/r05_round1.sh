# round 5, first measurement pass (GPU box): suite, root-of-eight emulation, Node host with / without the hardware-queue default, driver-style bench lines
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests -m gpu -x -q -rs > gpurun_out/r05/full2.log 2>&1; tail -4 gpurun_out/r05/full2.log
timeout -k 10 300 python tools/root_probe.py C3 8 2>&1 | grep -v amdgpu | tee gpurun_out/r05/root_probe.log
for e in "" "RT355_KEEP_HW_QUEUES=1"; do
  env -u GPU_MAX_HW_QUEUES $e timeout -k 10 200 node node/bench-frames.js C3 100 2>&1 | tail -1 | tee -a gpurun_out/r05/node_c3.log
done
for i in 1 2; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_C3_s20_$i.json 2> gpurun_out/r05/bench_C3_s20_$i.err; python -c "
import json,sys; d=json.load(open('gpurun_out/r05/bench_C3_s20_$i.json')); print({k:d.get(k) for k in ('ms_per_step','ms_per_step_median','ms_per_step_min','serial_ms_per_step','serial_ms_per_step_median','node_loop_ms_per_step','node_inflight_ms_per_step','loop_ms_per_step','clocks')}); print(d.get('node'))"
done
timeout -k 10 300 python bench.py --config REF --steps 20 --warmup 5 > gpurun_out/r05/bench_REF_s20.json 2> gpurun_out/r05/bench_REF_s20.err; python -c "
import json,sys; d=json.load(open('gpurun_out/r05/bench_REF_s20.json')); print({k:d.get(k) for k in ('ms_per_step','ms_per_step_median','ms_per_step_min','serial_ms_per_step','serial_ms_per_step_median','node_loop_ms_per_step','node_inflight_ms_per_step','loop_ms_per_step','animated_ms_per_step')}); print(d.get('node')); print(d.get('frame_check'))"
