mkdir -p gpurun_out/r05/reftrace2
timeout -k 10 600 python -m pytest tests/test_triangles_gpu.py tests/test_alternating_gpu.py tests/test_ref_pin.py tests/test_host_paths_gpu.py -m gpu -x -q > gpurun_out/r05/t5.log 2>&1; tail -3 gpurun_out/r05/t5.log
timeout -k 10 200 python tools/tri_ab_probe.py v0 "v5 epilogue in order_hist" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r05/tri_v5.log
export TMPDIR=/tmp RT355_BENCH_NO_CHILDREN=1
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05/reftrace2 -o t -- python3 tools/tri_frames.py REF 16 0 > gpurun_out/r05/reftrace2/out.txt 2>&1
