# round 5: how many of an awaited frame's longest tiles go as sixteenths (cap16) and from which cost on (mult16 / 2 throughput times)
mkdir -p gpurun_out/r05
export RT355_LIB=tools/bin/librt355_dev.so
for c16 in 64 128 256; do for m16 in 2 3 4; do
  RT355_TRI_CAP16=$c16 RT355_TRI_MULT16=$m16 timeout -k 10 200 python tools/tri_ab_probe.py REF TRI v0 "cap16=$c16 mult16=$m16" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r05/tri_split_sweep5.log
done; done
