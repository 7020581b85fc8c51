# Development probe: thresholds of the work list's sixteenths (RT355_TRI_MULT16, RT355_TRI_CAP16) with the five-wave kernel, awaited frames
export RT355_LIB=tools/bin/librt355_dev.so
for m in 2 4 8; do for c in 32 64 128; do
  RT355_TRI_MULT16=$m RT355_TRI_CAP16=$c timeout -k 10 200 python tools/tri_ab_probe.py REF TRI v0 "mult16=$m cap16=$c" 2>&1 | grep -v amdgpu.ids
done; done
