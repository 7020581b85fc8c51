export RT355_LIB=tools/bin/librt355_dev.so
for m in 1 2 3; do for c in 64 96; do
  RT355_TRI_MULT16=$m RT355_TRI_CAP16=$c timeout -k 10 200 python tools/tri_ab_probe.py REF TRI TRI4K v0 "mult16=$m cap16=$c" 2>&1 | grep -v amdgpu.ids
done; done
