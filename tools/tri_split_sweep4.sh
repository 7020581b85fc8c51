# round 5: from which cost on a tile of an awaited frame is rendered as quarters (mult4 / 2 times the frame's throughput time) and as sixteenths (mult16 / 2)
mkdir -p gpurun_out/r05
export RT355_LIB=tools/bin/librt355_dev.so
for m4 in 1 2 3 4 6; do for m16 in 4 8; do
  [ $m16 -lt $m4 ] && continue
  RT355_TRI_MULT4=$m4 RT355_TRI_MULT16=$m16 timeout -k 10 200 python tools/tri_ab_probe.py REF TRI v0 "mult4=$m4 mult16=$m16" 2>&1 | grep -v amdgpu | tee -a gpurun_out/r05/tri_split_sweep4.log
done; done
