mkdir -p gpurun_out/r05
L=gpurun_out/r05/tri_knobs.log
export RT355_LIB=tools/bin/librt355_dev.so
run() { timeout -k 10 200 python tools/tri_ab_probe.py v0 $CFGS "$1" 2>&1 | grep -v amdgpu | tee -a $L; }
CFGS="REF TRI"
for sm in 1 2; do
  for pad in 0 1280 2560 5120; do RT355_TRI_SMALL=$sm RT355_TRI_LDSPAD=$pad run "pad"; done
  for pr in 1 2; do RT355_TRI_SMALL=$sm RT355_TRI_PRIO=$pr run "prio"; done
done
