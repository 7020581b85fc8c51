# round 5: what the compute side allows at 1 / 2 / 4 / 8 ranks -- rank 0's share of a frame on one GPU (tools/knob_ab.py), frames in
# flight and one at a time, and the same with the ROOT's de-interleave of the whole frame behind it (dev hook RT355_DEV_ROOT_WORLD)
mkdir -p gpurun_out/r05
export RT355_LIB=tools/bin/librt355_dev.so
for cfg in C3 C5; do
  export KNOB_CONFIG=$cfg KNOB_BATCH=$([ $cfg = C5 ] && echo 8 || echo 64)
  for w in 1 2 4 8; do
    KNOB_WORLD=$w timeout -k 10 150 python tools/knob_ab.py "$cfg world=$w share" 2>&1 | grep --line-buffered "in flight" | tee -a gpurun_out/r05/scale_emulation.log
    KNOB_WORLD=$w timeout -k 10 150 python tools/knob_ab.py serial "$cfg world=$w share" 2>&1 | grep --line-buffered "serial" | tee -a gpurun_out/r05/scale_emulation.log
    [ $w -gt 1 ] && RT355_DEV_ROOT_WORLD=$w KNOB_WORLD=$w timeout -k 10 150 python tools/knob_ab.py "$cfg world=$w root (share + de-interleave)" 2>&1 | grep --line-buffered "in flight" | tee -a gpurun_out/r05/scale_emulation.log
  done
done
