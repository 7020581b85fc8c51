timeout -k 10 250 python bench.py --config C3 --steps 100 --warmup 5 > gpurun_out/r04/bench/C3_steps100.json 2> gpurun_out/r04/bench/C3_steps100.err
for cfg in C3 C5; do
  export KNOB_CONFIG=$cfg KNOB_BATCH=$([ $cfg = C5 ] && echo 8 || echo 64)
  for w in 1 8; do
    KNOB_WORLD=$w timeout -k 10 120 python tools/knob_ab.py serial "$cfg world=$w" 2>&1 | grep serial
    KNOB_WORLD=$w timeout -k 10 120 python tools/knob_ab.py "$cfg world=$w" 2>&1 | grep "in flight"
  done
done
timeout -k 10 200 python tools/tri_ab_probe.py v0 2>&1 | grep -v amdgpu
