# Development probe: C3 and C5, one at a time and in flight (dev library: RT355_BVH_CAP for the 16-wave form)
export RT355_LIB=tools/bin/librt355_dev.so
for cfg in C3 C5; do
  export KNOB_CONFIG=$cfg KNOB_BATCH=$([ $cfg = C5 ] && echo 8 || echo 64)
  timeout -k 10 120 python tools/knob_ab.py serial "$cfg" 2>&1 | grep serial
  timeout -k 10 120 python tools/knob_ab.py "$cfg" 2>&1 | grep "in flight"
done
for cap in 6 8 16; do
  KNOB_CONFIG=C5 KNOB_BATCH=8 RT355_BVH_CAP=$cap timeout -k 10 120 python tools/knob_ab.py "C5 cap=$cap" 2>&1 | grep "in flight"
done
