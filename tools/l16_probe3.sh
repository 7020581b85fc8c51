export RT355_LIB=tools/bin/librt355_dev.so KNOB_CONFIG=C3 KNOB_BATCH=64
for cap in 8 10 12 16; do
  RT355_BVH_CAP=$cap timeout -k 10 120 python tools/knob_ab.py serial "C3 cap=$cap" 2>&1 | grep serial
  RT355_BVH_CAP=$cap timeout -k 10 120 python tools/knob_ab.py "C3 cap=$cap" 2>&1 | grep "in flight"
done
