# Development probe: suspension threshold of the walk (RT355_BVH_TAIL) with the twelve-entry lists: C3 in flight and one at a time, C5 in flight
export RT355_LIB=tools/bin/librt355_dev.so
for t in 12 16 20 24; do KNOB_CONFIG=C3 KNOB_BATCH=64 RT355_BVH_TAIL=$t timeout -k 10 120 python tools/knob_ab.py "C3" 2>&1 | grep "in flight"; done
for t in 8 12 16; do KNOB_CONFIG=C3 RT355_BVH_TAIL=$t timeout -k 10 120 python tools/knob_ab.py serial "C3" 2>&1 | grep "serial"; done
for t in 20 28 36; do KNOB_CONFIG=C5 KNOB_BATCH=8 RT355_BVH_TAIL=$t timeout -k 10 120 python tools/knob_ab.py "C5" 2>&1 | grep "in flight"; done
