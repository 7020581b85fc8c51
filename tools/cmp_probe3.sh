export RT355_LIB=tools/bin/librt355_dev.so KNOB_CONFIG=C5 KNOB_BATCH=4
RT355_BVH_PRINT=1 RT355_BVH_CMP=1 RT355_BVH_ARITY=4 timeout -k 10 120 python tools/knob_ab.py serial 2>&1 | sort | uniq -c | tail -4
RT355_BVH_BLOCKS=256 RT355_BVH_CMP=1 RT355_BVH_ARITY=4 timeout -k 10 120 python tools/knob_ab.py serial "blocks256"
RT355_BVH_BLOCKS=512 RT355_BVH_CMP=1 RT355_BVH_ARITY=4 timeout -k 10 120 python tools/knob_ab.py serial "blocks512"
RT355_BVH_CMP=0 RT355_BVH_ARITY=4 timeout -k 10 120 python tools/knob_ab.py serial "exact"
