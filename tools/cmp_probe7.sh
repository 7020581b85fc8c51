export RT355_LIB=tools/bin/librt355_dev.so KNOB_CONFIG=C5 KNOB_BATCH=8
for ar in 4 6; do for cmp in 0 1; do
  RT355_BVH_PRINT=1 RT355_BVH_ARITY=$ar RT355_BVH_CMP=$cmp timeout -k 10 120 python tools/knob_ab.py serial "cmp=$cmp" 2>&1 | grep "serial\|bvh_pixels" | sort | uniq | tail -2
  RT355_BVH_ARITY=$ar RT355_BVH_CMP=$cmp timeout -k 10 120 python tools/knob_ab.py "cmp=$cmp" 2>&1 | grep "in flight"
done; done
