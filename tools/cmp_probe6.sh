export RT355_LIB=tools/bin/librt355_dev.so KNOB_CONFIG=C5 KNOB_BATCH=8 RT355_BVH_ARITY=6
for cmp in 0 1; do
  RT355_BVH_CMP=$cmp timeout -k 10 120 python tools/knob_ab.py serial "cmp=$cmp"
  RT355_BVH_CMP=$cmp timeout -k 10 120 python tools/knob_ab.py "cmp=$cmp"
done
for tail in 16 20 24 32 40; do
  RT355_BVH_TAIL=$tail RT355_BVH_CMP=1 timeout -k 10 120 python tools/knob_ab.py "cmp=1"
done
for b in 192 256 384 512; do
  RT355_BVH_BLOCKS=$b RT355_BVH_CMP=1 timeout -k 10 120 python tools/knob_ab.py "cmp=1"
done
RT355_BVH_ARITY=8 RT355_BVH_CMP=1 timeout -k 10 120 python tools/knob_ab.py "cmp=1 arity8"
RT355_BVH_ARITY=8 RT355_BVH_CMP=1 timeout -k 10 120 python tools/knob_ab.py serial "cmp=1 arity8"
RT355_BVH_ARITY=7 RT355_BVH_CMP=1 timeout -k 10 120 python tools/knob_ab.py "cmp=1 arity7"
