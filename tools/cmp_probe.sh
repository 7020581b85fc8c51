#!/bin/bash
# Development probe: C5 through the compact form (RT355_BVH_CMP) and the exact 16-wave form, hierarchy arity 4 / 6 (dev library).
export RT355_LIB=tools/bin/librt355_dev.so KNOB_CONFIG=C5 KNOB_BATCH=8
for cmp in 0 1; do for ar in 4 6; do
  RT355_BVH_CMP=$cmp RT355_BVH_ARITY=$ar timeout -k 10 120 python tools/knob_ab.py "cmp=$cmp arity=$ar" || exit 1
done; done
