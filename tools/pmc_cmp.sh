#!/bin/bash
# Development: C5 through the exact 16-wave form and the compact 12-wave form (dev library), SQ and LDS counter passes.
# usage (GPU box): tools/pmc_cmp.sh <out dir>
set -e
OUT=${1:-gpurun_out/r04/pmc_cmp}; mkdir -p $OUT
export TMPDIR=/tmp RT355_LIB=tools/bin/librt355_dev.so RT355_BVH_ARITY=${ARITY:-6}
ARGS="--config C5 --steps 4 --warmup 1 --no-cpu-baseline --serial-steps 0 --serial"
for cmp in 0 1; do
  export RT355_BVH_CMP=$cmp
  rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/sq$cmp -o sq -- python3 bench.py $ARGS > $OUT/sq$cmp.json 2> $OUT/sq$cmp.err
  rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/lds$cmp -o lds -- python3 bench.py $ARGS > $OUT/lds$cmp.json 2> $OUT/lds$cmp.err
  rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/stall$cmp -o stall -- python3 bench.py $ARGS > $OUT/stall$cmp.json 2> $OUT/stall$cmp.err
  for p in sq lds stall; do f=$(find $OUT/$p$cmp -name "*counter_collection.csv" | head -n 1); cp "$f" $OUT/cmp${cmp}__$p.csv; done
  rm -rf $OUT/sq$cmp $OUT/lds$cmp $OUT/stall$cmp
done
python3 tools/pmc_cmp_summary.py $OUT
