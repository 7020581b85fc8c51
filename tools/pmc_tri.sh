#!/bin/bash
# PMC passes of a triangle configuration one frame at a time: tools/pmc_tri.sh <out dir> <REF|TRI|TRI4K> <variant> ; then tools/pmc_tri_summary.py <out dir>
set -e
OUT=$1; CFG=${2:-REF}; V=${3:-0}
mkdir -p $OUT
export TMPDIR=/tmp
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
P2="SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
P3="TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
P4="SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $P -d $OUT/p$i -o p -- python3 tools/tri_frames.py $CFG 6 $V > $OUT/p$i.out 2> $OUT/p$i.err || { tail -5 $OUT/p$i.err; }
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && cp "$f" $OUT/${CFG}-v${V}__p$i.csv
  rm -rf $OUT/p$i
done
ls $OUT
