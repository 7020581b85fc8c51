#!/bin/bash
# Collects, on the GPU box, the rocprofv3 evidence bench.py's roofline refers to:
#   kernel-trace stats of the default bench run (pipelined) and of --serial, and the PMC passes of
#   --serial (SQ counters, FETCH_SIZE, WRITE_SIZE: separate runs, as the MI355X guide prescribes).
# usage: tools/collect_profiles.sh <round dir under gpurun_out, e.g. r02> [KEY=C3-fast-v0-n1] [bench args ...]
# Results land in gpurun_out/<round>/prof/; copy what is to be judged into profiles/<round>/ and run
# tools/pmc_summary.py.
set -e
RD=${1:-r02}; shift || true
KEY=${1:-C3-fast-v0-n1}; shift || true
OUT=gpurun_out/$RD/prof
mkdir -p $OUT
export TMPDIR=/tmp
export RT355_BENCH_NO_CHILDREN=1   # bench.py starts no child processes (amd-smi, node) under the profiler
ARGS="--steps 12 --warmup 3 --no-cpu-baseline --serial-steps 0 --repeats 1 --no-node $@"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pipelined -o t -- python3 bench.py $ARGS > $OUT/${KEY}__pipelined_bench.json 2> $OUT/trace_pipelined.err
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -o t -- python3 bench.py $ARGS --serial > $OUT/${KEY}__serial_bench.json 2> $OUT/trace_serial.err
timeout -k 5 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o sq -- python3 bench.py $ARGS --serial > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
# the same counters for the launches of frames in flight (one workgroup per CU, their own suspension threshold): the
# profiler serialises the dispatches, the launch configuration and with it the instruction count are those of the default run
timeout -k 5 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_sqflight -o sqflight -- python3 bench.py $ARGS > $OUT/pmc_sqflight.json 2> $OUT/pmc_sqflight.err
timeout -k 5 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python3 bench.py $ARGS --serial > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -k 5 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python3 bench.py $ARGS --serial > $OUT/pmc_write.json 2> $OUT/pmc_write.err
mkdir -p $OUT/collected/pmc
for p in sq sqflight fetch write; do
  f=$(find $OUT/pmc_$p -name "*counter_collection.csv" | head -n 1)
  [ -n "$f" ] && cp "$f" $OUT/collected/pmc/${KEY}__$p.csv
done
for m in pipelined serial; do
  f=$(find $OUT/trace_$m -name "*kernel_stats.csv" | head -n 1)
  [ -n "$f" ] && cp "$f" $OUT/collected/${KEY}__${m}_kernel_stats.csv
  cp $OUT/${KEY}__${m}_bench.json $OUT/collected/
done
ls -la $OUT/collected $OUT/collected/pmc
# memory side of the triangle kernel (RT_CACHE_PASSES=1): L1 -> L2 read requests and their latency, L2 hits / misses
if [ -n "$RT_CACHE_PASSES" ]; then
  timeout -k 5 300 rocprofv3 --output-format csv --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE -d $OUT/pmc_tcp -o tcp -- python3 bench.py $ARGS --serial > $OUT/pmc_tcp.json 2> $OUT/pmc_tcp.err
  timeout -k 5 300 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum GRBM_GUI_ACTIVE -d $OUT/pmc_tcc -o tcc -- python3 bench.py $ARGS --serial > $OUT/pmc_tcc.json 2> $OUT/pmc_tcc.err
  timeout -k 5 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc_stall -o stall -- python3 bench.py $ARGS --serial > $OUT/pmc_stall.json 2> $OUT/pmc_stall.err
  for p in tcp tcc stall; do
    f=$(find $OUT/pmc_$p -name "*counter_collection.csv" | head -n 1)
    [ -n "$f" ] && cp "$f" $OUT/collected/pmc/${KEY}__$p.csv
  done
fi
# development passes (not needed by bench.py): where the idle issue slots go
if [ -n "$RT_EXTRA_PASSES" ]; then
  timeout -k 5 300 rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o lds -- python3 bench.py $ARGS --serial > $OUT/pmc_lds.json 2> $OUT/pmc_lds.err
  timeout -k 5 300 rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc_stall -o stall -- python3 bench.py $ARGS --serial > $OUT/pmc_stall.json 2> $OUT/pmc_stall.err
  for p in lds stall; do
    f=$(find $OUT/pmc_$p -name "*counter_collection.csv" | head -n 1)
    [ -n "$f" ] && cp "$f" $OUT/collected/pmc/${KEY}__$p.csv
  done
fi
