#!/bin/bash
# Everything bench.py's roofline refers to, for every benched configuration, in one call on the GPU box:
#   tools/collect_all.sh <round, e.g. r03> [configs, default "C3 C5 REF TRI TRI4K": a call on the GPU box is limited to 20 minutes, two calls fit]
# kernel-trace stats (pipelined / serial), PMC passes (SQ, SQ in flight, FETCH_SIZE, WRITE_SIZE; LDS + stall passes for the
# sphere configs, TCP / TCC / stall passes for the triangle configs), the counting builds' totals (C3, C5).
# Then, in the build container:  cp -r gpurun_out/<round>/collected/* profiles/<round>/ && python tools/pmc_summary.py
set -e
RD=${1:-r03}
WHAT=${2:-C3 C5 REF TRI TRI4K}
want() { case " $WHAT " in *" $1 "*) return 0;; esac; return 1; }
mkdir -p gpurun_out/$RD/collected/pmc
run() {   # key, env assignment, bench args ...
  local key=$1 envs=$2; shift 2
  rm -rf gpurun_out/$RD/prof
  env $envs bash tools/collect_profiles.sh $RD $key "$@" > gpurun_out/$RD/collect_$key.log 2>&1 || { tail -20 gpurun_out/$RD/collect_$key.log; exit 1; }
  cp gpurun_out/$RD/prof/collected/pmc/* gpurun_out/$RD/collected/pmc/
  cp gpurun_out/$RD/prof/collected/*.json gpurun_out/$RD/prof/collected/*.csv gpurun_out/$RD/collected/
  echo "collected $key"
}
want C3 && run C3-fast-v0-n1 RT_EXTRA_PASSES=1
want C5 && run C5-fast-v0-n1 RT_EXTRA_PASSES=1 --config C5 --steps 6 --warmup 2
want REF && run REF-fast-v0-n1 RT_CACHE_PASSES=1 --config REF
want TRI && run TRI-fast-v0-n1 RT_CACHE_PASSES=1 --config TRI
want TRI4K && run TRI4K-fast-v0-n1 RT_CACHE_PASSES=1 --config TRI4K
# (the counting builds link the library's current objects: rebuild them first -- python tools/collect_counts.py --build, in the build container)
want C3 && python3 tools/collect_counts.py C3 gpurun_out/$RD/collected
want C5 && python3 tools/collect_counts.py C5 gpurun_out/$RD/collected
rm -rf gpurun_out/$RD/prof
ls gpurun_out/$RD/collected gpurun_out/$RD/collected/pmc | head -80
