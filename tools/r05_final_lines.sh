#!/bin/bash
# The final bench lines of a round, one per benched configuration, on the GPU box:  tools/r05_final_lines.sh <tag>
# (gpurun_out/r05/bench_<cfg>_<tag>.json; copy to profiles/r05/bench/ in the build container)
TAG=${1:-final}
mkdir -p gpurun_out/r05
for cfg in C3 REF TRI TRI4K C5; do
  timeout -k 10 400 python bench.py --config $cfg --steps 20 --warmup 5 > gpurun_out/r05/bench_${cfg}_$TAG.json 2> gpurun_out/r05/bench_${cfg}_$TAG.err || exit 1
  python - "$cfg" "gpurun_out/r05/bench_${cfg}_$TAG.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
keys = ("value", "ms_per_step", "ms_per_step_median", "serial_ms_per_step", "serial_ms_per_step_median",
        "node_loop_ms_per_step", "node_inflight_ms_per_step")
print(sys.argv[1], d["kernel"]["build_id"], {k: (round(d[k], 4) if isinstance(d.get(k), float) else d.get(k)) for k in keys},
      d["roofline"].get("frac"), d["roofline"].get("traffic"))
PY
done
