# round 5: the triangle kernel's in-flight form (six waves per SIMD) under the counters the awaited form is priced with -- wait share,
# VALU issue, L1 -> L2 latency -- from launches of frames in flight (the profiler serialises them; instruction counts and per-launch
# cycles are those of the launch configuration)
export TMPDIR=/tmp RT355_BENCH_NO_CHILDREN=1
mkdir -p gpurun_out/r05/inflight_pmc
for cfg in REF TRI4K; do
  i=0
  for P in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 5 200 rocprofv3 --output-format csv --pmc $P -d gpurun_out/r05/inflight_pmc/$cfg$i -o p -- python3 bench.py --config $cfg --steps 12 --warmup 3 --no-cpu-baseline --serial-steps 0 --repeats 1 --no-node > gpurun_out/r05/inflight_pmc/$cfg$i.json 2> gpurun_out/r05/inflight_pmc/$cfg$i.err < /dev/null
    f=$(find gpurun_out/r05/inflight_pmc/$cfg$i -name "*counter_collection.csv" 2>/dev/null | head -n 1)
    [ -n "$f" ] && python3 - "$f" $cfg <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:80]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in acc.items():
    if "trace_triangles" in k:
        m = {c: sum(x) / len(x) for c, x in v.items()}
        line = "%s %s launches %d mean %.1f us" % (sys.argv[2], k.replace("void rtk::", ""), len(dur[k]), sum(dur[k].values()) / len(dur[k]) / 1e3)
        if "SQ_WAIT_ANY" in m: line += " | wait share %.3f  VALU issue %.3f  waves %.0f" % (m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 2 * m["SQ_INSTS_VALU"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8), m["SQ_WAVES"])
        if "TCP_TCC_READ_REQ_sum" in m: line += " | L1->L2 reads %.3g  mean latency %.0f cycles" % (m["TCP_TCC_READ_REQ_sum"], m["TCP_TCC_READ_REQ_LATENCY_sum"] / m["TCP_TCC_READ_REQ_sum"])
        print(line, flush=True)
PY
  done
done
