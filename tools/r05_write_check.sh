# quick check (round 5): HBM write traffic and scratch of the triangle kernel after the spill removal -- WRITE_SIZE per launch, Scratch_Size column
export TMPDIR=/tmp
export RT355_BENCH_NO_CHILDREN=1   # bench.py starts no child processes (amd-smi, node) under the profiler
mkdir -p gpurun_out/r05/wcheck
rm -rf gpurun_out/r05/wcheck/*
for cfg in REF TRI4K; do
  for mode in "--serial" ""; do
    tag=$cfg$( [ -n "$mode" ] && echo _serial || echo _inflight )
    # (one counter per pass: WRITE_SIZE and FETCH_SIZE together exceed what the hardware collects at once, and the failed run hangs)
    timeout -k 5 90 rocprofv3 --output-format csv --pmc WRITE_SIZE -d gpurun_out/r05/wcheck/$tag -o w -- python3 bench.py --config $cfg --steps 12 --warmup 3 --no-cpu-baseline --serial-steps 0 --repeats 1 --no-node $mode > gpurun_out/r05/wcheck/$tag.json 2> gpurun_out/r05/wcheck/$tag.err < /dev/null
    f=$(find gpurun_out/r05/wcheck/$tag -name "*counter_collection.csv" 2>/dev/null | head -n 1)
    [ -n "$f" ] && python3 - "$f" $tag <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); scr = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:70]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"])); scr[k] = (r.get("Scratch_Size"), r.get("VGPR_Count"), r.get("LDS_Block_Size"))
for k, v in acc.items():
    if "trace_triangles" in k:
        print(sys.argv[2], k, {c: "%.1f MB x%d" % (sum(x) / len(x) * 64 / 1e6 if "SIZE" in c else sum(x) / len(x), len(x)) for c, x in v.items()}, "scratch/vgpr/lds", scr[k])
PY
  done
done
