"""FETCH_SIZE of tools/fetch_gather.hip's launches beside the bytes they request: python3 tools/fetch_gather_summary.py <counter_collection.csv> <stdout of the run>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
req = {}
order = []
for line in open(sys.argv[2]):
    p = line.split()
    if len(p) == 3 and p[1] == "requested_bytes":
        req[p[0]] = int(p[2]); order.append(p[0])
per = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == "FETCH_SIZE":
        per[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
# the two gather<64> variants share a kernel name: split by dispatch order (3 launches each)
seq = []
for k, v in per.items():
    for d, val in sorted(v): seq.append((d, k, val))
seq.sort()
groups = [seq[i:i + 3] for i in range(0, len(seq), 3)]
print("%-16s %14s %16s %8s %s" % ("launch", "requested MB", "FETCH_SIZE (KB)", "ratio", "(FETCH_SIZE is in KB; ratio = FETCH_SIZE bytes / requested bytes)"))
for name, g in zip(order, groups):
    fs = sum(v for _, _, v in g[1:]) / max(len(g) - 1, 1)
    print("%-16s %14.1f %16.0f %8.3f" % (name, req[name] / 1e6, fs, fs * 1024 / req[name]))
