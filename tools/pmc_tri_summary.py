"""Per-kernel averages of the PMC passes tools/pmc_tri.sh took: python3 tools/pmc_tri_summary.py <dir>"""
import csv, glob, os, sys, collections
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "*__p*.csv"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(os.path.basename(f))
    for k, cs in acc.items():
        if "trace_" not in k and "sky_resolve" not in k: continue
        print("  ", k, " ".join("%s=%.4g" % (c, sum(v[1:]) / max(len(v) - 1, 1)) for c, v in sorted(cs.items())))
