"""Averages per bvh_pixels launch of the counter passes tools/pmc_cmp.sh wrote (cmp0 = exact 16-wave form, cmp1 = compact)."""
import csv, glob, os, sys, collections
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "cmp*__*.csv"))):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "bvh_pixels" not in r["Kernel_Name"]: continue
        acc[(r["Kernel_Name"][:60], r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    per = collections.defaultdict(list)
    for (k, c, disp), v in acc.items(): per[(k, c)].append(sum(v))
    print(os.path.basename(f))
    for (k, c), v in sorted(per.items()):
        print("   %-28s %14.4e  (%d launches)  %s" % (c, sum(v) / len(v), len(v), k))
