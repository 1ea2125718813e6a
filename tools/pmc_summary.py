#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_pass.sh: python tools/pmc_summary.py <outdir> [name-filter]"""
import csv, glob, sys, collections
out = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        acc[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
for nm, cs in sorted(acc.items()):
    if flt not in nm:
        continue
    print(nm)
    for c, v in sorted(cs.items()):
        print(f"    {c:40s} {sum(v)/len(v):16.1f}   (n={len(v)})")
