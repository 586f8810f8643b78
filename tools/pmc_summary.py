#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs by kernel: sum of every counter over a kernel's dispatches.
usage: pmc_summary.py out.csv in1_counter_collection.csv [in2 ...]"""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][path].add(r["Dispatch_Id"])
names = sorted({c for k in agg for c in agg[k]})
with open(sys.argv[1], "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches_per_pass"] + names)
    for k in sorted(agg):
        w.writerow([k, max(len(v) for v in disp[k].values())] + [int(agg[k].get(c, 0)) for c in names])
print(open(sys.argv[1]).read())
