#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, mean of each counter per dispatch."""
import csv, glob, os, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if "bdsp" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-24s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
