#!/usr/bin/env python3
"""Per-kernel mean of every counter in a rocprofv3 counter_collection.csv."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    disp[k].add(r["Dispatch_Id"])
for k, cs in acc.items():
    for c, v in sorted(cs.items()):
        print(f"{k}\t{c}\tmean_per_dispatch={sum(v) / len(disp[k]):.6g}\tdispatches={len(disp[k])}")
