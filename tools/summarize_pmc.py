#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv by kernel name: mean value per dispatch and number of dispatches."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel,counter,dispatches,mean_per_dispatch,total")
for name in sorted(agg, key=lambda n: -sum(sum(v) for v in agg[n].values())):
    for c, v in sorted(agg[name].items()):
        print("%s,%s,%d,%.1f,%.0f" % (name, c, len(v), sum(v) / len(v), sum(v)))
