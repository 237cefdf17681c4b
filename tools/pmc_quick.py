#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counters per kernel name prefix.  usage: pmc_quick.py DIR [substring]"""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
            continue
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", max(len(v) for v in d.values()))
