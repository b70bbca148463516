#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel per dispatch."""
import csv
import collections
import glob
import sys

for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][-40:]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            n = max(len(v) for v in cs.values())
            print(f"{k}  dispatches={n}")
            for c, v in sorted(cs.items()):
                print(f"    {c:28s} mean={sum(v)/len(v):16.1f}")
