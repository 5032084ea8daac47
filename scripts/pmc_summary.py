#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean per dispatch of each counter."""
import csv, glob, os, sys, collections
root = sys.argv[1]
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "?")
            if not k.startswith("k4::"):
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", os.path.relpath(path, root))
    for k, cs in acc.items():
        print("  ", k.split("(")[0], {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "dispatches", len(next(iter(cs.values()))))
