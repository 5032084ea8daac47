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

# HBM traffic per launch for bench.py's roofline.traffic: (FETCH_SIZE + WRITE_SIZE) KiB -> bytes
import json
tr = {}
for kind in ("fetch", "write"):
    for path in glob.glob(os.path.join(root, kind, "*counter_collection.csv")):
        acc = collections.defaultdict(list)
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "")
                if k.startswith("k4::") and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    name = k.split("(")[0].replace("k4::", "")
                    acc[name].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            tr[k] = tr.get(k, 0.0) + 1024.0 * sum(v) / len(v)
if tr:
    # the encode call runs its two kernels side by side: report their sum under the LDS-table kernel's name
    if "k4_encode_fast_gtab_kernel" in tr:
        tr["k4_encode_fast_kernel"] = tr.get("k4_encode_fast_kernel", 0.0) + tr.pop("k4_encode_fast_gtab_kernel")
    out = {k: int(v) for k, v in tr.items()}
    print("traffic bytes per launch:", out)
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
