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

# HBM traffic per launch for bench.py's roofline.traffic.  FETCH_SIZE / WRITE_SIZE are reported in KiB; how many real bytes
# one reported KiB stands for depends on the access pattern, so the factors come from the calibration kernels of known byte
# counts (scripts/ubench/pmc_calib.hip) measured in the same session: streaming 16 B per lane for the decoder's and the
# LDS-table encoder's wide accesses, one 16 B piece / one 2 B store per 64 B line for the scattered ones.
import hashlib, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    """what bench.py compares: the kernels and the launcher the figures were measured on"""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "k4os", "compression", "lz4_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hpp", ".hip")):
            h.update(name.encode()); h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def per_kernel(kind, counter, prefix):
    acc = collections.defaultdict(list)
    paths = glob.glob(os.path.join(root, kind, "**", "*counter_collection.csv"), recursive=True) + \
        glob.glob(os.path.join(root, kind + "_counter_collection.csv"))          # (flattened copy under profiles/)
    for path in paths:
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "")
                if k.startswith(prefix) and row["Counter_Name"] == counter:
                    acc[k.split("(")[0].replace("k4::", "")].append(float(row["Counter_Value"]))
    return {k: 1024.0 * sum(v) / len(v) for k, v in acc.items()}


GiB = float(1 << 30)
known = {"calib_stream_read16": GiB, "calib_scatter_read16": GiB / 4 * 4, "calib_stream_write16": GiB, "calib_scatter_write2": GiB / 4 * 4}
useful = {"calib_stream_read16": GiB, "calib_scatter_read16": GiB / 4, "calib_stream_write16": GiB, "calib_scatter_write2": GiB / 64 / 2}
cal_f, cal_w = per_kernel("calfetch", "FETCH_SIZE", "calib_"), per_kernel("calwrite", "WRITE_SIZE", "calib_")
factors = {}
for k, v in list(cal_f.items()) + list(cal_w.items()):
    if v > 0 and k in known and ("read" in k) == (k in cal_f):
        factors[k] = round(known[k] / v, 4)      # real bytes moved (whole 64 B lines for the scattered kernels) per reported byte
print("calibration: reported bytes", {k: int(v) for k, v in {**cal_f, **cal_w}.items()}, "-> factors (line bytes / reported)", factors)
fetch, write = per_kernel("fetch", "FETCH_SIZE", "k4::"), per_kernel("write", "WRITE_SIZE", "k4::")
f_stream, f_scatter = factors.get("calib_stream_read16", 1.0), factors.get("calib_scatter_read16", 1.0)
w_stream, w_scatter = factors.get("calib_stream_write16", 1.0), factors.get("calib_scatter_write2", 1.0)
out = {}
for k in set(fetch) | set(write):
    scattered = k == "k4_encode_fast_gtab_kernel"      # 2-byte table accesses, a line each; everything else moves 16 B pieces
    out[k] = int(fetch.get(k, 0.0) * (f_scatter if scattered else f_stream) + write.get(k, 0.0) * (w_scatter if scattered else w_stream))
if out:
    doc = {"source_sha": source_hash(), "calibration_factors": factors,
           "raw_reported_bytes": {"fetch": {k: int(v) for k, v in fetch.items()}, "write": {k: int(v) for k, v in write.items()}},
           "traffic_bytes_per_launch": out}
    print("traffic bytes per launch:", out)
    if len(sys.argv) > 2:
        json.dump(doc, open(sys.argv[2], "w"), indent=1)
