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


def per_kernel_raw(kind, counter, prefix):
    return {k: v / 1024.0 for k, v in per_kernel(kind, counter, prefix).items()}     # (per_kernel scales KiB counters by 1024)


# what the calibration kernels are known to move: bytes read / written per launch, counting what a scattered access really
# costs at the memory side (a 128-byte line per 16-byte read, a 32-byte sector per 2-byte write)
known_rd = {"calib_stream_read16": GiB, "calib_scatter_read16": GiB / 4 / 16 * 128}
known_wr = {"calib_stream_write16": GiB, "calib_scatter_write2": GiB / 4 / 16 * 32}
cal_rd = per_kernel_raw("caldramrd", "TCC_EA0_RDREQ_DRAM_32B_sum", "calib_")
cal_wr = per_kernel_raw("caldramwr", "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", "calib_")
check = {}
for k, want in known_rd.items():
    if cal_rd.get(k):
        check[k] = round(32.0 * cal_rd[k] / want, 4)
for k, want in known_wr.items():
    if cal_wr.get(k):
        check[k] = round(32.0 * cal_wr[k] / want, 4)
print("calibration (32 B x DRAM request counters / known bytes, 1.0 = exact):", check)
cal_fetch, cal_write = per_kernel("calfetch", "FETCH_SIZE", "calib_"), per_kernel("calwrite", "WRITE_SIZE", "calib_")
print("for comparison, FETCH_SIZE / WRITE_SIZE on the same kernels (reported bytes / known bytes):",
      {k: round(cal_fetch[k] / known_rd[k], 3) for k in known_rd if cal_fetch.get(k)},
      {k: round(cal_write[k] / known_wr[k], 3) for k in known_wr if cal_write.get(k)})
rd = per_kernel_raw("dramrd", "TCC_EA0_RDREQ_DRAM_32B_sum", "k4::")
wr = per_kernel_raw("dramwr", "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", "k4::")
out = {k: int(32.0 * (rd.get(k, 0.0) + wr.get(k, 0.0))) for k in set(rd) | set(wr)}
# ---- the issue ports (bench.py's roofline.issue): per kernel, mean per dispatch, from the sq1 / sq2 / grbm passes --------------
def counters_of(kind, names):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    paths = glob.glob(os.path.join(root, kind, "**", "*counter_collection.csv"), recursive=True) + \
        glob.glob(os.path.join(root, kind + "_counter_collection.csv"))
    for path in paths:
        seen = set()
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "")
                if not k.startswith("k4::") or row["Counter_Name"] not in names:
                    continue
                k = k.split("(")[0].replace("k4::", "")
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row["Dispatch_Id"] not in seen:
                    seen.add(row["Dispatch_Id"])
                    dur[k].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    return ({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}, {k: sum(v) / len(v) for k, v in dur.items()})


SIMDS = 1024          # 256 CUs x 4
c1, _ = counters_of("sq1", {"SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"})
c2, _ = counters_of("sq2", {"SQ_INSTS_SMEM", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA"})
cg, dg = counters_of("grbm", {"GRBM_GUI_ACTIVE"})
issue = {}
for k in c1:
    if k not in c2 or k not in cg:
        continue
    cycles = cg[k]["GRBM_GUI_ACTIVE"] / 8.0                # summed over the 8 XCDs; the kernel alone (rocprofv3 serialises dispatches while it counts)
    insts = sum(c1[k].get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")) + c2[k].get("SQ_INSTS_SMEM", 0.0)
    issue[k] = {"insts_total": insts, "insts_valu": c1[k].get("SQ_INSTS_VALU", 0.0), "insts_salu": c1[k].get("SQ_INSTS_SALU", 0.0),
                "insts_lds": c1[k].get("SQ_INSTS_LDS", 0.0), "insts_vmem": c1[k].get("SQ_INSTS_VMEM_RD", 0.0) + c1[k].get("SQ_INSTS_VMEM_WR", 0.0),
                "active_valu_quadcycles": c2[k].get("SQ_ACTIVE_INST_VALU", 0.0), "active_scalar_quadcycles": c2[k].get("SQ_ACTIVE_INST_SCA", 0.0),
                "kernel_cycles_alone": cycles, "kernel_ns_alone": dg.get(k, 0.0),
                "valu_busy": 4.0 * c2[k].get("SQ_ACTIVE_INST_VALU", 0.0) / SIMDS / cycles if cycles else None,
                "salu_busy": 4.0 * c2[k].get("SQ_ACTIVE_INST_SCA", 0.0) / SIMDS / cycles if cycles else None}
if issue:
    print("issue ports (x4 quad-cycles / 1024 SIMDs / the kernel's own cycles):",
          {k: (round(v["valu_busy"], 3), round(v["salu_busy"], 3), int(v["insts_total"])) for k, v in issue.items()})
cl2, _ = counters_of("l2", {"TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"})
l2 = {k: {"hit": v.get("TCC_HIT_sum", 0.0), "miss": v.get("TCC_MISS_sum", 0.0), "req": v.get("TCC_REQ_sum", 0.0),
          "hit_rate": (v.get("TCC_HIT_sum", 0.0) / (v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 0.0))) if v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 0.0) else None}
      for k, v in cl2.items()}
if l2:
    print("L2 (TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum), requests per launch):", {k: (None if v["hit_rate"] is None else round(v["hit_rate"], 3), int(v["req"])) for k, v in l2.items()})
if out:
    doc = {"source_sha": source_hash(), "l2": l2, "counters": "32 B x (TCC_EA0_RDREQ_DRAM_32B_sum + TCC_EA0_WRREQ_WRITE_DRAM_32B_sum), separate passes",
           "calibration_vs_known_bytes": check,
           "read_bytes_per_launch": {k: int(32.0 * v) for k, v in rd.items()}, "write_bytes_per_launch": {k: int(32.0 * v) for k, v in wr.items()},
           "traffic_bytes_per_launch": out, "issue": issue}
    print("traffic bytes per launch:", out)
    if len(sys.argv) > 2:
        json.dump(doc, open(sys.argv[2], "w"), indent=1)
