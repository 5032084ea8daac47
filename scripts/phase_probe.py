#!/usr/bin/env python
"""Per-phase cycle breakdown of the encode / decode kernels on the bench workload, by data class
(uses the instrumented twin kernels behind k4lz4_profile_batch_device)."""
import os, sys, json, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from k4os.compression.lz4_amd import LZ4Codec, corpus
from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec

n, bs = int(os.environ.get("K4_BLOCKS", "4096")), int(os.environ.get("K4_BS", "65536"))
blocks = corpus.silesia_like_blocks(n, bs, seed=2)
dc = DeviceCodec(0)
lens = np.full(n, bs, np.int32)
off = np.arange(n, dtype=np.uint64) * bs
src = DeviceBatch.from_host(blocks.reshape(-1), off, lens, dc.device)
comp = DeviceBatch.empty_slots(np.full(n, LZ4Codec.MaximumOutputSize(bs)), dc.device)
back = DeviceBatch.empty_slots(lens, dc.device)

def timed(fn, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3

clen = dc.encode(src, comp)
torch.cuda.synchronize()
csrc = DeviceBatch(comp.data, comp.off, clen)
print("encode ms", round(timed(lambda: dc.encode(src, comp, clen)), 3), "decode ms", round(timed(lambda: dc.decode(csrc, back)), 3))
names = corpus.SILESIA_NAMES
for decode in (False, True):
    t0 = time.perf_counter()
    if decode:
        _, c = dc.profile(True, csrc, back)
    else:
        _, c = dc.profile(False, src, comp)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    c = c.cpu().numpy().astype(np.float64)
    hdr = ("total", "parse", "lit", "match", "batches", "rounds", "nseq", "coop") if decode else ("total", "probe", "ext", "emit", "nseq", "rounds", "dups", "winsum")
    print(("DECODE" if decode else "ENCODE"), "instrumented wall ms", round(wall, 2))
    print("%-8s " % "class" + " ".join("%10s" % h for h in hdr) + "   ratio")
    cl = clen.cpu().numpy()
    for ci, name in enumerate(names):
        idx = np.arange(ci, n, 12)
        m = c[idx].mean(axis=0)
        print("%-8s " % name + " ".join("%10.0f" % v for v in m[:8]) + "   %.3f" % (cl[idx].mean() / bs))
    m = c.mean(axis=0)
    print("%-8s " % "ALL" + " ".join("%10.0f" % v for v in m[:8]))
    if not decode:
        d = c[np.arange(0, n, 12)].mean(axis=0)
        print("dickens resolve split per round: initial candidates %.0f  hop loop %.0f  recompute %.0f  post %.0f ; recomputes/round %.2f" % (tuple(d[11:15] / d[5]) + (d[6] / d[5],)))
        print("ALL     resolve split per round: initial %.0f hop %.0f recompute %.0f post %.0f" % tuple(m[11:15] / m[5]))
        print("per class and round: probe initial hop recompute post emit | sequences, long counts, fall-backs per round, share of rounds that continue a search")
        for ci, name in enumerate(names):
            d = c[np.arange(ci, n, 12)].mean(axis=0)
            r = max(d[5], 1)
            print("%-8s %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f | %5.2f %5.2f %5.2f %5.2f" % (name, d[1] / r, d[11] / r, d[12] / r, d[13] / r, d[14] / r, d[3] / r, d[4] / r, d[7] / r, d[6] / r, d[15] / r))
    if decode:
        for name in ("dickens", "ALL"):
            sel = np.arange(n) if name == "ALL" else np.array([i for i in range(n) if names[i % len(names)] == name])
            r = c[sel][:, 15].mean()
            print(f"{name} PARSE split per speculative round ({r:.0f} rounds per block): ring+hypotheses {c[sel][:, 11].mean() / r:.0f}  chain {c[sel][:, 12].mean() / r:.0f}"
                  f"  scan+rules {c[sel][:, 13].mean() / r:.0f}  slots {c[sel][:, 14].mean() / r:.0f}  (whole PARSE {c[sel][:, 1].mean() / r:.0f})")
    st, en = c[:, 8], c[:, 9]
    ev = sorted([(t, 1) for t in st] + [(t, -1) for t in en])
    cur = mx = 0
    for _, d in ev:
        cur += d; mx = max(mx, cur)
    print("max concurrent waves", mx, "span ms", (en.max() - st.min()) / 1e5, "mean block ms", (en - st).mean() / 1e5)
    t00 = st.min()
    print("class start/end ms: " + " ".join("%s %.1f/%.1f" % (nm[:4], (st[np.arange(ci, n, 12)].mean() - t00) / 1e5, (en[np.arange(ci, n, 12)].mean() - t00) / 1e5) for ci, nm in enumerate(names)))
    hw = c[:, 10].astype(np.int64)
    print("distinct HW_ID (cu/sh/se) values", len(set((hw >> 8) & 0xffff)))
    print("max total cycles", c[:, 0].max(), "-> ms at 2.4GHz", c[:, 0].max() / 2.4e6)
