#!/usr/bin/env python
"""When every block of the bench batch starts and ends inside the ORDINARY kernels (k4lz4_profile_batch_device, modes 4 / 5:
only start / end / placement are recorded), by data class and, for encode, by kernel (1 LDS table, 2 global table of the one-kernel
encoders; the parse kernel: 4 table in LDS, 5 in memory, 6 in memory first and in an LDS table another block was done with later;
for those also when the parse was through, i.e. how long writing the block out took)."""
import os, sys
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
clen = dc.encode(src, comp)
torch.cuda.synchronize()
csrc = DeviceBatch(comp.data, comp.off, clen)
names = corpus.SILESIA_NAMES
for mode in (4, 5):
    for rep in range(2):
        _, c = dc.profile(mode, csrc, back) if mode == 5 else dc.profile(mode, src, comp)
        torch.cuda.synchronize()
    c = c.cpu().numpy().astype(np.float64)
    st, en, kern = c[:, 8], c[:, 9], c[:, 11].astype(int)
    t0 = st.min()
    print(("DECODE" if mode == 5 else "ENCODE"), "span ms %.3f" % ((en.max() - t0) / 1e5))
    for kk in sorted(set(kern)):
        sel = kern == kk
        print(" kernel", kk, "blocks", int(sel.sum()), "first start %.3f last end %.3f ms" % ((st[sel].min() - t0) / 1e5, (en[sel].max() - t0) / 1e5))
        for ci, name in enumerate(names):
            idx = np.array([i for i in range(ci, n, 12) if sel[i]])
            if idx.size == 0: continue
            d = (en[idx] - st[idx]) / 1e5
            extra = ""
            if mode == 4 and kk >= 4:
                w = (en[idx] - c[idx, 12]) / 1e5
                extra = " | writing out mean %.3f max %.3f" % (w.mean(), w.max())
            print("   %-8s n %4d  start mean %.3f max %.3f | duration mean %.3f p10 %.3f p90 %.3f max %.3f | end max %.3f" % (
                name, idx.size, (st[idx].mean() - t0) / 1e5, (st[idx].max() - t0) / 1e5, d.mean(), np.percentile(d, 10), np.percentile(d, 90), d.max(), (en[idx].max() - t0) / 1e5) + extra)
