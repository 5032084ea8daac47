#!/usr/bin/env python
"""Per-phase cycles of the parse kernel (k4lz4_parse.hpp) on the bench batch, by data class: needs a build with -DK4_PARSE_PROF
(scripts/build_variant.sh prof -DK4_PARSE_PROF ...), whose parse kernel fills the counters of k4lz4_profile_batch_device."""
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
clen = dc.encode(src, comp)
torch.cuda.synchronize()
_, c = dc.profile(False, src, comp)
torch.cuda.synchronize()
c = c.cpu().numpy().astype(np.uint64)
names = corpus.SILESIA_NAMES
print("class     kind   total    front  candwait  words+grp  chain  next+vis+rec  commit | rounds notplain strided subwin/round  seq/round lazy/round long/round groups/round")
def row(name, sel):
    cs = c[sel]
    lo = lambda col: (cs[:, col] & 0xffffffff).astype(np.float64).mean()
    hi = lambda col: (cs[:, col] >> 32).astype(np.float64).mean()
    tot = cs[:, 0].astype(np.float64).mean(); r = max(lo(11), 1)
    ph = [cs[:, 1 + i].astype(np.float64).mean() / r for i in range(6)]
    print("%-12s %8.0f " % (name, tot) + " ".join("%8.0f" % v for v in ph) + " | %6.0f %6.0f %6.0f %6.2f   %6.2f %6.2f %6.2f %6.2f" %
          (r, lo(12), hi(12), hi(11) / r, lo(7) / r, hi(7) / r, hi(14) / r, lo(14) / r) +
          "  | chain-asm %6.0f (%.2f entries) lazy %6.0f per round" % (lo(13) / r, cs[:, 10].astype(np.float64).mean() / r, hi(13) / r))
for kind in (1, 2):
    for ci, name in enumerate(names):
        sel = np.array([i for i in range(ci, n, 12) if c[i, 15] == kind])
        if sel.size: row(name + ("/lds" if kind == 1 else "/mem"), sel)
row("ALL", np.arange(n))
st, en = c[:, 8].astype(np.float64), c[:, 9].astype(np.float64)
print("span ms", (en.max() - st.min()) / 1e5, "mean block ms", (en - st).mean() / 1e5, "max block ms", (en - st).max() / 1e5)
for kind in (1, 2):
    sel = c[:, 15] == kind
    if sel.any(): print("kind", kind, "blocks", int(sel.sum()), "mean ms", (en[sel] - st[sel]).mean() / 1e5, "max ms", (en[sel] - st[sel]).max() / 1e5)
