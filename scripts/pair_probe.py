#!/usr/bin/env python
"""Where the two waves of a decoder pair spend their cycles, by data class (k4lz4_profile_batch_device, decode = 2)."""
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
ol, c = dc.profile(2, csrc, back)
torch.cuda.synchronize()
assert (ol.cpu().numpy() == bs).all()
c = c.cpu().numpy().astype(np.float64)
names = corpus.SILESIA_NAMES
print("PARSING wave: total, waiting for a free slot, parsing; batches, sequences, scalar-parser sequences | per speculative round: hypotheses chain rules slots")
for ci, name in list(enumerate(names)) + [(-1, "ALL")]:
    m = (c[np.arange(ci, n, 12)] if ci >= 0 else c).mean(axis=0)
    r = max(m[15], 1)
    print("%-8s %9.0f %9.0f %9.0f | %5.0f %6.0f %5.0f | %5.0f %5.0f %5.0f %5.0f" % (name, m[0], m[1], m[2], m[4], m[6], m[7], m[11] / r, m[12] / r, m[13] / r, m[14] / r))
print("COPYING wave: total, waiting for a batch, + descriptors; per batch: to end of search, fill (incl. search), rounds, flush | batches, rounds per batch, sequences per batch")
for ci, name in list(enumerate(names)) + [(-1, "ALL")]:
    m = (c[np.arange(ci, n, 12)] if ci >= 0 else c).mean(axis=0)[16:]
    nb = max(m[8], 1)
    print("%-8s %9.0f %9.0f %9.0f | %6.0f %6.0f %6.0f %6.0f (rest %6.0f) | %5.0f %5.2f %5.1f" % (name, m[0], m[1], m[2], m[3] / nb, m[4] / nb, m[5] / nb, m[6] / nb, (m[7] - m[5] - m[6]) / nb, nb, m[9] / nb, m[10] / nb))
