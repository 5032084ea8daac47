#!/usr/bin/env python
"""Where the two waves of a second-generation decoder pair (k4lz4_decode2.hpp) spend their cycles, by data class
(k4lz4_profile_batch_device, decode = 2 -> k4_decode_x_prof_kernel)."""
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
print("PARSER: total cycles, waiting for room | rounds, scalar sequences, records | per round: hyp, doubling, entries, rules+records | per scalar sequence")
for ci, name in list(enumerate(names)) + [(-1, "ALL")]:
    m = (c[np.arange(ci, n, 12)] if ci >= 0 else c).mean(axis=0)
    r = max(m[2], 1)
    print("%-8s %9.0f %9.0f | %5.0f %5.0f %6.0f | %5.0f %5.0f %5.0f %5.0f | %5.0f" % (name, m[0], m[1], m[2], m[3], m[4], m[5] / r, m[6] / r, m[7] / r, m[11] / r, m[12] / max(m[3], 1)))
print("COPIER: total cycles, waiting for records | windows, polls, windows with in-window sources, idle steps | cycles per window net of waiting")
for ci, name in list(enumerate(names)) + [(-1, "ALL")]:
    m = (c[np.arange(ci, n, 12)] if ci >= 0 else c).mean(axis=0)[16:]
    w = max(m[2], 1)
    print("%-8s %9.0f %9.0f | %5.0f %6.0f %5.0f %5.0f | %6.0f" % (name, m[0], m[1], m[2], m[3], m[4], m[5], (m[0] - m[1]) / w))
