#!/usr/bin/env python
"""Where the two waves of a decoder pair spend their time (k4_decode_pair_prof_kernel), by data class, on the bench batch."""
import os, sys
import numpy as np
import torch
os.environ["K4LZ4_PROF_PAIR"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from k4os.compression.lz4_amd import LZ4Codec, corpus
from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec

n, bs = int(os.environ.get("K4_BLOCKS", "4096")), 65536
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
_, c = dc.profile(True, csrc, back)
torch.cuda.synchronize()
c = c.cpu().numpy().astype(np.float64)
hdr = ("P total", "windows", "derive", "P wait", "batches", "nwin", "-", "slow", "C total", "C wait", "lit", "match", "batches", "rounds", "nseq")
print("%-8s " % "class" + " ".join("%9s" % h for h in hdr))
for ci, name in enumerate(corpus.SILESIA_NAMES):
    m = c[np.arange(ci, n, 12)].mean(axis=0)
    print("%-8s " % name + " ".join("%9.0f" % v for v in m[:15]))
print("%-8s " % "ALL" + " ".join("%9.0f" % v for v in c.mean(axis=0)[:15]))
print("max P total %.0f  max C total %.0f cycles" % (c[:, 0].max(), c[:, 8].max()))
