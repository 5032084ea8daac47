#!/usr/bin/env python
"""PCIe-inclusive rate of the host-pointer batch calls (k4lz4_encode_batch / k4lz4_decode_batch):
source in pageable host memory -> GPU -> results back in host memory.  Reported in DESIGN.md,
never as bench.py's `value`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from k4os.compression.lz4_amd import LZ4Codec, corpus, make_arena

n, bs = int(os.environ.get("K4_BLOCKS", "4096")), 65536
blocks = corpus.silesia_like_blocks(n, bs, seed=2)
src = blocks.reshape(-1)
off = np.arange(n, dtype=np.uint64) * bs
lens = np.full(n, bs, np.int32)
caps = np.full(n, LZ4Codec.MaximumOutputSize(bs), np.int32)
dst, doff = make_arena(caps)
back, boff = make_arena(lens)
best_e = best_d = 1e9
for _ in range(4):
    t = time.perf_counter(); out = LZ4Codec.EncodeBatchPacked(src, off, lens, dst, doff, caps); best_e = min(best_e, time.perf_counter() - t)
    t = time.perf_counter(); dl = LZ4Codec.DecodeBatchPacked(dst, doff, out, back, boff, lens); best_d = min(best_d, time.perf_counter() - t)
assert (dl == bs).all() and np.array_equal(back[:n * bs].reshape(n, bs), blocks)
g = n * bs / 2 ** 30
print(f"host-pointer path, {n} x {bs} B: encode {g / best_e:.2f} GiB/s, decode {g / best_d:.2f} GiB/s, "
      f"round trip {g / (best_e + best_d):.2f} GiB/s (PCIe + staging inclusive)")
