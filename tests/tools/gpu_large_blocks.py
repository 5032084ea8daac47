#!/usr/bin/env python
"""GPU vs oracle on a few large single blocks (byU32 table, positions far beyond 64 KiB, long literal runs)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
from k4os.compression.lz4_amd import LZ4Codec, LZ4Level, corpus
o = Oracle()
cases = [("dickens", 32 << 20), ("xml", 8 << 20), ("x-ray", 16 << 20), ("mozilla", 24 << 20)]
blocks = [corpus.class_bytes(n, s, 77) for n, s in cases] + [corpus.repeated(3, 20 << 20)]
for lvl in (LZ4Level.L00_FAST, LZ4Level.L03_HC):
    t = time.perf_counter(); enc = LZ4Codec.EncodeBatch(blocks, lvl); te = time.perf_counter() - t
    for b, e in zip(blocks, enc):
        if lvl == LZ4Level.L00_FAST:
            want = o.encode(b)
        else:
            r, w = o.compress_hc(b, int(lvl)); want = w[:r].tobytes()
        assert e == want, (lvl, b.size, len(e), len(want))
    t = time.perf_counter(); dec = LZ4Codec.DecodeBatch(enc, [b.size for b in blocks]); td = time.perf_counter() - t
    assert all(d == b.tobytes() for d, b in zip(dec, blocks))
    print(lvl.name, "ok:", [b.size for b in blocks], "encode s", round(te, 2), "decode s", round(td, 2))
