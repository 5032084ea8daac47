#!/usr/bin/env python
"""Random stress of k4lz4_segments.hpp under the wave emulator: messages stitched from random / repeated / corpus-class parts,
random segment sizes and warm-ups (most boundaries do not verify) -- every envelope against the oracle's.  Prints the totals of
[cut blocks, segments, blocks joined as planned, pieces kept behind a bad boundary, runs resumed by the join, of those stopped at
a verified boundary, blocks encoded again whole].  Usage: python tests/tools/emu_stress_segments.py [cases] [seed] [two_step]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from emu_lib import Emu
from oracle_lib import Oracle
from k4os.compression.lz4_amd import corpus
from test_emulated_kernels import pack, arena

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ENGINE = (1 << 30) if "two_step" in sys.argv[3:] else 0        # the runs by the two-step encoder (k4_parse_seg_kernel) instead of the one-kernel encoders
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 123)
emu, oracle = Emu(), Oracle()
names = list(corpus.SILESIA_NAMES)
tot = np.zeros(8, np.int64)
for case in range(cases):
    blocks = []
    for m in range(3):
        ps = []
        for q in range(int(rng.integers(1, 5))):
            n = int(rng.integers(20000, 120000))
            kind = rng.integers(0, 4)
            if kind == 0:
                ps.append(corpus.random_bytes(n, int(rng.integers(1e6))))
            elif kind == 1:
                ps.append(corpus.repeated(int(rng.integers(256)), n))
            else:
                ps.append(corpus.class_bytes(names[int(rng.integers(12))], n, int(rng.integers(1000))))
        blocks.append(np.concatenate(ps))
    src, soff, slen = pack(blocks)
    caps = [oracle.lib.k4o_pickle_bound(b.size) for b in blocks]
    t, w = int(rng.integers(12000, 70000)), int(rng.integers(1000, 90000))
    dst, doff, dcap = arena(caps)
    out, stats = emu.pickle_seg_batch(src, soff, slen, dst, doff, dcap, 70000, t, w, flags=ENGINE | (int(rng.integers(1, 17)) << 24 if ENGINE else 0))
    tot += stats.astype(np.int64)
    for i, b in enumerate(blocks):
        assert dst[int(doff[i]):int(doff[i]) + int(out[i])].tobytes() == oracle.pickle(b, 0, 0), (case, i, t, w)
print("ok", cases, "cases", tot.tolist())
