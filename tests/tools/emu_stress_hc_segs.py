#!/usr/bin/env python
"""Random stress of the level-3 HC parse by several waves per block (k4lz4_encode_hc.hpp, HcSegs) under the wave emulator: blocks of
8 192 .. 65 536 bytes stitched from random / repeated / periodic / corpus-class parts (long matches and runs across the waves' starts,
stretches without any match), two or four waves per block, ragged output limits -- every block against the oracle's bytes.
Usage: python tests/tools/emu_stress_hc_segs.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from emu_lib import Emu
from oracle_lib import Oracle
from k4os.compression.lz4_amd import corpus
from test_emulated_kernels import pack, arena, FLAG_RAW

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 321)
emu, oracle = Emu(), Oracle()
names = list(corpus.SILESIA_NAMES)
nblocks = 0
for case in range(cases):
    blocks = []
    for m in range(12):
        total = int(rng.integers(8192, 65537)) if rng.random() < 0.8 else int(rng.choice([8191, 8192, 8193, 8255, 8256, 65535, 65536, 12288, 16384, 32768, 49152]))
        ps, have = [], 0
        while have < total:
            n = min(total - have, int(rng.integers(1, 30000)))
            kind = int(rng.integers(0, 6))
            if kind == 0:
                ps.append(corpus.random_bytes(n, int(rng.integers(1e6))))
            elif kind == 1:
                ps.append(corpus.repeated(int(rng.integers(256)), n))
            elif kind == 2:
                unit = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)
                ps.append(np.tile(unit, n // unit.size + 1)[:n])
            elif kind == 3 and blocks:
                prev = blocks[int(rng.integers(len(blocks)))]
                o = int(rng.integers(0, max(1, prev.size - n)))
                ps.append(prev[o:o + n].copy())
            else:
                ps.append(corpus.class_bytes(names[int(rng.integers(12))], n, int(rng.integers(1000))))
            have += ps[-1].size
        blocks.append(np.concatenate(ps)[:total])
    full = [oracle.compress_hc(b, 3) for b in blocks]
    caps = [oracle.compress_bound(b.size) if rng.random() < 0.7 else int(rng.integers(0, r + 2)) for b, (r, w) in zip(blocks, full)]
    src, soff, slen = pack(blocks)
    for nseg_log2 in (1, 2):
        dst, doff, dcap = arena(caps)
        out = emu.encode_hc_batch(src, soff, slen, dst, doff, dcap, level=3, flags=FLAG_RAW | (1 << 30) | (nseg_log2 << 27))
        for i, b in enumerate(blocks):
            r, w = oracle.compress_hc(b, 3, cap=caps[i])
            assert out[i] == r, (case, i, b.size, caps[i], nseg_log2, int(out[i]), r)
            if r > 0:
                assert dst[int(doff[i]):int(doff[i]) + r].tobytes() == w[:r].tobytes(), (case, i, b.size, nseg_log2)
        nblocks += len(blocks)
print("ok", cases, "cases,", nblocks, "blocks")
