#!/usr/bin/env python
"""How long the matches of the bench corpus' classes are (from the oracle's encodings): what share of the sequences runs past the
12 / 28 / 44 bytes a parse round knows behind a probe (k4lz4_parse.hpp: the 28-known-bytes form of the round).
Usage: tests/tools/match_length_stats.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from k4os.compression.lz4_amd import corpus
from oracle_lib import Oracle


def sequences(buf):
    """(literal length, match length) of every sequence of an LZ4 block"""
    i, n, out = 0, len(buf), []
    while i < n:
        t = buf[i]; i += 1
        ll = t >> 4
        if ll == 15:
            while True:
                v = buf[i]; i += 1; ll += v
                if v != 255: break
        i += ll
        if i >= n: break
        i += 2
        ml = t & 15
        if ml == 15:
            while True:
                v = buf[i]; i += 1; ml += v
                if v != 255: break
        out.append((ll, ml + 4))
    return out


if __name__ == "__main__":
    o = Oracle()
    blocks = corpus.silesia_like_blocks(48, 65536, seed=2)
    print("class     seq/block  mean match  mean literals   e>12    e>28    e>44   (e = match length - 4)")
    for ci, name in enumerate(corpus.SILESIA_NAMES):
        s = [q for b in range(ci, 48, 12) for q in sequences(o.encode(blocks[b]))]
        if not s:
            print(f"{name:8s} {0:9d}"); continue
        e = np.array([m for _, m in s]) - 4
        print(f"{name:8s} {len(s) // 4:9d} {e.mean() + 4:10.1f} {np.mean([l for l, _ in s]):13.1f} {np.mean(e > 12) * 100:7.1f}% {np.mean(e > 28) * 100:6.1f}% {np.mean(e > 44) * 100:6.1f}%")
