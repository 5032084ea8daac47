#!/usr/bin/env python
"""Randomised stress of the fast encoder kernel (and, second half, of the decoder kernel) under the host wave emulator against the oracle:
inputs built to provoke equal hashes inside one 64-position window, matches ending near window
edges, long literal runs, limited output.  Usage: tests/tools/emu_stress_encode.py [rounds] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
from emu_lib import Emu
from k4os.compression.lz4_amd import pack_blocks, make_arena, LZ4Codec, corpus

def gen(rng, n):
    kind = rng.integers(0, 8)
    if kind == 0:      # tiny alphabet
        return rng.integers(0, rng.integers(2, 5), n).astype(np.uint8)
    if kind == 1:      # periodic with noise
        per = int(rng.integers(1, 80)); base = rng.integers(0, 256, per).astype(np.uint8)
        a = np.resize(base, n).copy(); m = rng.random(n) < rng.choice([0.0, 0.01, 0.05, 0.2]); a[m] = rng.integers(0, 256, int(m.sum()))
        return a
    if kind == 2:      # phrases from a small dictionary of short words
        words = [rng.integers(97, 97 + int(rng.integers(2, 26)), int(rng.integers(2, 9))).astype(np.uint8) for _ in range(int(rng.integers(2, 40)))]
        out = []; tot = 0
        while tot < n:
            w = words[int(rng.integers(0, len(words)))]; out.append(w); tot += w.size
        return np.concatenate(out)[:n]
    if kind == 3:      # random
        return rng.integers(0, 256, n).astype(np.uint8)
    if kind == 4:      # runs
        out = []; tot = 0
        while tot < n:
            r = int(rng.integers(1, 400)); out.append(np.full(r, rng.integers(0, 256), np.uint8)); tot += r
        return np.concatenate(out)[:n]
    if kind == 5:      # text-like class
        return corpus.class_bytes(str(rng.choice(["dickens", "xml", "osdb", "mr", "nci", "samba"])), n, int(rng.integers(0, 1000)))
    if kind == 6:      # copies of earlier pieces at random distances
        a = rng.integers(0, 256, n).astype(np.uint8); i = 16
        while i < n:
            L = int(rng.integers(4, 40)); d = int(rng.integers(1, min(i, 70000) + 1))
            if rng.random() < 0.7 and i + L <= n:
                for k in range(L): a[i + k] = a[i + k - d]
            i += L + int(rng.integers(0, 6))
        return a
    # mixture
    parts = [gen(rng, max(1, n // 3)) for _ in range(3)]
    return np.concatenate(parts)[:n]

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    run(rounds, seed, Oracle(), Emu())


def run(rounds, seed, oracle, emu, verbose=True):
    rng = np.random.default_rng(seed)
    for r in range(rounds):
        blocks = []
        for _ in range(48):
            n = int(rng.choice([rng.integers(0, 40), rng.integers(13, 300), rng.integers(300, 6000), rng.integers(6000, 70000)]))
            blocks.append(gen(rng, n) if n else np.zeros(0, np.uint8))
        if r % 5 == 0:
            blocks.append(gen(rng, int(rng.integers(65547, 140000))))
        src, soff, slen = pack_blocks(blocks)
        caps = []
        for b in blocks:
            bound = LZ4Codec.MaximumOutputSize(b.size)
            caps.append(bound if rng.random() < 0.6 else int(rng.integers(0, bound + 1)))
        caps = np.array(caps, np.int32)
        d1, o1 = make_arena(caps + 16, fill=0xCD); d2, o2 = make_arena(caps + 16, fill=0xCD)
        accel = 1 if r % 4 else int(rng.integers(1, 70))
        if accel == 1:
            want = oracle.encode_batch(src, soff, slen, d2, o2, caps, threads=8)
        else:
            want = np.zeros(len(blocks), np.int32)
            for i, b in enumerate(blocks):
                n, out = oracle.compress_fast(b, int(caps[i]), accel)
                want[i] = 0 if b.size == 0 else (-1 if n <= 0 else n)
                if n > 0: d2[int(o2[i]):int(o2[i]) + n] = out[:n]
        got = emu.encode_batch(src, soff, slen, d1, o1, caps, accel=accel)
        for i in range(len(blocks)):
            assert got[i] == want[i], (r, i, blocks[i].size, caps[i], got[i], want[i], accel)
            if want[i] > 0:
                a = d1[int(o1[i]):int(o1[i]) + want[i]]; b = d2[int(o2[i]):int(o2[i]) + want[i]]
                assert np.array_equal(a, b), (r, i, blocks[i].size, int(np.argmax(a != b)))
            assert (d1[int(o1[i]) + caps[i]:int(o1[i]) + caps[i] + 16] == 0xCD).all(), (r, i, 'guard')
        if verbose:
            print("round", r, "ok", len(blocks), "blocks accel", accel, flush=True)

def run_decode(rounds, seed, oracle, emu, verbose=True):
    """The decoder kernel on the oracle's streams of the same inputs: destinations of exactly the decoded size,
    larger, and too small; streams with a flipped byte; 16 guard bytes behind every destination (the lane copies
    write overlapping words, none may leave its run); results and bytes against the oracle's safe decoder."""
    rng = np.random.default_rng(seed + 1000)
    for r in range(rounds):
        blocks = []
        for _ in range(48):
            n = int(rng.choice([rng.integers(0, 40), rng.integers(13, 300), rng.integers(300, 6000), rng.integers(6000, 70000)]))
            blocks.append(gen(rng, n) if n else np.zeros(0, np.uint8))
        comp = []
        for b in blocks:
            if b.size == 0:
                comp.append(np.zeros(0, np.uint8)); continue
            n, out = oracle.compress_fast(b, LZ4Codec.MaximumOutputSize(b.size), 1)
            c = out[:n].copy()
            if rng.random() < 0.15 and n > 4:                 # hostile: one byte changed
                c[int(rng.integers(0, n))] ^= int(rng.integers(1, 256))
            comp.append(c)
        src, soff, slen = pack_blocks(comp)
        caps = []
        for b in blocks:
            k = rng.random()
            caps.append(b.size if k < 0.6 else (b.size + int(rng.integers(1, 100)) if k < 0.8 else int(rng.integers(0, b.size + 1))))
        caps = np.array(caps, np.int32)
        d1, o1 = make_arena(caps + 16, fill=0xCD)
        got = emu.decode_batch(src, soff, slen, d1, o1, caps, flags=1)    # raw engine results
        for i, b in enumerate(blocks):
            if comp[i].size == 0:
                continue
            n, ref = oracle.decompress_safe(comp[i], int(caps[i]))
            assert got[i] == n, (r, i, b.size, int(caps[i]), int(got[i]), n)
            if n > 0:
                a = d1[int(o1[i]):int(o1[i]) + n]
                assert np.array_equal(a, ref[:n]), (r, i, b.size, int(np.argmax(a != ref[:n])))
            assert (d1[int(o1[i]) + caps[i]:int(o1[i]) + caps[i] + 16] == 0xCD).all(), (r, i, 'guard')
        if verbose:
            print("decode round", r, "ok", len(blocks), "blocks", flush=True)


if __name__ == "__main__":
    main()
    run_decode(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1, Oracle(), Emu())
