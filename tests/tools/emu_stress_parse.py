#!/usr/bin/env python
"""Randomised stress of the two-step fast encoder (k4lz4_parse.hpp: parse + write-out, then the one-kernel encoder for the blocks it
leaves alone) under the host wave emulator against the oracle: the inputs of emu_stress_encode.py (equal hashes inside a window,
matches ending at window edges, long literal runs, incompressible stretches that make the step grow), ragged output limits, another
acceleration every sixth round, and a random configuration per round -- sub-windows per round, waves per workgroup (tables in LDS
and in memory), dispatch order, write-out by the parsing wave or by k4_emit_kernel, tables moving into LDS.
Usage: tests/tools/emu_stress_parse.py [rounds] [seed]     (K4_FIX_K=1 pins the number of sub-windows)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
from oracle_lib import Oracle
from emu_lib import Emu
from emu_stress_encode import gen
from k4os.compression.lz4_amd import pack_blocks, make_arena, LZ4Codec


def run(rounds, seed, oracle, emu, verbose=True):
    rng = np.random.default_rng(seed)
    bad = 0
    for r in range(rounds):
        blocks = []
        for _ in range(48):
            n = int(rng.choice([rng.integers(0, 40), rng.integers(100, 400), rng.integers(300, 6000), rng.integers(6000, 65547), rng.integers(64000, 65547)]))
            blocks.append(gen(rng, n) if n else np.zeros(0, np.uint8))
        if r % 5 == 0:
            blocks.append(gen(rng, int(rng.integers(65547, 100000))))
        src, soff, slen = pack_blocks(blocks)
        caps = []
        for b in blocks:
            bound = LZ4Codec.MaximumOutputSize(b.size)
            caps.append(bound if rng.random() < 0.6 else int(rng.integers(0, bound + 1)))
        caps = np.array(caps, np.int32)
        d1, o1 = make_arena(caps + 16, fill=0xCD); d2, o2 = make_arena(caps + 16, fill=0xCD)
        accel = 1 if r % 6 else 3
        if accel == 1:
            want = oracle.encode_batch(src, soff, slen, d2, o2, caps, threads=8)
        else:
            want = np.zeros(len(blocks), np.int32)
            for i, b in enumerate(blocks):
                n, out = oracle.compress_fast(b, int(caps[i]), accel)
                want[i] = 0 if b.size == 0 else (-1 if n <= 0 else n)
                if n > 0: d2[int(o2[i]):int(o2[i]) + n] = out[:n]
        k = int(os.environ.get("K4_FIX_K", rng.choice([1, 1, 2, 3, 4])))
        waves = int(rng.choice([1, 4, 9, 12, 16]))
        order = rng.permutation(len(blocks)).astype(np.uint32) if rng.random() < 0.5 else None
        got, _ = emu.encode_parse_batch(src, soff, slen, d1, o1, caps, accel=accel, k=k, waves=waves, order=order,
                                        inline_emit=rng.random() < 0.7, migrate=waves > 9, queue=rng.random() < 0.15)
        for i in range(len(blocks)):
            ok = got[i] == want[i]
            if ok and want[i] > 0:
                ok = bytes(d1[int(o1[i]):int(o1[i]) + want[i]]) == bytes(d2[int(o2[i]):int(o2[i]) + want[i]]) and \
                     (d1[int(o1[i]) + want[i]:int(o1[i]) + caps[i] + 16] == 0xCD).all()
            if not ok:
                bad += 1
                if verbose: print(f"round {r} k={k} waves={waves} block {i} len {blocks[i].size} cap {caps[i]} want {want[i]} got {got[i]}")
    return bad


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t = time.time()
    bad = run(rounds, seed, Oracle(), Emu())
    print(f"seed {seed}: {rounds} rounds, {bad} failures, {time.time() - t:.0f}s")
    sys.exit(1 if bad else 0)
