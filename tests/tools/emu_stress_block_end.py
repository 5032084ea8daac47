#!/usr/bin/env python
"""Randomised stress of the two-step fast encoder's last rounds under the emulator: blocks that end in a long literal run with something
matchable near the very end -- the search's 66-probe limit, its growing step and mflimitPlusOne / matchlimit all meet there
(LL64.fast.cs:156-172, :391, :469-503).  Text, a repeated phrase, 40 .. 200 bytes that match nothing (now and then with a
matchable fragment inside), then 4 .. 40 bytes copied from the text.  Usage: tests/tools/emu_stress_block_end.py [blocks] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
from emu_lib import Emu
from k4os.compression.lz4_amd import pack_blocks, make_arena, LZ4Codec, corpus


def block(rng):
    text = corpus.lorem(int(rng.integers(200, 3000)))
    a = int(rng.integers(0, text.size - 60)); rep = text[a:a + int(rng.integers(8, 60))]
    noise = rng.integers(128, 256, int(rng.integers(40, 200)), dtype=np.uint8)
    noise[0] = 0xFF
    if rng.random() < 0.3:
        at = int(rng.integers(1, noise.size - 8)); b = int(rng.integers(0, text.size - 8))
        noise[at:at + 6] = text[b:b + 6]
    t = int(rng.integers(0, text.size - 44)); tail = text[t:t + int(rng.integers(4, 41))]
    return np.concatenate([text, rep, noise, tail]).astype(np.uint8)


def run(nblocks, seed, oracle=None, emu=None):
    oracle = oracle or Oracle(); emu = emu or Emu()
    rng = np.random.default_rng(seed)
    bad = 0
    for lo in range(0, nblocks, 256):
        blocks = [block(rng) for _ in range(min(256, nblocks - lo))]
        src, soff, slen = pack_blocks(blocks)
        caps = np.array([LZ4Codec.MaximumOutputSize(b.size) for b in blocks], np.int32)
        d, do = make_arena(caps + 16, fill=0xCD)
        got, _ = emu.encode_parse_batch(src, soff, slen, d, do, caps, k=int(rng.choice([1, 1, 2])), waves=16, inline_emit=True, migrate=True)
        for i, b in enumerate(blocks):
            want = oracle.encode(b)
            if got[i] != len(want) or bytes(d[int(do[i]):int(do[i]) + len(want)]) != want:
                bad += 1
                print(f"block {lo + i} len {b.size} want {len(want)} got {got[i]}")
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t = time.time()
    bad = run(n, seed)
    print(f"seed {seed}: {n} blocks, {bad} failures, {time.time() - t:.0f}s")
    sys.exit(1 if bad else 0)
