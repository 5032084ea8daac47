#!/usr/bin/env python
"""Randomised stress of the fast encoder ON THE GPU against the oracle: batches of a few thousand ragged blocks (0 .. 100 000 bytes;
the adversarial inputs of emu_stress_encode.py -- equal hashes inside a window, matches ending at window edges, long literal runs,
incompressible stretches -- mixed with the corpus classes), ragged output limits, through the default path (k4_parse_kernel with
tables that move into LDS, then the one-kernel encoder for the blocks it leaves alone).  Every block: return value, bytes, and the
slack behind them untouched.  Usage: tests/tools/gpu_stress_encode.py [rounds] [seed] [blocks per round] [device] [big]     (K4_STRESS_REPEATS=n: every batch n times)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
from oracle_lib import Oracle
from emu_stress_encode import gen
from k4os.compression.lz4_amd import pack_blocks, make_arena, LZ4Codec, corpus


def run(rounds, seed, per, oracle=None, device=False, repeats=1, big=False):
    """device: through k4lz4_encode_batch_device on HBM-resident buffers (no host staging); repeats: every batch that many times;
    big: a third of the blocks of 65 547 .. 400 000 bytes (byU32 tables: k4_parse_big_kernel, round 6)"""
    oracle = oracle or Oracle()
    if device:
        import torch
        from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
        dc = DeviceCodec(0)
    rng = np.random.default_rng(seed)
    bad = total = 0
    t = time.time()
    for r in range(rounds):
        blocks = []
        for i in range(per):
            n = int(rng.choice([rng.integers(0, 40), rng.integers(100, 400), rng.integers(300, 6000), rng.integers(6000, 65547), rng.integers(64000, 65547), 65536]))
            if big and i % 3 == 0: n = int(rng.choice([rng.integers(65547, 66000), rng.integers(65547, 140000), rng.integers(100000, 400000)]))
            if n == 0: blocks.append(np.zeros(0, np.uint8))
            elif rng.random() < 0.5: blocks.append(gen(rng, n))
            else: blocks.append(corpus.class_bytes(corpus.SILESIA_NAMES[int(rng.integers(0, 12))], n, int(rng.integers(0, 1 << 30))))
        if r % 3 == 0: blocks.append(gen(rng, int(rng.integers(65547, 100000))))
        src, soff, slen = pack_blocks(blocks)
        caps = np.array([LZ4Codec.MaximumOutputSize(b.size) if rng.random() < 0.7 else int(rng.integers(0, LZ4Codec.MaximumOutputSize(b.size) + 1)) for b in blocks], np.int32)
        d1, o1 = make_arena(caps + 16, fill=0xCD); d2, o2 = make_arena(caps + 16, fill=0xCD)
        want = oracle.encode_batch(src, soff, slen, d2, o2, caps, threads=32)
        for rep in range(repeats):
            if device:
                sb = DeviceBatch.from_host(src, soff, slen, dc.device)
                db = DeviceBatch(torch.full((d1.size,), 0xCD, dtype=torch.uint8, device=dc.device), torch.from_numpy(o1.view(np.int64)).to(dc.device),
                                 torch.from_numpy(caps).to(dc.device))
                got = dc.encode(sb, db).cpu().numpy()
                d1 = db.data.cpu().numpy()
            else:
                d1[:] = 0xCD
                got = LZ4Codec.EncodeBatchPacked(src, soff, slen, d1, o1, caps)
            for i in range(len(blocks)):
                ok = got[i] == want[i]
                why = "length"
                if ok and want[i] > 0:
                    a, b = int(o1[i]), int(o2[i])
                    same = bytes(d1[a:a + want[i]]) == bytes(d2[b:b + want[i]])
                    clean = bool((d1[a + want[i]:a + caps[i] + 16] == 0xCD).all())
                    ok, why = same and clean, ("bytes" if not same else "slack")
                    if not same:
                        x, y = d1[a:a + want[i]], d2[b:b + want[i]]
                        k = int(np.nonzero(x != y)[0][0]); why += f" first at {k} of {want[i]}, {int((x != y).sum())} bytes differ: got {bytes(x[k:k + 12]).hex()} want {bytes(y[k:k + 12]).hex()}"
                if not ok:
                    bad += 1
                    print(f"round {r} rep {rep} block {i} len {blocks[i].size} cap {caps[i]} want {want[i]} got {got[i]}: {why}")
        total += len(blocks)
    print(f"seed {seed}: {rounds} rounds, {total} blocks, {bad} failures, {time.time() - t:.0f}s")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                      int(sys.argv[3]) if len(sys.argv) > 3 else 3000, device="device" in sys.argv[4:],
                      repeats=int(os.environ.get("K4_STRESS_REPEATS", "1")), big="big" in sys.argv[4:]) else 0)
