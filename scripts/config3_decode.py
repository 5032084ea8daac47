#!/usr/bin/env python
"""BASELINE.json configs[2]: decode-only, N pre-compressed 4 KiB blocks on one MI355X (HBM-resident),
text-like mix and random-bytes variants.  The compressed blocks come from the ORACLE (CPU), not from the GPU encoder, and
the decoded bytes are compared with the source: the path is never checked against itself.  Prints GiB/s and achieved
algorithmic GB/s."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
from k4os.compression.lz4_amd import LZ4Codec, corpus, make_arena
from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
oracle = Oracle()

n = int(os.environ.get("K4_BLOCKS", str(1 << 20)))
bs = 4096
dc = DeviceCodec(0)
for variant in ("silesia-like", "random"):
    if variant == "random":
        base = np.random.default_rng(7).integers(0, 256, size=(min(n, 1 << 16), bs), dtype=np.uint8)
    else:
        base = corpus.silesia_like_blocks(min(n, 1 << 16), bs, seed=3)
    reps = -(-n // base.shape[0])
    lens = np.full(n, bs, np.int32)
    off = np.arange(n, dtype=np.uint64) * bs
    # every block distinct (a per-block tag), compressed on the host by the oracle
    blocks = np.tile(base, (reps, 1))[:n].copy()
    blocks[:, 16:24] = np.arange(n, dtype="<u8").view(np.uint8).reshape(n, 8)
    bound = LZ4Codec.MaximumOutputSize(bs)
    caps = np.full(n, bound, np.int32)
    ref, ref_off = make_arena(caps)
    clen_h = oracle.encode_batch(blocks.reshape(-1), off, lens, ref, ref_off, caps, threads=os.cpu_count() or 8)
    comp = DeviceBatch.from_host(ref, ref_off, clen_h, dc.device)
    clen = comp.length
    src = DeviceBatch.from_host(blocks.reshape(-1), off, lens, dc.device)
    back = DeviceBatch.empty_slots(lens, dc.device)
    csrc = DeviceBatch(comp.data, comp.off, clen)
    dlen = dc.new_out_len(n)
    for _ in range(2):
        dc.decode(csrc, back, dlen)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    for i in range(10):
        ev[i].record(); dc.decode(csrc, back, dlen)
    ev[10].record(); torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(10))[5]
    ok = bool((dlen == bs).all().item()) and bool(torch.equal(back.data[:n * bs], src.data[:n * bs]))
    sum_c = int(clen.sum().item()); sum_u = n * bs
    alg = sum_u + sum_c + 12 * n
    print(json.dumps({"config": "configs[2] decode-only", "variant": variant, "blocks": n, "block_bytes": bs, "ratio": round(sum_c / sum_u, 4),
                      "decode_ms_median": round(ms, 3), "decode_GiBs": round(sum_u / 2 ** 30 / (ms * 1e-3), 2),
                      "achieved_GBs": round(alg / (ms * 1e-3) / 1e9, 1), "frac_of_8TBs": round(alg / (ms * 1e-3) / 8e12, 4),
                      "input": "oracle-encoded", "decoded_equals_source": ok}), flush=True)
    del src, comp, back, blocks
    torch.cuda.empty_cache()
