#!/usr/bin/env python
"""BASELINE.json configs[4]: L03_HC encode of N x 64 KiB Silesia-like blocks on one MI355X: ratio (must
equal the oracle's exactly) and GiB/s next to the oracle HC on the host cores."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from k4os.compression.lz4_amd import LZ4Codec, LZ4Level, corpus, make_arena
from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
from oracle_lib import Oracle

n, bs = int(os.environ.get("K4_BLOCKS", "4096")), 65536
level = int(os.environ.get("K4_LEVEL", "3"))
blocks = corpus.silesia_like_blocks(n, bs, seed=2)
dc = DeviceCodec(0)
lens = np.full(n, bs, np.int32); off = np.arange(n, dtype=np.uint64) * bs
src = DeviceBatch.from_host(blocks.reshape(-1), off, lens, dc.device)
bound = LZ4Codec.MaximumOutputSize(bs)
comp = DeviceBatch.empty_slots(np.full(n, bound), dc.device)
clen = dc.new_out_len(n)
ts = []
for _ in range(4):
    torch.cuda.synchronize(); t = time.perf_counter(); dc.encode(src, comp, clen, level=LZ4Level(level)); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
gpu_s = min(ts[1:])
clen_h = clen.cpu().numpy()
oracle = Oracle()
caps = np.full(n, bound, np.int32)
ref_dst, ref_off = make_arena(caps)
threads = os.cpu_count() or 1
t = time.perf_counter(); ref_len = oracle.encode_batch(blocks.reshape(-1), off, lens, ref_dst, ref_off, caps, level=level, threads=threads); cpu_s = time.perf_counter() - t
comp_h = comp.data.cpu().numpy(); coff = comp.off.cpu().numpy()
exact = bool(np.array_equal(ref_len, clen_h)) and all(
    np.array_equal(comp_h[coff[i]:coff[i] + clen_h[i]], ref_dst[int(ref_off[i]):int(ref_off[i]) + int(ref_len[i])]) for i in range(n))
back = DeviceBatch.empty_slots(lens, dc.device)
dlen = dc.decode(DeviceBatch(comp.data, comp.off, clen), back); torch.cuda.synchronize()
rt = bool((dlen == bs).all().item()) and bool(torch.equal(back.data[:n * bs], src.data[:n * bs]))
gib = n * bs / 2 ** 30
print(json.dumps({"config": "configs[4] HC encode", "level": level, "blocks": n, "block_bytes": bs,
                  "ratio_gpu": round(int(clen_h.sum()) / (n * bs), 5), "ratio_oracle": round(int(ref_len.sum()) / (n * bs), 5),
                  "bit_exact_all_blocks": exact, "decode_roundtrip": rt, "gpu_ms": round(gpu_s * 1e3, 2), "gpu_GiBs": round(gib / gpu_s, 2),
                  "cpu_oracle_GiBs": round(gib / cpu_s, 2), "cpu_threads": threads}))
