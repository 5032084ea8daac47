#!/usr/bin/env python
"""BASELINE.json configs[2]: decode-only, N pre-compressed 4 KiB blocks on one MI355X (HBM-resident),
text-like mix and random-bytes variants.  Prints GiB/s and achieved algorithmic GB/s."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from k4os.compression.lz4_amd import LZ4Codec, corpus
from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec

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
    # device-side tiling of the unique blocks; a per-block tag keeps every block distinct
    dev = torch.from_numpy(base).to(dc.device)
    data = dev.repeat(reps, 1)[:n].contiguous()
    tag = torch.arange(n, device=dc.device, dtype=torch.int64).view(-1, 1)
    data[:, 16:24] = ((tag >> (8 * torch.arange(8, device=dc.device))) & 0xFF).to(torch.uint8)
    src = DeviceBatch(data.view(-1), torch.from_numpy(off.view(np.int64)).to(dc.device), torch.from_numpy(lens).to(dc.device))
    bound = LZ4Codec.MaximumOutputSize(bs)
    comp = DeviceBatch.empty_slots(np.full(n, bound), dc.device)
    clen = dc.encode(src, comp)
    torch.cuda.synchronize()
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
    te = time.perf_counter(); dc.encode(src, comp, clen); torch.cuda.synchronize(); enc_ms = (time.perf_counter() - te) * 1e3
    ok = bool((dlen == bs).all().item()) and bool(torch.equal(back.data[:n * bs], src.data[:n * bs]))
    sum_c = int(clen.sum().item()); sum_u = n * bs
    alg = sum_u + sum_c + 12 * n
    print(json.dumps({"config": "configs[2] decode-only", "variant": variant, "blocks": n, "block_bytes": bs, "ratio": round(sum_c / sum_u, 4),
                      "decode_ms_median": round(ms, 3), "decode_GiBs": round(sum_u / 2 ** 30 / (ms * 1e-3), 2),
                      "achieved_GBs": round(alg / (ms * 1e-3) / 1e9, 1), "frac_of_8TBs": round(alg / (ms * 1e-3) / 8e12, 4),
                      "encode_ms_once": round(enc_ms, 2), "roundtrip_ok": ok}), flush=True)
    del src, comp, back, data, dev
    torch.cuda.empty_cache()
