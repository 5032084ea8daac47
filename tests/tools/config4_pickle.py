#!/usr/bin/env python
"""BASELINE.json configs[3] (one GPU's share): LZ4Pickler.Pickle / Unpickle over variable-length
(1 KiB - 4 MiB, log-uniform) random/text messages, HBM-resident, byte-balanced range of the batch
this rank would own (sharding.byte_balanced_ranges).  Envelopes are compared with the oracle for a
sample; sizes for all."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from k4os.compression.lz4_amd import corpus
from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
from k4os.compression.lz4_amd.sharding import byte_balanced_ranges
from oracle_lib import Oracle

n_total = int(os.environ.get("K4_MSGS", "100000"))
world = int(os.environ.get("K4_WORLD", "8"))
rank = int(os.environ.get("K4_RANK", "0"))
lens_all = corpus.config4_lengths(n_total)
lo, hi = byte_balanced_ranges(lens_all, world)[rank]
data, off, lens = corpus.config4_share(lens_all, lo, hi)
n = lens.size
total = int(lens.astype(np.int64).sum())
dc = DeviceCodec(0)
src = DeviceBatch.from_host(data, off, lens, dc.device)
env = DeviceBatch.empty_slots(lens.astype(np.int64) + 5, dc.device)
plen = dc.new_out_len(n)
def timed(fn, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return min(ts)
t_p = timed(lambda: dc.pickle(src, env, plen))
psrc = DeviceBatch(env.data, env.off, plen)
sizes = dc.unpickle_sizes(psrc); torch.cuda.synchronize()
assert bool((sizes == torch.from_numpy(lens).to(dc.device)).all().item())
back = DeviceBatch.empty_slots(lens, dc.device)
ulen = dc.new_out_len(n)
t_u = timed(lambda: dc.unpickle(psrc, back, ulen))
ok = bool((ulen == sizes).all().item())
bh = back.data.cpu().numpy(); boff = back.off.cpu().numpy()
ok = ok and all(np.array_equal(bh[boff[i]:boff[i] + lens[i]], data[int(off[i]):int(off[i]) + int(lens[i])]) for i in range(0, n, max(1, n // 200)))
oracle = Oracle()
eh = env.data.cpu().numpy(); eoff = env.off.cpu().numpy(); pl = plen.cpu().numpy()
exact = all(eh[eoff[i]:eoff[i] + pl[i]].tobytes() == oracle.pickle(data[int(off[i]):int(off[i]) + int(lens[i])]) for i in range(0, n, max(1, n // 60)))
gib = total / 2 ** 30
print(json.dumps({"config": "configs[3] pickle/unpickle, one GPU's share", "world": world, "rank": rank, "messages": n, "bytes": total,
                  "pickle_ms": round(t_p * 1e3, 1), "pickle_GiBs": round(gib / t_p, 2), "unpickle_ms": round(t_u * 1e3, 1),
                  "unpickle_GiBs": round(gib / t_u, 2), "envelope_bytes": int(pl.sum()), "sampled_envelopes_equal_oracle": exact,
                  "roundtrip_ok": ok}))
