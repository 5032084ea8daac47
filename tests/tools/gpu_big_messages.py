#!/usr/bin/env python
"""Big messages through LZ4Pickler.Pickle on the GPU (k4lz4_segments.hpp): every envelope byte for byte the oracle's, time of the
batch and of its largest message alone, with and without the segments (K4LZ4_NO_SEGMENTS=1 in a second process)."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from k4os.compression.lz4_amd import corpus
from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
from oracle_lib import Oracle

names = ["dickens", "xml", "samba", "webster", "nci", "mozilla", "reymont", "osdb"]
sizes = [4 << 20, 3 << 20, (5 << 19) + 12345, 2 << 20, 1600000, 4 << 20, 1 << 20, 3500000, 700000, 65536 + 4096, 2500000, 1000]
msgs = [corpus.class_bytes(names[i % len(names)], s, 7 + i) if i != 5 else corpus.random_bytes(s, 3) for i, s in enumerate(sizes)]
msgs.append(np.concatenate([corpus.class_bytes("dickens", 1500000, 1), corpus.random_bytes(900000, 4), corpus.class_bytes("dickens", 1200000, 2)]))
lens = np.array([m.size for m in msgs], np.int32)
off = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.uint64)
data = np.concatenate(msgs)
dc = DeviceCodec(0)
src = DeviceBatch.from_host(data, off, lens, dc.device)
env = DeviceBatch.empty_slots(lens.astype(np.int64) + 5, dc.device)
plen = dc.new_out_len(lens.size)
def timed(fn, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return min(ts)
t_all = timed(lambda: dc.pickle(src, env, plen))
oracle = Oracle()
eh = env.data.cpu().numpy(); eoff = env.off.cpu().numpy(); pl = plen.cpu().numpy()
bad = [i for i, m in enumerate(msgs) if eh[eoff[i]:eoff[i] + pl[i]].tobytes() != oracle.pickle(m)]
one = DeviceBatch.from_host(msgs[0], np.zeros(1, np.uint64), lens[:1], dc.device)
env1 = DeviceBatch.empty_slots(lens[:1].astype(np.int64) + 5, dc.device)
p1 = dc.new_out_len(1)
t_one = timed(lambda: dc.pickle(one, env1, p1))
e1 = env1.data.cpu().numpy(); o1 = int(env1.off.cpu().numpy()[0]); l1 = int(p1.cpu().numpy()[0])
ok1 = e1[o1:o1 + l1].tobytes() == oracle.pickle(msgs[0])
print(json.dumps({"messages": int(lens.size), "bytes": int(lens.sum()), "segments": os.environ.get("K4LZ4_NO_SEGMENTS") is None,
                  "batch_pickle_ms": round(t_all * 1e3, 2), "envelopes_differing_from_oracle": bad,
                  "one_4MiB_text_message_ms": round(t_one * 1e3, 2), "that_envelope_equals_oracle": ok1}))
