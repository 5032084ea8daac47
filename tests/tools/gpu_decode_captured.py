#!/usr/bin/env python
"""Decode the streams captured by gpu_stress_all.py (gpurun_out/stress_fail/*.npz) and tests/golden/mutated_stream_*.npy on DEVICE
buffers pre-filled with 0xCD and compare every byte of the slot with the oracle's (which decoded into a 0xCD-filled slot too)."""
import glob, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
from k4os.compression.lz4_amd import pack_blocks, make_arena
from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
o = Oracle(); dc = DeviceCodec(0)
items = []
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "stress_fail", "*.npz"))):
    z = np.load(f); items.append((os.path.basename(f), z["stream"], int(z["cap"])))
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mutated_stream_*.npy"))):
    items.append((os.path.basename(f), np.load(f), 5813))
src, soff, slen = pack_blocks([s for _, s, _ in items])
caps = np.array([c for _, _, c in items], np.int32)
d2, o2 = make_arena(caps + 16, fill=0xCD)
want = o.decode_batch(src, soff, slen, d2, o2, caps)
sb = DeviceBatch.from_host(src, soff, slen, dc.device)
db = DeviceBatch(torch.full((d2.size,), 0xCD, dtype=torch.uint8, device=dc.device), torch.from_numpy(o2.view(np.int64)).to(dc.device), torch.from_numpy(caps).to(dc.device))
got = dc.decode(sb, db).cpu().numpy()
d1 = db.data.cpu().numpy()
for i, (name, _, cap) in enumerate(items):
    a = int(o2[i])
    print(name, "want", want[i], "got", got[i], "differing bytes in the slot", int((d1[a:a + cap + 16] != d2[a:a + cap + 16]).sum()))
