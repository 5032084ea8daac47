#!/usr/bin/env python
"""GPU vs oracle per-block diff of the fast encoder on the bench workload, per dispatch variant."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
from k4os.compression.lz4_amd import LZ4Codec, corpus, make_arena

n, bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 65536
blocks = corpus.silesia_like_blocks(n, bs, seed=2)
src = blocks.reshape(-1); off = np.arange(n, dtype=np.uint64) * bs; lens = np.full(n, bs, np.int32)
caps = np.full(n, LZ4Codec.MaximumOutputSize(bs), np.int32)
o = Oracle()
rd, ro = make_arena(caps, fill=0)
want = o.encode_batch(src, off, lens, rd, ro, caps, threads=16)
for name, flags, env in (("nosplit", 16, None), ("noreorder", 4, None), ("default", 0, None), ("gtab99", 0, "1")):
    if env: os.environ["K4LZ4_SPLIT_PCT"] = env
    for rep in range(2):
        d, do = make_arena(caps, fill=0)
        got = LZ4Codec.EncodeBatchPacked(src, off, lens, d, do, caps, flags=flags)
        bad = [i for i in range(n) if got[i] != want[i] or not np.array_equal(d[int(do[i]):int(do[i]) + want[i]], rd[int(ro[i]):int(ro[i]) + want[i]])]
        print(name, "rep", rep, "mismatching blocks", len(bad), bad[:10])
        for i in bad[:3]:
            a = d[int(do[i]):int(do[i]) + max(got[i], 0)]; b = rd[int(ro[i]):int(ro[i]) + want[i]]
            m = min(a.size, b.size); k = int(np.argmax(a[:m] != b[:m])) if (a[:m] != b[:m]).any() else m
            print("   block", i, "got", got[i], "want", want[i], "first diff at", k, a[max(0, k - 4):k + 8].tolist(), b[max(0, k - 4):k + 8].tolist())
