#!/usr/bin/env python
"""Input for scripts/ubench/parse_only: one oracle-compressed 64 KiB block per Silesia-like class."""
import os, sys, struct
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import Oracle
from k4os.compression.lz4_amd import corpus
o = Oracle()
blocks = [np.frombuffer(o.encode(corpus.class_bytes(name, 131072, 2)[65536:]), np.uint8) for name in corpus.SILESIA_NAMES]
with open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/parse_only.bin", "wb") as f:
    f.write(struct.pack("<I", len(blocks)))
    f.write(np.array([b.size for b in blocks], np.uint32).tobytes())
    for b in blocks:
        f.write(b.tobytes())
print(" ".join(f"{n}:{b.size}" for n, b in zip(corpus.SILESIA_NAMES, blocks)))
