#!/usr/bin/env python
"""Does an L00_FAST encoder (byU32 table, hash5: blocks of 64 KiB + 11 bytes and more, LL64.fast.cs:526-544) that is started
`warm` bytes before a boundary with an EMPTY table fall into step with the true run by the boundary?  "In step" = the true
run ends a match exactly where the warm run ended its first match at or after the boundary, and the two hash tables agree
in every entry that can still be used (entries more than 65 535 bytes back are rejected by both, LL64.fast.cs:219-224).
From such a point on the two runs are the same run, byte for byte.  Usage: python probe.py   (builds spec.c with gcc)"""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from k4os.compression.lz4_amd import corpus
so = os.path.join("/tmp", "libk4spec.so")
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "spec.c")])
lib = C.CDLL(so)
lib.probe.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]


def run(data, seg, warm):
    ok = np.zeros(256, np.int32); q = np.zeros(256, np.int64); land = np.zeros(256, np.int64)
    nb = lib.probe(data.ctypes.data, data.size, seg, warm, ok.ctypes.data, q.ctypes.data, land.ctypes.data, 256)
    return ok[:nb]


sets = {"text 8 MiB": corpus.class_bytes("dickens", 8 << 20, 5)}
for nm in ("xml", "nci", "samba", "osdb", "webster", "reymont", "mozilla", "ooffice"):
    sets[nm + " 4 MiB"] = corpus.class_bytes(nm, 4 << 20, 3)
sets["12 classes x 512 KiB"] = np.concatenate([corpus.class_bytes(nm, 1 << 19, 9) for nm in corpus.SILESIA_NAMES])
warms = (65536, 131072, 196608, 262144, 393216)
print("boundaries every 512 KiB; in step / boundaries, by warm-up length")
print("%-22s " % "data" + " ".join("%8s" % f"{w >> 10} KiB" for w in warms))
for name, d in sets.items():
    cells = []
    for w in warms:
        ok = run(d, 524288, w)
        cells.append("%8s" % f"{int((ok == 1).sum())}/{len(ok)}")
    print("%-22s " % name + " ".join(cells))
