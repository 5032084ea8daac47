import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.init()
from k4os.compression.lz4_amd import _native
_native._lib = _native.load_library(os.path.join(ROOT, "k4os/compression/lz4_amd/libk4lz4_dbg.so"))
import torch
from k4os.compression.lz4_amd import LZ4Codec, corpus
from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
blocks = corpus.silesia_like_blocks(1024, 65536, seed=2)
b = blocks[int(sys.argv[1]) if len(sys.argv) > 1 else 6]
dc = DeviceCodec(0)
src = DeviceBatch.from_host(b, np.zeros(1, np.uint64), np.array([b.size], np.int32), dc.device)
comp = DeviceBatch.empty_slots(np.array([LZ4Codec.MaximumOutputSize(b.size)]), dc.device)
clen = dc.new_out_len(1)
dc.encode(src, comp, clen)
torch.cuda.synchronize()
t = comp.data.cpu().numpy()
print("n", clen.cpu().numpy(), t[:20].tolist())
d = t[40000:40000 + 8 * 128].view(np.uint32).reshape(8, 32)
names = "ip0 q f p jf match inf lit0 back code dirtyL dirtyH hmL hmH IL IH GL GH jl info h cand pos nextL nextH pre".split()
dec = ('ip0', 'q', 'f', 'p', 'match', 'lit0', 'back', 'code', 'h', 'cand', 'pos')
for r in d:
    print(" ".join(f"{n}={int(v) if n in dec else hex(int(v))}" for n, v in zip(names, r)))
