#!/usr/bin/env python
"""Measurements for the frame layer on one GPU:
  xxh32   block checksums: 4096 x 64 KiB HBM-resident buffers, and one 256 MiB stream (content checksum)
  frames  LZ4Frame.EncodeBatch / DecodeBatch through the host-pointer API (PCIe + staging inclusive)
  chain   in-order decode of linked-block frames written by liblz4, one frame per wavefront pair, HBM-resident"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from k4os.compression.lz4_amd import corpus, LZ4Frame, LZ4EncoderSettings, pack_blocks
from k4os.compression.lz4_amd import frames as F
from k4os.compression.lz4_amd.device import DeviceCodec

dc = DeviceCodec(0)
dev = dc.device
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return float(np.median(ts))

# xxh32: batch of blocks / one long stream
n, bs = 4096, 65536
blocks = corpus.silesia_like_blocks(n, bs, seed=2)
data = torch.from_numpy(blocks.reshape(-1)).to(dev)
off = torch.arange(n, dtype=torch.int64, device=dev) * bs
ln = torch.full((n,), bs, dtype=torch.int64, device=dev)
t = timed(lambda: dc.xxh32(data, off, ln))
got = dc.xxh32(data, off, ln).cpu().numpy().view(np.uint32)
import xxhash
ok = all(int(got[i]) == xxhash.xxh32(blocks[i].tobytes()).intdigest() for i in range(0, n, 97))
print(json.dumps({"config": "xxh32 block checksums", "buffers": n, "bytes_each": bs, "ms": round(t * 1e3, 3),
                  "GBs": round(n * bs / t / 1e9, 1), "frac_of_8TBs": round(n * bs / t / 8e12, 4), "sampled_equal_xxhash": ok}))
one_off = torch.zeros(1, dtype=torch.int64, device=dev); one_len = torch.full((1,), n * bs, dtype=torch.int64, device=dev)
t1 = timed(lambda: dc.xxh32(data, one_off, one_len), reps=3)
g1 = int(dc.xxh32(data, one_off, one_len).cpu().numpy().view(np.uint32)[0])
print(json.dumps({"config": "xxh32 one stream (content checksum)", "bytes": n * bs, "ms": round(t1 * 1e3, 1),
                  "GBs": round(n * bs / t1 / 1e9, 2), "equal_xxhash": g1 == xxhash.xxh32(blocks.tobytes()).intdigest()}))

# frames through the host API
contents = [blocks[i * 64:(i + 1) * 64].reshape(-1) for i in range(64)]          # 64 contents of 4 MiB
st = LZ4EncoderSettings(BlockChecksum=True)
t0 = time.perf_counter(); frames = LZ4Frame.EncodeBatch(contents, st); te = time.perf_counter() - t0
t0 = time.perf_counter(); frames = LZ4Frame.EncodeBatch(contents, st); te = min(te, time.perf_counter() - t0)
t0 = time.perf_counter(); back = LZ4Frame.DecodeBatch(frames); td = time.perf_counter() - t0
t0 = time.perf_counter(); back = LZ4Frame.DecodeBatch(frames); td = min(td, time.perf_counter() - t0)
tot = sum(c.size for c in contents)
print(json.dumps({"config": "LZ4Frame host API, 64 x 4 MiB, block checksums", "encode_GiBs": round(tot / te / 2**30, 2),
                  "decode_GiBs": round(tot / td / 2**30, 2), "ratio": round(sum(len(f) for f in frames) / tot, 4),
                  "roundtrip_ok": all(b == c.tobytes() for b, c in zip(back, contents)), "note": "pageable host memory, PCIe and python assembly inclusive"}))

# frame writer with everything in HBM
from k4os.compression.lz4_amd.frames import encode_frames_device
c_off = np.arange(64, dtype=np.int64) * (64 * bs); c_len = np.full(64, 64 * bs, np.int64)
for st_d in (LZ4EncoderSettings(), LZ4EncoderSettings(BlockChecksum=True), LZ4EncoderSettings(BlockChecksum=True, ContentChecksum=True)):
    tf = timed(lambda: encode_frames_device(dc, data, c_off, c_len, st_d), reps=5)
    fr, fo_, fl_ = encode_frames_device(dc, data, c_off, c_len, st_d); torch.cuda.synchronize()
    f0 = fr[int(fo_[0]):int(fo_[0]) + int(fl_[0])].cpu().numpy().tobytes()
    print(json.dumps({"config": "encode_frames_device, 64 contents x 4 MiB, HBM-resident", "block_checksum": st_d.BlockChecksum,
                      "content_checksum": st_d.ContentChecksum, "ms": round(tf * 1e3, 2), "GiBs": round(tot / tf / 2**30, 1),
                      "first_frame_equals_host_api": f0 == LZ4Frame.Encode(contents[0], st_d)}))

# chained decode, device resident
from test_frame_layer import LZ4F
lz = LZ4F()
nfr, size = 1024, 1 << 20
srcs = [corpus.class_bytes(corpus.SILESIA_NAMES[i % 12], size, i) for i in range(16)]
fr16 = [lz.compress(s, 4, linked=True) for s in srcs]
frames = [fr16[i % 16] for i in range(nfr)]
infos16 = [F.parse_frame(f) for f in fr16]
buf, foff, _ = pack_blocks([np.frombuffer(f, np.uint8) for f in frames])
blk_off, blk_len, first, nblk = [], [], [], []
for f in range(nfr):
    i = infos16[f % 16]
    first.append(len(blk_off)); nblk.append(len(i.block_off))
    blk_off += [int(foff[f]) + o for o in i.block_off]; blk_len += i.block_len
T = lambda a, dt: torch.from_numpy(np.asarray(a).astype(dt)).to(dev)
d_src = torch.from_numpy(buf).to(dev)
args = (d_src, T(blk_off, np.int64), T(np.array(blk_len, np.uint32).view(np.int32), np.int32), T(first, np.int64), T(nblk, np.int32),
        T([65536] * nfr, np.int32), torch.ones(nfr, dtype=torch.uint8, device=dev))
dst = torch.zeros(nfr * size, dtype=torch.uint8, device=dev)
doff = torch.arange(nfr, dtype=torch.int64, device=dev) * size
dcap = torch.full((nfr,), size, dtype=torch.int64, device=dev)
tc = timed(lambda: dc.decode_chain(*args, dst, doff, dcap), reps=3)
out = dc.decode_chain(*args, dst, doff, dcap).cpu().numpy()
okc = bool((out == size).all()) and all(bytes(dst[i * size:(i + 1) * size].cpu().numpy()) == srcs[i % 16].tobytes() for i in (0, 5, 1023))
print(json.dumps({"config": "chained (linked-block) frames, in-order decode, two wavefronts per frame (one with K4LZ4_NO_PAIR=1)", "frames": nfr, "bytes_each": size,
                  "ms": round(tc * 1e3, 1), "GiBs": round(nfr * size / tc / 2**30, 1), "roundtrip_ok": okc}))
