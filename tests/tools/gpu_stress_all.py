#!/usr/bin/env python
"""Randomised stress of the other batch paths ON THE GPU against the oracle (the fast encoder has gpu_stress_encode.py):
  decode   oracle-encoded ragged blocks into exact / oversize / short targets, through host pointers: lengths, error returns, bytes
  mutate   the same streams with random bytes flipped / truncated: return value, and the bytes where the oracle succeeds
  pickle   ragged messages (tiny ... 1 MiB) through LZ4Pickler.PickleBatch (both header rules) and back
  hc       small ragged batches at levels 3 and 9
  sizes    host-pointer encode + decode of equal blocks with the batch's total length swept around the 16 MiB staging chunks
  bigpickle / flags / many / frames / partial / dict / hclevels / envelopes   (not in the default set) messages up to 4 MiB through the segment path; FLAG_X32, FLAG_ALLOW_COPY,
           FLAG_NO_REORDER; 9 000 .. 14 000 small blocks in one call
Usage: tests/tools/gpu_stress_all.py [rounds] [seed] [which ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
from oracle_lib import Oracle
from emu_stress_encode import gen
from k4os.compression.lz4_amd import pack_blocks, make_arena, LZ4Codec, LZ4Pickler, LZ4Level, corpus


def blocks_of(rng, count, hi=65547):
    out = []
    for _ in range(count):
        n = int(rng.choice([rng.integers(0, 40), rng.integers(100, 400), rng.integers(300, 6000), rng.integers(6000, hi), rng.integers(max(1, hi - 1500), hi)]))
        if n == 0: out.append(np.zeros(0, np.uint8))
        elif rng.random() < 0.5: out.append(gen(rng, n))
        else: out.append(corpus.class_bytes(corpus.SILESIA_NAMES[int(rng.integers(0, 12))], n, int(rng.integers(0, 1 << 30))))
    return out


def check(name, r, i, ok, msg):
    if not ok: print(f"{name} round {r} item {i}: {msg}")
    return 0 if ok else 1


def decode_round(rng, oracle, r, mutate):
    bad = 0
    blocks = blocks_of(rng, 1500)
    enc = [np.frombuffer(oracle.encode(b), np.uint8) if b.size else np.zeros(0, np.uint8) for b in blocks]
    if mutate:
        enc2 = []
        for e in enc:
            e = e.copy()
            if e.size and rng.random() < 0.8:
                for _ in range(int(rng.integers(1, 4))):
                    e[int(rng.integers(0, e.size))] = rng.integers(0, 256)
                if rng.random() < 0.2: e = e[:int(rng.integers(0, e.size + 1))]
            enc2.append(e)
        enc = enc2
    src, soff, slen = pack_blocks(enc)
    caps = np.array([b.size if rng.random() < 0.6 else max(0, b.size + int(rng.integers(-40, 200))) for b in blocks], np.int32)
    d1, o1 = make_arena(caps + 16, fill=0xCD); d2, o2 = make_arena(caps + 16, fill=0xCD)
    want = oracle.decode_batch(src, soff, slen, d2, o2, caps, threads=32)
    if mutate:
        # on device buffers filled like the oracle's: a hostile offset of 0 leaves target bytes as they were (LL64.dec.cs:408-418), and
        # through host pointers "the target" is the context's staging buffer
        import torch
        from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
        dc = DeviceCodec(0)
        sb = DeviceBatch.from_host(src, soff, slen, dc.device)
        db = DeviceBatch(torch.full((d1.size,), 0xCD, dtype=torch.uint8, device=dc.device), torch.from_numpy(o1.view(np.int64)).to(dc.device), torch.from_numpy(caps).to(dc.device))
        got = dc.decode(sb, db).cpu().numpy()
        d1 = db.data.cpu().numpy()
    else:
        got = LZ4Codec.DecodeBatchPacked(src, soff, slen, d1, o1, caps)
    for i in range(len(blocks)):
        ok = got[i] == want[i]
        if ok and want[i] > 0:
            a, b = int(o1[i]), int(o2[i])
            ok = bytes(d1[a:a + want[i]]) == bytes(d2[b:b + want[i]]) and bool((d1[a + caps[i]:a + caps[i] + 16] == 0xCD).all())
        if not ok:
            os.makedirs(os.path.join(ROOT, "gpurun_out", "stress_fail"), exist_ok=True)
            a, b = int(o1[i]), int(o2[i])
            np.savez(os.path.join(ROOT, "gpurun_out", "stress_fail", f"{'mutate' if mutate else 'decode'}_{r}_{i}.npz"), stream=enc[i], cap=caps[i], want=want[i], got=got[i],
                     gpu=d1[a:a + caps[i] + 16], ref=d2[b:b + caps[i] + 16])
        bad += check("mutate" if mutate else "decode", r, i, ok, f"len {slen[i]} cap {caps[i]} want {want[i]} got {got[i]}")
    return bad, len(blocks)


def pickle_round(rng, oracle, r):
    bad = 0
    msgs = blocks_of(rng, 400, hi=int(rng.choice([70000, 200000, 600000])))
    for wm in (False, True):
        envs = LZ4Pickler.PickleBatch(msgs, writer_mode=wm)
        for i, m in enumerate(msgs):
            want = oracle.pickle(m, 0, 1 if wm else 0)
            bad += check("pickle", r, i, bytes(envs[i]) == want, f"len {m.size} writer {wm}: envelope {len(envs[i])} against {len(want)}")
        back = LZ4Pickler.UnpickleBatch(envs)
        for i, m in enumerate(msgs):
            bad += check("unpickle", r, i, bytes(back[i]) == m.tobytes(), f"len {m.size}")
    return bad, 2 * len(msgs)


def hc_round(rng, oracle, r):
    bad = 0
    blocks = blocks_of(rng, 300)
    src, soff, slen = pack_blocks(blocks)
    for level in (3, 9):
        caps = np.array([LZ4Codec.MaximumOutputSize(b.size) if rng.random() < 0.8 else int(rng.integers(0, LZ4Codec.MaximumOutputSize(b.size) + 1)) for b in blocks], np.int32)
        d1, o1 = make_arena(caps + 16, fill=0xCD); d2, o2 = make_arena(caps + 16, fill=0xCD)
        want = oracle.encode_batch(src, soff, slen, d2, o2, caps, level=level, threads=32)
        got = LZ4Codec.EncodeBatchPacked(src, soff, slen, d1, o1, caps, level=LZ4Level(level))
        for i in range(len(blocks)):
            ok = got[i] == want[i]
            if ok and want[i] > 0:
                a, b = int(o1[i]), int(o2[i])
                ok = bytes(d1[a:a + want[i]]) == bytes(d2[b:b + want[i]]) and bool((d1[a + want[i]:a + caps[i] + 16] == 0xCD).all())
            bad += check(f"hc{level}", r, i, ok, f"len {slen[i]} cap {caps[i]} want {want[i]} got {got[i]}")
    return bad, 2 * len(blocks)


def sizes_round(rng, oracle, r):
    """equal blocks whose total is a few bytes around multiples of the staging chunk and of the copy threads' split"""
    bad = 0
    bs = int(rng.choice([65536, 16384, 4096]))
    total = int(rng.choice([32, 40, 48, 64, 72])) * (1 << 20) + int(rng.integers(-70, 70)) + int(rng.choice([0, 1 << 19, 1 << 20, 3 << 20]))
    n = total // bs
    last = total - (n - 1) * bs                      # the last block takes the odd bytes
    if last > 65546: last = 65546
    blocks = [b for b in corpus.silesia_like_blocks(n - 1, bs, seed=int(rng.integers(0, 1 << 30)))] + [corpus.class_bytes("webster", last, int(rng.integers(0, 1 << 30)))]
    src, soff, slen = pack_blocks(blocks)
    caps = np.array([LZ4Codec.MaximumOutputSize(b.size) for b in blocks], np.int32)
    d1, o1 = make_arena(caps, fill=0xCD); d2, o2 = make_arena(caps, fill=0xCD)
    want = oracle.encode_batch(src, soff, slen, d2, o2, caps, threads=32)
    back, boff = make_arena(slen, fill=0xCD)
    reg = rng.random() < 0.5                          # every other round with the caller's three buffers page-locked (k4lz4_host_register)
    from k4os.compression.lz4_amd import host_register, host_unregister
    held = []
    try:
        if reg:
            for arr in (src, d1, back): host_register(arr); held.append(arr)
        got = LZ4Codec.EncodeBatchPacked(src, soff, slen, d1, o1, caps)
        for i in range(n):
            ok = got[i] == want[i] and bytes(d1[int(o1[i]):int(o1[i]) + want[i]]) == bytes(d2[int(o2[i]):int(o2[i]) + want[i]])
            bad += check("sizes-encode", r, i, ok, f"total {src.size} block {bs} registered {reg}")
        dl = LZ4Codec.DecodeBatchPacked(d1, o1, got, back, boff, slen)
        for i in range(n):
            ok = dl[i] == slen[i] and bytes(back[int(boff[i]):int(boff[i]) + slen[i]]) == blocks[i].tobytes()
            bad += check("sizes-decode", r, i, ok, f"total {src.size} block {bs} registered {reg}")
    finally:
        for arr in held: host_unregister(arr)
    return bad, 2 * n


def bigpickle_round(rng, oracle, r):
    """messages of 64 KiB .. 4 MiB: the segment path (k4lz4_segments.hpp) -- cut points, warm-up, joins -- against oracle.pickle"""
    bad = 0
    msgs = []
    for _ in range(int(rng.integers(20, 60))):
        n = int(rng.choice([rng.integers(65547, 200000), rng.integers(200000, 1 << 20), rng.integers(1 << 20, 4 << 20)]))
        kind = rng.random()
        if kind < 0.4: msgs.append(gen(rng, n))
        elif kind < 0.9: msgs.append(corpus.class_bytes(corpus.SILESIA_NAMES[int(rng.integers(0, 12))], n, int(rng.integers(0, 1 << 30))))
        else: msgs.append(np.concatenate([corpus.class_bytes("dickens", n // 2, 1), rng.integers(0, 256, n - n // 2, dtype=np.uint8)]))
    msgs += blocks_of(rng, 300)
    envs = LZ4Pickler.PickleBatch(msgs)
    for i, m in enumerate(msgs):
        want = oracle.pickle(m, 0, 0)
        bad += check("bigpickle", r, i, bytes(envs[i]) == want, f"len {m.size}: envelope {len(envs[i])} against {len(want)}")
    back = LZ4Pickler.UnpickleBatch(envs)
    for i, m in enumerate(msgs):
        bad += check("bigunpickle", r, i, bytes(back[i]) == m.tobytes(), f"len {m.size}")
    return bad, 2 * len(msgs)


def flags_round(rng, oracle, r):
    """LL32 for big blocks (FLAG_X32 / Enforce32), blocks stored raw when they do not shrink (FLAG_ALLOW_COPY), the order kept"""
    from k4os.compression.lz4_amd._native import FLAG_X32, FLAG_ALLOW_COPY, FLAG_NO_REORDER
    bad = 0
    blocks = blocks_of(rng, 400, hi=int(rng.choice([65547, 200000])))
    src, soff, slen = pack_blocks(blocks)
    caps = np.array([LZ4Codec.MaximumOutputSize(b.size) for b in blocks], np.int32)
    for flags in (FLAG_X32, FLAG_NO_REORDER, FLAG_ALLOW_COPY):
        d1, o1 = make_arena(caps + 16, fill=0xCD)
        got = LZ4Codec.EncodeBatchPacked(src, soff, slen, d1, o1, caps, flags=flags)
        for i, b in enumerate(blocks):
            if flags == FLAG_X32: n, out = oracle.compress_fast_x32(b, int(caps[i]))
            else: n, out = oracle.compress_fast(b, int(caps[i]))
            a = int(o1[i])
            if flags == FLAG_ALLOW_COPY and b.size and (n <= 0 or n >= b.size):
                ok = got[i] == -b.size and bytes(d1[a:a + b.size]) == b.tobytes()
            else:
                w = 0 if b.size == 0 else (-1 if n <= 0 else n)
                ok = got[i] == w and (w <= 0 or bytes(d1[a:a + w]) == bytes(out[:w]))
            bad += check(f"flags{flags}", r, i, ok, f"len {b.size} got {got[i]} oracle {n}")
    return bad, 3 * len(blocks)


def many_round(rng, oracle, r):
    """more blocks than one launch chunk holds (the parse kernel takes 16 per CU at a time)"""
    bad = 0
    n = int(rng.integers(9000, 14000)); bs = int(rng.choice([4096, 8192, 2000]))
    blocks = [b for b in corpus.silesia_like_blocks(n, bs, seed=int(rng.integers(0, 1 << 30)))]
    src, soff, slen = pack_blocks(blocks)
    caps = np.full(n, LZ4Codec.MaximumOutputSize(bs), np.int32)
    d1, o1 = make_arena(caps, fill=0xCD); d2, o2 = make_arena(caps, fill=0xCD)
    want = oracle.encode_batch(src, soff, slen, d2, o2, caps, threads=32)
    got = LZ4Codec.EncodeBatchPacked(src, soff, slen, d1, o1, caps)
    for i in range(n):
        ok = got[i] == want[i] and bytes(d1[int(o1[i]):int(o1[i]) + want[i]]) == bytes(d2[int(o2[i]):int(o2[i]) + want[i]])
        bad += check("many", r, i, ok, f"blocks {n} of {bs}")
    return bad, n


def frames_round(rng, oracle, r):
    """LZ4 frames: random contents (0 .. 3 MB), block sizes, checksums -- bit-exact with the frame oracle, decodable by liblz4, and
    liblz4's own frames (linked blocks included) decodable here"""
    from oracle_lib import FrameOracle
    from test_frame_layer import LZ4F
    from k4os.compression.lz4_amd import LZ4Frame, LZ4EncoderSettings
    fo = FrameOracle(oracle)
    try: lz4f = LZ4F()
    except OSError: lz4f = None
    bad = 0
    contents = []
    for _ in range(24):
        n = int(rng.choice([rng.integers(0, 100), rng.integers(100, 70000), rng.integers(60000, 300000), rng.integers(300000, 3000000)]))
        contents.append(gen(rng, n) if rng.random() < 0.5 and n else corpus.class_bytes(corpus.SILESIA_NAMES[int(rng.integers(0, 12))], n, int(rng.integers(0, 1 << 30))) if n else np.zeros(0, np.uint8))
    bs = int(rng.choice([65536, 262144, 1 << 20, 4 << 20])); bsum = bool(rng.random() < 0.5); csum = bool(rng.random() < 0.5)
    frames = LZ4Frame.EncodeBatch(contents, LZ4EncoderSettings(BlockSize=bs, BlockChecksum=bsum, ContentChecksum=csum))
    for i, (data, fr) in enumerate(zip(contents, frames)):
        bad += check("frame-encode", r, i, fr == fo.frame_encode(data, bs, 0, bsum, csum), f"len {data.size} block {bs} sums {bsum} {csum}")
        if lz4f is not None:
            rr, out, used = lz4f.decompress(fr, data.size + 16)
            bad += check("frame-liblz4", r, i, rr == 0 and used == len(fr) and out == data.tobytes(), f"len {data.size}")
    back = LZ4Frame.DecodeBatch(frames)
    for i, data in enumerate(contents):
        bad += check("frame-decode", r, i, bytes(back[i]) == data.tobytes(), f"len {data.size}")
    n_items = 3 * len(contents)
    if lz4f is not None:
        theirs = [lz4f.compress(c, block_id=int(rng.choice([4, 5, 6, 7])), linked=bool(rng.random() < 0.5), block_checksum=bool(rng.random() < 0.5),
                                content_checksum=bool(rng.random() < 0.5), content_size=bool(rng.random() < 0.5)) for c in contents]
        back = LZ4Frame.DecodeBatch(theirs)
        for i, data in enumerate(contents):
            bad += check("frame-from-liblz4", r, i, bytes(back[i]) == data.tobytes(), f"len {data.size}")
        n_items += len(contents)
    return bad, n_items


def partial_round(rng, oracle, r):
    """LZ4Codec.PartialDecode as a batch (FLAG_PARTIAL: the slot capacity is the number of bytes wanted), valid and mutated streams"""
    from k4os.compression.lz4_amd._native import FLAG_PARTIAL
    bad = 0
    blocks = [b for b in blocks_of(rng, 500) if b.size]
    enc = []
    for b in blocks:
        e = np.frombuffer(oracle.encode(b), np.uint8).copy()
        if rng.random() < 0.25 and e.size: e[int(rng.integers(0, e.size))] = rng.integers(0, 256)
        enc.append(e)
    src, soff, slen = pack_blocks(enc)
    wants = np.array([int(rng.choice([0, 1, rng.integers(0, b.size + 1), b.size, b.size + int(rng.integers(1, 100))])) for b in blocks], np.int32)
    d1, o1 = make_arena(wants + 32, fill=0xCD)
    # (device buffers: some of the streams are mutated, and a hostile offset of 0 leaves target bytes as they were -- see decode_round)
    import torch
    from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
    dc = DeviceCodec(0)
    sb = DeviceBatch.from_host(src, soff, slen, dc.device)
    db = DeviceBatch(torch.full((d1.size,), 0xCD, dtype=torch.uint8, device=dc.device), torch.from_numpy(o1.view(np.int64)).to(dc.device), torch.from_numpy(wants).to(dc.device))
    got = dc.decode(sb, db, flags=FLAG_PARTIAL).cpu().numpy()
    d1 = db.data.cpu().numpy()
    for i, b in enumerate(blocks):
        n, ref = oracle.decompress_partial(enc[i], int(wants[i]), int(wants[i]))
        n = -1 if n <= 0 else n
        a = int(o1[i])
        same = n <= 0 or bytes(d1[a:a + n]) == bytes(ref[:n])
        clean = bool((d1[a + wants[i]:a + wants[i] + 16] == 0xCD).all())
        ok = got[i] == n and same and clean
        if not ok and got[i] == n and n > 0:
            x, y = d1[a:a + n], np.asarray(ref[:n]); dd = np.nonzero(x != y)[0]
            print(f"   bytes same {same} slack clean {clean} differing {dd.size} first {dd[:6]} mutated {not np.array_equal(enc[i], np.frombuffer(oracle.encode(b), np.uint8))}")
            os.makedirs(os.path.join(ROOT, "gpurun_out", "stress_fail"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "stress_fail", f"partial_{r}_{i}.npz"), stream=enc[i], want=wants[i], gpu=d1[a:a + wants[i] + 16], ref=np.asarray(ref))
        bad += check("partial", r, i, ok, f"len {b.size} wanted {wants[i]} oracle {n} got {got[i]}")
    return bad, len(blocks)


def dict_round(rng, oracle, r):
    """Decode(source, target, dictionary): blocks compressed by liblz4 against a dictionary (external: anywhere; prefix: right in
    front of the target -- not reachable through packed host arenas, so external only), valid and mutated"""
    from oracle_lib import SystemLZ4
    try: sys4 = SystemLZ4()
    except OSError: return 0, 0
    bad = 0
    blocks = [b for b in blocks_of(rng, 300) if b.size >= 16]
    dicts = [corpus.class_bytes(corpus.SILESIA_NAMES[int(rng.integers(0, 12))], int(rng.integers(4, 70000)), int(rng.integers(0, 1 << 30))) for _ in blocks]
    # make the dictionary worth something: the block repeats parts of it
    for i, b in enumerate(blocks):
        d = dicts[i]
        if d.size >= 64 and b.size >= 200:
            k = int(rng.integers(0, d.size - 60)); at = int(rng.integers(0, b.size - 60)); b = b.copy(); b[at:at + 50] = d[k:k + 50]; blocks[i] = b
    enc = []
    for b, d in zip(blocks, dicts):
        e = sys4.compress_with_dict(b, d).copy()
        if rng.random() < 0.2 and e.size: e[int(rng.integers(0, e.size))] = rng.integers(0, 256)
        enc.append(e)
    src, soff, slen = pack_blocks(enc)
    dsrc, dictoff, dictlen = pack_blocks(dicts)
    caps = np.array([b.size if rng.random() < 0.7 else b.size + int(rng.integers(-30, 60)) for b in blocks], np.int32).clip(min=0)
    d1, o1 = make_arena(caps + 16, fill=0xCD)
    got = LZ4Codec.DecodeDictBatchPacked(src, soff, slen, d1, o1, caps, dsrc, dictoff, dictlen)
    for i, b in enumerate(blocks):
        n, ref = oracle.decompress_using_dict(enc[i], int(caps[i]), dicts[i])
        n = -1 if n < 0 or (n == 0 and enc[i].size) else n
        a = int(o1[i])
        ok = (got[i] == n or (n <= 0 and got[i] <= 0)) and (n <= 0 or bytes(d1[a:a + n]) == bytes(ref[:n]))
        bad += check("dict", r, i, ok, f"len {b.size} dict {dicts[i].size} cap {caps[i]} oracle {n} got {got[i]}")
    return bad, len(blocks)


def hclevels_round(rng, oracle, r):
    """every HC level on a small ragged batch (chain levels 3..9, the optimal parser 10..12)"""
    bad = 0
    blocks = blocks_of(rng, 60, hi=int(rng.choice([20000, 65547])))
    src, soff, slen = pack_blocks(blocks)
    caps = np.array([LZ4Codec.MaximumOutputSize(b.size) for b in blocks], np.int32)
    for level in (4, 5, 6, 7, 8, 10, 11, 12):
        d1, o1 = make_arena(caps + 16, fill=0xCD); d2, o2 = make_arena(caps + 16, fill=0xCD)
        want = oracle.encode_batch(src, soff, slen, d2, o2, caps, level=level, threads=32)
        got = LZ4Codec.EncodeBatchPacked(src, soff, slen, d1, o1, caps, level=LZ4Level(level))
        for i in range(len(blocks)):
            ok = got[i] == want[i] and (want[i] <= 0 or bytes(d1[int(o1[i]):int(o1[i]) + want[i]]) == bytes(d2[int(o2[i]):int(o2[i]) + want[i]]))
            bad += check(f"hc{level}", r, i, ok, f"len {slen[i]} want {want[i]} got {got[i]}")
    return bad, 8 * len(blocks)


def envelopes_round(rng, oracle, r):
    """hostile pickles: mutated / truncated envelopes through UnpickleBatch's kernels against oracle.unpickle (None = the reference
    throws InvalidDataException)"""
    from k4os.compression.lz4_amd import _native
    bad = 0
    msgs = [m for m in blocks_of(rng, 300, hi=70000) if m.size]
    envs = []
    for m in msgs:
        e = bytearray(oracle.pickle(m, 0, int(rng.integers(0, 2))))
        for _ in range(int(rng.integers(0, 3))): e[int(rng.integers(0, len(e)))] = int(rng.integers(0, 256))
        if rng.random() < 0.1: e = e[:int(rng.integers(1, len(e) + 1))]
        envs.append(bytes(e))
    # batch form: one bad envelope fails the call as a whole in the reference's terms, so go one by one on a sample
    for i in rng.permutation(len(envs))[:80]:
        e = envs[int(i)]
        want = oracle.unpickle(e)
        try: got = bytes(LZ4Pickler.Unpickle(e))
        except Exception: got = None
        ok = (got is None and want is None) or (got is not None and want is not None and got == want)
        bad += check("envelope", r, int(i), ok, f"len {len(e)}: oracle {'throws' if want is None else len(want)} here {'throws' if got is None else len(got)}")
    return bad, 80


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    which = sys.argv[3:] or ["decode", "mutate", "pickle", "hc", "sizes"]
    oracle = Oracle()
    rng = np.random.default_rng(seed)
    bad = total = 0
    t = time.time()
    for r in range(rounds):
        for w in which:
            b, n = {"decode": lambda: decode_round(rng, oracle, r, False), "mutate": lambda: decode_round(rng, oracle, r, True),
                    "pickle": lambda: pickle_round(rng, oracle, r), "hc": lambda: hc_round(rng, oracle, r), "sizes": lambda: sizes_round(rng, oracle, r),
                    "bigpickle": lambda: bigpickle_round(rng, oracle, r), "flags": lambda: flags_round(rng, oracle, r), "many": lambda: many_round(rng, oracle, r), "frames": lambda: frames_round(rng, oracle, r), "partial": lambda: partial_round(rng, oracle, r),
                    "dict": lambda: dict_round(rng, oracle, r), "hclevels": lambda: hclevels_round(rng, oracle, r), "envelopes": lambda: envelopes_round(rng, oracle, r)}[w]()
            bad += b; total += n
    print(f"seed {seed}: {rounds} rounds of {which}, {total} items, {bad} failures, {time.time() - t:.0f}s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
