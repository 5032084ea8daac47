"""Parity tests proper: the HIP path (through the C ABI of libk4lz4.so) against the oracle and the
reference's own fixtures.  They read like the reference's tests:
  BlockRoundtripTests.cs:45-125, SpanTests.cs:11-83, PicklingTests.cs:11-172, Issue64.cs:16-55,
plus batch parity on the BASELINE.json configurations."""
import os
import struct

import numpy as np
import pytest

from k4os.compression.lz4_amd import (LZ4Codec, LZ4Level, LZ4Pickler, InvalidDataException, corpus, pack_blocks,
                                      make_arena)
from k4os.compression.lz4_amd._native import FLAG_RAW_RETURN, FLAG_PICKLE_WRITER

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def roundtrip(data: np.ndarray, oracle):
    """TestBase-style helper: encode, compare with the oracle, decode into 2x buffer"""
    target = np.full(LZ4Codec.MaximumOutputSize(data.size), 0xCD, np.uint8)
    n = LZ4Codec.Encode(data, target)
    want = oracle.encode(data)
    assert n == len(want) and target[:n].tobytes() == want
    assert (target[n:] == 0xCD).all()
    out = np.full(2 * data.size + 8, 0xCD, np.uint8)
    m = LZ4Codec.Decode(target[:n].copy(), out)
    assert m == data.size and out[:m].tobytes() == data.tobytes()
    assert (out[m:] == 0xCD).all()


# ---- BlockRoundtripTests.cs -------------------------------------------------------------------
def test_quick_fox(oracle):                                         # :45-61
    roundtrip(np.frombuffer(corpus.QUICK_FOX, np.uint8), oracle)


def test_single_byte(oracle):                                       # :63-68
    roundtrip(np.array([0x7A], np.uint8), oracle)


@pytest.mark.parametrize("n", [13, 15, 17, 33, 67, 1000, 0x10000])   # :70-84
def test_repeated_byte(oracle, n):
    roundtrip(corpus.repeated(0xAA, n), oracle)


@pytest.mark.parametrize("n", [1, 1000, 0x7FFF, 0xFFFF, 0x123456])   # :86-98
def test_lorem(oracle, n):
    roundtrip(corpus.lorem(n), oracle)


@pytest.mark.parametrize("seed,n", [(0, 1000), (1, 0x7FFF), (2, 0xFFFF), (3, 0x123456)])   # :100-112
def test_incompressible(oracle, seed, n):
    roundtrip(corpus.random_bytes(n, seed), oracle)


def test_borderline_compressions(oracle):                           # :114-125
    data = corpus.class_bytes("x-ray", 0x10000, 1)
    required = len(oracle.encode(data))
    exact = np.zeros(required, np.uint8)
    assert LZ4Codec.Encode(data, exact) == required
    assert exact.tobytes() == oracle.encode(data)
    assert LZ4Codec.Encode(data, np.zeros(required - 1, np.uint8)) < 0


# ---- SpanTests.cs -------------------------------------------------------------------------------
def test_span_offsets_and_guards(oracle):                           # :11-83
    payload = corpus.lorem(5000)
    src = np.full(8000, 0xEE, np.uint8)
    src[1234:1234 + 5000] = payload
    tgt = np.full(9000, 0xCD, np.uint8)
    n = LZ4Codec.Encode(src, 1234, 5000, tgt, 777, 6000)
    want = oracle.encode(payload)
    assert n == len(want) and tgt[777:777 + n].tobytes() == want
    assert (tgt[:777] == 0xCD).all() and (tgt[777 + n:] == 0xCD).all()
    out = np.full(9000, 0xCD, np.uint8)
    m = LZ4Codec.Decode(tgt, 777, n, out, 333, 7000)
    assert m == 5000 and out[333:5333].tobytes() == payload.tobytes()
    assert (out[:333] == 0xCD).all() and (out[5333:] == 0xCD).all()


def test_decode_too_small_target_is_negative(oracle):
    data = corpus.lorem(3000)
    comp = np.frombuffer(oracle.encode(data), np.uint8)
    assert LZ4Codec.Decode(comp, np.zeros(2999, np.uint8)) < 0
    assert LZ4Codec.Decode(comp, np.zeros(3000, np.uint8)) == 3000


# ---- Issue64.cs: golden decode fixture from the reference repo ---------------------------------
def test_issue64_golden_record0():
    raw = open(os.path.join(GOLDEN, "issue64_input.bin"), "rb").read()
    want = open(os.path.join(GOLDEN, "issue64_output.bin"), "rb").read()
    u, c = struct.unpack_from("<II", raw, 24)
    out = np.zeros(u, np.uint8)
    assert LZ4Codec.Decode(np.frombuffer(raw[32:32 + c], np.uint8), out) == 65536
    assert out.tobytes() == want[:u]


# ---- PicklingTests.cs -----------------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 10, 32, 200, 1337, 0x10000, 0x172a5, 4 << 20])   # :11-50
def test_pickle_lorem_and_random(oracle, n):
    for data in (corpus.lorem(n), corpus.random_bytes(n, n)):
        p = LZ4Pickler.Pickle(data)
        assert p == oracle.pickle(data)
        assert LZ4Pickler.Unpickle(p) == data.tobytes()
        assert LZ4Pickler.UnpickledSize(p) == n


def test_pickle_writer_and_span_forms(oracle):                      # :52-147
    import io
    data = corpus.lorem(70000)
    w = io.BytesIO()
    LZ4Pickler.Pickle(data, w)
    assert w.getvalue() == oracle.pickle(data, 0, 1)
    assert LZ4Pickler.Unpickle(w.getvalue()) == data.tobytes()
    buf = np.full(80000, 0xEE, np.uint8)
    buf[100:70100] = data
    assert LZ4Pickler.Pickle(buf, 100, 70000) == LZ4Pickler.Pickle(data)    # array vs span: identical bytes
    out = np.zeros(70000, np.uint8)
    LZ4Pickler.Unpickle(LZ4Pickler.Pickle(data), out)
    assert out.tobytes() == data.tobytes()
    w2 = io.BytesIO()
    LZ4Pickler.Unpickle(LZ4Pickler.Pickle(data), w2)
    assert w2.getvalue() == data.tobytes()


def test_pickle_corruption_raises(oracle):                          # :149-172
    p = bytearray(LZ4Pickler.Pickle(corpus.lorem(5000)))
    bad_version = bytes([p[0] | 1]) + bytes(p[1:])
    with pytest.raises(InvalidDataException):
        LZ4Pickler.Unpickle(bad_version)
    with pytest.raises(InvalidDataException):
        LZ4Pickler.Unpickle(bytes(p[:-7]))
    with pytest.raises(InvalidDataException):
        LZ4Pickler.Unpickle(bytes(p), np.zeros(4999, np.uint8))
    flipped = bytearray(p)
    flipped[1] ^= 0x40
    with pytest.raises(InvalidDataException):
        LZ4Pickler.Unpickle(bytes(flipped))


# ---- batch parity on the BASELINE.json configurations -----------------------------------------
def test_batch_encode_decode_silesia_like_vs_oracle(oracle):
    """configs[1] at reduced count: 12 classes x 64 KiB, every block compared with the oracle"""
    blocks = corpus.silesia_like_blocks(240, 65536, seed=2)
    n = blocks.shape[0]
    src = blocks.reshape(-1)
    off = np.arange(n, dtype=np.uint64) * 65536
    lens = np.full(n, 65536, np.int32)
    caps = np.full(n, LZ4Codec.MaximumOutputSize(65536), np.int32)
    dst, doff = make_arena(caps, fill=0xCD)
    out = LZ4Codec.EncodeBatchPacked(src, off, lens, dst, doff, caps)
    want_len = oracle.encode_batch(src, off, lens, np.empty_like(dst), doff, caps, threads=8)
    assert np.array_equal(out, want_len)
    ref = np.full_like(dst, 0xCD)
    oracle.encode_batch(src, off, lens, ref, doff, caps, threads=8)
    assert np.array_equal(dst, ref)          # bytes and untouched slack identical
    back, boff = make_arena(lens, fill=0xCD)
    dl = LZ4Codec.DecodeBatchPacked(dst, doff, out, back, boff, lens)
    assert (dl == 65536).all() and np.array_equal(back[:n * 65536].reshape(n, 65536), blocks)


def test_batch_small_blocks_4k_vs_oracle(oracle):
    """configs[2] shape at reduced count: 4 KiB blocks, text + random"""
    blocks = np.concatenate([corpus.silesia_like_blocks(600, 4096, seed=5),
                             corpus.random_bytes(200 * 4096, 1).reshape(200, 4096)])
    n = blocks.shape[0]
    enc = LZ4Codec.EncodeBatch(list(blocks))
    for i in range(n):
        assert enc[i] == oracle.encode(blocks[i]), i
    dec = LZ4Codec.DecodeBatch(enc, [4096] * n)
    assert all(d == blocks[i].tobytes() for i, d in enumerate(dec))


def test_batch_ragged_and_empty(oracle):
    """empty, tiny, table-switch sizes and > 64 KiB blocks in one batch"""
    sizes = [0, 1, 5, 12, 13, 14, 100, 65535, 65536, 65546, 65547, 70000, 300000, 0, 1 << 20]
    blocks = [corpus.class_bytes(corpus.SILESIA_NAMES[i % 12], s, i) if s else np.zeros(0, np.uint8)
              for i, s in enumerate(sizes)]
    # matches of tens of thousands of bytes, in a small-table and in two big-table blocks
    blocks += [corpus.repeated(0x55, 65000), np.concatenate([corpus.random_bytes(30000, 4)] * 3), corpus.repeated(7, 100000)]
    sizes = sizes + [65000, 90000, 100000]
    enc = LZ4Codec.EncodeBatch(blocks)
    for i, b in enumerate(blocks):
        assert enc[i] == (b"" if b.size == 0 else oracle.encode(b)), i
    dec = LZ4Codec.DecodeBatch(enc, sizes)
    assert all(d == b.tobytes() for d, b in zip(dec, blocks))


def test_batch_malformed_streams_raw_returns(oracle):
    """accept/reject + LLxx return values + bytes identical to the oracle; guards untouched"""
    rng = np.random.default_rng(31)
    comps, caps = [], []
    for name, n in (("dickens", 3000), ("xml", 6000), ("mr", 66000)):
        data = corpus.class_bytes(name, n, 4)
        good = np.frombuffer(oracle.encode(data), np.uint8)
        for t in range(400):
            bad = good.copy()
            k = t % 4
            if k == 0:
                bad = bad[:rng.integers(1, good.size)]
            elif k == 1:
                for _ in range(int(rng.integers(1, 4))):
                    bad[rng.integers(0, good.size)] = rng.integers(0, 256)
            elif k == 2:
                bad = np.concatenate([bad, rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8)])
            comps.append(bad)
            caps.append(n + int(rng.integers(-20, 21)) if k != 3 else int(rng.integers(0, n)))
    src, soff, slen = pack_blocks(comps)
    caps = np.array(caps, np.int32)
    dst, doff = make_arena(caps + 32, fill=0xCD)
    out = LZ4Codec.DecodeBatchPacked(src, soff, slen, dst, doff, caps, flags=FLAG_RAW_RETURN)
    for i, (c, cap) in enumerate(zip(comps, caps)):
        n, ref = oracle.decompress_safe(c, int(cap))
        assert out[i] == n, i
        if n > 0:
            assert dst[int(doff[i]):int(doff[i]) + n].tobytes() == ref[:n].tobytes()
        assert (dst[int(doff[i]) + max(n, 0):int(doff[i]) + int(cap) + 32] == 0xCD).all()


def test_batch_limited_output(oracle):
    blocks, caps, wants = [], [], []
    for name in corpus.SILESIA_NAMES:
        b = corpus.class_bytes(name, 65536, 6)
        full = oracle.encode(b)
        for cap in (len(full), len(full) - 1, len(full) // 2, 0):
            blocks.append(b); caps.append(cap); wants.append(full if cap >= len(full) else None)
    src, soff, slen = pack_blocks(blocks)
    caps = np.array(caps, np.int32)
    dst, doff = make_arena(caps + 16, fill=0xCD)
    out = LZ4Codec.EncodeBatchPacked(src, soff, slen, dst, doff, caps)
    for i, w in enumerate(wants):
        if w is None:
            assert out[i] == -1
        else:
            assert out[i] == len(w) and dst[int(doff[i]):int(doff[i]) + len(w)].tobytes() == w
        assert (dst[int(doff[i]) + int(caps[i]):int(doff[i]) + int(caps[i]) + 16] == 0xCD).all()


def test_pickle_batch_variable_messages(oracle):
    """configs[3] shape at reduced size: variable-length random/text messages, both header rules"""
    data, off, lens = corpus.variable_messages(200, seed=4, lo=1024, hi=1 << 20, budget_bytes=24 << 20)
    msgs = [data[int(o):int(o) + int(l)] for o, l in zip(off, lens)]
    for writer in (False, True):
        ps = LZ4Pickler.PickleBatch(msgs, writer_mode=writer)
        for i, m in enumerate(msgs):
            assert ps[i] == oracle.pickle(m, 0, int(writer)), i
        back = LZ4Pickler.UnpickleBatch(ps)
        assert all(b == m.tobytes() for b, m in zip(back, msgs))


def test_full_size_config2_properties(oracle):
    """configs[1] at full size (4096 x 64 KiB) on device-resident buffers: round trip + checksum of
    sizes against the oracle's total, bytes of all 4096 blocks and the untouched rest of every slot."""
    import torch
    from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
    blocks = corpus.silesia_like_blocks(4096, 65536, seed=2)
    n = blocks.shape[0]
    dc = DeviceCodec(0)
    lens = np.full(n, 65536, np.int32)
    off = np.arange(n, dtype=np.uint64) * 65536
    src = DeviceBatch.from_host(blocks.reshape(-1), off, lens, dc.device)
    comp = DeviceBatch.empty_slots(np.full(n, LZ4Codec.MaximumOutputSize(65536)), dc.device, fill=0xCD)
    clen = dc.encode(src, comp)
    back = DeviceBatch.empty_slots(lens, dc.device)
    dlen = dc.decode(DeviceBatch(comp.data, comp.off, clen), back)
    torch.cuda.synchronize()
    assert (dlen.cpu().numpy() == 65536).all()
    assert torch.equal(back.data[:n * 65536], src.data[:n * 65536])
    clen_h = clen.cpu().numpy()
    caps = np.full(n, LZ4Codec.MaximumOutputSize(65536), np.int32)
    ref_dst, ref_off = make_arena(caps)
    want = oracle.encode_batch(blocks.reshape(-1), off, lens, ref_dst, ref_off, caps, threads=os.cpu_count() or 8)
    assert np.array_equal(clen_h, want)
    comp_h = comp.data.cpu().numpy()
    coff = comp.off.cpu().numpy()
    for i in range(n):
        assert comp_h[coff[i]:coff[i] + clen_h[i]].tobytes() == ref_dst[int(ref_off[i]):int(ref_off[i]) + int(want[i])].tobytes(), i
        assert (comp_h[coff[i] + clen_h[i]:coff[i] + caps[i]] == 0xCD).all()


# ---- HC levels (BASELINE.json configs[4]) -----------------------------------
@pytest.mark.parametrize("level", [LZ4Level.L03_HC, LZ4Level.L04_HC, LZ4Level.L06_HC, LZ4Level.L08_HC, LZ4Level.L09_HC,
                                   LZ4Level.L10_OPT, LZ4Level.L11_OPT, LZ4Level.L12_MAX])
def test_hc_single_block_roundtrip(oracle, level):
    for data in (corpus.lorem(0x172a5), corpus.class_bytes("webster", 65536, 3), corpus.repeated(7, 1000)):
        target = np.full(LZ4Codec.MaximumOutputSize(data.size), 0xCD, np.uint8)
        n = LZ4Codec.Encode(data, target, level)
        r, w = oracle.compress_hc(data, int(level))
        assert n == r and target[:n].tobytes() == w[:r].tobytes() and (target[n:] == 0xCD).all()
        out = np.zeros(data.size, np.uint8)
        assert LZ4Codec.Decode(target[:n].copy(), out) == data.size and out.tobytes() == data.tobytes()


def test_hc_batch_silesia_like_vs_oracle(oracle):
    """configs[4] at reduced count: 64 KiB blocks, L03_HC, every block byte-compared; ratio equal"""
    blocks = corpus.silesia_like_blocks(120, 65536, seed=2)
    enc = LZ4Codec.EncodeBatch(list(blocks), LZ4Level.L03_HC)
    total = 0
    for i in range(blocks.shape[0]):
        r, w = oracle.compress_hc(blocks[i], 3)
        assert enc[i] == w[:r].tobytes(), i
        total += r
    assert sum(len(e) for e in enc) == total
    dec = LZ4Codec.DecodeBatch(enc, [65536] * blocks.shape[0])
    assert all(d == blocks[i].tobytes() for i, d in enumerate(dec))


def test_hc_ragged_batch_and_limits(oracle):
    sizes = [0, 1, 12, 13, 100, 4096, 65536, 70000, 300000]
    blocks = [corpus.class_bytes(corpus.SILESIA_NAMES[(3 * i) % 12], s, i) if s else np.zeros(0, np.uint8)
              for i, s in enumerate(sizes)]
    enc = LZ4Codec.EncodeBatch(blocks, LZ4Level.L05_HC)
    for i, b in enumerate(blocks):
        if b.size == 0:
            assert enc[i] == b""
        else:
            r, w = oracle.compress_hc(b, 5)
            assert enc[i] == w[:r].tobytes(), i
    # limitedOutput: exact size fits, one byte less does not
    b = blocks[6]
    r, w = oracle.compress_hc(b, 3)
    assert LZ4Codec.Encode(b, np.zeros(r, np.uint8), LZ4Level.L03_HC) == r
    assert LZ4Codec.Encode(b, np.zeros(r - 1, np.uint8), LZ4Level.L03_HC) < 0


def test_hc_pickle_levels(oracle):
    """PicklingTests.cs:11-50 at an HC level"""
    msgs = [corpus.lorem(n) for n in (0, 10, 200, 1337, 0x10000)] + [corpus.random_bytes(5000, 3)]
    ps = LZ4Pickler.PickleBatch(msgs, LZ4Level.L03_HC)
    for m, p in zip(msgs, ps):
        assert p == oracle.pickle(m, 3)
    assert [LZ4Pickler.Unpickle(p) for p in ps] == [m.tobytes() for m in msgs]


def test_hc_level9_pattern_analysis_batch(oracle):
    """L09_HC = 256 attempts + pattern analysis (LL64.high.cs:208-337): runs of 1/2/4-byte patterns, and the
    12 classes, byte-compared with the oracle (itself byte-equal to liblz4's level 9)"""
    rng = np.random.default_rng(9)
    blocks = []
    for unit in (b"a", b"ab", b"abcd", b"aaab", b"abc"):
        parts = []
        for _ in range(60):
            parts.append(np.frombuffer(unit * int(rng.integers(1, 400)), np.uint8))
            parts.append(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8))
        blocks.append(np.concatenate(parts))
    blocks.append(np.concatenate([np.zeros(70000, np.uint8), np.frombuffer(b"xyz", np.uint8), np.zeros(70000, np.uint8)]))
    blocks += [corpus.class_bytes(name, 40000, 11) for name in corpus.SILESIA_NAMES]
    enc = LZ4Codec.EncodeBatch(blocks, LZ4Level.L09_HC)
    for i, b in enumerate(blocks):
        r, w = oracle.compress_hc(b, 9)
        assert enc[i] == w[:r].tobytes(), i
    dec = LZ4Codec.DecodeBatch(enc, [b.size for b in blocks])
    assert all(d == b.tobytes() for d, b in zip(dec, blocks))


@pytest.mark.parametrize("level", [LZ4Level.L10_OPT, LZ4Level.L11_OPT, LZ4Level.L12_MAX])
def test_optimal_parser_levels_batch(oracle, level):
    """L10_OPT..L12_MAX = LZ4HC_compress_optimal (LL64.high.cs:802-1122): pattern runs, the 12 classes, limited output;
    byte-compared with the oracle (itself byte-equal to liblz4 at these levels)"""
    rng = np.random.default_rng(10)
    blocks = []
    for unit in (b"a", b"ab", b"abcd", b"abc"):
        parts = []
        for _ in range(40):
            parts.append(np.frombuffer(unit * int(rng.integers(1, 400)), np.uint8))
            parts.append(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8))
        blocks.append(np.concatenate(parts))
    blocks += [corpus.class_bytes(name, 30000, 12) for name in corpus.SILESIA_NAMES]
    blocks += [corpus.lorem(70000), np.zeros(0, np.uint8), corpus.lorem(12), corpus.lorem(13)]
    enc = LZ4Codec.EncodeBatch(blocks, level)
    for i, b in enumerate(blocks):
        if b.size == 0:
            assert enc[i] == b""
            continue
        r, w = oracle.compress_hc(b, int(level))
        assert enc[i] == w[:r].tobytes(), i
    dec = LZ4Codec.DecodeBatch(enc, [b.size for b in blocks])
    assert all(d == b.tobytes() for d, b in zip(dec, blocks))
    b = blocks[6]
    r, w = oracle.compress_hc(b, int(level))
    assert LZ4Codec.Encode(b, np.zeros(r, np.uint8), level) == r
    assert LZ4Codec.Encode(b, np.zeros(r - 1, np.uint8), level) < 0


def test_every_level_of_the_enum_is_implemented():
    data = corpus.lorem(5000)
    for level in LZ4Level:
        tgt = np.zeros(LZ4Codec.MaximumOutputSize(data.size), np.uint8)
        n = LZ4Codec.Encode(data, tgt, level)
        out = np.zeros(data.size, np.uint8)
        assert n > 0 and LZ4Codec.Decode(tgt[:n].copy(), out) == data.size and out.tobytes() == data.tobytes()


def test_dispatch_variants_give_identical_bytes(oracle):
    """cost-ordered dispatch and the LDS-table / global-memory-table split are scheduling only:
    K4LZ4_FLAG_NO_REORDER (4) and K4LZ4_FLAG_NO_SPLIT (16) must not change a byte"""
    blocks = corpus.silesia_like_blocks(1200, 16384, seed=9)
    n = blocks.shape[0]
    src = blocks.reshape(-1)
    off = np.arange(n, dtype=np.uint64) * 16384
    lens = np.full(n, 16384, np.int32)
    caps = np.full(n, LZ4Codec.MaximumOutputSize(16384), np.int32)
    outs = []
    for flags in (0, 4, 16, 4 | 16):
        dst, doff = make_arena(caps, fill=0xCD)
        out = LZ4Codec.EncodeBatchPacked(src, off, lens, dst, doff, caps, flags=flags)
        outs.append((out.copy(), dst.copy()))
    ref_dst = np.full_like(outs[0][1], 0xCD)
    want = oracle.encode_batch(src, off, lens, ref_dst, make_arena(caps)[1], caps, threads=8)
    for out, dst in outs:
        assert np.array_equal(out, want) and np.array_equal(dst, ref_dst)


def _zero_one_offset(enc: bytes, rng) -> bytes:
    """the stream with the offset of one of its sequences set to 0 (LL64.dec.cs:408-418 lets it through: the match copies output
    bytes onto themselves, i.e. leaves what the target held)"""
    e = bytearray(enc)
    i, offs = 0, []
    while i < len(e):
        t = e[i]; i += 1
        ll = t >> 4
        if ll == 15:
            while True:
                x = e[i]; i += 1; ll += x
                if x != 255: break
        i += ll
        if i >= len(e): break
        offs.append(i); i += 2
        if (t & 15) == 15:
            while e[i] == 255: i += 1
            i += 1
    if offs:
        k = offs[int(rng.integers(0, len(offs)))]
        e[k] = 0; e[k + 1] = 0
    return bytes(e)


def test_streams_with_an_offset_of_zero_leave_the_target_as_it_was(oracle):
    """Hostile streams the reference decodes without complaint: a match with offset 0 copies output bytes onto themselves, so those
    bytes of the target stay what they were (0xCD here).  The captured stream (tests/golden/mutated_stream_offset0_in_shortcut.npy:
    found by tests/tools/gpu_stress_all.py in round 5) and 300 made ones."""
    rng = np.random.default_rng(77)
    here = os.path.dirname(os.path.abspath(__file__))
    streams = [np.load(os.path.join(here, "golden", "mutated_stream_offset0_in_shortcut.npy")).tobytes()]
    sizes = [5813]
    for i in range(300):
        b = corpus.class_bytes(corpus.SILESIA_NAMES[i % 12], int(rng.integers(200, 20000)), i)
        streams.append(_zero_one_offset(oracle.encode(b), rng))
        sizes.append(b.size + int(rng.choice([0, 0, 7, 100])))
    src, soff, slen = pack_blocks([np.frombuffer(x, np.uint8) for x in streams])
    caps = np.array(sizes, np.int32)
    import torch
    from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
    ref, roff = make_arena(caps + 16, fill=0xCD)
    want = oracle.decode_batch(src, soff, slen, ref, roff, caps, threads=8)
    # on DEVICE buffers: "as it was" is what the target slot held.  (Through host pointers the target slot is the context's staging
    # buffer: there these bytes are zeroed -- the next test.)
    dc = DeviceCodec(0)
    sb = DeviceBatch.from_host(src, soff, slen, dc.device)
    db = DeviceBatch(torch.full((ref.size,), 0xCD, dtype=torch.uint8, device=dc.device), torch.from_numpy(roff.view(np.int64)).to(dc.device),
                     torch.from_numpy(caps).to(dc.device))
    got = dc.decode(sb, db).cpu().numpy()
    dst, doff = db.data.cpu().numpy(), roff
    assert np.array_equal(got, want)
    assert (want > 0).sum() > 200
    for i in range(len(streams)):
        if want[i] > 0:
            a, b = int(doff[i]), int(roff[i])
            assert np.array_equal(dst[a:a + want[i]], ref[b:b + want[i]]), i          # the bytes left alone included
            assert (dst[a + caps[i]:a + caps[i] + 16] == 0xCD).all(), i


def test_offset_zero_through_host_pointers_never_returns_another_calls_bytes(oracle):
    """a host-pointer decode runs on the context's staging buffer, which holds what earlier calls left there; the
    bytes an offset-0 match "copies onto themselves" (LL64.dec.cs:408-418) are therefore ZEROED on this path (the reference leaves
    them what the caller's buffer held -- undefined bytes either way, but never another caller's data).  The staging buffer is first
    filled with a recognisable pattern by an ordinary decode through the same context; return values and every defined byte are
    the oracle's, the offset-0 bytes are 0."""
    rng = np.random.default_rng(78)
    marker = np.full(400000, 0xEE, np.uint8)
    streams, sizes = [], []
    for i in range(120):
        b = corpus.class_bytes(corpus.SILESIA_NAMES[i % 12], int(rng.integers(200, 20000)), i)
        streams.append(_zero_one_offset(oracle.encode(b), rng))
        sizes.append(b.size)
    # plus the hand-made one of the emulator test: 8 literals, a "match" of 6 bytes at offset 0, then ordinary sequences
    streams.append(bytes([0x82]) + b"ABCDEFGH" + bytes([0, 0]) + bytes([0x21]) + b"ij" + bytes([8, 0]) + bytes([0xC0]) + b"123456789012")
    sizes.append(33)
    src, soff, slen = pack_blocks([np.frombuffer(x, np.uint8) for x in streams])
    caps = np.array(sizes, np.int32)
    ref, roff = make_arena(caps + 16, fill=0x00)            # the oracle leaves the bytes as they are: a zeroed target is what the host path gives
    want = oracle.decode_batch(src, soff, slen, ref, roff, caps, threads=8)
    for attempt in range(2):                                 # (second time: the staging buffer holds the first attempt's output)
        back = LZ4Codec.DecodeBatchPacked(*pack_blocks([np.frombuffer(oracle.encode(marker), np.uint8)]), *make_arena(np.array([marker.size], np.int32)), np.array([marker.size], np.int32))
        assert back[0] == marker.size
        dst, doff = make_arena(caps + 16, fill=0xCD)
        got = LZ4Codec.DecodeBatchPacked(src, soff, slen, dst, doff, caps, flags=FLAG_RAW_RETURN)
        assert np.array_equal(got, want)
        assert (want > 0).sum() > 60
        for i in range(len(streams)):
            if want[i] > 0:
                a, b = int(doff[i]), int(roff[i])
                assert np.array_equal(dst[a:a + want[i]], ref[b:b + want[i]]), (attempt, i)
                assert (dst[a + want[i]:a + caps[i] + 16] == 0xCD).all(), (attempt, i)
    a = int(doff[-1])
    assert got[-1] == 33 and dst[a:a + 8].tobytes() == b"ABCDEFGH" and (dst[a + 8:a + 14] == 0).all()


def test_randomised_ragged_batches_through_the_default_fast_encoder(oracle):
    """tests/tools/gpu_stress_encode.py, two rounds: 2 x 1500 blocks of 0 .. 100 000 bytes (adversarial generator + corpus classes),
    ragged output limits, through k4_parse_kernel and the one-kernel encoder behind it: return value, bytes, untouched slack"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import gpu_stress_encode
    assert gpu_stress_encode.run(2, 5, 1500, oracle) == 0          # (seed 5 holds the block that found the 66th-probe corner, round 5)
    assert gpu_stress_encode.run(2, 6, 3000, oracle) == 0          # round 6: two more rounds of 3 000 blocks, another seed ...
    assert gpu_stress_encode.run(1, 7, 3000, oracle, device=True) == 0   # ... and one on device buffers (no host staging in between)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_batches_beyond_one_residency_through_the_persistent_launch(oracle):
    """round 6: more blocks than one residency of the parse kernel (16 per CU) go through ONE persistent launch -- every wave takes
    the next block of the cost order when it is done with one, the records live in per-wave slots that are reused block after
    block, a wave that finds the queue empty frees its LDS table for a block that still parses with its table in memory.  9 400
    ragged blocks (many more than are resident), ragged output limits: the oracle's bytes, lengths and failures, and the same
    from launches of one residency each (K4LZ4_NO_PERSIST)."""
    import os
    from k4os.compression.lz4_amd import _native
    import adversarial_blocks
    rng = np.random.default_rng(61)
    blocks = [b[:int(rng.integers(130, 8193))] for b in corpus.silesia_like_blocks(9000, 8192, seed=8)]
    blocks += [b for b in corpus.silesia_like_blocks(300, 65536, seed=9)]
    blocks += adversarial_blocks.search_limit_at_block_end() + [adversarial_blocks.dense_four_byte_matches(128, 65546, 128)]
    blocks += [corpus.lorem(n) for n in (0, 1, 13, 127, 128, 65546, 65547, 70000)]
    order = rng.permutation(len(blocks))
    blocks = [blocks[i] for i in order]
    caps = np.array([LZ4Codec.MaximumOutputSize(b.size) if rng.random() < 0.85 else int(rng.integers(0, LZ4Codec.MaximumOutputSize(b.size) + 1)) for b in blocks], np.int32)
    src, soff, slen = pack_blocks(blocks)
    ref_dst, ref_off = make_arena(caps + 16, fill=0xCD)
    want = oracle.encode_batch(src, soff, slen, ref_dst, ref_off, caps, threads=32)
    for env in ({}, {"K4LZ4_NO_PERSIST": "1"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ctx = _native.Context(-1)
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        for rep in range(2):                     # (the second call reuses the slots the first one filled)
            dst, doff = make_arena(caps + 16, fill=0xCD)
            out = LZ4Codec.EncodeBatchPacked(src, soff, slen, dst, doff, caps, ctx=ctx)
            assert np.array_equal(out, want), (env, rep, np.nonzero(out != want)[0][:5])
            for i in np.nonzero(want > 0)[0]:
                a = dst[int(doff[i]):int(doff[i]) + caps[i] + 16]; b = ref_dst[int(ref_off[i]):int(ref_off[i]) + caps[i] + 16]
                assert np.array_equal(a[:want[i]], b[:want[i]]) and (a[want[i]:] == 0xCD).all(), (env, rep, int(i))
        ctx.close()


def test_fast_encoder_paths_give_identical_bytes(oracle):
    """Which kernels encode a fast-level batch is scheduling only: the two-kernel path of k4lz4_parse.hpp (parse + write-out by
    the parsing wave, or by k4_emit_kernel: K4LZ4_NO_INLINE_EMIT; 16 or fewer blocks per workgroup: K4LZ4_PARSE_WAVES; waves that
    take their blocks from a queue: K4LZ4_PARSE_QUEUE; tables that stay in memory: K4LZ4_NO_MIGRATE) with the one-kernel encoder behind it for the blocks it leaves alone, or
    the one-kernel encoders alone (K4LZ4_NO_PARSE).  Every one of them: the oracle's bytes, length and failures included."""
    import os
    from k4os.compression.lz4_amd import _native
    rng = np.random.default_rng(21)
    blocks = [b for b in corpus.silesia_like_blocks(1500, 16384, seed=5)]
    blocks += [b for b in corpus.silesia_like_blocks(60, 65536, seed=6)]
    blocks += [corpus.lorem(n) for n in (0, 1, 12, 13, 64, 127, 128, 129, 200, 65546, 65547, 70000)]
    blocks += [corpus.class_bytes("samba", 150000, 3), corpus.random_bytes(65000, 9), corpus.repeated(7, 65536)]
    import adversarial_blocks
    hard = adversarial_blocks.search_limit_at_block_end() + [adversarial_blocks.dense_four_byte_matches(128, 65546, 128)]
    blocks += hard
    caps = []
    for i, b in enumerate(blocks):
        bound = LZ4Codec.MaximumOutputSize(b.size)
        caps.append(bound if i >= len(blocks) - len(hard) or rng.random() < 0.8 else int(rng.integers(0, bound + 1)))
    caps = np.array(caps, np.int32)
    src, soff, slen = pack_blocks(blocks)
    ref_dst, ref_off = make_arena(caps + 16, fill=0xCD)
    want = oracle.encode_batch(src, soff, slen, ref_dst, ref_off, caps, threads=8)
    envs = [{}, {"K4LZ4_NO_PARSE": "1"}, {"K4LZ4_NO_INLINE_EMIT": "1"}, {"K4LZ4_PARSE_WAVES": "5"},
            {"K4LZ4_PARSE_QUEUE": "1", "K4LZ4_PARSE_WAVES": "3"}, {"K4LZ4_PCOST": "1"}, {"K4LZ4_NO_MIGRATE": "1"}]
    for env in envs:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ctx = _native.Context(-1)            # (the switches are read when a context is made)
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        for flags in (0, 4):                     # cost order / index order
            dst, doff = make_arena(caps + 16, fill=0xCD)
            out = LZ4Codec.EncodeBatchPacked(src, soff, slen, dst, doff, caps, flags=flags, ctx=ctx)
            assert np.array_equal(out, want), (env, flags, np.nonzero(out != want)[0][:5])
            for i in np.nonzero(want > 0)[0]:
                a = dst[int(doff[i]):int(doff[i]) + caps[i] + 16]; b = ref_dst[int(ref_off[i]):int(ref_off[i]) + caps[i] + 16]
                assert np.array_equal(a[:want[i]], b[:want[i]]) and (a[want[i]:] == 0xCD).all(), (env, flags, int(i))
        ctx.close()


def test_hc_level3_paths_give_identical_bytes(oracle):
    """round 6: how a level-3 batch of blocks up to 64 KiB is parsed is scheduling only -- one, two or four waves per block, each
    from its own start and joined where their cursors meet (K4LZ4_HC_SEGS; by default four), the candidate records out of LDS or from memory (K4LZ4_HC_CAND_MEM), the chains by eight waves per block or
    by the LDS-table kernel of rounds 3-5 (K4LZ4_HC_CHAIN_OLD), sequence records or LZ4HC_encodeSequence in the loop
    (K4LZ4_NO_HC_RECORDS).  Every one of them: the oracle's bytes, ragged lengths and ragged output limits included."""
    import os
    from k4os.compression.lz4_amd import _native
    rng = np.random.default_rng(33)
    blocks = [corpus.class_bytes(name, int(rng.integers(200, 65537)), 11 + i) for i, name in enumerate(corpus.SILESIA_NAMES * 8)]
    blocks += [b for b in corpus.silesia_like_blocks(96, 65536, seed=8)]
    blocks += [corpus.lorem(n) for n in (0, 1, 12, 13, 14, 8191, 8192, 8193, 65536)] + [corpus.repeated(0xAA, n) for n in (13, 8192, 65536)]
    blocks += [np.concatenate([np.zeros(30000, np.uint8), corpus.random_bytes(100, 1), np.zeros(30000, np.uint8)]), corpus.random_bytes(65536, 4)]
    import adversarial_blocks
    blocks.append(adversarial_blocks.dense_four_byte_matches(128, 65536, 128))
    caps = []
    for b in blocks:
        bound = LZ4Codec.MaximumOutputSize(b.size)
        caps.append(bound if rng.random() < 0.8 else int(rng.integers(0, bound + 1)))
    caps = np.array(caps, np.int32)
    src, soff, slen = pack_blocks(blocks)
    ref_dst, ref_off = make_arena(caps + 16, fill=0xCD)
    want = oracle.encode_batch(src, soff, slen, ref_dst, ref_off, caps, level=3, threads=8)
    envs = [{}, {"K4LZ4_HC_SEGS": "1"}, {"K4LZ4_HC_SEGS": "2"}, {"K4LZ4_HC_SEGS": "4"}, {"K4LZ4_HC_CHAIN_OLD": "1"}, {"K4LZ4_NO_HC_RECORDS": "1"},
            {"K4LZ4_HC_CAND_MEM": "1"}]
    for env in envs:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ctx = _native.Context(-1)            # (the switches are read when a context is made)
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        dst, doff = make_arena(caps + 16, fill=0xCD)
        out = LZ4Codec.EncodeBatchPacked(src, soff, slen, dst, doff, caps, level=LZ4Level.L03_HC, ctx=ctx)
        assert np.array_equal(out, want), (env, np.nonzero(out != want)[0][:5])
        for i in np.nonzero(want > 0)[0]:
            a = dst[int(doff[i]):int(doff[i]) + caps[i] + 16]; b = ref_dst[int(ref_off[i]):int(ref_off[i]) + caps[i] + 16]
            assert np.array_equal(a[:want[i]], b[:want[i]]) and (a[want[i]:] == 0xCD).all(), (env, int(i))
        ctx.close()


def test_decoder_kernel_variants_give_identical_results(oracle):
    """Which decoder kernel a batch gets is scheduling only: two waves per block (up to 16 blocks per CU), one wave
    per block (K4LZ4_NO_PAIR=1, or up to 24 per CU), the dense build (more).  Same bytes and the same verdict on
    damaged streams from all of them."""
    import os
    rng = np.random.default_rng(12)
    blocks = [b for b in corpus.silesia_like_blocks(480, 8192, seed=13)]
    blocks += [corpus.class_bytes("dickens", 70000, 3), corpus.lorem(3), np.zeros(0, np.uint8), corpus.repeated(9, 40000)]
    enc = [np.frombuffer(oracle.encode(b), np.uint8).copy() if b.size else np.zeros(0, np.uint8) for b in blocks]
    for i in range(0, 480, 7):                               # hostile: one byte changed
        if enc[i].size > 8:
            enc[i][int(rng.integers(0, enc[i].size))] ^= int(rng.integers(1, 256))
    sizes = [b.size for b in blocks]

    def run(batch_enc, batch_sizes, flags=1):                 # 1 = raw engine results
        src, soff, slen = pack_blocks(batch_enc)
        caps = np.array(batch_sizes, np.int32)
        dst, doff = make_arena(caps + 16, fill=0xCD)
        out = LZ4Codec.DecodeBatchPacked(src, soff, slen, dst, doff, caps, flags=flags)
        return out.copy(), dst.copy()

    pair = run(enc, sizes)
    longest_first = run(enc, sizes, flags=1 | 8)             # K4LZ4_FLAG_REORDER: dispatch order only
    assert np.array_equal(pair[0], longest_first[0]) and np.array_equal(pair[1], longest_first[1])
    os.environ["K4LZ4_NO_PAIR"] = "1"
    try:
        single = run(enc, sizes)
    finally:
        del os.environ["K4LZ4_NO_PAIR"]
    assert np.array_equal(pair[0], single[0])
    _, off = make_arena(np.array(sizes, np.int32) + 16)
    for i, b in enumerate(blocks):
        want, ref = oracle.decompress_safe(enc[i], b.size) if enc[i].size else (pair[0][i], None)
        assert pair[0][i] == want, (i, int(pair[0][i]), want)
        if want > 0:
            a = pair[1][int(off[i]):int(off[i]) + want]
            c = single[1][int(off[i]):int(off[i]) + want]
            assert np.array_equal(a, ref[:want]) and np.array_equal(c, ref[:want]), i
        for arr in (pair[1], single[1]):
            assert (arr[int(off[i]) + b.size:int(off[i]) + b.size + 16] == 0xCD).all(), (i, "guard")
    # the dense build: the same blocks 14 times over (> 24 blocks per CU)
    many_enc, many_sizes = enc * 14, sizes * 14
    dense = run(many_enc, many_sizes)
    assert np.array_equal(dense[0], np.tile(pair[0], 14))


# ---- PartialDecode (LZ4Codec.cs:123-173; reference test PartialDecodeTests) ---------------------
@pytest.mark.parametrize("cls", ["dickens", "xml", "x-ray", "nci"])
def test_partial_decode_matches_oracle(oracle, cls):
    data = corpus.class_bytes(cls, 40000, 5)
    comp = np.frombuffer(oracle.encode(data), np.uint8)
    for want in (0, 1, 7, 64, 65, 1000, 12345, 39990, 39995, 40000, 40001, 50000):
        tgt = np.full(want + 32, 0xCD, np.uint8)
        n = LZ4Codec.PartialDecode(comp, 0, comp.size, tgt, 0, want)
        ref_n, ref = oracle.decompress_partial(comp, want, want)
        ref_n = -1 if ref_n <= 0 else ref_n
        assert n == ref_n, (cls, want, n, ref_n)
        if n > 0:
            assert tgt[:n].tobytes() == ref[:n].tobytes() == data[:n].tobytes()
        assert (tgt[max(want, 0):] == 0xCD).all()


# ---- Decode with dictionary (LZ4Codec.cs:144-160; Issue64.cs: dictionary-chained records) -------
def _issue64_records():
    raw = open(os.path.join(GOLDEN, "issue64_input.bin"), "rb").read()
    want = np.frombuffer(open(os.path.join(GOLDEN, "issue64_output.bin"), "rb").read(), np.uint8)
    pos, recs = 20, []
    while raw[pos:pos + 4] == b"bv41":
        u, c = struct.unpack_from("<II", raw, pos + 4)
        recs.append((u, np.frombuffer(raw[pos + 12:pos + 12 + c], np.uint8)))
        pos += 12 + c
    return recs, want


def test_issue64_chained_record_with_dictionary(oracle):
    recs, want = _issue64_records()
    u0, c0 = recs[0]
    u1, c1 = recs[1]
    first = np.zeros(u0, np.uint8)
    assert LZ4Codec.Decode(c0, first) == u0 and first.tobytes() == want[:u0].tobytes()
    # external dictionary
    second = np.full(u1 + 16, 0xCD, np.uint8)
    assert LZ4Codec.Decode(c1, second[:u1], first) == u1
    assert second[:u1].tobytes() == want[u0:u0 + u1].tobytes() and (second[u1:] == 0xCD).all()
    # prefix placement (the dictionary ends where the target starts)
    both = np.full(u0 + u1, 0xCD, np.uint8)
    both[:u0] = first
    assert LZ4Codec.Decode(c1, both[u0:], both[:u0]) == u1
    assert both.tobytes() == want[:u0 + u1].tobytes()
    # without its dictionary the record is rejected, exactly as the oracle rejects it
    n, _ = oracle.decompress_safe(c1, u1)
    assert n < 0 and LZ4Codec.Decode(c1, np.zeros(u1, np.uint8)) == -1
    # too small a target
    assert LZ4Codec.Decode(c1, np.zeros(u1 - 1, np.uint8), first) == -1
    # the overload with offsets and lengths (LZ4Codec.cs:250-266), dictionary may be null when empty
    big = np.full(u1 + 20, 0xCD, np.uint8)
    assert LZ4Codec.Decode(c1, 0, c1.size, big, 10, u1, first, 0, first.size) == u1
    assert big[10:10 + u1].tobytes() == want[u0:u0 + u1].tobytes() and (big[:10] == 0xCD).all() and (big[10 + u1:] == 0xCD).all()
    assert LZ4Codec.Decode(c0, 0, c0.size, np.zeros(u0, np.uint8), 0, u0, None, 0, 0) == u0


def test_dictionary_batch_matches_oracle(oracle):
    """independent streams, each with its own dictionary: liblz4-made streams (LZ4_compress_fast_continue
    after LZ4_loadDict is what the reference's chain encoder does) are not available here, so the streams are
    issue64's record plus blocks whose matches are re-pointed into the dictionary by construction"""
    recs, want = _issue64_records()
    u0 = recs[0][0]
    u1, c1 = recs[1]
    d0 = want[:u0]
    rng = np.random.default_rng(9)
    cases = []
    for k in range(40):
        cut = int(rng.integers(0, u0))
        cases.append((c1, int(u1 + rng.integers(-2, 3)), d0[cut:] if k % 3 else d0))
    from oracle_lib import SystemLZ4
    sysl = SystemLZ4()
    if sysl.available:
        for k, cls in enumerate(corpus.SILESIA_NAMES):
            dn, n = (65536, 65536) if k % 2 else (int(rng.integers(100, 70000)), int(rng.integers(1000, 65536)))
            text = corpus.class_bytes(cls, dn + n, 4)
            dd, data = text[:dn].copy(), text[dn:].copy()
            c = sysl.compress_with_dict(data, dd)
            got = np.zeros(n, np.uint8)
            assert LZ4Codec.Decode(c, got, dd) == n and got.tobytes() == data.tobytes()
            cases += [(c, n, dd), (c, n - 1, dd), (c, n + 7, dd), (c, n, dd[1:])]
    comps = [c for c, _, _ in cases]
    src, soff, slen = pack_blocks(comps)
    dpk, doffs, dlens = pack_blocks([d for _, _, d in cases])
    caps = np.array([c for _, c, _ in cases], np.int32)
    dst, doff = make_arena(caps)
    out = LZ4Codec.DecodeDictBatchPacked(src, soff, slen, dst, doff, caps, dpk, doffs, dlens, flags=FLAG_RAW_RETURN)
    for i, (c, cap, d) in enumerate(cases):
        n, ref = oracle.decompress_using_dict(c, cap, np.ascontiguousarray(d))
        assert out[i] == n, (i, out[i], n)
        if n > 0:
            assert dst[int(doff[i]):int(doff[i]) + n].tobytes() == ref[:n].tobytes()


def test_enforce32_switch(oracle):
    """LZ4Codec.Enforce32 = true: the 32-bit engine's bytes for blocks of 64 KiB and more, everything else unchanged;
    the switch is process-wide like the reference's (LL.Enforce32) and is restored here"""
    big = [corpus.class_bytes("dickens", 70000, 1), corpus.lorem(150000)]
    small = corpus.class_bytes("dickens", 60000, 2)
    assert LZ4Codec.Enforce32 is False
    try:
        LZ4Codec.Enforce32 = True
        enc = LZ4Codec.EncodeBatch(big + [small])
        for b, e in zip(big, enc):
            r, w = oracle.compress_fast_x32(b)
            assert e == w[:r].tobytes() and e != oracle.encode(b)
        assert enc[2] == oracle.encode(small)
        assert LZ4Codec.DecodeBatch(enc, [b.size for b in big] + [small.size]) == [b.tobytes() for b in big] + [small.tobytes()]
        p = LZ4Pickler.Pickle(big[0])
        assert LZ4Pickler.Unpickle(p) == big[0].tobytes()
    finally:
        LZ4Codec.Enforce32 = False
    assert LZ4Codec.EncodeBatch(big)[0] == oracle.encode(big[0])
