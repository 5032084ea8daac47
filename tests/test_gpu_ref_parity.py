"""The HIP path against THE REFERENCE ITSELF, compiled here (oracle/_ref/libk4ref.so = the reference's own LL64 / LL32
engine files respelled as C++ by oracle/make_ref.py; the prebuilt library travels to the GPU box with the snapshot).
The other GPU tests compare with the oracle restatement, which tests/test_ref_pins.py pins to this library byte for byte;
these compare the kernels with it directly, through the C ABI, on the graded data."""
import ctypes as C

import numpy as np
import pytest

from k4os.compression.lz4_amd import LZ4Codec, LZ4Level, LZ4Pickler, corpus, pack_blocks, make_arena
from k4os.compression.lz4_amd._native import FLAG_RAW_RETURN, load_library
from oracle_lib import RefEngine, REFERENCE_PRESENT
from test_oracle_pins import _pool_map

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref():
    try:
        return RefEngine()
    except FileNotFoundError as e:
        assert not REFERENCE_PRESENT
        pytest.skip(str(e))


def _enc(ref, b, level=0):
    r, d = ref.compress_fast(b) if level == 0 else ref.compress_hc(b, level)
    return d[:r].tobytes()


def test_bench_batch_l00_every_block_vs_compiled_reference(ref):
    """configs[1]: all 4096 blocks of bench.py's batch, encoded on the GPU, bytes equal to LL64.LZ4_compress_fast's"""
    blocks = corpus.silesia_like_blocks(4096, 65536, seed=2)
    n = blocks.shape[0]
    off = np.arange(n, dtype=np.uint64) * 65536
    lens = np.full(n, 65536, np.int32)
    caps = np.full(n, LZ4Codec.MaximumOutputSize(65536), np.int32)
    dst, doff = make_arena(caps, fill=0xCD)
    out = LZ4Codec.EncodeBatchPacked(blocks.reshape(-1), off, lens, dst, doff, caps)
    want = _pool_map(lambda i: _enc(ref, blocks[i]), range(n))
    bad = [i for i in range(n) if out[i] != len(want[i]) or dst[int(doff[i]):int(doff[i]) + int(out[i])].tobytes() != want[i]]
    assert not bad, bad[:10]
    # and the decoder on the reference's own streams: return values and bytes
    src, soff, slen = pack_blocks([np.frombuffer(w, np.uint8) for w in want])
    back, boff = make_arena(lens, fill=0xCD)
    dl = LZ4Codec.DecodeBatchPacked(src, soff, slen, back, boff, lens, flags=FLAG_RAW_RETURN)
    assert (dl == 65536).all() and np.array_equal(back[:n * 65536].reshape(n, 65536), blocks)


@pytest.mark.parametrize("level,count", [(3, 384), (9, 96), (10, 48), (12, 24)])
def test_bench_blocks_hc_levels_vs_compiled_reference(ref, level, count):
    """configs[4] and the other levels ChecksumBlockTests.cs:125-172 holds goldens for"""
    blocks = corpus.silesia_like_blocks(4096, 65536, seed=2)[:count]
    enc = LZ4Codec.EncodeBatch(list(blocks), level=LZ4Level(level))
    want = _pool_map(lambda i: _enc(ref, blocks[i], level), range(count))
    bad = [i for i in range(count) if enc[i] != want[i]]
    assert not bad, bad[:10]


def test_ragged_messages_both_engines_vs_compiled_reference(ref):
    """byU16 / byU32 switch at 65 547 and messages up to 4 MiB (configs[3] shapes); with Enforce32 the 32-bit engine's bytes
    (x32/LL32.tools.cs:141-148) -- this arm was compared with an unpinned oracle arm until round 4"""
    sizes = [1, 12, 13, 100, 65535, 65536, 65546, 65547, 70000, 300000, 1 << 20, (4 << 20) - 3]
    blocks = [corpus.class_bytes(corpus.SILESIA_NAMES[i % 12], s, 40 + i) for i, s in enumerate(sizes)]
    enc = LZ4Codec.EncodeBatch(blocks)
    for b, e in zip(blocks, enc):
        assert e == _enc(ref, b), b.size
    try:
        LZ4Codec.Enforce32 = True
        enc32 = LZ4Codec.EncodeBatch(blocks)
        text = corpus.class_bytes("dickens", 70000, 3)           # compressible, so the envelope carries a block and not the raw bytes
        pick = LZ4Pickler.Pickle(text)
    finally:
        LZ4Codec.Enforce32 = False
    differ = 0
    for b, e, e64 in zip(blocks, enc32, enc):
        r, d = ref.compress_fast_x32(b)
        assert e == d[:r].tobytes(), b.size
        differ += e != e64
    assert differ >= 4
    # the pickle of a >= 64 KiB message under Enforce32 carries the LL32 block (the flag was dropped)
    r, d = ref.compress_fast_x32(text)
    assert pick[0] != 0 and pick.endswith(d[:r].tobytes())


def test_acceleration_through_the_llxx_seam(ref):
    """LLxx.LZ4_compress_fast passes `acceleration` through (LLxx.cs:65-75): k4lz4_compress_fast with 1, 2, 8 and an out of
    range value (< 1 -> ACCELERATION_DEFAULT, LL64.fast.cs:522)"""
    lib = load_library()          # (the argument types are the ones _native.SYMBOLS declared: nothing is changed on the shared handle)
    for i, size in enumerate((3000, 65536, 65600, 200000)):
        for cls in ("dickens", "xml", "mr", "sao"):
            data = corpus.class_bytes(cls, size, i)
            cap = LZ4Codec.MaximumOutputSize(size)
            for acc in (1, 2, 8, 0, 70):
                dst = np.full(cap, 0xCD, np.uint8)
                r = lib.k4lz4_compress_fast(data.ctypes.data, dst.ctypes.data, size, cap, acc)
                want_r, want = ref.compress_fast(data, accel=acc)
                assert r == want_r and dst.tobytes() == want.tobytes(), (cls, size, acc)


def test_mutated_streams_vs_compiled_reference(ref):
    """LZ4_decompress_safe's return value (negative error position included) and bytes on 1 200 mutants"""
    rng = np.random.default_rng(32)
    comps, caps = [], []
    for name, n in (("dickens", 3000), ("xml", 6000), ("mr", 66000)):
        data = corpus.class_bytes(name, n, 4)
        r, d = ref.compress_fast(data)
        good = d[:r].copy()
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
        n, want = ref.decompress_safe(c, int(cap))
        assert out[i] == n, i
        if n > 0:
            assert dst[int(doff[i]):int(doff[i]) + n].tobytes() == want[:n].tobytes()


@pytest.mark.parametrize("level", [4, 5, 6, 7, 8, 11])
def test_remaining_hc_levels_vs_compiled_reference(ref, level):
    """the rest of LZ4Level (clTable, LL64.high.cs:1124-1138): 24 bench blocks (two of every class) per level"""
    blocks = corpus.silesia_like_blocks(4096, 65536, seed=2)[:24]
    enc = LZ4Codec.EncodeBatch(list(blocks), level=LZ4Level(level))
    want = _pool_map(lambda i: _enc(ref, blocks[i], level), range(24))
    assert [i for i in range(24) if enc[i] != want[i]] == []


def test_partial_and_dictionary_decode_vs_compiled_reference(ref, syslz4):
    """next-row N1 against LL64.LZ4_decompress_safe_partial / _usingDict themselves (LL64.dec.cs:479-556): return values and bytes"""
    from k4os.compression.lz4_amd._native import FLAG_RAW_RETURN
    for cls in ("dickens", "xml", "mr"):
        data = corpus.class_bytes(cls, 30000, 8)
        r, d = ref.compress_fast(data)
        comp = d[:r].copy()
        for want in (0, 1, 13, 64, 1000, 12345, 29990, 29999, 30000, 30001, 40000):
            tgt = np.full(want + 32, 0xCD, np.uint8)
            n = LZ4Codec.PartialDecode(comp, 0, comp.size, tgt, 0, want)
            rn, rd = ref.decompress_partial(comp, want, want)
            assert n == (-1 if rn <= 0 else rn), (cls, want, n, rn)
            if n > 0:
                assert tgt[:n].tobytes() == rd[:n].tobytes()
        dictionary = corpus.class_bytes(cls, 40000, 9)
        block = syslz4.compress_with_dict(data, dictionary)
        cases = [(block, data.size, dictionary), (block, data.size - 1, dictionary), (block, data.size + 9, dictionary),
                 (block, data.size, dictionary[1:]), (block[:-3], data.size, dictionary)]
        src, soff, slen = pack_blocks([c for c, _, _ in cases])
        dpk, doffs, dlens = pack_blocks([x for _, _, x in cases])
        caps = np.array([c for _, c, _ in cases], np.int32)
        dst, doff = make_arena(caps)
        out = LZ4Codec.DecodeDictBatchPacked(src, soff, slen, dst, doff, caps, dpk, doffs, dlens, flags=FLAG_RAW_RETURN)
        for i, (c, cap, dd) in enumerate(cases):
            rn, rd = ref.decompress_using_dict(c, cap, np.ascontiguousarray(dd))
            assert out[i] == rn, (cls, i, out[i], rn)
            if rn > 0:
                assert dst[int(doff[i]):int(doff[i]) + rn].tobytes() == rd[:rn].tobytes()


def test_long_length_fields_vs_compiled_reference(ref, oracle):
    """length fields that are long runs of 255 (4 KiB of random bytes: 17 bytes; 70 000 equal bytes: 274), whole, cut and damaged
    inside the runs: LZ4_decompress_safe's return value (error position included) and bytes, against LL64.dec.cs compiled here"""
    from stream_cases import long_field_streams
    cases = long_field_streams(oracle, np.random.default_rng(78))
    src, soff, slen = pack_blocks([c for c, _, _ in cases])
    caps = np.array([k for _, k, _ in cases], np.int32)
    dst, doff = make_arena(caps + 32, fill=0xCD)
    out = LZ4Codec.DecodeBatchPacked(src, soff, slen, dst, doff, caps, flags=FLAG_RAW_RETURN)
    for i, (c, cap, defined) in enumerate(cases):
        n, want = ref.decompress_safe(c, int(cap))
        assert out[i] == n, (i, c.size, cap, out[i], n)
        if n > 0 and defined:
            assert dst[int(doff[i]):int(doff[i]) + n].tobytes() == want[:n].tobytes()
        assert (dst[int(doff[i]) + int(cap):int(doff[i]) + int(cap) + 32] == 0xCD).all()
