"""Frame layer without a GPU: oracle pins, host logic, kernels under the emulator.

Pins of the oracle (oracle/k4lz4_oracle_frame.c):
  * XXH32 against the python `xxhash` package (bindings of the xxHash reference implementation) -- the
    reference's own XXH32 is NuGet K4os.Hash.xxHash 1.0.8, not under /root/reference;
  * whole frames against the system liblz4's LZ4F_* (lz4 1.9.3): oracle frames decode there, liblz4
    frames (independent AND linked blocks, with checksums and content size) decode in the oracle.
The product's host logic (parse_frame / assemble_frame / frame_header) is checked against the oracle's
frames byte for byte; the kernels (k4_xxh32_kernel, k4_allow_copy_kernel, k4_decode_chain_kernel) run
under the wave emulator against the oracle."""
import ctypes as C
import struct

import numpy as np
import pytest

from emu_lib import Emu
from oracle_lib import FrameOracle
from k4os.compression.lz4_amd import corpus, LZ4Codec, pack_blocks
from k4os.compression.lz4_amd import frames as F
from k4os.compression.lz4_amd.encoders import LZ4BlockEncoder, LZ4BlockDecoder, EncoderAction


@pytest.fixture(scope="module")
def emu():
    return Emu()


@pytest.fixture(scope="module")
def fo(oracle):
    return FrameOracle(oracle)


class LZ4F:
    """liblz4's frame API through ctypes"""

    class Prefs(C.Structure):
        _fields_ = [("blockSizeID", C.c_int), ("blockMode", C.c_int), ("contentChecksumFlag", C.c_int), ("frameType", C.c_int),
                    ("contentSize", C.c_ulonglong), ("dictID", C.c_uint), ("blockChecksumFlag", C.c_int),
                    ("compressionLevel", C.c_int), ("autoFlush", C.c_uint), ("favorDecSpeed", C.c_uint), ("reserved", C.c_uint * 3)]

    def __init__(self):
        self.lib = C.CDLL("liblz4.so.1")
        L = self.lib
        L.LZ4F_compressFrameBound.restype = C.c_size_t
        L.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
        L.LZ4F_compressFrame.restype = C.c_size_t
        L.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.LZ4F_isError.argtypes = [C.c_size_t]
        L.LZ4F_createDecompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        L.LZ4F_freeDecompressionContext.argtypes = [C.c_void_p]
        L.LZ4F_decompress.restype = C.c_size_t
        L.LZ4F_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]

    def compress(self, data: np.ndarray, block_id=4, linked=False, content_checksum=False, block_checksum=False, content_size=False):
        p = LZ4F.Prefs()
        p.blockSizeID, p.blockMode = block_id, 0 if linked else 1
        p.contentChecksumFlag, p.blockChecksumFlag = int(content_checksum), int(block_checksum)
        p.contentSize = data.size if content_size else 0
        cap = self.lib.LZ4F_compressFrameBound(data.size, C.byref(p))
        dst = np.zeros(cap, np.uint8)
        n = self.lib.LZ4F_compressFrame(dst.ctypes.data, cap, data.ctypes.data, data.size, C.byref(p))
        assert not self.lib.LZ4F_isError(n)
        return dst[:n].tobytes()

    def decompress(self, frame: bytes, cap: int):
        ctx = C.c_void_p()
        assert self.lib.LZ4F_createDecompressionContext(C.byref(ctx), 100) == 0
        try:
            src = np.frombuffer(frame, np.uint8)
            dst = np.zeros(max(cap, 1), np.uint8)
            ds, ss = C.c_size_t(cap), C.c_size_t(src.size)
            r = self.lib.LZ4F_decompress(ctx, dst.ctypes.data, C.byref(ds), src.ctypes.data, C.byref(ss), None)
            return r, dst[:ds.value].tobytes(), ss.value
        finally:
            self.lib.LZ4F_freeDecompressionContext(ctx)


@pytest.fixture(scope="module")
def lz4f():
    try:
        return LZ4F()
    except OSError:
        pytest.skip("liblz4.so.1 not present")


def _contents():
    return [corpus.class_bytes("dickens", 200000, 1), corpus.class_bytes("x-ray", 70000, 2), corpus.lorem(5),
            corpus.random_bytes(65536, 3), np.zeros(0, np.uint8), corpus.class_bytes("xml", 65537, 4), corpus.repeated(7, 300000)]


# ---- oracle pins -------------------------------------------------------------------------------
def test_oracle_xxh32_equals_reference_implementation(fo):
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(1)
    for n in list(range(0, 50)) + [100, 255, 256, 1000, 65536, 70001, 1 << 20]:
        d = rng.integers(0, 256, n, dtype=np.uint8)
        for seed in (0, 1, 0xDEADBEEF):
            assert fo.xxh32(d, seed) == xxhash.xxh32(d.tobytes(), seed=seed).intdigest(), (n, seed)
    assert fo.xxh32(b"") == 0x02CC5D05 and fo.xxh32(b"abc") == 0x32D153FF      # published test vectors


def test_oracle_frames_interoperate_with_liblz4(fo, lz4f):
    for data in _contents():
        for bs, bid in ((65536, 4), (262144, 5)):
            for bsum in (False, True):
                for csum in (False, True):
                    fr = fo.frame_encode(data, bs, 0, bsum, csum)
                    r, out, used = lz4f.decompress(fr, data.size + 16)
                    assert r == 0 and used == len(fr) and out == data.tobytes()
                    n, out2, used2 = fo.frame_decode(fr, data.size + 16)
                    assert n == data.size and out2 == data.tobytes() and used2 == len(fr)
                    for linked in (False, True):
                        theirs = lz4f.compress(data, bid, linked, csum, bsum, content_size=csum)
                        n, out3, used3 = fo.frame_decode(theirs, data.size + 16)
                        assert n == data.size and out3 == data.tobytes() and used3 == len(theirs), (data.size, bs, linked)


def test_oracle_frame_rejects_corruption(fo):
    data = corpus.class_bytes("dickens", 100000, 5)
    fr = bytearray(fo.frame_encode(data, 65536, 0, True, True))
    def dec(b):
        return fo.frame_decode(bytes(b), 200000)[0]
    assert dec(fr) == data.size
    bad = bytearray(fr); bad[0] ^= 1; assert dec(bad) == -1
    bad = bytearray(fr); bad[6] ^= 1; assert dec(bad) == -3            # header checksum byte
    bad = bytearray(fr); bad[4] ^= 0x10; assert dec(bad) == -3         # FLG changed under the checksum
    bad = bytearray(fr); bad[20] ^= 1; assert dec(bad) == -5           # payload byte: block checksum
    bad = bytearray(fr); bad[-1] ^= 1; assert dec(bad) == -7           # content checksum
    assert dec(fr[:-3]) == -4 and dec(fr[:30]) == -4


# ---- host logic of the product against the oracle ----------------------------------------------
def test_parse_and_assemble_match_oracle_frames(fo, oracle):
    for data in _contents():
        for bsum in (False, True):
            for csum in (False, True):
                fr = fo.frame_encode(data, 65536, 0, bsum, csum)
                info = F.parse_frame(fr)
                d = info.descriptor
                assert (d.BlockSize, d.Chaining, d.BlockChecksum, d.ContentChecksum, d.ContentLength, d.Dictionary) == \
                       (65536, False, bsum, csum, None, None)
                assert info.consumed == len(fr) and info.header == F.frame_header(d)
                assert ((fo.xxh32(info.header) >> 8) & 0xFF) == info.header_checksum
                payloads = [fr[o:o + (l & 0x7FFFFFFF)] for o, l in zip(info.block_off, info.block_len)]
                raw = [bool(l & 0x80000000) for l in info.block_len]
                # blocks: raw exactly where the encoder did not shrink them, else the oracle's encoding
                blocks = [data[p:p + 65536] for p in range(0, data.size, 65536)]
                assert len(blocks) == len(payloads)
                for b, p, r in zip(blocks, payloads, raw):
                    enc = oracle.encode(b)
                    assert r == (len(enc) >= b.size) and p == (b.tobytes() if r else enc)
                if bsum:
                    assert info.block_checksum == [fo.xxh32(p) for p in payloads]
                if csum:
                    assert info.content_checksum == fo.xxh32(data)
                again = F.assemble_frame(d, fo.xxh32(info.header), payloads, raw, info.block_checksum if bsum else None,
                                         info.content_checksum)
                assert again == fr


def test_parse_frame_errors_and_foreign_headers(lz4f):
    data = corpus.lorem(600000)                            # more than one block: liblz4 keeps the linked mode
    theirs = lz4f.compress(data, 5, linked=True, content_checksum=True, block_checksum=True, content_size=True)
    info = F.parse_frame(theirs)
    d = info.descriptor
    assert d.Chaining and d.BlockChecksum and d.ContentChecksum and d.ContentLength == data.size and d.BlockSize == 262144
    assert info.consumed == len(theirs) and len(info.block_checksum) == len(info.block_off)
    with pytest.raises(F.InvalidDataException):
        F.parse_frame(b"\x00" + theirs[1:])
    with pytest.raises(EOFError):
        F.parse_frame(theirs[:-2])
    with pytest.raises(EOFError):
        F.parse_frame(theirs[:5])
    with pytest.raises(ValueError):
        F.max_block_size_code((4 << 20) + 1)
    assert [F.max_block_size_code(x) for x in (1, 65536, 65537, 1 << 20, 4 << 20)] == [4, 4, 5, 6, 7]


# ---- kernels under the emulator ------------------------------------------------------------------
def test_xxh32_kernel_matches_oracle(emu, fo):
    rng = np.random.default_rng(2)
    lens = list(range(0, 70)) + [255, 256, 257, 1000, 4099, 65536, 65551, 200001] + [int(x) for x in rng.integers(0, 3000, 40)]
    bufs = [rng.integers(0, 256, n, dtype=np.uint8) for n in lens]
    data, off, _ = pack_blocks(bufs)
    ln = np.array(lens, np.uint64)
    for seed in (0, 0x9E3779B1):
        got = emu.xxh32_batch(data, off, ln, seed)
        want = [fo.xxh32(b, seed) for b in bufs]
        assert got.tolist() == want


def test_allow_copy_kernel(emu, oracle):
    blocks = [corpus.random_bytes(5000, 1), corpus.lorem(5000), corpus.random_bytes(1, 2), corpus.lorem(13), np.zeros(0, np.uint8),
              corpus.class_bytes("x-ray", 65536, 3), corpus.repeated(1, 4)]
    src, soff, slen = pack_blocks(blocks)
    caps = np.array([LZ4Codec.MaximumOutputSize(b.size) for b in blocks], np.int32)
    caps[1] = 100                                      # too small for the text block: encoder reports failure
    from emu_lib import arena
    dst, doff, dcap = arena(caps)
    out = emu.encode_batch(src, soff, slen, dst, doff, dcap)
    out = emu.allow_copy(src, soff, slen, dst, doff, dcap, out)
    for i, b in enumerate(blocks):
        enc = oracle.encode(b) if b.size else b""
        if b.size == 0 or i == 1:
            assert out[i] == 0
        elif len(enc) >= b.size:
            assert out[i] == -b.size and dst[int(doff[i]):int(doff[i]) + b.size].tobytes() == b.tobytes()
        else:
            assert out[i] == len(enc) and dst[int(doff[i]):int(doff[i]) + len(enc)].tobytes() == enc
        assert (dst[int(doff[i]) + caps[i]:int(doff[i]) + caps[i] + 8] == 0xCD).all()


def _streams_from_frames(frames):
    src, foff, _ = pack_blocks([np.frombuffer(f, np.uint8) for f in frames])
    infos = [F.parse_frame(f) for f in frames]
    blk_off, blk_len, first, nblk = [], [], [], []
    for f, i in enumerate(infos):
        first.append(len(blk_off)); nblk.append(len(i.block_off))
        blk_off += [int(foff[f]) + o for o in i.block_off]; blk_len += i.block_len
    return (src, np.array(blk_off or [0], np.uint64), np.array(blk_len or [0], np.uint32), np.array(first, np.uint64),
            np.array(nblk, np.uint32), np.array([i.descriptor.BlockSize for i in infos], np.int32),
            np.array([int(i.descriptor.Chaining) for i in infos], np.uint8), infos)


def test_chain_decode_kernel_matches_oracle(emu, fo, lz4f):
    """whole frames through k4_decode_chain_kernel: oracle-made independent frames and liblz4-made LINKED
    frames (every block needs the previous 64 KiB), plus truncated targets and a corrupted block"""
    contents = _contents()
    frames, want = [], []
    for k, data in enumerate(contents):
        frames.append(fo.frame_encode(data, 65536, 0, False, False)); want.append(data)
        frames.append(lz4f.compress(data, 4, linked=True)); want.append(data)
        if k % 2 == 0:
            frames.append(lz4f.compress(data, 5, linked=True, block_checksum=True, content_checksum=True)); want.append(data)
    src, bo, bl, first, nblk, bsize, chained, infos = _streams_from_frames(frames)
    caps = np.array([max(w.size, 1) for w in want], np.uint64)
    caps[3] = want[3].size - 1 if want[3].size > 1 else caps[3]          # one target too small
    doff = np.zeros(len(frames), np.uint64); doff[1:] = np.cumsum(caps[:-1] + 32)
    dst = np.full(int(caps.sum()) + 32 * len(frames) + 64, 0xCD, np.uint8)
    out = emu.decode_chain_batch(src, bo, bl, first, nblk, bsize, chained, dst, doff, caps)
    for f, w in enumerate(want):
        n_ref, ref, _ = fo.frame_decode(frames[f], int(caps[f]))
        if n_ref >= 0:
            assert out[f] == w.size == n_ref and dst[int(doff[f]):int(doff[f]) + w.size].tobytes() == w.tobytes(), f
        else:
            assert out[f] == n_ref, (f, out[f], n_ref)                    # -9: target too small
        assert (dst[int(doff[f]) + int(caps[f]):int(doff[f]) + int(caps[f]) + 32] == 0xCD).all()
    # a corrupted compressed block
    bad = bytearray(frames[1]); info = infos[1]
    bad[info.block_off[1] + 3] ^= 0xFF; bad[info.block_off[1] + 4] ^= 0xFF
    src, bo, bl, first, nblk, bsize, chained, _ = _streams_from_frames([bytes(bad)])
    cap1 = np.array([want[1].size], np.uint64)
    d1 = np.zeros(want[1].size + 64, np.uint8)
    o1 = emu.decode_chain_batch(src, bo, bl, first, nblk, bsize, chained, d1, np.zeros(1, np.uint64), cap1)
    n_ref = fo.frame_decode(bytes(bad), want[1].size)[0]
    assert (o1[0] < 0) == (n_ref < 0) and (o1[0] == n_ref or n_ref >= 0)


def test_chain_decode_pair_kernel_matches_oracle(emu, fo, lz4f):
    """the same through k4_decode_chain_pair_kernel: one wave parses the stream's blocks (and may be blocks ahead), the
    other copies"""
    emu.pair = True
    try:
        test_chain_decode_kernel_matches_oracle(emu, fo, lz4f)
    finally:
        emu.pair = False


# ---- block encoder / decoder state machines (host logic; compute calls need the device) -------------
def test_block_encoder_topup_bookkeeping():
    enc = LZ4BlockEncoder(blockSize=1000)                    # rounded up to 1 KiB
    assert enc.BlockSize == 1024 and enc.BytesReady == 0
    data = corpus.lorem(3000)
    assert enc.Topup(data) == 1024 and enc.BytesReady == 1024
    assert enc.Topup(data, 1024) == 0                       # block full
    assert LZ4BlockEncoder(blockSize=65537).BlockSize == 66560
    dec = LZ4BlockDecoder(blockSize=100)
    assert dec.BlockSize == 1024 and dec.BytesReady == 0
    assert dec.Inject(data[:500]) == 500 and dec.BytesReady == 500
    out = np.zeros(200, np.uint8)
    dec.Drain(out, -500, 200)
    assert out.tobytes() == data[:200].tobytes()
    dec.Drain(out, -100, 100)
    assert out[:100].tobytes() == data[400:500].tobytes()
    from k4os.compression.lz4_amd.encoders import InvalidOperationException
    with pytest.raises(InvalidOperationException):
        dec.Drain(out, -501, 10)
    with pytest.raises(InvalidOperationException):
        dec.Drain(out, -10, 11)
    with pytest.raises(InvalidOperationException):
        dec.Inject(np.zeros(2000, np.uint8))
    assert dec.Peek(-100).tobytes() == data[400:500].tobytes()
    assert list(EncoderAction) == [EncoderAction.None_, EncoderAction.Loaded, EncoderAction.Encoded, EncoderAction.Copied]


def test_frame_decode_arena_follows_the_input_not_the_header(oracle, monkeypatch):
    """A frame that claims 4 MiB blocks but holds thousands of tiny ones must not make the decoder size its arena from the
    claim (N x 4 MiB): slots follow what the stored bytes can produce (255 per input byte at most) and launches are cut at a
    byte budget.  The kernels are replaced by the oracle here (no GPU): this is about the host's index work."""
    seen = []

    def fake_decode(src, soff, slen, dst, doff, cap, flags=0, ctx=None):
        seen.append((len(slen), int(np.asarray(cap, np.int64).sum())))
        out = np.empty(len(slen), np.int32)
        for i in range(len(slen)):
            r, d = oracle.decompress_safe(src[int(soff[i]):int(soff[i]) + int(slen[i])], int(cap[i]))
            out[i] = r if r > 0 else -1
            if r > 0:
                dst[int(doff[i]):int(doff[i]) + r] = d[:r]
        return out

    monkeypatch.setattr(LZ4Codec, "DecodeBatchPacked", staticmethod(fake_decode))
    pieces = [np.tile(corpus.lorem(37 + (i % 11)), 8 + i % 5) for i in range(3000)]
    records = []
    for p in pieces:
        c = oracle.encode(p)
        records.append((c, False) if len(c) < p.size else (p.tobytes(), True))
    desc = F.LZ4Descriptor(None, False, False, False, None, 4 << 20)          # BD = 7: 4 MiB blocks, independent
    import xxhash
    frame = F.assemble_frame(desc, xxhash.xxh32(F.frame_header(desc)).intdigest(), [r[0] for r in records], [r[1] for r in records], None, None)
    info = F.parse_frame(np.frombuffer(frame, np.uint8))
    monkeypatch.setattr(F.LZ4Frame, "ARENA_BUDGET", 1 << 20)
    outs = F.LZ4Frame._decode_streams([np.frombuffer(frame, np.uint8)], [info], None)
    assert outs[0].tobytes() == b"".join(p.tobytes() for p in pieces)
    assert len(seen) > 1 and all(total <= (1 << 20) for _, total in seen)     # several launches, each under the budget
    assert sum(total for _, total in seen) < 255 * len(frame) + 32 * len(pieces)   # nowhere near 3000 x 4 MiB
    assert F.LZ4Frame._block_cap(4 << 20, 10) == 255 * 10 + 32 and F.LZ4Frame._block_cap(65536, 60000) == 65536
