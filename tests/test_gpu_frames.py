"""GPU parity of the block-stream / frame layer, through the C ABI.
Reads like the reference's Streams tests: encode with one side, decode with the other
(Streams.Tests/EncoderTests.cs, DecoderTests.cs use lz4.exe as the other side; here it is the system
liblz4's LZ4F_* and the oracle), checksum variants (ChecksumTests.cs), block encoders (Tests/EncoderTests)."""
import numpy as np
import pytest
import torch   # noqa: F401  (before libk4lz4 is loaded: torch must initialise its HIP runtime first)

from oracle_lib import FrameOracle
from test_frame_layer import LZ4F, _contents
from k4os.compression.lz4_amd import (LZ4Frame, LZ4EncoderSettings, LZ4Level, LZ4BlockEncoder, LZ4BlockDecoder, EncoderAction,
                                      TopupAndEncode, FlushAndEncode, DecodeAndDrain, LZ4Codec, corpus, xxh32_many)
from k4os.compression.lz4_amd import frames as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fo(oracle):
    return FrameOracle(oracle)


@pytest.fixture(scope="module")
def lz4f():
    try:
        return LZ4F()
    except OSError:
        pytest.skip("liblz4.so.1 not present")


def test_xxh32_many_matches_reference_implementation(fo):
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(3)
    bufs = [rng.integers(0, 256, n, dtype=np.uint8) for n in list(range(0, 40)) + [1000, 65536, 65551, 1 << 20, 3_000_001]]
    got = xxh32_many(bufs)
    assert got.tolist() == [xxhash.xxh32(b.tobytes(), seed=0).intdigest() for b in bufs] == [fo.xxh32(b) for b in bufs]


@pytest.mark.parametrize("bsum,csum", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("block_size", [65536, 262144])
def test_frames_bit_exact_with_oracle_and_decodable_by_liblz4(fo, lz4f, bsum, csum, block_size):
    contents = _contents()
    s = LZ4EncoderSettings(BlockSize=block_size, BlockChecksum=bsum, ContentChecksum=csum)
    frames = LZ4Frame.EncodeBatch(contents, s)
    for data, fr in zip(contents, frames):
        assert fr == fo.frame_encode(data, block_size, 0, bsum, csum)
        r, out, used = lz4f.decompress(fr, data.size + 16)
        assert r == 0 and used == len(fr) and out == data.tobytes()
    back = LZ4Frame.DecodeBatch(frames)
    assert [b for b in back] == [c.tobytes() for c in contents]


def test_frame_levels_and_single_calls(fo):
    data = corpus.class_bytes("webster", 300000, 7)
    for level in (LZ4Level.L00_FAST, LZ4Level.L03_HC, LZ4Level.L06_HC):
        fr = LZ4Frame.Encode(data, LZ4EncoderSettings(CompressionLevel=level, BlockChecksum=True))
        assert fr == fo.frame_encode(data, 65536, int(level), True, False)
        assert LZ4Frame.Decode(fr) == data.tobytes()
    assert LZ4Frame.Decode(LZ4Frame.Encode(b"")) == b""
    with pytest.raises(F.NotImplementedException):
        LZ4Frame.Encode(data, LZ4EncoderSettings(ChainBlocks=True))
    fr = LZ4Frame.Encode(data, LZ4EncoderSettings(ContentLength=data.size))
    assert F.parse_frame(fr).descriptor.ContentLength == data.size and LZ4Frame.Decode(fr) == data.tobytes()


def test_decodes_frames_written_by_liblz4_including_linked_blocks(lz4f):
    contents = _contents() + [corpus.class_bytes("mozilla", 1_500_000, 9)]
    frames, want = [], []
    for data in contents:
        for linked in (False, True):
            for sums in (False, True):
                for bid in (4, 6):
                    frames.append(lz4f.compress(data, bid, linked, content_checksum=sums, block_checksum=sums, content_size=sums))
                    want.append(data.tobytes())
    assert any(F.parse_frame(f).descriptor.Chaining for f in frames)
    assert LZ4Frame.DecodeBatch(frames) == want


def test_frame_corruption_is_rejected(fo):
    data = corpus.class_bytes("dickens", 200000, 5)
    fr = LZ4Frame.Encode(data, LZ4EncoderSettings(BlockChecksum=True, ContentChecksum=True))
    info = F.parse_frame(fr)
    def flip(pos):
        b = bytearray(fr); b[pos] ^= 0x01
        return bytes(b)
    with pytest.raises(F.InvalidDataException, match="magic"):
        LZ4Frame.Decode(flip(0))
    with pytest.raises(F.InvalidDataException, match="header checksum"):
        LZ4Frame.Decode(flip(6))
    with pytest.raises(F.InvalidDataException, match="block checksum"):
        LZ4Frame.Decode(flip(info.block_off[1] + 10))
    with pytest.raises(F.InvalidDataException, match="content checksum"):
        LZ4Frame.Decode(flip(len(fr) - 1))
    with pytest.raises(EOFError):
        LZ4Frame.Decode(fr[:-5])
    # without block checksums a damaged block surfaces as a decode failure or a content checksum mismatch
    fr2 = LZ4Frame.Encode(data, LZ4EncoderSettings(ContentChecksum=True))
    i2 = F.parse_frame(fr2)
    b = bytearray(fr2); b[i2.block_off[0] + 5] ^= 0xFF
    with pytest.raises(F.InvalidDataException):
        LZ4Frame.Decode(bytes(b))


# ---- LZ4BlockEncoder / LZ4BlockDecoder (Encoders/*.cs) ------------------------------------------
def test_block_encoder_loop_like_the_frame_writer(oracle):
    """LZ4FrameWriter.WriteManyBytes: TopupAndEncode(forceEncode=false, allowCopy=true) until the source is
    consumed, then FlushAndEncode -- the blocks must be the oracle's (raw where they do not shrink)"""
    data = np.concatenate([corpus.class_bytes("dickens", 100000, 1), corpus.random_bytes(70000, 2), corpus.lorem(3000)])
    enc = LZ4BlockEncoder(LZ4Level.L00_FAST, 65536)
    target = np.zeros(LZ4Codec.MaximumOutputSize(enc.BlockSize), np.uint8)
    got, pos = [], 0
    while pos < data.size:
        action, loaded, encoded = TopupAndEncode(enc, data[pos:], target, False, True)
        pos += loaded
        if action in (EncoderAction.Encoded, EncoderAction.Copied):
            got.append((action, target[:encoded].tobytes()))
    action, encoded = FlushAndEncode(enc, target)
    if action in (EncoderAction.Encoded, EncoderAction.Copied):
        got.append((action, target[:encoded].tobytes()))
    want = []
    for p in range(0, data.size, 65536):
        b = data[p:p + 65536]
        e = oracle.encode(b)
        want.append((EncoderAction.Copied, b.tobytes()) if len(e) >= b.size else (EncoderAction.Encoded, e))
    assert got == want
    assert enc.EncodeBlocks([data[p:p + 65536] for p in range(0, data.size, 65536)]) == want
    # without allowCopy an incompressible block stays encoded
    enc2 = LZ4BlockEncoder(LZ4Level.L00_FAST, 65536)
    rnd = corpus.random_bytes(65536, 4)
    assert enc2.Topup(rnd) == 65536
    n = enc2.Encode(target, allowCopy=False)
    assert n == len(oracle.encode(rnd)) > rnd.size and enc2.BytesReady == 0
    # target too small: the reference throws
    from k4os.compression.lz4_amd import InvalidOperationException
    enc2.Topup(corpus.lorem(5000))
    with pytest.raises(InvalidOperationException):
        enc2.Encode(np.zeros(10, np.uint8))


def test_block_decoder_decode_drain(oracle):
    data = corpus.class_bytes("osdb", 65536, 3)
    comp = oracle.encode(data)
    dec = LZ4BlockDecoder(65536)
    assert dec.Decode(np.frombuffer(comp, np.uint8)) == data.size == dec.BytesReady
    out = np.zeros(data.size, np.uint8)
    dec.Drain(out, -data.size, data.size)
    assert out.tobytes() == data.tobytes()
    ok, n = DecodeAndDrain(dec, np.frombuffer(comp, np.uint8), out)
    assert ok and n == data.size
    ok, n = DecodeAndDrain(dec, np.frombuffer(comp, np.uint8), np.zeros(100, np.uint8))
    assert not ok
    from k4os.compression.lz4_amd import InvalidOperationException
    with pytest.raises(InvalidOperationException):
        dec.Decode(np.frombuffer(comp[:-3], np.uint8))
    small = LZ4BlockDecoder(1024)
    with pytest.raises(InvalidOperationException):
        small.Decode(np.frombuffer(comp, np.uint8))             # decoded block larger than the decoder's buffer


def test_device_resident_frame_encoder_equals_host_api(fo, lz4f):
    """encode_frames_device: contents stay in HBM, frames are laid out by k4_frame_blocks_kernel / k4_frame_edges_kernel;
    byte-identical to LZ4Frame.EncodeBatch and to the oracle"""
    import torch
    from k4os.compression.lz4_amd.device import DeviceCodec
    from k4os.compression.lz4_amd.frames import encode_frames_device
    from k4os.compression.lz4_amd import pack_blocks
    dc = DeviceCodec(0)
    contents = _contents()
    data_h, off, ln = pack_blocks(contents)
    data = torch.from_numpy(data_h).to(dc.device)
    for bsum, csum, bs, clen in ((False, False, 65536, False), (True, True, 65536, False), (True, False, 262144, True)):
        s = LZ4EncoderSettings(BlockSize=bs, BlockChecksum=bsum, ContentChecksum=csum, ContentLength=0 if clen else None)
        frames, foff, flen = encode_frames_device(dc, data, off.astype(np.int64), np.array([c.size for c in contents], np.int64), s)
        torch.cuda.synchronize()
        fh, fl = frames.cpu().numpy(), flen.cpu().numpy()
        for f, c in enumerate(contents):
            got = fh[int(foff[f]):int(foff[f]) + int(fl[f])].tobytes()
            if not clen:
                assert got == fo.frame_encode(c, bs, 0, bsum, csum), (f, bsum, csum)
            r, out, used = lz4f.decompress(got, c.size + 16)
            assert r == 0 and used == len(got) and out == c.tobytes()
            assert LZ4Frame.Decode(got) == c.tobytes()
