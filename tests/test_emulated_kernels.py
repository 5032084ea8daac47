"""The product's kernel source (k4os/compression/lz4_amd/csrc/*.hpp), compiled by g++ against the
host wave emulator (tests/emu), compared with the oracle.  This checks the kernels' *logic*
(speculative probe rounds, in-window duplicate resolution, token parse, accept/reject rules,
envelope arithmetic) on the CPU; the same comparisons run against the real gfx950 build in the
`-m gpu` tests.  Nothing here is a product path."""
import os

import numpy as np
import pytest

import adversarial_blocks

from emu_lib import Emu, pack, arena
from k4os.compression.lz4_amd import corpus

FLAG_RAW = 1
FLAG_WRITER = 2


@pytest.fixture(scope="module")
def emu():
    return Emu()


def _fixture_blocks():
    b = [np.frombuffer(corpus.QUICK_FOX, np.uint8)]
    b += [corpus.lorem(n) for n in (1, 2, 11, 12, 13, 14, 15, 64, 1000, 4096, 0x7FFF, 65535, 65536, 65546)]
    b += [corpus.repeated(0xAA, n) for n in (1, 13, 15, 17, 33, 67, 1000, 65536)]
    b += [corpus.repeated(0, 300), np.zeros(0, np.uint8)]
    b += [corpus.random_bytes(n, n) for n in (5, 100, 5000, 66000)]
    b += [corpus.class_bytes(name, 20000 + 3000 * i, 11) for i, name in enumerate(corpus.SILESIA_NAMES)]
    # short-period and near-window-edge patterns
    b += [np.tile(np.arange(p, dtype=np.uint8), 3000 // p + 1)[:3000] for p in (1, 2, 3, 5, 7, 8, 9, 63, 64, 65, 255)]
    rng = np.random.default_rng(3)
    pat = rng.integers(0, 256, 40, dtype=np.uint8)
    b += [np.concatenate([pat, rng.integers(0, 256, 60000, dtype=np.uint8), pat, pat])]
    return b


@pytest.mark.parametrize("variant", ["lds_table", "global_table", "lds_table_28_known_bytes"])
def test_encode_fast_matches_oracle(emu, oracle, variant):
    """the three builds of the encoder: table in LDS (chain with the pair fall-back inside), table in memory (plain chain),
    and the LDS-table kernel that knows 28 bytes behind every probe (what small batches run)"""
    blocks = _fixture_blocks()
    blocks += [corpus.class_bytes(name, 65536, 21) for name in ("nci", "samba", "osdb", "xml")]      # matches of 13 .. 28 bytes and beyond
    src, soff, slen = pack(blocks)
    caps = [oracle.compress_bound(b.size) for b in blocks]
    dst, doff, dcap = arena(caps)
    out = emu.encode_batch(src, soff, slen, dst, doff, dcap, gtab=variant == "global_table", more=variant.endswith("bytes"))
    for i, b in enumerate(blocks):
        want = oracle.encode(b)
        if b.size == 0:
            assert out[i] == 0
            continue
        got = dst[int(doff[i]):int(doff[i]) + max(int(out[i]), 0)].tobytes()
        assert out[i] == len(want) and got == want, f"block {i} ({b.size} B)"
    # guard bands and slack untouched
    mask = np.ones(dst.size, bool)
    for i in range(len(blocks)):
        mask[int(doff[i]):int(doff[i]) + max(int(out[i]), 0)] = False
    assert (dst[mask] == 0xCD).all()


@pytest.mark.parametrize("k,waves,how", [(1, 1, ""), (1, 16, "inline+migrate"), (1, 12, "inline+queue"), (1, 11, "migrate"), (2, 16, ""), (2, 9, "inline"), (3, 12, "queue"), (4, 5, ""),
                                         (1, 16, "inline+migrate+slots"), (1, 16, "inline+queue+slots"), (1, 3, "inline+queue+slots"), (1, 16, "inline+queue+migrate+slots"), (1, 11, "inline+queue+migrate+slots")])
def test_encode_parse_emit_matches_oracle(emu, oracle, k, waves, how):
    """the two-kernel fast encoder (k4lz4_parse.hpp: which sequences, then their bytes): k sub-windows of 64 positions per
    round, `waves` blocks per workgroup of which those beyond nine keep their table in memory; blocks it leaves alone
    (under 128 bytes, 65 547 and more) come out of the one-kernel encoder behind it.  "slots": the records in one slot per wave of
    the launch instead of one per block (what the launcher does: the persistent launch of a batch beyond one residency reuses
    them block after block).  The emulator's entry point also checks that nothing was written behind a block's last counted record
    (the record slot is sized by PARSE_REC_STRIDE's static_assert: what is counted is all that is written)."""
    blocks = _fixture_blocks()
    blocks += [corpus.lorem(n) for n in (127, 128, 129, 140, 200, 65546, 65547)] + [corpus.repeated(0xAA, n) for n in (128, 129, 141)]
    blocks += [corpus.class_bytes(name, 65536, 21) for name in ("nci", "samba", "osdb", "xml", "x-ray", "sao")]
    # an incompressible stretch long enough for the step to grow (LL64.fast.cs:156-172), then matches again -- also at the block's end
    rng = np.random.default_rng(8)
    noise = rng.integers(0, 256, 9000, dtype=np.uint8)
    blocks += [np.concatenate([corpus.lorem(700), noise, corpus.lorem(900)]), np.concatenate([noise, noise[:5000]]), noise[:4000].copy()]
    blocks.append(adversarial_blocks.dense_four_byte_matches(128, 65546, 128))      # ~10 500 sequences: a hit on every fourth lane, round after round
    blocks += adversarial_blocks.search_limit_at_block_end()                            # the search's 66th probe on either side of mflimitPlusOne
    src, soff, slen = pack(blocks)
    dst, doff, dcap = arena([oracle.compress_bound(b.size) for b in blocks])
    order = np.random.default_rng(k).permutation(len(blocks)).astype(np.uint32) if waves != 9 else None
    out, nseq = emu.encode_parse_batch(src, soff, slen, dst, doff, dcap, k=k, waves=waves, order=order, inline_emit="inline" in how, queue="queue" in how, migrate="migrate" in how, slot_recs="slots" in how)
    for i, b in enumerate(blocks):
        want = oracle.encode(b)
        if b.size == 0:
            assert out[i] == 0
            continue
        got = dst[int(doff[i]):int(doff[i]) + max(int(out[i]), 0)].tobytes()
        assert out[i] == len(want) and got == want, f"block {i} ({b.size} B)"
        # (65 547 bytes and more: k4_parse_big_kernel where the parsing waves write out themselves, else the one-kernel encoder)
        assert (nseq[i] == 0xFFFFFFFF) == (b.size < 128 or (b.size >= 65547 and "inline" not in how)), f"block {i} ({b.size} B): who encoded it"
        if nseq[i] != 0xFFFFFFFF and b.size < 65547:
            assert nseq[i] == oracle.count_sequences(np.frombuffer(want, np.uint8)) - 1, f"block {i}: sequences"
    mask = np.ones(dst.size, bool)
    for i in range(len(blocks)):
        mask[int(doff[i]):int(doff[i]) + max(int(out[i]), 0)] = False
    assert (dst[mask] == 0xCD).all()


@pytest.mark.parametrize("waves,how,x32", [(16, "inline+migrate+slots", False), (5, "inline+queue+slots", False), (12, "inline+slots", True), (16, "inline+queue+migrate+slots", True)])
def test_encode_parse_big_blocks_matches_oracle(emu, oracle, waves, how, x32):
    """round 6: blocks of 65 547 bytes and more -- byU32 table, hash5 (LL64.fast.cs:526-544), or hash4 with LZ4Codec.Enforce32
    (x32/LL32.tools.cs:141-148) -- through k4_parse_big_kernel: candidates more than 65 535 bytes back are no matches
    (LL64.fast.cs:219-224), match lengths beyond the record's 16 bits are counted again at the write-out, and a block with more
    sequences than its record slot holds is written out a slot-full at a time (the dense block below: ~21 000 sequences against
    16 448 records).  Full and ragged output limits; small blocks in the same batch keep their kernel."""
    rng = np.random.default_rng(12)
    sizes = [65547, 65548, 70000, 100000, 131072, 200000, 262144 + 77, 400000]
    blocks = [corpus.class_bytes(corpus.SILESIA_NAMES[(3 * i) % 12], n, 31 + i) for i, n in enumerate(sizes)]
    blocks.append(np.concatenate([corpus.lorem(3000), corpus.repeated(0x55, 150000), corpus.lorem(2000)]))             # a match of 150 000 bytes: length code beyond 16 bits
    blocks.append(np.concatenate([corpus.class_bytes("xml", 40000, 2), rng.integers(0, 256, 80000, dtype=np.uint8), corpus.class_bytes("xml", 40000, 2)]))   # the same text 120 000 bytes on: too far
    blocks.append(np.concatenate([rng.integers(0, 256, 70000, dtype=np.uint8), corpus.lorem(5000)]))                  # the step grows and grows, then matches again
    import adversarial_blocks
    blocks.append(np.tile(adversarial_blocks.dense_four_byte_matches(128, 65536, 128)[:43000], 2))                      # more sequences than a record slot
    blocks += [corpus.lorem(n) for n in (100, 5000, 65546)] + [corpus.class_bytes("mr", 40000, 3)]                      # the other kernel's
    caps = []
    for b in blocks:
        bound = oracle.compress_bound(b.size)
        caps.append(bound)
    blocks2, caps2 = list(blocks), list(caps)
    for b in blocks[:6]:                                    # the same blocks with output limits: exact fit, one byte short, half
        r, _ = (oracle.compress_fast_x32 if x32 else oracle.compress_fast)(b)
        for c in (r, r - 1, r // 2):
            blocks2.append(b); caps2.append(c)
    src, soff, slen = pack(blocks2)
    dst, doff, dcap = arena(caps2)
    out, nseq = emu.encode_parse_batch(src, soff, slen, dst, doff, dcap, flags=FLAG_RAW | (128 if x32 else 0), k=1, waves=waves, inline_emit=True, queue="queue" in how, migrate="migrate" in how, slot_recs=True, threads=8)
    for i, (b, cap) in enumerate(zip(blocks2, caps2)):
        r, w = (oracle.compress_fast_x32 if x32 else oracle.compress_fast)(b, cap)
        assert out[i] == max(r, 0), (i, b.size, cap, out[i], r)
        if r > 0:
            assert dst[int(doff[i]):int(doff[i]) + r].tobytes() == w[:r].tobytes(), (i, b.size, cap)
            assert (dst[int(doff[i]) + r:int(doff[i]) + cap + 16] == 0xCD).all(), (i, b.size, cap)
        assert (nseq[i] == 0xFFFFFFFF) == (b.size < 128)      # nothing of 128 bytes and more is the one-kernel encoder's


def test_encode_parse_block_ends_randomised(emu, oracle):
    """tests/tools/emu_stress_block_end.py, 4096 blocks: a long literal run with something matchable near the very end of the block --
    where the search's 66-probe limit, its growing step and mflimitPlusOne meet (the round-5 fix; the code before it fails this seed)"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import emu_stress_block_end
    assert emu_stress_block_end.run(4096, 3, oracle, emu) == 0


def test_encode_parse_emit_limited_output_and_acceleration(emu, oracle):
    """output limits are the emit kernel's (LL64.fast.cs:251-255, :346-350, :471-476): cap == size succeeds, one less fails;
    another acceleration than 1 is not the parse kernel's case and comes out of the one-kernel encoder all the same"""
    blocks, caps, wants = [], [], []
    for name in ("x-ray", "dickens", "xml", "sao", "nci"):
        b = corpus.class_bytes(name, 30000, 17)
        n = len(oracle.encode(b))
        for cap in (n, n - 1, n + 1, n // 2, 0, 13):
            blocks.append(b); caps.append(cap)
            r, enc = oracle.compress_fast(b, cap)
            wants.append((r, bytes(enc[:max(r, 0)])))
    src, soff, slen = pack(blocks)
    dst, doff, dcap = arena(caps)
    out, _ = emu.encode_parse_batch(src, soff, slen, dst, doff, dcap, flags=FLAG_RAW, k=1, waves=7, inline_emit=True)
    for i, (r, enc) in enumerate(wants):
        assert out[i] == (r if r > 0 else 0), (i, caps[i])
        if r > 0:
            assert dst[int(doff[i]):int(doff[i]) + r].tobytes() == enc
    dst2, doff2, dcap2 = arena([oracle.compress_bound(b.size) for b in blocks])
    out2, nseq2 = emu.encode_parse_batch(src, soff, slen, dst2, doff2, dcap2, accel=5, flags=FLAG_RAW, k=2, waves=4)
    assert (nseq2 == 0xFFFFFFFF).all()
    for i, b in enumerate(blocks):
        r, enc = oracle.compress_fast(b, oracle.compress_bound(b.size), 5)
        assert out2[i] == r and dst2[int(doff2[i]):int(doff2[i]) + r].tobytes() == bytes(enc[:r])


def test_encode_fast_big_block_hash5(emu, oracle):
    """>= 65547 bytes switches to the u32 table + hash5 (LL64.fast.cs:526-544)"""
    blocks = [corpus.lorem(65547), corpus.class_bytes("samba", 200000, 5), corpus.class_bytes("mozilla", 140000, 5),
              np.concatenate([corpus.class_bytes("dickens", 70000, 9)] * 2),
              # matches far longer than a pending-sequence record can say (12 bits): counted again when written out
              np.concatenate([corpus.random_bytes(30000, 4)] * 3), corpus.repeated(7, 100000)]
    src, soff, slen = pack(blocks)
    dst, doff, dcap = arena([oracle.compress_bound(b.size) for b in blocks])
    out = emu.encode_batch(src, soff, slen, dst, doff, dcap)
    for i, b in enumerate(blocks):
        want = oracle.encode(b)
        assert out[i] == len(want)
        assert dst[int(doff[i]):int(doff[i]) + int(out[i])].tobytes() == want


def test_encode_limited_output_boundary(emu, oracle):
    """BlockRoundtripTests.cs:114-125 BorderLineCompressions: cap == size succeeds, size-1 fails"""
    blocks, caps, wants = [], [], []
    for name in ("x-ray", "dickens", "xml", "sao"):
        b = corpus.class_bytes(name, 30000, 2)
        full = oracle.encode(b)
        for cap in (len(full), len(full) - 1, len(full) + 1, 10, 0):
            blocks.append(b); caps.append(cap); wants.append(full if cap >= len(full) else None)
    src, soff, slen = pack(blocks)
    dst, doff, dcap = arena(caps)
    out = emu.encode_batch(src, soff, slen, dst, doff, dcap)
    raw = emu.encode_batch(src, soff, slen, arena(caps)[0], doff, dcap, flags=FLAG_RAW)
    for i, w in enumerate(wants):
        if w is None:
            assert out[i] == -1 and raw[i] == 0
        else:
            assert out[i] == len(w) and dst[int(doff[i]):int(doff[i]) + len(w)].tobytes() == w
    for i in range(len(blocks)):   # never writes past its slot
        lo, hi = int(doff[i]) + int(dcap[i]), int(doff[i]) + int(dcap[i]) + 16
        assert (dst[lo:hi] == 0xCD).all()


def test_decode_matches_oracle_and_guards(emu, oracle):
    blocks = [b for b in _fixture_blocks() if b.size]
    comp = [np.frombuffer(oracle.encode(b), np.uint8) for b in blocks]
    src, soff, slen = pack(comp)
    for slack in (0, 37):
        dst, doff, dcap = arena([b.size + slack for b in blocks])
        out = emu.decode_batch(src, soff, slen, dst, doff, dcap)
        for i, b in enumerate(blocks):
            assert out[i] == b.size
            assert dst[int(doff[i]):int(doff[i]) + b.size].tobytes() == b.tobytes()
            assert (dst[int(doff[i]) + b.size:int(doff[i]) + b.size + slack + 16] == 0xCD).all()


def test_decode_golden_issue64(emu, oracle):
    import os, struct
    raw = open(os.path.join(os.path.dirname(__file__), "golden", "issue64_input.bin"), "rb").read()
    want = open(os.path.join(os.path.dirname(__file__), "golden", "issue64_output.bin"), "rb").read()
    u, c = struct.unpack_from("<II", raw, 24)
    comp = np.frombuffer(raw[32:32 + c], np.uint8)
    src, soff, slen = pack([comp])
    dst, doff, dcap = arena([u])
    out = emu.decode_batch(src, soff, slen, dst, doff, dcap)
    assert out[0] == 65536 and dst[int(doff[0]):int(doff[0]) + u].tobytes() == want[:u]


def test_decode_malformed_parity_with_oracle(emu, oracle):
    """accept/reject, LLxx-level error codes and bytes identical to the oracle (LL64.dec.cs rules)"""
    rng = np.random.default_rng(17)
    comps, caps = [], []
    for name, n in (("dickens", 3000), ("xml", 5000), ("x-ray", 2000), ("nci", 70000)):
        data = corpus.class_bytes(name, n, 4)
        good = np.frombuffer(oracle.encode(data), np.uint8)
        for t in range(150):
            bad = good.copy()
            kind = t % 4
            if kind == 0:
                bad = bad[:rng.integers(1, good.size)]
            elif kind == 1:
                for _ in range(int(rng.integers(1, 4))):
                    bad[rng.integers(0, good.size)] = rng.integers(0, 256)
            elif kind == 2:
                bad = np.concatenate([bad, rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8)])
            comps.append(bad)
            caps.append(n + int(rng.integers(-20, 21)) if kind != 3 else int(rng.integers(0, n)))
    src, soff, slen = pack(comps)
    dst, doff, dcap = arena(caps)
    out = emu.decode_batch(src, soff, slen, dst, doff, dcap, flags=FLAG_RAW)
    for i, (c, cap) in enumerate(zip(comps, caps)):
        n, ref = oracle.decompress_safe(c, cap)
        assert out[i] == n, f"stream {i}: kernel {out[i]} oracle {n}"
        if n >= 0:
            assert dst[int(doff[i]):int(doff[i]) + n].tobytes() == ref[:n].tobytes()
        assert (dst[int(doff[i]) + cap:int(doff[i]) + cap + 16] == 0xCD).all()
        assert (dst[int(doff[i]) - 16:int(doff[i])] == 0xCD).all()


from stream_cases import long_field_streams as _long_field_streams


def test_long_length_fields_a_wave_full_at_a_time(emu, oracle):
    """the scalar parser reads length fields 64 bytes per step (read_vle): same return values -- the error position included --
    and bytes as the oracle's byte-at-a-time loop on long literal and match runs, cut and damaged inside the fields"""
    rng = np.random.default_rng(77)
    cases = _long_field_streams(oracle, rng)
    comps, caps = [c for c, _, _ in cases], [k for _, k, _ in cases]
    src, soff, slen = pack(comps)
    dst, doff, dcap = arena(caps)
    out = emu.decode_batch(src, soff, slen, dst, doff, dcap, flags=FLAG_RAW)
    for i, (c, cap, defined) in enumerate(cases):
        n, ref = oracle.decompress_safe(c, cap)
        assert out[i] == n, f"stream {i} ({c.size} B, cap {cap}): kernel {out[i]} oracle {n}"
        if n >= 0 and defined:
            assert dst[int(doff[i]):int(doff[i]) + n].tobytes() == ref[:n].tobytes()
        assert (dst[int(doff[i]) + cap:int(doff[i]) + cap + 16] == 0xCD).all()


def test_decode_special_cases(emu, oracle):
    """LL64.dec.cs:160-172 and the LZ4Codec mapping (LZ4Codec.cs:104-115)"""
    comps = [np.array([0], np.uint8), np.array([0], np.uint8), np.array([0x10, 0x41], np.uint8), np.zeros(0, np.uint8),
             np.array([0x00, 0x00], np.uint8)]
    caps = [0, 5, 0, 10, 0]
    src, soff, slen = pack(comps)
    dst, doff, dcap = arena(caps)
    raw = emu.decode_batch(src, soff, slen, dst, doff, dcap, flags=FLAG_RAW)
    cod = emu.decode_batch(src, soff, slen, dst, doff, dcap)
    for i, (c, cap) in enumerate(zip(comps, caps)):
        n, _ = oracle.decompress_safe(c, cap)
        assert raw[i] == n
        want = oracle.decode(c, cap)
        assert cod[i] == (0 if c.size == 0 else (-1 if want is None else len(want)))


def test_pickle_matches_oracle(emu, oracle):
    """PicklingTests.cs:11-50 lengths, both header rules"""
    blocks = [np.zeros(0, np.uint8)]
    for n in (1, 10, 32, 200, 1023, 1024, 1025, 1337, 0x10000, 0x172a5):
        blocks += [corpus.lorem(n), corpus.random_bytes(n, n)]
    blocks += [corpus.class_bytes("xml", 300000, 1), corpus.repeated(7, 70000)]
    src, soff, slen = pack(blocks)
    caps = [oracle.lib.k4o_pickle_bound(b.size) for b in blocks]
    for writer in (0, 1):
        dst, doff, dcap = arena(caps)
        out = emu.pickle_batch(src, soff, slen, dst, doff, dcap, flags=FLAG_WRITER if writer else 0)
        pickles = []
        for i, b in enumerate(blocks):
            want = oracle.pickle(b, 0, writer)
            got = dst[int(doff[i]):int(doff[i]) + int(out[i])].tobytes()
            assert got == want, f"message {i} ({b.size} B) writer={writer}"
            assert (dst[int(doff[i]) + max(int(dcap[i]), int(out[i])):int(doff[i]) + int(dcap[i]) + 16] == 0xCD).all()
            pickles.append(np.frombuffer(got, np.uint8))
        # unpickle through the kernels
        psrc, poff, plen = pack(pickles)
        sizes = emu.unpickle_sizes(psrc, poff, plen)
        assert list(sizes) == [b.size for b in blocks]
        udst, uoff, ucap = arena(sizes)
        uout = emu.unpickle_batch(psrc, poff, plen, udst, uoff, ucap)
        for i, b in enumerate(blocks):
            assert uout[i] == b.size
            assert udst[int(uoff[i]):int(uoff[i]) + b.size].tobytes() == b.tobytes()


TWO_STEP = 1 << 30          # the emulator's flag: the segments' runs by the two-step encoder (k4_parse_kernel + k4_parse_seg_kernel, round 6)


@pytest.mark.parametrize("engine", [0, TWO_STEP, TWO_STEP | (5 << 24)], ids=["one_kernel", "two_step", "two_step_5_waves"])
def test_pickle_in_segments_matches_oracle(emu, oracle, engine):
    """k4lz4_segments.hpp: big messages cut into segments, every piece by a wave of its own, are byte for byte the oracle's
    pickles -- where a boundary verifies (the pieces are joined) and where it does not (the message is encoded again).  Small
    segment sizes so that the emulator gets through it; both outcomes must occur over the set."""
    blocks = [corpus.lorem(1000), corpus.class_bytes("dickens", 120000, 1), corpus.class_bytes("xml", 150000, 1), corpus.random_bytes(90000, 5),
              corpus.repeated(7, 100000), corpus.class_bytes("webster", 90000, 3),
              np.concatenate([corpus.class_bytes("nci", 60000, 1), corpus.random_bytes(40000, 9), corpus.class_bytes("nci", 50000, 1)]),
              corpus.lorem(70000)]
    src, soff, slen = pack(blocks)
    caps = [oracle.lib.k4o_pickle_bound(b.size) for b in blocks]
    joined = cut = 0
    for (seg_target, seg_warm) in ((32768, 65536), (40000, 8192)):
        dst, doff, dcap = arena(caps)
        out, stats = emu.pickle_seg_batch(src, soff, slen, dst, doff, dcap, 70000, seg_target, seg_warm, flags=engine)
        for i, b in enumerate(blocks):
            want = oracle.pickle(b, 0, 0)
            got = dst[int(doff[i]):int(doff[i]) + int(out[i])].tobytes()
            assert got == want, f"message {i} ({b.size} B), segments of {seg_target} B, warm-up {seg_warm} B"
            assert (dst[int(doff[i]) + max(int(dcap[i]), int(out[i])):int(doff[i]) + int(dcap[i]) + 16] == 0xCD).all()
        assert stats[0] >= 5 and stats[1] >= 2 * stats[0]
        cut += int(stats[0]); joined += int(stats[2])
    assert 0 < joined < cut, (joined, cut)


@pytest.mark.parametrize("engine", [0, TWO_STEP], ids=["one_kernel", "two_step"])
def test_segments_behind_a_bad_boundary_are_kept(emu, oracle, engine):
    """round 4: a piece whose successor is not in step keeps its output and leaves the cut's table in its item's slot (state 4;
    the successor's published snapshot stays what it was: overwriting it once made a later check pass against the wrong table),
    and the join's one wave goes on from there only as far as the next boundary whose piece IS in step.  Messages whose middle
    part breaks the warm-up's assumption (random bytes between text) at several segment sizes: state-4 pieces and resumed
    runs must occur, one of the settings has a resumed run stop at a verified boundary, every envelope is the oracle's."""
    parts = lambda *p: np.concatenate(p)
    blocks = [parts(corpus.class_bytes("dickens", 100000, 1), corpus.random_bytes(60000, 1), corpus.class_bytes("dickens", 200000, 2)),
              parts(corpus.class_bytes("xml", 90000, 3), corpus.random_bytes(50000, 2), corpus.class_bytes("webster", 160000, 4),
                    corpus.random_bytes(30000, 5), corpus.class_bytes("webster", 100000, 8)),
              parts(corpus.lorem(40000), corpus.class_bytes("nci", 50000, 5), corpus.lorem(70000))]
    src, soff, slen = pack(blocks)
    caps = [oracle.lib.k4o_pickle_bound(b.size) for b in blocks]
    kept = resumed = stops = 0
    for (seg_target, seg_warm) in ((32768, 65536), (49152, 70000), (16384, 4096), (24576, 66000)):
        dst, doff, dcap = arena(caps)
        out, stats = emu.pickle_seg_batch(src, soff, slen, dst, doff, dcap, 70000, seg_target, seg_warm, flags=engine)
        for i, b in enumerate(blocks):
            assert dst[int(doff[i]):int(doff[i]) + int(out[i])].tobytes() == oracle.pickle(b, 0, 0), (i, seg_target, seg_warm)
            assert (dst[int(doff[i]) + max(int(dcap[i]), int(out[i])):int(doff[i]) + int(dcap[i]) + 16] == 0xCD).all()
        kept += int(stats[3]); resumed += int(stats[4]); stops += int(stats[5])
    assert kept > 0 and 0 < stops <= resumed, (kept, resumed, stops)


def test_unpickle_corruption(emu, oracle):
    """PicklingTests.cs:149-172: corrupted pickles are rejected exactly where the oracle rejects"""
    rng = np.random.default_rng(23)
    base = np.frombuffer(oracle.pickle(corpus.lorem(5000)), np.uint8)
    pickles = []
    for t in range(200):
        p = base.copy()
        k = t % 4
        if k == 0:
            p[0] = rng.integers(0, 256)
        elif k == 1:
            p = p[:rng.integers(0, p.size)]
        elif k == 2:
            p[rng.integers(0, p.size)] ^= 1 << rng.integers(0, 8)
        else:
            p = np.concatenate([p, rng.integers(0, 256, 3, dtype=np.uint8)])
        pickles.append(p)
    psrc, poff, plen = pack(pickles)
    sizes = emu.unpickle_sizes(psrc, poff, plen)
    caps = np.where(sizes < 0, 0, np.minimum(sizes, 1 << 20)).astype(np.int32)
    udst, uoff, ucap = arena(caps)
    uout = emu.unpickle_batch(psrc, poff, plen, udst, uoff, ucap)
    for i, p in enumerate(pickles):
        want = oracle.unpickle(p.tobytes())
        rc, _, rl, _ = oracle.unpickle_header(p.tobytes()) if p.size else (0, 0, 0, 0)
        if p.size == 0:
            assert sizes[i] == 0 and uout[i] == 0
            continue
        assert sizes[i] == (rl if rc == 0 and rl >= 0 else -1)
        if want is None or caps[i] != sizes[i]:
            assert uout[i] == -1
        else:
            assert uout[i] == len(want) and udst[int(uoff[i]):int(uoff[i]) + len(want)].tobytes() == want


def test_decode_all_classes_various_sizes(emu, oracle):
    """speculative parse rounds + scalar fallbacks on every data class, 4 KiB .. 64 KiB blocks,
    odd source alignments (the packed buffer puts blocks at arbitrary byte offsets)"""
    rng = np.random.default_rng(41)
    blocks = []
    for i, name in enumerate(corpus.SILESIA_NAMES):
        data = corpus.class_bytes(name, 200000, 13)
        for size in (65536, 4096, 4097, int(rng.integers(100, 30000))):
            start = int(rng.integers(0, data.size - size))
            blocks.append(data[start:start + size])
    comp = [np.frombuffer(oracle.encode(b), np.uint8) for b in blocks]
    src, soff, slen = pack(comp)
    dst, doff, dcap = arena([b.size for b in blocks])
    out = emu.decode_batch(src, soff, slen, dst, doff, dcap)
    for i, b in enumerate(blocks):
        assert out[i] == b.size, i
        assert dst[int(doff[i]):int(doff[i]) + b.size].tobytes() == b.tobytes(), i
    mask = np.ones(dst.size, bool)
    for i, b in enumerate(blocks):
        mask[int(doff[i]):int(doff[i]) + b.size] = False
    assert (dst[mask] == 0xCD).all()


def test_decode_hostile_streams_random(emu, oracle):
    """random byte soup and heavy mutations: identical LLxx returns, no write outside the slot"""
    rng = np.random.default_rng(43)
    comps, caps = [], []
    good = np.frombuffer(oracle.encode(corpus.class_bytes("xml", 20000, 1)), np.uint8)
    for t in range(300):
        if t % 3 == 0:
            c = rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8)
        else:
            c = good[:int(rng.integers(20, good.size))].copy()
            for _ in range(int(rng.integers(1, 30))):
                c[rng.integers(0, c.size)] = rng.integers(0, 256)
        comps.append(c)
        caps.append(int(rng.integers(0, 25000)))
    src, soff, slen = pack(comps)
    dst, doff, dcap = arena(caps)
    out = emu.decode_batch(src, soff, slen, dst, doff, dcap, flags=FLAG_RAW)
    for i, (c, cap) in enumerate(zip(comps, caps)):
        n, ref = oracle.decompress_safe(c, cap)
        assert out[i] == n, f"stream {i}: kernel {out[i]} oracle {n}"
        if n > 0:
            assert dst[int(doff[i]):int(doff[i]) + n].tobytes() == ref[:n].tobytes()
        assert (dst[int(doff[i]) + cap:int(doff[i]) + cap + 16] == 0xCD).all()
        assert (dst[int(doff[i]) - 16:int(doff[i])] == 0xCD).all()


def test_dispatch_order_is_a_permutation_by_cost(emu):
    """the scheduling kernels: a dry encoder run over a sample estimates cost (never touches dst --
    it is given a null output), blocks are ordered most-expensive-bucket first"""
    blocks = [corpus.class_bytes(name, 20000, 3) for name in corpus.SILESIA_NAMES]
    blocks += [corpus.random_bytes(20000, 5), corpus.lorem(50), np.zeros(0, np.uint8), corpus.lorem(300000)]
    src, soff, slen = pack(blocks)
    cost, order = emu.order(src, soff, slen)
    assert sorted(order.tolist()) == list(range(len(blocks)))
    assert (np.diff(cost[order].astype(np.int64)) <= 0).all()      # non-increasing cost buckets
    names = list(corpus.SILESIA_NAMES)
    # text-like classes (many short sequences) rank above incompressible ones of the same size
    assert cost[names.index("dickens")] > cost[names.index("x-ray")]
    assert cost[len(blocks) - 1] == cost.max()                       # the 300 KB block is the most expensive
    cost_l, order_l = emu.order(src, soff, slen, by_length=1)
    assert sorted(order_l.tolist()) == list(range(len(blocks)))
    assert (np.diff(slen[order_l].astype(np.int64) // 1) <= 0).all() or (np.diff(cost_l[order_l].astype(np.int64)) <= 0).all()


@pytest.mark.parametrize("level", [3, 4, 6, 8, 9, 10, 11, 12])
def test_encode_hc_matches_oracle(emu, oracle, level):
    """HC chain + parse kernels (levels 3..9; 9 adds pattern analysis) and the optimal parser (10..12) against the
    oracle's LL64.high.cs restatement"""
    blocks = [np.frombuffer(corpus.QUICK_FOX, np.uint8), np.zeros(0, np.uint8)]
    if level >= 9:      # runs of 1/2/4-byte patterns of many lengths, next to each other and far apart
        rng = np.random.default_rng(9)
        for unit in (b"a", b"ab", b"abcd", b"aaab", b"abc"):
            parts = []
            for _ in range(60):
                parts.append(np.frombuffer(unit * int(rng.integers(1, 400)), np.uint8))
                parts.append(rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8))
            blocks.append(np.concatenate(parts))
        blocks.append(np.concatenate([np.zeros(70000, np.uint8), np.frombuffer(b"xyz", np.uint8), np.zeros(70000, np.uint8)]))
    blocks += [corpus.lorem(n) for n in (1, 12, 13, 14, 1000, 65536)]
    blocks += [corpus.repeated(0xAA, n) for n in (13, 33, 1000, 70000)]
    blocks += [corpus.random_bytes(5000, 5), np.tile(np.frombuffer(b"abcdabcdabcdabcd" * 4 + b"xyz", np.uint8), 300)]
    blocks += [np.concatenate([np.zeros(20000, np.uint8), corpus.random_bytes(100, 1), np.zeros(20000, np.uint8)])]
    blocks += [corpus.class_bytes(name, 30000 + 5000 * (i % 3), 7) for i, name in enumerate(corpus.SILESIA_NAMES)]
    blocks += [corpus.class_bytes("samba", 150000, 3)]          # > 64 KiB: lowestMatchIndex slides
    if level >= 10:     # the optimal parser is slow under the emulator: a smaller set (the GPU tests run the full one)
        blocks = [b[:12000] for b in blocks[:8]] + [b[:6000] for b in blocks[-18:-12]] + [b[:9000] for b in blocks[-9:-1:2]] + [blocks[-1][60000:90000]]
    src, soff, slen = pack(blocks)
    caps = [oracle.compress_bound(b.size) for b in blocks]
    dst, doff, dcap = arena(caps)
    out = emu.encode_hc_batch(src, soff, slen, dst, doff, dcap, level=level)
    for i, b in enumerate(blocks):
        if b.size == 0:
            assert out[i] == 0
            continue
        r, w = oracle.compress_hc(b, level)
        assert out[i] == r, f"block {i} ({b.size} B) level {level}"
        assert dst[int(doff[i]):int(doff[i]) + r].tobytes() == w[:r].tobytes(), f"block {i} level {level}"
    mask = np.ones(dst.size, bool)
    for i in range(len(blocks)):
        mask[int(doff[i]):int(doff[i]) + max(int(out[i]), 0)] = False
    assert (dst[mask] == 0xCD).all()


@pytest.mark.parametrize("nseg_log2", [0, 1, 2])
def test_encode_hc_level3_from_sequence_records(emu, oracle, nseg_log2):
    """round 6: at level 3, on blocks of at most 64 KiB, k4_hc_parse_kernel only decides -- 8-byte sequence records -- and the bytes are
    written from the records afterwards (emit_block<true>: LZ4HC_encodeSequence's format and ITS output-limit tests,
    LL64.high.cs:435-510).  The oracle's bytes at full capacity, at exact fit, one byte short and far too small.
    nseg_log2 1 / 2: the same blocks parsed by two / four waves each (HcSegs: every wave from its own start, joined where their cursors
    meet) -- byte for byte the one-wave parse."""
    rng = np.random.default_rng(44)
    blocks = [corpus.class_bytes(name, int(rng.integers(2000, 65537)), 7) for name in corpus.SILESIA_NAMES]
    blocks += [corpus.class_bytes(name, 65536, 9) for name in ("dickens", "xml", "nci", "sao")]
    blocks += [corpus.lorem(n) for n in (1, 12, 13, 14, 1000, 65536)] + [corpus.repeated(0xAA, n) for n in (13, 33, 1000, 65536)]
    blocks += [corpus.random_bytes(5000, 5), np.tile(np.frombuffer(b"abcdabcdabcdabcd" * 4 + b"xyz", np.uint8), 300)]
    blocks += [np.concatenate([np.zeros(20000, np.uint8), corpus.random_bytes(100, 1), np.zeros(20000, np.uint8)])]
    import adversarial_blocks
    blocks.append(adversarial_blocks.dense_four_byte_matches(128, 65536, 128))
    full = [oracle.compress_hc(b, 3) for b in blocks]
    cases = []
    for b, (r, w) in zip(blocks, full):
        for cap in (oracle.compress_bound(b.size), r, r - 1, max(r // 2, 0), 0):
            cases.append((b, cap))
    src, soff, slen = pack([b for b, _ in cases])
    dst, doff, dcap = arena([c for _, c in cases])
    # (nseg_log2 0 also runs the candidate records from memory -- emulator flag bit 26 -- and the chains by one wave per block -- bit 29)
    out = emu.encode_hc_batch(src, soff, slen, dst, doff, dcap, level=3, flags=FLAG_RAW | (1 << 30) | (nseg_log2 << 27) | ((1 << 26) | (1 << 29) if nseg_log2 == 0 else 0))
    for i, (b, cap) in enumerate(cases):
        r, w = oracle.compress_hc(b, 3, cap=cap)
        assert out[i] == r, (i, b.size, cap, out[i], r)
        if r > 0:
            assert dst[int(doff[i]):int(doff[i]) + r].tobytes() == w[:r].tobytes(), (i, b.size, cap)
    mask = np.ones(dst.size, bool)                         # nothing outside a slot, nothing behind a successful block's bytes
    for i, (b, cap) in enumerate(cases):                   # (what a failed call leaves INSIDE its slot is no contract)
        mask[int(doff[i]):int(doff[i]) + (int(out[i]) if out[i] > 0 else cap)] = False
    assert (dst[mask] == 0xCD).all()


def test_encode_hc_level3_several_waves_per_block_stress():
    """tests/tools/emu_stress_hc_segs.py, six cases (144 blocks): blocks of 8 192 .. 65 536 bytes stitched from random / repeated / periodic / copied /
    corpus-class parts (matches and runs across the waves' starts, stretches without a match), two and four waves per block, ragged
    output limits -- the oracle's bytes."""
    import subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "emu_stress_hc_segs.py")
    out = subprocess.run([sys.executable, tool, "6", "7"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().startswith("ok 6 cases"), (out.stdout[-500:], out.stderr[-1500:])


def test_encode_hc_limited_output(emu, oracle):
    blocks, caps, wants = [], [], []
    for name in ("dickens", "xml", "x-ray"):
        b = corpus.class_bytes(name, 20000, 2)
        r, w = oracle.compress_hc(b, 3)
        for cap in (r, r - 1, r + 1, 10, 0):
            blocks.append(b); caps.append(cap); wants.append(w[:r].tobytes() if cap >= r else None)
    src, soff, slen = pack(blocks)
    dst, doff, dcap = arena(caps)
    out = emu.encode_hc_batch(src, soff, slen, dst, doff, dcap, level=3)
    for i, w in enumerate(wants):
        if w is None:
            assert out[i] == -1
        else:
            assert out[i] == len(w) and dst[int(doff[i]):int(doff[i]) + len(w)].tobytes() == w
        assert (dst[int(doff[i]) + int(dcap[i]):int(doff[i]) + int(dcap[i]) + 16] == 0xCD).all()


def test_partial_decode_matches_oracle(emu, oracle):
    """LZ4_decompress_safe_partial (LL64.dec.cs:548-556; PartialDecompressionTests.cs:9-46):
    exactly `target` bytes, the rest untouched; also on truncated / corrupt streams"""
    rng = np.random.default_rng(53)
    comps, targets = [], []
    for name, n in (("dickens", 3000), ("xml", 20000), ("mr", 66000), ("x-ray", 5000)):
        data = corpus.class_bytes(name, n, 8)
        good = np.frombuffer(oracle.encode(data), np.uint8)
        for t in (0, 1, 5, 17, 100, n // 3, n - 13, n - 12, n - 5, n - 1, n, n + 50):
            comps.append(good); targets.append(t)
        for t in range(40):
            bad = good.copy()
            if t % 2:
                bad = bad[:rng.integers(1, good.size)]
            else:
                bad[rng.integers(0, good.size)] = rng.integers(0, 256)
            comps.append(bad); targets.append(int(rng.integers(0, n + 20)))
    src, soff, slen = pack(comps)
    dst, doff, dcap = arena(targets)
    out = emu.decode_batch(src, soff, slen, dst, doff, dcap, flags=FLAG_RAW | 32)
    for i, (c, t) in enumerate(zip(comps, targets)):
        n, ref = oracle.decompress_partial(c, t, t)
        assert out[i] == n, f"stream {i}: kernel {out[i]} oracle {n} target {t}"
        if n > 0:
            assert dst[int(doff[i]):int(doff[i]) + n].tobytes() == ref[:n].tobytes()
        assert (dst[int(doff[i]) + t:int(doff[i]) + t + 16] == 0xCD).all()


def _dict_streams(oracle, syslz4_lib=None):
    """blocks compressed against a dictionary: made by compressing dict+data as one block with the
    oracle and cutting the stream is not possible, so use chained data: second half of a text
    compressed with LZ4_compress_fast_continue semantics emulated by liblz4's usingDict decode
    inverse -- simpler: take issue64's real dictionary-chained record, plus synthetic streams whose
    offsets reach before the block start."""
    import os, struct
    g = os.path.join(os.path.dirname(__file__), "golden")
    raw = open(os.path.join(g, "issue64_input.bin"), "rb").read()
    want = open(os.path.join(g, "issue64_output.bin"), "rb").read()
    pos, recs = 20, []
    while raw[pos:pos + 4] == b"bv41":
        u, c = struct.unpack_from("<II", raw, pos + 4)
        recs.append((u, np.frombuffer(raw[pos + 12:pos + 12 + c], np.uint8)))
        pos += 12 + c
    return recs, np.frombuffer(want, np.uint8)


def test_decode_with_dictionary_issue64_and_synthetic(emu, oracle):
    """LZ4Codec.Decode(source, target, dictionary) (LZ4Codec.cs:144-160; LL64.dec.cs:523-546):
    the reference repo's chained fixture (record 1 needs record 0's output as dictionary) in external
    and prefix placement, plus hand-made streams with offsets reaching before the block start."""
    recs, want = _dict_streams(oracle)
    d0 = want[:recs[0][0]].copy()
    u1, c1 = recs[1]
    # hand-made: token with 4 literals + match(offset into dict), etc.
    def seq(lits, offset, mlen, last=b"ABCDE"):
        out = bytearray()
        ml = mlen - 4
        tok = (min(len(lits), 15) << 4) | min(ml, 15)
        out.append(tok)
        if len(lits) >= 15:
            r = len(lits) - 15
            while r >= 255: out.append(255); r -= 255
            out.append(r)
        out += lits
        out += bytes([offset & 255, offset >> 8])
        if ml >= 15:
            r = ml - 15
            while r >= 255: out.append(255); r -= 255
            out.append(r)
        out.append(len(last) << 4)
        out += last
        return np.frombuffer(bytes(out) + b"", np.uint8)
    dct = corpus.lorem(3000)
    cases = [(c1, u1, d0), (c1, u1 + 10, d0), (c1, u1 - 1, d0), (c1, u1, d0[-100:]), (c1, u1, d0[:0])]
    for lits, off, ml in ((b"wxyz", 100, 20), (b"", 3000, 8), (b"q", 3001, 40), (b"abcdefgh", 9, 300), (b"ab", 3002, 5),
                          (b"", 2, 50), (b"zzzzzzzzzzzzzzzzzzzz", 24, 4), (b"k", 60000, 10)):
        cases.append((seq(lits, off, ml), len(lits) + ml + 5, dct))
        cases.append((seq(lits, off, ml), len(lits) + ml + 5 + 40, dct))
    from oracle_lib import SystemLZ4
    sysl = SystemLZ4()
    if sysl.available:                      # real dictionary streams (LZ4_loadDict + LZ4_compress_fast_continue)
        for cls, dn, n in (("dickens", 30000, 20000), ("xml", 70000, 30000), ("osdb", 500, 9000), ("nci", 65536, 65536)):
            text = corpus.class_bytes(cls, dn + n, 4)
            dd, data = text[:dn].copy(), text[dn:].copy()
            c = sysl.compress_with_dict(data, dd)
            cases += [(c, n, dd), (c, n - 1, dd), (c, n + 7, dd), (c, n, dd[1:])]
    # external placement: packed dictionaries in their own buffer
    comps = [c for c, _, _ in cases]
    src, soff, slen = pack(comps)
    dpk, doffs, dlens = pack([d for _, _, d in cases])
    dst, doff, dcap = arena([cap for _, cap, _ in cases])
    out = emu.decode_dict_batch(src, soff, slen, dst, doff, dcap, dpk, doffs, dlens, flags=FLAG_RAW)
    for i, (c, cap, d) in enumerate(cases):
        n, ref = oracle.decompress_using_dict(c, cap, d) if d.size else oracle.decompress_safe(c, cap)
        assert out[i] == n, f"ext case {i}: kernel {out[i]} oracle {n}"
        if n > 0:
            assert dst[int(doff[i]):int(doff[i]) + n].tobytes() == ref[:n].tobytes(), i
        assert (dst[int(doff[i]) + cap:int(doff[i]) + cap + 16] == 0xCD).all()
    assert out[0] == u1 and dst[int(doff[0]):int(doff[0]) + u1].tobytes() == want[recs[0][0]:recs[0][0] + u1].tobytes()
    # prefix placement: dictionary immediately before each output slot
    for i, (c, cap, d) in enumerate(cases):
        if d.size == 0:
            continue
        buf = np.full(d.size + cap + 16, 0xCD, np.uint8)
        buf[:d.size] = d
        o = emu.decode_dict_batch(np.ascontiguousarray(c), np.zeros(1, np.uint64), np.array([c.size], np.int32), buf,
                                  np.array([d.size], np.uint64), np.array([cap], np.int32), buf, np.zeros(1, np.uint64),
                                  np.array([d.size], np.int32), flags=FLAG_RAW)
        n, ref = oracle.decompress_using_prefix_dict(c, cap, d)
        assert o[0] == n, f"prefix case {i}: kernel {o[0]} oracle {n}"
        if n > 0:
            assert buf[d.size:d.size + n].tobytes() == ref[:n].tobytes(), i
        assert (buf[d.size + cap:] == 0xCD).all() and buf[:d.size].tobytes() == d.tobytes()


def test_encode_random_stress_28_known_bytes(emu, oracle):
    """the same stress through the small-batch variant of the LDS-table kernel (k4_encode_fast_more_kernel)"""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "emu_stress_encode.py")
    spec = importlib.util.spec_from_file_location("emu_stress_encode", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    emu.more = True
    try:
        mod.run(2, 9, oracle, emu, verbose=False)
    finally:
        emu.more = False


def test_encode_random_stress(emu, oracle):
    """tests/tools/emu_stress_encode.py (two rounds of it): inputs built to provoke equal hashes inside one
    64-position window, matches ending at window edges, long literal runs, limited output, accel > 1"""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "emu_stress_encode.py")
    spec = importlib.util.spec_from_file_location("emu_stress_encode", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.run(2, 5, oracle, emu, verbose=False)


def test_decode_random_stress(emu, oracle):
    """tests/tools/emu_stress_encode.py run_decode (three rounds): exact / oversized / undersized destinations, streams
    with a flipped byte, guard bytes behind every destination"""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "emu_stress_encode.py")
    spec = importlib.util.spec_from_file_location("emu_stress_encode", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.run_decode(3, 7, oracle, emu, verbose=False)


def test_encode_x32_arm_matches_oracle(emu, oracle):
    """LZ4Codec.Enforce32 (K4LZ4_FLAG_X32 = 128): inputs of 64 KiB and more are hashed with LZ4_hash4 (12 bits) into
    the byU32 table (x32/LL32.tools.cs:141-148, x32/LL32.fast.cs:543-545); smaller inputs are unaffected"""
    blocks = [corpus.class_bytes("dickens", 65547, 1), corpus.class_bytes("xml", 100000, 2), corpus.lorem(200000),
              corpus.class_bytes("samba", 65546, 3), corpus.lorem(5000), corpus.random_bytes(70000, 4)]
    src, soff, slen = pack(blocks)
    caps = [oracle.compress_bound(b.size) for b in blocks]
    dst, doff, dcap = arena(caps)
    out = emu.encode_batch(src, soff, slen, dst, doff, dcap, flags=128)
    differs = 0
    for i, b in enumerate(blocks):
        r, w = oracle.compress_fast_x32(b)
        assert out[i] == r and dst[int(doff[i]):int(doff[i]) + r].tobytes() == w[:r].tobytes(), i
        n, back = oracle.decompress_safe(w[:r], b.size)
        assert n == b.size and back[:n].tobytes() == b.tobytes()
        differs += w[:r].tobytes() != oracle.encode(b)
        if b.size < 65547:
            assert w[:r].tobytes() == oracle.encode(b)
    assert differs >= 3            # the large compressible blocks really take the other hash


@pytest.fixture()
def emu_pair(emu):
    """the same emulator driver with decode routed to k4_decode_pair_kernel (parse wave + copy wave per block)"""
    emu.pair = True
    yield emu
    emu.pair = False


@pytest.mark.parametrize("case", [test_decode_matches_oracle_and_guards, test_decode_golden_issue64,
                                  test_decode_malformed_parity_with_oracle, test_decode_special_cases,
                                  test_decode_all_classes_various_sizes, test_decode_hostile_streams_random,
                                  test_partial_decode_matches_oracle, test_decode_with_dictionary_issue64_and_synthetic,
                                  test_decode_random_stress, test_pickle_matches_oracle, test_unpickle_corruption],
                         ids=lambda f: f.__name__)
def test_pair_kernel_passes_the_decoder_tests(case, emu_pair, oracle):
    """every decoder test above, run once more through the two-waves-per-block kernel"""
    case(emu_pair, oracle)


@pytest.mark.parametrize("mode", [0, 1], ids=["lane_copy32", "lane_move32_slack"])
def test_lane_run_copies_every_length_and_alignment(emu, mode):
    """k4lz4_common.hpp LaneRun / lane_move32_slack: chunks plus an overlapping last word, no byte tails -- every length
    0..32 at every source and destination alignment, nothing outside [d, d + len) touched; lane_copy32 also with the
    source ending exactly where the run ends (no read past it)"""
    rng = np.random.default_rng(8)
    for trial in range(40):
        length = np.array([(l + trial) % 33 for l in range(64)], np.uint32)
        sal, dal = rng.integers(0, 8, 64), rng.integers(0, 8, 64)
        soff = (np.arange(64) * 48 + sal).astype(np.uint32)
        doff = (np.arange(64) * 64 + 16 + dal).astype(np.uint32)
        if mode == 0:
            # pack the runs back to back at the very end of the source buffer for the last lanes: readable == len there
            size = int(soff[-1] + length[-1])
        else:
            size = int(soff[-1] + 32 + 8)
        src = rng.integers(0, 256, size, dtype=np.uint8)
        dst = np.full(64 * 64 + 64, 0xCD, np.uint8)
        emu.lane_copy(src, soff, dst, doff, length, mode)
        want = np.full_like(dst, 0xCD)
        for l in range(64):
            n = int(length[l])
            want[int(doff[l]):int(doff[l]) + n] = src[int(soff[l]):int(soff[l]) + n]
        assert np.array_equal(dst, want), (trial, int(np.argmax(dst != want)))


def test_staged_batches_hold_long_and_overlapping_copies(emu, oracle):
    """batches that fit the LDS stage are assembled there whatever they contain: literal runs and far match sources of more
    than 32 bytes (16-byte pieces dealt out to the lanes), long copies inside the stage (several rounds of a lane, or the
    whole wave), periods of 1..40 bytes (byte-serial semantics, LL64.dec.cs:408-450), and the offset-0 sequence of a hostile
    stream, whose bytes stay as they are (such a batch is not staged)"""
    rng = np.random.default_rng(5)
    blocks = []
    for period in list(range(1, 41)) + [63, 64, 65, 100, 300]:
        parts = []
        for rep in range(6):
            pat = rng.integers(0, 256, period, dtype=np.uint8)
            parts.append(rng.integers(0, 256, int(rng.integers(1, 70)), dtype=np.uint8))            # literals, some over 32 bytes
            parts.append(np.tile(pat, int(rng.integers(40, 700)) // period + 2))                     # an overlapping match
        blocks.append(np.concatenate(parts))
    text = corpus.class_bytes("xml", 30000, 3)
    far = np.concatenate([text[:5000], rng.integers(0, 256, 3000, dtype=np.uint8), text[100:1400], rng.integers(0, 256, 50, dtype=np.uint8),
                          text[2000:2090], text[3000:3033], text[:600]])
    blocks.append(far)
    comp = [np.frombuffer(oracle.encode(b), np.uint8) for b in blocks]
    # a hand-made stream with an offset of 0: 8 literals, "match" of 6 bytes at offset 0, then ordinary sequences
    hostile = bytes([0x82]) + b"ABCDEFGH" + bytes([0, 0]) + bytes([0x21]) + b"ij" + bytes([8, 0]) + bytes([0xC0]) + b"123456789012"
    comp.append(np.frombuffer(hostile, np.uint8))
    for pair in (False, True):
        emu.pair = pair
        src, soff, slen = pack(comp)
        caps = [b.size for b in blocks] + [64]
        dst, doff, dcap = arena(caps)
        out = emu.decode_batch(src, soff, slen, dst, doff, dcap, flags=FLAG_RAW)
        for i, b in enumerate(blocks):
            assert out[i] == b.size and dst[int(doff[i]):int(doff[i]) + b.size].tobytes() == b.tobytes(), (pair, i)
        n, ref = oracle.decompress_safe(comp[-1], 64)
        assert out[-1] == n == 33
        got = dst[int(doff[-1]):int(doff[-1]) + n]
        assert got[:8].tobytes() == b"ABCDEFGH" and (got[8:14] == 0xCD).all() and got[14:].tobytes() == ref[14:n].tobytes()
    emu.pair = False
    mask = np.ones(dst.size, bool)
    for i in range(len(caps)):
        mask[int(doff[i]):int(doff[i]) + caps[i]] = False
    assert (dst[mask] == 0xCD).all()
