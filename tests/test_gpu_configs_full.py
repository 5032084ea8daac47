"""BASELINE.json configs[2], [3], [4] at FULL size against the oracle (the -m gpu suite's other files cover them at reduced
size).  The compressed inputs of the decode run come from the oracle, not from the GPU encoder, so the path is never
compared with itself.  Budget on the GPU box: about a minute each (host-side oracle work included)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from k4os.compression.lz4_amd import LZ4Codec, LZ4Level, LZ4Pickler, corpus, make_arena, pack_blocks
from k4os.compression.lz4_amd._native import FLAG_RAW_RETURN, FLAG_SEGMENTS, FLAG_X32
from k4os.compression.lz4_amd.sharding import byte_balanced_ranges

pytestmark = pytest.mark.gpu
THREADS = os.cpu_count() or 8


def test_config2_decode_only_1m_blocks_of_4k_oracle_encoded(oracle):
    """configs[2]: 1 048 576 pre-compressed 4 KiB blocks, decode-only, both variants of SURVEY.md 8(d): the text-like mix
    and random bytes.  Every block's size and every output byte is checked (on the device) against the source."""
    import torch
    from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
    n, bs = 1 << 20, 4096
    dc = DeviceCodec(0)
    for variant in ("silesia-like", "random"):
        if variant == "random":
            blocks = np.random.default_rng(7).integers(0, 256, size=(n, bs), dtype=np.uint8)
        else:
            blocks = corpus.silesia_like_blocks(n, bs, seed=3, unique_bytes_per_class=1 << 24)
        lens = np.full(n, bs, np.int32)
        off = np.arange(n, dtype=np.uint64) * bs
        caps = np.full(n, LZ4Codec.MaximumOutputSize(bs), np.int32)
        ref, ref_off = make_arena(caps)
        clen = oracle.encode_batch(blocks.reshape(-1), off, lens, ref, ref_off, caps, threads=THREADS)   # the ORACLE's streams
        assert (clen > 0).all()
        comp = DeviceBatch.from_host(ref, ref_off, clen, dc.device)
        back = DeviceBatch.empty_slots(lens, dc.device, fill=0xCD)
        dlen = dc.decode(comp, back)
        torch.cuda.synchronize()
        assert bool((dlen == bs).all().item()), variant
        want = torch.from_numpy(blocks.reshape(-1)).to(dc.device)
        assert torch.equal(back.data[:n * bs], want), variant
        assert bool((back.data[n * bs:] == 0xCD).all().item())
        del comp, back, want
        torch.cuda.empty_cache()


def test_config3_pickle_one_ranks_share_every_envelope_vs_oracle(oracle):
    """configs[3]: rank 0's byte-balanced share of the 100 000-message batch (1 KiB - 4 MiB, random / text alternating):
    EVERY envelope the GPU writes equals oracle.pickle of the message; every message comes back from Unpickle."""
    import torch
    from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
    lens_all = corpus.config4_lengths()
    lo, hi = byte_balanced_ranges(lens_all, 8)[0]
    data, off, lens = corpus.config4_share(lens_all, lo, hi)
    n = lens.size
    dc = DeviceCodec(0)
    src = DeviceBatch.from_host(data, off, lens, dc.device)
    env = DeviceBatch.empty_slots(lens.astype(np.int64) + 5, dc.device, fill=0xCD)
    plen = dc.pickle(src, env)
    psrc = DeviceBatch(env.data, env.off, plen)
    sizes = dc.unpickle_sizes(psrc)
    back = DeviceBatch.empty_slots(lens, dc.device)
    ulen = dc.unpickle(psrc, back)
    torch.cuda.synchronize()
    assert bool((sizes == torch.from_numpy(lens).to(dc.device)).all().item())
    assert bool((ulen == sizes).all().item())
    boff = back.off.cpu().numpy()
    packed = bool(np.array_equal(boff, off.view(np.int64))) if (lens % 16 == 0).all() else False
    bh = back.data.cpu().numpy()
    eh, eoff, pl = env.data.cpu().numpy(), env.off.cpu().numpy(), plen.cpu().numpy()

    def check(i):
        m = data[int(off[i]):int(off[i]) + int(lens[i])]
        return eh[eoff[i]:eoff[i] + pl[i]].tobytes() == oracle.pickle(m) and bh[boff[i]:boff[i] + lens[i]].tobytes() == m.tobytes()

    with ThreadPoolExecutor(THREADS) as pool:
        bad = [i for i, ok in enumerate(pool.map(check, range(n))) if not ok]
    assert not bad, f"{len(bad)} of {n} messages differ, first {bad[:5]} (packed={packed})"


def test_big_messages_in_segments_every_envelope_vs_oracle(oracle, monkeypatch):
    """k4lz4_segments.hpp on the device: messages cut into segments that are encoded by a wave each and joined -- with the
    shipped sizes (1.5 MiB and more, 384 KiB of warm-up: the text-like ones join) and with tiny segments and warm-ups (most
    boundaries do not verify: those messages are encoded again by the join kernel).  Every envelope equals oracle.pickle,
    nothing outside a slot's envelope is written, and a batch of one big message alone takes the same path."""
    import torch
    from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
    names = ["dickens", "xml", "samba", "webster", "nci", "mozilla", "reymont", "osdb"]
    sizes = [4 << 20, 3 << 20, (5 << 19) + 12345, 2 << 20, 1600000, 4 << 20, 1 << 20, 3500000, 700000, 65536 + 4096, 2500000, 1000]
    msgs = [corpus.class_bytes(names[i % len(names)], s, 7 + i) if i != 5 else corpus.random_bytes(s, 3) for i, s in enumerate(sizes)]
    msgs.append(np.concatenate([corpus.class_bytes("dickens", 1500000, 1), corpus.random_bytes(900000, 4), corpus.class_bytes("dickens", 1200000, 2)]))
    lens = np.array([m.size for m in msgs], np.int32)
    off = np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))]).astype(np.uint64)
    data = np.concatenate(msgs)
    want = [oracle.pickle(m) for m in msgs]
    for env_vars in ({}, {"K4LZ4_SEG_MIN": "70000", "K4LZ4_SEG_TARGET": "49152", "K4LZ4_SEG_WARM": "24576", "K4LZ4_SEG_DIV": "0"},
                     {"K4LZ4_SEG_MIN": "300000", "K4LZ4_SEG_TARGET": "200000", "K4LZ4_SEG_WARM": "400000", "K4LZ4_SEG_DIV": "0"},
                     # a run that waits for its successor's cut gives up after ONE poll (SEG_SPIN_MAX is 2^20): nearly every
                     # boundary then fails to verify in time and the join kernel encodes those blocks again -- the timeout exit
                     {"K4LZ4_SEG_SPIN_MAX": "1"},
                     {"K4LZ4_SEG_SPIN_MAX": "1", "K4LZ4_SEG_MIN": "70000", "K4LZ4_SEG_TARGET": "49152", "K4LZ4_SEG_WARM": "24576", "K4LZ4_SEG_DIV": "0"}):
        for k in ("K4LZ4_SEG_MIN", "K4LZ4_SEG_TARGET", "K4LZ4_SEG_WARM", "K4LZ4_SEG_DIV", "K4LZ4_SEG_SPIN_MAX"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env_vars.items():
            monkeypatch.setenv(k, v)
        dc = DeviceCodec(0)                                   # the switches are read when a context is created
        src = DeviceBatch.from_host(data, off, lens, dc.device)
        env = DeviceBatch.empty_slots(lens.astype(np.int64) + 5, dc.device, fill=0xCD)
        plen = dc.pickle(src, env)
        torch.cuda.synchronize()
        eh, eoff, pl = env.data.cpu().numpy(), env.off.cpu().numpy(), plen.cpu().numpy()
        for i, m in enumerate(msgs):
            assert eh[eoff[i]:eoff[i] + pl[i]].tobytes() == want[i], f"message {i} ({m.size} B) with {env_vars}"
            end = int(eoff[i + 1]) if i + 1 < len(msgs) else eh.size
            assert (eh[eoff[i] + max(int(pl[i]), int(lens[i]) + 5):end] == 0xCD).all(), f"message {i}: bytes behind the slot touched"
        # ... and LZ4Codec.Encode of the same blocks with K4LZ4_FLAG_SEGMENTS
        comp = DeviceBatch.empty_slots([LZ4Codec.MaximumOutputSize(int(n)) for n in lens], dc.device, fill=0xCD)
        clen = dc.encode(src, comp, flags=FLAG_SEGMENTS)
        torch.cuda.synchronize()
        ch, coff, cl = comp.data.cpu().numpy(), comp.off.cpu().numpy(), clen.cpu().numpy()
        for i, m in enumerate(msgs):
            assert ch[coff[i]:coff[i] + cl[i]].tobytes() == oracle.encode(m), f"block {i} ({m.size} B) with {env_vars}"
        # ... the same with the 32-bit engine's hash (K4LZ4_FLAG_X32): a subset is enough, the cut does not look at the hash
        sub = [0, 2, 12]
        slens = lens[sub]
        soff = np.concatenate([[0], np.cumsum(slens[:-1].astype(np.int64))]).astype(np.uint64)
        ssrc = DeviceBatch.from_host(np.concatenate([msgs[i] for i in sub]), soff, slens, dc.device)
        scomp = DeviceBatch.empty_slots([LZ4Codec.MaximumOutputSize(int(n)) for n in slens], dc.device, fill=0xCD)
        sclen = dc.encode(ssrc, scomp, flags=FLAG_SEGMENTS | FLAG_X32 | FLAG_RAW_RETURN)
        torch.cuda.synchronize()
        sh, so, sl = scomp.data.cpu().numpy(), scomp.off.cpu().numpy(), sclen.cpu().numpy()
        for j, i in enumerate(sub):
            r, w = oracle.compress_fast_x32(msgs[i])
            assert int(sl[j]) == r and sh[so[j]:so[j] + r].tobytes() == w[:r].tobytes(), f"block {i} with the x32 hash and {env_vars}"
        one = DeviceBatch.from_host(msgs[0], np.zeros(1, np.uint64), lens[:1], dc.device)
        env1 = DeviceBatch.empty_slots(lens[:1].astype(np.int64) + 5, dc.device, fill=0xCD)
        p1 = dc.pickle(one, env1)
        torch.cuda.synchronize()
        o1 = int(env1.off.cpu().numpy()[0])
        assert env1.data.cpu().numpy()[o1:o1 + int(p1.cpu().numpy()[0])].tobytes() == want[0]
    # ... and through the host-pointer API (LZ4Pickler.Pickle on host memory: staged up, pickled in segments, copied back)
    from k4os.compression.lz4_amd import LZ4Pickler
    for i in (0, 12, 9):
        assert bytes(LZ4Pickler.Pickle(msgs[i])) == want[i], f"host-pointer pickle of message {i}"


def test_config4_hc_l03_all_4096_blocks_vs_oracle(oracle):
    """configs[4]: L03_HC over the 4096 x 64 KiB batch: every block's bytes and the ratio equal the oracle's"""
    import torch
    from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
    n, bs = 4096, 65536
    blocks = corpus.silesia_like_blocks(n, bs, seed=2)
    lens = np.full(n, bs, np.int32)
    off = np.arange(n, dtype=np.uint64) * bs
    caps = np.full(n, LZ4Codec.MaximumOutputSize(bs), np.int32)
    ref, ref_off = make_arena(caps)
    want = oracle.encode_batch(blocks.reshape(-1), off, lens, ref, ref_off, caps, level=3, threads=THREADS)
    dc = DeviceCodec(0)
    src = DeviceBatch.from_host(blocks.reshape(-1), off, lens, dc.device)
    comp = DeviceBatch.empty_slots(caps, dc.device, fill=0xCD)
    clen = dc.encode(src, comp, level=LZ4Level.L03_HC)
    torch.cuda.synchronize()
    clen_h = clen.cpu().numpy()
    assert np.array_equal(clen_h, want)
    assert int(clen_h.sum()) == int(want.sum())            # the ratio, exactly
    ch, coff = comp.data.cpu().numpy(), comp.off.cpu().numpy()
    for i in range(n):
        assert ch[coff[i]:coff[i] + clen_h[i]].tobytes() == ref[int(ref_off[i]):int(ref_off[i]) + int(want[i])].tobytes(), i


def test_hc_device_call_with_reservation_only_enqueues_and_a_small_one_fails_loudly(oracle):
    """k4lz4_ctx_reserve_hc: device-resident HC encodes sized from the reservation give the same bytes; a batch bigger
    than the reservation is not encoded and the context says so at the next synchronising call (never a silent -1)."""
    import torch
    from k4os.compression.lz4_amd import _native
    from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
    blocks = corpus.silesia_like_blocks(96, 65536, seed=5)
    n = blocks.shape[0]
    lens = np.full(n, 65536, np.int32)
    off = np.arange(n, dtype=np.uint64) * 65536
    dc = DeviceCodec(0)
    src = DeviceBatch.from_host(blocks.reshape(-1), off, lens, dc.device)
    caps = np.full(n, LZ4Codec.MaximumOutputSize(65536), np.int32)
    comp = DeviceBatch.empty_slots(caps, dc.device, fill=0xCD)
    dc.ctx.check(dc.lib.k4lz4_ctx_reserve_hc(dc.ctx.handle, n * 65536, 65536))
    clen = dc.encode(src, comp, level=LZ4Level.L06_HC)
    dc.ctx.check(dc.lib.k4lz4_synchronize(dc.ctx.handle, None))
    torch.cuda.synchronize()
    ch, coff, cl = comp.data.cpu().numpy(), comp.off.cpu().numpy(), clen.cpu().numpy()
    for i in range(n):
        r, w = oracle.compress_hc(blocks[i], 6)
        assert cl[i] == r and ch[coff[i]:coff[i] + r].tobytes() == w[:r].tobytes(), i
    # too small: half the bytes
    dc.ctx.check(dc.lib.k4lz4_ctx_reserve_hc(dc.ctx.handle, n * 65536 // 2, 65536))
    clen2 = dc.encode(src, comp, level=LZ4Level.L03_HC)
    torch.cuda.synchronize()
    rc = dc.lib.k4lz4_synchronize(dc.ctx.handle, None)
    assert rc == _native.E_NOMEM and b"reserve" in dc.lib.k4lz4_last_error(dc.ctx.handle)
    assert (clen2.cpu().numpy() == -1).all()
    dc.ctx.check(dc.lib.k4lz4_ctx_reserve_hc(dc.ctx.handle, 0, 0))
    clen3 = dc.encode(src, comp, level=LZ4Level.L03_HC)      # sized by asking the device again
    torch.cuda.synchronize()
    dc.ctx.check(dc.lib.k4lz4_synchronize(dc.ctx.handle, None))
    assert (clen3.cpu().numpy() > 0).all()


def test_status_word_is_per_context_and_a_short_hc_reservation_never_yields_raw_pickles(oracle):
    """The status word belongs to the context whose call launched the kernels: another context on the same device neither
    sees nor clears it.  And the pickle path: an HC reservation that is too small must not come back as valid raw
    (uncompressed) envelopes -- outLen says failure, the context says why."""
    import torch
    from k4os.compression.lz4_amd import _native
    from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
    blocks = corpus.silesia_like_blocks(48, 65536, seed=6)
    n = blocks.shape[0]
    lens = np.full(n, 65536, np.int32)
    off = np.arange(n, dtype=np.uint64) * 65536
    a, b = DeviceCodec(0), DeviceCodec(0)
    src = DeviceBatch.from_host(blocks.reshape(-1), off, lens, a.device)
    env = DeviceBatch.empty_slots(np.full(n, 65536 + 5, np.int32), a.device, fill=0xCD)
    a.ctx.check(a.lib.k4lz4_ctx_reserve_hc(a.ctx.handle, n * 65536 // 2, 65536))
    plen = a.pickle(src, env, level=LZ4Level.L03_HC)
    torch.cuda.synchronize()
    # context b synchronises first: it finds nothing, and it does not take a's report away
    comp = DeviceBatch.empty_slots(np.full(n, LZ4Codec.MaximumOutputSize(65536), np.int32), b.device)
    clen = b.encode(src, comp)
    b.ctx.check(b.lib.k4lz4_synchronize(b.ctx.handle, None))
    assert (clen.cpu().numpy() > 0).all()
    rc = a.lib.k4lz4_synchronize(a.ctx.handle, None)
    assert rc == _native.E_NOMEM and b"reserve" in a.lib.k4lz4_last_error(a.ctx.handle)
    assert (plen.cpu().numpy() == -1).all(), "a failed HC pickle must not look like a raw envelope"
    assert a.lib.k4lz4_synchronize(a.ctx.handle, None) == 0        # read once, cleared
    a.ctx.check(a.lib.k4lz4_ctx_reserve_hc(a.ctx.handle, 0, 0))
    plen = a.pickle(src, env, level=LZ4Level.L03_HC)
    a.ctx.check(a.lib.k4lz4_synchronize(a.ctx.handle, None))
    eh, eoff, pl = env.data.cpu().numpy(), env.off.cpu().numpy(), plen.cpu().numpy()
    for i in range(0, n, 5):
        want = oracle.pickle(blocks[i], 3)
        assert pl[i] == len(want) and eh[eoff[i]:eoff[i] + pl[i]].tobytes() == want, i


def test_one_context_two_streams_do_not_race_on_its_scratch(oracle):
    """the context's dispatch-order / hash-table scratch is shared by all calls: two calls in a row on different streams
    must give the results of two calls in a row on one stream"""
    import torch
    from k4os.compression.lz4_amd.device import DeviceBatch, DeviceCodec
    dc = DeviceCodec(0)
    sets = []
    for seed in (11, 12):
        blocks = corpus.silesia_like_blocks(1536, 65536, seed=seed)
        n = blocks.shape[0]
        lens = np.full(n, 65536, np.int32)
        off = np.arange(n, dtype=np.uint64) * 65536
        src = DeviceBatch.from_host(blocks.reshape(-1), off, lens, dc.device)
        comp = DeviceBatch.empty_slots(np.full(n, LZ4Codec.MaximumOutputSize(65536)), dc.device)
        sets.append((blocks, src, comp, dc.new_out_len(n)))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(dc.device), torch.cuda.Stream(dc.device)]
    for rep in range(3):
        for (blocks, src, comp, clen), st in zip(sets, streams):
            with torch.cuda.stream(st):
                dc.encode(src, comp, clen)
    torch.cuda.synchronize()
    for blocks, src, comp, clen in sets:
        cl, ch, coff = clen.cpu().numpy(), comp.data.cpu().numpy(), comp.off.cpu().numpy()
        for i in range(blocks.shape[0]):
            assert ch[coff[i]:coff[i] + cl[i]].tobytes() == oracle.encode(blocks[i]), i


def test_compress_hc_refuses_levels_below_3():
    from k4os.compression.lz4_amd import _native
    lib = _native.load_library()
    data = corpus.lorem(1000)
    dst = np.zeros(2000, np.uint8)
    assert lib.k4lz4_compress_hc(data.ctypes.data, dst.ctypes.data, data.size, dst.size, 1) == 0
    assert lib.k4lz4_last_status() == _native.E_ARG
    assert lib.k4lz4_compress_hc(data.ctypes.data, dst.ctypes.data, data.size, dst.size, 3) > 0
    assert lib.k4lz4_last_status() == 0


def test_hand_written_chains_agree_with_their_c_twins_on_the_device():
    """follow_tokens / hop_chain / hop_chain_pairs are inline ISA; the emulator suite runs their C twins.  Both forms run
    here on the GPU over random well-formed hop words and must agree in every output (k4_chain_selftest_kernel)."""
    import ctypes as C
    from k4os.compression.lz4_amd import _native
    ctx = _native.default_context()
    res = (C.c_uint32 * 3)()
    for seed in (1, 2, 0xC0FFEE):
        ctx.check(ctx.lib.k4lz4_selftest_chains(ctx.handle, 512, 200, seed, res))
        assert list(res) == [0, 0, 0], f"ISA and C disagree (token chain, hop chain, pair chain): {list(res)}"


def test_host_pointer_batches_big_enough_for_the_staged_path_round_trip(oracle):
    """k4lz4_encode_batch / k4lz4_decode_batch on pageable host memory, 1024 x 64 KiB: the source goes up through the pinned
    double buffers, the compressed blocks come back packed and are scattered into worst-case slots, and the decode call
    finds them far apart there and sends only the blocks (staged_upload_packed).  Bytes against the oracle both ways."""
    n, bs = 1024, 65536
    blocks = corpus.silesia_like_blocks(n, bs, seed=9)
    src = blocks.reshape(-1)
    off = np.arange(n, dtype=np.uint64) * bs
    lens = np.full(n, bs, np.int32)
    caps = np.full(n, LZ4Codec.MaximumOutputSize(bs), np.int32)
    dst, doff = make_arena(caps, fill=0xCD)
    out = LZ4Codec.EncodeBatchPacked(src, off, lens, dst, doff, caps)
    ref, roff = make_arena(caps, fill=0xCD)
    want = oracle.encode_batch(src, off, lens, ref, roff, caps, threads=os.cpu_count() or 8)
    assert np.array_equal(out, want)
    for i in range(n):                                          # every block, and the rest of every slot untouched
        assert np.array_equal(dst[int(doff[i]):int(doff[i]) + caps[i]], ref[int(roff[i]):int(roff[i]) + caps[i]]), i
    back, boff = make_arena(lens, fill=0xCD)
    got = LZ4Codec.DecodeBatchPacked(dst, doff, out, back, boff, lens)
    assert (got == bs).all()
    for i in range(n):
        assert np.array_equal(back[int(boff[i]):int(boff[i]) + bs], blocks[i]), i


@pytest.mark.parametrize("tail", [5, 1, 7, 64 * 8 + 3])
def test_host_pointer_upload_whose_last_chunk_does_not_divide_by_the_copy_threads(oracle, tail):
    """A source span of 40 MiB + `tail` bytes goes up in chunks of 16 MiB, each copied into its pinned buffer by up to eight
    threads: the last chunk is 8 MiB + tail, an eighth of it a whole number of 64-byte lines with `tail` bytes left over -- which
    the split used to drop (the last bytes of the upload stayed what the buffer held before: round 5, found by
    tests/tools/gpu_stress_encode.py as 5 wrong bytes at the end of one block in 12 000).  The buffers are dirtied by a first
    call with other bytes; then every block against the oracle, the last one to its last literal."""
    bs = 65536
    n = 640
    dirt = np.full(n * bs + 4096, 0x5A, np.uint8)
    off = np.arange(n, dtype=np.uint64) * bs
    lens = np.full(n, bs, np.int32)
    caps = np.full(n, LZ4Codec.MaximumOutputSize(bs + 4096), np.int32)
    dst, doff = make_arena(caps, fill=0xCD)
    LZ4Codec.EncodeBatchPacked(dirt, off, lens, dst, doff, caps)
    blocks = [b for b in corpus.silesia_like_blocks(n - 1, bs, seed=13)] + [corpus.class_bytes("dickens", bs + tail, 4)]
    src, off, lens = pack_blocks(blocks)
    assert src.size == 40 * (1 << 20) + tail
    dst, doff = make_arena(caps, fill=0xCD)
    out = LZ4Codec.EncodeBatchPacked(src, off, lens, dst, doff, caps)
    ref, roff = make_arena(caps, fill=0xCD)
    want = oracle.encode_batch(src, off, lens, ref, roff, caps, threads=os.cpu_count() or 8)
    assert np.array_equal(out, want)
    for i in range(n):
        assert np.array_equal(dst[int(doff[i]):int(doff[i]) + want[i]], ref[int(roff[i]):int(roff[i]) + want[i]]), i


def test_host_pointer_staging_branches_big_ragged_shuffled(oracle):
    """The staged host path's other branches: a decode call with more than 128 MiB of compressed input in worst-case slots
    (packed upload, two parts), with empty blocks, a corrupt block, a destination that is too small and one block of more
    than a staging chunk (17 MiB) among them; and the same blocks handed over in shuffled order (srcOff not ascending:
    the unpacked fall-back).  Results and bytes against the oracle."""
    rng = np.random.default_rng(77)
    bs = 65536
    rand = corpus.random_bytes(1200 * bs, 5).reshape(1200, bs)          # incompressible: 65 809-byte streams
    text = corpus.silesia_like_blocks(900, bs, seed=13)
    big = corpus.class_bytes("webster", 17 << 20, 3)
    blocks = [rand[i] for i in range(1200)] + [text[i] for i in range(900)] + [big, np.zeros(0, np.uint8), np.zeros(0, np.uint8)]
    order = rng.permutation(len(blocks))
    blocks = [blocks[i] for i in order]
    n = len(blocks)
    lens = np.array([b.size for b in blocks], np.int32)
    off = np.concatenate(([0], np.cumsum(lens[:-1].astype(np.int64)))).astype(np.uint64)
    src = np.concatenate([b for b in blocks if b.size])
    caps = np.array([LZ4Codec.MaximumOutputSize(int(l)) for l in lens], np.int32)
    comp, coff = make_arena(caps, fill=0xCD)
    clen = LZ4Codec.EncodeBatchPacked(src, off, lens, comp, coff, caps)
    ref, roff = make_arena(caps, fill=0xCD)
    want = oracle.encode_batch(src, off, lens, ref, roff, caps, threads=os.cpu_count() or 8)
    assert np.array_equal(clen, want)
    assert np.array_equal(comp, ref)                                  # every byte of every slot, slack included
    assert int(clen[clen > 0].sum()) > (128 << 20) * 3 // 4 and int(caps.sum()) > (128 << 20)
    # decode: exact destinations, except one corrupt stream and one destination that is too small
    i_bad, i_small = int(np.flatnonzero(lens == bs)[3]), int(np.flatnonzero(lens == bs)[11])
    comp[int(coff[i_bad]) + 7] ^= 0x55
    comp[int(coff[i_bad]) + 200:int(coff[i_bad]) + 208] = 0
    dcap = lens.copy()
    dcap[i_small] = bs - 9
    back, boff = make_arena(dcap, fill=0xCD)
    got = LZ4Codec.DecodeBatchPacked(comp, coff, clen, back, boff, dcap)
    for i in range(n):
        c = comp[int(coff[i]):int(coff[i]) + max(int(clen[i]), 0)]
        if lens[i] == 0:
            assert got[i] == 0, i
            continue
        r, out = oracle.decompress_safe(c, int(dcap[i]))
        assert got[i] == (r if r > 0 else -1), (i, int(got[i]), r)
        if r > 0:
            assert np.array_equal(back[int(boff[i]):int(boff[i]) + r], out[:r]), i
    assert got[i_small] == -1 and (got[np.arange(n) != i_bad] != 0)[lens[np.arange(n) != i_bad] > 0].all()
    # the same compressed blocks in shuffled order: offsets no longer ascend
    perm = rng.permutation(n)
    got2 = LZ4Codec.DecodeBatchPacked(comp, coff[perm], clen[perm], back, boff[perm], dcap[perm])
    assert np.array_equal(got2, got[perm])


def test_registered_host_buffers_give_the_same_bytes_as_pageable_ones():
    """k4lz4_host_register: with the caller's buffers page-locked the source goes up straight from its pages and a decoded batch in
    adjacent slots comes back in one copy per part (direct_download); compressed blocks in worst-case slots (every block a
    run of its own), ragged lengths, a corrupt stream and a destination that is too small keep to the staged way or mix both.
    Every byte of every destination arena -- slack of the slots included -- must equal what the same calls write without
    registration; registering twice / unregistering something unknown are refused."""
    from k4os.compression.lz4_amd import host_register, host_unregister
    n, bs = 4096, 65536                                 # (4096 blocks: a registered destination comes back in eight parts)
    blocks = corpus.silesia_like_blocks(n, bs, seed=21)
    src = np.ascontiguousarray(blocks.reshape(-1))
    off = np.arange(n, dtype=np.uint64) * bs
    lens = np.full(n, bs, np.int32)
    lens[5::97] -= 13                                   # ragged: these blocks end short of the next one's start
    caps = np.full(n, LZ4Codec.MaximumOutputSize(bs), np.int32)
    plain_c, coff = make_arena(caps, fill=0xCD)
    plain_len = LZ4Codec.EncodeBatchPacked(src, off, lens, plain_c, coff, caps)
    dcap = lens.copy()
    i_bad, i_small = 77, 3500
    dcap[i_small] -= 9
    plain_c[int(coff[i_bad]) + 9] ^= 0x5A
    plain_c[int(coff[i_bad]) + 300:int(coff[i_bad]) + 308] = 0
    plain_b, boff = make_arena(dcap, fill=0xCD)
    plain_got = LZ4Codec.DecodeBatchPacked(plain_c, coff, plain_len, plain_b, boff, dcap)
    assert plain_got[i_small] == -1 and (np.delete(plain_got, [i_bad, i_small]) == np.delete(lens, [i_bad, i_small])).all()

    reg_c, _ = make_arena(caps, fill=0xCD)
    reg_b, _ = make_arena(dcap, fill=0xCD)
    held = []
    try:
        for arr in (src, reg_c, reg_b):
            host_register(arr); held.append(arr)
        reg_len = LZ4Codec.EncodeBatchPacked(src, off, lens, reg_c, coff, caps)
        assert np.array_equal(reg_len, plain_len)
        reg_c[int(coff[i_bad]) + 9] ^= 0x5A
        reg_c[int(coff[i_bad]) + 300:int(coff[i_bad]) + 308] = 0
        assert np.array_equal(reg_c, plain_c)
        reg_got = LZ4Codec.DecodeBatchPacked(reg_c, coff, reg_len, reg_b, boff, dcap)
        assert np.array_equal(reg_got, plain_got)
        assert np.array_equal(reg_b, plain_b)
        # a batch that lies only partly inside a registered range is staged as a whole
        part = np.concatenate([src[:bs * 8], src[:bs * 8]])
        got8 = LZ4Codec.EncodeBatchPacked(part, off[:16], lens[:16] * 0 + bs, reg_c, coff[:16], caps[:16])
        assert (got8[:8] == got8[8:]).all() and (got8 > 0).all()
        with pytest.raises(ValueError):
            host_register(src[100:200])                 # overlaps a registered range
    finally:
        for arr in held:
            host_unregister(arr)
    with pytest.raises(ValueError):
        host_unregister(src)                            # no longer registered


def test_two_ranks_on_one_gpu_run_the_real_multi_rank_backend():
    """`bench.py --strong --gpus 2` as two processes (torch.distributed.run) that share GPU 0 (K4LZ4_RANK_DEVICE=0): a context
    and a DevicePickleBackend per rank, byte-balanced ranges, the size vector gathered (over gloo: RCCL refuses two ranks
    on one device) -- and rank 0 compares the WHOLE gathered vector with a single-rank run of the same batch."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, K4LZ4_RANK_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(root, "bench.py"), "--strong", "--gpus", "2", "--messages", "3000"]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and lines, p.stderr[-2000:]
    r = json.loads(lines[-1])
    assert r["n_gpus"] == 2 and r["roundtrip_ok_all_ranks"] and r["size_vector_sample_equals_oracle"]
    assert r["size_vector_equals_single_rank_run"] is True
    assert len(r["rank_busy_ms"]) == 2 and r["critical_path_ms"] > 0


@pytest.mark.parametrize("n", [5001, 8192])
def test_uniform_cost_pickle_batch_leaves_no_block_to_neither_encoder_kernel(oracle, n):
    """a batch that lands in ONE cost bucket made k4_order_kernel round the device-side split up to
    n_lds + 1 while the host launched the LDS-table kernel with n_lds slots -- order[n_lds] was encoded by neither kernel.
    n uniform messages (same class, same length) through the unchunked pickle path: every envelope vs the oracle."""
    msgs = [corpus.class_bytes("dickens", 3000, 900 + (i % 64)) for i in range(n)]
    got = LZ4Pickler.PickleBatch(msgs)
    want = {}
    for i in range(n):
        k = i % 64
        if k not in want:
            want[k] = oracle.pickle(msgs[i])
        assert got[i] == want[k], i


def test_pickle_batch_honours_the_per_call_x32_flag(oracle):
    """K4LZ4_FLAG_X32 on a pickle call was dropped by the segment-capable pickle path; a >= 64 KiB
    message must carry the 32-bit engine's block (oracle arm pinned to LL32 compiled here, tests/test_ref_pins.py)"""
    from k4os.compression.lz4_amd import _native
    from k4os.compression.lz4_amd.codec import _batch_args, pack_blocks, make_arena
    ctx = _native.default_context()
    msgs = [corpus.class_bytes("dickens", 70000, 1), corpus.lorem(150000), corpus.class_bytes("xml", 3000, 2)]
    src, soff, slen = pack_blocks(msgs)
    caps = np.array([ctx.lib.k4lz4_pickle_bound(m.size) for m in msgs], dtype=np.int32)
    dst, doff = make_arena(caps)
    out = np.empty(len(msgs), dtype=np.int32)
    ctx.check(ctx.lib.k4lz4_pickle_batch(ctx.handle, *_batch_args(src, soff, slen, dst, doff, caps, out), 0, _native.FLAG_X32))
    for i, m in enumerate(msgs):
        env = dst[int(doff[i]):int(doff[i]) + int(out[i])].tobytes()
        r, w = oracle.compress_fast_x32(m)
        assert env.endswith(w[:r].tobytes()), i
        if m.size >= 65547:
            assert env != oracle.pickle(m)


def test_headline_bench_with_two_ranks_is_the_command_the_scaling_run_launches(oracle):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` -- the weak-scaling (headline) path with
    world > 1: rank r encodes + decodes its own batch (seed 2 + r), barrier + max-over-ranks timing, the size vector
    all-gathered.  Two processes share GPU 0 (K4LZ4_RANK_DEVICE=0, gloo for the three small exchanges: RCCL refuses two ranks
    on one device).  n_gpus, bit_exact and the gathered total of compressed bytes = what the oracle gives for seeds 2 and 3."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, K4LZ4_RANK_DEVICE="0", MASTER_ADDR="127.0.0.1")
    nb = 1024
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29579", os.path.join(root, "bench.py"), "--gpus", "2", "--blocks", str(nb), "--steps", "3", "--warmup", "1",
           "--no-host-path", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and lines, p.stderr[-2000:]
    r = json.loads(lines[-1])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["bit_exact"] is True
    want = 0
    for seed in (2, 3):
        blocks = corpus.silesia_like_blocks(nb, 65536, seed=seed)
        off = np.arange(nb, dtype=np.uint64) * 65536
        lens = np.full(nb, 65536, np.int32)
        caps = np.full(nb, LZ4Codec.MaximumOutputSize(65536), np.int32)
        dst, doff = make_arena(caps)
        want += int(oracle.encode_batch(blocks.reshape(-1), off, lens, dst, doff, caps, threads=THREADS).astype(np.int64).sum())
    assert r["config"]["total_compressed_bytes_all_gpus"] == want
    assert r["value"] > 0 and abs(r["value"] - 2 * nb * 65536 / 2 ** 30 / (r["ms_per_step"] * 1e-3)) < 0.01 * r["value"]
    assert "all 2 rank(s)" in r["bit_exact_scope"] and r["roundtrip_all_ranks"] is True
    # `bit_exact` vouches for EVERY rank's blocks: one byte of rank 1's output flipped (in its host copy, before its own comparison
    # with the oracle) must turn the line's flag to false although rank 0's share is intact
    env_bad = dict(env, K4LZ4_TEST_CORRUPT_RANK="1")
    cmd_bad = [c if c != "29579" else "29581" for c in cmd]
    p = subprocess.run(cmd_bad, cwd=root, env=env_bad, capture_output=True, text=True, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and lines, p.stderr[-2000:]
    assert json.loads(lines[-1])["bit_exact"] is False
