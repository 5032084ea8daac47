"""LZ4 frame format over batched blocks, backed by libk4lz4.so.

Mirrors (K4os.Compression.LZ4.Streams):
  LZ4EncoderSettings / LZ4Descriptor                 LZ4EncoderSettings.cs:8-47, Frames/LZ4Descriptor.cs
  LZ4Frame.Encode / LZ4Frame.Decode                  LZ4Frame.cs (span / buffer-writer overloads)
  frame writer   magic, FLG/BD, header checksum byte, block length with raw bit, block checksum,
                 EndMark, content checksum           Frames/LZ4FrameWriter.cs:57-108,:159-189,
                                                     Frames/LZ4FrameWriter.async.cs:15-27,:75-90
  frame reader                                       Frames/LZ4FrameReader.async.cs:52-136
What runs where: splitting into blocks, the header and the block table are host index work; block
encoding (with the allowCopy rule), block decoding (independent blocks as one batch, chained blocks as
one in-order stream per wavefront) and every XXH32 -- header byte included -- run in the HIP kernels.

Differences from the reference, all deliberate:
  * the writer only makes frames of INDEPENDENT blocks (ChainBlocks=False); the reference's default
    is chained blocks, whose encoder is serial (*_continue).  The READER takes both.
  * ContentLength in the header is written when asked for (the reference's writer throws
    NotImplemented, LZ4FrameWriter.cs:86-88) and verified by the reader, like the reference's reader.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _native
from .codec import LZ4Codec, LZ4Level, _ro_view, pack_blocks, make_arena, _batch_args
from .encoders import encode_blocks_packed

MAGIC = 0x184D2204
K64, K256, M1, M4 = 64 << 10, 256 << 10, 1 << 20, 4 << 20


class InvalidDataException(Exception):
    """System.IO.InvalidDataException (magic number, header checksum, block / content checksum)"""


class NotImplementedException(Exception):
    """System.NotImplementedException (predefined dictionaries)"""


@dataclass
class LZ4EncoderSettings:                     # LZ4EncoderSettings.cs (ChainBlocks default differs, see module docstring)
    ContentLength: Optional[int] = None
    ChainBlocks: bool = False
    BlockSize: int = K64
    ContentChecksum: bool = False
    BlockChecksum: bool = False
    CompressionLevel: LZ4Level = LZ4Level.L00_FAST
    ExtraMemory: int = 0

    @property
    def Dictionary(self):
        return None


@dataclass
class LZ4Descriptor:                          # Frames/LZ4Descriptor.cs
    ContentLength: Optional[int]
    ContentChecksum: bool
    Chaining: bool
    BlockChecksum: bool
    Dictionary: Optional[int]
    BlockSize: int


@dataclass
class FrameInfo:
    """what the reader learns from walking a frame (no payload is touched)"""
    descriptor: LZ4Descriptor
    header: bytes                 # FLG .. before HC: the bytes the header checksum covers
    header_checksum: int
    block_off: List[int] = field(default_factory=list)     # payload offsets inside the frame
    block_len: List[int] = field(default_factory=list)     # payload lengths, bit 31 = raw
    block_checksum: List[int] = field(default_factory=list)
    content_checksum: Optional[int] = None
    consumed: int = 0


def max_block_size_code(block_size: int) -> int:          # LZ4FrameWriter.cs:184-189
    if block_size <= K64:
        return 4
    if block_size <= K256:
        return 5
    if block_size <= M1:
        return 6
    if block_size <= M4:
        return 7
    raise ValueError(f"Invalid block size ${block_size} for this operation")


def max_block_size(code: int) -> int:                     # LZ4FrameReader.cs:56-59
    return {7: M4, 6: M1, 5: K256, 4: K64}.get(code, K64)


def frame_header(d: LZ4Descriptor) -> bytes:
    """FLG, BD [, content size] -- the bytes covered by the header checksum (LZ4FrameWriter.cs:65-100)"""
    if d.Dictionary is not None:
        raise NotImplementedException("Predefined dictionaries feature is not implemented")
    flg = (1 << 6) | ((0 if d.Chaining else 1) << 5) | ((1 if d.BlockChecksum else 0) << 4) | \
          ((1 if d.ContentLength is not None else 0) << 3) | ((1 if d.ContentChecksum else 0) << 2)
    bd = max_block_size_code(d.BlockSize) << 4
    out = bytes([flg & 0xFF, bd & 0xFF])
    if d.ContentLength is not None:
        out += struct.pack("<Q", d.ContentLength)
    return out


def parse_frame(frame, start: int = 0) -> FrameInfo:
    """walks one frame: header fields, block table, checksums as stored (LZ4FrameReader.async.cs:52-136).
    Checksums are NOT verified here -- that is batch work for the device."""
    buf = _ro_view(frame, "source")
    pos, end = int(start), buf.size

    def need(n):
        if end - pos < n:
            raise EOFError("Unexpected end of stream")               # EndOfStream()
    need(4)
    if struct.unpack_from("<I", buf, pos)[0] != MAGIC:
        raise InvalidDataException("LZ4 frame magic number expected")
    pos += 4
    hdr0 = pos
    need(2)
    flg, bd = int(buf[pos]), int(buf[pos + 1])
    pos += 2
    version = (flg >> 6) & 0x11                                     # as written at LZ4FrameReader.async.cs:72
    if version != 1:
        raise InvalidDataException(f"LZ4 frame version unknown: {version}")
    chaining = ((flg >> 5) & 1) == 0
    bsum = ((flg >> 4) & 1) != 0
    has_size = ((flg >> 3) & 1) != 0
    csum = ((flg >> 2) & 1) != 0
    has_dict = (flg & 1) != 0
    content_length = None
    if has_size:
        need(8)
        content_length = struct.unpack_from("<Q", buf, pos)[0]
        pos += 8
    dict_id = None
    if has_dict:
        need(4)
        dict_id = struct.unpack_from("<I", buf, pos)[0]
        pos += 4
    header = buf[hdr0:pos].tobytes()
    need(1)
    hc = int(buf[pos])
    pos += 1
    info = FrameInfo(LZ4Descriptor(content_length, csum, chaining, bsum, dict_id, max_block_size((bd >> 4) & 7)), header, hc)
    while True:
        need(4)
        lc = struct.unpack_from("<I", buf, pos)[0]
        pos += 4
        if lc == 0:
            break
        n = lc & 0x7FFFFFFF
        need(n + (4 if bsum else 0))
        info.block_off.append(pos)
        info.block_len.append(lc)
        pos += n
        if bsum:
            info.block_checksum.append(struct.unpack_from("<I", buf, pos)[0])
            pos += 4
    if csum:
        need(4)
        info.content_checksum = struct.unpack_from("<I", buf, pos)[0]
        pos += 4
    info.consumed = pos - int(start)
    return info


def assemble_frame(d: LZ4Descriptor, header_hash: int, payloads: Sequence[bytes], raw: Sequence[bool],
                   block_hashes: Optional[Sequence[int]], content_hash: Optional[int]) -> bytes:
    """lays the pieces out as the writer does (LZ4FrameWriter.async.cs:15-27,:75-90)"""
    parts = [struct.pack("<I", MAGIC), frame_header(d), bytes([(header_hash >> 8) & 0xFF])]
    for i, p in enumerate(payloads):
        parts.append(struct.pack("<I", len(p) | (0x80000000 if raw[i] else 0)))     # BlockLengthCode
        parts.append(p)
        if d.BlockChecksum:
            parts.append(struct.pack("<I", block_hashes[i]))
    parts.append(struct.pack("<I", 0))                                                  # EndMark
    if d.ContentChecksum:
        parts.append(struct.pack("<I", content_hash))
    return b"".join(parts)


def xxh32_many(buffers: Sequence, ctx: Optional[_native.Context] = None) -> np.ndarray:
    """XXH32.DigestOf of every buffer, one kernel launch (seed 0, as every call site of the reference)"""
    ctx = ctx or _native.default_context()
    views = [_ro_view(b, "buffer") for b in buffers]
    if not views:
        return np.zeros(0, np.uint32)
    data, off, _ = pack_blocks(views)
    lens = np.array([v.size for v in views], dtype=np.uint64)
    out = np.zeros(len(views), dtype=np.uint32)
    ctx.check(ctx.lib.k4lz4_xxh32_batch(ctx.handle, data.ctypes.data, off.ctypes.data, lens.ctypes.data, out.ctypes.data,
                                        len(views), 0))
    return out


class LZ4Frame:
    """LZ4Frame.Encode / Decode for whole buffers, plus their batch forms."""

    # ---- encode -----------------------------------------------------------------------------------
    @staticmethod
    def Encode(source, settings: Optional[LZ4EncoderSettings] = None, level: Optional[LZ4Level] = None) -> bytes:
        return LZ4Frame.EncodeBatch([source], settings, level)[0]

    @staticmethod
    def EncodeBatch(sources: Sequence, settings: Optional[LZ4EncoderSettings] = None, level: Optional[LZ4Level] = None,
                    ctx: Optional[_native.Context] = None) -> List[bytes]:
        s = settings or LZ4EncoderSettings()
        if level is not None:
            s = LZ4EncoderSettings(**{**s.__dict__, "CompressionLevel": LZ4Level(level)})
        if s.ChainBlocks:
            raise NotImplementedException("chained blocks are encoded serially (LZ4_compress_*_continue): not offered; "
                                          "use ChainBlocks=False")
        max_block_size_code(s.BlockSize)
        ctx = ctx or _native.default_context()
        contents = [_ro_view(x, "source") for x in sources]
        bs = int(s.BlockSize)
        blocks, owner = [], []
        for f, c in enumerate(contents):
            if s.ContentLength is not None and s.ContentLength != c.size:
                raise ValueError("ContentLength does not match the source length")
            for p in range(0, c.size, bs):
                blocks.append(c[p:p + bs])
                owner.append(f)
        out, arena, aoff = encode_blocks_packed(blocks, s.CompressionLevel, True, ctx) if blocks else (np.zeros(0, np.int32), None, None)
        payloads, raw = [], []
        for n, o in zip(out, aoff if blocks else []):
            if n == 0:
                raise RuntimeError("Failed to encode chunk. Target buffer too small.")       # LZ4EncoderBase.cs:75-77
            payloads.append(arena[int(o):int(o) + abs(int(n))])
            raw.append(n < 0)
        descs = [LZ4Descriptor(s.ContentLength, s.ContentChecksum, False, s.BlockChecksum, None, bs) for _ in contents]
        # every XXH32 of the batch in one launch: headers, then block payloads, then contents
        to_hash = [np.frombuffer(frame_header(d), np.uint8) for d in descs]
        if s.BlockChecksum:
            to_hash += payloads
        if s.ContentChecksum:
            to_hash += contents
        hashes = xxh32_many(to_hash, ctx)
        nf, nb = len(contents), len(blocks)
        bh = hashes[nf:nf + nb] if s.BlockChecksum else None
        ch = hashes[nf + (nb if s.BlockChecksum else 0):] if s.ContentChecksum else None
        frames = []
        k = 0
        for f, d in enumerate(descs):
            k0 = k
            while k < nb and owner[k] == f:
                k += 1
            frames.append(assemble_frame(d, int(hashes[f]), [payloads[i].tobytes() for i in range(k0, k)], raw[k0:k],
                                         None if bh is None else [int(x) for x in bh[k0:k]],
                                         None if ch is None else int(ch[f])))
        return frames

    # ---- decode -----------------------------------------------------------------------------------
    @staticmethod
    def Decode(source, settings=None) -> bytes:
        return LZ4Frame.DecodeBatch([source])[0]

    @staticmethod
    def DecodeBatch(sources: Sequence, ctx: Optional[_native.Context] = None) -> List[bytes]:
        """every element holds one frame (bytes after it are ignored, as a reader that stops at EndMark does)"""
        ctx = ctx or _native.default_context()
        bufs = [_ro_view(x, "source") for x in sources]
        infos = [parse_frame(b) for b in bufs]
        for i in infos:
            if i.descriptor.Dictionary is not None:
                raise NotImplementedException("Predefined dictionaries feature is not implemented")
        # header and block checksums: one launch
        to_hash = [np.frombuffer(i.header, np.uint8) for i in infos]
        for b, i in zip(bufs, infos):
            if i.descriptor.BlockChecksum:
                to_hash += [b[o:o + (l & 0x7FFFFFFF)] for o, l in zip(i.block_off, i.block_len)]
        hashes = xxh32_many(to_hash, ctx)
        k = len(infos)
        for f, i in enumerate(infos):
            if ((int(hashes[f]) >> 8) & 0xFF) != i.header_checksum:
                raise InvalidDataException("Invalid LZ4 frame header checksum")
            if i.descriptor.BlockChecksum:
                for want in i.block_checksum:
                    if int(hashes[k]) != want:
                        raise InvalidDataException("Invalid block checksum")
                    k += 1
        outs = LZ4Frame._decode_streams(bufs, infos, ctx)
        with_sum = [f for f, i in enumerate(infos) if i.descriptor.ContentChecksum]
        if with_sum:
            got = xxh32_many([outs[f] for f in with_sum], ctx)
            for f, g in zip(with_sum, got):
                if int(g) != infos[f].content_checksum:
                    raise InvalidDataException("Invalid content checksum")
        for f, i in enumerate(infos):
            if i.descriptor.ContentLength is not None and i.descriptor.ContentLength != outs[f].size:
                raise InvalidDataException("Content length does not match the frame header")
        return [o.tobytes() for o in outs]

    # What a frame's metadata may make the decoder allocate.  The reference's reader holds ONE block-sized buffer at a time
    # (LZ4FrameReader.async.cs:108-136); a batch decoder sizes an arena up front, so the size must come from what the input
    # can actually produce, not from what the header claims: an LZ4 block never decodes to more than 255 bytes per input
    # byte (a match costs at least 3 bytes + 1 per 255 bytes of length), and launches are cut at an arena budget.
    ARENA_BUDGET = 1 << 30

    @staticmethod
    def _block_cap(block_size: int, stored: int) -> int:
        return int(min(block_size, 255 * stored + 32))

    @staticmethod
    def _decode_streams(bufs, infos, ctx) -> List[np.ndarray]:
        """frames of independent blocks: all their blocks as batches (LZ4BlockDecoder per block, parallel) of at most
        ARENA_BUDGET bytes of output slots each; frames of chained blocks: one in-order stream per frame (LZ4ChainDecoder,
        k4lz4_decode_chain_batch)"""
        res: List[Optional[np.ndarray]] = [None] * len(infos)
        indep = [f for f, i in enumerate(infos) if not i.descriptor.Chaining]
        chain = [f for f, i in enumerate(infos) if i.descriptor.Chaining]
        if indep:
            blocks, caps, where = [], [], []
            for f in indep:
                i, b = infos[f], bufs[f]
                for k, (o, l) in enumerate(zip(i.block_off, i.block_len)):
                    if not (l & 0x80000000):
                        blocks.append(b[o:o + l])
                        caps.append(LZ4Frame._block_cap(i.descriptor.BlockSize, l))
                        where.append((f, k))
            decoded = {}
            lo = 0
            while lo < len(blocks):
                hi, room = lo, LZ4Frame.ARENA_BUDGET
                while hi < len(blocks) and (hi == lo or caps[hi] <= room):
                    room -= caps[hi]
                    hi += 1
                src, soff, slen = pack_blocks(blocks[lo:hi])
                cap = np.array(caps[lo:hi], np.int32)
                dst, doff = make_arena(cap)
                out = LZ4Codec.DecodeBatchPacked(src, soff, slen, dst, doff, cap, ctx=ctx)
                for (f, k), n, o in zip(where[lo:hi], out, doff):
                    if n < 0:
                        raise InvalidDataException("LZ4 block does not decode")      # LZ4BlockDecoder.cs:50-52
                    decoded[(f, k)] = dst[int(o):int(o) + int(n)].copy()             # (the arena goes away with this launch)
                lo = hi
            for f in indep:
                i, b = infos[f], bufs[f]
                parts = [decoded[(f, k)] if not (l & 0x80000000) else b[o:o + (l & 0x7FFFFFFF)]
                         for k, (o, l) in enumerate(zip(i.block_off, i.block_len))]
                res[f] = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        if chain:
            cb = [bufs[f] for f in chain]
            ci = [infos[f] for f in chain]
            nf = len(ci)
            src, foff, _ = pack_blocks(cb)
            blk_off, blk_len, first, nblk = [], [], np.zeros(nf, np.uint64), np.zeros(nf, np.uint32)
            for f, i in enumerate(ci):
                first[f] = len(blk_off)
                nblk[f] = len(i.block_off)
                blk_off += [int(foff[f]) + o for o in i.block_off]
                blk_len += i.block_len
            bsize = np.array([i.descriptor.BlockSize for i in ci], np.int32)
            chained = np.ones(nf, np.uint8)
            produced = [sum(LZ4Frame._block_cap(i.descriptor.BlockSize, l & 0x7FFFFFFF) if not (l & 0x80000000) else (l & 0x7FFFFFFF)
                            for l in i.block_len) for i in ci]          # what the blocks can produce, not what the header claims
            caps = np.array([p if i.descriptor.ContentLength is None else min(i.descriptor.ContentLength, p)
                             for i, p in zip(ci, produced)], np.uint64)
            doff = np.zeros(nf, np.uint64)
            if nf > 1:
                doff[1:] = np.cumsum(caps[:-1])
            dst = np.zeros(max(int(caps.sum()), 1), np.uint8)
            out = np.zeros(nf, np.int64)
            bo = np.array(blk_off if blk_off else [0], np.uint64)
            bl = np.array(blk_len if blk_len else [0], np.uint32)
            ctx.check(ctx.lib.k4lz4_decode_chain_batch(ctx.handle, src.ctypes.data, bo.ctypes.data, bl.ctypes.data, len(blk_off),
                                                       first.ctypes.data, nblk.ctypes.data, bsize.ctypes.data, chained.ctypes.data,
                                                       dst.ctypes.data, doff.ctypes.data, caps.ctypes.data, out.ctypes.data, nf))
            for k, f in enumerate(chain):
                if out[k] < 0:
                    raise InvalidDataException("LZ4 block does not decode" if out[k] == -6
                                               else "Decoded frame does not fit its declared size")
                res[f] = dst[int(doff[k]):int(doff[k]) + int(out[k])]
        return res


def encode_frames_device(dc, data, off: np.ndarray, length: np.ndarray, settings: Optional[LZ4EncoderSettings] = None):
    """LZ4Frame.EncodeBatch on HBM-resident contents, nothing leaves the device: `data` is a uint8 torch tensor holding
    the contents, content f = data[off[f] : off[f]+length[f]] (off / length: host arrays -- the block split is host index
    work).  Returns (frames, frame_off, frame_len): frame f = frames[frame_off[f] : frame_off[f] + frame_len[f]], with
    frame_off a host array of (worst-case spaced) positions and frame_len a device tensor.  Asynchronous on the current
    torch stream; `dc` is a device.DeviceCodec."""
    import ctypes as C
    import torch
    from ._native import FLAG_ALLOW_COPY
    from .device import DeviceBatch, _dp
    s = settings or LZ4EncoderSettings()
    if s.ChainBlocks:
        raise NotImplementedException("chained blocks are encoded serially: use ChainBlocks=False")
    bs = int(s.BlockSize)
    max_block_size_code(bs)
    off = np.asarray(off, dtype=np.int64)
    length = np.asarray(length, dtype=np.int64)
    nf = len(off)
    nblk = (length + bs - 1) // bs
    first = np.concatenate(([0], np.cumsum(nblk)))[:-1]
    nb = int(nblk.sum())
    owner = np.repeat(np.arange(nf), nblk)
    k_in = np.arange(nb) - first[owner]
    boff = off[owner] + k_in * bs
    blen = np.minimum(bs, length[owner] - k_in * bs).astype(np.int32)
    bound = LZ4Codec.MaximumOutputSize(bs)
    dev = dc.device
    src = DeviceBatch(data, torch.from_numpy(boff).to(dev), torch.from_numpy(blen).to(dev))
    arena = DeviceBatch.empty_slots(np.full(nb, bound, np.int64), dev)
    out_len = dc.encode(src, arena, level=s.CompressionLevel, flags=FLAG_ALLOW_COPY) if nb else torch.zeros(0, dtype=torch.int32, device=dev)
    stored = out_len.abs().to(torch.int64)
    # header bytes (host: two to ten bytes per frame) and every XXH32 of the batch
    hdrs = [frame_header(LZ4Descriptor(int(length[f]) if s.ContentLength is not None else None, s.ContentChecksum, False,
                                       s.BlockChecksum, None, bs)) for f in range(nf)]
    hl = len(hdrs[0]) if nf else 2
    hdr_h = np.zeros((max(nf, 1), 16), np.uint8)
    for f, h in enumerate(hdrs):
        hdr_h[f, :hl] = np.frombuffer(h, np.uint8)
    hdr_d = torch.from_numpy(hdr_h.reshape(-1)).to(dev)
    hdr_len = torch.full((max(nf, 1),), hl, dtype=torch.int64, device=dev)
    hdr_sum = dc.xxh32(hdr_d, torch.arange(max(nf, 1), dtype=torch.int64, device=dev) * 16, hdr_len)
    blk_sum = dc.xxh32(arena.data, arena.off, stored) if (s.BlockChecksum and nb) else None
    con_sum = dc.xxh32(data, torch.from_numpy(off).to(dev), torch.from_numpy(length).to(dev)) if (s.ContentChecksum and nf) else None
    # layout: frames sit at worst-case spaced bases (known without a sync), records are packed inside each frame
    head = 4 + hl + 1
    per_block = 4 + bound + (4 if s.BlockChecksum else 0)
    frame_cap = head + nblk * per_block + 8
    frame_off = np.concatenate(([0], np.cumsum(frame_cap)))[:-1].astype(np.int64)
    rec_size = stored + (8 if s.BlockChecksum else 4)
    excl = torch.cumsum(rec_size, 0) - rec_size if nb else rec_size
    owner_d = torch.from_numpy(owner).to(dev)
    first_d = torch.from_numpy(np.minimum(first, max(nb - 1, 0))).to(dev)
    base_d = torch.from_numpy(frame_off).to(dev)
    start_of_frame = excl[first_d] if nb else torch.zeros(nf, dtype=torch.int64, device=dev)
    rec_off = (base_d[owner_d] + head + excl - start_of_frame[owner_d]) if nb else torch.zeros(0, dtype=torch.int64, device=dev)
    total = torch.zeros(nf, dtype=torch.int64, device=dev)
    if nb:
        total.index_add_(0, owner_d, rec_size)
    tail_off = base_d + head + total
    frames = torch.empty(int(frame_cap.sum()) + 64, dtype=torch.uint8, device=dev)
    frame_len = torch.zeros(max(nf, 1), dtype=torch.int64, device=dev)
    hdr_len32 = hdr_len.to(torch.int32)
    rc = dc.lib.k4lz4_frame_assemble_device(dc.ctx.handle, _dp(arena.data), _dp(arena.off), _dp(out_len), _dp(blk_sum), _dp(rec_off), nb,
                                            _dp(hdr_d), _dp(hdr_len32), _dp(hdr_sum), _dp(base_d), _dp(tail_off), _dp(con_sum), _dp(frames),
                                            _dp(frame_len), nf, C.c_void_p(dc._stream()))
    dc.ctx.check(rc)
    return frames, frame_off, frame_len[:nf]
