"""Host-side mirror of the reference's public block API, backed by libk4lz4.so.

Mirrors (same names, argument meaning and error behaviour):
  LZ4Level                         src/K4os.Compression.LZ4/LZ4Level.cs:6-39
  LZ4Codec.MaximumOutputSize       LZ4Codec.cs:30-31
  LZ4Codec.Encode (3 overloads)    LZ4Codec.cs:40-96
  LZ4Codec.Decode (3 overloads)    LZ4Codec.cs:104-115,:179-191,:225-237
  argument validation              Internal/Extensions.cs:37-52 (ArgumentNullException ->
                                   TypeError, ArgumentException -> ValueError)
plus the batch entry points that are the reason a GPU backend exists:
  LZ4Codec.EncodeBatch / DecodeBatch            convenience: sequences of byte strings
  LZ4Codec.EncodeBatchPacked / DecodeBatchPacked packed buffer + offsets + lengths (host numpy)
All arithmetic happens in the HIP kernels; this module only marshals buffers.
"""
from __future__ import annotations

import ctypes as C
import enum
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from ._native import FLAG_RAW_RETURN


class LZ4Level(enum.IntEnum):
    L00_FAST = 0
    L03_HC = 3
    L04_HC = 4
    L05_HC = 5
    L06_HC = 6
    L07_HC = 7
    L08_HC = 8
    L09_HC = 9
    L10_OPT = 10
    L11_OPT = 11
    L12_MAX = 12


def _ro_view(obj, what: str) -> np.ndarray:
    if obj is None:
        raise TypeError(f"{what} is null")            # ArgumentNullException
    if isinstance(obj, np.ndarray):
        if obj.dtype != np.uint8 or not obj.flags["C_CONTIGUOUS"]:
            raise ValueError(f"{what} must be a contiguous uint8 array")
        return obj.reshape(-1)
    return np.frombuffer(obj, dtype=np.uint8)


def _rw_view(obj, what: str) -> np.ndarray:
    if obj is None:
        raise TypeError(f"{what} is null")
    if isinstance(obj, np.ndarray):
        if obj.dtype != np.uint8 or not obj.flags["C_CONTIGUOUS"] or not obj.flags["WRITEABLE"]:
            raise ValueError(f"{what} must be a writable contiguous uint8 array")
        return obj.reshape(-1)
    mv = memoryview(obj)
    if mv.readonly:
        raise ValueError(f"{what} is read-only")
    return np.frombuffer(mv, dtype=np.uint8)


def _validate(buf: np.ndarray, offset: int, length: int, what: str):
    """Internal/Extensions.cs:37-52"""
    if offset < 0 or length < 0 or offset + length > buf.size:
        raise ValueError(f"invalid index/length combination: {what}[{offset}:{offset}+{length}] of {buf.size}")


def _split_args(args, n_expected_short: int, name: str):
    """(source, target) | (source, sOff, sLen, target, tOff, tLen)"""
    if len(args) == n_expected_short:
        return None
    if len(args) == 6:
        return args
    raise TypeError(f"{name}: expected (source, target) or (source, sourceOffset, sourceLength, target, targetOffset, targetLength)")


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


class _LZ4CodecMeta(type):
    """LZ4Codec.Enforce32 (LZ4Codec.cs:14-25): process-wide switch to the reference's 32-bit engine.  In a 64-bit
    process that engine differs from LL64 only in the fast encoder's hash for inputs of 64 KiB and more
    (x32/LL32.tools.cs:141-148, x32/LL32.fast.cs:543-545); the switch is kept in libk4lz4 (k4lz4_set_enforce32)."""

    @property
    def Enforce32(cls) -> bool:
        return bool(_native.load_library().k4lz4_get_enforce32())

    @Enforce32.setter
    def Enforce32(cls, value: bool) -> None:
        _native.load_library().k4lz4_set_enforce32(1 if value else 0)


class LZ4Codec(metaclass=_LZ4CodecMeta):
    """Static class for compressing and decompressing LZ4 blocks (reference LZ4Codec.cs)."""

    Version = 192  # block format of lz4 1.9.2 (LZ4Codec.cs:13)

    @staticmethod
    def RecommendedMinBatch(kind: int = 0, block_bytes: int = 65536, host_GiBs: float = 0.0) -> int:
        """k4lz4_recommended_min_batch: the batch size below which the managed engine on the host is the faster choice
        (kind 0 fast encode, 1 decode, 2 HC encode; INTEGRATION.md "Crossover")"""
        n = _native.load_library().k4lz4_recommended_min_batch(kind, block_bytes, float(host_GiBs))
        if n < 0:
            raise ValueError("kind must be 0, 1 or 2 and block_bytes positive")
        return int(n)

    @staticmethod
    def MaximumOutputSize(length: int) -> int:
        return _native.load_library().k4lz4_compress_bound(int(length))

    # ---- single block -------------------------------------------------------------------------
    @staticmethod
    def Encode(*args, level: LZ4Level = LZ4Level.L00_FAST) -> int:
        """Encode(source, target, level) or Encode(source, sourceOffset, sourceLength, target,
        targetOffset, targetLength, level).  Returns bytes written, 0 for an empty source, or a
        negative value if the target is too small."""
        if len(args) in (3, 7):
            args, level = args[:-1], args[-1]
        long = _split_args(args, 2, "Encode")
        if long is None:
            src, dst = _ro_view(args[0], "source"), _rw_view(args[1], "target")
            so, sl, to, tl = 0, src.size, 0, dst.size
        else:
            src, dst = _ro_view(long[0], "source"), _rw_view(long[3], "target")
            so, sl, to, tl = int(long[1]), int(long[2]), int(long[4]), int(long[5])
            _validate(src, so, sl, "source")
            _validate(dst, to, tl, "target")
        if sl <= 0:
            return 0                                             # LZ4Codec.cs:45-46
        lib = _native.load_library()
        lvl = int(level)
        if lvl < LZ4Level.L03_HC:
            n = lib.k4lz4_compress_fast(_ptr(src) + so, _ptr(dst) + to, sl, tl, 1)
        else:
            n = lib.k4lz4_compress_hc(_ptr(src) + so, _ptr(dst) + to, sl, tl, lvl)
        _raise_if_native_failed(lib)
        return -1 if n <= 0 else n                               # LZ4Codec.cs:51

    @staticmethod
    def Decode(*args) -> int:
        """Decode(source, target) or Decode(source, sourceOffset, sourceLength, target,
        targetOffset, targetLength).  Returns bytes written, 0 for an empty source, or a negative
        value if the target is too small / the block is corrupt."""
        if len(args) == 3:                                       # Decode(source, target, dictionary): LZ4Codec.cs:198-216
            return LZ4Codec._decode_with_dictionary(*args)
        if len(args) == 9:                                       # ... with offsets and lengths: LZ4Codec.cs:250-266
            src, dst = _ro_view(args[0], "source"), _rw_view(args[3], "target")
            so, sl, to, tl = int(args[1]), int(args[2]), int(args[4]), int(args[5])
            _validate(src, so, sl, "source")
            _validate(dst, to, tl, "target")
            if args[6] is None:                                  # dictionary.Validate(..., allowNullIfEmpty: true)
                if int(args[7]) != 0 or int(args[8]) != 0:
                    raise ValueError("dictionary is null but offset / length are not zero")
                dct = np.zeros(0, np.uint8)
            else:
                d = _ro_view(args[6], "dictionary")
                _validate(d, int(args[7]), int(args[8]), "dictionary")
                dct = d[int(args[7]):int(args[7]) + int(args[8])]
            return LZ4Codec._decode_with_dictionary(src[so:so + sl], dst[to:to + tl], dct)
        long = _split_args(args, 2, "Decode")
        if long is None:
            src, dst = _ro_view(args[0], "source"), _rw_view(args[1], "target")
            so, sl, to, tl = 0, src.size, 0, dst.size
        else:
            src, dst = _ro_view(long[0], "source"), _rw_view(long[3], "target")
            so, sl, to, tl = int(long[1]), int(long[2]), int(long[4]), int(long[5])
            _validate(src, so, sl, "source")
            _validate(dst, to, tl, "target")
        if sl <= 0:
            return 0                                             # LZ4Codec.cs:108-109
        lib = _native.load_library()
        scratch = dst if dst.size else np.zeros(1, np.uint8)
        n = lib.k4lz4_decompress_safe(_ptr(src) + so, _ptr(scratch) + to, sl, tl)
        _raise_if_native_failed(lib)
        return -1 if n <= 0 else n                               # LZ4Codec.cs:114

    @staticmethod
    def _decode_with_dictionary(source, target, dictionary) -> int:
        src, dst, dct = _ro_view(source, "source"), _rw_view(target, "target"), _ro_view(dictionary, "dictionary")
        if src.size <= 0:
            return 0                                             # LZ4Codec.cs:149-150
        lib = _native.load_library()
        scratch = dst if dst.size else np.zeros(1, np.uint8)
        dpt = _ptr(dct) if dct.size else None
        n = lib.k4lz4_decompress_safe_using_dict(_ptr(src), _ptr(scratch), src.size, dst.size, dpt, dct.size)
        _raise_if_native_failed(lib)
        return -1 if n <= 0 else n                               # LZ4Codec.cs:156

    @staticmethod
    def PartialDecode(*args) -> int:
        """PartialDecode(source, target) or (source, sourceOffset, sourceLength, target, targetOffset,
        targetLength): decoding stops once the target is full (LZ4Codec.cs:123-173)."""
        long = _split_args(args, 2, "PartialDecode")
        if long is None:
            src, dst = _ro_view(args[0], "source"), _rw_view(args[1], "target")
            so, sl, to, tl = 0, src.size, 0, dst.size
        else:
            src, dst = _ro_view(long[0], "source"), _rw_view(long[3], "target")
            so, sl, to, tl = int(long[1]), int(long[2]), int(long[4]), int(long[5])
            _validate(src, so, sl, "source")
            _validate(dst, to, tl, "target")
        if sl <= 0:
            return 0                                             # LZ4Codec.cs:127-128
        lib = _native.load_library()
        scratch = dst if dst.size else np.zeros(1, np.uint8)
        n = lib.k4lz4_decompress_safe_partial(_ptr(src) + so, _ptr(scratch) + to, sl, tl)
        _raise_if_native_failed(lib)
        return -1 if n <= 0 else n                               # LZ4Codec.cs:133

    # ---- batches --------------------------------------------------------------------------------
    @staticmethod
    def EncodeBatchPacked(src: np.ndarray, src_off: np.ndarray, src_len: np.ndarray, dst: np.ndarray,
                          dst_off: np.ndarray, dst_cap: np.ndarray, level: LZ4Level = LZ4Level.L00_FAST,
                          flags: int = 0, ctx: Optional[_native.Context] = None) -> np.ndarray:
        ctx = ctx or _native.default_context()
        out = np.empty(len(src_len), dtype=np.int32)
        a = _batch_args(src, src_off, src_len, dst, dst_off, dst_cap, out)
        ctx.check(ctx.lib.k4lz4_encode_batch(ctx.handle, *a, int(level), flags))
        return out

    @staticmethod
    def DecodeBatchPacked(src: np.ndarray, src_off: np.ndarray, src_len: np.ndarray, dst: np.ndarray,
                          dst_off: np.ndarray, dst_cap: np.ndarray, flags: int = 0,
                          ctx: Optional[_native.Context] = None) -> np.ndarray:
        ctx = ctx or _native.default_context()
        out = np.empty(len(src_len), dtype=np.int32)
        a = _batch_args(src, src_off, src_len, dst, dst_off, dst_cap, out)
        ctx.check(ctx.lib.k4lz4_decode_batch(ctx.handle, *a, flags))
        return out

    @staticmethod
    def DecodeDictBatchPacked(src: np.ndarray, src_off: np.ndarray, src_len: np.ndarray, dst: np.ndarray,
                              dst_off: np.ndarray, dst_cap: np.ndarray, dictionaries: np.ndarray, dict_off: np.ndarray,
                              dict_len: np.ndarray, flags: int = 0, ctx: Optional[_native.Context] = None) -> np.ndarray:
        """Batched Decode(source, target, dictionary): block i uses dictionaries[dict_off[i]:+dict_len[i]]."""
        ctx = ctx or _native.default_context()
        out = np.empty(len(src_len), dtype=np.int32)
        a = _batch_args(src, src_off, src_len, dst, dst_off, dst_cap, out)
        dct = np.ascontiguousarray(dictionaries, dtype=np.uint8)
        doff = np.ascontiguousarray(dict_off, dtype=np.uint64)
        dlen = np.ascontiguousarray(dict_len, dtype=np.int32)
        if len(doff) != len(src_len) or len(dlen) != len(src_len):
            raise ValueError("dict_off / dict_len must have one entry per block")
        ctx.check(ctx.lib.k4lz4_decode_dict_batch(ctx.handle, *a, flags, _ptr(dct) if dct.size else None,
                                                  doff.ctypes.data, dlen.ctypes.data))
        return out

    @staticmethod
    def EncodeBatch(sources: Sequence, level: LZ4Level = LZ4Level.L00_FAST) -> List[Optional[bytes]]:
        """Each element as LZ4Codec.Encode into a MaximumOutputSize target; None where Encode < 0."""
        blocks = [_ro_view(s, "source") for s in sources]
        src, soff, slen = pack_blocks(blocks)
        caps = np.array([LZ4Codec.MaximumOutputSize(b.size) for b in blocks], dtype=np.int32)
        dst, doff = make_arena(caps)
        out = LZ4Codec.EncodeBatchPacked(src, soff, slen, dst, doff, caps, level)
        return [None if n < 0 else dst[int(o):int(o) + int(n)].tobytes() for n, o in zip(out, doff)]

    @staticmethod
    def DecodeBatch(sources: Sequence, target_lengths: Sequence[int]) -> List[Optional[bytes]]:
        blocks = [_ro_view(s, "source") for s in sources]
        src, soff, slen = pack_blocks(blocks)
        caps = np.asarray(target_lengths, dtype=np.int32)
        dst, doff = make_arena(caps)
        out = LZ4Codec.DecodeBatchPacked(src, soff, slen, dst, doff, caps)
        return [None if n < 0 else dst[int(o):int(o) + int(n)].tobytes() for n, o in zip(out, doff)]


def _raise_if_native_failed(lib):
    _native.check_last_status(lib)


def _batch_args(src, src_off, src_len, dst, dst_off, dst_cap, out):
    n = len(src_len)
    for name, arr, dt in (("src", src, np.uint8), ("dst", dst, np.uint8), ("src_off", src_off, np.uint64),
                          ("dst_off", dst_off, np.uint64), ("src_len", src_len, np.int32),
                          ("dst_cap", dst_cap, np.int32)):
        if not isinstance(arr, np.ndarray) or arr.dtype != dt or not arr.flags["C_CONTIGUOUS"]:
            raise ValueError(f"{name} must be a contiguous numpy array of {np.dtype(dt).name}")
    if not (len(src_off) == len(dst_off) == len(dst_cap) == n):
        raise ValueError("metadata vectors differ in length")
    if n:
        if int((src_off + src_len.clip(min=0).astype(np.uint64)).max()) > src.size:
            raise ValueError("a source block exceeds the source buffer")
        if int((dst_off + dst_cap.clip(min=0).astype(np.uint64)).max()) > dst.size:
            raise ValueError("a target slot exceeds the target buffer")
    return (_ptr(src), _ptr(src_off), _ptr(src_len), _ptr(dst), _ptr(dst_off), _ptr(dst_cap), _ptr(out), n)


def pack_blocks(blocks: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """list of uint8 arrays -> (packed buffer, uint64 offsets, int32 lengths)"""
    lens = np.array([b.size for b in blocks], dtype=np.int32)
    off = np.zeros(len(blocks), dtype=np.uint64)
    if len(blocks) > 1:
        off[1:] = np.cumsum(lens[:-1].astype(np.uint64))
    buf = np.concatenate(blocks) if len(blocks) else np.zeros(0, np.uint8)
    if buf.size == 0:
        buf = np.zeros(1, np.uint8)
    return np.ascontiguousarray(buf, dtype=np.uint8), off, lens


def make_arena(caps: np.ndarray, fill: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
    """target arena with one slot of caps[i] bytes per block -> (buffer, uint64 offsets)"""
    caps = np.asarray(caps, dtype=np.int64).clip(min=0)
    off = np.zeros(len(caps), dtype=np.uint64)
    if len(caps) > 1:
        off[1:] = np.cumsum(caps[:-1]).astype(np.uint64)
    total = int(caps.sum()) if len(caps) else 0
    buf = np.empty(max(total, 1), dtype=np.uint8) if fill is None else np.full(max(total, 1), fill, dtype=np.uint8)
    return buf, off
