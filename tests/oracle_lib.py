"""ctypes bindings for the CPU oracle (oracle/libk4lz4_oracle.so) and, when present, the
system liblz4.so.1 second opinion.  Test infrastructure only -- never imported by the
product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libk4lz4_oracle.so")

_u8p = C.POINTER(C.c_uint8)


def _ptr(a: np.ndarray):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u8p)


def build_oracle(force: bool = False) -> str:
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("k4lz4_oracle.c", "k4lz4_oracle_hc.c")]
    stale = (not os.path.exists(ORACLE_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "-B", "libk4lz4_oracle.so"])
    return ORACLE_SO


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        L = self.lib
        L.k4o_compress_bound.argtypes = [C.c_int]
        L.k4o_compress_fast.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
        L.k4o_compress_hc.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
        L.k4o_decompress_safe.argtypes = [_u8p, _u8p, C.c_int, C.c_int]
        L.k4o_decompress_safe_partial.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
        L.k4o_decompress_safe_using_dict.argtypes = [_u8p, _u8p, C.c_int, C.c_int, _u8p, C.c_int]
        L.k4o_codec_encode.argtypes = [_u8p, C.c_int, _u8p, C.c_int, C.c_int]
        L.k4o_codec_decode.argtypes = [_u8p, C.c_int, _u8p, C.c_int]
        L.k4o_pickle_bound.argtypes = [C.c_int]
        L.k4o_pickle.argtypes = [_u8p, C.c_int, _u8p, _u8p, C.c_int, C.c_int]
        L.k4o_unpickle_header.argtypes = [_u8p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.k4o_unpickle.argtypes = [_u8p, C.c_int, _u8p, C.c_int]
        L.k4o_adler32.argtypes = [_u8p, C.c_int64]
        L.k4o_adler32.restype = C.c_uint32
        batch = [_u8p, C.c_void_p, C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.k4o_encode_batch.argtypes = batch + [C.c_int, C.c_int]
        L.k4o_decode_batch.argtypes = batch + [C.c_int]

    # ---- raw engine entry points (LLxx-level returns) --------------------------------------
    def compress_bound(self, n: int) -> int:
        return self.lib.k4o_compress_bound(n)

    def compress_fast(self, src: np.ndarray, cap: int | None = None, accel: int = 1):
        """returns (ret, dst array of length cap)"""
        src = np.ascontiguousarray(src, dtype=np.uint8)
        if cap is None:
            cap = self.compress_bound(src.size)
        dst = np.full(max(cap, 1), 0xCD, dtype=np.uint8)
        ret = self.lib.k4o_compress_fast(_ptr(src if src.size else np.zeros(1, np.uint8)), _ptr(dst), src.size, cap, accel)
        return ret, dst[:cap]

    def compress_hc(self, src: np.ndarray, level: int, cap: int | None = None):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        if cap is None:
            cap = self.compress_bound(src.size)
        dst = np.full(max(cap, 1), 0xCD, dtype=np.uint8)
        ret = self.lib.k4o_compress_hc(_ptr(src if src.size else np.zeros(1, np.uint8)), _ptr(dst), src.size, cap, level)
        return ret, dst[:cap]

    def compress_fast_x32(self, src: np.ndarray, cap: int | None = None, accel: int = 1):
        """LL32.LZ4_compress_fast as LZ4Codec.Enforce32 runs it in a 64-bit process (pinned to oracle/_ref's LL32)"""
        src = np.ascontiguousarray(src, dtype=np.uint8)
        if cap is None:
            cap = self.compress_bound(src.size)
        dst = np.full(max(cap, 1), 0xCD, dtype=np.uint8)
        self.lib.k4o_compress_fast_x32.argtypes = self.lib.k4o_compress_fast.argtypes
        ret = self.lib.k4o_compress_fast_x32(_ptr(src if src.size else np.zeros(1, np.uint8)), _ptr(dst), src.size, cap, accel)
        return ret, dst[:cap]

    def decompress_safe(self, src: np.ndarray, cap: int, fill: int = 0xCD):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dst = np.full(max(cap, 1), fill, dtype=np.uint8)
        ret = self.lib.k4o_decompress_safe(_ptr(src if src.size else np.zeros(1, np.uint8)), _ptr(dst), src.size, cap)
        return ret, dst[:cap]

    def decompress_partial(self, src: np.ndarray, target: int, cap: int):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dst = np.full(max(cap, 1), 0xCD, dtype=np.uint8)
        ret = self.lib.k4o_decompress_safe_partial(_ptr(src), _ptr(dst), src.size, target, cap)
        return ret, dst[:cap]

    def decompress_using_dict(self, src: np.ndarray, cap: int, dictionary: np.ndarray):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dictionary = np.ascontiguousarray(dictionary, dtype=np.uint8)
        dst = np.full(max(cap, 1), 0xCD, dtype=np.uint8)
        ret = self.lib.k4o_decompress_safe_using_dict(_ptr(src), _ptr(dst), src.size, cap, _ptr(dictionary), dictionary.size)
        return ret, dst[:cap]

    def decompress_using_prefix_dict(self, src: np.ndarray, cap: int, dictionary: np.ndarray):
        """dictionary placed immediately before the output (the reference's withPrefix modes)"""
        src = np.ascontiguousarray(src, dtype=np.uint8)
        buf = np.full(dictionary.size + max(cap, 1), 0xCD, dtype=np.uint8)
        buf[:dictionary.size] = dictionary
        dptr = C.cast(buf.ctypes.data, _u8p)
        optr = C.cast(buf.ctypes.data + dictionary.size, _u8p)
        ret = self.lib.k4o_decompress_safe_using_dict(_ptr(src), optr, src.size, cap, dptr, dictionary.size)
        return ret, buf[dictionary.size:dictionary.size + cap]

    # ---- LZ4Codec / LZ4Pickler level --------------------------------------------------------
    def encode(self, src: np.ndarray, level: int = 0, cap: int | None = None) -> bytes | None:
        src = np.ascontiguousarray(src, dtype=np.uint8)
        if cap is None:
            cap = self.compress_bound(src.size)
        dst = np.empty(max(cap, 1), dtype=np.uint8)
        n = self.lib.k4o_codec_encode(_ptr(src if src.size else np.zeros(1, np.uint8)), src.size, _ptr(dst), cap, level)
        return None if n < 0 else dst[:n].tobytes()

    def decode(self, src, cap: int) -> bytes | None:
        src = np.frombuffer(bytes(src), dtype=np.uint8) if not isinstance(src, np.ndarray) else src
        dst = np.empty(max(cap, 1), dtype=np.uint8)
        n = self.lib.k4o_codec_decode(_ptr(src if src.size else np.zeros(1, np.uint8)), src.size, _ptr(dst), cap)
        return None if n < 0 else dst[:n].tobytes()

    def pickle(self, src: np.ndarray, level: int = 0, writer_mode: int = 0) -> bytes:
        src = np.ascontiguousarray(src, dtype=np.uint8)
        if src.size == 0:
            return b""
        dst = np.empty(self.lib.k4o_pickle_bound(src.size), dtype=np.uint8)
        scratch = np.empty(max(src.size, 1024), dtype=np.uint8)
        n = self.lib.k4o_pickle(_ptr(src), src.size, _ptr(dst), _ptr(scratch), level, writer_mode)
        return dst[:n].tobytes()

    def pickle_header(self, src_len: int, enc_len: int, writer_mode: int = 0) -> bytes:
        """the V0 header alone for a block of src_len bytes whose LZ4 block is enc_len bytes"""
        self.lib.k4o_pickle_header.argtypes = [C.c_int, C.c_int, C.c_int, _u8p]
        out = np.zeros(8, dtype=np.uint8)
        n = self.lib.k4o_pickle_header(src_len, enc_len, writer_mode, _ptr(out))
        return out[:n].tobytes()

    def unpickle_header(self, src: bytes):
        a = np.frombuffer(src, dtype=np.uint8)
        off, rl, comp = C.c_int(), C.c_int(), C.c_int()
        rc = self.lib.k4o_unpickle_header(_ptr(a if a.size else np.zeros(1, np.uint8)), a.size, C.byref(off), C.byref(rl), C.byref(comp))
        return rc, off.value, rl.value, comp.value

    def unpickle(self, src: bytes) -> bytes | None:
        """None when the reference would throw InvalidDataException."""
        if len(src) == 0:
            return b""
        rc, off, rl, comp = self.unpickle_header(src)
        if rc < 0 or rl < 0:
            return None
        a = np.frombuffer(src, dtype=np.uint8)
        dst = np.empty(max(rl, 1), dtype=np.uint8)
        n = self.lib.k4o_unpickle(_ptr(a), a.size, _ptr(dst), rl)
        return None if n < 0 else dst[:rl].tobytes()

    def count_sequences(self, src: np.ndarray) -> int:
        src = np.ascontiguousarray(src, dtype=np.uint8)
        self.lib.k4o_count_sequences.restype = C.c_int64
        self.lib.k4o_count_sequences.argtypes = [_u8p, C.c_int]
        return int(self.lib.k4o_count_sequences(_ptr(src if src.size else np.zeros(1, np.uint8)), src.size))

    def adler32(self, data) -> int:
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
        return int(self.lib.k4o_adler32(_ptr(a if a.size else np.zeros(1, np.uint8)), a.size))

    # ---- batch (threaded) -------------------------------------------------------------------
    def encode_batch(self, src, src_off, src_len, dst, dst_off, dst_cap, level=0, threads=1):
        out = np.empty(len(src_len), dtype=np.int32)
        rc = self.lib.k4o_encode_batch(_ptr(src), src_off.ctypes.data, src_len.ctypes.data, _ptr(dst),
                                       dst_off.ctypes.data, dst_cap.ctypes.data, out.ctypes.data,
                                       len(src_len), level, threads)
        assert rc == 0
        return out

    def decode_batch(self, src, src_off, src_len, dst, dst_off, dst_cap, threads=1):
        out = np.empty(len(src_len), dtype=np.int32)
        rc = self.lib.k4o_decode_batch(_ptr(src), src_off.ctypes.data, src_len.ctypes.data, _ptr(dst),
                                       dst_off.ctypes.data, dst_cap.ctypes.data, out.ctypes.data,
                                       len(src_len), threads)
        assert rc == 0
        return out


class SystemLZ4:
    """liblz4.so.1 (1.9.3 in this image) via dlopen -- independent cross-check, not the authority
    (SURVEY.md 8c).  `available` is False when the library is missing."""

    def __init__(self):
        self.lib = None
        for name in ("liblz4.so.1", "/lib/x86_64-linux-gnu/liblz4.so.1"):
            try:
                self.lib = C.CDLL(name)
                break
            except OSError:
                continue
        self.available = self.lib is not None
        if not self.available:
            return
        L = self.lib
        L.LZ4_versionNumber.restype = C.c_int
        L.LZ4_compress_fast.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
        L.LZ4_compress_HC.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
        L.LZ4_decompress_safe.argtypes = [_u8p, _u8p, C.c_int, C.c_int]
        L.LZ4_decompress_safe_partial.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
        L.LZ4_decompress_safe_usingDict.argtypes = [_u8p, _u8p, C.c_int, C.c_int, _u8p, C.c_int]
        L.LZ4_createStream.restype = C.c_void_p
        L.LZ4_freeStream.argtypes = [C.c_void_p]
        L.LZ4_loadDict.argtypes = [C.c_void_p, _u8p, C.c_int]
        L.LZ4_compress_fast_continue.argtypes = [C.c_void_p, _u8p, _u8p, C.c_int, C.c_int, C.c_int]
        self.version = L.LZ4_versionNumber()

    def compress_with_dict(self, src: np.ndarray, dictionary: np.ndarray) -> np.ndarray:
        """LZ4_loadDict + LZ4_compress_fast_continue: a block whose matches may reach into the dictionary
        (what the reference's chain encoder emits for every block after the first)"""
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dictionary = np.ascontiguousarray(dictionary, dtype=np.uint8)
        cap = src.size + src.size // 255 + 16
        dst = np.zeros(cap, np.uint8)
        st = self.lib.LZ4_createStream()
        try:
            self.lib.LZ4_loadDict(st, _ptr(dictionary), dictionary.size)
            n = self.lib.LZ4_compress_fast_continue(st, _ptr(src), _ptr(dst), src.size, cap, 1)
        finally:
            self.lib.LZ4_freeStream(st)
        assert n > 0
        return dst[:n].copy()

    def compress_fast(self, src: np.ndarray, cap: int, accel: int = 1):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dst = np.full(max(cap, 1), 0xCD, dtype=np.uint8)
        ret = self.lib.LZ4_compress_fast(_ptr(src if src.size else np.zeros(1, np.uint8)), _ptr(dst), src.size, cap, accel)
        return ret, dst[:cap]

    def compress_hc(self, src: np.ndarray, cap: int, level: int):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dst = np.full(max(cap, 1), 0xCD, dtype=np.uint8)
        ret = self.lib.LZ4_compress_HC(_ptr(src if src.size else np.zeros(1, np.uint8)), _ptr(dst), src.size, cap, level)
        return ret, dst[:cap]

    def decompress_safe(self, src: np.ndarray, cap: int):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dst = np.full(max(cap, 1), 0xCD, dtype=np.uint8)
        ret = self.lib.LZ4_decompress_safe(_ptr(src if src.size else np.zeros(1, np.uint8)), _ptr(dst), src.size, cap)
        return ret, dst[:cap]

    def decompress_using_dict(self, src, cap, dictionary):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dictionary = np.ascontiguousarray(dictionary, dtype=np.uint8)
        dst = np.full(max(cap, 1), 0xCD, dtype=np.uint8)
        ret = self.lib.LZ4_decompress_safe_usingDict(_ptr(src), _ptr(dst), src.size, cap, _ptr(dictionary), dictionary.size)
        return ret, dst[:cap]


def _frame_api(lib):
    lib.k4o_xxh32.restype = C.c_uint32
    lib.k4o_xxh32.argtypes = [_u8p, C.c_int64, C.c_uint32]
    lib.k4o_frame_bound.restype = C.c_int64
    lib.k4o_frame_bound.argtypes = [C.c_int64, C.c_int]
    lib.k4o_frame_encode.restype = C.c_int64
    lib.k4o_frame_encode.argtypes = [_u8p, C.c_int64, _u8p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _u8p]
    lib.k4o_frame_decode.restype = C.c_int64
    lib.k4o_frame_decode.argtypes = [_u8p, C.c_int64, _u8p, C.c_int64, C.POINTER(C.c_int64)]


class FrameOracle:
    """frame layer of the oracle (oracle/k4lz4_oracle_frame.c): XXH32, frame writer for independent
    blocks, frame reader (independent and chained blocks)"""

    def __init__(self, oracle: "Oracle"):
        self.o = oracle
        self.lib = oracle.lib
        _frame_api(self.lib)

    def xxh32(self, data, seed: int = 0) -> int:
        a = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        return int(self.lib.k4o_xxh32(_ptr(a if a.size else np.zeros(1, np.uint8)), a.size, seed))

    def frame_encode(self, data: np.ndarray, block_size=65536, level=0, block_checksum=False, content_checksum=False) -> bytes:
        data = np.ascontiguousarray(data, dtype=np.uint8)
        cap = int(self.lib.k4o_frame_bound(data.size, block_size))
        dst = np.zeros(cap, np.uint8)
        scratch = np.zeros(self.o.compress_bound(block_size), np.uint8)
        n = self.lib.k4o_frame_encode(_ptr(data if data.size else np.zeros(1, np.uint8)), data.size, _ptr(dst), cap, block_size, level,
                                      int(block_checksum), int(content_checksum), _ptr(scratch))
        assert n > 0, n
        return dst[:n].tobytes()

    def frame_decode(self, frame, cap: int):
        f = np.frombuffer(bytes(frame), np.uint8)
        dst = np.full(max(cap, 1), 0xCD, np.uint8)
        used = C.c_int64(0)
        n = self.lib.k4o_frame_decode(_ptr(f), f.size, _ptr(dst), cap, C.byref(used))
        return int(n), dst[:max(n, 0)].tobytes(), int(used.value)


# ---- oracle/_ref: the reference itself, compiled here (oracle/make_ref.py) --------------------------------------------
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
REFERENCE_PRESENT = os.path.isdir("/root/reference/src/K4os.Compression.LZ4")


def build_ref() -> str | None:
    """(re)builds oracle/_ref where /root/reference is present; elsewhere (the GPU box) returns the prebuilt library that
    travelled with the snapshot, or None when there is none"""
    so = os.path.join(REF_DIR, "libk4ref.so")
    if REFERENCE_PRESENT:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "ref"])
    return so if os.path.exists(so) else None


class RefEngine:
    """LL64 / LL32 of the reference, respelled by oracle/make_ref.py and compiled by g++ (oracle/_ref/libk4ref*.so).
    Same call shapes as `Oracle`'s raw entry points.  variant: "" (NET5_0_OR_GREATER), "_net462", "_debug" (Assert()s on)."""

    def __init__(self, variant: str = ""):
        so = build_ref()
        if so is None:
            raise FileNotFoundError("oracle/_ref/libk4ref.so (needs /root/reference at build time)")
        so = so.replace("libk4ref.so", f"libk4ref{variant}.so")
        if variant == "_debug" and REFERENCE_PRESENT:
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "_ref/libk4ref_debug.so"])
        self.lib = L = C.CDLL(so)
        enc = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
        for f in ("k4ref_compress_fast", "k4ref_compress_fast_x32", "k4ref_compress_hc", "k4ref_compress_hc_x32"):
            getattr(L, f).argtypes = enc
        dec = [_u8p, _u8p, C.c_int, C.c_int]
        L.k4ref_decompress_safe.argtypes = dec
        L.k4ref_decompress_safe_x32.argtypes = dec
        L.k4ref_decompress_safe_partial.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
        L.k4ref_decompress_safe_using_dict.argtypes = [_u8p, _u8p, C.c_int, C.c_int, _u8p, C.c_int]
        L.k4ref_compress_bound.argtypes = [C.c_int]
        L.k4ref_inputs_sha256.restype = C.c_char_p
        self.inputs_sha256 = L.k4ref_inputs_sha256().decode()

    def compress_bound(self, n):
        return self.lib.k4ref_compress_bound(n)

    def _enc(self, fn, src, cap, arg):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        if cap is None:
            cap = self.compress_bound(src.size)
        dst = np.full(max(cap, 1), 0xCD, dtype=np.uint8)
        ret = fn(_ptr(src if src.size else np.zeros(1, np.uint8)), _ptr(dst), src.size, cap, arg)
        return ret, dst[:cap]

    def compress_fast(self, src, cap=None, accel=1):
        return self._enc(self.lib.k4ref_compress_fast, src, cap, accel)

    def compress_fast_x32(self, src, cap=None, accel=1):
        return self._enc(self.lib.k4ref_compress_fast_x32, src, cap, accel)

    def compress_hc(self, src, level, cap=None):
        return self._enc(self.lib.k4ref_compress_hc, src, cap, level)

    def compress_hc_x32(self, src, level, cap=None):
        return self._enc(self.lib.k4ref_compress_hc_x32, src, cap, level)

    def decompress_safe(self, src, cap, fill=0xCD, x32=False):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dst = np.full(max(cap, 1), fill, dtype=np.uint8)
        fn = self.lib.k4ref_decompress_safe_x32 if x32 else self.lib.k4ref_decompress_safe
        ret = fn(_ptr(src if src.size else np.zeros(1, np.uint8)), _ptr(dst), src.size, cap)
        return ret, dst[:cap]

    def decompress_partial(self, src, target, cap):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dst = np.full(max(cap, 1), 0xCD, dtype=np.uint8)
        ret = self.lib.k4ref_decompress_safe_partial(_ptr(src), _ptr(dst), src.size, target, cap)
        return ret, dst[:cap]

    # ---- LZ4Pickler's own header helpers (LZ4Pickler.pickle.cs:161-229, LZ4Pickler.unpickle.cs:131-148), no block encoded
    def pickle_effective_size_of(self, value):
        return self.lib.k4ref_pickle_effective_size_of(C.c_int(value))

    def pickle_encode_size_of(self, size):
        return self.lib.k4ref_pickle_encode_size_of(C.c_int(size))

    def pickle_header_byte_v0(self, size_of_diff):
        return self.lib.k4ref_pickle_header_byte_v0(C.c_int(size_of_diff))

    def pickle_header(self, src_len, enc_len, writer_mode=0, version=0, cap=8):
        """bytes of the header the reference writes, or the negated exception kind (1 Argument, 2 InvalidData, 3 range, 4 Debug.Assert)"""
        self.lib.k4ref_pickle_header.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        n = self.lib.k4ref_pickle_header(writer_mode, version, src_len, enc_len, _ptr(out), cap)
        return n if n < 0 else out[:n].tobytes()

    def unpickle_header(self, src: bytes):
        """(rc, DataOffset, ResultLength, IsCompressed, Flags); rc < 0 = the negated exception kind"""
        a = np.frombuffer(src, dtype=np.uint8)
        out = np.zeros(4, dtype=np.int32)
        self.lib.k4ref_unpickle_header.argtypes = [_u8p, C.c_int, C.c_void_p]
        rc = self.lib.k4ref_unpickle_header(_ptr(a if a.size else np.zeros(1, np.uint8)), a.size, out.ctypes.data)
        return (rc,) + tuple(int(v) for v in out)

    def encode_batch(self, src, src_off, src_len, dst, dst_off, dst_cap, level=0, threads=1):
        out = np.empty(len(src_len), dtype=np.int32)
        self.lib.k4ref_encode_batch.argtypes = [_u8p, C.c_void_p, C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int]
        rc = self.lib.k4ref_encode_batch(_ptr(src), src_off.ctypes.data, src_len.ctypes.data, _ptr(dst), dst_off.ctypes.data,
                                         dst_cap.ctypes.data, out.ctypes.data, len(src_len), level, threads)
        assert rc == 0
        return out

    def decode_batch(self, src, src_off, src_len, dst, dst_off, dst_cap, threads=1):
        out = np.empty(len(src_len), dtype=np.int32)
        self.lib.k4ref_decode_batch.argtypes = [_u8p, C.c_void_p, C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        rc = self.lib.k4ref_decode_batch(_ptr(src), src_off.ctypes.data, src_len.ctypes.data, _ptr(dst), dst_off.ctypes.data,
                                         dst_cap.ctypes.data, out.ctypes.data, len(src_len), threads)
        assert rc == 0
        return out

    def decompress_using_dict(self, src, cap, dictionary):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dictionary = np.ascontiguousarray(dictionary, dtype=np.uint8)
        dst = np.full(max(cap, 1), 0xCD, dtype=np.uint8)
        ret = self.lib.k4ref_decompress_safe_using_dict(_ptr(src), _ptr(dst), src.size, cap, _ptr(dictionary), dictionary.size)
        return ret, dst[:cap]
