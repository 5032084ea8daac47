"""ctypes driver for tests/emu/libk4lz4_emu.so: the product's HIP kernel source compiled by g++
against the host wave emulator (tests/emu/hip/hip_runtime.h).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import glob
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_SO = os.path.join(EMU_DIR, "libk4lz4_emu.so")
_u8p = C.POINTER(C.c_uint8)


def build_emu() -> str:
    srcs = glob.glob(os.path.join(EMU_DIR, "*.cpp")) + glob.glob(os.path.join(EMU_DIR, "hip", "*.h")) + \
        glob.glob(os.path.join(ROOT, "k4os", "compression", "lz4_amd", "csrc", "*.hpp"))
    def stale():
        return not os.path.exists(EMU_SO) or any(os.path.getmtime(s) > os.path.getmtime(EMU_SO) for s in srcs)
    if stale():
        # several processes may want it at once (the two ranks of test_distributed_cpu): one builds, under a lock, into a file of
        # its own that takes the library's name only when it is complete
        import fcntl
        with open(os.path.join(EMU_DIR, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if stale():
                tmp = f"libk4lz4_emu.so.{os.getpid()}.tmp"
                subprocess.check_call(["make", "-C", EMU_DIR, "-s", "-B", f"OUT={tmp}"])
                os.replace(os.path.join(EMU_DIR, tmp), EMU_SO)
    return EMU_SO


def pack(blocks, caps=None, guard=0):
    """blocks: list of uint8 arrays -> (packed, off u64, len i32).  With caps: an output arena
    (filled 0xCD) with `guard` bytes between slots."""
    lens = np.array([b.size for b in blocks], dtype=np.int32)
    off = np.zeros(len(blocks), dtype=np.uint64)
    if len(blocks):
        off[1:] = np.cumsum(lens[:-1].astype(np.uint64))
    buf = np.concatenate([np.asarray(b, np.uint8) for b in blocks]) if len(blocks) else np.zeros(0, np.uint8)
    if buf.size == 0:
        buf = np.zeros(1, np.uint8)
    return np.ascontiguousarray(buf), off, lens


def arena(caps, guard=16):
    caps = np.asarray(caps, dtype=np.int32)
    sizes = np.maximum(caps.astype(np.int64), 0) + guard
    off = np.zeros(len(caps), dtype=np.uint64)
    off[:] = guard + np.concatenate(([0], np.cumsum(sizes[:-1]))) if len(caps) else 0
    total = int(sizes.sum()) + guard + 16
    return np.full(total, 0xCD, dtype=np.uint8), off, caps


class Emu:
    def __init__(self):
        self.lib = C.CDLL(build_emu())
        b = [_u8p, C.c_void_p, C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
        self.lib.k4emu_decode_batch.argtypes = b + [C.c_int, C.c_int]
        self.lib.k4emu_encode_batch.argtypes = b + [C.c_int, C.c_int, C.c_int, C.c_int]
        self.lib.k4emu_encode_gtab_batch.argtypes = b + [C.c_int, C.c_int, C.c_int, C.c_int]
        self.lib.k4emu_encode_more_batch.argtypes = b + [C.c_int, C.c_int, C.c_int, C.c_int]
        self.more = False      # True: encode_batch runs the 28-known-bytes variant of the LDS-table kernel
        self.lib.k4emu_encode_parse_batch.argtypes = b + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        self.lib.k4emu_decode_dict_batch.argtypes = b + [C.c_int, _u8p, C.c_void_p, C.c_void_p, C.c_int]
        self.lib.k4emu_decode_pair_batch.argtypes = b + [C.c_int, _u8p, C.c_void_p, C.c_void_p, C.c_int]
        self.pair = False      # True: decode_batch / decode_dict_batch run the two-waves-per-block kernel
        self.lib.k4emu_xxh32_batch.argtypes = [_u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_uint32, C.c_int]
        self.lib.k4emu_allow_copy.argtypes = b + [C.c_int]
        self.lib.k4emu_decode_chain_batch.argtypes = [_u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _u8p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int]
        self.lib.k4emu_encode_hc_batch.argtypes = b + [C.c_int, C.c_int, C.c_int]
        self.lib.k4emu_order.argtypes = [_u8p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.lib.k4emu_pickle_batch.argtypes = b + [C.c_int, C.c_int, C.c_int]
        self.lib.k4emu_pickle_seg_batch.argtypes = b + [C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_int]
        self.lib.k4emu_unpickle_batch.argtypes = b + [C.c_int, C.c_int]
        self.lib.k4emu_unpickle_pair_batch.argtypes = b + [C.c_int, C.c_int]
        self.lib.k4emu_unpickle_sizes.argtypes = [_u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int]

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(_u8p)

    def decode_batch(self, src, src_off, src_len, dst, dst_off, dst_cap, flags=0, threads=0):
        out = np.full(len(src_len), -12345, dtype=np.int32)
        if self.pair:
            rc = self.lib.k4emu_decode_pair_batch(self._p(src), src_off.ctypes.data, src_len.ctypes.data, self._p(dst),
                                                  dst_off.ctypes.data, dst_cap.ctypes.data, out.ctypes.data, len(src_len),
                                                  flags, None, None, None, threads)
            assert rc == 0
            return out
        rc = self.lib.k4emu_decode_batch(self._p(src), src_off.ctypes.data, src_len.ctypes.data, self._p(dst),
                                         dst_off.ctypes.data, dst_cap.ctypes.data, out.ctypes.data,
                                         len(src_len), flags, threads)
        assert rc == 0
        return out

    def encode_batch(self, src, src_off, src_len, dst, dst_off, dst_cap, level=0, accel=1, flags=0, threads=0, gtab=False, more=False):
        out = np.full(len(src_len), -12345, dtype=np.int32)
        fn = self.lib.k4emu_encode_gtab_batch if gtab else (self.lib.k4emu_encode_more_batch if more or self.more else self.lib.k4emu_encode_batch)
        rc = fn(self._p(src), src_off.ctypes.data, src_len.ctypes.data, self._p(dst),
                                         dst_off.ctypes.data, dst_cap.ctypes.data, out.ctypes.data,
                                         len(src_len), level, accel, flags, threads)
        assert rc == 0
        return out

    def encode_parse_batch(self, src, src_off, src_len, dst, dst_off, dst_cap, accel=1, flags=0, k=1, waves=16, order=None, threads=0, inline_emit=False, queue=False, migrate=False, slot_recs=False):
        """the two-kernel fast encoder (k4lz4_parse.hpp): parse with k sub-windows per round and `waves` blocks per workgroup
        (those beyond 9 keep their table in memory), emit, then the one-kernel encoder for the blocks the parse left alone.
        Returns (outLen, sequences per block -- 0xffffffff where the parse left the block alone)."""
        out = np.full(len(src_len), -12345, dtype=np.int32)
        nseq = np.zeros(len(src_len), dtype=np.uint32)
        rc = self.lib.k4emu_encode_parse_batch(self._p(src), src_off.ctypes.data, src_len.ctypes.data, self._p(dst), dst_off.ctypes.data,
                                               dst_cap.ctypes.data, out.ctypes.data, len(src_len), accel, flags, k + (16 if inline_emit else 0) + (32 if queue else 0) + (64 if migrate else 0) + (128 if slot_recs else 0), waves,
                                               order.ctypes.data if order is not None else None, nseq.ctypes.data, threads)
        assert rc == 0
        return out, nseq

    def pickle_batch(self, src, src_off, src_len, dst, dst_off, dst_cap, level=0, flags=0, threads=0):
        out = np.full(len(src_len), -12345, dtype=np.int32)
        rc = self.lib.k4emu_pickle_batch(self._p(src), src_off.ctypes.data, src_len.ctypes.data, self._p(dst),
                                         dst_off.ctypes.data, dst_cap.ctypes.data, out.ctypes.data,
                                         len(src_len), level, flags, threads)
        assert rc == 0
        return out

    def pickle_seg_batch(self, src, src_off, src_len, dst, dst_off, dst_cap, seg_min, seg_target, seg_warm, flags=0, threads=0):
        """pickles with big messages cut into segments (k4lz4_segments.hpp): returns (lengths, [cut blocks, segments, blocks joined as planned, pieces kept behind a bad boundary, runs resumed by the join, of those stopped at a verified boundary, blocks encoded again whole])"""
        out = np.zeros(src_len.size, np.int32)
        stats = np.zeros(8, np.uint32)
        rc = self.lib.k4emu_pickle_seg_batch(self._p(src), src_off.ctypes.data, src_len.ctypes.data, self._p(dst),
                                             dst_off.ctypes.data, dst_cap.ctypes.data, out.ctypes.data, src_len.size, flags,
                                             seg_min, seg_target, seg_warm, stats.ctypes.data, threads)
        assert rc == 0
        return out, stats

    def unpickle_batch(self, src, src_off, src_len, dst, dst_off, dst_cap, flags=0, threads=0):
        out = np.full(len(src_len), -12345, dtype=np.int32)
        rc = (self.lib.k4emu_unpickle_pair_batch if self.pair else self.lib.k4emu_unpickle_batch)(self._p(src), src_off.ctypes.data, src_len.ctypes.data, self._p(dst),
                                           dst_off.ctypes.data, dst_cap.ctypes.data, out.ctypes.data,
                                           len(src_len), flags, threads)
        assert rc == 0
        return out

    def unpickle_sizes(self, src, src_off, src_len, threads=0):
        out = np.full(len(src_len), -12345, dtype=np.int32)
        rc = self.lib.k4emu_unpickle_sizes(self._p(src), src_off.ctypes.data, src_len.ctypes.data,
                                           out.ctypes.data, len(src_len), threads)
        assert rc == 0
        return out

    def porder(self, src, src_off, src_len, threads=0):
        """k4_pcost_kernel + k4_porder_kernel: (cost bucket per block, dispatch order)"""
        n = len(src_len)
        cost = np.zeros(n, np.uint32); order = np.full(n, 0xFFFFFFFF, np.uint32)
        self.lib.k4emu_porder.argtypes = [_u8p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int]
        rc = self.lib.k4emu_porder(self._p(src), src_off.ctypes.data, src_len.ctypes.data, n, cost.ctypes.data, order.ctypes.data, threads)
        assert rc == 0
        return cost, order

    def order(self, src, src_off, src_len, by_length=0, threads=0):
        n = len(src_len)
        cost = np.zeros(n, np.uint32); hist = np.zeros(128, np.uint32); order = np.full(n, 0xFFFFFFFF, np.uint32)
        rc = self.lib.k4emu_order(self._p(src), src_off.ctypes.data, src_len.ctypes.data, n, by_length,
                                  cost.ctypes.data, hist.ctypes.data, order.ctypes.data, threads)
        assert rc == 0
        return cost, order

    def encode_hc_batch(self, src, src_off, src_len, dst, dst_off, dst_cap, level=3, flags=0, threads=0):
        out = np.full(len(src_len), -12345, dtype=np.int32)
        rc = self.lib.k4emu_encode_hc_batch(self._p(src), src_off.ctypes.data, src_len.ctypes.data, self._p(dst),
                                            dst_off.ctypes.data, dst_cap.ctypes.data, out.ctypes.data,
                                            len(src_len), level, flags, threads)
        assert rc == 0
        return out

    def decode_dict_batch(self, src, src_off, src_len, dst, dst_off, dst_cap, dct, dict_off, dict_len, flags=0, threads=0):
        out = np.full(len(src_len), -12345, dtype=np.int32)
        fn = self.lib.k4emu_decode_pair_batch if self.pair else self.lib.k4emu_decode_dict_batch
        rc = fn(self._p(src), src_off.ctypes.data, src_len.ctypes.data, self._p(dst),
                                              dst_off.ctypes.data, dst_cap.ctypes.data, out.ctypes.data, len(src_len),
                                              flags, self._p(dct), dict_off.ctypes.data, dict_len.ctypes.data, threads)
        assert rc == 0
        return out

    def decode_pair_prof(self, src, src_off, src_len, dst, dst_off, dst_cap, threads=0):
        """the pair kernel's instrumented twin: (outLen, counters[n, 32]) -- parsing wave [0, 16), copying wave [16, 32)"""
        out = np.full(len(src_len), -12345, dtype=np.int32)
        counters = np.zeros((len(src_len), 32), dtype=np.uint64)
        self.lib.k4emu_decode_pair_prof_batch.argtypes = [_u8p, C.c_void_p, C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong,
                                                          C.c_void_p, C.c_int]
        rc = self.lib.k4emu_decode_pair_prof_batch(self._p(src), src_off.ctypes.data, src_len.ctypes.data, self._p(dst), dst_off.ctypes.data,
                                                   dst_cap.ctypes.data, out.ctypes.data, len(src_len), counters.ctypes.data, threads)
        assert rc == 0
        return out, counters

    def lane_copy(self, src, soff, dst, doff, length, mode):
        self.lib.k4emu_lane_copy.argtypes = [_u8p, C.c_void_p, _u8p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
        rc = self.lib.k4emu_lane_copy(self._p(src), soff.ctypes.data, self._p(dst), doff.ctypes.data, length.ctypes.data,
                                      src.size, mode)
        assert rc == 0

    def xxh32_batch(self, data, off, length, seed=0, threads=0):
        out = np.zeros(len(off), dtype=np.uint32)
        rc = self.lib.k4emu_xxh32_batch(self._p(data), off.ctypes.data, length.ctypes.data, out.ctypes.data, len(off), seed, threads)
        assert rc == 0
        return out

    def allow_copy(self, src, src_off, src_len, dst, dst_off, dst_cap, out_len, threads=0):
        rc = self.lib.k4emu_allow_copy(self._p(src), src_off.ctypes.data, src_len.ctypes.data, self._p(dst), dst_off.ctypes.data,
                                       dst_cap.ctypes.data, out_len.ctypes.data, len(src_len), threads)
        assert rc == 0
        return out_len

    def decode_chain_batch(self, src, blk_off, blk_len, first, nblk, block_size, chained, dst, dst_off, dst_cap, threads=0):
        out = np.full(len(first), -12345, dtype=np.int64)
        fn = self.lib.k4emu_decode_chain_pair_batch if self.pair else self.lib.k4emu_decode_chain_batch
        fn.argtypes = self.lib.k4emu_decode_chain_batch.argtypes
        rc = fn(self._p(src), blk_off.ctypes.data, blk_len.ctypes.data, first.ctypes.data,
                                               nblk.ctypes.data, block_size.ctypes.data, chained.ctypes.data, self._p(dst),
                                               dst_off.ctypes.data, dst_cap.ctypes.data, out.ctypes.data, len(first), threads)
        assert rc == 0
        return out
