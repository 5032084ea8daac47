"""Device-resident batches: the roofline path.  Buffers are torch tensors on the GPU (torch is
used for device memory and streams only); the kernels are reached through the *_device entry
points of the C ABI with raw device pointers."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _native
from .codec import LZ4Level


def _dp(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


@dataclass
class DeviceBatch:
    """n independent blocks in one packed device buffer"""
    data: torch.Tensor      # uint8
    off: torch.Tensor       # int64 holding uint64 offsets
    length: torch.Tensor    # int32 (lengths for a source batch, capacities for a target batch)

    @property
    def n(self) -> int:
        return int(self.length.numel())

    @staticmethod
    def from_host(data: np.ndarray, off: np.ndarray, length: np.ndarray, device) -> "DeviceBatch":
        d = torch.from_numpy(np.ascontiguousarray(data)).to(device, non_blocking=False)
        o = torch.from_numpy(np.ascontiguousarray(off).view(np.int64)).to(device)
        ln = torch.from_numpy(np.ascontiguousarray(length, dtype=np.int32)).to(device)
        return DeviceBatch(d, o, ln)

    @staticmethod
    def empty_slots(caps: np.ndarray, device, align: int = 16, fill: Optional[int] = None) -> "DeviceBatch":
        caps = np.asarray(caps, dtype=np.int64).clip(min=0)
        padded = (caps + (align - 1)) // align * align
        off = np.zeros(len(caps), dtype=np.int64)
        if len(caps) > 1:
            off[1:] = np.cumsum(padded[:-1])
        total = int(padded.sum()) + 64
        if fill is None:
            data = torch.empty(total, dtype=torch.uint8, device=device)
        else:
            data = torch.full((total,), fill, dtype=torch.uint8, device=device)
        return DeviceBatch(data, torch.from_numpy(off).to(device), torch.from_numpy(caps.astype(np.int32)).to(device))


class DeviceCodec:
    """Batched LZ4Codec / LZ4Pickler on HBM-resident data; asynchronous on the current torch stream."""

    def __init__(self, device: Optional[int] = None):
        if not torch.cuda.is_available():
            raise _native.NativeLibraryError("no GPU visible: the device path has no CPU fallback")
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.ctx = _native.Context(self.device_index)
        self.lib = self.ctx.lib

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _call(self, fn, src: DeviceBatch, dst: DeviceBatch, out_len: torch.Tensor, *tail):
        assert src.n == dst.n == out_len.numel() and out_len.dtype == torch.int32
        rc = fn(self.ctx.handle, _dp(src.data), _dp(src.off), _dp(src.length), _dp(dst.data), _dp(dst.off),
                _dp(dst.length), _dp(out_len), src.n, *tail, C.c_void_p(self._stream()))
        self.ctx.check(rc)
        return out_len

    def new_out_len(self, n: int) -> torch.Tensor:
        return torch.empty(n, dtype=torch.int32, device=self.device)

    def encode(self, src: DeviceBatch, dst: DeviceBatch, out_len: Optional[torch.Tensor] = None,
               level: LZ4Level = LZ4Level.L00_FAST, flags: int = 0) -> torch.Tensor:
        out_len = self.new_out_len(src.n) if out_len is None else out_len
        return self._call(self.lib.k4lz4_encode_batch_device, src, dst, out_len, int(level), flags)

    def decode(self, src: DeviceBatch, dst: DeviceBatch, out_len: Optional[torch.Tensor] = None,
               flags: int = 0) -> torch.Tensor:
        out_len = self.new_out_len(src.n) if out_len is None else out_len
        return self._call(self.lib.k4lz4_decode_batch_device, src, dst, out_len, flags)

    def pickle(self, src: DeviceBatch, dst: DeviceBatch, out_len: Optional[torch.Tensor] = None,
               level: LZ4Level = LZ4Level.L00_FAST, flags: int = 0) -> torch.Tensor:
        out_len = self.new_out_len(src.n) if out_len is None else out_len
        return self._call(self.lib.k4lz4_pickle_batch_device, src, dst, out_len, int(level), flags)

    def unpickle(self, src: DeviceBatch, dst: DeviceBatch, out_len: Optional[torch.Tensor] = None) -> torch.Tensor:
        out_len = self.new_out_len(src.n) if out_len is None else out_len
        return self._call(self.lib.k4lz4_unpickle_batch_device, src, dst, out_len, 0)

    def unpickle_sizes(self, src: DeviceBatch) -> torch.Tensor:
        out = self.new_out_len(src.n)
        rc = self.lib.k4lz4_unpickle_sizes_device(self.ctx.handle, _dp(src.data), _dp(src.off), _dp(src.length),
                                                  _dp(out), src.n, C.c_void_p(self._stream()))
        self.ctx.check(rc)
        return out

    def xxh32(self, data: torch.Tensor, off: torch.Tensor, length: torch.Tensor, seed: int = 0) -> torch.Tensor:
        """XXH32 of n HBM-resident buffers (off, length: int64/uint64 tensors) -> uint32 digests as an int64 tensor's low half
        (torch has no uint32 arithmetic; the tensor is int32 storage, view it as uint32 on the host)"""
        n = off.numel()
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        rc = self.lib.k4lz4_xxh32_batch_device(self.ctx.handle, _dp(data), _dp(off), _dp(length), _dp(out), n, seed,
                                               C.c_void_p(self._stream()))
        self.ctx.check(rc)
        return out

    def decode_chain(self, src: torch.Tensor, blk_off: torch.Tensor, blk_len: torch.Tensor, first: torch.Tensor, nblk: torch.Tensor,
                     block_size: torch.Tensor, chained: torch.Tensor, dst: torch.Tensor, dst_off: torch.Tensor,
                     dst_cap: torch.Tensor) -> torch.Tensor:
        """block streams decoded in order, one stream per wavefront (k4lz4_decode_chain_batch_device)"""
        n = first.numel()
        out = torch.empty(n, dtype=torch.int64, device=self.device)
        rc = self.lib.k4lz4_decode_chain_batch_device(self.ctx.handle, _dp(src), _dp(blk_off), _dp(blk_len), _dp(first), _dp(nblk),
                                                      _dp(block_size), _dp(chained), _dp(dst), _dp(dst_off), _dp(dst_cap), _dp(out), n,
                                                      C.c_void_p(self._stream()))
        self.ctx.check(rc)
        return out

    def profile(self, decode: bool, src: DeviceBatch, dst: DeviceBatch, out_len: Optional[torch.Tensor] = None):
        """instrumented twin kernels: returns (out_len, counters[n, 16] int64 tensor); decode == 2: the two-waves-per-block
        decoder, counters[n, 32] (parsing wave, copying wave); decode == 4 / 5: the ordinary encode / decode kernels, which
        then only stamp every block's start, end and placement (counters[:, 8:12])"""
        out_len = self.new_out_len(src.n) if out_len is None else out_len
        counters = torch.zeros((src.n, 32 if int(decode) == 2 else 16), dtype=torch.int64, device=self.device)
        rc = self.lib.k4lz4_profile_batch_device(self.ctx.handle, int(decode), _dp(src.data), _dp(src.off),
                                                 _dp(src.length), _dp(dst.data), _dp(dst.off), _dp(dst.length),
                                                 _dp(out_len), src.n, _dp(counters), C.c_void_p(self._stream()))
        self.ctx.check(rc)
        return out_len, counters


class DevicePickleBackend:
    """per-rank work of sharding.sharded_pickle_roundtrip on one GPU: HBM-resident Pickle + Unpickle of a share of messages"""

    def __init__(self, device: int):
        self.dc = DeviceCodec(device)

    def time_alone(self, message: np.ndarray) -> float:
        """Pickle + Unpickle seconds of ONE message with the GPU to itself (the critical path of a ragged batch)"""
        import time
        dc = self.dc
        lens = np.array([message.size], np.int32)
        src = DeviceBatch.from_host(message, np.zeros(1, np.uint64), lens, dc.device)
        env = DeviceBatch.empty_slots(lens.astype(np.int64) + 5, dc.device)
        back = DeviceBatch.empty_slots(lens, dc.device)
        plen, ulen = dc.new_out_len(1), dc.new_out_len(1)
        best = float("inf")
        for _ in range(2):
            torch.cuda.synchronize(dc.device)
            t = time.perf_counter()
            dc.pickle(src, env, plen)
            dc.unpickle(DeviceBatch(env.data, env.off, plen), back, ulen)
            torch.cuda.synchronize(dc.device)
            best = min(best, time.perf_counter() - t)
        return best

    def pickle_unpickle(self, data: np.ndarray, off: np.ndarray, lens: np.ndarray):
        import time
        dc = self.dc
        n = int(lens.size)
        if n == 0:
            return torch.zeros(0, dtype=torch.int32, device=dc.device), 0.0, 0.0, True
        src = DeviceBatch.from_host(data, off, lens, dc.device)
        env = DeviceBatch.empty_slots(lens.astype(np.int64) + 5, dc.device)
        plen = dc.new_out_len(n)
        back = DeviceBatch.empty_slots(lens, dc.device)
        ulen = dc.new_out_len(n)
        dc.pickle(src, env, plen)                      # warm-up (scratch allocation)
        torch.cuda.synchronize(dc.device)
        t = time.perf_counter(); dc.pickle(src, env, plen); torch.cuda.synchronize(dc.device); t_p = time.perf_counter() - t
        psrc = DeviceBatch(env.data, env.off, plen)
        dc.unpickle(psrc, back, ulen)
        torch.cuda.synchronize(dc.device)
        t = time.perf_counter(); dc.unpickle(psrc, back, ulen); torch.cuda.synchronize(dc.device); t_u = time.perf_counter() - t
        ok = bool((ulen == torch.from_numpy(lens).to(dc.device)).all().item())
        if ok:                                         # every byte back, compared on the device
            boff = back.off.cpu().numpy()
            want = torch.from_numpy(np.ascontiguousarray(data)).to(dc.device)
            pos = 0
            if all(int(boff[i]) == int(off[i]) for i in range(0, n, max(1, n // 64))) and int(boff[-1]) == int(off[-1]):
                ok = bool(torch.equal(back.data[:want.numel()], want))
            else:                                      # slots are 16-byte aligned: compare message by message
                for i in range(n):
                    ln = int(lens[i])
                    if not torch.equal(back.data[int(boff[i]):int(boff[i]) + ln], want[int(off[i]):int(off[i]) + ln]):
                        ok = False
                        break
        return plen, t_p, t_u, ok
