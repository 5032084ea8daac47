"""Host-side mirror of LZ4Pickler, backed by libk4lz4.so.

Mirrors:
  LZ4Pickler.Pickle(source[, index, count], level)       LZ4Pickler.pickle.cs:24-106
  LZ4Pickler.Pickle(source, writer, level)               LZ4Pickler.pickle.cs:113-158
  LZ4Pickler.Unpickle(source[, index, count])            LZ4Pickler.unpickle.cs:18-50
  LZ4Pickler.Unpickle(source, writer) / (source, output) LZ4Pickler.unpickle.cs:55-107
  LZ4Pickler.UnpickledSize                               LZ4Pickler.unpickle.cs:83-84
InvalidDataException("Pickle is corrupted: ...") is raised where the reference raises it
(unpickle.cs:160-161).  PickleBatch / UnpickleBatch are the batch forms.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import _native
from ._native import FLAG_PICKLE_WRITER
from .codec import LZ4Level, _ro_view, _rw_view, _validate, _batch_args, pack_blocks, make_arena


class InvalidDataException(Exception):
    """System.IO.InvalidDataException"""


def _corrupted(msg: str) -> InvalidDataException:
    return InvalidDataException(f"Pickle is corrupted: {msg}")


class LZ4Pickler:
    """Self-describing single-block envelope around LZ4Codec (reference LZ4Pickler.*.cs)."""

    # ---- Pickle -----------------------------------------------------------------------------------
    @staticmethod
    def Pickle(source, *args, level: LZ4Level = LZ4Level.L00_FAST):
        """Pickle(source, level) -> bytes | Pickle(source, index, count, level) -> bytes |
        Pickle(source, writer, level) -> None (writer: object with write(bytes-like))."""
        src = _ro_view(source, "source")
        writer = None
        if args and isinstance(args[-1], LZ4Level):
            args, level = args[:-1], args[-1]
        if len(args) == 2:
            index, count = int(args[0]), int(args[1])
            _validate(src, index, count, "source")
            src = src[index:index + count]
        elif len(args) == 1:
            writer = args[0]
            if writer is None:
                raise TypeError("writer is null")                  # pickle.cs:118-119
        elif len(args) != 0:
            raise TypeError("Pickle(source[, index, count | writer][, level])")
        out = LZ4Pickler.PickleBatch([src], level, writer_mode=writer is not None)[0]
        if writer is None:
            return out
        if len(out):
            writer.write(out)
        return None

    @staticmethod
    def PickleBatch(sources: Sequence, level: LZ4Level = LZ4Level.L00_FAST, writer_mode: bool = False,
                    ctx: Optional[_native.Context] = None) -> List[bytes]:
        ctx = ctx or _native.default_context()
        blocks = [_ro_view(s, "source") for s in sources]
        src, soff, slen = pack_blocks(blocks)
        caps = np.array([ctx.lib.k4lz4_pickle_bound(b.size) for b in blocks], dtype=np.int32)
        dst, doff = make_arena(caps)
        out = np.empty(len(blocks), dtype=np.int32)
        a = _batch_args(src, soff, slen, dst, doff, caps, out)
        ctx.check(ctx.lib.k4lz4_pickle_batch(ctx.handle, *a, int(level), FLAG_PICKLE_WRITER if writer_mode else 0))
        if (out < 0).any():
            raise _native.NativeLibraryError("pickle kernel reported a slot too small (internal error)")
        return [dst[int(o):int(o) + int(n)].tobytes() for n, o in zip(out, doff)]

    # ---- Unpickle ---------------------------------------------------------------------------------
    @staticmethod
    def UnpickledSize(source) -> int:
        src = _ro_view(source, "source")
        if src.size == 0:
            return 0
        n = _native.load_library().k4lz4_unpickle_size(src.ctypes.data, src.size)
        if n < 0:
            raise _corrupted("header")
        return n

    @staticmethod
    def Unpickle(source, *args):
        """Unpickle(source) -> bytes | Unpickle(source, index, count) -> bytes |
        Unpickle(source, writer) -> None | Unpickle(source, output: writable buffer) -> None"""
        src = _ro_view(source, "source")
        if len(args) == 2:
            index, count = int(args[0]), int(args[1])
            _validate(src, index, count, "source")
            src = src[index:index + count]
            args = ()
        if len(args) == 0:
            return LZ4Pickler.UnpickleBatch([src])[0]
        if len(args) != 1:
            raise TypeError("Unpickle(source[, index, count | writer | output])")
        target = args[0]
        if target is None:
            raise TypeError("writer is null")                      # unpickle.cs:60-61
        if hasattr(target, "write"):
            data = LZ4Pickler.UnpickleBatch([src])[0]
            if len(data):
                target.write(data)
            return None
        out = _rw_view(target, "output")
        if src.size == 0:
            return None                                            # unpickle.cs:103
        expected = LZ4Pickler.UnpickledSize(src)
        if out.size != expected:                                   # unpickle.cs:115-117
            raise _corrupted(f"Output buffer size ({out.size}) does not match expected value ({expected})")
        out[:] = np.frombuffer(LZ4Pickler.UnpickleBatch([src])[0], dtype=np.uint8)
        return None

    @staticmethod
    def UnpickleBatch(sources: Sequence, ctx: Optional[_native.Context] = None) -> List[bytes]:
        ctx = ctx or _native.default_context()
        blocks = [_ro_view(s, "source") for s in sources]
        sizes = []
        for i, b in enumerate(blocks):
            n = 0 if b.size == 0 else ctx.lib.k4lz4_unpickle_size(b.ctypes.data, b.size)
            if n < 0:
                raise _corrupted(f"header of message {i}")
            sizes.append(n)
        src, soff, slen = pack_blocks(blocks)
        caps = np.array(sizes, dtype=np.int32)
        dst, doff = make_arena(caps)
        out = np.empty(len(blocks), dtype=np.int32)
        a = _batch_args(src, soff, slen, dst, doff, caps, out)
        ctx.check(ctx.lib.k4lz4_unpickle_batch(ctx.handle, *a, 0))
        bad = np.nonzero(out < 0)[0]
        if bad.size:
            raise _corrupted(f"message {int(bad[0])} does not decode to {sizes[int(bad[0])]} bytes")
        return [dst[int(o):int(o) + int(n)].tobytes() for n, o in zip(out, doff)]
