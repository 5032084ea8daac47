"""Host-side mirror of the independent-block stream encoder / decoder, backed by
libk4lz4.so.

Mirrors:
  ILZ4Encoder  (Topup / Encode / BlockSize / BytesReady)      Encoders/ILZ4Encoder.cs
  LZ4EncoderBase                                              Encoders/LZ4EncoderBase.cs:28-97
  LZ4BlockEncoder(level, blockSize)                           Encoders/LZ4BlockEncoder.cs:7-23
  ILZ4Decoder  (Decode / Inject / Drain / Peek / BytesReady)  Encoders/ILZ4Decoder.cs
  LZ4BlockDecoder(blockSize)                                  Encoders/LZ4BlockDecoder.cs:11-106
  LZ4EncoderExtensions.TopupAndEncode / FlushAndEncode /
      DecodeAndDrain, EncoderAction                           Encoders/LZ4EncoderExtensions.cs:8-205, EncoderAction.cs
The chained encoders (LZ4FastChainEncoder / LZ4HighChainEncoder -> *_continue) are serial across
blocks and are not offered; chained *decoding* is (frames.py, k4lz4_decode_chain_batch).

`LZ4BlockEncoder.EncodeBlocks` is the batching front-end the frame writer uses: K blocks, one launch,
with the reference's allowCopy rule applied on the device.
"""
from __future__ import annotations

import enum
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from ._native import FLAG_ALLOW_COPY
from .codec import LZ4Codec, LZ4Level, _ro_view, _rw_view, _batch_args, pack_blocks, make_arena

K1 = 1024


class InvalidOperationException(Exception):
    """System.InvalidOperationException"""


class EncoderAction(enum.IntEnum):          # Encoders/EncoderAction.cs
    None_ = 0
    Loaded = 1
    Encoded = 2
    Copied = 3


def _round_block_size(block_size: int) -> int:
    block_size = max(int(block_size), K1)
    return (block_size + K1 - 1) // K1 * K1                      # Mem.RoundUp(Math.Max(blockSize, Mem.K1), Mem.K1)


class LZ4BlockEncoder:
    """Independent block encoder (LZ4BlockEncoder.cs): every block is LZ4Codec.Encode of its own bytes."""

    def __init__(self, level: LZ4Level = LZ4Level.L00_FAST, blockSize: int = 65536):
        self._level = LZ4Level(level)
        self._block_size = _round_block_size(blockSize)
        self._input = np.zeros(self._block_size + 32 + 8, np.uint8)   # LZ4EncoderBase.cs:34-37, no dictionary part
        self._index = 0
        self._pointer = 0

    @property
    def BlockSize(self) -> int:
        return self._block_size

    @property
    def BytesReady(self) -> int:
        return self._pointer - self._index

    def Topup(self, source, offset: int = 0, length: Optional[int] = None) -> int:
        """copies up to the free space of the current block; returns bytes taken (LZ4EncoderBase.cs:46-62)"""
        src = _ro_view(source, "source")
        length = src.size - offset if length is None else int(length)
        if length == 0:
            return 0
        space = self._index + self._block_size - self._pointer
        if space <= 0:
            return 0
        chunk = min(space, length)
        self._input[self._pointer:self._pointer + chunk] = src[offset:offset + chunk]
        self._pointer += chunk
        return chunk

    def Encode(self, target, offset: int = 0, length: Optional[int] = None, allowCopy: bool = False) -> int:
        """encodes the pending bytes as one block into target; with allowCopy a block that does not shrink
        is stored raw and -length is returned (LZ4EncoderBase.cs:66-88)"""
        dst = _rw_view(target, "target")
        length = dst.size - offset if length is None else int(length)
        n = self._pointer - self._index
        if n <= 0:
            return 0
        encoded = LZ4Codec.Encode(self._input, self._index, n, dst, offset, length, self._level)
        if encoded <= 0:
            raise InvalidOperationException("Failed to encode chunk. Target buffer too small.")
        if allowCopy and encoded >= n:
            dst[offset:offset + n] = self._input[self._index:self._index + n]
            encoded = -n
        self._index = self._pointer = 0                              # Commit(): CopyDict returns 0 for independent blocks
        return encoded

    # ---- batching front-end ---------------------------------------------------------------------
    def EncodeBlocks(self, sources: Sequence, allowCopy: bool = True,
                     ctx: Optional[_native.Context] = None) -> List[Tuple[EncoderAction, bytes]]:
        """each element (at most BlockSize bytes) as Topup + Encode(allowCopy) would produce it, one launch"""
        blocks = [_ro_view(s, "source") for s in sources]
        for b in blocks:
            if b.size > self._block_size:
                raise InvalidOperationException("block larger than BlockSize")
        out, dst, doff = encode_blocks_packed(blocks, self._level, allowCopy, ctx)
        res = []
        for n, o, b in zip(out, doff, blocks):
            if b.size == 0:
                res.append((EncoderAction.None_, b""))
            elif n == 0:
                raise InvalidOperationException("Failed to encode chunk. Target buffer too small.")
            elif n < 0:
                res.append((EncoderAction.Copied, dst[int(o):int(o) - int(n)].tobytes()))
            else:
                res.append((EncoderAction.Encoded, dst[int(o):int(o) + int(n)].tobytes()))
        return res


def encode_blocks_packed(blocks, level: LZ4Level, allow_copy: bool, ctx: Optional[_native.Context] = None):
    """-> (outLen int32 (negative: stored raw), arena, arena offsets)"""
    ctx = ctx or _native.default_context()
    src, soff, slen = pack_blocks(blocks)
    caps = np.array([LZ4Codec.MaximumOutputSize(b.size) for b in blocks], dtype=np.int32)
    dst, doff = make_arena(caps)
    out = np.empty(len(blocks), dtype=np.int32)
    a = _batch_args(src, soff, slen, dst, doff, caps, out)
    ctx.check(ctx.lib.k4lz4_encode_batch(ctx.handle, *a, int(level), FLAG_ALLOW_COPY if allow_copy else 0))
    return out, dst, doff


class LZ4BlockDecoder:
    """Decoder for independent blocks (LZ4BlockDecoder.cs)."""

    def __init__(self, blockSize: int = 65536):
        self._block_size = _round_block_size(blockSize)
        self._output_length = self._block_size + 8
        self._output = np.zeros(self._output_length + 8, np.uint8)
        self._output_index = 0

    @property
    def BlockSize(self) -> int:
        return self._block_size

    @property
    def BytesReady(self) -> int:
        return self._output_index

    def Decode(self, source, offset: int = 0, length: Optional[int] = None, blockSize: int = 0) -> int:
        src = _ro_view(source, "source")
        length = src.size - offset if length is None else int(length)
        if blockSize <= 0:
            blockSize = self._block_size
        if blockSize > self._block_size:
            raise InvalidOperationException()
        decoded = LZ4Codec.Decode(src, offset, length, self._output, 0, self._output_length)
        if decoded < 0:
            raise InvalidOperationException()
        self._output_index = decoded
        return decoded

    def Inject(self, source, offset: int = 0, length: Optional[int] = None) -> int:
        src = _ro_view(source, "source")
        length = src.size - offset if length is None else int(length)
        if length <= 0:
            self._output_index = 0
            return 0
        if length > self._output_length:
            raise InvalidOperationException()
        self._output[:length] = src[offset:offset + length]
        self._output_index = length
        return length

    def Drain(self, target, offset: int, length: int, targetOffset: int = 0) -> None:
        """offset is relative to the end of the decoded data (negative), LZ4BlockDecoder.cs:75-85"""
        dst = _rw_view(target, "target")
        start = self._output_index + int(offset)
        if start < 0 or length < 0 or start + length > self._output_index:
            raise InvalidOperationException()
        dst[targetOffset:targetOffset + length] = self._output[start:start + length]

    def Peek(self, offset: int) -> np.ndarray:
        start = self._output_index + int(offset)
        if start < 0 or start > self._output_index:
            raise InvalidOperationException()
        return self._output[start:self._output_index]


# ---- LZ4EncoderExtensions ---------------------------------------------------------------------------
def TopupAndEncode(encoder: LZ4BlockEncoder, source, target, forceEncode: bool, allowCopy: bool):
    """-> (action, loaded, encoded)   (LZ4EncoderExtensions.cs:117-133, :183-205)"""
    src = _ro_view(source, "source")
    loaded = encoder.Topup(src) if src.size > 0 else 0
    action, encoded = FlushAndEncode(encoder, target, forceEncode, allowCopy, loaded)
    return action, loaded, encoded


def FlushAndEncode(encoder: LZ4BlockEncoder, target, forceEncode: bool = True, allowCopy: bool = True, loaded: int = 0):
    """-> (action, encoded)"""
    if encoder.BytesReady < (1 if forceEncode else encoder.BlockSize):
        return (EncoderAction.Loaded if loaded > 0 else EncoderAction.None_), 0
    encoded = encoder.Encode(target, allowCopy=allowCopy)
    if not allowCopy or encoded >= 0:
        return EncoderAction.Encoded, encoded
    return EncoderAction.Copied, -encoded


def DecodeAndDrain(decoder: LZ4BlockDecoder, source, target):
    """-> (ok, decoded): decodes one block and copies it to the start of target
    (LZ4EncoderExtensions.cs:288-305: false for an empty source, a failed decode or a target too small)"""
    src = _ro_view(source, "source")
    if src.size <= 0:
        return False, 0
    decoded = decoder.Decode(src)
    dst = _rw_view(target, "target")
    if decoded <= 0 or dst.size < decoded:
        return False, decoded
    decoder.Drain(dst, -decoded, decoded)
    return True, decoded
