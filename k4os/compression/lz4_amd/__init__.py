"""k4os.compression.lz4_amd -- MI355X (gfx950) LZ4 block codec behind the K4os.Compression.LZ4
block API: LZ4Codec.Encode / Decode, LZ4Pickler.Pickle / Unpickle and their batch forms.
All compute runs in hand-written HIP kernels (csrc/) reached through the C ABI of libk4lz4.so
(include/k4lz4.h).  There is no CPU fallback."""
from .codec import LZ4Codec, LZ4Level, pack_blocks, make_arena
from .pickler import LZ4Pickler, InvalidDataException
from .encoders import (LZ4BlockEncoder, LZ4BlockDecoder, EncoderAction, InvalidOperationException, TopupAndEncode,
                       FlushAndEncode, DecodeAndDrain)
from .frames import LZ4Frame, LZ4EncoderSettings, LZ4Descriptor, parse_frame, xxh32_many
from ._native import NativeLibraryError, Context, load_library, default_context, host_register, host_unregister

__all__ = ["LZ4Codec", "LZ4Level", "LZ4Pickler", "InvalidDataException", "NativeLibraryError", "Context",
           "load_library", "default_context", "host_register", "host_unregister", "pack_blocks", "make_arena", "LZ4BlockEncoder", "LZ4BlockDecoder",
           "EncoderAction", "InvalidOperationException", "TopupAndEncode", "FlushAndEncode", "DecodeAndDrain", "LZ4Frame",
           "LZ4EncoderSettings", "LZ4Descriptor", "parse_frame", "xxh32_many"]
