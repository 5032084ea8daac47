// The Algorithm.Native arm: EDITS to existing reference files, written out as the members they become.  Compile-unverified.
//
// Engine/Algorithm.cs:4-10
//     internal enum Algorithm { X32, X64, Native }
//
// Engine/LL.tools.cs:29-36 -- who selects it (opt-in switch next to Enforce32; default stays X64/X32):
//     public static bool UseNative { get; set; } = false;
//     public static Algorithm Algorithm => UseNative ? Algorithm.Native : Enforce32 || Mem.System32 ? Algorithm.X32 : Algorithm.X64;
//   With UseNative, LZ4Codec.Enforce32 (LZ4Codec.cs:21-25) additionally forwards to LLNative.k4lz4_set_enforce32(value ? 1 : 0):
//   the library then produces the 32-bit engine's bytes for fast-level inputs of 64 KiB and more (the only place where LL32
//   and LL64 differ in a 64-bit process) -- pinned to LL32 itself, compiled here from the reference's x32/LL32.*.cs
//   (oracle/_ref; tests/test_ref_pins.py, tests/test_gpu_ref_parity.py).
//
// Engine/LLxx.cs -- every switch gets the third arm; the members below replace :17-26, :29-39, :41-55, :65-75, :94-103.
using System;

namespace K4os.Compression.LZ4.Engine
{
	internal static unsafe partial class LLxx
	{
		public static int LZ4_decompress_safe(byte* source, byte* target, int sourceLength, int targetLength) =>
			LL.Algorithm switch {
				Algorithm.X64 => LL64.LZ4_decompress_safe(source, target, sourceLength, targetLength),
				Algorithm.X32 => LL32.LZ4_decompress_safe(source, target, sourceLength, targetLength),
				Algorithm.Native => LLNative.Checked(LLNative.k4lz4_decompress_safe(source, target, sourceLength, targetLength)),
				_ => throw AlgorithmNotImplemented(nameof(LZ4_decompress_safe))
			};

		public static int LZ4_decompress_safe_partial(byte* source, byte* target, int sourceLength, int targetLength) =>
			LL.Algorithm switch {
				Algorithm.X64 => LL64.LZ4_decompress_safe_partial(source, target, sourceLength, targetLength, targetLength),
				Algorithm.X32 => LL32.LZ4_decompress_safe_partial(source, target, sourceLength, targetLength, targetLength),
				// decoding stops at targetLength bytes, as the managed arms (which pass targetLength as capacity too, LLxx.cs:29-39)
				Algorithm.Native => LLNative.Checked(LLNative.k4lz4_decompress_safe_partial(source, target, sourceLength, targetLength)),
				_ => throw AlgorithmNotImplemented(nameof(LZ4_decompress_safe_partial))
			};

		public static int LZ4_decompress_safe_usingDict(
			byte* source, byte* target, int sourceLength, int targetLength, byte* dictionary, int dictionaryLength) =>
			LL.Algorithm switch {
				Algorithm.X64 => LL64.LZ4_decompress_safe_usingDict(source, target, sourceLength, targetLength, dictionary, dictionaryLength),
				Algorithm.X32 => LL32.LZ4_decompress_safe_usingDict(source, target, sourceLength, targetLength, dictionary, dictionaryLength),
				Algorithm.Native => LLNative.Checked(LLNative.k4lz4_decompress_safe_using_dict(
					source, target, sourceLength, targetLength, dictionary, dictionaryLength)),
				_ => throw AlgorithmNotImplemented(nameof(LZ4_decompress_safe_usingDict))
			};

		public static int LZ4_compress_fast(byte* source, byte* target, int sourceLength, int targetLength, int acceleration) =>
			LL.Algorithm switch {
				Algorithm.X64 => LL64.LZ4_compress_fast(source, target, sourceLength, targetLength, acceleration),
				Algorithm.X32 => LL32.LZ4_compress_fast(source, target, sourceLength, targetLength, acceleration),
				Algorithm.Native => LLNative.Checked(LLNative.k4lz4_compress_fast(source, target, sourceLength, targetLength, acceleration)),
				_ => throw AlgorithmNotImplemented(nameof(LZ4_compress_fast))
			};

		public static int LZ4_compress_HC(byte* source, byte* target, int sourceLength, int targetLength, int compressionLevel) =>
			LL.Algorithm switch {
				Algorithm.X64 => LL64.LZ4_compress_HC(source, target, sourceLength, targetLength, compressionLevel),
				Algorithm.X32 => LL32.LZ4_compress_HC(source, target, sourceLength, targetLength, compressionLevel),
				// LZ4Codec only reaches this member with level >= L03_HC (LZ4Codec.cs:48-50); lower values fail with ArgumentException
				Algorithm.Native => LLNative.Checked(LLNative.k4lz4_compress_hc(source, target, sourceLength, targetLength, compressionLevel)),
				_ => throw AlgorithmNotImplemented(nameof(LZ4_compress_HC))
			};

		// The *_continue members (LLxx.cs:57-62, :78-92, :105-113: chained blocks) keep their two managed arms: consecutive
		// blocks of one stream depend on each other, nothing to batch.  With Algorithm.Native they use LL64.
	}
}
