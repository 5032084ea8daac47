// Engine/LLNative.cs -- P/Invoke table of libk4lz4.so (include/k4lz4.h).  Compile-unverified (no dotnet in the build image).
// Every entry mirrors one K4LZ4_API declaration; the per-block calls have the argument order and return values of the
// LLxx members they stand in for (Engine/LLxx.cs:17-26, :29-39, :41-55, :65-75, :94-103).
using System;
using System.IO;
using System.Runtime.InteropServices;

namespace K4os.Compression.LZ4.Engine
{
	internal static unsafe class LLNative
	{
		private const string Lib = "k4lz4"; // libk4lz4.so on the probing path

		// k4lz4_status
		public const int OK = 0, E_HIP = -1, E_ARG = -2, E_NOMEM = -3, E_NO_DEVICE = -4, E_UNSUPPORTED = -5;

		// k4lz4_flags
		public const int FLAG_RAW_RETURN = 1, FLAG_PICKLE_WRITER = 2, FLAG_NO_REORDER = 4, FLAG_REORDER = 8, FLAG_NO_SPLIT = 16,
			FLAG_PARTIAL = 32, FLAG_ALLOW_COPY = 64, FLAG_X32 = 128, FLAG_SEGMENTS = 256;

		[DllImport(Lib)] public static extern int k4lz4_version();
		[DllImport(Lib)] public static extern int k4lz4_device_count();
		[DllImport(Lib)] public static extern long k4lz4_recommended_min_batch(int kind, int blockBytes, double hostGiBs);
		[DllImport(Lib)] public static extern int k4lz4_ctx_create(out IntPtr ctx, int device);
		[DllImport(Lib)] public static extern void k4lz4_ctx_destroy(IntPtr ctx);
		[DllImport(Lib)] public static extern IntPtr k4lz4_last_error(IntPtr ctx);
		[DllImport(Lib)] public static extern int k4lz4_ctx_device(IntPtr ctx);
		[DllImport(Lib)] public static extern int k4lz4_synchronize(IntPtr ctx, IntPtr stream);
		[DllImport(Lib)] public static extern int k4lz4_ctx_reserve_hc(IntPtr ctx, long totalSrcBytes, int longestBlock);
		[DllImport(Lib)] public static extern int k4lz4_host_register(IntPtr ptr, UIntPtr bytes);
		[DllImport(Lib)] public static extern int k4lz4_host_unregister(IntPtr ptr);
		[DllImport(Lib)] public static extern void k4lz4_set_enforce32(int on);
		[DllImport(Lib)] public static extern int k4lz4_get_enforce32();
		[DllImport(Lib)] public static extern int k4lz4_compress_bound(int n);
		[DllImport(Lib)] public static extern int k4lz4_last_status();

		// ---- the LLxx seam, one block per call
		[DllImport(Lib)] public static extern int k4lz4_compress_fast(byte* src, byte* dst, int srcLen, int dstCap, int acceleration);
		[DllImport(Lib)] public static extern int k4lz4_compress_hc(byte* src, byte* dst, int srcLen, int dstCap, int level);
		[DllImport(Lib)] public static extern int k4lz4_decompress_safe(byte* src, byte* dst, int srcLen, int dstCap);
		[DllImport(Lib)] public static extern int k4lz4_decompress_safe_partial(byte* src, byte* dst, int srcLen, int targetLen);
		[DllImport(Lib)] public static extern int k4lz4_decompress_safe_using_dict(
			byte* src, byte* dst, int srcLen, int dstCap, byte* dict, int dictLen);

		// ---- batches of independent blocks (host pointers)
		[DllImport(Lib)] public static extern int k4lz4_encode_batch(
			IntPtr ctx, byte* src, ulong* srcOff, int* srcLen, byte* dst, ulong* dstOff, int* dstCap, int* outLen, long n, int level, int flags);
		[DllImport(Lib)] public static extern int k4lz4_decode_batch(
			IntPtr ctx, byte* src, ulong* srcOff, int* srcLen, byte* dst, ulong* dstOff, int* dstCap, int* outLen, long n, int flags);
		[DllImport(Lib)] public static extern int k4lz4_decode_dict_batch(
			IntPtr ctx, byte* src, ulong* srcOff, int* srcLen, byte* dst, ulong* dstOff, int* dstCap, int* outLen, long n, int flags,
			byte* dict, ulong* dictOff, int* dictLen);

		// ---- LZ4Pickler envelope
		[DllImport(Lib)] public static extern int k4lz4_pickle_bound(int srcLen);
		[DllImport(Lib)] public static extern int k4lz4_unpickle_size(byte* pickle, int pickleLen);
		[DllImport(Lib)] public static extern int k4lz4_pickle_batch(
			IntPtr ctx, byte* src, ulong* srcOff, int* srcLen, byte* dst, ulong* dstOff, int* dstCap, int* outLen, long n, int level, int flags);
		[DllImport(Lib)] public static extern int k4lz4_unpickle_batch(
			IntPtr ctx, byte* src, ulong* srcOff, int* srcLen, byte* dst, ulong* dstOff, int* dstCap, int* outLen, long n, int flags);

		// ---- frame layer
		[DllImport(Lib)] public static extern int k4lz4_xxh32_batch(IntPtr ctx, byte* data, ulong* off, ulong* len, uint* digests, long n, uint seed);
		[DllImport(Lib)] public static extern int k4lz4_decode_chain_batch(
			IntPtr ctx, byte* src, ulong* blkOff, uint* blkLen, long nBlocks, ulong* firstBlk, uint* nBlk, int* blockSize, byte* chained,
			byte* dst, ulong* dstOff, ulong* dstCap, long* outLen, long nStreams);

		// ---- device-resident variants: every pointer is a device pointer of the context's GPU, stream = hipStream_t
		[DllImport(Lib)] public static extern int k4lz4_encode_batch_device(
			IntPtr ctx, IntPtr src, IntPtr srcOff, IntPtr srcLen, IntPtr dst, IntPtr dstOff, IntPtr dstCap, IntPtr outLen, long n, int level, int flags, IntPtr stream);
		[DllImport(Lib)] public static extern int k4lz4_decode_batch_device(
			IntPtr ctx, IntPtr src, IntPtr srcOff, IntPtr srcLen, IntPtr dst, IntPtr dstOff, IntPtr dstCap, IntPtr outLen, long n, int flags, IntPtr stream);
		[DllImport(Lib)] public static extern int k4lz4_pickle_batch_device(
			IntPtr ctx, IntPtr src, IntPtr srcOff, IntPtr srcLen, IntPtr dst, IntPtr dstOff, IntPtr dstCap, IntPtr outLen, long n, int level, int flags, IntPtr stream);
		[DllImport(Lib)] public static extern int k4lz4_unpickle_batch_device(
			IntPtr ctx, IntPtr src, IntPtr srcOff, IntPtr srcLen, IntPtr dst, IntPtr dstOff, IntPtr dstCap, IntPtr outLen, long n, int flags, IntPtr stream);

		/// <summary>Call-level status -> the exception type the managed code base uses for the same situation.</summary>
		internal static void ThrowIfFailed(int status, IntPtr ctx)
		{
			if (status == OK) return;
			var msg = Marshal.PtrToStringAnsi(k4lz4_last_error(ctx)) ?? "libk4lz4 call failed";
			switch (status)
			{
				case E_ARG: throw new ArgumentException(msg); // Internal/Extensions.cs:37-52 semantics
				case E_NOMEM: throw new OutOfMemoryException(msg);
				case E_UNSUPPORTED: throw new NotImplementedException(msg);
				case E_NO_DEVICE: throw new PlatformNotSupportedException(msg); // never a silent fall back to LL64
				default: throw new InvalidOperationException(msg); // E_HIP, incl. "decoder wave pair timed out"
			}
		}

		/// <summary>After an LLxx-shaped call, whose int return cannot carry an infrastructure failure.</summary>
		internal static int Checked(int result)
		{
			ThrowIfFailed(k4lz4_last_status(), IntPtr.Zero);
			return result; // LLxx-level value: bytes, 0 = did not fit, decode error = -(pos) - 1
		}
	}
}
