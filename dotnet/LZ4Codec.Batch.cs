// LZ4Codec.Batch.cs -- the batch surface beside LZ4Codec.Encode / Decode (LZ4Codec.cs:40-115): n independent blocks, one call,
// one kernel launch.  Per-block results follow Encode / Decode: > 0 bytes written, 0 for an empty source, -1 failure.
// Compile-unverified.
using System;
using K4os.Compression.LZ4.Engine;

namespace K4os.Compression.LZ4
{
	public static partial class LZ4Codec
	{
		/// <summary>What this host's threads sustain together on block work (GiB/s), for the batch-size crossover; 0 = the
		/// library's own figures for the box it was measured on.</summary>
		public static double HostGiBs { get; set; } = 0;

		/// <summary>Upper bound on the threads the managed fallback of a small batch may use (0: Environment.ProcessorCount).</summary>
		public static int MaxHostThreads { get; set; } = 0;

		/// <summary>Compresses n independent blocks. Block i is source[sourceOffsets[i] .. +sourceLengths[i]) and goes to
		/// target[targetOffsets[i] .. +targetLengths[i]); encodedLengths[i] is what Encode(...) would return for it.</summary>
		public static unsafe void EncodeBatch(
			ReadOnlySpan<byte> source, ReadOnlySpan<ulong> sourceOffsets, ReadOnlySpan<int> sourceLengths,
			Span<byte> target, ReadOnlySpan<ulong> targetOffsets, ReadOnlySpan<int> targetLengths,
			Span<int> encodedLengths, LZ4Level level = LZ4Level.L00_FAST)
		{
			var n = ValidateBatch(source.Length, sourceOffsets, sourceLengths, target.Length, targetOffsets, targetLengths, encodedLengths.Length);
			if (n == 0) return;
			using var lease = NativeContext.Rent();
			var ctx = lease.Handle;
			fixed (byte* s = source, t = target)
			fixed (ulong* so = sourceOffsets, to = targetOffsets)
			fixed (int* sl = sourceLengths, tl = targetLengths, ol = encodedLengths)
				LLNative.ThrowIfFailed(LLNative.k4lz4_encode_batch(ctx, s, so, sl, t, to, tl, ol, n, (int) level, 0), ctx);
		}

		/// <summary>Decompresses n independent blocks; decodedLengths[i] is what Decode(...) would return (-1: corrupt / too small).</summary>
		public static unsafe void DecodeBatch(
			ReadOnlySpan<byte> source, ReadOnlySpan<ulong> sourceOffsets, ReadOnlySpan<int> sourceLengths,
			Span<byte> target, ReadOnlySpan<ulong> targetOffsets, ReadOnlySpan<int> targetLengths,
			Span<int> decodedLengths)
		{
			var n = ValidateBatch(source.Length, sourceOffsets, sourceLengths, target.Length, targetOffsets, targetLengths, decodedLengths.Length);
			if (n == 0) return;
			using var lease = NativeContext.Rent();
			var ctx = lease.Handle;
			fixed (byte* s = source, t = target)
			fixed (ulong* so = sourceOffsets, to = targetOffsets)
			fixed (int* sl = sourceLengths, tl = targetLengths, ol = decodedLengths)
				LLNative.ThrowIfFailed(LLNative.k4lz4_decode_batch(ctx, s, so, sl, t, to, tl, ol, n, 0), ctx);
		}

		/// <summary>Convenience form: every block compressed into a fresh array.  The blocks travel packed into one managed
		/// buffer per native call; a byte[] holds less than 2 GiB, so the batch is cut into runs of blocks whose sources AND
		/// MaximumOutputSize targets both stay below <see cref="MaxPackedBytes"/> -- one native call (one launch) per run.</summary>
		public static byte[][] EncodeBatch(byte[][] blocks, LZ4Level level = LZ4Level.L00_FAST)
		{
			if (blocks is null) throw new ArgumentNullException(nameof(blocks));
			var n = blocks.Length;
			var result = new byte[n][];
			// (the per-element check comes first: the device path and the managed path must fail the same way on a null element)
			long total = 0;
			for (var i = 0; i < n; i++)
			{
				if (blocks[i] is null) throw new ArgumentNullException($"{nameof(blocks)}[{i}]");
				total += blocks[i].Length;
			}
			// A device call has a floor (one block's time on its wavefront, about 2 ms for 64 KiB at the fast level): below the batch
			// size the library recommends for this host, the managed engine is the faster one (INTEGRATION.md, "Crossover").  The
			// size asked about is the batch's MEAN block (ragged batches), the host rate the one of ALL host threads -- which is what
			// the fallback below then really uses: Parallel.For over the managed engine itself (LL64 / LL32), never LLxx's Native
			// arm, which would be one device call per block.
			var meanBlock = (int) Math.Max(1, n > 0 ? total / n : 1);
			if (n > 0 && n < LLNative.k4lz4_recommended_min_batch(level < LZ4Level.L03_HC ? 0 : 2, meanBlock, HostGiBs))
			{
				void One(int i)
				{
					var buf = new byte[MaximumOutputSize(blocks[i].Length)];
					var k = ManagedEncode(blocks[i], buf, level);
					result[i] = k <= 0 && blocks[i].Length > 0 ? null : buf.AsSpan(0, Math.Max(k, 0)).ToArray();
				}
				// one or two blocks: the caller's thread (what the serial loop did); more: at most one thread per block and never more than
				// MaxHostThreads (default: the processor count -- HostGiBs is read as the rate of that many threads), and an exception
				// inside the loop reaches the caller as itself, as from the serial loop and from the device path (ADVICE round 5)
				if (n <= 2) { for (var i = 0; i < n; i++) One(i); return result; }
				var options = new System.Threading.Tasks.ParallelOptions { MaxDegreeOfParallelism = Math.Max(1, Math.Min(n, MaxHostThreads > 0 ? MaxHostThreads : Environment.ProcessorCount)) };
				try { System.Threading.Tasks.Parallel.For(0, n, options, One); }
				catch (AggregateException e) when (e.InnerExceptions.Count == 1)
				{
					System.Runtime.ExceptionServices.ExceptionDispatchInfo.Capture(e.InnerExceptions[0]).Throw();
				}
				return result;
			}
			for (var first = 0; first < n;)
			{
				long st = 0, dt = 0;
				var last = first;
				while (last < n)
				{
					var bound = MaximumOutputSize(blocks[last].Length);
					if (last > first && (st + blocks[last].Length > MaxPackedBytes || dt + bound > MaxPackedBytes)) break;
					st += blocks[last].Length; dt += bound; last++;
				}
				EncodeRun(blocks, first, last, (int) Math.Max(1, st), (int) Math.Max(1, dt), level, result);
				first = last;
			}
			return result;
		}

		/// <summary>The reference's managed engine on one block, whatever LL.UseNative says (LZ4Codec.cs:40-52 with the
		/// Algorithm switch of Engine/LLxx.cs:65-103 resolved to X64 / X32 by hand).</summary>
		private static unsafe int ManagedEncode(byte[] source, byte[] target, LZ4Level level)
		{
			if (source.Length == 0) return 0;
			fixed (byte* s = source, t = target)
			{
				var x32 = LL.Enforce32 || Mem.System32;
				if (level < LZ4Level.L03_HC)
					return x32 ? LL32.LZ4_compress_fast(s, t, source.Length, target.Length, 1) : LL64.LZ4_compress_fast(s, t, source.Length, target.Length, 1);
				return x32 ? LL32.LZ4_compress_HC(s, t, source.Length, target.Length, (int) level) : LL64.LZ4_compress_HC(s, t, source.Length, target.Length, (int) level);
			}
		}

		/// <summary>What one packed native call may carry (sources, and targets): below the 2 GiB a byte[] can index.</summary>
		internal const long MaxPackedBytes = 0x7FF00000;

		private static void EncodeRun(byte[][] blocks, int first, int last, int srcBytes, int dstBytes, LZ4Level level, byte[][] result)
		{
			var n = last - first;
			var srcOff = new ulong[n]; var srcLen = new int[n]; var dstOff = new ulong[n]; var dstCap = new int[n]; var outLen = new int[n];
			var src = new byte[srcBytes];
			var dst = new byte[dstBytes];
			int st = 0, dt = 0;
			for (var i = 0; i < n; i++)
			{
				var b = blocks[first + i];
				srcOff[i] = (ulong) st; srcLen[i] = b.Length;
				Buffer.BlockCopy(b, 0, src, st, b.Length);
				st += b.Length;
				dstOff[i] = (ulong) dt; dstCap[i] = MaximumOutputSize(b.Length); dt += dstCap[i];
			}
			EncodeBatch(src, srcOff, srcLen, dst, dstOff, dstCap, outLen, level);
			for (var i = 0; i < n; i++)
			{
				if (outLen[i] < 0) throw new InvalidOperationException($"block {first + i} did not fit into MaximumOutputSize bytes"); // cannot happen
				result[first + i] = new byte[outLen[i]];
				Buffer.BlockCopy(dst, (int) dstOff[i], result[first + i], 0, outLen[i]);
			}
		}

		// the checks Encode/Decode make per call (Internal/Extensions.cs:37-52), once per batch
		private static long ValidateBatch(
			int sourceLength, ReadOnlySpan<ulong> sourceOffsets, ReadOnlySpan<int> sourceLengths,
			int targetLength, ReadOnlySpan<ulong> targetOffsets, ReadOnlySpan<int> targetLengths, int results)
		{
			var n = sourceOffsets.Length;
			if (sourceLengths.Length != n || targetOffsets.Length != n || targetLengths.Length != n || results != n)
				throw new ArgumentException("batch vectors differ in length");
			for (var i = 0; i < n; i++)
			{
				if (sourceLengths[i] < 0 || sourceOffsets[i] + (ulong) sourceLengths[i] > (ulong) sourceLength)
					throw new ArgumentException($"block {i}: source range outside the buffer");
				if (targetLengths[i] < 0 || targetOffsets[i] + (ulong) targetLengths[i] > (ulong) targetLength)
					throw new ArgumentException($"block {i}: target range outside the buffer");
			}
			return n;
		}
	}
}
