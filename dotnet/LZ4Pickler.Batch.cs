// LZ4Pickler.Batch.cs -- Pickle / Unpickle for many messages at once (LZ4Pickler.pickle.cs:51-106, LZ4Pickler.unpickle.cs:39-50),
// byte-identical to calling them in a loop.  Compile-unverified.
using System;
using System.Collections.Generic;
using System.IO;
using K4os.Compression.LZ4.Engine;

namespace K4os.Compression.LZ4
{
	public static partial class LZ4Pickler
	{
		/// <summary>Pickle(message, level) for every message.</summary>
		public static unsafe byte[][] PickleBatch(IReadOnlyList<ReadOnlyMemory<byte>> messages, LZ4Level level = LZ4Level.L00_FAST)
		{
			if (messages is null) throw new ArgumentNullException(nameof(messages));
			var n = messages.Count;
			var result = new byte[n][];
			if (n == 0) return result;
			var srcOff = new ulong[n]; var srcLen = new int[n]; var dstOff = new ulong[n]; var dstCap = new int[n]; var outLen = new int[n];
			ulong st = 0, dt = 0;
			for (var i = 0; i < n; i++)
			{
				srcOff[i] = st; srcLen[i] = messages[i].Length; st += (ulong) srcLen[i];
				dstOff[i] = dt; dstCap[i] = LLNative.k4lz4_pickle_bound(srcLen[i]); dt += (ulong) dstCap[i];
			}
			var src = new byte[Math.Max(1UL, st)];
			var dst = new byte[Math.Max(1UL, dt)];
			for (var i = 0; i < n; i++) messages[i].Span.CopyTo(src.AsSpan((int) srcOff[i], srcLen[i]));
			var ctx = NativeContext.Current;
			fixed (byte* s = src, t = dst)
			fixed (ulong* so = srcOff, to = dstOff)
			fixed (int* sl = srcLen, tl = dstCap, ol = outLen)
				LLNative.ThrowIfFailed(LLNative.k4lz4_pickle_batch(ctx, s, so, sl, t, to, tl, ol, n, (int) level, 0), ctx);
			for (var i = 0; i < n; i++)
			{
				// Pickle of an empty message is an empty array (pickle.cs:54); outLen < 0 cannot happen with bound-sized targets
				if (outLen[i] < 0) throw new InvalidOperationException($"message {i} could not be pickled");
				result[i] = outLen[i] == 0 ? Array.Empty<byte>() : dst.AsSpan((int) dstOff[i], outLen[i]).ToArray();
			}
			return result;
		}

		/// <summary>Unpickle(pickle) for every pickle; throws InvalidDataException for the first corrupted one, as Unpickle does
		/// (unpickle.cs:160-161): bad version, short header, size mismatch, or a block that does not decode to its stated size.</summary>
		public static unsafe byte[][] UnpickleBatch(IReadOnlyList<ReadOnlyMemory<byte>> pickles)
		{
			if (pickles is null) throw new ArgumentNullException(nameof(pickles));
			var n = pickles.Count;
			var result = new byte[n][];
			if (n == 0) return result;
			var srcOff = new ulong[n]; var srcLen = new int[n]; var dstOff = new ulong[n]; var dstCap = new int[n]; var outLen = new int[n];
			ulong st = 0, dt = 0;
			for (var i = 0; i < n; i++)
			{
				srcOff[i] = st; srcLen[i] = pickles[i].Length; st += (ulong) srcLen[i];
				int size;
				fixed (byte* p = pickles[i].Span) size = LLNative.k4lz4_unpickle_size(p, srcLen[i]);   // header arithmetic, on the host
				if (size < 0) throw new InvalidDataException($"Pickle is corrupted: message {i}: header");
				dstOff[i] = dt; dstCap[i] = size; dt += (ulong) size;
			}
			var src = new byte[Math.Max(1UL, st)];
			var dst = new byte[Math.Max(1UL, dt)];
			for (var i = 0; i < n; i++) pickles[i].Span.CopyTo(src.AsSpan((int) srcOff[i], srcLen[i]));
			var ctx = NativeContext.Current;
			fixed (byte* s = src, t = dst)
			fixed (ulong* so = srcOff, to = dstOff)
			fixed (int* sl = srcLen, tl = dstCap, ol = outLen)
				LLNative.ThrowIfFailed(LLNative.k4lz4_unpickle_batch(ctx, s, so, sl, t, to, tl, ol, n, 0), ctx);
			for (var i = 0; i < n; i++)
			{
				if (outLen[i] < 0) throw new InvalidDataException($"Pickle is corrupted: message {i}: expected {dstCap[i]} bytes");
				result[i] = dstCap[i] == 0 ? Array.Empty<byte>() : dst.AsSpan((int) dstOff[i], dstCap[i]).ToArray();
			}
			return result;
		}

		/// <summary>Pickle(source, writer) header rule (pickle.cs:113-158): the writer path sizes the header from the SOURCE length,
		/// so its bytes can differ from the array path.  The batch form of that is K4LZ4_FLAG_PICKLE_WRITER.</summary>
		public static unsafe byte[][] PickleBatchForWriter(IReadOnlyList<ReadOnlyMemory<byte>> messages, LZ4Level level = LZ4Level.L00_FAST)
		{
			if (messages is null) throw new ArgumentNullException(nameof(messages));
			var n = messages.Count;
			var result = new byte[n][];
			if (n == 0) return result;
			var srcOff = new ulong[n]; var srcLen = new int[n]; var dstOff = new ulong[n]; var dstCap = new int[n]; var outLen = new int[n];
			ulong st = 0, dt = 0;
			for (var i = 0; i < n; i++)
			{
				srcOff[i] = st; srcLen[i] = messages[i].Length; st += (ulong) srcLen[i];
				dstOff[i] = dt; dstCap[i] = LLNative.k4lz4_pickle_bound(srcLen[i]); dt += (ulong) dstCap[i];
			}
			var src = new byte[Math.Max(1UL, st)];
			var dst = new byte[Math.Max(1UL, dt)];
			for (var i = 0; i < n; i++) messages[i].Span.CopyTo(src.AsSpan((int) srcOff[i], srcLen[i]));
			var ctx = NativeContext.Current;
			fixed (byte* s = src, t = dst)
			fixed (ulong* so = srcOff, to = dstOff)
			fixed (int* sl = srcLen, tl = dstCap, ol = outLen)
				LLNative.ThrowIfFailed(LLNative.k4lz4_pickle_batch(ctx, s, so, sl, t, to, tl, ol, n, (int) level, LLNative.FLAG_PICKLE_WRITER), ctx);
			for (var i = 0; i < n; i++)
				result[i] = outLen[i] <= 0 ? Array.Empty<byte>() : dst.AsSpan((int) dstOff[i], outLen[i]).ToArray();
			return result;
		}
	}
}
