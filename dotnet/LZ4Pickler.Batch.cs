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
		/// <summary>What one packed native call may carry (sources, and targets): below the 2 GiB a byte[] can index.</summary>
		internal const long MaxPackedBytes = 0x7FF00000;

		private enum Op { Pickle, PickleForWriter, Unpickle }

		/// <summary>Pickle(message, level) for every message.</summary>
		public static byte[][] PickleBatch(IReadOnlyList<ReadOnlyMemory<byte>> messages, LZ4Level level = LZ4Level.L00_FAST) =>
			RunBatch(messages ?? throw new ArgumentNullException(nameof(messages)), Op.Pickle, level);

		/// <summary>Pickle(source, writer) header rule (pickle.cs:113-158): the writer path sizes the header from the SOURCE length,
		/// so its bytes can differ from the array path.  The batch form of that is K4LZ4_FLAG_PICKLE_WRITER.</summary>
		public static byte[][] PickleBatchForWriter(IReadOnlyList<ReadOnlyMemory<byte>> messages, LZ4Level level = LZ4Level.L00_FAST) =>
			RunBatch(messages ?? throw new ArgumentNullException(nameof(messages)), Op.PickleForWriter, level);

		/// <summary>Unpickle(pickle) for every pickle; throws InvalidDataException for the first corrupted one, as Unpickle does
		/// (unpickle.cs:160-161): bad version, short header, size mismatch, or a block that does not decode to its stated size.</summary>
		public static byte[][] UnpickleBatch(IReadOnlyList<ReadOnlyMemory<byte>> pickles) =>
			RunBatch(pickles ?? throw new ArgumentNullException(nameof(pickles)), Op.Unpickle, LZ4Level.L00_FAST);

		// The items travel packed into one managed buffer per native call; a byte[] holds less than 2 GiB (rank 0's share of the
		// configs[3] batch is 6.3 GB), so the batch is cut into runs whose sources AND targets both stay below MaxPackedBytes.
		private static unsafe byte[][] RunBatch(IReadOnlyList<ReadOnlyMemory<byte>> items, Op op, LZ4Level level)
		{
			var n = items.Count;
			var result = new byte[n][];
			var caps = new int[n];
			for (var i = 0; i < n; i++)
			{
				if (op == Op.Unpickle)
				{
					fixed (byte* p = items[i].Span) caps[i] = LLNative.k4lz4_unpickle_size(p, items[i].Length);   // header arithmetic, on the host
					if (caps[i] < 0) throw new InvalidDataException($"Pickle is corrupted: message {i}: header");
				}
				else caps[i] = LLNative.k4lz4_pickle_bound(items[i].Length);
			}
			using var lease = NativeContext.Rent();
			var ctx = lease.Handle;
			for (var first = 0; first < n;)
			{
				long st = 0, dt = 0;
				var last = first;
				while (last < n && (last == first || (st + items[last].Length <= MaxPackedBytes && dt + caps[last] <= MaxPackedBytes)))
				{
					st += items[last].Length; dt += caps[last]; last++;
				}
				var m = last - first;
				var srcOff = new ulong[m]; var srcLen = new int[m]; var dstOff = new ulong[m]; var dstCap = new int[m]; var outLen = new int[m];
				var src = new byte[Math.Max(1, st)];
				var dst = new byte[Math.Max(1, dt)];
				int so = 0, dof = 0;
				for (var i = 0; i < m; i++)
				{
					var item = items[first + i];
					srcOff[i] = (ulong) so; srcLen[i] = item.Length;
					item.Span.CopyTo(src.AsSpan(so, item.Length));
					so += item.Length;
					dstOff[i] = (ulong) dof; dstCap[i] = caps[first + i]; dof += dstCap[i];
				}
				fixed (byte* s = src, t = dst)
				fixed (ulong* pso = srcOff, pto = dstOff)
				fixed (int* sl = srcLen, tl = dstCap, ol = outLen)
				{
					var status = op == Op.Unpickle
						? LLNative.k4lz4_unpickle_batch(ctx, s, pso, sl, t, pto, tl, ol, m, 0)
						: LLNative.k4lz4_pickle_batch(ctx, s, pso, sl, t, pto, tl, ol, m, (int) level, op == Op.PickleForWriter ? LLNative.FLAG_PICKLE_WRITER : 0);
					LLNative.ThrowIfFailed(status, ctx);
				}
				for (var i = 0; i < m; i++)
				{
					if (op == Op.Unpickle)
					{
						if (outLen[i] < 0) throw new InvalidDataException($"Pickle is corrupted: message {first + i}: expected {dstCap[i]} bytes");
						result[first + i] = dstCap[i] == 0 ? Array.Empty<byte>() : dst.AsSpan((int) dstOff[i], dstCap[i]).ToArray();
					}
					else
					{
						// Pickle of an empty message is an empty array (pickle.cs:54); outLen < 0 cannot happen with bound-sized targets
						if (outLen[i] < 0) throw new InvalidOperationException($"message {first + i} could not be pickled");
						result[first + i] = outLen[i] == 0 ? Array.Empty<byte>() : dst.AsSpan((int) dstOff[i], outLen[i]).ToArray();
					}
				}
				first = last;
			}
			return result;
		}
	}
}
