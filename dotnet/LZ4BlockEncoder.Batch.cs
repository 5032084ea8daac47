// Encoders/LZ4BlockEncoder.Batch.cs -- the batching front-end of the independent-block encoder / decoder
// (Encoders/LZ4BlockEncoder.cs:7-23, Encoders/LZ4EncoderBase.cs:46-97, Encoders/LZ4BlockDecoder.cs:39-71).
// LZ4BlockEncoder.Encode hands ONE block to LZ4Codec.Encode; a GPU wants many.  EncodeBlocks / DecodeBlocks take K blocks of
// one stream (or of many) and make one native call -- the allowCopy rule of LZ4EncoderBase.Encode (:79-83: a block that did
// not shrink is stored raw and reported as -length) is applied on the device (K4LZ4_FLAG_ALLOW_COPY), so the frame writer's
// raw bit comes back with the lengths.  Compile-unverified.
using System;
using K4os.Compression.LZ4.Engine;

namespace K4os.Compression.LZ4.Encoders
{
	public unsafe partial class LZ4BlockEncoder
	{
		/// <summary>Encodes source as consecutive independent blocks of BlockSize bytes (the last one shorter).  Block i goes to
		/// target[i * slot .. ) with slot = MaximumOutputSize(BlockSize); encoded[i] is what Encode(target, slot, allowCopy) returns
		/// for it: bytes written, or -length when the block was stored raw (allowCopy).  Returns the number of blocks.</summary>
		public int EncodeBlocks(ReadOnlySpan<byte> source, Span<byte> target, Span<int> encoded, bool allowCopy)
		{
			var blockSize = BlockSize;
			var slot = LZ4Codec.MaximumOutputSize(blockSize);
			var n = (int) (((long) source.Length + blockSize - 1) / blockSize);
			if (encoded.Length < n) throw new ArgumentException("one result per block", nameof(encoded));
			if ((long) n * slot > target.Length) throw new ArgumentException("target: MaximumOutputSize(BlockSize) bytes per block", nameof(target));
			if (n == 0) return 0;
			var srcOff = new ulong[n]; var srcLen = new int[n]; var dstOff = new ulong[n]; var dstCap = new int[n];
			for (var i = 0; i < n; i++)
			{
				srcOff[i] = (ulong) i * (ulong) blockSize;
				srcLen[i] = Math.Min(blockSize, source.Length - i * blockSize);
				dstOff[i] = (ulong) i * (ulong) slot;
				dstCap[i] = slot;
			}
			using var lease = NativeContext.Rent();
			var ctx = lease.Handle;
			fixed (byte* s = source, t = target)
			fixed (ulong* so = srcOff, to = dstOff)
			fixed (int* sl = srcLen, tl = dstCap, ol = encoded)
				LLNative.ThrowIfFailed(
					LLNative.k4lz4_encode_batch(ctx, s, so, sl, t, to, tl, ol, n, (int) _level, allowCopy ? LLNative.FLAG_ALLOW_COPY : 0), ctx);
			for (var i = 0; i < n; i++)
				if (encoded[i] == 0 || (encoded[i] < 0 && !allowCopy))
					throw new InvalidOperationException("Failed to encode chunk. Target buffer too small."); // LZ4EncoderBase.cs:75-77
			return n;
		}
	}

	public unsafe partial class LZ4BlockDecoder
	{
		/// <summary>Decodes n independent blocks of one stream: block i is source[offsets[i] .. +lengths[i]) -- a negative length
		/// marks a block stored raw (LZ4BlockDecoder.Inject, :39-50) -- and goes to target[i * BlockSize ..).  decoded[i] = bytes
		/// produced; throws as Decode does (:52-71) when a block does not decode.</summary>
		public void DecodeBlocks(ReadOnlySpan<byte> source, ReadOnlySpan<ulong> offsets, ReadOnlySpan<int> lengths, Span<byte> target, Span<int> decoded)
		{
			var n = offsets.Length;
			if (lengths.Length != n || decoded.Length != n) throw new ArgumentException("batch vectors differ in length");
			if ((long) n * BlockSize > target.Length) throw new ArgumentException("target: BlockSize bytes per block", nameof(target));
			var srcLen = new int[n]; var dstOff = new ulong[n]; var dstCap = new int[n];
			var raw = 0;
			for (var i = 0; i < n; i++)
			{
				dstOff[i] = (ulong) i * (ulong) BlockSize;
				dstCap[i] = BlockSize;
				if (lengths[i] < 0)
				{   // stored raw: a copy, no kernel
					var len = -lengths[i];
					if (len > BlockSize) throw new InvalidOperationException("Block is too large"); // LZ4BlockDecoder.cs:44
					source.Slice((int) offsets[i], len).CopyTo(target.Slice(i * BlockSize, len));
					decoded[i] = len; srcLen[i] = 0; raw++;
				}
				else srcLen[i] = lengths[i];
			}
			if (raw == n) return;
			var results = new int[n];
			using var lease = NativeContext.Rent();
			var ctx = lease.Handle;
			fixed (byte* s = source, t = target)
			fixed (ulong* so = offsets, to = dstOff)
			fixed (int* sl = srcLen, tl = dstCap, ol = results)
				LLNative.ThrowIfFailed(LLNative.k4lz4_decode_batch(ctx, s, so, sl, t, to, tl, ol, n, 0), ctx);
			for (var i = 0; i < n; i++)
			{
				if (lengths[i] < 0) continue;
				if (results[i] < 0) throw new InvalidOperationException("Failed to decode chunk"); // LZ4BlockDecoder.cs:62-63 via LZ4Codec.Decode < 0
				decoded[i] = results[i];
			}
		}
	}
}
