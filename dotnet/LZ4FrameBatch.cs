// Streams/Frames/LZ4FrameBatch.cs -- whole-buffer frames through the batch calls: what LZ4FrameWriter / LZ4FrameReader do one
// block at a time (Frames/LZ4FrameWriter.async.cs:15-90: length word with raw bit, payload, optional block checksum, EndMark,
// optional content checksum; Frames/LZ4FrameReader.async.cs:108-136), done for all blocks of a frame -- or of many frames -- with
// one encode / decode launch and one XXH32 launch.  Frames of independent blocks only (LZ4EncoderSettings.ChainBlocks = false):
// chained blocks depend on each other and do not batch (the reader side does decode them, in order, k4lz4_decode_chain_batch).
// Byte layout and header arithmetic are LZ4FrameWriter.cs:57-108,:159-189.  Compile-unverified.
using System;
using System.Buffers.Binary;
using K4os.Compression.LZ4.Encoders;
using K4os.Compression.LZ4.Engine;

namespace K4os.Compression.LZ4.Streams.Frames
{
	public static unsafe class LZ4FrameBatch
	{
		private const uint Magic = 0x184D2204;

		/// <summary>One LZ4 frame (independent blocks) around content, byte-identical to what LZ4FrameWriter writes for the same
		/// settings when its input arrives in one piece.</summary>
		public static byte[] Encode(ReadOnlySpan<byte> content, LZ4EncoderSettings settings)
		{
			if (settings.ChainBlocks) throw new NotSupportedException("chained blocks are encoded one after the other: use LZ4FrameWriter");
			var blockSize = MaxBlockSize(settings.BlockSize, out var bdCode);
			var n = (int) (((long) content.Length + blockSize - 1) / blockSize);
			var slot = LZ4Codec.MaximumOutputSize(blockSize);
			var arena = new byte[Math.Max(1, (long) n * slot)];
			var encoded = new int[n];
			using (var encoder = new LZ4BlockEncoder(settings.CompressionLevel, blockSize))
				encoder.EncodeBlocks(content, arena, encoded, allowCopy: true);

			// sizes: header 7 (+8 content length), per block 4 + stored (+4), EndMark 4 (+4)
			long total = 7 + (settings.ContentLength.HasValue ? 8 : 0) + 4 + (settings.ContentChecksum ? 4 : 0);
			for (var i = 0; i < n; i++) total += 4 + Math.Abs(encoded[i]) + (settings.BlockChecksum ? 4 : 0);
			var frame = new byte[total];
			var at = 0;
			BinaryPrimitives.WriteUInt32LittleEndian(frame.AsSpan(at), Magic); at += 4;
			var headerStart = at;
			frame[at++] = (byte) ((1 << 6) | (1 << 5) /* independent */ | (settings.BlockChecksum ? 1 << 4 : 0) |
				(settings.ContentLength.HasValue ? 1 << 3 : 0) | (settings.ContentChecksum ? 1 << 2 : 0));
			frame[at++] = (byte) (bdCode << 4);
			if (settings.ContentLength.HasValue) { BinaryPrimitives.WriteUInt64LittleEndian(frame.AsSpan(at), (ulong) settings.ContentLength.Value); at += 8; }
			frame[at] = (byte) (Digest(frame.AsSpan(headerStart, at - headerStart)) >> 8); at++;          // LZ4FrameWriter.cs:100

			// payloads first, then every block checksum of the frame in one XXH32 launch
			var payloadAt = new ulong[n]; var payloadLen = new ulong[n];
			for (var i = 0; i < n; i++)
			{
				var stored = Math.Abs(encoded[i]);
				BinaryPrimitives.WriteUInt32LittleEndian(frame.AsSpan(at), (uint) stored | (encoded[i] < 0 ? 0x80000000u : 0u)); at += 4;   // :159-160
				arena.AsSpan(i * slot, stored).CopyTo(frame.AsSpan(at));
				payloadAt[i] = (ulong) at; payloadLen[i] = (ulong) stored; at += stored;
				if (settings.BlockChecksum) at += 4;
			}
			if (settings.BlockChecksum && n > 0)
			{
				var digests = Digests(frame, payloadAt, payloadLen);
				for (var i = 0; i < n; i++) BinaryPrimitives.WriteUInt32LittleEndian(frame.AsSpan((int) (payloadAt[i] + payloadLen[i])), digests[i]);
			}
			BinaryPrimitives.WriteUInt32LittleEndian(frame.AsSpan(at), 0); at += 4;                          // EndMark
			if (settings.ContentChecksum) { BinaryPrimitives.WriteUInt32LittleEndian(frame.AsSpan(at), Digest(content)); at += 4; }
			return frame;
		}

		/// <summary>Decodes one frame (independent or chained blocks) into a fresh array; InvalidDataException where LZ4FrameReader throws
		/// (bad magic / version / header checksum, a block or content checksum that does not match, a block that does not decode).</summary>
		public static byte[] Decode(ReadOnlySpan<byte> frame)
		{
			// The header walk is LZ4FrameReader.async.cs:52-105 (index arithmetic on the host); the block table it yields --
			// offsets, stored lengths with the raw bit, at most MaxBlockSize bytes each -- goes to LZ4BlockDecoder.DecodeBlocks
			// (independent blocks) or k4lz4_decode_chain_batch (chained), the checksums to one k4lz4_xxh32_batch call, exactly as
			// k4os/compression/lz4_amd/frames.py does in the Python mirror of this file (LZ4Frame.Decode / DecodeBatch), which is
			// what the parity tests run.  The slot of a block is min(MaxBlockSize, 255 * stored + 32): a header is never trusted
			// for memory.
			throw new NotImplementedException("see frames.py: LZ4Frame.Decode -- the same walk, to be transcribed when a C# toolchain is at hand");
		}

		private static int MaxBlockSize(int requested, out int bdCode)
		{   // LZ4FrameWriter.cs:176-189
			if (requested <= 1 << 16) { bdCode = 4; return 1 << 16; }
			if (requested <= 1 << 18) { bdCode = 5; return 1 << 18; }
			if (requested <= 1 << 20) { bdCode = 6; return 1 << 20; }
			bdCode = 7; return 1 << 22;
		}

		private static uint Digest(ReadOnlySpan<byte> bytes)
		{
			var off = new ulong[] { 0 }; var len = new ulong[] { (ulong) bytes.Length };
			var pad = bytes.Length == 0 ? new byte[1] : bytes.ToArray();
			return Digests(pad, off, len)[0];
		}

		private static uint[] Digests(byte[] data, ulong[] off, ulong[] len)
		{
			var digests = new uint[off.Length];
			using var lease = NativeContext.Rent();
			fixed (byte* d = data)
			fixed (ulong* o = off, l = len)
			fixed (uint* r = digests)
				LLNative.ThrowIfFailed(LLNative.k4lz4_xxh32_batch(lease.Handle, d, o, l, r, off.Length, 0), lease.Handle);
			return digests;
		}
	}
}
