// Engine/NativeContext.cs -- k4lz4_ctx handles for managed callers (include/k4lz4.h: "a k4lz4_ctx ... may be used by one host
// thread at a time; different contexts are independent").  A context owns device scratch, pinned staging buffers and a few
// helper threads, so they are POOLED, not made per thread: a call rents one for its duration and gives it back; the pool grows
// to the number of calls that are in flight at once and keeps at most MaxIdle of them.  Compile-unverified.
using System;
using System.Collections.Concurrent;
using System.Threading;

namespace K4os.Compression.LZ4.Engine
{
	internal sealed class NativeContext: IDisposable
	{
		private static readonly ConcurrentBag<NativeContext> Idle = new ConcurrentBag<NativeContext>();
		private static int _idleCount;

		/// <summary>Idle contexts kept for reuse; more than this are destroyed when they come back.</summary>
		public static int MaxIdle { get; set; } = 4;

		private IntPtr _handle;

		private NativeContext(int device)
		{
			var status = LLNative.k4lz4_ctx_create(out _handle, device);
			if (status != LLNative.OK)
			{
				_handle = IntPtr.Zero;
				LLNative.ThrowIfFailed(status, IntPtr.Zero); // E_NO_DEVICE -> PlatformNotSupportedException: no CPU fallback behind this arm
			}
		}

		/// <summary>A context for the duration of one call: <c>using var lease = NativeContext.Rent();</c></summary>
		public static Lease Rent()
		{
			if (Idle.TryTake(out var ctx)) Interlocked.Decrement(ref _idleCount);
			else ctx = new NativeContext(-1);
			return new Lease(ctx);
		}

		public readonly struct Lease: IDisposable
		{
			private readonly NativeContext _ctx;
			internal Lease(NativeContext ctx) => _ctx = ctx;
			public IntPtr Handle => _ctx._handle;
			public int Device => LLNative.k4lz4_ctx_device(_ctx._handle);

			public void Dispose()
			{
				if (Interlocked.Increment(ref _idleCount) <= MaxIdle) Idle.Add(_ctx);
				else { Interlocked.Decrement(ref _idleCount); _ctx.Dispose(); }
			}
		}

		public void Dispose()
		{
			var h = Interlocked.Exchange(ref _handle, IntPtr.Zero);
			if (h != IntPtr.Zero) LLNative.k4lz4_ctx_destroy(h);
			GC.SuppressFinalize(this);
		}

		~NativeContext() => Dispose();

		/// <summary>Destroys every idle context (host shutdown hook).</summary>
		public static void DisposeAll()
		{
			while (Idle.TryTake(out var c)) { Interlocked.Decrement(ref _idleCount); c.Dispose(); }
		}
	}
}
