// Engine/NativeContext.cs -- one k4lz4_ctx per managed thread (include/k4lz4.h: "a k4lz4_ctx ... may be used by one host
// thread at a time; different contexts are independent").  Compile-unverified.
using System;
using System.Threading;

namespace K4os.Compression.LZ4.Engine
{
	internal sealed class NativeContext: IDisposable
	{
		// ThreadLocal with trackAllValues so that AppDomain/process shutdown can dispose what threads left behind
		private static readonly ThreadLocal<NativeContext> PerThread =
			new ThreadLocal<NativeContext>(() => new NativeContext(-1), trackAllValues: true);

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

		/// <summary>The calling thread's context on its current HIP device (created on first use).</summary>
		public static IntPtr Current => PerThread.Value._handle;

		/// <summary>GPU the calling thread's context is bound to.</summary>
		public static int Device => LLNative.k4lz4_ctx_device(Current);

		public void Dispose()
		{
			var h = Interlocked.Exchange(ref _handle, IntPtr.Zero);
			if (h != IntPtr.Zero) LLNative.k4lz4_ctx_destroy(h);
			GC.SuppressFinalize(this);
		}

		~NativeContext() => Dispose();

		/// <summary>Disposes every context created so far (host shutdown hook).</summary>
		public static void DisposeAll()
		{
			foreach (var c in PerThread.Values) c.Dispose();
		}
	}
}
