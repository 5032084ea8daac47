// oracle/ref_entry.cpp -- C entry points of oracle/_ref/libk4ref.so (test infrastructure).
// The bodies of the engine are NOT here: k4ref_engine.hpp is generated from /root/reference (into a scratch directory) by
// oracle/make_ref.py at build time.  These wrappers are the `Algorithm.X64` / `Algorithm.X32` arms of
// Engine/LLxx.cs:17-103 (which is managed `switch`-expression code the translator does not take) with
// the engine chosen by the caller instead of by LL.Enforce32.
#include "ref_prelude.hpp"
#include "k4ref_engine.hpp"

using namespace k4ref;
#define K4REF_API extern "C" __attribute__((visibility("default")))

K4REF_API int k4ref_compress_bound(int n) { return LL::LZ4_compressBound(n); }                                         // LL.tools.cs:38-40
// LLxx.cs:65-75
K4REF_API int k4ref_compress_fast(const uint8_t* s, uint8_t* d, int n, int cap, int acc) { return LL64::LZ4_compress_fast((byte*) s, d, n, cap, acc); }
K4REF_API int k4ref_compress_fast_x32(const uint8_t* s, uint8_t* d, int n, int cap, int acc) { return LL32::LZ4_compress_fast((byte*) s, d, n, cap, acc); }
// LLxx.cs:94-103
K4REF_API int k4ref_compress_hc(const uint8_t* s, uint8_t* d, int n, int cap, int level) { return LL64::LZ4_compress_HC((byte*) s, d, n, cap, level); }
K4REF_API int k4ref_compress_hc_x32(const uint8_t* s, uint8_t* d, int n, int cap, int level) { return LL32::LZ4_compress_HC((byte*) s, d, n, cap, level); }
// LLxx.cs:17-26
K4REF_API int k4ref_decompress_safe(const uint8_t* s, uint8_t* d, int n, int cap) { return LL64::LZ4_decompress_safe((byte*) s, d, n, cap); }
K4REF_API int k4ref_decompress_safe_x32(const uint8_t* s, uint8_t* d, int n, int cap) { return LL32::LZ4_decompress_safe((byte*) s, d, n, cap); }
// LLxx.cs:28-39
K4REF_API int k4ref_decompress_safe_partial(const uint8_t* s, uint8_t* d, int n, int target, int cap) { return LL64::LZ4_decompress_safe_partial((byte*) s, d, n, target, cap); }
// LLxx.cs:41-51
K4REF_API int k4ref_decompress_safe_using_dict(const uint8_t* s, uint8_t* d, int n, int cap, const uint8_t* dict, int dictLen) {
	return LL64::LZ4_decompress_safe_usingDict((byte*) s, d, n, cap, (byte*) dict, dictLen); }

// ---- the envelope's header arithmetic: LZ4Pickler's own private helpers (the reference's statements, respelled), called the way
// its two Pickle bodies call them.  No LZ4 block is encoded here: `encodedLength` is an argument, so the whole (sourceLength,
// encodedLength) plane can be swept.  A managed exception comes back as -(kind) (ref_prelude.hpp K4RefManaged).
K4REF_API int k4ref_pickle_effective_size_of(int value) { return LZ4Pickler::EffectiveSizeOf(value); }        // LZ4Pickler.pickle.cs:224-225
K4REF_API int k4ref_pickle_encode_size_of(int size) { return LZ4Pickler::EncodeSizeOf(size); }               // :227-228
K4REF_API int k4ref_pickle_header_byte_v0(int sizeOfDiff) { return LZ4Pickler::EncodeHeaderByteV0(sizeOfDiff); }   // :221-222
// writer == 0: PickleWithBuffer (LZ4Pickler.pickle.cs:85-105); writer == 1: Pickle<TBufferWriter> (:128,:135-148).  `out` holds
// `cap` bytes (the reference's span is headerSize + the payload).  Returns the header's length.
K4REF_API int k4ref_pickle_header(int writer, int version, int sourceLength, int encodedLength, uint8_t* out, int cap) {
	try {
		Span<byte> target(out, cap);
		if (encodedLength <= 0 || encodedLength >= sourceLength) {                                                // :85, :135
			int headerSize = LZ4Pickler::GetUncompressedHeaderSize(version, sourceLength);                      // :87
			int offset = LZ4Pickler::EncodeUncompressedHeader(target, version, sourceLength);                   // :90, :137
			Debug::Assert(writer || headerSize == offset);                                                          // :91
			return offset;
		}
		int headerSize = writer ? LZ4Pickler::GetPessimisticHeaderSize(version, sourceLength)                   // :128
		                        : LZ4Pickler::GetCompressedHeaderSize(version, sourceLength, encodedLength);    // :97
		int offset = LZ4Pickler::EncodeCompressedHeader(target, version, headerSize, sourceLength, encodedLength);  // :100-101, :143-144
		Debug::Assert(headerSize == offset);                                                                    // :102, :145
		return offset;
	} catch (const K4RefManaged& e) { return -e.kind; }
}
// LZ4Pickler.unpickle.cs:131-148 (DecodeHeader): out = { DataOffset, ResultLength, IsCompressed, Flags }
K4REF_API int k4ref_unpickle_header(const uint8_t* src, int len, int32_t* out) {
	try {
		PickleHeader h = LZ4Pickler::DecodeHeader(ReadOnlySpan<byte>(src, len));
		out[0] = h.DataOffset; out[1] = h.ResultLength; out[2] = h.IsCompressed() ? 1 : 0; out[3] = h.Flags;
		return 0;
	} catch (const K4RefManaged& e) { return -e.kind; }
}

K4REF_API const char* k4ref_inputs_sha256() { return K4REF_INPUTS_SHA256; }

// ---- threaded batch drivers: bench.py's cpu_baseline (kind "reference") times the reference's own engine on the host cores.
// The per-block mapping is LZ4Codec's (LZ4Codec.cs:40-52,104-115: empty -> 0, <= 0 -> -1); static partition over T threads.
#include <pthread.h>
namespace {
struct Job { int op; const uint8_t* src; const uint64_t* so; const int32_t* sl; uint8_t* dst; const uint64_t* dof; const int32_t* dc; int32_t* out; int64_t lo, hi; int level; };
void* worker(void* p) {
	Job* j = (Job*) p;
	for (int64_t i = j->lo; i < j->hi; i++) {
		byte* s = (byte*) j->src + j->so[i]; byte* d = j->dst + j->dof[i];
		int r;
		if (j->sl[i] <= 0) r = 0;
		else if (j->op == 0) r = j->level < 3 ? LL64::LZ4_compress_fast(s, d, j->sl[i], j->dc[i], 1) : LL64::LZ4_compress_HC(s, d, j->sl[i], j->dc[i], j->level);
		else r = LL64::LZ4_decompress_safe(s, d, j->sl[i], j->dc[i]);
		j->out[i] = j->sl[i] <= 0 ? 0 : (r <= 0 ? -1 : r);
	}
	return nullptr;
}
int run(int op, const uint8_t* src, const uint64_t* so, const int32_t* sl, uint8_t* dst, const uint64_t* dof, const int32_t* dc, int32_t* out, int64_t n, int level, int threads) {
	if (threads < 1) threads = 1;
	if (threads > 256) threads = 256;
	pthread_t tid[256]; Job jobs[256]; bool started[256];
	for (int t = 0; t < 256; t++) started[t] = true;
	for (int t = 0; t < threads; t++) {
		jobs[t] = Job{op, src, so, sl, dst, dof, dc, out, n * t / threads, n * (t + 1) / threads, level};
		if (threads == 1) worker(&jobs[t]);
		/* (a thread that cannot be started: its range is done here, and the threads already running are still joined below --
		 * jobs[] and tid[] live on this stack) */
		else if (pthread_create(&tid[t], nullptr, worker, &jobs[t]) != 0) { worker(&jobs[t]); started[t] = false; }
	}
	if (threads > 1) for (int t = 0; t < threads; t++) if (started[t]) pthread_join(tid[t], nullptr);
	return 0;
}
}
K4REF_API int k4ref_encode_batch(const uint8_t* src, const uint64_t* so, const int32_t* sl, uint8_t* dst, const uint64_t* dof, const int32_t* dc, int32_t* out, int64_t n, int level, int threads) {
	return run(0, src, so, sl, dst, dof, dc, out, n, level, threads); }
K4REF_API int k4ref_decode_batch(const uint8_t* src, const uint64_t* so, const int32_t* sl, uint8_t* dst, const uint64_t* dof, const int32_t* dc, int32_t* out, int64_t n, int threads) {
	return run(1, src, so, sl, dst, dof, dc, out, n, 0, threads); }
