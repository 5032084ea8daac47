// oracle/ref_entry.cpp -- C entry points of oracle/_ref/libk4ref.so (test infrastructure).
// The bodies of the engine are NOT here: k4ref_engine.hpp is generated from /root/reference by
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
K4REF_API const char* k4ref_inputs_sha256() { return K4REF_INPUTS_SHA256; }
