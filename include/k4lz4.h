/*
 * k4lz4.h -- C ABI of libk4lz4.so: the MI355X (gfx950) LZ4 block codec that sits behind the
 * K4os.Compression.LZ4 block API.  Plain pointers and sizes only; no HIP or torch types.
 *
 * Every entry point names the reference interface (paths relative to the reference repository,
 * src/K4os.Compression.LZ4/...) it replaces.  INTEGRATION.md shows the C# P/Invoke stubs and
 * the `Algorithm.Native` arm a maintainer would add to Engine/LLxx.cs.
 *
 * Conventions
 *   - All block lengths are `int32_t` like the reference (`int`), offsets into batch buffers are
 *     `uint64_t`.  Input size limit: 0x7E000000 (Engine/LL.types.cs:19).
 *   - Per-block results in `outLen[i]` follow LZ4Codec (LZ4Codec.cs:40-52, :104-115):
 *       > 0 bytes written, 0 for an empty input, -1 on failure (output too small / corrupt input).
 *     With K4LZ4_FLAG_RAW_RETURN they are the LLxx-level returns instead (Engine/LLxx.cs:17-26,
 *     :65-75): bytes written, 0 = did not fit, decode error = -(input position) - 1.
 *   - Bytes of dst[i] beyond outLen[i] are never modified on success
 *     (src/K4os.Compression.LZ4.Tests/SpanTests.cs:36-44).
 *   - Call-level return: K4LZ4_OK or a negative k4lz4_status; text via k4lz4_last_error().
 *     The library never throws, aborts or calls back.  No pointer is retained after return
 *     (the *_device calls return after enqueueing on the given stream; the buffers must stay
 *     valid until that stream work completes).
 *   - There is NO CPU fallback: without a usable gfx950 device every compute entry point fails
 *     with K4LZ4_E_NO_DEVICE.
 *   - A k4lz4_ctx is bound to one GPU and may be used by one host thread at a time; different
 *     contexts are independent (reentrant like the reference's static API).  The context owns device
 *     scratch (dispatch order, hash tables, HC work areas) that every call reuses: calls on ONE context
 *     are therefore serialised on the device even when they are given different streams (the second
 *     call's stream waits for the first call's work); use one context per stream for concurrency.
 *   - Trouble that is not a property of a block's data (a decoder wave pair timing out on its partner,
 *     an HC scratch reservation that was too small) is never reported through outLen alone: the blocks
 *     concerned say "failed" AND the next synchronising call on the context (every host-pointer call,
 *     k4lz4_synchronize) returns K4LZ4_E_HIP / K4LZ4_E_NOMEM with the reason(s) in k4lz4_last_error().
 *     The report is kept per CONTEXT (a word of device memory the context owns and hands to its kernels):
 *     contexts that share a device -- one per managed thread in the .NET shim -- never see or clear each
 *     other's.  This holds for pickles too: an HC pickle that was not encoded for want of reserved scratch
 *     has outLen = -1, never a valid raw envelope.
 */
#ifndef K4LZ4_H
#define K4LZ4_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define K4LZ4_API __attribute__((visibility("default")))
#define K4LZ4_VERSION 100 /* 0.1.0 */

typedef struct k4lz4_ctx k4lz4_ctx;

enum k4lz4_status {
    K4LZ4_OK = 0,
    K4LZ4_E_HIP = -1,         /* HIP runtime error, see k4lz4_last_error */
    K4LZ4_E_ARG = -2,         /* NULL pointer / negative count (the C# shim raises ArgumentException) */
    K4LZ4_E_NOMEM = -3,
    K4LZ4_E_NO_DEVICE = -4,   /* no gfx950 device visible */
    K4LZ4_E_UNSUPPORTED = -5  /* reserved: every LZ4Level is implemented (levels above L12_MAX behave as L12_MAX, LL64.high.cs:1160) */
};

/* LZ4Level.cs:6-39 -- the numeric value is part of the ABI */
enum k4lz4_level {
    K4LZ4_L00_FAST = 0,
    K4LZ4_L03_HC = 3, K4LZ4_L04_HC = 4, K4LZ4_L05_HC = 5, K4LZ4_L06_HC = 6, K4LZ4_L07_HC = 7,
    K4LZ4_L08_HC = 8, K4LZ4_L09_HC = 9, K4LZ4_L10_OPT = 10, K4LZ4_L11_OPT = 11, K4LZ4_L12_MAX = 12
};

enum k4lz4_flags {
    K4LZ4_FLAG_RAW_RETURN = 1,    /* outLen = LLxx-level return values */
    K4LZ4_FLAG_PICKLE_WRITER = 2, /* LZ4Pickler IBufferWriter path header rule (LZ4Pickler.pickle.cs:113-158) */
    K4LZ4_FLAG_NO_REORDER = 4,    /* encode/pickle/unpickle: dispatch blocks in index order instead of most-expensive-first */
    K4LZ4_FLAG_REORDER = 8,       /* decode: dispatch longest inputs first (useful for ragged batches; unpickle does it by default) */
    K4LZ4_FLAG_NO_SPLIT = 16,     /* encode: do not run part of the batch on the global-memory-table kernel */
    K4LZ4_FLAG_ALLOW_COPY = 64,   /* encode: LZ4EncoderBase.Encode(allowCopy) -- a block that does not shrink is stored raw, outLen = -srcLen;
                                     outLen = 0 where the reference throws "target buffer too small" (Encoders/LZ4EncoderBase.cs:66-88) */
    K4LZ4_FLAG_X32 = 128,         /* fast encode / pickle: the 32-bit engine's bytes (LZ4Codec.Enforce32, LZ4Codec.cs:14-25): inputs of
                                     64 KiB and more are hashed with LZ4_hash4 instead of LZ4_hash5 (x32/LL32.tools.cs:141-148) */
    K4LZ4_FLAG_PARTIAL = 32,      /* decode: LZ4Codec.PartialDecode -- stop once dstCap[i] bytes are produced (LZ4Codec.cs:123-173) */
    K4LZ4_FLAG_SEGMENTS = 256     /* fast encode: blocks of 1.5 MiB and more that are also a large share of the batch may be encoded by several
                                     wavefronts (segments whose joints are verified; a block that does not verify is encoded again by one), see
                                     DESIGN.md 4.6 -- same bytes, a 4 MiB block in 50 ms instead of 165; needs dstCap[i] >= srcLen[i] - 1 and
                                     may write anywhere inside a block's slot before outLen is final.  The pickle calls do this by themselves. */
};

K4LZ4_API int k4lz4_version(void);
K4LZ4_API int k4lz4_device_count(void);

/* Below which batch size a caller should stay on the managed engine (LZ4Codec.cs:40-52 -> LLxx -> LL64 on the host).
 * One wavefront encodes a 64 KiB block in about 2 ms and decodes it in 0.6 ms however small the batch is (the chip is fast
 * because it runs thousands of blocks side by side, not because a block is fast), so a device call has a floor; a host that
 * sustains `hostGiBs` on this work beats it below
 *     floor(kind, blockBytes) * hostGiBs / blockBytes   blocks.
 * kind: 0 fast-level encode, 1 decode, 2 HC encode (level 3); blockBytes: uncompressed bytes per block;
 * hostGiBs: what the caller's host threads sustain together on such blocks (the reference's engine, measured on the 256-thread
 * host of the MI355X box: 0.8 GiB/s encode / 4 GiB/s decode per thread, 32 / 35 GiB/s with all threads); <= 0: those box figures.
 * For data that starts and ends in host memory use the host-pointer rate (about 22 GiB/s encode, 28-37 decode) as the device's
 * ceiling as well: a host faster than that never gains.  Returns the number of blocks (>= 1), or a negative error. */
K4LZ4_API int64_t k4lz4_recommended_min_batch(int kind, int32_t blockBytes, double hostGiBs);

/* device < 0: the calling thread's current HIP device */
K4LZ4_API int k4lz4_ctx_create(k4lz4_ctx **out, int device);
K4LZ4_API void k4lz4_ctx_destroy(k4lz4_ctx *ctx);
/* ctx may be NULL: last error of the calling thread's implicit context / of ctx creation */
K4LZ4_API const char *k4lz4_last_error(const k4lz4_ctx *ctx);
K4LZ4_API int k4lz4_ctx_device(const k4lz4_ctx *ctx);
/* blocks until everything this ctx enqueued on `stream` (NULL = default stream) has finished; returns the call-level
 * status of that work (see "Trouble that is not a property of a block's data" above) */
K4LZ4_API int k4lz4_synchronize(k4lz4_ctx *ctx, void *stream);
/* HC levels (L03_HC and up) need work areas proportional to the batch: 20 bytes per input byte (36 before round 6).  A device-resident call
 * does not know its batch's size on the host, so by default it reads it back -- ONE synchronisation per 4096 blocks.
 * After this call, device-resident HC encodes / pickles on ctx whose blocks total at most totalSrcBytes (per call) and are
 * at most longestBlock bytes each only enqueue, like every other *_device call; a batch that exceeds the reservation is
 * not encoded (outLen = failure, K4LZ4_E_NOMEM at the next synchronising call).  (0, 0) removes the reservation. */
K4LZ4_API int k4lz4_ctx_reserve_hc(k4lz4_ctx *ctx, int64_t totalSrcBytes, int32_t longestBlock);
/* Page-locks [ptr, ptr + bytes) of the caller's memory (hipHostRegister) and remembers the range: host-pointer batch calls
 * whose source span, or whose destination slots, lie inside a registered range move those bytes straight between the
 * caller's pages and the GPU instead of through the context's pinned staging buffers (no counterpart in the reference; a
 * .NET caller registers a pinned array / NativeMemory block it reuses across calls).  Results are the same byte for byte:
 * exactly outLen[i] bytes are written to slot i either way.  Registering costs about as much as the copy it saves, so it
 * pays for buffers that are used more than once.  The range must stay registered until every call that uses it has
 * returned; unregister with the same ptr.  Process-wide, any thread. */
K4LZ4_API int k4lz4_host_register(void *ptr, size_t bytes);
K4LZ4_API int k4lz4_host_unregister(void *ptr);
/* Diagnostic, no counterpart in the reference: the three serial chains of the kernels exist as hand-written scalar ISA and
 * as C (the form the CPU wave emulator of the test suite runs).  Runs both forms on the device over `waves` x `rounds`
 * pseudo-random well-formed inputs; mismatches[0..2] = rounds in which they disagreed (token chain of the decoder, hop
 * chain of the fast encoder, its variant with pair fall-backs).  All zero on a healthy build. */
K4LZ4_API int k4lz4_selftest_chains(k4lz4_ctx *ctx, int waves, int rounds, uint32_t seed, uint32_t mismatches[3]);

/* LZ4Codec.MaximumOutputSize (LZ4Codec.cs:30-31) == LL.LZ4_compressBound (Engine/LL.tools.cs:38-40).
 * Pure host arithmetic. */
/* LZ4Codec.Enforce32 (process-wide, like LL.Enforce32): every fast-level encode and pickle behaves as with K4LZ4_FLAG_X32 */
K4LZ4_API void k4lz4_set_enforce32(int on);
K4LZ4_API int k4lz4_get_enforce32(void);
K4LZ4_API int k4lz4_compress_bound(int n);

/* ---- per-block entry points: the Engine/LLxx.cs seam, same argument order and returns ------
 * (host pointers; each call is a batch of one on the calling thread's implicit context) */
/* These mirror the reference's `int` returns, which cannot carry an infrastructure failure (no
 * GPU, HIP error, unsupported level): such a failure returns 0 (encode) / -1 (decode) and sets the
 * thread's status, which the host shim must check: K4LZ4_OK or a negative k4lz4_status. */
K4LZ4_API int k4lz4_last_status(void);
/* LLxx.LZ4_compress_fast (Engine/LLxx.cs:65-75) */
K4LZ4_API int k4lz4_compress_fast(const uint8_t *src, uint8_t *dst, int srcLen, int dstCap, int acceleration);
/* LLxx.LZ4_compress_HC (Engine/LLxx.cs:94-103); level >= K4LZ4_L03_HC (LZ4Level has nothing between FAST and L03_HC, and
 * LZ4Codec routes lower values to the fast encoder, LZ4Codec.cs:48-50): lower values fail with K4LZ4_E_ARG */
K4LZ4_API int k4lz4_compress_hc(const uint8_t *src, uint8_t *dst, int srcLen, int dstCap, int level);
/* LLxx.LZ4_decompress_safe (Engine/LLxx.cs:17-26) */
K4LZ4_API int k4lz4_decompress_safe(const uint8_t *src, uint8_t *dst, int srcLen, int dstCap);
/* LLxx.LZ4_decompress_safe_partial (Engine/LLxx.cs:29-39): decoding stops at targetLen bytes */
K4LZ4_API int k4lz4_decompress_safe_partial(const uint8_t *src, uint8_t *dst, int srcLen, int targetLen);
/* LLxx.LZ4_decompress_safe_usingDict (Engine/LLxx.cs:41-55; LL64.dec.cs:523-546) behind
 * LZ4Codec.Decode(source, target, dictionary) (LZ4Codec.cs:144-160).  A dictionary that ends exactly at dst is
 * decoded with the reference's prefix semantics, any other one with its external-dictionary semantics. */
K4LZ4_API int k4lz4_decompress_safe_using_dict(const uint8_t *src, uint8_t *dst, int srcLen, int dstCap,
                                               const uint8_t *dict, int dictLen);

/* ---- batches of independent blocks (what LZ4Codec.Encode / Decode callers loop over) --------
 * block i: input  src + srcOff[i], srcLen[i] bytes;  output dst + dstOff[i], dstCap[i] bytes.
 * Host variants take host pointers and stage through the GPU; they return when done.      */
K4LZ4_API int k4lz4_encode_batch(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                                 uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen,
                                 int64_t n, int level, int flags);
K4LZ4_API int k4lz4_decode_batch(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                                 uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen,
                                 int64_t n, int flags);

/* Device-resident variants: every pointer (including srcOff/srcLen/dstOff/dstCap/outLen) is a
 * device pointer on ctx's GPU; `stream` is a hipStream_t (NULL = default stream).  Asynchronous. */
K4LZ4_API int k4lz4_encode_batch_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff,
                                        const int32_t *srcLen, uint8_t *dst, const uint64_t *dstOff,
                                        const int32_t *dstCap, int32_t *outLen, int64_t n, int level, int flags,
                                        void *stream);
K4LZ4_API int k4lz4_decode_batch_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff,
                                        const int32_t *srcLen, uint8_t *dst, const uint64_t *dstOff,
                                        const int32_t *dstCap, int32_t *outLen, int64_t n, int flags, void *stream);

/* Batched LZ4Codec.Decode(source, target, dictionary): block i is decoded against
 * dict[dictOff[i] .. dictOff[i]+dictLen[i]) (dictLen[i] <= 0: no dictionary).  Chained blocks of one
 * stream (LZ4ChainDecoder's layout: each block's dictionary is the previous 64 KiB of output) are NOT a
 * batch -- they depend on each other; independent streams each holding a dictionary are. */
K4LZ4_API int k4lz4_decode_dict_batch(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                                      uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen,
                                      int64_t n, int flags, const uint8_t *dict, const uint64_t *dictOff,
                                      const int32_t *dictLen);
K4LZ4_API int k4lz4_decode_dict_batch_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff,
                                             const int32_t *srcLen, uint8_t *dst, const uint64_t *dstOff,
                                             const int32_t *dstCap, int32_t *outLen, int64_t n, int flags,
                                             const uint8_t *dict, const uint64_t *dictOff, const int32_t *dictLen,
                                             void *stream);

/* ---- LZ4Pickler envelope, version 0 (LZ4Pickler.pickle.cs:51-228, LZ4Pickler.unpickle.cs:18-158)
 * pickle:   outLen[i] = envelope bytes written to dst + dstOff[i]; dstCap[i] must be at least
 *           k4lz4_pickle_bound(srcLen[i]); 0 for an empty message (Pickle returns an empty array).
 * unpickle: dstCap[i] must equal the size k4lz4_unpickle_size reports (the reference allocates
 *           exactly that); outLen[i] = that size, or -1 where the reference throws
 *           InvalidDataException (bad version, short header, size mismatch, corrupt block).  */
K4LZ4_API int k4lz4_pickle_bound(int srcLen);
/* host arithmetic on one envelope: unpickled size, or -1 when the header is corrupt */
K4LZ4_API int k4lz4_unpickle_size(const uint8_t *pickle, int pickleLen);
K4LZ4_API int k4lz4_pickle_batch(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                                 uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen,
                                 int64_t n, int level, int flags);
K4LZ4_API int k4lz4_unpickle_batch(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                                   uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen,
                                   int64_t n, int flags);
K4LZ4_API int k4lz4_pickle_batch_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff,
                                        const int32_t *srcLen, uint8_t *dst, const uint64_t *dstOff,
                                        const int32_t *dstCap, int32_t *outLen, int64_t n, int level, int flags,
                                        void *stream);
K4LZ4_API int k4lz4_unpickle_batch_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff,
                                          const int32_t *srcLen, uint8_t *dst, const uint64_t *dstOff,
                                          const int32_t *dstCap, int32_t *outLen, int64_t n, int flags,
                                          void *stream);
/* device-side k4lz4_unpickle_size for a whole batch: outLen[i] = unpickled size or -1 */
K4LZ4_API int k4lz4_unpickle_sizes_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff,
                                          const int32_t *srcLen, int32_t *outLen, int64_t n, void *stream);

/* ---- diagnostics -----------------------------------------------------------------------------
 * Runs the encode (decode = 0, L00_FAST, blocks < 65547 B) or decode (decode = 1) kernel's
 * instrumented twin on a device-resident batch and fills `counters` (device pointer, 16 x uint64
 * per block): [0] total shader cycles, [1..3] cycles per phase (encode: probe / extend / emit;
 * decode: parse / literals / matches), [4..7] event counts, [8]/[9] start/end on the 100 MHz
 * real-time counter, [10] HW_ID of the executing wave.  Results in dst/outLen
 * are identical to the normal kernels; timing is perturbed (each phase drains its memory traffic).
 * decode = 4 (encode) / 5 (decode): the ORDINARY kernels run and only record, per block, [8] start and [9] end on the
 * real-time counter, [10] HW_ID and -- encode -- [11] which kernel took the block (1 LDS table, 2 global table, 3 the
 * 28-known-bytes variant; 4 / 5 / 6 the two-step encoder of k4lz4_parse.hpp: table in LDS, in memory, moved into LDS on the way):
 * when each block of a batch really starts and ends (scripts/stamp_probe.py).
 * NOTE: in a default build, decode = 0 (encode phases) instruments the ONE-KERNEL encoder of rounds 1-4, not the two-step encoder
 * that LZ4Codec.Encode batches ship through; its phase probe is a build with -DK4_PARSE_PROF (scripts/parse_probe.py). */
K4LZ4_API int k4lz4_profile_batch_device(k4lz4_ctx *ctx, int decode, const uint8_t *src, const uint64_t *srcOff,
                                         const int32_t *srcLen, uint8_t *dst, const uint64_t *dstOff,
                                         const int32_t *dstCap, int32_t *outLen, int64_t n, uint64_t *counters,
                                         void *stream);

/* ---- frame layer (K4os.Compression.LZ4.Streams: Frames/LZ4FrameWriter.cs, LZ4FrameReader.async.cs) ----------- */

/* XXH32.DigestOf for n buffers (K4os.Hash.xxHash 1.0.8, NuGet; call sites Frames/LZ4FrameWriter.cs:100,:162-182,
 * Internal/Stash.cs:149-150): out[i] = xxHash32(data[off[i] .. off[i]+len[i]), seed). */
K4LZ4_API int k4lz4_xxh32_batch(k4lz4_ctx *ctx, const uint8_t *data, const uint64_t *off, const uint64_t *len,
                                uint32_t *out, int64_t n, uint32_t seed);
K4LZ4_API int k4lz4_xxh32_batch_device(k4lz4_ctx *ctx, const uint8_t *data, const uint64_t *off, const uint64_t *len,
                                       uint32_t *out, int64_t n, uint32_t seed, void *stream);

/* Block streams decoded in order, one stream per wavefront: ILZ4Decoder.Decode / Inject over the blocks of a frame
 * (Frames/LZ4FrameReader.async.cs:108-136; Encoders/LZ4BlockDecoder.cs:39-71 for independent blocks,
 * Encoders/LZ4ChainDecoder.cs -> LL64.LZ4_decompress_safe_continue for chained ones).  Stream s owns blocks
 * firstBlk[s] .. firstBlk[s]+nBlk[s]-1; blkLen has bit 31 set for blocks stored raw (LZ4FrameWriter.cs:159-160).
 * outLen[s] = bytes produced, -6 when a block does not decode, -9 when dstCap[s] is too small. */
K4LZ4_API int k4lz4_decode_chain_batch(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *blkOff, const uint32_t *blkLen,
                                       int64_t nBlocks, const uint64_t *firstBlk, const uint32_t *nBlk,
                                       const int32_t *blockSize, const uint8_t *chained, uint8_t *dst,
                                       const uint64_t *dstOff, const uint64_t *dstCap, int64_t *outLen, int64_t nStreams);
K4LZ4_API int k4lz4_decode_chain_batch_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *blkOff,
                                              const uint32_t *blkLen, const uint64_t *firstBlk, const uint32_t *nBlk,
                                              const int32_t *blockSize, const uint8_t *chained, uint8_t *dst,
                                              const uint64_t *dstOff, const uint64_t *dstCap, int64_t *outLen,
                                              int64_t nStreams, void *stream);

/* Frame writer on device-resident data: after k4lz4_encode_batch_device(..., K4LZ4_FLAG_ALLOW_COPY) and
 * k4lz4_xxh32_batch_device, lays the frames out (Frames/LZ4FrameWriter.cs:57-108 header, LZ4FrameWriter.async.cs:15-27
 * block records, :75-90 EndMark + content checksum).  The caller computes the positions (recOff, frameOff, tailOff) from
 * the block split and the encoded lengths; hdr holds 16 bytes per frame of which hdrLen[f] (FLG, BD [, content size]) are
 * used, hdrSum[f] their XXH32.  frameLen[f] receives each frame's length. */
K4LZ4_API int k4lz4_frame_assemble_device(k4lz4_ctx *ctx, const uint8_t *arena, const uint64_t *slotOff, const int32_t *outLen,
                                          const uint32_t *blkSum, const uint64_t *recOff, int64_t nBlocks, const uint8_t *hdr,
                                          const uint32_t *hdrLen, const uint32_t *hdrSum, const uint64_t *frameOff,
                                          const uint64_t *tailOff, const uint32_t *contentSum, uint8_t *frames,
                                          uint64_t *frameLen, int64_t nFrames, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* K4LZ4_H */
