/*
 * k4lz4_pickle.hpp -- batched LZ4Pickler envelope (version 0) on top of the block kernels.
 *
 * Replaces, for batches of independent messages,
 *   LZ4Pickler.Pickle / PickleWithBuffer   src/K4os.Compression.LZ4/LZ4Pickler.pickle.cs:51-106
 *   LZ4Pickler.Pickle(IBufferWriter)       LZ4Pickler.pickle.cs:113-158  (FLAG_PICKLE_WRITER)
 *   header encode / EffectiveSizeOf        LZ4Pickler.pickle.cs:161-228
 *   LZ4Pickler.Unpickle / UnpickleCore     LZ4Pickler.unpickle.cs:18-129
 *   DecodeHeaderV0 / PeekN                 LZ4Pickler.unpickle.cs:131-158
 *
 * Envelope: byte0 = version(3 bits, 0) | sizeCode << 6, sizeCode 0 = raw, 1/2/3 = 1/2/4-byte LE
 * `diff = U - C`, then the LZ4 block (or the raw bytes).
 *
 * The reference encodes into a scratch buffer of max(U, 1024) (array path) or U (writer path)
 * bytes and stores raw when `encoded <= 0 || encoded >= U`.  Every output-limit test of the block
 * encoder is a lower bound of the final size, so "limited to cap" succeeds exactly when the
 * unlimited size is <= cap, with identical bytes; all paths therefore reduce to: compressed iff
 * C < U.  The kernel encodes straight into the envelope slot behind the widest header with
 * cap = U - 1 and then closes the 0..3 byte gap; no scratch buffer exists.
 */
#pragma once
#include "k4lz4_decode.hpp"
#include "k4lz4_encode_fast.hpp"

namespace k4 {

/* LZ4Pickler.pickle.cs:225-226 */
__device__ __forceinline__ int effective_size_of(int value)
{
    return (value > 0xffff || value < 0) ? 4 : (value > 0xff ? 2 : 1);
}

/* envelope around an already encoded block sitting at dst + 5 (C = encoder result with cap U - 1) */
__device__ __forceinline__ int pickle_finish(const uint8_t *src, int U, uint8_t *dst, int C, int flags, int lane)
{
    if (C <= 0 || C >= U) {                                 /* pickle.cs:85,:135 raw */
        wave_sync();
        if (lane == 0) dst[0] = 0;
        wave_copy(dst + 1, src, (uint32_t)U, lane);
        return 1 + U;
    }
    const int diff = U - C;
    const int sod = (flags & FLAG_PICKLE_WRITER) ? effective_size_of(U) : effective_size_of(diff);
    const int code = sod == 4 ? 3 : sod;                    /* pickle.cs:228 */
    wave_sync();
    if (sod != 4) wave_shift_down(dst + 1 + sod, dst + 5, (uint32_t)C, lane);
    if (lane == 0) {
        dst[0] = (uint8_t)((code & 3) << 6);                /* pickle.cs:221-222, version 0 */
        for (int i = 0; i < sod; i++) dst[1 + i] = (uint8_t)((uint32_t)diff >> (8 * i));
    }
    return 1 + sod + C;
}

/* one message -> envelope; returns envelope length, 0 for an empty message, -1 if dst too small */
__device__ __forceinline__ int pickle_block(const uint8_t *src, int U, uint8_t *dst, int cap, int level, int flags,
                                            uint32_t *tabw, int lane)
{
    (void)level;
    if (U <= 0) return 0;                                   /* pickle.cs:53-54 */
    if (cap < 1 + 4 + U) return -1;
    int C = 0;
    if (U > 1) C = compress_fast_block(src, U, dst + 5, U - 1, 1, tabw, lane, nullptr, (flags & FLAG_X32) != 0);
    return pickle_finish(src, U, dst, C, flags, lane);
}

struct PickleHeader { int data_offset; int result_len; bool compressed; bool ok; };

/* LZ4Pickler.unpickle.cs:131-158; all lanes compute the same header */
__device__ __forceinline__ PickleHeader unpickle_header(const uint8_t *src, int len)
{
    PickleHeader h{0, 0, false, false};
    if (len <= 0) return h;
    const uint32_t b0 = src[0];
    if ((b0 & 7u) != 0u) return h;                          /* version */
    const int code = (int)((b0 >> 6) & 3u);
    const int sod = code == 3 ? 4 : code;
    const int off = 1 + sod;
    const int data_len = len - off;
    if (data_len < 0) return h;
    uint32_t diff = 0;
    for (int i = 0; i < sod; i++) diff |= (uint32_t)src[1 + i] << (8 * i);
    h.data_offset = off;
    h.result_len = (int)((uint32_t)data_len + diff);        /* C# unchecked int add */
    h.compressed = diff != 0;
    h.ok = true;
    return h;
}

/* Unpickle(source, output): returns the unpickled size (== cap), 0 for an empty pickle,
 * -1 where the reference throws (unpickle.cs:115-128,:134,:143-144) */
template <int ROLE = 0, bool HOP2 = false>
__device__ __forceinline__ int unpickle_block(const uint8_t *src, int len, uint8_t *dst, int cap, int lane, uint32_t *lds,
                                              uint32_t *pipe = nullptr, uint32_t *pace = nullptr)
{
    if (len == 0) return 0;
    const PickleHeader h = unpickle_header(src, len);
    if (!h.ok || h.result_len < 0) return -1;
    if (cap != h.result_len) return -1;
    const int data_len = len - h.data_offset;
    if (!h.compressed) {
        if (ROLE != 1) wave_copy(dst, src + h.data_offset, (uint32_t)data_len, lane);
        return h.result_len;
    }
    int decoded = 0;                                        /* LZ4Codec.Decode: empty -> 0 */
    if (data_len > 0) {
        decoded = decode_block<false, ROLE, HOP2>(src + h.data_offset, data_len, dst, cap, lane, lds, nullptr, false,
                                                  DecodeDict{nullptr, 0u, 0}, pipe, nullptr, pace);
        if (decoded <= 0) decoded = -1;
    }
    return decoded == h.result_len ? decoded : -1;
}

__global__ __launch_bounds__(64) void k4_pickle_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[ENCODE_LDS_DWORDS];
    const int lane = lane_id();
    const long long b = a.order ? (long long)uni(a.order[blockIdx.x]) : (long long)blockIdx.x;
    const int r = pickle_block(a.src + a.srcOff[b], a.srcLen[b], a.dst + a.dstOff[b], a.dstCap[b], a.level,
                               a.flags, tab, lane);
    if (lane == 0) a.outLen[b] = r;
}

__global__ __launch_bounds__(64 * DECODE_WAVES_PER_WG) void k4_unpickle_kernel(BatchArgs a)
{
    __shared__ uint32_t lds[DECODE_WAVES_PER_WG][DECODE_LDS_DWORDS];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const long long slot = (long long)blockIdx.x * DECODE_WAVES_PER_WG + (long long)wave;
    if (slot >= a.n) return;
    const long long b = a.order ? (long long)uni(a.order[slot]) : slot;    /* longest envelope first */
    const int r = unpickle_block(a.src + a.srcOff[b], a.srcLen[b], a.dst + a.dstOff[b], a.dstCap[b], lane, lds[wave]);
    if (lane == 0) a.outLen[b] = r;
}

/* Pickles are ragged (1 KiB .. 4 MiB in BASELINE configs[3]) and, started longest first, the call lasts as long as
 * its biggest message: the two waves per message of k4_decode_pair_kernel shorten exactly that. */
template <bool HOP2>
__device__ __forceinline__ void unpickle_pair_kernel_body(const BatchArgs &a, uint32_t (*lds)[DECODE_PAIR_LDS_DWORDS])
{
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t pair = wave >> 1, role = (wave ^ blockIdx.x) & 1u;   /* as in k4_decode_pair_kernel */
    const long long slot = (long long)blockIdx.x * DECODE_PAIRS_PER_WG + (long long)pair;
    uint32_t *ring = lds[pair], *pipe = lds[pair] + RING_DWORDS;
    if (role == 0) pipe_init(pipe, a.status, lane);
    __syncthreads();
    if (slot >= a.n) return;
    const long long b = a.order ? (long long)uni(a.order[slot]) : slot;
    const int len = a.srcLen[b];
    const uint8_t *src = a.src + a.srcOff[b];
    uint8_t *dst = a.dst + a.dstOff[b];
    const int cap = a.dstCap[b];
    if (role == 0) {
        if (K4_DEC_PACE) Pace::begin(a.pace, pipe + 2, lane);
        unpickle_block<1, HOP2>(src, len, dst, cap, lane, ring, pipe, a.pace);
    } else {
        const int r = unpickle_block<2>(src, len, dst, cap, lane, ring, pipe, a.pace);
        if (lane == 0) a.outLen[b] = r;
    }
}
__global__ __launch_bounds__(128 * DECODE_PAIRS_PER_WG) __attribute__((amdgpu_waves_per_eu(4, 8))) void k4_unpickle_pair_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[DECODE_PAIRS_PER_WG][DECODE_PAIR_LDS_DWORDS];
    unpickle_pair_kernel_body<false>(a, lds);
}
/* the token chain two links per hop (k4lz4_decode.hpp, follow_tokens): launches that leave wave slots free */
__global__ __launch_bounds__(128 * DECODE_PAIRS_PER_WG) __attribute__((amdgpu_waves_per_eu(4, 8))) void k4_unpickle_pair2_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[DECODE_PAIRS_PER_WG][DECODE_PAIR_LDS_DWORDS];
    unpickle_pair_kernel_body<true>(a, lds);
}

/* HC levels: the block encoder runs as its own kernels between these two.
 * prep: encoder slot = envelope slot + 5, capacity U - 1 (or -1 = slot too small / empty message) */
__global__ __launch_bounds__(256) void k4_pickle_prep_kernel(BatchArgs a, uint64_t *encOff, int32_t *encCap)
{
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= a.n) return;
    const int U = a.srcLen[b];
    encOff[b] = a.dstOff[b] + 5u;
    encCap[b] = (U > 0 && a.dstCap[b] >= 1 + 4 + U) ? U - 1 : 0;
}
/* finish: encLen[i] = LLxx-level encoder result for message i */
constexpr int PICKLE_FINISH_WAVES_PER_WG = 4;        /* (one-wave workgroups are placed badly: see k4_hc_parse_kernel) */
__global__ __launch_bounds__(64 * PICKLE_FINISH_WAVES_PER_WG) void k4_pickle_finish_kernel(BatchArgs a, const int32_t *encLen)
{
    const int lane = lane_id();
    const long long b = (long long)blockIdx.x * PICKLE_FINISH_WAVES_PER_WG + (long long)uni(threadIdx.x >> 6);
    if (b >= a.n) return;
    const int U = a.srcLen[b];
    int r;
    if (U <= 0) r = 0;
    else if (a.dstCap[b] < 1 + 4 + U) r = -1;
    else if (U > 1 && encLen[b] == HC_NO_SCRATCH) r = -1;   /* the HC reservation was too small: not pickled (the status word says why), never a raw envelope */
    else r = pickle_finish(a.src + a.srcOff[b], U, a.dst + a.dstOff[b], U > 1 ? encLen[b] : 0, a.flags, lane);
    if (lane == 0) a.outLen[b] = r;
}

/* sizes only: outLen[i] = unpickled size or -1 (one thread per message) */
__global__ __launch_bounds__(256) void k4_unpickle_sizes_kernel(BatchArgs a)
{
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= a.n) return;
    const int len = a.srcLen[b];
    int r = 0;
    if (len != 0) {
        const PickleHeader h = unpickle_header(a.src + a.srcOff[b], len);
        r = (h.ok && h.result_len >= 0) ? h.result_len : -1;
    }
    a.outLen[b] = r;
}

}  // namespace k4
