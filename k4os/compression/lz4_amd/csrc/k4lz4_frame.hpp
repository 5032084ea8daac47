/*
 * k4lz4_frame.hpp -- device side of the block-stream / frame layer.
 *
 *   k4_allow_copy_kernel    LZ4EncoderBase.Encode(allowCopy: true), Encoders/LZ4EncoderBase.cs:66-88:
 *                           a block that did not shrink is stored raw and reported as -length
 *   k4_decode_chain_kernel  a whole block stream per wavefront, blocks in order:
 *                           LZ4BlockDecoder.Decode / Inject (Encoders/LZ4BlockDecoder.cs:39-71) for
 *                           independent blocks, LZ4ChainDecoder (-> LL64.LZ4_decompress_safe_continue,
 *                           LL64.dec.cs:479-521) for chained ones: every block is decoded with the
 *                           output so far (at most 64 KiB of it) as its prefix dictionary.
 *                           Chained streams are serial by construction; streams are parallel.
 */
#pragma once
#include "k4lz4_decode.hpp"

namespace k4 {

/* after an encode launch with FLAG_ALLOW_COPY: outLen[b] >= srcLen[b] -> raw copy, outLen[b] = -srcLen[b];
 * outLen[b] <= 0 (target too small: the reference throws) -> 0 */
__global__ __launch_bounds__(256) void k4_allow_copy_kernel(BatchArgs a)
{
    const int lane = lane_id();
    const long long b = (long long)blockIdx.x * 4 + (long long)uni(threadIdx.x >> 6);
    if (b >= a.n) return;
    const int src_len = a.srcLen[b];
    int got = a.outLen[b];
    if (src_len <= 0 || got <= 0) {
        got = 0;
    } else if (got >= src_len) {
        if (a.dstCap[b] >= src_len) {
            wave_copy(a.dst + a.dstOff[b], a.src + a.srcOff[b], (uint32_t)src_len, lane);
            got = -src_len;
        } else {
            got = 0;
        }
    }
    wave_sync();                                 /* every lane has read outLen[b] before it changes */
    if (lane == 0) a.outLen[b] = got;
}

struct ChainArgs {
    const uint8_t *src;          /* stored blocks, packed */
    const uint64_t *blkOff;      /* per block: offset of its payload in src */
    const uint32_t *blkLen;      /* per block: payload length, bit 31 = stored raw (LZ4FrameWriter.cs:159-160) */
    const uint64_t *firstBlk;    /* per stream: index of its first block */
    const uint32_t *nBlk;        /* per stream: number of blocks */
    const int32_t *blockSize;    /* per stream: maximum block size */
    const uint8_t *chained;      /* per stream: blocks depend on the previous output */
    uint8_t *dst;
    const uint64_t *dstOff;      /* per stream */
    const uint64_t *dstCap;
    long long *outLen;           /* per stream: bytes produced, -6 a block does not decode, -9 target too small */
    long long n;
};

__global__ __launch_bounds__(64 * DECODE_WAVES_PER_WG) void k4_decode_chain_kernel(ChainArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[DECODE_WAVES_PER_WG][DECODE_LDS_DWORDS];
    const int lane = lane_id();
    const int wave = (int)uni(threadIdx.x >> 6);
    const long long s = (long long)blockIdx.x * DECODE_WAVES_PER_WG + wave;
    if (s >= a.n) return;
    uint8_t *out = a.dst + a.dstOff[s];
    const uint64_t cap = a.dstCap[s];
    const uint64_t first = a.firstBlk[s];
    const uint32_t count = a.nBlk[s];
    const int block_size = a.blockSize[s];
    const bool chained = a.chained[s] != 0;
    uint64_t op = 0;
    long long result = 0;
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t lc = a.blkLen[first + k];
        const uint32_t n = lc & 0x7fffffffu;
        const uint8_t *in = a.src + a.blkOff[first + k];
        const uint64_t room = cap - op;
        if (lc & 0x80000000u) {                                  /* LZ4BlockDecoder.Inject */
            if (n > room) { result = -9; break; }
            wave_sync();
            wave_copy(out + op, in, n, lane);
            op += n;
        } else {
            const int want = (int)(room < (uint64_t)block_size ? room : (uint64_t)block_size);
            DecodeDict dict{nullptr, 0u, 0};
            if (chained && op > 0) {
                dict.end = out + op;
                dict.size = op >= 65535u ? 65536u : (uint32_t)op;
                dict.mode = 1;
            }
            wave_sync();                                         /* the previous block's stores -> this block's loads */
            const int d = decode_block(in, (int)n, out + op, want, lane, lds[wave], nullptr, false, dict);
            if (d < 0) { result = room < (uint64_t)block_size ? -9 : -6; break; }
            op += (uint64_t)d;
        }
    }
    if (lane == 0) a.outLen[s] = result < 0 ? result : (long long)op;
}

}  // namespace k4
