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

/* host-pointer calls: the bytes each block produced, moved next to each other (dst + dstOff[b], 16-byte aligned positions)
 * so that only they travel back over PCIe, not the slots' worst-case capacities.  outLen < 0 with `raw_negative` = a block
 * stored raw by FLAG_ALLOW_COPY (-length bytes). */
__global__ __launch_bounds__(256) void k4_compact_kernel(const uint8_t *src, const uint64_t *srcOff, const int32_t *outLen, uint8_t *dst,
                                                         const uint64_t *dstOff, long long n, int raw_negative)
{
    const int lane = lane_id();
    const long long b = (long long)blockIdx.x * 4 + (long long)uni(threadIdx.x >> 6);
    if (b >= n) return;
    int len = outLen[b];
    if (len < 0 && raw_negative) len = -len;
    if (len > 0) wave_copy(dst + dstOff[b], src + srcOff[b], (uint32_t)len, lane);
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
    uint32_t *status;            /* the context's status word (k4lz4_common.hpp), or nullptr */
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


/* The same with two waves per stream (see k4_decode_pair_kernel): the parsing wave walks the stream's blocks on its own --
 * what it needs of a block's outcome, the decoded size, it computes itself -- and may be several blocks ahead of the
 * wave that copies; both make the same decisions about stored blocks, full targets and failures from the same numbers. */
__global__ __launch_bounds__(128 * DECODE_PAIRS_PER_WG) __attribute__((amdgpu_waves_per_eu(4, 8))) void k4_decode_chain_pair_kernel(ChainArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[DECODE_PAIRS_PER_WG][DECODE_PAIR_LDS_DWORDS];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t pair = wave >> 1, role = (wave ^ blockIdx.x) & 1u;   /* as in k4_decode_pair_kernel */
    const long long s = (long long)blockIdx.x * DECODE_PAIRS_PER_WG + (long long)pair;
    uint32_t *ring = lds[pair], *pipe = lds[pair] + RING_DWORDS;
    if (role == 0) pipe_init(pipe, a.status, lane);
    __syncthreads();
    if (s >= a.n) return;
    uint8_t *out = a.dst + a.dstOff[s];
    const uint64_t cap = a.dstCap[s];
    const uint64_t first = a.firstBlk[s];
    const uint32_t count = a.nBlk[s];
    const int block_size = a.blockSize[s];
    const bool chained = a.chained[s] != 0;
    uint64_t op = 0;
    long long result = 0;
    uint32_t seq = 0;
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t lc = a.blkLen[first + k];
        const uint32_t n = lc & 0x7fffffffu;
        const uint8_t *in = a.src + a.blkOff[first + k];
        const uint64_t room = cap - op;
        if (lc & 0x80000000u) {                                  /* LZ4BlockDecoder.Inject */
            if (n > room) { result = -9; break; }
            if (role == 1) {
                wave_sync();
                wave_copy(out + op, in, n, lane);
            }
            op += n;
        } else {
            const int want = (int)(room < (uint64_t)block_size ? room : (uint64_t)block_size);
            DecodeDict dict{nullptr, 0u, 0};
            if (chained && op > 0) {
                dict.end = out + op;
                dict.size = op >= 65535u ? 65536u : (uint32_t)op;
                dict.mode = 1;
            }
            int d;
            if (role == 0) {
                d = decode_block<false, 1>(in, (int)n, out + op, want, lane, ring, nullptr, false, dict, pipe, &seq);
            } else {
                wave_sync();                                     /* the previous block's stores -> this block's loads */
                d = decode_block<false, 2>(in, (int)n, out + op, want, lane, ring, nullptr, false, dict, pipe, &seq);
            }
            if (d < 0) { result = room < (uint64_t)block_size ? -9 : -6; break; }
            op += (uint64_t)d;
        }
    }
    if (role == 1 && lane == 0) a.outLen[s] = result < 0 ? result : (long long)op;
}


/* ---- frame writer on the device (Frames/LZ4FrameWriter.async.cs:15-27 per block, :75-90 tail; LZ4FrameWriter.cs:57-108
 * header).  The host lays out WHERE things go (it knows the block split); the bytes are moved here. ----------------- */
struct FrameBlocksArgs {
    const uint8_t *arena;        /* encoder output slots */
    const uint64_t *slotOff;     /* per block: its slot in the arena */
    const int32_t *outLen;       /* per block: k4lz4_encode_batch result with FLAG_ALLOW_COPY (< 0: stored raw) */
    const uint32_t *blkSum;      /* per block: XXH32 of the stored payload, or nullptr */
    const uint64_t *recOff;      /* per block: where its record (length word, payload, checksum) starts in `frames` */
    uint8_t *frames;
    long long n;
};

/* one wavefront per block: BlockLengthCode, payload, optional block checksum */
__global__ __launch_bounds__(256) void k4_frame_blocks_kernel(FrameBlocksArgs a)
{
    const int lane = lane_id();
    const long long b = (long long)blockIdx.x * 4 + (long long)uni(threadIdx.x >> 6);
    if (b >= a.n) return;
    const int32_t got = a.outLen[b];
    const uint32_t stored = (uint32_t)(got < 0 ? -got : got);
    uint8_t *rec = a.frames + a.recOff[b];
    if (lane == 0) ((U32u *)rec)->v = stored | (got < 0 ? 0x80000000u : 0u);       /* LZ4FrameWriter.cs:159-160 */
    wave_copy(rec + 4, a.arena + a.slotOff[b], stored, lane);
    if (a.blkSum && lane == 0) ((U32u *)(rec + 4 + stored))->v = a.blkSum[b];
}

struct FrameEdgesArgs {
    const uint8_t *hdr;          /* per frame: 16 bytes, the header bytes FLG.. (hdrLen of them used) */
    const uint32_t *hdrLen;
    const uint32_t *hdrSum;      /* XXH32 of those bytes */
    const uint64_t *frameOff;    /* per frame: where it starts in `frames` */
    const uint64_t *tailOff;     /* per frame: where its EndMark goes */
    const uint32_t *contentSum;  /* per frame, or nullptr */
    uint8_t *frames;
    uint64_t *frameLen;          /* per frame: total length, written here */
    long long n;
};

/* one thread per frame: magic, header, header checksum byte; EndMark and content checksum */
__global__ __launch_bounds__(256) void k4_frame_edges_kernel(FrameEdgesArgs a)
{
    const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
    if (f >= a.n) return;
    uint8_t *p = a.frames + a.frameOff[f];
    ((U32u *)p)->v = 0x184D2204u;
    const uint32_t hl = a.hdrLen[f];
    for (uint32_t i = 0; i < hl; i++) p[4 + i] = a.hdr[16 * f + i];
    p[4 + hl] = (uint8_t)(a.hdrSum[f] >> 8);
    uint8_t *t = a.frames + a.tailOff[f];
    ((U32u *)t)->v = 0u;
    uint64_t end = a.tailOff[f] + 4u;
    if (a.contentSum) { ((U32u *)(t + 4))->v = a.contentSum[f]; end += 4u; }
    a.frameLen[f] = end - a.frameOff[f];
}

}  // namespace k4
