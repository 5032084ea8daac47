/*
 * k4lz4_decode.hpp -- batched LZ4 block decoder for gfx950, one wavefront per block.
 *
 * Replaces (for batches of independent blocks) the reference's
 *   LZ4Codec.Decode            src/K4os.Compression.LZ4/LZ4Codec.cs:104-115
 *   LLxx.LZ4_decompress_safe   Engine/LLxx.cs:17-26
 *   LL64.LZ4_decompress_safe   Engine/x64/LL64.dec.cs:469-477  (endOnInputSize, full, noDict)
 *   LL64.LZ4_decompress_generic Engine/x64/LL64.dec.cs:123-467
 * Accept/reject decisions, the error position and the produced bytes follow that function
 * (including its two-stage shortcut at :191-225, whose relaxed end-of-block rules are observable
 * on malformed input); how the bytes are moved is entirely different:
 *
 *   - The compressed stream is pulled with coalesced dword loads into a 512-byte window that
 *     lives in two VGPRs per lane (InputWindow).  Token / offset / length bytes are picked out of
 *     the window with v_readlane into SGPRs, so the serial parse chain of a block runs on the
 *     scalar unit and never waits on a memory round trip.
 *   - Literal runs and matches are moved by all 64 lanes at once: byte-per-lane for the short
 *     ones, 16 B per lane (1 KiB per wave instruction) for long runs.  Overlapping matches
 *     (offset < length) read the already-final first period, so no lane depends on a byte written
 *     by the same instruction.
 *   - The match source is the block's own earlier output in HBM/L2; the wave's stores and loads
 *     to it are ordered by program order (wave_sync() pins the compiler).
 */
#pragma once
#include "k4lz4_common.hpp"

namespace k4 {

struct InputWindow {
    const uint32_t *base;  /* dword-aligned address at or below the first stream byte */
    uint32_t a0;           /* misalignment of the stream start: 0..3 */
    uint32_t ndw;          /* dwords that contain stream bytes */
    uint32_t wd;           /* dword index held by lane 0 of w0 (multiple of 64) */
    uint32_t w0, w1;       /* lane l: dwords wd + l and wd + 64 + l (0 beyond the stream) */

    __device__ __forceinline__ uint32_t load(uint32_t dw) const { return dw < ndw ? base[dw] : 0u; }

    __device__ __forceinline__ void init(const uint8_t *in, uint32_t len, int lane)
    {
        a0 = (uint32_t)((uintptr_t)in & 3u);
        base = (const uint32_t *)(in - a0);
        ndw = (a0 + len + 3u) >> 2;
        wd = 0;
        w0 = load((uint32_t)lane);
        w1 = load(64u + (uint32_t)lane);
    }

    /* the 4 stream bytes at wave-uniform position p, little endian; bytes past the end read 0 */
    __device__ __forceinline__ uint32_t fetch(uint32_t p, int lane)
    {
        const uint32_t q = p + a0;
        const uint32_t d = q >> 2;
        uint32_t rel = d - wd;
        if (rel >= 64u) {
            if (rel < 128u) {
                w0 = w1;
                wd += 64u;
            } else {
                wd = d & ~63u;
                w0 = load(wd + (uint32_t)lane);
            }
            w1 = load(wd + 64u + (uint32_t)lane);
            rel = d - wd;
        }
        const uint32_t lo = __builtin_amdgcn_readlane(w0, (int)rel);
        const uint32_t hi = rel == 63u ? __builtin_amdgcn_readlane(w1, 0) : __builtin_amdgcn_readlane(w0, (int)rel + 1);
        const uint64_t v = ((uint64_t)hi << 32) | lo;
        return (uint32_t)(v >> ((q & 3u) * 8u));
    }
};

/* Match copy inside the output block: out[op + i] = out[op - offset + i] with the byte-serial
 * (replicating) semantics of LL64.dec.cs:408-450.  offset >= 1. */
__device__ __forceinline__ void wave_match_copy(uint8_t *out, uint32_t op, uint32_t offset, uint32_t len, int lane)
{
    const uint8_t *m = out + op - offset;
    uint8_t *d = out + op;
    wave_sync();  /* earlier stores of any lane -> these loads */
    if (offset >= len) {
        wave_copy(d, m, len, lane);
    } else if (offset >= 64u) {
        for (uint32_t k0 = 0; k0 < len; k0 += 64u) {
            const uint32_t k = k0 + (uint32_t)lane;
            if (k < len) d[k] = m[k];
            wave_sync();
        }
    } else {
        /* period < 64: every lane reads from the first period, which is final */
        const uint32_t chunk = (64u / offset) * offset;
        const uint32_t r = (uint32_t)lane % offset;
        for (uint32_t k0 = 0; k0 < len; k0 += chunk) {
            const uint32_t k = k0 + (uint32_t)lane;
            if ((uint32_t)lane < chunk && k < len) d[k] = m[r];
        }
    }
}

/*
 * Decode one block.  Returns what LL64.LZ4_decompress_safe returns: the number of bytes written,
 * or -(input position) - 1 when the stream is malformed (LL64.dec.cs:465).
 */
__device__ __forceinline__ int decode_block(const uint8_t *in, int src_size, uint8_t *out, int out_size, int lane)
{
    if (out_size == 0) {                                   /* LL64.dec.cs:162-168 */
        if (src_size == 1) {
            uint32_t b = uni(lane == 0 ? (uint32_t)in[0] : 0u);
            return b == 0 ? 0 : -1;
        }
        return -1;
    }
    if (src_size <= 0) return -1;                          /* :172 (negative sizes cannot come through the API) */

    InputWindow win;
    win.init(in, (uint32_t)src_size, lane);

    const int64_t iend = src_size;
    const int64_t oend = out_size;
    const int64_t shortiend = iend - 14 - 2;               /* :152 */
    const int64_t shortoend = oend - 14 - 18;              /* :153 */
    int64_t ip = 0, op = 0;

    for (;;) {
        uint32_t w = win.fetch((uint32_t)ip, lane);
        const uint32_t token = w & 0xffu;
        ip++;
        uint32_t length = token >> ML_BITS;
        uint32_t offset;
        int64_t match;
        bool have_match = false;

        if (length != RUN_MASK && ip < shortiend && op <= shortoend) {   /* :191-225 */
            if ((uint32_t)lane < length) out[op + lane] = in[ip + lane];
            op += length;
            ip += length;
            /* offset: inside the token's dword when the literal run is short */
            const uint32_t ow = length <= 1 ? (w >> (8u * (1u + length))) : win.fetch((uint32_t)ip, lane);
            offset = ow & 0xffffu;
            ip += 2;
            match = op - (int64_t)offset;
            length = token & ML_MASK;
            if (length != ML_MASK && offset >= 8u && match >= 0) {
                const uint32_t n = length + MINMATCH;      /* <= 18, period >= 8 */
                wave_sync();
                uint32_t r = (uint32_t)lane;
                if (r >= offset) r -= offset;
                if (r >= offset) r -= offset;
                if ((uint32_t)lane < n) out[op + lane] = out[match + r];
                op += n;
                continue;
            }
            have_match = true;                             /* :222 goto _copy_match */
        }

        if (!have_match) {
            if (length == RUN_MASK) {                      /* :228-243, LL.tools.cs:165-193 */
                const int64_t lencheck = iend - RUN_MASK;
                if (ip >= lencheck) return (int)(-ip) - 1; /* initial_error */
                uint32_t s;
                do {
                    s = win.fetch((uint32_t)ip, lane) & 0xffu;
                    ip++;
                    length += s;
                    if (ip >= lencheck) break;             /* loop_error: not fatal here */
                } while (s == 255u);
            }
            const int64_t cpy = op + (int64_t)length;      /* :246-315 */
            if (cpy > oend - MFLIMIT || ip + (int64_t)length > iend - (2 + 1 + LASTLITERALS)) {
                if (ip + (int64_t)length != iend || cpy > oend) return (int)(-ip) - 1;
                wave_copy(out + op, in + ip, length, lane);
                ip += length;
                op += length;
                break;                                     /* last sequence */
            }
            wave_copy(out + op, in + ip, length, lane);
            ip += length;
            op = cpy;
            offset = win.fetch((uint32_t)ip, lane) & 0xffffu;  /* :318-323 */
            ip += 2;
            match = op - (int64_t)offset;
            length = token & ML_MASK;
        }

        /* _copy_match */
        if (length == ML_MASK) {                           /* :326-334: any error is fatal */
            const int64_t lencheck = iend - LASTLITERALS + 1;
            uint32_t s;
            do {
                s = win.fetch((uint32_t)ip, lane) & 0xffu;
                ip++;
                length += s;
                if (ip >= lencheck) return (int)(-ip) - 1;
            } while (s == 255u);
        }
        length += MINMATCH;
        if (match < 0) return (int)(-ip) - 1;              /* :338 offset before block start */
        const int64_t cpy = op + (int64_t)length;
        if (cpy > oend - MATCH_SAFEGUARD) {                /* :427-443 */
            if (cpy > oend - LASTLITERALS) return (int)(-ip) - 1;
        }
        if (offset != 0u) wave_match_copy(out, (uint32_t)op, offset, length, lane);
        op = cpy;
    }
    return (int)op;
}

/* LZ4Codec.Decode mapping (LZ4Codec.cs:104-115): empty input -> 0, engine result <= 0 -> -1 */
__device__ __forceinline__ int codec_decode_result(int src_len, int ret, int flags)
{
    if (flags & FLAG_RAW_RETURN) return ret;
    if (src_len <= 0) return 0;
    return ret <= 0 ? -1 : ret;
}

constexpr int DECODE_WAVES_PER_WG = 4;

__global__ __launch_bounds__(64 * DECODE_WAVES_PER_WG) void k4_decode_kernel(BatchArgs a)
{
    const int lane = lane_id();
    const long long b = (long long)blockIdx.x * DECODE_WAVES_PER_WG + (long long)(threadIdx.x >> 6);
    if (b >= a.n) return;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    const uint8_t *in = a.src + a.srcOff[b];
    uint8_t *out = a.dst + a.dstOff[b];
    int ret = 0;
    if (src_len > 0 || (a.flags & FLAG_RAW_RETURN)) ret = decode_block(in, src_len, out, cap < 0 ? 0 : cap, lane);
    if (lane == 0) a.outLen[b] = codec_decode_result(src_len, ret, a.flags);
}

}  // namespace k4
