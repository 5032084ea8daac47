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
 *   - Copies are batched per 64 sequences: every lane moves the literal run / the match of its
 *     own sequence (all loads before the first store: one memory round trip per batch phase, not
 *     per sequence).  Long runs move 16 B per lane (1 KiB per wave instruction) with the whole
 *     wave; overlapping matches (offset < length) read the already-final first period, so no lane
 *     depends on a byte written by the same instruction.
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

/* up to 7 bytes at p (fewer than 8 readable): little-endian assemble without reading past them */
__device__ __forceinline__ uint64_t load_tail(const uint8_t *p, uint32_t avail)
{
    uint64_t v = 0;
    for (uint32_t i = 0; i < 8u && i < avail; i++) v |= (uint64_t)p[i] << (8u * i);
    return v;
}

/* Per-lane copy of len <= 32 bytes, regions must not overlap.  All (up to four) 8-byte loads are
 * issued before the first store, so a lane pays one memory round trip; the stores write exactly
 * len bytes.  `readable` = bytes that may be read starting at s (>= len). */
__device__ __forceinline__ void lane_copy32(uint8_t *d, const uint8_t *s, uint32_t len, uint32_t readable)
{
    uint64_t v[4];
#pragma unroll
    for (uint32_t c = 0; c < 4u; c++) {
        v[c] = 0;
        if (8u * c < len) v[c] = (8u * c + 8u <= readable) ? ld64u(s + 8u * c) : load_tail(s + 8u * c, readable - 8u * c);
    }
#pragma unroll
    for (uint32_t c = 0; c < 4u; c++) {
        if (8u * c >= len) break;
        uint32_t rem = len - 8u * c;
        uint8_t *q = d + 8u * c;
        uint64_t x = v[c];
        if (rem >= 8u) {
            ((U64u *)q)->v = x;
        } else {
            if (rem & 4u) { ((U32u *)q)->v = (uint32_t)x; x >>= 32; q += 4; }
            if (rem & 2u) { ((U16u *)q)->v = (uint16_t)x; x >>= 16; q += 2; }
            if (rem & 1u) { *q = (uint8_t)x; }
        }
    }
}

constexpr uint32_t LANE_COPY_MAX = 32;

/*
 * Decode one block.  Returns what LL64.LZ4_decompress_safe returns: the number of bytes written,
 * or -(input position) - 1 when the stream is malformed (LL64.dec.cs:465).
 *
 * Structure: repeat { PARSE up to 64 sequences on the scalar unit (no memory waits: the stream
 * comes out of the register window) and drop each sequence's (literal position, literal length,
 * output position, offset, match length) into lane k of five VGPRs;  LITERALS: every lane moves
 * its own literal run (one memory round trip for 64 sequences);  MATCHES: lanes whose source lies
 * entirely below the first unfinished match copy in parallel, round by round }.  Long runs and
 * overlapping matches are moved by the whole wave.  All accept/reject decisions depend only on
 * positions and lengths, so they are taken in PARSE exactly in the reference's order.
 */
template <bool PROF = false>
__device__ __forceinline__ int decode_block(const uint8_t *in, int src_size, uint8_t *out, int out_size, int lane,
                                        unsigned long long *pc = nullptr)
{
    unsigned long long c_parse = 0, c_lit = 0, c_match = 0, n_batch = 0, n_round = 0, n_seq = 0, n_coop = 0;
    prof_place<PROF>(pc, 8, lane);
    const unsigned long long t_begin = prof_now<PROF>();
    if (out_size == 0) {                                   /* LL64.dec.cs:162-168 */
        if (src_size == 1) {
            uint32_t b = uni(lane == 0 ? (uint32_t)in[0] : 0u);
            return b == 0 ? 0 : -1;
        }
        return -1;
    }
    if (src_size <= 0) return -1;                          /* :172 */

    InputWindow win;
    win.init(in, (uint32_t)src_size, lane);

    const int64_t iend = src_size;
    const int64_t oend = out_size;
    const int64_t shortiend = iend - 14 - 2;               /* :152 */
    const int64_t shortoend = oend - 14 - 18;              /* :153 */
    int64_t ip = 0, op = 0;

    for (;;) {
        /* ---------------- PARSE ---------------- */
        const unsigned long long t0 = prof_now<PROF>();
        uint32_t v_lpos = 0, v_llen = 0, v_out = 0, v_moff = 0, v_mlen = 0;
        int nseq = 0;
        int err = 0;
        bool done = false;
        while (nseq < 64) {
            uint32_t w = win.fetch((uint32_t)ip, lane);
            const uint32_t token = w & 0xffu;
            ip++;
            uint32_t length = token >> ML_BITS;
            uint32_t offset = 0;
            int64_t match = 0;
            uint32_t s_lpos, s_llen, s_out, s_moff = 0, s_mlen = 0, adv = 0;
            bool last = false, need_match = true;

            if (length != RUN_MASK && ip < shortiend && op <= shortoend) {   /* :191-225 */
                s_lpos = (uint32_t)ip; s_llen = length; s_out = (uint32_t)op;
                op += length;
                ip += length;
                const uint32_t ow = length <= 1 ? (w >> (8u * (1u + length))) : win.fetch((uint32_t)ip, lane);
                offset = ow & 0xffffu;
                ip += 2;
                match = op - (int64_t)offset;
                length = token & ML_MASK;
                if (length != ML_MASK && offset >= 8u && match >= 0) {
                    s_moff = offset; s_mlen = length + MINMATCH; adv = s_mlen;
                    need_match = false;
                }
            } else {
                if (length == RUN_MASK) {                      /* :228-243, LL.tools.cs:165-193 */
                    const int64_t lencheck = iend - RUN_MASK;
                    if (ip >= lencheck) { err = (int)(-ip) - 1; break; }   /* initial_error */
                    uint32_t s;
                    do {
                        s = win.fetch((uint32_t)ip, lane) & 0xffu;
                        ip++;
                        length += s;
                        if (ip >= lencheck) break;             /* loop_error: not fatal here */
                    } while (s == 255u);
                }
                const int64_t cpy = op + (int64_t)length;      /* :246-315 */
                s_lpos = (uint32_t)ip; s_llen = length; s_out = (uint32_t)op;
                if (cpy > oend - MFLIMIT || ip + (int64_t)length > iend - (2 + 1 + LASTLITERALS)) {
                    if (ip + (int64_t)length != iend || cpy > oend) { err = (int)(-ip) - 1; break; }
                    ip += length;
                    op += length;
                    last = true;
                    need_match = false;
                } else {
                    ip += length;
                    op = cpy;
                    offset = win.fetch((uint32_t)ip, lane) & 0xffffu;  /* :318-323 */
                    ip += 2;
                    match = op - (int64_t)offset;
                    length = token & ML_MASK;
                }
            }
            if (need_match) {                                  /* _copy_match */
                if (length == ML_MASK) {                       /* :326-334: any error is fatal */
                    const int64_t lencheck = iend - LASTLITERALS + 1;
                    uint32_t s;
                    do {
                        s = win.fetch((uint32_t)ip, lane) & 0xffu;
                        ip++;
                        length += s;
                        if (ip >= lencheck) { err = (int)(-ip) - 1; break; }
                    } while (s == 255u);
                    if (err) break;
                }
                length += MINMATCH;
                if (match < 0) { err = (int)(-ip) - 1; break; }              /* :338 */
                const int64_t cpy = op + (int64_t)length;
                if (cpy > oend - MATCH_SAFEGUARD && cpy > oend - LASTLITERALS) { err = (int)(-ip) - 1; break; }  /* :427-433 */
                s_moff = offset;
                s_mlen = offset != 0u ? length : 0u;           /* offset 0 (hostile): output left as is */
                adv = length;
            }
            if (lane == nseq) {   /* drop the sequence into lane `nseq` */
                v_lpos = s_lpos; v_llen = s_llen; v_out = s_out; v_moff = s_moff; v_mlen = s_mlen;
            }
            nseq++;
            op += adv;
            if (last) { done = true; break; }
        }
        if (err) return err;
        const unsigned long long t1 = prof_now<PROF>();

        /* ---------------- LITERALS ---------------- */
        {
            const bool mine = lane < nseq;
            if (mine && v_llen != 0u && v_llen <= LANE_COPY_MAX)
                lane_copy32(out + v_out, in + v_lpos, v_llen, (uint32_t)src_size - v_lpos);
            unsigned long long big = __ballot(mine && v_llen > LANE_COPY_MAX);
            while (big) {
                const int f = ctz64(big);
                big &= big - 1;
                wave_copy(out + __builtin_amdgcn_readlane(v_out, f), in + __builtin_amdgcn_readlane(v_lpos, f),
                          __builtin_amdgcn_readlane(v_llen, f), lane);
            }
        }

        const unsigned long long t2 = prof_now<PROF>();
        /* ---------------- MATCHES ---------------- */
        {
            const uint32_t mdst = v_out + v_llen;
            const uint32_t msrc = mdst - v_moff;
            unsigned long long pend = __ballot(lane < nseq && v_mlen != 0u);
            while (pend) {
                const int f = ctz64(pend);
                const uint32_t F = __builtin_amdgcn_readlane(mdst, f);
                const uint32_t f_len = __builtin_amdgcn_readlane(v_mlen, f);
                const uint32_t f_off = __builtin_amdgcn_readlane(v_moff, f);
                const bool f_coop = f_len > LANE_COPY_MAX || f_off < f_len;
                if (PROF) { n_round++; n_coop += f_coop ? 1 : 0; }
                if (f_coop) wave_match_copy(out, F, f_off, f_len, lane);   /* includes the wave_sync */
                else wave_sync();
                const bool pending = ((pend >> lane) & 1ull) != 0;
                const bool ready = pending && lane != f && msrc + v_mlen <= F;
                const bool go = ready || (lane == f && !f_coop);
                if (go && v_mlen <= LANE_COPY_MAX)
                    lane_copy32(out + mdst, out + msrc, v_mlen, (uint32_t)out_size - msrc);
                unsigned long long big = __ballot(ready && v_mlen > LANE_COPY_MAX);
                const unsigned long long gone = __ballot(go) | (1ull << f);
                while (big) {
                    const int g = ctz64(big);
                    big &= big - 1;
                    const uint32_t g_dst = __builtin_amdgcn_readlane(mdst, g);
                    const uint32_t g_len = __builtin_amdgcn_readlane(v_mlen, g);
                    wave_copy(out + g_dst, out + g_dst - __builtin_amdgcn_readlane(v_moff, g), g_len, lane);
                }
                pend &= ~gone;
            }
            wave_sync();
        }
        if (PROF) {
            const unsigned long long t3 = prof_now<PROF>();
            c_parse += t1 - t0; c_lit += t2 - t1; c_match += t3 - t2; n_batch++; n_seq += (unsigned long long)nseq;
        }
        if (done) break;
    }
    if (PROF && pc && lane == 0) {
        pc[0] = prof_now<PROF>() - t_begin; pc[1] = c_parse; pc[2] = c_lit; pc[3] = c_match;
        pc[4] = n_batch; pc[5] = n_round; pc[6] = n_seq; pc[7] = n_coop;
    }
    prof_place<PROF>(pc, 9, lane);
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
    const long long b = (long long)blockIdx.x * DECODE_WAVES_PER_WG + (long long)uni(threadIdx.x >> 6);
    if (b >= a.n) return;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    const uint8_t *in = a.src + a.srcOff[b];
    uint8_t *out = a.dst + a.dstOff[b];
    int ret = 0;
    if (src_len > 0 || (a.flags & FLAG_RAW_RETURN)) ret = decode_block(in, src_len, out, cap < 0 ? 0 : cap, lane);
    if (lane == 0) a.outLen[b] = codec_decode_result(src_len, ret, a.flags);
}

/* diagnostic twin: same decode with per-phase cycle counters (a.prof, 8 per block) */
__global__ __launch_bounds__(64 * DECODE_WAVES_PER_WG) void k4_decode_prof_kernel(BatchArgs a)
{
    const int lane = lane_id();
    const long long b = (long long)blockIdx.x * DECODE_WAVES_PER_WG + (long long)uni(threadIdx.x >> 6);
    if (b >= a.n) return;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    int ret = 0;
    if (src_len > 0) ret = decode_block<true>(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, lane, a.prof + PROF_STRIDE * b);
    if (lane == 0) a.outLen[b] = codec_decode_result(src_len, ret, a.flags);
}

}  // namespace k4
