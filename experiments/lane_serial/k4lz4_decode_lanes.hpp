/*
 * k4lz4_decode_lanes.hpp -- LZ4 block decoder for gfx950 in which every lane decodes sequences of its own.
 *
 * Replaces, for batches of independent blocks without dictionary (the LZ4Codec.Decode / LL64.LZ4_decompress_safe case,
 * Engine/x64/LL64.dec.cs:469-477 over :123-467), the batch-of-64-sequences organisation of k4lz4_decode.hpp:
 *
 *   WINDOW  parse_window (k4lz4_decode_parse.hpp) finds the token positions of the next 52 x 64 bytes of the compressed
 *           stream; the result stays where it was computed: lane j holds the bit set of the tokens in segment j.
 *   SIZES   every lane walks its own tokens and adds up what they produce (literals + match); a prefix sum over the
 *           lanes gives lane j the output position of its first sequence.  Tokens that need more than the common
 *           case (long lengths, the end zone, anything malformed) end the lane-parallel part of the window there.
 *   COPY    every lane decodes its sequences one after the other, like the reference's loop does, straight from the
 *           stream to the output in memory: literal run, then match.  A match may read what another lane (an
 *           earlier segment of the same window) writes, so lanes publish how far they are (`prog`), and a lane whose
 *           match source is not there yet tries again in the next step.  Lane 0 never waits and a lane only waits for
 *           lanes below it, so the steps always make progress.
 *   CAREFUL the sequences left over -- the first one that is not of the common kind, and the tail of the block
 *           where the end-of-block rules matter -- go one at a time through a scalar parser that follows
 *           LL64.dec.cs:175-451 check by check (same accept/reject decisions, same error positions).
 *
 * No staging of the output in LDS, no descriptor queue, no dependency search over 64 sequences: a step of COPY costs
 * about a hundred wave instructions and retires up to 64 sequences.
 */
#pragma once
#include "k4lz4_common.hpp"
#include "k4lz4_decode.hpp"
#include "k4lz4_decode_parse.hpp"

namespace k4 {

/* LDS per wave, dwords: the parser's ring, marks, bit set, entry offsets | first-token positions | output bases | progress */
constexpr int LANES_OFF_FIRST = PARSE_OFF_TOK, LANES_OFF_BASE = LANES_OFF_FIRST + 72, LANES_OFF_PROG = LANES_OFF_BASE + 72,
              LANES_LDS_DWORDS = LANES_OFF_PROG + 64;
constexpr uint32_t LANES_NONE = 0xffffffffu;

/* the common kind of sequence, as the batch decoder's `fast` (k4lz4_decode.hpp): literal length in the token or one
 * extension byte, match length likewise, everything well inside the input.  P = token position in aligned
 * coordinates, IEND = input length + a0.  next = where the following token starts. */
struct LaneSeq { uint32_t lpos, L, offset, mlen, next; bool simple; };
__device__ __forceinline__ LaneSeq lane_seq(const ParseWin &win, uint32_t P, uint32_t IEND)
{
    LaneSeq q;
    const uint32_t t4 = win.read4(P);
    uint32_t L = (t4 >> 4) & 15u;
    const uint32_t M = t4 & 15u;
    const bool cls_g = L == RUN_MASK;
    uint32_t hdr = 1u;
    bool ok = cls_g ? true : P + 1u + 16u < IEND;          /* :191 ip < shortiend */
    if (cls_g) {
        const uint32_t ext = (t4 >> 8) & 255u;
        ok = ext != 255u;
        L += ext;
        hdr = 2u;
        ok = ok && P + hdr + L + (2u + 1u + LASTLITERALS) <= IEND;   /* :247 */
    }
    const uint32_t o4 = win.read4(P + hdr + L);
    q.offset = o4 & 0xffffu;
    q.next = P + hdr + L + 2u;
    q.mlen = M + MINMATCH;
    ok = ok && q.offset != 0u;
    if (M == ML_MASK) {
        const uint32_t ext = (o4 >> 16) & 255u;
        q.mlen += ext;
        q.next += 1u;
        ok = ok && ext != 255u && q.next + LASTLITERALS < IEND + 1u;   /* :328 */
    }
    q.lpos = P + hdr;
    q.L = L;
    q.simple = ok;
    return q;
}

/* n bytes src -> dst by one lane, the regions do not overlap; only [dst, dst + n) is written, at most `readable` bytes
 * are read from src on */
__device__ __forceinline__ void lane_copy_any(uint8_t *d, const uint8_t *s, uint32_t n, uint32_t readable)
{
    if (n <= LANE_COPY_MAX) {
        lane_copy32(d, s, n, readable);
        return;
    }
    uint32_t k = 0;
    for (; k + 8u <= n; k += 8u) ((U64u *)(d + k))->v = ld64u(s + k);
    if (k < n) ((U64u *)(d + n - 8u))->v = ld64u(s + n - 8u);   /* the tail once more as the run's last 8 bytes */
}

/* out[op + i] = out[op - offset + i], i < n, with the byte-serial semantics of LL64.dec.cs:408-450, by one lane */
__device__ __forceinline__ void lane_match_copy(uint8_t *out, uint32_t op, uint32_t offset, uint32_t n, uint32_t out_size)
{
    uint8_t *d = out + op;
    const uint8_t *m = d - offset;
    if (offset >= n) {
        lane_copy_any(d, m, n, out_size - (op - offset));
        return;
    }
    /* overlapping: the first `step` bytes one at a time (step = the period's first multiple of 8 or more), after that
     * 8 at a time from `step` bytes back, which are final by then */
    uint32_t step = offset;
    while (step < 8u) step += offset;
    uint32_t k = 0;
    if (offset < 8u) {
        const uint32_t head = step < n ? step : n;
        for (; k < head; k++) d[k] = m[k];
    }
    for (; k + 8u <= n; k += 8u) ((U64u *)(d + k))->v = ld64u(d + k - step);
    for (; k < n; k++) d[k] = d[k - step];
}

/* One sequence the careful way: LL64.LZ4_decompress_generic's loop body (Engine/x64/LL64.dec.cs:175-451, endOnInputSize,
 * full, noDict), wave-uniform, with the copies done by the whole wave.  Returns 0 to go on, 1 when the block is
 * finished (result in op), a negative LLxx error code otherwise. */
__device__ __forceinline__ int careful_sequence(ParseWin &win, const uint8_t *in, uint8_t *out, int64_t &ip, int64_t &op,
                                                int64_t iend, int64_t oend, int lane)
{
    const int64_t shortiend = iend - 14 - 2, shortoend = oend - 14 - 18;
    uint32_t w = win.fetch((uint32_t)ip, lane);
    const uint32_t token = w & 0xffu;
    ip++;
    uint32_t length = token >> ML_BITS;
    uint32_t offset = 0;
    int64_t match = 0;
    bool copy_match_checked = false;

    if (length != RUN_MASK && ip < shortiend && op <= shortoend) {   /* :191-225 the shortcut */
        wave_copy(out + op, in + ip, length, lane);
        op += length;
        ip += length;
        const uint32_t ow = length <= 1 ? (w >> (8u * (1u + length))) : win.fetch((uint32_t)ip, lane);
        offset = ow & 0xffffu;
        ip += 2;
        match = op - (int64_t)offset;
        length = token & ML_MASK;
        if (length != ML_MASK && offset >= 8u && match >= 0) {       /* :213 */
            wave_match_copy(out, (uint32_t)op, offset, length + MINMATCH, lane);
            op += length + MINMATCH;
            return 0;
        }
        copy_match_checked = true;                                   /* -> _copy_match */
    }
    if (!copy_match_checked) {
        if (length == RUN_MASK) {                                    /* :228-243, LL.tools.cs:165-193 */
            const int64_t lencheck = iend - RUN_MASK;
            if (ip >= lencheck) return (int)(-ip) - 1;               /* initial_error */
            uint32_t s;
            do {
                s = win.fetch((uint32_t)ip, lane) & 0xffu;
                ip++;
                length += s;
                if (ip >= lencheck) break;                           /* loop_error: not fatal here */
            } while (s == 255u);
        }
        const int64_t cpy = op + (int64_t)length;                    /* :246-315 */
        if (cpy > oend - MFLIMIT || ip + (int64_t)length > iend - (2 + 1 + LASTLITERALS)) {
            if (ip + (int64_t)length != iend || cpy > oend) return (int)(-ip) - 1;
            wave_copy(out + op, in + ip, length, lane);
            ip += length;
            op += length;
            return 1;                                                /* the last sequence */
        }
        wave_copy(out + op, in + ip, length, lane);
        ip += length;
        op = cpy;
        offset = win.fetch((uint32_t)ip, lane) & 0xffffu;            /* :318-323 */
        ip += 2;
        match = op - (int64_t)offset;
        length = token & ML_MASK;
    }
    /* _copy_match */
    if (length == ML_MASK) {                                         /* :326-334: any error is fatal */
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
    if (match < 0) return (int)(-ip) - 1;                            /* :338 offset before the start of the output */
    const int64_t cpy = op + (int64_t)length;
    if (cpy > oend - MATCH_SAFEGUARD && cpy > oend - LASTLITERALS) return (int)(-ip) - 1;   /* :427-433 */
    if (offset != 0u) wave_match_copy(out, (uint32_t)op, offset, length, lane);             /* offset 0 (hostile): output left as is */
    op = cpy;
    return 0;
}

/*
 * Decode one block.  Returns what LL64.LZ4_decompress_safe returns: the number of bytes written, or
 * -(input position) - 1 when the stream is malformed.  `lds`: LANES_LDS_DWORDS dwords owned by this wave.
 * PROF: pc[0] total cycles, [1] windows, [2] sizes, [3] copy, [4] careful; [5] windows, [6] copy steps, [7] sequences
 * decoded by lanes, [8] sequences decoded the careful way.
 */
template <bool PROF = false>
__device__ __forceinline__ int decode_block_lanes(const uint8_t *in, int src_size, uint8_t *out, int out_size, int lane, uint32_t *lds,
                                                  unsigned long long *pc = nullptr)
{
    if (out_size == 0) {                                   /* LL64.dec.cs:162-168 */
        if (src_size == 1) {
            const uint32_t b = uni(lane == 0 ? (uint32_t)in[0] : 0u);
            return b == 0 ? 0 : -1;
        }
        return -1;
    }
    if (src_size <= 0) return -1;                          /* :172 */

    unsigned long long c_win = 0, c_size = 0, c_copy = 0, c_care = 0, n_win = 0, n_step = 0, n_lane = 0, n_care = 0;
    const unsigned long long t_begin = prof_now<PROF>();
    ParseWin win;
    win.init(lds, in, (uint32_t)src_size, lane);
    uint32_t *fp = lds + LANES_OFF_FIRST, *basev = lds + LANES_OFF_BASE, *prog = lds + LANES_OFF_PROG;
    const int64_t iend = src_size, oend = out_size;
    const uint32_t a0 = win.a0, IEND = (uint32_t)src_size + a0;
    int64_t ip = 0, op = 0;
    bool tail = false;                                     /* the end of the output is near: one sequence at a time from here on */
    int ret = 0;

    for (;;) {
        bool progressed = false;
        if (!tail && ip < iend - 16) {
            unsigned long long tm = 0;
            uint32_t end_ip = 0;
            const unsigned long long t0 = prof_now<PROF>();
            uint32_t ntok = parse_window<false, false>(win, (uint32_t)ip, (uint32_t)(iend - 16), lane, lds, end_ip, nullptr, &tm);
            const uint32_t W = (uint32_t)ip + a0;
            const uint32_t seg0 = W + 64u * (uint32_t)lane;
            if (PROF) { c_win += prof_now<PROF>() - t0; n_win++; }
            /* as long as the window has tokens left: the lane-parallel part, then one careful sequence where it stopped */
            while (ntok) {
                const unsigned long long t1 = prof_now<PROF>();
                /* ---- SIZES ---- */
                const uint32_t first = tm ? seg0 + (uint32_t)__ffsll((long long)tm) - 1u : LANES_NONE;
                fp[lane] = first;
                if (lane < 8) fp[64 + lane] = end_ip + a0;
                lds_sync();
                uint32_t after = LANES_NONE;               /* where the chain goes after this lane's last token */
                if (tm) {
                    uint32_t k = (uint32_t)lane + 1u;
                    after = fp[k];
                    while (after == LANES_NONE) after = fp[++k];   /* fp[64..71] = end of the window's chain: found there at the latest */
                }
                uint32_t sum = 0, cnt = 0, cut = LANES_NONE;
                {
                    unsigned long long m = tm;
                    while (__ballot(m != 0ull && cut == LANES_NONE)) {
                        if (m != 0ull && cut == LANES_NONE) {
                            const uint32_t P = seg0 + (uint32_t)__ffsll((long long)m) - 1u;
                            m &= m - 1ull;
                            const LaneSeq q = lane_seq(win, P, IEND);
                            const uint32_t succ = m ? seg0 + (uint32_t)__ffsll((long long)m) - 1u : after;
                            if (q.simple && q.next == succ) { sum += q.L + q.mlen; cnt++; }
                            else cut = P;
                        }
                    }
                }
                /* the first token (in stream order) that is not of the common kind ends the lane-parallel part */
                uint32_t wcut = cut;
#pragma unroll
                for (int sh = 1; sh < 64; sh <<= 1) {
                    const uint32_t o = (uint32_t)__shfl_xor((int)wcut, sh);
                    wcut = o < wcut ? o : wcut;
                }
                if (seg0 > wcut) { sum = 0; cnt = 0; }     /* (a lane's own cut is in its segment: lanes at or below the cut keep their sums) */
                const uint32_t incl = wave_inclusive_scan(sum);
                const uint32_t total = readlane_u32(incl, 63);
                const uint32_t ntake = readlane_u32(wave_inclusive_scan(cnt), 63);
                if (PROF) c_size += prof_now<PROF>() - t1;
                if (ntake == 0u && wcut == LANES_NONE) break;              /* (a list without tokens: let the careful parser look) */
                if (ntake != 0u && op + (int64_t)total > oend - 64) {      /* :427-443 the end-of-block rules start to matter */
                    tail = true;
                    break;
                }
                if (ntake != 0u) {
                    /* ---- COPY ---- */
                    const unsigned long long t2 = prof_now<PROF>();
                    const uint32_t o0 = (uint32_t)op;      /* the window's output starts here; everything below is final */
                    const uint32_t base = o0 + incl - sum;
                    basev[lane] = base;
                    prog[lane] = base;
                    if (lane < 8) basev[64 + lane] = o0 + total;
                    lds_sync();
                    unsigned long long m = cnt ? tm : 0ull;
                    if (cut != LANES_NONE && m) m &= (1ull << (cut - seg0)) - 1ull;   /* own tokens below the cut only */
                    uint32_t cur = base;
                    bool have = false, viol = false;       /* have: literals of the current sequence done, its match is waiting */
                    LaneSeq q{};
                    uint32_t need_lane = 0, need_pos = 0, viol_ip = LANES_NONE, viol_op = 0;
                    while (__ballot(m != 0ull || have)) {
                        if (PROF) n_step++;
                        if (!have && m) {
                            const uint32_t tokP = seg0 + (uint32_t)__ffsll((long long)m) - 1u;
                            q = lane_seq(win, tokP, IEND);
                            if (q.offset > cur + q.L) {    /* :338 the match starts before the output: the careful way reports it */
                                viol = true; viol_ip = tokP - a0; viol_op = cur;
                                m = 0ull;
                                cur = basev[lane + 1];     /* (nobody may wait for this lane any more) */
                            } else {
                                lane_copy_any(out + cur, in + (q.lpos - a0), q.L, (uint32_t)src_size - (q.lpos - a0));
                                cur += q.L;
                                have = true;
                                /* what the match waits for: the last source byte that another lane of this window writes */
                                const uint32_t s = cur - q.offset;
                                uint32_t e = s + q.mlen;
                                if (e > base) e = base;    /* from `base` on it is this lane's own output */
                                need_pos = 0;
                                if (e > o0 && s < base) {  /* owner of byte e - 1: the last lane whose base is <= e - 1 */
                                    uint32_t lo = 0;
#pragma unroll
                                    for (uint32_t stepw = 32; stepw != 0; stepw >>= 1)
                                        if (basev[lo + stepw] <= e - 1u) lo += stepw;
                                    need_lane = lo;
                                    need_pos = e;
                                }
                            }
                        }
                        lds_sync();
                        bool ready = have;
                        if (have && need_pos) {
                            /* the owner of the last source byte must be past it; if the source starts below that lane's
                             * region, every lane from the owner of its first byte up to there must be finished */
                            ready = prog[need_lane] >= need_pos;
                            const uint32_t s = cur - q.offset > o0 ? cur - q.offset : o0;
                            if (ready && s < basev[need_lane]) {
                                uint32_t lo = 0;
#pragma unroll
                                for (uint32_t stepw = 32; stepw != 0; stepw >>= 1)
                                    if (basev[lo + stepw] <= s) lo += stepw;
                                for (uint32_t l2 = lo; l2 < need_lane && ready; l2++) ready = prog[l2] >= basev[l2 + 1u];
                            }
                        }
                        if (ready) {
                            lane_match_copy(out, cur, q.offset, q.mlen, (uint32_t)out_size);
                            cur += q.mlen;
                            m &= m - 1ull;
                            have = false;
                        }
                        wave_sync();                       /* this step's stores before the next step's loads */
                        prog[lane] = cur;
                        lds_sync();
                    }
                    const unsigned long long vmask = __ballot(viol);
                    if (PROF) { c_copy += prof_now<PROF>() - t2; n_lane += ntake; }
                    if (vmask) {                           /* go back to the first sequence that broke the rule */
                        uint32_t vi = viol ? viol_ip : LANES_NONE, vo = viol_op;
#pragma unroll
                        for (int sh = 1; sh < 64; sh <<= 1) {
                            const uint32_t oi = (uint32_t)__shfl_xor((int)vi, sh), oo = (uint32_t)__shfl_xor((int)vo, sh);
                            if (oi < vi) { vi = oi; vo = oo; }
                        }
                        ip = (int64_t)vi;
                        op = (int64_t)vo;
                        break;
                    }
                    op += total;
                    progressed = true;
                }
                if (wcut == LANES_NONE) {                  /* the whole list is done */
                    ip = end_ip;
                    break;
                }
                /* the sequence at the cut, the careful way; then on with the rest of the window if the chain still fits */
                ip = (int64_t)(wcut - a0);
                {
                    const unsigned long long t3 = prof_now<PROF>();
                    ret = careful_sequence(win, in, out, ip, op, iend, oend, lane);
                    if (PROF) { c_care += prof_now<PROF>() - t3; n_care++; }
                    if (ret != 0) goto finished;
                    progressed = true;
                }
                {
                    const uint32_t cs = (wcut - W) >> 6, cb = (wcut - W) & 63u;
                    if ((uint32_t)lane < cs) tm = 0ull;
                    else if ((uint32_t)lane == cs) tm &= cb == 63u ? 0ull : ~0ull << (cb + 1u);
                    uint32_t nxt = tm ? seg0 + (uint32_t)__ffsll((long long)tm) - 1u : LANES_NONE;
#pragma unroll
                    for (int sh = 1; sh < 64; sh <<= 1) {
                        const uint32_t o = (uint32_t)__shfl_xor((int)nxt, sh);
                        nxt = o < nxt ? o : nxt;
                    }
                    ntok = readlane_u32(wave_inclusive_scan((uint32_t)__popcll(tm)), 63);
                    /* the rest stands if the careful parser arrived exactly at the next token of the list */
                    if (nxt == LANES_NONE || (int64_t)(nxt - a0) != ip || !(ip < iend - 16)) ntok = 0;
                }
            }
        }
        if (progressed) continue;
        /* nothing lane-parallel to do here (no token the chains could follow, the tail of the block, a broken rule): one
         * sequence the careful way */
        {
            const unsigned long long t3 = prof_now<PROF>();
            ret = careful_sequence(win, in, out, ip, op, iend, oend, lane);
            if (PROF) { c_care += prof_now<PROF>() - t3; n_care++; }
            if (ret != 0) goto finished;
        }
    }
finished:
    if (PROF && pc && lane == 0) {
        pc[0] = prof_now<PROF>() - t_begin; pc[1] = c_win; pc[2] = c_size; pc[3] = c_copy; pc[4] = c_care;
        pc[5] = n_win; pc[6] = n_step; pc[7] = n_lane; pc[8] = n_care;
    }
    return ret < 0 ? ret : (int)op;
}

constexpr int LANES_WAVES_PER_WG = 2;

__global__ __launch_bounds__(64 * LANES_WAVES_PER_WG) void k4_decode_lanes_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[LANES_WAVES_PER_WG][LANES_LDS_DWORDS];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const long long slot = (long long)blockIdx.x * LANES_WAVES_PER_WG + (long long)wave;
    if (slot >= a.n) return;
    const long long b = a.order ? (long long)uni(a.order[slot]) : slot;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    int ret = 0;
    if (src_len > 0 || (a.flags & FLAG_RAW_RETURN))
        ret = decode_block_lanes(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, lane, lds[wave]);
    if (lane == 0) a.outLen[b] = codec_decode_result(src_len, ret, a.flags);
}

/* diagnostic twin: per-phase cycle counters (a.prof, PROF_STRIDE per block; see decode_block_lanes) */
__global__ __launch_bounds__(64 * LANES_WAVES_PER_WG) void k4_decode_lanes_prof_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[LANES_WAVES_PER_WG][LANES_LDS_DWORDS];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const long long b = (long long)blockIdx.x * LANES_WAVES_PER_WG + (long long)wave;
    if (b >= a.n) return;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    int ret = 0;
    if (src_len > 0) ret = decode_block_lanes<true>(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, lane, lds[wave], a.prof + PROF_STRIDE * b);
    if (lane == 0) a.outLen[b] = codec_decode_result(src_len, ret, a.flags);
}

}  // namespace k4
