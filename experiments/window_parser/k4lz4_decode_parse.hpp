/*
 * k4lz4_decode_parse.hpp -- finding the token positions of an LZ4 block 64 chains at a time.
 *
 * What the reference does one token after the other (LL64.LZ4_decompress_generic, Engine/x64/LL64.dec.cs:175-451:
 * token -> literal length -> skip the literals -> offset -> match length -> next token) is a dependent chain; a
 * wavefront that follows it token by token uses one lane and pays the chain's latency per sequence.  Here a
 * window of the compressed stream (PARSE_NL segments of 64 bytes) is parsed by all lanes at once:
 *
 *   MAIN   lane j follows the chain that starts at the first byte of segment j -- a guess, only lane 0 starts on
 *          a real token -- to the end of its segment and marks the positions it visits (64-bit mask per lane).
 *          A chain is a pure function of the position it starts at, and chains that meet stay together: a wrong
 *          guess falls into step with the real tokens after a few hops.
 *   EXT    each lane goes on past its segment until it lands on a position the owner of that segment marked
 *          (from there on the two chains are the same), remembering the positions it visited on the way.
 *   WALK   the real chain is then read off: lane 0's marks, its extension, the lane it merged into from the merge
 *          position on, that lane's extension, ...  Where an extension gave up before merging (its list is
 *          full), the walk follows the chain itself, one token at a time, until it meets a mark again.
 *   LIST   the real token positions (a bit set over the window) are compacted into a list in LDS.
 *
 * decode_block then takes up to 64 positions of the list per batch and derives the sequence fields of all of
 * them at once -- every lane works on a real token, instead of 64 hypotheses for the dozen tokens that a window
 * of 64 bytes holds.  The list is advisory: every token derived from it is checked against the reference's
 * rules, tokens that need more (long lengths, block end, anything malformed) go through the scalar parser, and a
 * list that disagrees with where the scalar parser ends up is dropped.
 */
#pragma once
#include "k4lz4_common.hpp"

namespace k4 {

/* A window over the compressed stream in LDS: DW dwords (power of two), slot = dword index & (DW - 1), filled
 * with coalesced dword loads; slot DW mirrors slot 0, so two consecutive dwords can always be read as one
 * ds_read2_b32.  Positions given to cover()/read4() are "aligned" positions: stream position + a0. */
template <int DW> struct StreamWin {
    uint32_t *ring;        /* DW + 1 dwords */
    const uint32_t *base;  /* dword-aligned address at or below the first stream byte */
    const uint8_t *bytes;  /* the stream itself */
    uint32_t a0;           /* misalignment of the stream start: 0..3 */
    uint32_t ndw;          /* dwords that contain stream bytes */
    uint32_t len;          /* stream bytes */
    uint32_t rlo, rhi;     /* dwords [rlo, rhi) are in the ring (multiples of 64, rhi - rlo <= DW) */

    __device__ __forceinline__ uint32_t load(uint32_t dw) const { return dw < ndw ? base[dw] : 0u; }

    __device__ __forceinline__ void init(uint32_t *lds, const uint8_t *in, uint32_t n, int lane)
    {
        ring = lds;
        bytes = in;
        len = n;
        a0 = (uint32_t)((uintptr_t)in & 3u);
        base = (const uint32_t *)(in - a0);
        ndw = (a0 + n + 3u) >> 2;
        rlo = rhi = 0u;
        (void)lane;
    }

    /* make the ring hold the aligned byte positions [qlo, qhi); qhi - qlo <= 4 * DW - 512 */
    __device__ __forceinline__ void cover(uint32_t qlo, uint32_t qhi, int lane)
    {
        const uint32_t dlo = qlo >> 2, dhi = (qhi + 3u) >> 2;
        if (dlo >= rlo && dhi <= rhi) return;
        wave_sync();                                     /* earlier reads of the slots that are overwritten */
        if (dlo < rlo || dlo > rhi) rlo = rhi = dlo & ~63u;   /* a jump: start again there */
        while (rhi < dhi) {                               /* up to eight chunks' loads in flight before the first store */
            uint32_t v[8];
#pragma unroll
            for (uint32_t c = 0; c < 8u; c++) v[c] = rhi + 64u * c < dhi ? load(rhi + 64u * c + (uint32_t)lane) : 0u;
#pragma unroll
            for (uint32_t c = 0; c < 8u; c++) {
                if (rhi + 64u * c < dhi) {
                    const uint32_t slot = (rhi + 64u * c + (uint32_t)lane) & (uint32_t)(DW - 1);
                    ring[slot] = v[c];
                    if (slot == 0u) ring[DW] = v[c];
                }
            }
            const uint32_t todo = (dhi - rhi + 63u) >> 6;
            rhi += 64u * (todo < 8u ? todo : 8u);
        }
        if (rhi - rlo > (uint32_t)DW) rlo = rhi - (uint32_t)DW;
        wave_sync();
    }

    /* one past the last stream position that can be read from the ring */
    __device__ __forceinline__ uint32_t hi_pos() const { return (rhi << 2) - a0; }

    /* 4 stream bytes at aligned byte position q (per lane) */
    __device__ __forceinline__ uint32_t read4(uint32_t q) const
    {
        const uint32_t *s = ring + ((q >> 2) & (uint32_t)(DW - 1));
        const uint32_t lo = s[0], hi = s[1];
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> ((q & 3u) * 8u));
    }
    /* ... at stream position p */
    __device__ __forceinline__ uint32_t at(uint32_t p) const { return read4(p + a0); }

    /* one stream byte, from the ring where it has it, else from memory; 0 past the end of the stream */
    __device__ __forceinline__ uint32_t byte_at(uint32_t p) const
    {
        if (p >= len) return 0u;
        const uint32_t q = p + a0;
        if ((q >> 2) >= rlo && (q >> 2) < rhi) return (ring[(q >> 2) & (uint32_t)(DW - 1)] >> ((q & 3u) * 8u)) & 255u;
        return (uint32_t)bytes[p];
    }

    /* scalar parser: the 4 stream bytes at wave-uniform stream position p */
    __device__ __forceinline__ uint32_t fetch(uint32_t p, int lane)
    {
        cover(p + a0, p + a0 + 8u, lane);
        return uni(read4(p + a0));
    }
};

constexpr int PARSE_RING_DW = 1024;            /* 4 KiB of stream per parsing wave */
constexpr int PARSE_NL = 52;                   /* segments (chains) per window: 52 * 64 + slack < 4 KiB */
constexpr int PARSE_SLACK = 296;               /* bytes past the window that are kept in the ring */
constexpr int PARSE_EXT_TRIPS = 32;            /* steps a chain takes past its own segment before it gives up */
constexpr int PARSE_TOK_MAX = PARSE_NL * 22;   /* a sequence has at least 3 stream bytes */
/* LDS of the parser, in dwords: ring (+ mirror) | main marks (2 per lane) | real-token bit set | entry offsets.  The result --
 * the window's token list -- is written over the last three once they have served: one BYTE per token, its distance
 * from the token before it (255 = 255 or more: the caller knows where that token is by other means or not at all). */
constexpr int PARSE_OFF_MARK = PARSE_RING_DW + 2, PARSE_OFF_TRUE = PARSE_OFF_MARK + 128, PARSE_OFF_RIN = PARSE_OFF_TRUE + 128,
              PARSE_LDS_DWORDS = PARSE_OFF_RIN + 64, PARSE_OFF_LIST = PARSE_OFF_MARK;
static_assert(2 * PARSE_NL + 6 <= 128, "the real-token bit set has two dwords per segment plus the extension marks' reach");
static_assert(PARSE_TOK_MAX <= 4 * (PARSE_LDS_DWORDS - PARSE_OFF_LIST), "the token list fits over the parser's scratch");
constexpr uint32_t PARSE_FAR = 255u;
typedef StreamWin<PARSE_RING_DW> ParseWin;

/*
 * One step of a chain.  A chain's state is (pt, pend): pend = 1 says that the sequence before had a match
 * length of 19 or more (ML field 15), its extension byte is at pt - 1 and has not been looked at -- the token is at
 * pt only if that byte is not 255.  Keeping that byte for the next step's read means a step needs ONE look at the
 * stream: 4 bytes that hold (extension byte,) token and the first literal-length byte.  The step follows the
 * layout of a sequence only (LL64.dec.cs:177-178,:228-243,:318-336: token, 255-runs of the two lengths, 2 offset
 * bytes).  Returns the token's position (pt itself unless the extension ran on); (nx, npend) = the next state.
 * `fail`: the sequence does not lie inside the input or a length run is longer than chains follow -- such a
 * token ends the chain and is left to the scalar parser.
 */
__device__ __forceinline__ uint32_t chain_hop_slow(const ParseWin &win, uint32_t pt, uint32_t pend, uint32_t &nx, uint32_t &npend, bool &fail)
{
    uint32_t p = pt;
    fail = false;
    if (pend) {
        uint32_t e = win.byte_at(p - 1u);
        for (uint32_t k = 0; e == 255u && k < 24u; k++) e = win.byte_at(p++);
        if (e == 255u) fail = true;
    }
    const uint32_t tok = win.byte_at(p);
    uint32_t L = tok >> 4, q = p + 1u;
    if (L == RUN_MASK) {
        uint32_t e = 255u;
        for (uint32_t k = 0; e == 255u && k < 24u; k++) { e = win.byte_at(q++); L += e; }
        if (e == 255u) fail = true;
    }
    nx = q + L + 2u;
    npend = (tok & 15u) == ML_MASK ? 1u : 0u;
    nx += npend;
    if (nx > win.len) fail = true;
    return p;
}

/* the common case: at most one extension byte per length.  P = the expected token position in ALIGNED coordinates
 * (stream position + a0, which is what indexes the ring); LEN = stream length + a0.  One ds_read2_b32: the four bytes
 * from P - 1 on are (extension byte of the sequence before,) token, first literal-length byte.  `rare` = this lane
 * needs chain_hop_slow instead. */
__device__ __forceinline__ void chain_hop(const uint32_t *ring, uint32_t P, uint32_t pend, uint32_t LEN, uint32_t &nx, uint32_t &npend, bool &rare)
{
    const uint32_t a = P - 1u;
    const uint32_t *s = ring + ((a >> 2) & (uint32_t)(PARSE_RING_DW - 1));
    const uint32_t lo = s[0], hi = s[1];
    const uint32_t t = (uint32_t)((((uint64_t)hi << 32) | lo) >> ((a & 3u) * 8u));
    const uint32_t L = (t >> 12) & 15u, e1 = (t >> 16) & 255u;
    const bool g = L == RUN_MASK;
    npend = ((t >> 8) & 15u) == ML_MASK ? 1u : 0u;
    nx = P + (g ? 19u + e1 : 3u + L) + npend;
    rare = (pend != 0u && (t & 255u) == 255u) || (g && e1 == 255u) || nx > LEN;
}

/* the real chain from lane to lane: lane l's word = the lane its chain merged into (bits 0-5), or l itself with bit 7 set
 * where the chain did not merge.  From lane `from` on, marks every lane passed in T and returns the lane where the hops
 * end.  In ISA for the same reason as the encoder's hop chain: s_bitset1 and v_readlane take the lane from the low six
 * bits of the word just read, so one read feeds the next with nothing but the wait states in between; a lane where the
 * chain ends points at itself, so hopping on is harmless and the exit is tested once per eight hops. */
__device__ __forceinline__ uint32_t walk_lanes(uint32_t word, uint32_t from, unsigned long long &T)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t pk = from;
#define K4_WHOP "s_bitset1_b64 %[T], %[pk]\n\ts_nop 2\n\tv_readlane_b32 %[pk], %[word], %[pk]\n\t"
#define K4_WHOP8 K4_WHOP K4_WHOP K4_WHOP K4_WHOP K4_WHOP K4_WHOP K4_WHOP K4_WHOP
    asm volatile(
        ".Lwalk_more%=:\n\t"
        K4_WHOP8
        "s_bitcmp0_b32 %[pk], 7\n\t"
        "s_cbranch_scc1 .Lwalk_more%="
        : [T] "+s"(T), [pk] "+s"(pk)
        : [word] "v"(word)
        : "scc");
#undef K4_WHOP8
#undef K4_WHOP
    return pk & 63u;
#else
    uint32_t cur = from;
    for (;;) {
        T |= 1ull << cur;
        const uint32_t w = readlane_u32(word, (int)cur);
        if (w & 0x80u) return cur;
        cur = w & 63u;
    }
#endif
}

/*
 * One window: the token positions of the chain that starts at stream position wb (a real token), as far as the
 * window reaches.  Returns their number n; list[k] (bytes at area + PARSE_OFF_LIST) = distance of token k from token
 * k - 1 (list[0] = 0: the first token is at wb; PARSE_FAR = that far or more); end_ip = where the chain goes on after
 * them (the first position not in the list).  n == 0: the token at wb is left to the scalar
 * parser.  clim = iend - 16: chains only visit positions below it.  `area` = this wave's PARSE_LDS_DWORDS.
 */
template <bool PROF = false, bool LIST = true>
__device__ __forceinline__ uint32_t parse_window(ParseWin &win, uint32_t wb, uint32_t clim, int lane, uint32_t *area, uint32_t &end_ip,
                                                 unsigned long long *pc = nullptr, unsigned long long *bits = nullptr)
{
    /* diagnostics (PROF): pc[0..4] cycles of cover / MAIN / EXT / WALK / LIST, pc[5..9] their loop trips (WALK: hops, then
     * tokens followed one at a time) */
    unsigned long long tq0 = prof_now<PROF>(), n_main = 0, n_ext = 0, n_hop = 0, n_ser = 0, n_list = 0;
    uint32_t *pm = area + PARSE_OFF_MARK, *pt_ = area + PARSE_OFF_TRUE, *prin = area + PARSE_OFF_RIN;
    uint8_t *plist = (uint8_t *)(area + PARSE_OFF_LIST);
    const uint32_t *ring = win.ring;
    /* everything below in aligned coordinates (stream position + a0) */
    const uint32_t a0 = win.a0, W = wb + a0, CL = clim + a0, LEN = win.len + a0;
    const uint32_t span = clim - wb;
    const uint32_t nl = span > (uint32_t)PARSE_NL * 64u ? (uint32_t)PARSE_NL : (span + 63u) >> 6;
    const uint32_t WEND = W + nl * 64u;
    const uint32_t STOP = WEND < CL ? WEND : CL;          /* chains do not visit positions from here on */
    win.cover(W, WEND + (uint32_t)PARSE_SLACK, lane);
    const unsigned long long tq1 = prof_now<PROF>();

    /* ---- MAIN: every lane its own segment ---- */
    const uint32_t seg0 = W + 64u * (uint32_t)lane;
    const uint32_t seg_stop = seg0 + 64u < CL ? seg0 + 64u : CL;
    constexpr uint32_t LEFT = 0x80000000u;                 /* set in P: left the loop on a resolved token position */
    uint32_t P = (uint32_t)lane < nl ? seg0 : 0xffffffffu, pend = 0;      /* the state: token expected at P */
    unsigned long long mask = 0;
    bool stopped = false;
    for (;;) {
        const bool live = P < seg_stop;
        if (!__ballot(live)) break;
        if (PROF) n_main++;
        uint32_t nx, npend, tp = P;
        bool rare;
        chain_hop(ring, P, pend, LEN, nx, npend, rare);
        rare = rare && live;
        if (__ballot(rare)) {
            if (rare) {
                bool fail;
                tp = chain_hop_slow(win, P - a0, pend, nx, npend, fail) + a0;
                nx += a0;
                if (fail || tp >= seg_stop) {             /* the token is not this segment's after all, or cannot be followed */
                    stopped = fail || tp >= CL;
                    nx = tp | LEFT;
                    npend = 0;
                    tp = 0xffffffffu;
                }
            }
        }
        if (live) {
            if (tp != 0xffffffffu) mask |= 1ull << (tp - seg0);
            P = nx;
            pend = npend;
        }
    }
    if ((uint32_t)lane >= nl) P = WEND;
    else if (P & LEFT) P &= ~LEFT;
    else if (P >= CL && P < seg0 + 64u) stopped = true;
    const uint32_t mlo = (uint32_t)mask, mhi = (uint32_t)(mask >> 32);
    if (uni(mlo | mhi) == 0u) { end_ip = wb; return 0u; } /* the real chain's first token cannot be followed */
    pm[2 * lane] = mlo;
    pm[2 * lane + 1] = mhi;
    pt_[2 * lane] = 0u;
    pt_[2 * lane + 1] = 0u;
    lds_sync();

    /* ---- EXT: on until the chain meets the marks of a segment's owner ---- */
    const unsigned long long tq2 = prof_now<PROF>();
    enum : uint32_t { ST_RUN = 0, ST_MERGED = 1, ST_END = 2, ST_OPEN = 3 };
    uint32_t st = (uint32_t)lane < nl && !stopped ? ST_RUN : ST_END;
    unsigned long long e1 = 0, e2 = 0;                    /* positions visited in the next segment and the one after it */
    for (uint32_t trip = 0;; trip++) {
        const bool run = st == ST_RUN;
        if (!__ballot(run)) break;
        if (PROF) n_ext++;
        const bool off_end = P >= STOP;                   /* (P is approximate while an extension byte is pending: allowed) */
        uint32_t nx, npend, tp = P;
        bool rare, fail = false;
        chain_hop(ring, P, pend, LEN, nx, npend, rare);
        rare = rare && run && !off_end;
        if (__ballot(rare)) {
            if (rare) {
                tp = chain_hop_slow(win, P - a0, pend, nx, npend, fail) + a0;
                nx += a0;
            }
        }
        const uint32_t r = (tp - W) & 4095u;
        const uint32_t mw = pm[r >> 5];
        const uint32_t rel = tp - seg0 - 64u;             /* 0..127: inside the two segments this lane keeps marks for */
        if (run) {
            if (off_end) {
                st = ST_END;
            } else if (fail || tp >= STOP) {
                st = ST_END; P = tp; pend = 0;
            } else if ((mw >> (r & 31u)) & 1u) {
                st = ST_MERGED; P = tp; pend = 0;
            } else if (rel >= 128u || trip >= (uint32_t)PARSE_EXT_TRIPS) {
                st = ST_OPEN; P = tp; pend = 0;
            } else {
                const unsigned long long bit = 1ull << (rel & 63u);
                if (rel < 64u) e1 |= bit; else e2 |= bit;
                P = nx; pend = npend;
            }
        }
    }

    /* ---- WALK: the real chain, from lane to lane ---- */
    const unsigned long long tq3 = prof_now<PROF>();
    const uint32_t word = st == ST_MERGED ? (P - W) >> 6 : 0x80u | (uint32_t)lane;
    unsigned long long onwalk = 0;
    uint32_t cur = 0, endp;
    for (;;) {
        cur = walk_lanes(word, cur, onwalk);
        if (PROF) n_hop++;
        uint32_t s = readlane_u32(st, (int)cur);
        uint32_t x = readlane_u32(P, (int)cur);
        if (s == ST_OPEN) {                               /* its list was full: follow the chain itself */
            uint32_t xp = 0;
            for (;;) {
                if (x >= STOP) { s = ST_END; break; }
                if (PROF) n_ser++;
                uint32_t nx, npend, tp = x;
                bool rare, fail = false;
                chain_hop(ring, x, xp, LEN, nx, npend, rare);
                if (uni(rare ? 1u : 0u)) {
                    tp = uni(chain_hop_slow(win, x - a0, xp, nx, npend, fail)) + a0;
                    nx += a0;
                    if (uni(fail ? 1u : 0u) || tp >= STOP) { x = tp; s = ST_END; break; }
                }
                const uint32_t r = tp - W;
                if ((uni(pm[r >> 5]) >> (r & 31u)) & 1u) { x = tp; s = ST_MERGED; break; }
                if (lane == 0) atomicOr(&pt_[r >> 5], 1u << (r & 31u));
                x = uni(nx);
                xp = uni(npend);
            }
            if (s == ST_MERGED) {
                cur = (x - W) >> 6;
                if (lane == 0) prin[cur] = (x - W) & 63u;             /* entry offset handed over directly */
                continue;
            }
        }
        endp = x;
        break;
    }

    /* ---- LIST: bit set of the real tokens -> ascending positions ---- */
    const unsigned long long tq4 = prof_now<PROF>();
    const bool on = ((onwalk >> lane) & 1ull) != 0;
    lds_sync();
    if (on && st == ST_MERGED) prin[(P - W) >> 6] = (P - W) & 63u;   /* tell the lane this one merged into where the real chain enters it */
    lds_sync();
    if (on) {
        const uint32_t rin = lane == 0 ? 0u : prin[lane];
        const unsigned long long mine = mask & (~0ull << rin);
        if ((uint32_t)mine) atomicOr(&pt_[2 * lane], (uint32_t)mine);
        if ((uint32_t)(mine >> 32)) atomicOr(&pt_[2 * lane + 1], (uint32_t)(mine >> 32));
        if ((uint32_t)e1) atomicOr(&pt_[2 * lane + 2], (uint32_t)e1);
        if ((uint32_t)(e1 >> 32)) atomicOr(&pt_[2 * lane + 3], (uint32_t)(e1 >> 32));
        if ((uint32_t)e2) atomicOr(&pt_[2 * lane + 4], (uint32_t)e2);
        if ((uint32_t)(e2 >> 32)) atomicOr(&pt_[2 * lane + 5], (uint32_t)(e2 >> 32));
    }
    lds_sync();
    unsigned long long tm = (uint32_t)lane < nl ? ((unsigned long long)pt_[2 * lane + 1] << 32) | pt_[2 * lane] : 0ull;
    const uint32_t c = (uint32_t)__popcll(tm);
    const uint32_t incl = wave_inclusive_scan(c);
    const uint32_t total = readlane_u32(incl, 63);
    if (!LIST) {                                          /* the caller works from the segments' bit sets themselves */
        *bits = tm;
        end_ip = endp - a0;
        return total;
    }
    /* the token before this lane's first one: the last token of the nearest lane below that has any */
    const uint32_t rel0 = 64u * (uint32_t)lane;
    uint32_t last1 = tm ? rel0 + 64u - (uint32_t)__clzll((long long)tm) : 0u;   /* position + 1 of this lane's last token */
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)last1, (unsigned)sh);
        if (lane >= sh && o > last1) last1 = o;
    }
    uint32_t prev = (uint32_t)__shfl_up((int)last1, 1u);   /* (lane 0: its own, unused) */
    prev = lane == 0 ? 0u : prev;                          /* position + 1 of the token before, 0 = none */
    lds_sync();                                            /* every lane has its bits: the list may now go over them */
    uint8_t *dst = plist + (incl - c);
    while (__ballot(tm != 0ull)) {                         /* four per trip: the exit test costs as much as a position */
        if (PROF) n_list++;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (tm) {
                const uint32_t pos = rel0 + (uint32_t)__ffsll((long long)tm) - 1u;
                const uint32_t d = prev ? pos + 1u - prev : 0u;
                *dst++ = (uint8_t)(d < PARSE_FAR ? d : PARSE_FAR);
                prev = pos + 1u;
                tm &= tm - 1ull;
            }
        }
    }
    lds_sync();
    if (PROF && pc && lane == 0) {
        const unsigned long long tq5 = prof_now<PROF>();
        pc[0] += tq1 - tq0; pc[1] += tq2 - tq1; pc[2] += tq3 - tq2; pc[3] += tq4 - tq3; pc[4] += tq5 - tq4;
        pc[5] += n_main; pc[6] += n_ext; pc[7] += n_hop; pc[8] += n_ser; pc[9] += n_list;
    }
    end_ip = endp - a0;
    return total;
}

}  // namespace k4
