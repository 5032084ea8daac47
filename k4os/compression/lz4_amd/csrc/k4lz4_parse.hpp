/*
 * k4lz4_parse.hpp -- the fast-level encoder in two kernels: PARSE (which sequences) and EMIT (their bytes).
 *
 * Replaces, for batches of blocks below 65 547 bytes at acceleration 1 -- LZ4Codec.Encode's case
 * (src/K4os.Compression.LZ4/LZ4Codec.cs:40-52 -> Engine/LLxx.cs:65-75) --, the same reference code as
 * k4lz4_encode_fast.hpp:
 *   LL64.LZ4_compress_generic, byU16 arm    Engine/x64/LL64.fast.cs:34-513
 *   hash4 / table get / put                 Engine/LL.tools.cs:46-51,:80-148
 *   LZ4_count                               Engine/x64/LL64.tools.cs:86-133
 * with byte-identical output.  Blocks outside that case are left to the kernels of k4lz4_encode_fast.hpp
 * (k4_encode_fast_rest_kernel below).
 *
 * Why two kernels.  One wavefront owns one block and its time is one long chain of dependent instructions; what the
 * reference's loop decides -- where a match starts, which earlier position it refers to, how long it is -- is
 * 8 bytes per sequence, and everything else (backward extension LL64.fast.cs:237-242, token and length bytes :244-382,
 * the literal copies, the output-limit checks) follows from those 8 bytes and the source alone.  k4_parse_kernel
 * writes the 8 bytes; k4_emit_kernel, a throughput kernel, turns them into the block.
 *
 * How the parse works.  A ROUND looks at 64 * K consecutive probe positions from the cursor (K sub-windows of one
 * position per lane); all of a round's trips to memory -- the source bytes, the table look-ups, the 16 bytes around every
 * candidate -- are made for the K sub-windows together, before the first decision, so a round waits for each of them
 * once per 64 * K positions.  The decisions are then a scalar chain over one sub-window after the other: first lane at
 * or after the cursor that stops the chain -> lane after its match (k4lz4_encode_fast.hpp, hop_chain).
 *   A probe's candidate is the table entry as it stood when the round began unless an earlier VISITED position of the
 * round has the same hash.  Lanes whose hash another lane of the round shares are found through one LDS bit per hash
 * value; such a lane stops the chain, and its candidate is worked out right there from what has been visited so far --
 * lazily, in scalar code, for the few lanes the cursor really reaches -- instead of for every lane up front.
 *   The puts of a round (every visited position, and position end-2 behind every match, LL64.fast.cs:394) are made at
 * its end: lanes nobody shares a hash with in one store, the others one by one in position order.
 * After 66 probes without a match the reference's step grows (LL64.fast.cs:156-172): those rounds probe the strided
 * positions with the same machinery and end at their first match.
 */
#pragma once
#include "k4lz4_encode_fast.hpp"
#include <type_traits>
#ifdef K4_PARSE_DEBUG
#include <stdio.h>
#include <stdlib.h>
#endif

namespace k4 {

#ifndef K4_PARSE_K
#define K4_PARSE_K 1
#endif

/* records per block.  A sequence is at least four bytes of match, the block's first byte and its last five are literals, so a block
 * below LIMIT_64K has at most (65546 - 6) / 4 = 16 385 sequences -- on paper: that many would need nothing but 4-byte matches from
 * the second byte on, and a match needs its bytes to have stood there before (the densest block the tests could build,
 * tests/adversarial_blocks.py, has 10 500).  The slot holds the paper bound plus a round's worth, so nothing hangs on that argument. */
constexpr uint32_t PARSE_REC_STRIDE = 16448u;
static_assert(PARSE_REC_STRIDE >= ((uint32_t)LIMIT_64K - 1u - 6u) / 4u + 1u + 62u, "a block below LIMIT_64K has at most (LIMIT_64K - 1 - 6) / 4 sequences; a round adds at most 64 / 4 + 1");
constexpr uint32_t PARSE_MIN_LEN = 128u;             /* shorter blocks go to the other kernels (the clamped loads below want 16 readable bytes somewhere) */
constexpr uint32_t PARSE_REST = 0xffffffffu;         /* meta[2 b]: this block is for k4_encode_fast_rest_kernel */
constexpr uint32_t PARSE_BIG = 0xfffffffeu;          /* ... for k4_parse_big_kernel (65 547 bytes and more: byU32 table), which then writes its count */
constexpr int PARSE_MAX_WAVES = 16;                  /* waves (= blocks) per workgroup: one workgroup per CU */
constexpr int PARSE_LDS_TABLES = 9;                  /* of which so many have their table in LDS (9 x 16 KiB + 16 x 512 B of 160 KiB) */
constexpr int PARSE_SEEN_DWORDS = 128;               /* one bit per two hash values */

constexpr int PARSE_LDS_DWORDS = 4096 * PARSE_LDS_TABLES + PARSE_SEEN_DWORDS * PARSE_MAX_WAVES + 4;      /* + the workgroup's free-table word */

struct ParseArgs {
    uint2 *recs;            /* PARSE_REC_STRIDE records per block: x = position of the match (before backward extension),
                             * y = offset | (match length - MINMATCH) << 16 */
    uint32_t *meta;         /* per block: [0] number of sequences or PARSE_REST, [1] spare */
    uint32_t *gtab;         /* 4096 dwords per (workgroup, wave) for the waves without an LDS table */
    uint32_t nwg;           /* workgroups of the parse launch: block of (workgroup w, wave s) = order[s * nwg + w] */
    uint32_t migrate;       /* != 0: blocks without an LDS table move into one when a block of their workgroup is done with it (ParseCtl) */
    uint32_t inline_emit;   /* != 0: the wave that parsed a block writes it out as well (k4_emit_kernel is not launched): the blocks that are
                             * through early do that while the others still parse, only the last ones' bytes come on top of the launch */
    uint32_t *nbig;         /* nullptr, or a zeroed word: k4_parse_kernel counts the blocks it marks PARSE_BIG there, and the launch behind it leaves at
                             * once when there are none (a batch of 262 144 small blocks otherwise costs it a ticket per block for nothing) */
    uint32_t big;           /* != 0: k4_parse_big_kernel follows this launch and takes the blocks of 65 547 bytes and more (needs inline_emit) */
    uint32_t slot_recs;     /* != 0 (with inline_emit): a block's records are written out by the wave that made them, right behind its parse,
                             * so `recs` holds one slot per WAVE of the launch (workgroup x waves per workgroup + wave) instead of one per block:
                             * a launch's scratch is what is resident, whatever the number of blocks */
    uint32_t *queue;        /* nullptr, or three zeroed words: the launch has fewer waves than blocks and every wave takes the next block
                             * when it is done with one -- [0] tickets handed out (never more than there are blocks), [1] taken from the front
                             * of the order (the most expensive: by the waves with a table in LDS), [2] taken from its back (by the others) */
};

/* ------------------------------------------------------------------------------------------------------------------ */

template <int N> using ic = std::integral_constant<int, N>;
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) { return ((unsigned long long)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }

/* bytes [s, s+16) of a stream of which w holds [a, a+16), a <= s (zeros behind the end) */
__device__ __forceinline__ void shift16(uint32_t &w0, uint32_t &w1, uint32_t &w2, uint32_t &w3, uint32_t sh)
{
    if (sh == 0u) return;
    unsigned long long lo = ((unsigned long long)w1 << 32) | w0, hi = ((unsigned long long)w3 << 32) | w2;
    if (sh >= 16u) { lo = 0; hi = 0; }
    else if (sh >= 8u) { lo = sh == 8u ? hi : hi >> (8u * (sh - 8u)); hi = 0; }
    else { lo = (lo >> (8u * sh)) | (hi << (64u - 8u * sh)); hi >>= 8u * sh; }
    w0 = (uint32_t)lo; w1 = (uint32_t)(lo >> 32); w2 = (uint32_t)hi; w3 = (uint32_t)(hi >> 32);
}

/* equal bytes of two 12-byte strings given as xors of their words (no branches: every lane of a round does this) */
__device__ __forceinline__ uint32_t ext12(uint32_t x0, uint32_t x1, uint32_t x2)
{
    const uint32_t t0 = min((uint32_t)(__ffs((int)x0) - 1), 32u);       /* __ffs(0) - 1 = 0xffffffff */
    const uint32_t t1 = min((uint32_t)(__ffs((int)x1) - 1), 32u);
    const uint32_t t2 = min((uint32_t)(__ffs((int)x2) - 1), 32u);
    const uint32_t hi = t1 + (t1 == 32u ? t2 : 0u);
    return (t0 + (t0 == 32u ? hi : 0u)) >> 3;
}

/* the same over 28 bytes (seven words) */
__device__ __forceinline__ uint32_t ext28(const uint32_t *x)
{
    uint32_t acc = min((uint32_t)(__ffs((int)x[6]) - 1), 32u);
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        const uint32_t t = min((uint32_t)(__ffs((int)x[i]) - 1), 32u);
        acc = t + (t == 32u ? acc : 0u);
    }
    return acc >> 3;
}

/* a sequence record: written once by the parse, read once by the write-out */
__device__ __forceinline__ void rec_store(uint2 *r, uint32_t x, uint32_t y)
{
#if defined(K4_NT_RECS) && !defined(K4_HOST_EMU)
    __builtin_nontemporal_store((((unsigned long long)y) << 32) | x, (unsigned long long *)r);
#else
    *r = make_uint2(x, y);
#endif
}
__device__ __forceinline__ uint2 rec_load(const uint2 *r)
{
#if defined(K4_NT_RECS) && !defined(K4_HOST_EMU)
    const unsigned long long v = __builtin_nontemporal_load((const unsigned long long *)r);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
#else
    return *r;
#endif
}

/* hop word: bits 0-6 lane after the match (64 and more: outside the sub-window, 127 = "127 or more"), and the reasons to
 * leave the tight chain (all inside hop_chain's 0xf40): 0x40 = bit 6 of the lane, 0x100 the match runs past the 12 known
 * bytes, 0x200 a lane that shares its hash with an earlier lane of the round (0x1000 on top: its table candidate is no
 * hit), 0x400 the block ends behind this match (LL64.fast.cs:391), 0x800 no probe here: the block's last literals begin
 * (LL64.fast.cs:172) */
constexpr uint32_t HOP_LONG = 0x100u, HOP_LAZY = 0x200u, HOP_END = 0x400u, HOP_INVALID = 0x800u, HOP_TABMISS = 0x1000u;

/*
 * LL64.LZ4_compress_generic (byU16, noDict, acceleration 1) for one block, sequences only.  `tab`: the block's 8192 x u16
 * table (LDS, or global memory with GT), `seen`: PARSE_SEEN_DWORDS dwords of LDS.  Returns the number of records written.
 */
struct ParseStats { uint32_t rounds, slow, groups, lazies, longs; };
constexpr int PARSE_FLUSH = -2;      /* ParseCtl::claimed: the record slot is nearly full -- write the records out, then resume */

/* A block whose table lives in memory may move into an LDS table that another block of its workgroup has finished with: the
 * waves with LDS tables set their bit in *free_slots when their parse is through; a wave without one looks at the word every 16
 * rounds, claims a bit (atomic AND) and returns from parse_block with its state here; the kernel copies the table from memory into
 * the slot and calls parse_block's LDS form with `resume`.  The encoder's state between two rounds is just these words. */
struct ParseCtl {
    uint32_t *free_slots;       /* LDS word of the workgroup, or nullptr */
    uint32_t rec_cap;           /* in: 0, or the records the block's slot holds -- parse_block returns with claimed = PARSE_FLUSH once a round may overrun it */
    int claimed;                /* out: the slot claimed (parse_block returned early), PARSE_FLUSH, else -1 */
    bool resume;                /* in: go on from the state below (the table is in place) */
    uint32_t c, sbase, sj, nrec, long_seen, long_rounds;
    bool test, more;
    uint32_t cut_pos;           /* segments (SegRun): where the run's next cut lies, whether it is still warming up; */
    bool dry, stopped;          /* out: parse_block returned because the run stopped at its boundary (sr->state says how), not at the block's end */
};      /* what a DRY run counts (the cost estimate's input) */

/* TT: the table and hash of the block (k4lz4_encode_fast.hpp, FastTable): 1 = byU16 + hash4 (blocks below LIMIT_64K: the case this file
 * was written for), 0 = byU32 + hash5 (LL64.fast.cs:526-544: 65 547 bytes and more), 2 = byU32 + hash4 (LL32: LZ4Codec.Enforce32).  What
 * changes with a byU32 table (round 6): a candidate more than 65 535 bytes back is no match (LL64.fast.cs:219-224; it is still put over),
 * one bit of `seen` per hash value, hash5 reads five bytes, and a match length no longer fits the record's 16 bits -- 0xffff there says
 * "this or more", counted again where the record is written out.  The record slot of a block holds PARSE_REC_STRIDE records: a longer
 * block returns to its caller when the slot is nearly full (ParseCtl::flush, like a table migration), has them written out and goes on. */
/* SEG (byU32 blocks only): the block is one of several runs over a big block (k4lz4_encode_fast.hpp, SegRun; k4lz4_segments.hpp says who
 * gets what).  The run starts at sr->begin as a fresh encoder would; a warm run (emit_from != 0) writes no record before its CUT -- its
 * first match end at or behind emit_from --, where it publishes cut and table and goes on as the block's encoder; every run stops at
 * its first match end at or behind stop_at, and says whether the next run's published cut and table are that very state.  A round
 * may end at any match end (a match that runs out of the window does just that), which is all the rounds need for it: hop words of
 * matches that end at or behind the next such position carry HOP_END, and the chain's exits look at where the match ended. */
template <int K, bool GT, bool DRY = false, int TT = 1, bool SEG = false>
__device__ __forceinline__ uint32_t parse_block(const uint8_t *src, const uint32_t U, uint2 *recs, void *tabv, uint32_t *seen, const int lane, unsigned long long *pc = nullptr,
                                                ParseStats *stats = nullptr, ParseCtl *ctl = nullptr, SegRun *sr = nullptr)
{
    static_assert(!SEG || (TT != 1 && K == 1 && !DRY), "segments: byU32 blocks, one sub-window per round");
    typedef FastTable<TT> Table;
    static_assert(TT == 1 || K == 1, "byU32 tables: one sub-window per round");
    constexpr bool U32 = TT != 1;
    Table tab;
    tab.t = (decltype(tab.t))tabv;
    constexpr uint32_t SEEN_SHIFT = TT == 1 ? 1u : 0u;          /* hash value -> bit of `seen` */
    ParseStats st = {0u, 0u, 0u, 0u, 0u};
#define K4_ST(field, v) do { if (DRY) st.field += (v); } while (0)
#ifdef K4_PARSE_PROF
    /* diagnostic build: cycles per phase of a round (each phase ends by draining its own memory traffic) and event counts */
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pn[8] = {0, 0, 0, 0, 0, 0, 0, 0}, plast = 0;
    const unsigned long long pbegin = prof_now<true>();
#define K4_PT(i) do { const unsigned long long t_ = prof_now<true>(); pt[i] += t_ - plast; plast = t_; } while (0)
#define K4_PN(i, v) (pn[i] += (v))
#define K4_TIC() const unsigned long long tic_ = (unsigned long long)__builtin_readcyclecounter()
#define K4_TOC(i) (pt[i] += (unsigned long long)__builtin_readcyclecounter() - tic_)
#else
#define K4_TIC() ((void)0)
#define K4_TOC(i) ((void)0)
#define K4_PT(i) ((void)0)
#define K4_PN(i, v) ((void)0)
#endif
/* fast_round's two spare counters: the chain's scalar loop and its lazy resolutions, or (K4_PARSE_PROF_SPLIT) the part of the
 * phase behind the chain up to the next round's loads, and the records */
#if defined(K4_PARSE_PROF) && !defined(K4_PARSE_PROF_SPLIT)
#define K4_TICA() K4_TIC()
#define K4_TOCA(i) K4_TOC(i)
#else
#define K4_TICA() ((void)0)
#define K4_TOCA(i) ((void)0)
#endif
#if defined(K4_PARSE_PROF) && defined(K4_PARSE_PROF_SPLIT)
#define K4_TICB() const unsigned long long ticb_ = (unsigned long long)__builtin_readcyclecounter()
#define K4_TOCB(i) (pt[i] += (unsigned long long)__builtin_readcyclecounter() - ticb_)
#define K4_TICC() const unsigned long long ticc_ = (unsigned long long)__builtin_readcyclecounter()
#define K4_TOCC(i) (pt[i] += (unsigned long long)__builtin_readcyclecounter() - ticc_)
#else
#define K4_TICB() ((void)0)
#define K4_TOCB(i) ((void)0)
#define K4_TICC() ((void)0)
#define K4_TOCC(i) ((void)0)
#endif
    const uint32_t mfl1 = U - (uint32_t)MFLIMIT + 1u;          /* mflimitPlusOne */
    const uint32_t matchlimit = U - (uint32_t)LASTLITERALS;
    const unsigned long long me = 1ull << lane, below_me = me - 1ull;

    const bool resumed = ctl && ctl->resume;
    if (!resumed) for (int k = lane; k < 1024; k += 64) ((uint4 *)tabv)[k] = make_uint4(0u, 0u, 0u, 0u);     /* LZ4_initStream; put(hash(0), 0) stores a 0 (:119-122) */
    for (int k = lane; k < PARSE_SEEN_DWORDS; k += 64) seen[k] = 0u;
    wave_sync();

    /* segments: the next match end at or behind cut_pos ends its round and is looked at (handle_cut); a warm run counts no records */
    uint32_t cut_pos = 0xffffffffu;
    bool dry = false, cut_hit = false;
    uint32_t seg_begin = 0u;
    if (SEG) {
        const uint32_t ef = uni(sr->emit_from), sa = uni(sr->stop_at);
        seg_begin = uni(sr->begin);
        dry = ef != 0u;
        cut_pos = ef ? ef : sa;
    }
    if (SEG && !resumed && seg_begin != 0u) {                 /* the run's first position, inserted unsearched (:119-122) */
        if (lane == 0) tab.put(Table::hash(src + seg_begin), seg_begin);
        wave_sync();
    }
    uint32_t c = seg_begin + 1u;              /* position of lane 0 of the round */
    uint32_t sbase = seg_begin + 1u;          /* first probe of the running search (:466: the position behind a match + 1) */
    bool test = false;            /* c is the position right behind a match (:393-463) */
    uint32_t sj = 0u;             /* != 0: the search has used up its 66 contiguous probes; next probe is number sj (0-based) */
    uint32_t nrec = 0u;

    /* per lane and sub-window */
    uint32_t seq[K], n0[K], n1[K], n2[K];     /* the 16 source bytes at the position */
    uint32_t h[K], cpos[K], hop[K], ecode[K], pos[K];
    bool val[K];
    /* wave-uniform per sub-window */
    unsigned long long hmx[K], hits[K], Dm[K], Gall[K], vis[K];
    uint32_t qent[K];
    bool tent[K];
    /* the next round's source bytes */
    U128u pw[K];
    uint32_t pre2 = 0u, pre2b = 0u;
    /* Blocks whose matches often run past the 12 bytes a round knows behind every probe (:326-329 then costs a trip to memory in the
     * middle of the chain, ~1500 cycles under load) switch to rounds that know 28: 16 more bytes of every probe and candidate, about
     * 25 more instructions per round.  Decided every 32 rounds from what the rounds before met. */
#ifdef K4_PARSE_FORCE_MORE
    bool more = true;                 /* (test builds: every round of the 28-byte form from the first on) */
#else
    bool more = false;
#endif
    U128u pw2;                      /* ... bytes 16 .. 31 at the probe positions of the next round */
    uint32_t ahead = 0u;
    uint32_t long_seen = 0u, long_rounds = 0u;

    auto prepare = [&]() {
        if (sj == 0u) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint32_t p = c + 64u * (uint32_t)k + (uint32_t)lane;
                const uint32_t a = p < U - 16u ? p : U - 16u;
                pw[k] = ld128u(src + a);
            }
            if (K == 1 && more) {
                const uint32_t p2 = c + 16u + (uint32_t)lane;
                pw2 = ld128u(src + (p2 < U - 16u ? p2 : U - 16u));
            }
            /* (the word at c - 2 is the same in every lane, and a compiler that knows it moves it to a scalar register on the spot --
             * `s_waitcnt vmcnt(0)` right here, in front of everything the loads above were issued early to overlap with; an address
             * it cannot see through keeps the word a vector register until the next round's front looks at it) */
            uint32_t o2 = c >= 2u ? c - 2u : 0u;
#if !defined(K4_HOST_EMU) && !defined(K4_EXP_PRE2_SCALAR)
            asm("" : "+v"(o2));
#endif
            pre2 = ld32u(src + o2);
            if (TT == 0) pre2b = ld32u(src + o2 + 4u);
#ifdef K4_PARSE_AHEAD
            /* the lines two rounds on: asked for now (after the loads this round's successor waits for, so that its wait does not
             * include them), looked at never -- the asm below only keeps the compiler from dropping the load */
            {
                asm volatile("" :: "v"(ahead));
                const uint32_t pa = c + (uint32_t)K4_PARSE_AHEAD + 4u * (uint32_t)lane;
                ahead = ld32u(src + (pa < U - 4u ? pa : U - 4u));
            }
#endif
        } else {
            const uint32_t p = sbase + probe_offset(sj + (uint32_t)lane, 1u);
            const uint32_t a = p < U - 16u ? p : U - 16u;
            pw[0] = ld128u(src + a);
        }
    };
    if (resumed) {
        c = uni(ctl->c); sbase = uni(ctl->sbase); sj = uni(ctl->sj); nrec = uni(ctl->nrec); test = ctl->test; more = ctl->more;
        long_seen = uni(ctl->long_seen); long_rounds = uni(ctl->long_rounds);
        if (SEG) { cut_pos = uni(ctl->cut_pos); dry = uni(ctl->dry ? 1u : 0u) != 0u; }      /* (uni(): a loop that leaves on a condition the compiler takes for per-lane makes
                                                                                             * everything it carries per-lane -- and the scalar chain's operands are carried) */
    }
    prepare();
    (void)ahead;
    uint32_t mig_tick = 0u;

    /* lanes of sub-window k whose position was put into the table, from the hits `hk` of its chain: not the lanes before its entry
     * cursor (except the one two before it when the cursor came from a match, :394), not the lanes inside matches (except the lane
     * two before a match's end), none from lane `upto` on */
    auto visited = [&](auto kc, unsigned long long hk, uint32_t upto) -> unsigned long long {
        constexpr int k = decltype(kc)::value;
        const unsigned long long hb = hk & below_me;
        const int ph = hb ? 63 - (int)__clzll((long long)hb) : 0;
        const uint32_t qp = (uint32_t)__shfl((int)hop[k], ph) & 127u;
        const bool in = hb != 0ull && (uint32_t)lane < qp && (uint32_t)lane + 2u != qp;
        const bool before = (uint32_t)lane < qent[k] && !(tent[k] && (uint32_t)lane + 2u == qent[k]);
        const unsigned long long m = ballot(!in && !before && val[k]);
        return upto < 64u ? m & ((1ull << upto) - 1ull) : m;
    };

    /* One round.  PLAIN: every lane of every sub-window is a probe, its 16 bytes and its candidate's lie inside the block, no match
     * measured from them reaches matchlimit or the last probe position -- no clamping, no lane masks anywhere; the other form does
     * the block's last rounds and the strided ones.  Returns false when the block is done. */
    auto round = [&](auto plain_c) -> bool {
        constexpr bool plain = decltype(plain_c)::value;
        const bool strided = plain ? false : sj != 0u;
        const uint32_t c0 = c;
        const int KK = strided ? 1 : K;
#ifdef K4_PARSE_PROF
        plast = prof_now<true>();
        K4_PN(0, 1); if (!plain) K4_PN(1, 1); if (strided) K4_PN(2, 1);
#endif
        K4_ST(rounds, 1u); K4_ST(slow, 1u);

        /* ---------------- front: positions, hashes, candidates, what each lane would do as a hit ---------------- */
        K4_PHASE("front");
        uint32_t cand[K];
        bool flagged[K];
        uint32_t hE2 = 0xffffffffu;
        if (test) hE2 = Table::hash_of(pre2, pre2b);         /* the put of c - 2 (:394): made with the round's other puts, seen by its look-ups */
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (k >= KK) { val[k] = false; h[k] = 0u; pos[k] = 0u; hop[k] = 0u; cpos[k] = 0u; ecode[k] = 0u; seq[k] = n0[k] = n1[k] = n2[k] = 0u; cand[k] = 0u; flagged[k] = false; continue; }
            uint32_t p;
            if (strided) {
                p = sbase + probe_offset(sj + (uint32_t)lane, 1u);
                const uint32_t pn = sbase + probe_offset(sj + (uint32_t)lane + 1u, 1u);
                val[k] = pn <= mfl1 && pn > p;
            } else {
                p = c0 + 64u * (uint32_t)k + (uint32_t)lane;
                val[k] = plain || p + 1u <= mfl1;         /* (:172 for a step of one; the search's probe 65 is the chain's business) */
            }
            pos[k] = p;
            uint32_t w0 = pw[k].v[0], w1 = pw[k].v[1], w2 = pw[k].v[2], w3 = pw[k].v[3];
            if (!plain) shift16(w0, w1, w2, w3, p < U - 16u ? 0u : p - (U - 16u));
            seq[k] = w0; n0[k] = w1; n1[k] = w2; n2[k] = w3;
            h[k] = Table::hash_of(w0, w1);
            uint32_t cd = 0u;
            bool fl = false;
            if (val[k]) {
                cd = tab.get(h[k]);
                const uint32_t bit = h[k] >> SEEN_SHIFT;
                fl = ((atomicOr(&seen[bit >> 5], 1u << (bit & 31u)) >> (bit & 31u)) & 1u) != 0u;
            }
            if (h[k] == hE2) cd = c0 - 2u;
            cand[k] = cd;
            flagged[k] = fl;
        }
        /* the candidates' bytes: the round's one dependent trip to memory */
        K4_PT(0);
        U128u cw[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (k >= KK) { cw[k].v[0] = cw[k].v[1] = cw[k].v[2] = cw[k].v[3] = 0u; continue; }
            const uint32_t a = plain ? cand[k] : (cand[k] < U - 16u ? cand[k] : U - 16u);
            cw[k] = ld128u(src + a);
        }
        /* (every lane has recorded its hash by now: wipe the words this round touched) */
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < K; k++) if (k < KK && val[k]) seen[h[k] >> (5u + SEEN_SHIFT)] = 0u;

        /* lanes that share a hash: Dm = those with an earlier lane of the round in their group, Gall = all of them */
        K4_PT(1);
        K4_PHASE("groups");
#pragma unroll
        for (int k = 0; k < K; k++) { Dm[k] = 0ull; Gall[k] = 0ull; }
        {
            unsigned long long fl[K];
            unsigned long long any = 0ull;
#pragma unroll
            for (int k = 0; k < K; k++) { fl[k] = ballot(flagged[k]); any |= fl[k]; }
            if (any) {
#pragma unroll
                for (int k = 0; k < K; k++) {
                    while (fl[k]) {
                        K4_PN(3, 1); K4_ST(groups, 1u);
                        const int j = ctz64(fl[k]);
                        const uint32_t hj = readlane_u32(h[k], j);
                        unsigned long long m[K];
                        uint32_t members = 0u;
#pragma unroll
                        for (int kk = 0; kk < K; kk++) { m[kk] = ballot(val[kk] && h[kk] == hj); members += (uint32_t)__popcll(m[kk]); fl[kk] &= ~m[kk]; }
                        fl[k] &= ~(1ull << j);
                        if (members >= 2u) {
                            bool first = true;
#pragma unroll
                            for (int kk = 0; kk < K; kk++) {
                                Gall[kk] |= m[kk];
                                if (m[kk]) { Dm[kk] |= first ? (m[kk] & (m[kk] - 1ull)) : m[kk]; first = false; }
                            }
                        }
                    }
                }
            }
        }

        /* now the candidate bytes: hit or not, how far the match goes, the hop word */
        K4_PHASE("words");
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (k >= KK) { hmx[k] = 0ull; hits[k] = 0ull; vis[k] = 0ull; qent[k] = 64u; tent[k] = false; continue; }
            uint32_t v0 = cw[k].v[0], v1 = cw[k].v[1], v2 = cw[k].v[2], v3 = cw[k].v[3];
            if (!plain) shift16(v0, v1, v2, v3, cand[k] < U - 16u ? 0u : cand[k] - (U - 16u));
            const bool hit = val[k] && v0 == seq[k] && (!U32 || pos[k] - cand[k] <= (uint32_t)DISTANCE_MAX);      /* LL64.fast.cs:219-224 */
            uint32_t e = ext12(v1 ^ n0[k], v2 ^ n1[k], v3 ^ n2[k]);
            uint32_t word;
            if (plain) {
                word = ((uint32_t)lane + (uint32_t)MINMATCH + e) | (e == 12u ? HOP_LONG : 0u);
                if (SEG && pos[k] + (uint32_t)MINMATCH + e >= cut_pos) word |= HOP_END;
            } else {
                const uint32_t fwd_max = matchlimit - (pos[k] + (uint32_t)MINMATCH);       /* (of no consequence where the lane is no probe) */
                const bool lng = e == 12u && fwd_max > 12u && fwd_max < 0x80000000u;
                if (e > fwd_max) e = fwd_max;
                const uint32_t qn = strided ? 127u : (uint32_t)lane + (uint32_t)MINMATCH + e;
                word = qn | (lng ? HOP_LONG : 0u) | (pos[k] + (uint32_t)MINMATCH + e >= (SEG && cut_pos < mfl1 ? cut_pos : mfl1) ? HOP_END : 0u) | (val[k] ? 0u : HOP_INVALID);
            }
            const bool lazy = ((Dm[k] >> lane) & 1ull) != 0ull;
            if (lazy) word |= HOP_LAZY | (hit ? 0u : HOP_TABMISS);
            hop[k] = word;
            ecode[k] = e;
            cpos[k] = cand[k];
            hmx[k] = ballot(hit) | Dm[k] | (plain ? 0ull : ballot(!val[k]));
            hits[k] = 0ull;
            vis[k] = 0ull;
            qent[k] = 64u;
            tent[k] = false;
        }

        /* ---------------- the chain ---------------- */
        K4_PT(2);
        K4_PHASE("chain");
        /* a hop word for lane f of sub-window k worked out in scalar code: the lane's candidate is lane j of sub-window kk */
        auto word_from_lane = [&](auto kc, int f, auto kkc, int j) -> uint32_t {
            constexpr int k = decltype(kc)::value, kk = decltype(kkc)::value;
            if (readlane_u32(seq[k], f) != readlane_u32(seq[kk], j)) return 0xffffffffu;
            const uint32_t p = readlane_u32(pos[k], f), cp = readlane_u32(pos[kk], j);
            uint32_t e = ext12(readlane_u32(n0[k], f) ^ readlane_u32(n0[kk], j), readlane_u32(n1[k], f) ^ readlane_u32(n1[kk], j),
                               readlane_u32(n2[k], f) ^ readlane_u32(n2[kk], j));
            const uint32_t fwd_max = matchlimit - (p + (uint32_t)MINMATCH);
            const bool lng = e == 12u && fwd_max > 12u;
            if (e > fwd_max) e = fwd_max;
            const uint32_t qn = strided ? 127u : (uint32_t)f + (uint32_t)MINMATCH + e;
            const uint32_t word = qn | (lng ? HOP_LONG : 0u) | (p + (uint32_t)MINMATCH + e >= mfl1 ? HOP_END : 0u);
            if (lane == f) { hop[k] = word; ecode[k] = e; cpos[k] = cp; }
            return word;
        };
        /* lane f of sub-window k shares its hash with earlier lanes of the round: its candidate is the latest of them that has
         * been visited, else what the table held.  Returns its hop word, 0xffffffff when it is no hit. */
        auto resolve = [&](auto kc, int f) -> uint32_t {
            constexpr int k = decltype(kc)::value;
            const uint32_t hf = readlane_u32(h[k], f);
            {
                const unsigned long long m = ballot(h[k] == hf) & ((1ull << f) - 1ull) & visited(kc, hits[k] & ~(1ull << f), 64u);
                if (m) return word_from_lane(kc, f, kc, 63 - (int)__clzll((long long)m));
            }
            if constexpr (k >= 1) {
                const unsigned long long m = ballot(val[k - 1] && h[k - 1] == hf) & visited(ic<k - 1>{}, hits[k - 1], 64u);
                if (m) return word_from_lane(kc, f, ic<k - 1>{}, 63 - (int)__clzll((long long)m));
            }
            if constexpr (k >= 2) {
                const unsigned long long m = ballot(val[k - 2] && h[k - 2] == hf) & visited(ic<k - 2>{}, hits[k - 2], 64u);
                if (m) return word_from_lane(kc, f, ic<k - 2>{}, 63 - (int)__clzll((long long)m));
            }
            if constexpr (k >= 3) {
                const unsigned long long m = ballot(val[k - 3] && h[k - 3] == hf) & visited(ic<k - 3>{}, hits[k - 3], 64u);
                if (m) return word_from_lane(kc, f, ic<k - 3>{}, 63 - (int)__clzll((long long)m));
            }
            const uint32_t w = readlane_u32(hop[k], f);
            if (w & HOP_TABMISS) return 0xffffffffu;
            const uint32_t word = w & ~HOP_LAZY;
            if (lane == f) hop[k] = word;
            return word;
        };

        int outcome = 0;               /* 0 every sub-window walked, the search goes on; 1 the next round starts behind a match, at `anchor`;
                                        * 2 the block's last literals begin; 3 the search's contiguous probes are used up */
        uint32_t anchor = 0u;
        uint32_t upto_last = 64u;      /* the sub-window the round ended in: lanes from here on were not visited */
        int klast = KK - 1;            /* the sub-window the round ended in */
        {
            uint32_t q = 0u;
            bool etest = test;
            bool done = false;
            auto walk = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (done || k >= KK) return;
                const uint32_t w0 = c0 + 64u * (uint32_t)k;
                qent[k] = q;
                tent[k] = etest;
                klast = k;
                bool limited = false;
                unsigned long long limmask = ~0ull;
                uint32_t limend = 64u;         /* the first lane the running search does not probe one by one */
                if (!strided) {
                    /* the search's 66th probe, number 65 (:156-172: the step grows behind it) */
                    const int lim = (int)(sbase + 65u) - (int)w0;
                    if (lim < (int)q) { outcome = 3; upto_last = q; done = true; return; }
                    /* ... which is itself left with a step of two already (:170-171 work the step out an iteration ahead of its use), so
                     * at the very end of a block it is not made where a probe with a step of one still would be (:172) */
                    const bool cut65 = lim <= 63 && w0 + (uint32_t)lim + 2u > mfl1;
                    if (lim < 63 || cut65) {
                        limited = true;
                        limend = cut65 ? (uint32_t)lim : (uint32_t)lim + 1u;
                        limmask = limend >= 64u ? ~0ull : (1ull << limend) - 1ull;
                    }
                }
                for (;;) {
                    uint32_t hv = 0u;
                    int f = 0;
                    unsigned long long stop;
                    unsigned long long hm = hmx[k];
                    if (limited) {
                        /* only stops up to the limit count, and only until the first hop (a new search begins behind it): one at a time */
                        const unsigned long long t = hm & limmask & (~0ull << q);
                        if (!t) { outcome = 3; upto_last = limend; done = true; return; }
                        hm = t & (0ull - t);
                    }
                    { K4_TIC(); hop_chain(hm, hop[k], q, hits[k], f, hv, stop); K4_TOC(6); K4_PN(7, 1); }
                    if (!stop) {
                        if (limited) { limited = false; continue; }        /* that one stop was a plain hop */
                        if (hits[k]) {             /* where the running search began: behind the last match */
                            const int lf = 63 - (int)__clzll((long long)hits[k]);
                            sbase = w0 + (readlane_u32(hop[k], lf) & 127u) + 1u;
                        }
                        q = 0u;
                        etest = false;
                        return;
                    }
                    if (hv & HOP_INVALID) { hits[k] &= ~(1ull << f); outcome = 2; upto_last = (uint32_t)f; done = true; return; }
                    if (hv & HOP_LAZY) {
                        K4_PN(4, 1); K4_ST(lazies, 1u);
                        { K4_TIC(); hv = resolve(kc, f); K4_TOC(7); }
                        if (hv == 0xffffffffu) {
                            hits[k] &= ~(1ull << f);
                            hmx[k] &= ~(1ull << f);
                            q = (uint32_t)f + 1u;
                            if (q >= 64u) {        /* (the lane was the sub-window's last) */
                                if (limited) { outcome = 3; upto_last = limend; done = true; return; }
                                if (hits[k]) { const int lf = 63 - (int)__clzll((long long)hits[k]); sbase = w0 + (readlane_u32(hop[k], lf) & 127u) + 1u; }
                                q = 0u; etest = false; return;
                            }
                            continue;
                        }
                    }
                    const uint32_t p = strided ? readlane_u32(pos[k], f) : w0 + (uint32_t)f;
                    uint32_t e_end;
                    if (hv & HOP_LONG) {                               /* :326-329 beyond the 12 known bytes */
                        K4_PN(5, 1); K4_ST(longs, 1u);
                        const uint32_t match = readlane_u32(cpos[k], f);
                        const uint32_t code = 12u + wave_count(src + p + 16u, src + match + 16u, matchlimit - (p + 16u), lane);
                        e_end = p + (uint32_t)MINMATCH + code;
                        const uint32_t qf = (uint32_t)f + (uint32_t)MINMATCH + code;
                        if (lane == f) { ecode[k] = code; hop[k] = (!strided && qf < 127u) ? qf : 127u; }
                    } else {
                        e_end = strided ? p + (uint32_t)MINMATCH + readlane_u32(ecode[k], f) : w0 + (hv & 127u);
                    }
                    anchor = e_end;
                    sbase = e_end + 1u;
                    limited = false;
                    if (e_end >= mfl1) { outcome = 2; upto_last = (uint32_t)f + 1u; done = true; return; }      /* :391 */
                    if (SEG && e_end >= cut_pos) { cut_hit = true; outcome = 1; upto_last = (uint32_t)f + 1u; done = true; return; }
                    const uint32_t nq = e_end - w0;
                    if (strided || nq >= 128u || (nq >= 64u && k + 1 >= KK)) { outcome = 1; upto_last = (uint32_t)f + 1u; done = true; return; }
                    if (nq >= 64u) { q = nq - 64u; etest = true; return; }
                    q = nq;
                }
            };
            walk(ic<0>{});
            if constexpr (K >= 2) walk(ic<1>{});
            if constexpr (K >= 3) walk(ic<2>{});
            if constexpr (K >= 4) walk(ic<3>{});
        }

        /* ---------------- where the next round starts; its source loads go out now ---------------- */
        K4_PT(3);
        K4_PN(6, (unsigned long long)klast + 1ull);
        K4_PHASE("next");
        if (outcome == 1) { c = anchor; test = true; sj = 0u; }
        else if (outcome == 0) {
            if (strided) sj += 64u;
            else {
                c = c0 + 64u * (uint32_t)K;
                test = false;
                if (c - sbase >= 66u) sj = 66u;
            }
        }
        else if (outcome == 3) { test = false; sj = 66u; }
        if (outcome != 2) prepare();

        /* ---------------- the visited lanes ---------------- */
        K4_PHASE("visited");
        {
            auto fin = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (k >= KK) return;
                if (k > klast) { vis[k] = 0ull; return; }
                uint32_t upto = k == klast ? upto_last : 64u;
                /* a match that ends the round: the lanes behind its first are not visited -- except, where the round goes on behind the
                 * match, the one two before its end if that is still a lane here (:394; the next round puts it once more) */
                unsigned long long m = visited(kc, hits[k], 64u);
                if (upto < 64u) {
                    unsigned long long keep = (1ull << upto) - 1ull;
                    if (outcome == 1 && k == klast && !strided) {
                        const uint32_t e2 = anchor - 2u - (c0 + 64u * (uint32_t)k);
                        if (e2 < 64u && e2 >= upto) keep |= 1ull << e2;
                    }
                    m &= keep;
                }
                vis[k] = m;
            };
            fin(ic<0>{});
            if constexpr (K >= 2) fin(ic<1>{});
            if constexpr (K >= 3) fin(ic<2>{});
            if constexpr (K >= 4) fin(ic<3>{});
        }

        /* ---------------- records ---------------- */
        K4_PHASE("records");
        {
            uint32_t at = nrec;
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (k >= KK) continue;
                if (hits[k] && !(SEG && dry)) {
                    if (!DRY && ((hits[k] >> lane) & 1ull))
                        rec_store(recs + at + (uint32_t)__popcll(hits[k] & below_me), pos[k], (pos[k] - cpos[k]) | ((U32 && ecode[k] > 0xffffu ? 0xffffu : ecode[k]) << 16));
                    at += (uint32_t)__popcll(hits[k]);
                }
            }
            nrec = at;
        }
        if (outcome == 2) return false;

        /* ---------------- the round's puts ---------------- */
        K4_PT(4);
        K4_PHASE("commit");
#ifdef K4_PARSE_DEBUG
        if (lane == 0 && getenv("K4DBG") && c0 + 64u * K > (uint32_t)atoi(getenv("K4DBG")) && c0 < (uint32_t)atoi(getenv("K4DBG")) + 200u) {
            printf("round c0=%u test=%d strided=%d plain=%d outcome=%d anchor=%u klast=%d upto=%u sbase=%u hE2=%x\n", c0, (int)(hE2 != 0xffffffffu), (int)strided, (int)plain, outcome, anchor, klast, upto_last, sbase, hE2);
            for (int k = 0; k < K; k++) printf("   k=%d qent=%u tent=%d hmx=%016llx hits=%016llx vis=%016llx Dm=%016llx Gall=%016llx\n", k, qent[k], (int)tent[k], hmx[k], hits[k], vis[k], Dm[k], Gall[k]);
        }
#endif
        if (hE2 != 0xffffffffu && lane == 0) tab.put(hE2, c0 - 2u);
        if (GT) wave_sync(); else __builtin_amdgcn_wave_barrier();      /* a lane of the round may put the same slot: it comes second */
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (k >= KK) continue;
            if ((vis[k] & ~Gall[k]) >> lane & 1ull) tab.put(h[k], pos[k]);
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (k >= KK) continue;
            unsigned long long g = vis[k] & Gall[k];
            while (g) {
                const int j = ctz64(g);
                g &= g - 1ull;
                if (GT) wave_sync(); else __builtin_amdgcn_wave_barrier();
                if (lane == j) tab.put(h[k], pos[k]);
            }
        }
        if (GT) wave_sync(); else lds_sync();       /* (never a wait for the records' stores or the next round's loads) */
        K4_PT(5);
        return true;
    };

    /* The same round once more for the case that is nearly all of a block's rounds -- one sub-window, every lane a probe with its
     * bytes inside the block, the search's 66-probe limit out of reach -- written for the number of instructions it takes: a wave
     * alone issues one instruction every five to six cycles whatever the instruction is, so a round costs what it counts.
     * Differences from the general form: the lanes inside matches come from a DPP max-scan instead of a shuffle, the round's puts are
     * one store with ONE writer per slot (of the visited lanes of an equal-hash group the highest: the latest position is what a slot
     * holds in the end; every lane carries its group's lane mask G), a lane that stops the chain for its group's sake has its
     * candidate's bytes compared in vector code against one broadcast lane. */
    auto fast_round = [&](auto more_c) -> bool {
        constexpr bool MORE = decltype(more_c)::value;
        constexpr uint32_t KNOWN = MORE ? 28u : 12u;
        const uint32_t c0 = c;
#ifdef K4_PARSE_PROF
        plast = prof_now<true>();
        K4_PN(0, 1);
#endif
        K4_ST(rounds, 1u);
        K4_PHASE("front");
        const uint32_t p = c0 + (uint32_t)lane;
        const uint32_t w0 = pw[0].v[0], w1 = pw[0].v[1], w2 = pw[0].v[2], w3 = pw[0].v[3];
        const uint32_t hh = Table::hash_of(w0, w1);
#if defined(K4_NT_GTAB)
        uint32_t cd;
        if constexpr (GT && TT == 1) cd = tab.get_nt(hh); else cd = tab.get(hh);
#else
        uint32_t cd = tab.get(hh);
#endif
        const uint32_t bit = hh >> SEEN_SHIFT;
        const bool flg = ((atomicOr(&seen[bit >> 5], 1u << (bit & 31u)) >> (bit & 31u)) & 1u) != 0u;
        uint32_t hE2 = 0xffffffffu;
        if (test) {
            hE2 = Table::hash_of(pre2, pre2b);
            if (hh == hE2) cd = c0 - 2u;
        }
        K4_PT(0);
#ifdef K4_NT_CAND
        const U128u cw = ld128u_nt(src + cd);
        U128u cw2 = {{0u, 0u, 0u, 0u}};
        if (MORE) cw2 = ld128u_nt(src + cd + 16u);
#else
        const U128u cw = ld128u(src + cd);
        U128u cw2 = {{0u, 0u, 0u, 0u}};
        if (MORE) cw2 = ld128u(src + cd + 16u);
#endif
        __builtin_amdgcn_wave_barrier();
        seen[hh >> (5u + SEEN_SHIFT)] = 0u;
        /* groups */
        K4_PT(1);
        K4_PHASE("groups");
        unsigned long long D0 = 0ull;
        unsigned long long G = me;            /* per lane: the lanes of the window with its hash */
        {
            unsigned long long fl = ballot(flg);
            while (fl) {
                K4_PN(3, 1); K4_ST(groups, 1u);
                const uint32_t hj = readlane_u32(hh, ctz64(fl));
                const bool same = hh == hj;
                const unsigned long long m = ballot(same);
                fl &= ~m;
                if (same) G = m;
                D0 |= m & (m - 1ull);
            }
        }
        K4_PHASE("words");
        const bool hit = cw.v[0] == w0 && (!U32 || p - cd <= (uint32_t)DISTANCE_MAX);      /* LL64.fast.cs:219-224 */
        uint32_t e;
        if (MORE) {
            const uint32_t x[7] = {cw.v[1] ^ w1, cw.v[2] ^ w2, cw.v[3] ^ w3, cw2.v[0] ^ pw2.v[0], cw2.v[1] ^ pw2.v[1], cw2.v[2] ^ pw2.v[2], cw2.v[3] ^ pw2.v[3]};
            e = ext28(x);
        } else e = ext12(cw.v[1] ^ w1, cw.v[2] ^ w2, cw.v[3] ^ w3);
        uint32_t word = ((uint32_t)lane + (uint32_t)MINMATCH + e) | (e == KNOWN ? HOP_LONG : 0u);
        if (SEG && p + (uint32_t)MINMATCH + e >= cut_pos) word |= HOP_END;
        if ((D0 >> lane) & 1ull) word |= HOP_LAZY | (hit ? 0u : HOP_TABMISS);
        uint32_t ec = e, cp = cd;
        unsigned long long hm = ballot(hit) | D0;
        K4_PT(2);
        K4_PHASE("chain");
        unsigned long long hts = 0ull;
        uint32_t q = 0u, anchor = 0u, upto = 64u;
        int outcome = 0;
        for (;;) {
            uint32_t hv = 0u;
            int f = 0;
            unsigned long long stop;
            { K4_TICA(); hop_chain(hm, word, q, hts, f, hv, stop); K4_TOCA(6); K4_PN(7, 1); }
            if (!stop) break;
            if (hv & HOP_LAZY) {
                K4_PN(4, 1); K4_ST(lazies, 1u);
                K4_TICA();
                /* the latest VISITED lane below f with f's hash, if any: a lane is inside a match -- never put -- when it lies below the
                 * landing place of the nearest hit below it, other than two before it (:394) */
                unsigned long long cm = ballot(hh == readlane_u32(hh, f)) & ((1ull << f) - 1ull);
                const unsigned long long hb = hts & ((1ull << f) - 1ull);
                unsigned long long m = 0ull;
                while (cm) {
                    const int j = 63 - (int)__clzll((long long)cm);
                    cm &= ~(1ull << j);
                    const unsigned long long hbj = hb & ((1ull << j) - 1ull);
                    bool v = true;
                    if (hbj) { const uint32_t qj = readlane_u32(word, 63 - (int)__clzll((long long)hbj)) & 127u; v = (uint32_t)j >= qj || (uint32_t)j + 2u == qj; }
                    if (v) { m = 1ull << j; break; }
                }
                if (m) {
                    const int j = 63 - (int)__clzll((long long)m);
                    const uint32_t s0 = readlane_u32(w0, j), s1 = readlane_u32(w1, j), s2 = readlane_u32(w2, j), s3 = readlane_u32(w3, j);
                    uint32_t e2;
                    if (MORE) {
                        const uint32_t x[7] = {s1 ^ w1, s2 ^ w2, s3 ^ w3, readlane_u32(pw2.v[0], j) ^ pw2.v[0], readlane_u32(pw2.v[1], j) ^ pw2.v[1],
                                               readlane_u32(pw2.v[2], j) ^ pw2.v[2], readlane_u32(pw2.v[3], j) ^ pw2.v[3]};
                        e2 = ext28(x);
                    } else e2 = ext12(s1 ^ w1, s2 ^ w2, s3 ^ w3);
                    const uint32_t wj = s0 != w0 ? 0xffffffffu : (((uint32_t)lane + (uint32_t)MINMATCH + e2) | (e2 == KNOWN ? HOP_LONG : 0u));
                    hv = readlane_u32(wj, f);
                    if (lane == f) { word = wj; ec = e2; cp = c0 + (uint32_t)j; }
                } else {
                    hv = (hv & HOP_TABMISS) ? 0xffffffffu : (hv & ~HOP_LAZY);
                    if (lane == f) word = hv;
                }
                K4_TOCA(7);
                if (hv == 0xffffffffu) {
                    hts &= ~(1ull << f);
                    hm &= ~(1ull << f);
                    q = (uint32_t)f + 1u;
                    if (q >= 64u) break;
                    continue;
                }
            }
            uint32_t e_end = c0 + (hv & 127u);
            if (hv & HOP_LONG) {                               /* :326-329 beyond the 12 known bytes */
                K4_PN(5, 1); K4_ST(longs, 1u);
                const uint32_t pf = c0 + (uint32_t)f;
                const uint32_t match = readlane_u32(cp, f);
                const uint32_t code = KNOWN + wave_count(src + pf + 4u + KNOWN, src + match + 4u + KNOWN, matchlimit - (pf + 4u + KNOWN), lane);
                e_end = pf + (uint32_t)MINMATCH + code;
                const uint32_t qf = (uint32_t)f + (uint32_t)MINMATCH + code;
                if (lane == f) { ec = code; word = qf < 127u ? qf : 127u; }
                long_seen++;
                if (e_end >= mfl1) { anchor = e_end; outcome = 2; upto = (uint32_t)f + 1u; break; }      /* :391 */
            }
            if (SEG && e_end >= cut_pos) { cut_hit = true; anchor = e_end; outcome = 1; upto = (uint32_t)f + 1u; break; }
            const uint32_t nq = e_end - c0;
            if (nq >= 64u) { anchor = e_end; outcome = 1; upto = (uint32_t)f + 1u; break; }
            q = nq;
        }
        K4_PT(3);
        K4_PN(6, 1);
        K4_PHASE("next");
        K4_TICB();
        /* (uni(): what the scalar chain hands back counts as per-lane for the compiler, and these go into the next chain's operands) */
        if (outcome == 1) { c = uni(anchor); test = true; sbase = c + 1u; }
        else if (outcome == 0) {
            c = c0 + 64u;
            test = false;
            if (hts) sbase = uni(c0 + (readlane_u32(word, 63 - (int)__clzll((long long)hts)) & 127u) + 1u);
            if (c - sbase >= 66u) sj = 66u;
        }
        c = uni(c); sbase = uni(sbase); sj = uni(sj); test = uni(test ? 1u : 0u) != 0u;
        const bool mine = ((hts >> lane) & 1ull) != 0ull;
        if (MORE) long_seen += (uint32_t)__popcll(ballot(mine && ec > 12u));      /* (what a round that knows 12 bytes would have had to count) */
#ifndef K4_PARSE_FORCE_MORE
        if (++long_rounds == 32u) { more = uni(long_seen) >= 12u; long_seen = 0u; long_rounds = 0u; }
#endif
        if (outcome != 2) prepare();
        K4_TOCB(6);
        K4_PHASE("records");
        K4_TICC();
        if (hts && !(SEG && dry)) {
            if (!DRY && mine) rec_store(recs + nrec + (uint32_t)__popcll(hts & below_me), p, (p - cp) | ((U32 && ec > 0xffffu ? 0xffffu : ec) << 16));
            nrec += (uint32_t)__popcll(hts);
        }
        K4_TOCC(7);
        if (outcome == 2) return false;
        K4_PHASE("visited");
        /* lanes inside matches: below the landing place of the nearest hit below them, except the lane two before it (:394) */
        unsigned long long vm;
        {
            uint32_t x = mine ? (word & 127u) : 0u;
            x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));
            x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));
            x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));
            x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
            x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
            x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
            const uint32_t below = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, true);      /* wave_shr:1 */
            vm = ballot(!((uint32_t)lane < below && (uint32_t)lane + 2u != below));
            if (upto < 64u) vm &= (1ull << upto) - 1ull;
        }
        K4_PT(4);
        K4_PHASE("commit");
        /* one writer per slot: of the visited lanes of a group the highest (the latest position is what a slot holds in the end); the
         * put of c0 - 2 (:394) came before all of them and stands only where none of them has its hash */
        if (hE2 != 0xffffffffu && (ballot(hh == hE2) & vm) == 0ull && lane == 0) tab.put(hE2, c0 - 2u);
#if defined(K4_NT_GTAB)
        if (((vm >> lane) & 1ull) && (G & vm & ~(below_me | me)) == 0ull) { if constexpr (GT && TT == 1) tab.put_nt(hh, p); else tab.put(hh, p); }
#else
        if (((vm >> lane) & 1ull) && (G & vm & ~(below_me | me)) == 0ull) tab.put(hh, p);
#endif
        if (GT) wave_sync(); else lds_sync();       /* (never a wait for the records' stores or the next round's loads) */
        K4_PT(5);
        return true;
    };
    /* a round has ended at a match end at or behind cut_pos (c is that match end, the round's puts are made): the table holds every
     * visited position before it -- and position c - 2, which the reference puts right behind a match (:394) and this encoder with the
     * round that follows: here, so that both runs hold it whichever way their windows fell (the round that follows puts it once more).
     * Returns true when the run is over (sr->state says how). */
    auto handle_cut = [&]() -> bool {
        const uint32_t cut = c;
        if (lane == 0) tab.put(Table::hash(src + cut - 2u), cut - 2u);
        wave_sync();
        if (dry) {
            /* the warm run has reached its cut: publish it with the table, count records from here on */
            uint32_t *pub = sr->snap_pub;
#pragma unroll 2
            for (int k = lane; k < 4096; k += 64) pub[16 + k] = tab.get((uint32_t)k);
            wave_sync();
            agent_publish(pub, cut + 1u);            /* (all lanes, the same word: see the end of this function) */
            dry = false;
            nrec = 0u;
            sr->cut = cut;
            cut_pos = uni(sr->stop_at);
            return false;
        }
        /* the true run has reached a match end behind the next segment's boundary: is that segment's run in step here? */
        uint32_t theirs = 0u;
        const uint32_t spin_max = uni(sr->spin_max) ? uni(sr->spin_max) : SEG_SPIN_MAX;
        for (uint32_t spin = 0; spin < spin_max; spin++) {
            theirs = uni(agent_peek(sr->snap_chk));
            if (theirs != 0u) break;
            __builtin_amdgcn_s_sleep(8);
        }
        bool same = theirs == cut + 1u;
        if (same) {
            agent_acquire();
            bool differ = false;
#pragma unroll 2
            for (int k = lane; k < 4096; k += 64) {
                const uint32_t mine = tab.get((uint32_t)k), other = sr->snap_chk[16 + k];
                differ = differ || (mine != other && !(cut - mine > (uint32_t)DISTANCE_MAX && cut - other > (uint32_t)DISTANCE_MAX));
            }
            same = ballot(differ) == 0ull;
        }
        sr->stop = cut;
        if (!same && (theirs == 0u || !sr->fix)) { sr->state = 3u; return true; }        /* gave up waiting: nothing to offer */
        if (!same) {               /* this run's records stand if IT began in step; its table is left for a run that goes on from here */
            uint32_t *fix = sr->fix;
#pragma unroll 2
            for (int k = lane; k < 4096; k += 64) fix[k] = tab.get((uint32_t)k);
            wave_sync();
        }
        sr->state = same ? 1u : 4u;
        return true;
    };
    if (ctl) ctl->stopped = false;
    const uint32_t rec_cap = ctl ? ctl->rec_cap : 0u;
    for (;;) {
        if (SEG) {
            /* (what the rounds and handle_cut leave in these is the same in every lane; uni() says so to the compiler -- a value it
             * takes for per-lane, carried around this loop, would make the scalar chains' operands per-lane with it) */
            cut_hit = uni(cut_hit ? 1u : 0u) != 0u;
            if (cut_hit) {
                cut_hit = false;
                if (handle_cut()) { ctl->stopped = true; return nrec; }
            }
            cut_pos = uni(cut_pos); nrec = uni(nrec); dry = uni(dry ? 1u : 0u) != 0u;
            c = uni(c); sbase = uni(sbase);
        }
        if (U32 && rec_cap && nrec + 64u * (uint32_t)K + 2u > rec_cap) {
            ctl->c = c; ctl->sbase = sbase; ctl->sj = sj; ctl->nrec = nrec; ctl->test = test; ctl->more = more;
            ctl->long_seen = long_seen; ctl->long_rounds = long_rounds;
            if (SEG) { ctl->cut_pos = cut_pos; ctl->dry = dry; }
            ctl->claimed = PARSE_FLUSH;
            return nrec;
        }
        if (GT && ctl && ctl->free_slots && (++mig_tick & 15u) == 0u) {
            const uint32_t fs = uni(*(volatile uint32_t *)ctl->free_slots);
            if (fs) {
                const uint32_t sl = (uint32_t)__ffs((int)fs) - 1u;
                uint32_t old = 0u;
                if (lane == 0) old = atomicAnd(ctl->free_slots, ~(1u << sl));
                if (uni(old) & (1u << sl)) {
                    ctl->c = c; ctl->sbase = sbase; ctl->sj = sj; ctl->nrec = nrec; ctl->test = test; ctl->more = more;
                    ctl->long_seen = long_seen; ctl->long_rounds = long_rounds;
                    if (SEG) { ctl->cut_pos = cut_pos; ctl->dry = dry; }
                    ctl->claimed = (int)sl;
                    wave_sync();             /* the table's last puts are in memory before anybody copies it */
                    return nrec;
                }
            }
        }
        const bool plain = sj == 0u && c + 64u * (uint32_t)K + 28u <= U;
        /* (the search's limit: probes up to sbase + 65 are contiguous, the window ends at c + 63) */
        if (K == 1 && plain && c + 63u <= sbase + 65u) {
            if (more && c + 108u <= U) { if (!fast_round(std::true_type{})) break; }
            else if (!fast_round(std::false_type{})) break;
        }
        else if (plain) { if (!round(std::true_type{})) break; }
        else if (!round(std::false_type{})) break;
    }
#ifdef K4_PARSE_PROF
    if (pc && lane == 0) {
        pc[0] = prof_now<true>() - pbegin;
        for (int i = 0; i < 6; i++) pc[1 + i] = pt[i];
        /* [8..10] are the placement stamps, [15] says where the table lived */
        pc[7] = (unsigned long long)nrec | (pn[4] << 32);              /* sequences | lazy stops << 32 */
        pc[11] = pn[0] | (pn[6] << 32);                                /* rounds | sub-windows entered << 32 */
        pc[12] = pn[1] | (pn[2] << 32);                                /* rounds of the careful form | strided ones << 32 */
        pc[13] = (pt[6] & 0xffffffffull) | (pt[7] << 32);              /* cycles inside the tight chain | inside lazy resolutions << 32 (both include ~2 counter reads each) */
        pc[10] = pn[7];                                                /* entries into the tight chain (overwrites the HW_ID stamp) */
        pc[14] = pn[3] | (pn[5] << 32);                                /* groups | long counts << 32 */
    }
#endif
    if (DRY && stats) *stats = st;
#undef K4_ST
    if (SEG) {
        if (uni(dry ? 1u : 0u) != 0u) {           /* a warm run that never found its cut: nothing to offer */
            /* (every lane stores the same word: a fence inside `if (lane == 0)` at this place makes the compiler move scalar values that
             * the chains' assembly produced into vector registers -- "illegal VGPR to SGPR copy") */
            agent_publish(sr->snap_pub, SEG_NONE);
            sr->state = 3u;
            ctl->stopped = true;
            nrec = 0u;
        } else {
            sr->stop = U; sr->state = 2u;
        }
    }
    return nrec;
}

/* ------------------------------------------------------------------------------------------------------------------ */

/* where a block's write-out stands between two calls (a block longer than its record slot is written out a slot-full at a time) */
struct EmitState { uint32_t op, emitted_to; };
template <bool HC = false, bool BIG = false>
__device__ __forceinline__ bool emit_records(const uint8_t *src, const uint32_t U, uint8_t *dst, const int dst_cap, const uint2 *recs, const uint32_t nseq, const int lane, EmitState &st);
__device__ __forceinline__ int emit_tail(const uint8_t *src, const uint32_t U, uint8_t *dst, const int dst_cap, const int lane, const EmitState &st);
template <bool HC = false, bool BIG = false>
__device__ __forceinline__ int emit_block(const uint8_t *src, const uint32_t U, uint8_t *dst, const int dst_cap, const uint2 *recs, const uint32_t nseq, const int lane)
{
    EmitState st = {0u, 0u};
    if (!emit_records<HC, BIG>(src, U, dst, dst_cap, recs, nseq, lane, st)) return 0;
    return emit_tail(src, U, dst, dst_cap, lane, st);
}

/* One block (or, SEG, one run over a big block: `sr`) by the calling wave: parse, the records written out by the same wave -- a slot-full
 * at a time where the block has more sequences than its record slot holds --, the table in LDS (`in_lds`: the wave's own slot) or in
 * memory until a table of the workgroup becomes free.  Returns what the encoder call returns (bytes written to dst, 0: no room); for a
 * run that stopped at its boundary (sr->state 1 / 4) that is the piece without last literals.  *nrec_out: records of the last parse call
 * (the whole block's where it fits its slot). */
template <int K, int TT, bool SEG>
__device__ __forceinline__ int parse_one(const BatchArgs &a, const ParseArgs &p, uint32_t *lds, uint32_t *seen, const bool in_lds, const uint32_t wave, const uint32_t waves,
                                         const int lane, const long long b, const uint8_t *src, const int src_len, uint8_t *dst, const int cap, SegRun *sr, uint32_t *nrec_out)
{
    constexpr bool U32 = TT != 1;
    uint2 *recs = p.recs + (p.slot_recs ? (unsigned long long)blockIdx.x * waves + wave : (unsigned long long)b) * PARSE_REC_STRIDE;
    uint32_t n;
    unsigned long long *pc = nullptr;
#ifdef K4_PARSE_PROF
    if (a.prof) { pc = a.prof + PROF_STRIDE * b; prof_place<true>(pc, 8, lane); }
#endif
    uint32_t *free_slots = p.migrate ? lds + PARSE_LDS_DWORDS - 1 : nullptr;
    bool moved = false;
#ifndef K4_PARSE_PROF
    if (a.prof && !SEG) prof_place<true>(a.prof + PROF_STRIDE * b, 8, lane);
#endif
    /* One call of parse_block is the whole block unless it comes back early: with a claim on an LDS table that has become free
     * (the table moves, the LDS form goes on), or -- byU32 blocks, which may hold more sequences than a record slot -- with a
     * slot-full of records to be written out before it goes on (PARSE_FLUSH). */
    EmitState est = {0u, 0u};
    bool room = true, started = false;
    auto write_out = [&](uint32_t cnt) {         /* the records so far; a warm run's output begins at its cut */
        if (SEG && !started) { est.emitted_to = uni(sr->cut); started = true; }
        wave_sync();                             /* the records are this wave's own stores: in order with the loads that follow */
        if (room && !emit_records<false, U32>(src, (uint32_t)src_len, dst, cap < 0 ? 0 : cap, recs, cnt, lane, est)) room = false;
        wave_sync();
    };
    uint32_t *gt = p.gtab + 4096ull * ((unsigned long long)blockIdx.x * PARSE_MAX_WAVES + wave);
    uint32_t *table = in_lds ? lds + 4096u * wave : gt;
    bool table_in_lds = in_lds;
    int my_slot = -1;                            /* an LDS table this wave moved into (to be given back) */
    ParseCtl ctl = {};
    ctl.free_slots = in_lds ? nullptr : free_slots; ctl.claimed = -1; ctl.resume = false;
    ctl.rec_cap = U32 ? PARSE_REC_STRIDE : 0u;
    for (;;) {
        if (table_in_lds) n = parse_block<K, false, false, TT, SEG>(src, (uint32_t)src_len, recs, table, seen, lane, pc, nullptr, (U32 || ctl.resume) ? &ctl : nullptr, sr);
        else n = parse_block<K, true, false, TT, SEG>(src, (uint32_t)src_len, recs, table, seen, lane, pc, nullptr, &ctl, sr);
        if (U32 && ctl.claimed == PARSE_FLUSH) {
            if (p.inline_emit) write_out(n);
            if (!room) { n = 0u; break; }            /* the output does not fit (:251-255, :346-350): the block fails, no need to go on */
            ctl.nrec = 0u; ctl.resume = true; ctl.claimed = -1;
            continue;
        }
        if (ctl.claimed >= 0) {                  /* a table of the workgroup has become free: move in */
            moved = true;
            uint32_t *slot = lds + 4096u * (uint32_t)ctl.claimed;
#pragma unroll 4
            for (int k = lane; k < 1024; k += 64) ((uint4 *)slot)[k] = ((const uint4 *)gt)[k];
            wave_sync();
            my_slot = ctl.claimed;
            table = slot; table_in_lds = true;
            ctl.resume = true; ctl.free_slots = nullptr; ctl.claimed = -1;
            continue;
        }
        break;
    }
    if (free_slots && lane == 0) {
        if (my_slot >= 0) atomicOr(free_slots, 1u << my_slot);                 /* ... and free again for the next one */
        else if (in_lds && !p.queue) atomicOr(free_slots, 1u << wave);          /* this wave's table is free now */
    }
#ifdef K4_PARSE_PROF
    if (a.prof) { if (lane == 0) { pc[9] = __builtin_amdgcn_s_memrealtime(); pc[15] = in_lds ? 1u : 2u; } }
#endif
    *nrec_out = n;
#ifndef K4_PARSE_PROF
    /* k4lz4_profile_batch_device mode 4 (stamps only): [12] when the parse was through, [9] when the block was written, [11]
     * where its table lived (4 LDS, 5 memory, 6 memory first and an LDS table from some round on) */
    if (a.prof && !SEG) { prof_place<true>(a.prof + PROF_STRIDE * b, 12, lane); if (lane == 0) a.prof[PROF_STRIDE * b + 11] = in_lds ? 4u : (moved ? 6u : 5u); }
#endif
    int ret = 0;
    if (p.inline_emit) {
        const bool stopped = SEG && ctl.stopped;                  /* the run ended at its boundary: a piece, no last literals */
        if (!(SEG && stopped && uni(sr->state) == 3u)) {
            if (room) write_out(n);
            if (room) ret = stopped ? (int)est.op : emit_tail(src, (uint32_t)src_len, dst, cap < 0 ? 0 : cap, lane, est);
        }
    }
#ifndef K4_PARSE_PROF
    if (a.prof && !SEG) prof_place<true>(a.prof + PROF_STRIDE * b, 9, lane);
#endif
    (void)moved;
    return ret;
}

template <int K, int TT = 1>
__device__ __forceinline__ void parse_kernel_body(const BatchArgs &a, const ParseArgs &p, uint32_t *lds)
{
    constexpr bool U32 = TT != 1;
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t waves = blockDim.x >> 6;
    const uint32_t lds_tables = waves < (uint32_t)PARSE_LDS_TABLES ? waves : (uint32_t)PARSE_LDS_TABLES;
    uint32_t *seen = lds + 4096u * (uint32_t)PARSE_LDS_TABLES + (uint32_t)PARSE_SEEN_DWORDS * wave;
    const bool in_lds = wave < lds_tables;
    if (U32 && p.nbig && uni(*(volatile uint32_t *)p.nbig) == 0u) return;          /* the launch before this one found nothing for it */
    if (p.migrate) {
        if (threadIdx.x == 0) lds[PARSE_LDS_DWORDS - 1] = 0u;
        __syncthreads();
    }
    for (long long idx = (long long)wave * (long long)p.nwg + (long long)blockIdx.x;;) {
        if (p.queue) {
            uint32_t t = 0u;
            if (lane == 0) {
                t = atomicAdd(p.queue, 1u);
                if (t < (uint32_t)a.n) t = in_lds ? atomicAdd(p.queue + 1, 1u) : (uint32_t)a.n - 1u - atomicAdd(p.queue + 2, 1u);
                else t = 0xffffffffu;
            }
            t = uni(t);
            if (t == 0xffffffffu) {
                /* no block left for this wave: its LDS table is for a block of the workgroup that still parses with its table in memory */
                if (p.migrate && in_lds && lane == 0) atomicOr(lds + PARSE_LDS_DWORDS - 1, 1u << wave);
                return;
            }
            idx = (long long)t;
        }
        if (idx >= a.n) return;
        const long long b = a.order ? (long long)uni(a.order[idx]) : idx;
        const int src_len = a.srcLen[b];
        uint32_t *meta = p.meta + 2ull * (unsigned long long)b;
        /* whose block: this launch's (byU16: 128 .. 65 546 bytes; the byU32 launch behind it: 65 547 and more), the byU32 launch's
         * (marked PARSE_BIG by the first one when there is going to be one), or the one-kernel encoder's (PARSE_REST) */
        const bool big = src_len >= LIMIT_64K && a.accel == 1;
        const bool mine = a.accel == 1 && (U32 ? big : (src_len >= (int)PARSE_MIN_LEN && src_len < LIMIT_64K));
        if (!mine) {
            if (!U32 && lane == 0) {
                meta[0] = (big && p.big) ? PARSE_BIG : PARSE_REST; meta[1] = 0u;
                if (big && p.big && p.nbig) atomicAdd(p.nbig, 1u);
            }
        } else {
            uint32_t n = 0u;
            const int ret = parse_one<K, TT, false>(a, p, lds, seen, in_lds, wave, waves, lane, b, a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], a.dstCap[b], nullptr, &n);
            if (lane == 0) {
                meta[0] = n; meta[1] = 0u;
                if (p.inline_emit) a.outLen[b] = codec_encode_result(src_len, ret, a.flags);
            }
        }
        if (!p.queue) return;
    }
}


__global__ __launch_bounds__(64 * PARSE_MAX_WAVES) void k4_parse_kernel(BatchArgs a, ParseArgs p)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[PARSE_LDS_DWORDS];
    parse_kernel_body<K4_PARSE_K>(a, p, lds);
}

/* The same for the blocks of 65 547 bytes and more (round 6): byU32 table, hash5 (LL64.fast.cs:526-544) or -- LZ4Codec.Enforce32 --
 * hash4 (x32/LL32.tools.cs:141-148); a kernel of its own so that k4_parse_kernel stays the code the bench batch was tuned on. */
__global__ __launch_bounds__(64 * PARSE_MAX_WAVES) void k4_parse_big_kernel(BatchArgs a, ParseArgs p)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[PARSE_LDS_DWORDS];
    if (a.flags & FLAG_X32) parse_kernel_body<1, 2>(a, p, lds);
    else parse_kernel_body<1, 0>(a, p, lds);
}

/* ------------------------------------------------------------------------------------------------------------------ */

/*
 * Which blocks get the tables in LDS.  All blocks of a launch start together and the launch lasts as long as its slowest block; a
 * block whose table lives in memory pays about two more trips to memory per round (the look-up, the wait for its puts), so the
 * seven of sixteen waves per workgroup that have no LDS table should get the blocks that are through soonest WITH that penalty --
 * few rounds count for more there than few sequences.  The estimate: the parse itself, without output, over the block's first
 * bytes, in cycles by what it met (rounds, sequences, groups of equal hashes, stops for them, matches counted in memory; weights
 * fitted to the phase probe, profiles/r5*_parse_probe.txt), scaled to the block's length, plus the penalty per round.  The order
 * is by that figure, most expensive first, in 16 steps per octave.
 */
constexpr uint32_t PCOST_SAMPLE = 3072u;
constexpr int PCOST_BUCKETS = 512;               /* 16 per octave from 2^8 on */
constexpr int PCOST_WAVES_PER_WG = 8;

__device__ __forceinline__ uint32_t pcost_bucket(unsigned long long cost)
{
    if (cost < 256ull) return 0u;
    const uint32_t l = 63u - (uint32_t)__clzll((long long)cost);
    const uint32_t b = 16u * (l - 8u) + (uint32_t)((cost >> (l - 4u)) & 15ull) + 1u;
    return b < (uint32_t)PCOST_BUCKETS ? b : (uint32_t)PCOST_BUCKETS - 1u;
}

__global__ __launch_bounds__(64 * PCOST_WAVES_PER_WG) void k4_pcost_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[PCOST_WAVES_PER_WG][4096 + PARSE_SEEN_DWORDS];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const long long b = (long long)blockIdx.x * PCOST_WAVES_PER_WG + (long long)wave;
    if (b >= a.n) return;
    const int src_len = a.srcLen[b];
    unsigned long long cost = 0ull;
    if (src_len >= (int)PARSE_MIN_LEN && src_len < LIMIT_64K) {
        const uint32_t sample = (uint32_t)src_len < PCOST_SAMPLE ? (uint32_t)src_len : PCOST_SAMPLE;
        ParseStats st;
        const uint32_t nseq = parse_block<1, false, true>(a.src + a.srcOff[b], sample, nullptr, (uint16_t *)lds[wave], lds[wave] + 4096, lane, nullptr, &st);
        const unsigned long long cyc = 6500ull * st.rounds + 3000ull * st.slow + 120ull * nseq + 150ull * st.groups + 480ull * st.lazies + 1500ull * st.longs;
        cost = cyc * (unsigned long long)(uint32_t)src_len / sample;
    } else if (src_len > 0) {
        cost = 120ull * (unsigned long long)(uint32_t)src_len;              /* (the one-kernel encoder's blocks: by length) */
    }
    if (lane == 0) {
        const uint32_t bkt = pcost_bucket(cost);
        a.cost[b] = bkt;
        atomicAdd(&a.hist[bkt], 1u);
    }
}

/* order[] = block indices, most expensive bucket first (hist: PCOST_BUCKETS counts, then PCOST_BUCKETS cursors, zeroed) */
__global__ __launch_bounds__(256) void k4_porder_kernel(BatchArgs a)
{
    __shared__ uint32_t before[PCOST_BUCKETS];
    for (int k = (int)threadIdx.x; k < PCOST_BUCKETS; k += 256) before[k] = a.hist[k];
    __syncthreads();
    if (threadIdx.x == 0) {                     /* suffix sums: blocks in more expensive buckets */
        uint32_t acc = 0u;
        for (int k = PCOST_BUCKETS - 1; k >= 0; k--) { const uint32_t c = before[k]; before[k] = acc; acc += c; }
    }
    __syncthreads();
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= a.n) return;
    const uint32_t bkt = a.cost[b];
    a.order_out[before[bkt] + atomicAdd(&a.hist[PCOST_BUCKETS + bkt], 1u)] = (uint32_t)b;
}

/* ------------------------------------------------------------------------------------------------------------------ */

/*
 * EMIT: LL64.fast.cs:237-382 and :469-503 for one block from its records, one wavefront per block, 64 sequences at a time, one per
 * lane: backward extension (:237-242), the sizes, their prefix sum, the output-limit checks (:251-255, :346-350, :471-476), token /
 * length bytes / literals / offset.  Runs of more than 32 literals and length fields of more than one byte are moved by the whole
 * wave.
 */
constexpr int EMIT_WAVES_PER_WG = 4;

/* HC: the records of the hash-chain parse (k4lz4_encode_hc.hpp, round 6) -- LZ4HC_encodeSequence (LL64.high.cs:435-510) writes the same
 * format from the same four numbers; what differs is that its match starts are final (no backward extension here: LZ4HC_countBack
 * ran inside the search) and its second output-limit test (:484: op + length / 255 + (1 + LASTLITERALS) > oend, against :346-350) */
/* BIG: records of a byU32 block -- a match length code of 0xffff says "this or more" and is counted again here (LL64.fast.cs:326-329) */
template <bool HC, bool BIG>
__device__ __forceinline__ bool emit_records(const uint8_t *src, const uint32_t U, uint8_t *dst, const int dst_cap, const uint2 *recs, const uint32_t nseq, const int lane, EmitState &st)
{
    const bool limited = dst_cap < compress_bound((int)U);         /* :524 */
    const uint64_t olimit = (uint64_t)(dst_cap < 0 ? 0 : dst_cap);
    uint32_t op = st.op, emitted_to = st.emitted_to;
    uint2 rn = make_uint2(0u, 0u);
    if ((uint32_t)lane < nseq) rn = rec_load(recs + lane);
    for (uint32_t base = 0u; base < nseq; base += 64u) {
        const uint32_t n = nseq - base < 64u ? nseq - base : 64u;
        const bool mine = (uint32_t)lane < n;
        const uint2 r = rn;
        if (base + 64u + (uint32_t)lane < nseq) rn = rec_load(recs + base + 64u + (uint32_t)lane);      /* the next 64, while these are written */
        const uint32_t pos = r.x, cpos = r.x - (r.y & 0xffffu);
        uint32_t code = r.y >> 16;
        if (BIG) {
            unsigned long long lng = ballot(mine && code == 0xffffu);
            while (lng) {
                const int g = ctz64(lng);
                lng &= lng - 1ull;
                const uint32_t gp = readlane_u32(pos, g), gc = readlane_u32(cpos, g);
                const uint32_t full = wave_count(src + gp + (uint32_t)MINMATCH, src + gc + (uint32_t)MINMATCH, (U - (uint32_t)LASTLITERALS) - (gp + (uint32_t)MINMATCH), lane);
                if (lane == g) code = full;
            }
        }
        const uint32_t end = mine ? pos + (uint32_t)MINMATCH + code : 0u;
        const uint32_t prev = (uint32_t)__shfl_up((int)end, 1);
        const uint32_t ls = lane == 0 ? emitted_to : prev;
        const uint32_t lit0 = mine ? pos - ls : 0u;
        const uint32_t maxback = HC ? 0u : (lit0 < cpos ? lit0 : cpos);    /* 0 right after a match */
        /* everything this pass needs from the source, asked for together -- the four bytes before position and candidate for the
         * backward extension, and the literal run in 8-byte pieces (a piece that holds a literal lies inside the block: a match
         * and the last literals follow) -- so that a pass waits for memory once */
        const bool near = mine && maxback != 0u && cpos >= 8u;
        uint64_t pb = 0, cb = 1;
        if (near) { pb = ld64u(src + pos - 8u); cb = ld64u(src + cpos - 8u); }
        uint64_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        const bool short_run = mine && lit0 != 0u && lit0 <= LANE_COPY_MAX;
        if (short_run) {
            v0 = ld64u(src + ls);
            if (lit0 > 8u) v1 = ld64u(src + ls + 8u);
            if (lit0 > 16u) v2 = ld64u(src + ls + 16u);
            if (lit0 > 24u) v3 = ld64u(src + ls + 24u);
        }
        uint32_t back = 0u;
        if (mine && maxback != 0u) {
            if (cpos >= 8u) {                                           /* eight bytes at once, the rare longer run eight at a time */
                const uint64_t y = pb ^ cb;
                back = y ? (uint32_t)__clzll((long long)y) >> 3 : 8u;
                if (back > maxback) back = maxback;
                bool full = y == 0ull;
                uint32_t done = 8u;
                while (full && done < maxback) {
                    if (done + 8u <= cpos) {
                        const uint64_t y2 = ld64u(src + pos - 8u - done) ^ ld64u(src + cpos - 8u - done);
                        back = done + (y2 ? (uint32_t)__clzll((long long)y2) >> 3 : 8u);
                        if (back > maxback) back = maxback;
                        full = y2 == 0ull;
                        done += 8u;
                    } else {
                        back = done;
                        while (back < maxback && src[pos - 1u - back] == src[cpos - 1u - back]) back++;
                        break;
                    }
                }
            } else {
                while (back < maxback && src[pos - 1u - back] == src[cpos - 1u - back]) back++;
            }
        }
        const uint32_t ll = lit0 - back, mc = mine ? code + back : 0u;
        const uint32_t lx = ll >= (uint32_t)RUN_MASK ? (ll - RUN_MASK) / 255u + 1u : 0u;
        const uint32_t mx = mc >= (uint32_t)ML_MASK ? (mc - ML_MASK) / 255u + 1u : 0u;
        const uint32_t sz = mine ? 1u + lx + ll + 2u + mx : 0u;
        const uint32_t incl = wave_inclusive_scan(sz);
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        const uint32_t o_tok = op + incl - sz;
        const uint32_t o_lit = o_tok + 1u + lx, o_off = o_lit + ll, o_mx = o_off + 2u;
        if (limited) {                                      /* :251-255, :346-350 */
            const bool fail = mine && ((uint64_t)o_tok + 1u + ll + (2 + 1 + LASTLITERALS) + ll / 255u > olimit ||
                                       (uint64_t)o_mx + (1 + LASTLITERALS) + (HC ? mc / 255u : (mc + 240u) / 255u) > olimit);
            if (ballot(fail)) return false;
        }
        if (mine) {
            st8_out(dst + o_tok, (uint8_t)(((ll < (uint32_t)RUN_MASK ? ll : (uint32_t)RUN_MASK) << ML_BITS) |
                                           (mc < (uint32_t)ML_MASK ? mc : (uint32_t)ML_MASK)));
            if (lx == 1u) st8_out(dst + o_tok + 1u, (uint8_t)(ll - RUN_MASK));
            st16u_out(dst + o_off, (uint16_t)(pos - cpos));   /* :299-304 */
            if (mx == 1u) st8_out(dst + o_mx, (uint8_t)(mc - ML_MASK));
        }
        if (short_run && ll != 0u) {                  /* exactly ll bytes, from the pieces */
            uint8_t *d = dst + o_lit;
            if (ll >= 8u) st64u_out(d, v0);
            if (ll >= 16u) st64u_out(d + 8, v1);
            if (ll >= 24u) st64u_out(d + 16, v2);
            if (ll >= 32u) st64u_out(d + 24, v3);
            const uint32_t n8 = ll >> 3;
            uint64_t vt = n8 == 0u ? v0 : n8 == 1u ? v1 : n8 == 2u ? v2 : v3;
            d += 8u * n8;
            if (ll & 4u) { st32u_out(d, (uint32_t)vt); vt >>= 32; d += 4; }
            if (ll & 2u) { st16u_out(d, (uint16_t)vt); vt >>= 16; d += 2; }
            if (ll & 1u) st8_out(d, (uint8_t)vt);
        }
        unsigned long long big = ballot(mine && (lit0 > LANE_COPY_MAX || lx > 1u || mx > 1u));
        while (big) {
            const int g = ctz64(big);
            big &= big - 1ull;
            const uint32_t g_ll = readlane_u32(ll, g), g_mc = readlane_u32(mc, g);
            const uint32_t g_tok = readlane_u32(o_tok, g);
            if (g_ll >= (uint32_t)RUN_MASK + 255u) emit_length_run(dst, g_tok + 1u, g_ll - RUN_MASK, lane);
            if (readlane_u32(lit0, g) > LANE_COPY_MAX) wave_copy(dst + readlane_u32(o_lit, g), src + readlane_u32(ls, g), g_ll, lane);
            if (g_mc >= (uint32_t)ML_MASK + 255u) emit_length_run(dst, readlane_u32(o_mx, g), g_mc - ML_MASK, lane);
        }
        op += total;
        emitted_to = readlane_u32(end, (int)n - 1);
    }
    st.op = op; st.emitted_to = emitted_to;
    return true;
}

/* ---- _last_literals (:469-503; LZ4HC's :751-787 are the same bytes and the same test) ---- */
__device__ __forceinline__ int emit_tail(const uint8_t *src, const uint32_t U, uint8_t *dst, const int dst_cap, const int lane, const EmitState &st)
{
    const bool limited = dst_cap < compress_bound((int)U);
    const uint64_t olimit = (uint64_t)(dst_cap < 0 ? 0 : dst_cap);
    uint32_t op = st.op;
    const uint32_t emitted_to = st.emitted_to;
    const uint32_t last_run = U - emitted_to;
    if (limited && (uint64_t)op + last_run + 1u + (last_run + 255u - RUN_MASK) / 255u > olimit) return 0;
    if (last_run >= (uint32_t)RUN_MASK) {
        if (lane == 0) dst[op] = (uint8_t)(RUN_MASK << ML_BITS);
        emit_length_run(dst, op + 1u, last_run - RUN_MASK, lane);
        op += 2u + (last_run - RUN_MASK) / 255u;
    } else {
        if (lane == 0) dst[op] = (uint8_t)(last_run << ML_BITS);
        op++;
    }
    wave_copy(dst + op, src + emitted_to, last_run, lane);
    op += last_run;
    return (int)op;
}

__global__ __launch_bounds__(64 * EMIT_WAVES_PER_WG) void k4_emit_kernel(BatchArgs a, ParseArgs p)
{
    const int lane = lane_id();
    const long long b = (long long)blockIdx.x * EMIT_WAVES_PER_WG + (long long)uni(threadIdx.x >> 6);
    if (b >= a.n) return;
    const uint32_t nseq = uni(p.meta[2ull * (unsigned long long)b]);
    if (nseq == PARSE_REST || nseq == PARSE_BIG) return;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    const int ret = emit_block(a.src + a.srcOff[b], (uint32_t)src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap,
                               p.recs + (unsigned long long)b * PARSE_REC_STRIDE, nseq, lane);
    if (lane == 0) a.outLen[b] = codec_encode_result(src_len, ret, a.flags);
}

/* the blocks the parse kernel left alone (PARSE_REST): the one-kernel encoder of k4lz4_encode_fast.hpp, table in LDS */
__global__ __launch_bounds__(64 * ENCODE_WAVES_PER_WG) void k4_encode_fast_rest_kernel(BatchArgs a, ParseArgs p)
{
    __shared__ __attribute__((aligned(16))) uint32_t tabs[ENCODE_WAVES_PER_WG][ENCODE_LDS_DWORDS];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const long long b = (long long)blockIdx.x * ENCODE_WAVES_PER_WG + (long long)wave;
    if (b >= a.n) return;
    if (uni(p.meta[2ull * (unsigned long long)b]) != PARSE_REST) return;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    int ret = 0;
    if (src_len > 0 || (a.flags & FLAG_RAW_RETURN))
        ret = compress_fast_block<true, false>(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, a.accel, tabs[wave], lane, nullptr, (a.flags & FLAG_X32) != 0, nullptr);
    if (lane == 0) a.outLen[b] = codec_encode_result(src_len, ret, a.flags);
}

}  // namespace k4
