/*
 * k4lz4_encode_fast.hpp -- batched L00_FAST LZ4 block encoder for gfx950, one wavefront per block.
 *
 * Replaces (for batches of independent blocks) the reference's
 *   LZ4Codec.Encode (level < L03_HC)   src/K4os.Compression.LZ4/LZ4Codec.cs:40-52
 *   LLxx.LZ4_compress_fast             Engine/LLxx.cs:65-75
 *   LL64.LZ4_compress_fast(_extState)  Engine/x64/LL64.fast.cs:517-576
 *   LL64.LZ4_compress_generic          Engine/x64/LL64.fast.cs:34-513   (noDict, noDictIssue;
 *                                      notLimited|limitedOutput; byU16 | byU32 + hash5)
 *   hash / table helpers               Engine/LL.tools.cs:46-148, x64/LL64.tools.cs:86-153
 * and produces byte-identical blocks.  The reference is a serial greedy state machine whose hash
 * table is mutated by every visited position; bit-exactness therefore forbids "hash everything in
 * parallel".  What the wavefront does instead: one ROUND handles the 64 consecutive probe
 * positions from the cursor, one per lane, and resolves every sequence that starts among them
 * (about 8 per round on text), see encode_fast_block.  The serial semantics "a probe sees the
 * puts of every earlier visited position, and positions inside a match are never put" are kept
 * exactly: lanes with equal hashes form groups, a lane's candidate is the highest VISITED lane of
 * its group below it, else the table entry as it stood at the start of the round.
 *
 * Table slots hold positions relative to the block start (currentOffset == 0, empty slot == 0).
 */
#pragma once
#include "k4lz4_common.hpp"

namespace k4 {

/* The *_seg twins of the global-table kernels (and k4_encode_seg_kernel) may take the registers of five waves per SIMD instead of
 * six: at 80 VGPRs they spilled to scratch memory inside the block loop (12-80 bytes per lane), at 85 / 82 they do not. */
#ifndef K4_SEG_WAVES_MIN
#define K4_SEG_WAVES_MIN 5
#endif
#ifndef K4_GTAB_DUTY
#define K4_GTAB_DUTY 3
#endif
#ifndef K4_PACE
#define K4_PACE 1
#endif

/* TYPE 0: byU32 + hash5 (LL64), 1: byU16 + hash4, 2: byU32 + hash4 (LL32 in a 64-bit process, LZ4Codec.Enforce32) */
template <int TYPE> struct FastTable;

template <> struct FastTable<1> {   /* byU16: 8192 x u16, hash4 >> 19 (LL.tools.cs:46-51) */
    uint16_t *t;
    __device__ __forceinline__ static uint32_t hash(const uint8_t *p) { return (ld32u(p) * 2654435761u) >> (32 - 13); }
    /* same hash from bytes already in registers: seq = bytes p..p+3, n0 = bytes p+4..p+7 */
    __device__ __forceinline__ static uint32_t hash_of(uint32_t seq, uint32_t) { return (seq * 2654435761u) >> (32 - 13); }
    __device__ __forceinline__ uint32_t get(uint32_t h) const { return t[h]; }
    __device__ __forceinline__ void put(uint32_t h, uint32_t pos) const { t[h] = (uint16_t)pos; }
    /* (an A/B build, K4_NT_GTAB: a table in memory read and written with the non-temporal hint, so that the seven such tables of a
     * workgroup -- 3.5 MiB per XCD, every line of them touched every round or two -- do not hold the L2 against the candidates' lines) */
    __device__ __forceinline__ uint32_t get_nt(uint32_t h) const
    {
#ifndef K4_HOST_EMU
        return __builtin_nontemporal_load(t + h);
#else
        return t[h];
#endif
    }
    __device__ __forceinline__ void put_nt(uint32_t h, uint32_t pos) const
    {
#ifndef K4_HOST_EMU
        __builtin_nontemporal_store((uint16_t)pos, t + h);
#else
        t[h] = (uint16_t)pos;
#endif
    }
};
template <> struct FastTable<0> {  /* byU32: 4096 x u32, hash5 (LL.tools.cs:53-58, LL64.tools.cs:135-143) */
    uint32_t *t;
    __device__ __forceinline__ static uint32_t hash(const uint8_t *p)
    {
        return (uint32_t)(((ld64u(p) << 24) * 889523592379ull) >> (64 - 12));
    }
    __device__ __forceinline__ static uint32_t hash_of(uint32_t seq, uint32_t n0)   /* five bytes: seq and the low byte of n0 */
    {
        return (uint32_t)((((((uint64_t)n0) << 32) | seq) << 24) * 889523592379ull >> (64 - 12));
    }
    __device__ __forceinline__ uint32_t get(uint32_t h) const { return t[h]; }
    __device__ __forceinline__ void put(uint32_t h, uint32_t pos) const { t[h] = pos; }
};
template <> struct FastTable<2> {  /* byU32: 4096 x u32, hash4 >> 20: x32/LL32.tools.cs:141-148 has no hash5 arm; x32/LL32.fast.cs:543-545 */
    uint32_t *t;
    __device__ __forceinline__ static uint32_t hash(const uint8_t *p) { return (ld32u(p) * 2654435761u) >> (32 - 12); }
    __device__ __forceinline__ static uint32_t hash_of(uint32_t seq, uint32_t) { return (seq * 2654435761u) >> (32 - 12); }
    __device__ __forceinline__ uint32_t get(uint32_t h) const { return t[h]; }
    __device__ __forceinline__ void put(uint32_t h, uint32_t pos) const { t[h] = pos; }
};

/* distance from the search start to the j-th probe of LL64.fast.cs:156-172
 * (step_0 = 1, step_i = (accel*64 + i - 1) >> 6). */
__device__ __forceinline__ uint32_t probe_offset(uint32_t j, uint32_t accel)
{
    if (j == 0) return 0;
    const uint32_t M = accel * 64u - 2u + j;
    const uint32_t q = M >> 6;
    return 1u + 32u * q * (q - 1u) + q * (M - 64u * q + 1u) - 32u * accel * (accel - 1u);
}

/* number of equal bytes at a[]/b[] (b < a), at most maxn  -- LL64.tools.cs:86-133 */
__device__ __forceinline__ uint32_t wave_count(const uint8_t *a, const uint8_t *b, uint32_t maxn, int lane)
{
    uint32_t done = 0;
    for (;;) {
        const uint32_t i = done + 4u * (uint32_t)lane;
        uint32_t neq = 0;
        if (i < maxn) {
            const uint32_t x = ld32u(a + i) ^ ld32u(b + i);
            const uint32_t avail = maxn - i < 4u ? maxn - i : 4u;
            const uint32_t e = x ? (uint32_t)(__ffs(x) - 1) >> 3 : 4u;
            neq = e < avail ? e : avail;
        }
        const unsigned long long notfull = ballot(neq != 4u);
        if (!notfull) { done += 256u; continue; }
        const int fl = ctz64(notfull);
        return done + 4u * (uint32_t)fl + __builtin_amdgcn_readlane(neq, fl);
    }
}

constexpr int ENCODE_SCRATCH_BYTES = 512;      /* same-hash detection: 4096 bits, one per hash value (per two of the byU16 table's) */
/* Sequences found but not written out yet, 8 bytes each: position; offset (16 bits), what is known about the bytes
 * before position and match (4 bits), match length code (12 bits; REC_CODE_MAX = "this or more", counted again when it
 * is written out).  A round finds ~8 on text, at most 16; they are written out a wave-full at a time, one per lane,
 * instead of by the ~8 lanes that found them. */
constexpr int ENCODE_REC_SLOTS = 64;
constexpr int ENCODE_REC_DWORDS = 2 * ENCODE_REC_SLOTS;
constexpr uint32_t REC_CODE_MAX = 4095u;
constexpr int ENCODE_STAGE_DWORDS = ENCODE_SCRATCH_BYTES / 4 + ENCODE_REC_DWORDS + 4;   /* + the wave's word for Pace */   /* what every encoder wave needs besides its table */
constexpr int ENCODE_LDS_DWORDS = 4096 + ENCODE_STAGE_DWORDS;                       /* hash table + the above */

/* length field tail: `rem` encoded as 255-run + final byte (LL64.fast.cs:262-272,:365-381,:484-495) */
__device__ __forceinline__ void emit_length_run(uint8_t *dst, uint32_t op, uint32_t rem, int lane)
{
    const uint32_t nb = rem / 255u;
    wave_fill(dst + op, 255, nb, lane);
    if (lane == 0) dst[op + nb] = (uint8_t)(rem - nb * 255u);
}

/* the 16 source bytes around position p: 4 before, the 4 compared ones, 8 after */
struct Around {
    uint32_t pre, seq;
    uint32_t n0, n1;           /* the 8 bytes after seq, as two words: a merged 12-byte load then needs no aligned register
                                * pair, hence no copy that would have to wait for the load right where it was issued */
    bool pre_ok;
};
__device__ __forceinline__ Around load_around(const uint8_t *src, uint32_t p)
{
    Around a;
    a.pre_ok = p >= 4u;
    if (a.pre_ok) {
        const U128u v = ld128u(src + p - 4u);
        a.pre = v.v[0]; a.seq = v.v[1]; a.n0 = v.v[2]; a.n1 = v.v[3];
    } else {
        a.pre = 0; a.seq = ld32u(src + p); a.n0 = ld32u(src + p + 4u); a.n1 = ld32u(src + p + 8u);
    }
    return a;
}

__device__ __forceinline__ unsigned long long geq_of(uint32_t q) { return ~0ull << q; }

/* What the 16+16 bytes around a position and its candidate say about the match extension:
 * bits 0-4 forward bytes beyond MINMATCH (0..8; 0..12 where the four bytes after those are known as well, `has2` with `a_n2` /
 * `b_n2`; 0..24 with the 16 further bytes `ax` / `bx` of both, `more`; capped at fwd_max), bits 5-7 equal bytes backwards
 * (0..4), 0x100 forward run continues past the known bytes, 0x200 the 4 bytes before both are known */
constexpr uint32_t EXT_FWD_MASK = 31u, EXT_BACK_SHIFT = 5u;
__device__ __forceinline__ uint32_t extension_info(uint32_t a_pre, uint32_t a_n0, uint32_t a_n1, bool a_ok, uint32_t b_pre, uint32_t b_n0,
                                                   uint32_t b_n1, bool b_ok, uint32_t fwd_max, bool more = false,
                                                   uint4 ax = make_uint4(0u, 0u, 0u, 0u), uint4 bx = make_uint4(0u, 0u, 0u, 0u),
                                                   bool has2 = false, uint32_t a_n2 = 0u, uint32_t b_n2 = 0u)
{
    const uint32_t x0 = a_n0 ^ b_n0, x1 = a_n1 ^ b_n1;
    uint32_t e = x0 ? (uint32_t)(__ffs((int)x0) - 1) >> 3 : (x1 ? 4u + ((uint32_t)(__ffs((int)x1) - 1) >> 3) : 8u);
    uint32_t known = 8u;
    if (has2 && !more) {
        const uint32_t x2 = a_n2 ^ b_n2;
        if (e == 8u) e += x2 ? (uint32_t)(__ffs((int)x2) - 1) >> 3 : 4u;
        known = 12u;
    }
    if (more) {
        const uint32_t y0 = ax.x ^ bx.x, y1 = ax.y ^ bx.y, y2 = ax.z ^ bx.z, y3 = ax.w ^ bx.w;
        const uint32_t e2 = y0 ? (uint32_t)(__ffs((int)y0) - 1) >> 3
                               : y1 ? 4u + ((uint32_t)(__ffs((int)y1) - 1) >> 3)
                                    : y2 ? 8u + ((uint32_t)(__ffs((int)y2) - 1) >> 3) : y3 ? 12u + ((uint32_t)(__ffs((int)y3) - 1) >> 3) : 16u;
        if (e == 8u) e += e2;
        known = 24u;
    }
    const uint32_t c = e < fwd_max ? e : fwd_max;
    const bool ok = a_ok && b_ok;
    const uint32_t y = a_pre ^ b_pre;
    const uint32_t nb = ok ? (y ? (uint32_t)__clz(y) >> 3 : 4u) : 0u;
    return c | (nb << EXT_BACK_SHIFT) | ((e == known && fwd_max > known) ? 0x100u : 0u) | (ok ? 0x200u : 0u);
}

/* The serial part of a round: from cursor lane q follow stop -> end of its match -> next stop, until a stop whose
 * word carries one of the 0xf40 flags (hit needs attention / window left / invalid lane) or no stop is left
 * (stop == 0).  Written out in scalar ISA: the compiler's version of this two-exit uniform loop carries its exit
 * reasons in extra mask registers and takes 18 instructions and three branches per hop; this takes 9 and two. */
__device__ __forceinline__ void hop_chain_ref(unsigned long long hmx, uint32_t hopv, uint32_t &q, unsigned long long &hits,
                                              int &f, uint32_t &hv, unsigned long long &stop)
{   /* what the ISA below does, in C: the emulator build runs this, k4_chain_selftest_kernel compares the two on the GPU */
    for (;;) {
        stop = hmx & (~0ull << q);
        if (!stop) break;
        f = ctz64(stop);
        hv = readlane_u32(hopv, f);
        hits |= 1ull << f;
        if (hv & 0xf40u) break;
        q = hv & 63u;
    }
}
__device__ __forceinline__ void hop_chain(unsigned long long hmx, uint32_t hopv, uint32_t &q, unsigned long long &hits,
                                          int &f, uint32_t &hv, unsigned long long &stop)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t t;
    asm volatile(
        "s_lshl_b64 %[stop], -1, %[q]\n\t"
        "s_and_b64 %[stop], %[stop], %[hmx]\n\t"
        "s_cbranch_scc0 .Lhop_end%=\n"
        ".Lhop_next%=:\n\t"
        "s_ff1_i32_b64 %[f], %[stop]\n\t"
        "v_readlane_b32 %[hv], %[hopv], %[f]\n\t"
        "s_bitset1_b64 %[hits], %[f]\n\t"
        "s_and_b32 %[t], %[hv], 0xf40\n\t"
        "s_cbranch_scc1 .Lhop_end%=\n\t"
        "s_and_b32 %[q], %[hv], 63\n\t"
        "s_lshl_b64 %[stop], -1, %[q]\n\t"
        "s_and_b64 %[stop], %[stop], %[hmx]\n\t"
        "s_cbranch_scc1 .Lhop_next%=\n"
        ".Lhop_end%=:"
        : [stop] "=&s"(stop), [q] "+s"(q), [hits] "+s"(hits), [f] "+s"(f), [hv] "+s"(hv), [t] "=&s"(t)
        : [hmx] "s"(hmx), [hopv] "v"(hopv)
        : "scc");
#else
    hop_chain_ref(hmx, hopv, q, hits, f, hv, stop);
#endif
}

/* The same chain with the one thing that used to interrupt it most done on the way: when a match covers lanes that later
 * lanes count on as candidates (flag 0x400 and no other), those later lanes fall back to their table candidates -- the
 * second word every lane at or above the cursor has ready (hopB, hit mask hmB) replaces the first in place.  `lostC` collects the lanes such
 * matches covered; `j1c` = 63 - the lane's candidate lane (63 - the lane itself where it has none, a lane the cursor has
 * not passed is never covered). */
__device__ __forceinline__ void hop_chain_pairs_ref(unsigned long long &hmx, unsigned long long hmB, uint32_t &hopv, uint32_t hopB, uint32_t j1c,
                                                    unsigned long long &lostC, uint32_t &q, unsigned long long &hits, int &f,
                                                    uint32_t &hv, unsigned long long &stop)
{
    for (;;) {
        stop = hmx & (~0ull << q);
        if (!stop) break;
        f = ctz64(stop);
        hv = readlane_u32(hopv, f);
        hits |= 1ull << f;
        const uint32_t t = hv & 0xf40u;
        if (t) {
            if (t != 0x400u) break;
            const uint32_t qn = hv & 63u;
            lostC |= (~1ull << f) & ~(~0ull << qn) & ~(1ull << ((qn - 2u) & 63u));
            const unsigned long long now = ballot((long long)(lostC << j1c) < 0);
            hmx ^= (hmx ^ hmB) & now;
            if (((now >> lane_id()) & 1ull) && (uint32_t)lane_id() >= qn) hopv = hopB;   /* below the cursor: final already */
        }
        q = hv & 63u;
    }
}
__device__ __forceinline__ void hop_chain_pairs(unsigned long long &hmx, unsigned long long hmB, uint32_t &hopv, uint32_t hopB, uint32_t j1c,
                                                unsigned long long &lostC, uint32_t &q, unsigned long long &hits, int &f,
                                                uint32_t &hv, unsigned long long &stop)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t t;
    unsigned long long sk, t2, vt;
    asm volatile(
        "s_lshl_b64 %[stop], -1, %[q]\n\t"
        "s_and_b64 %[stop], %[stop], %[hmx]\n\t"
        "s_cbranch_scc0 .Lh2_end%=\n"
        ".Lh2_next%=:\n\t"
        "s_ff1_i32_b64 %[f], %[stop]\n\t"
        "v_readlane_b32 %[hv], %[hopv], %[f]\n\t"
        "s_bitset1_b64 %[hits], %[f]\n\t"
        "s_and_b32 %[t], %[hv], 0xf40\n\t"
        "s_cbranch_scc1 .Lh2_special%=\n"
        ".Lh2_cont%=:\n\t"
        "s_and_b32 %[q], %[hv], 63\n\t"
        "s_lshl_b64 %[stop], -1, %[q]\n\t"
        "s_and_b64 %[stop], %[stop], %[hmx]\n\t"
        "s_cbranch_scc1 .Lh2_next%=\n\t"
        "s_branch .Lh2_end%=\n"
        ".Lh2_special%=:\n\t"
        "s_cmpk_lg_u32 %[t], 0x400\n\t"
        "s_cbranch_scc1 .Lh2_end%=\n\t"
        "s_and_b32 %[q], %[hv], 63\n\t"
        "s_lshl_b64 %[sk], -2, %[f]\n\t"
        "s_lshl_b64 %[t2], -1, %[q]\n\t"
        "s_andn2_b64 %[sk], %[sk], %[t2]\n\t"
        "s_sub_u32 %[t], %[q], 2\n\t"
        "s_bitset0_b64 %[sk], %[t]\n\t"
        "s_or_b64 %[lostC], %[lostC], %[sk]\n\t"
        "v_lshlrev_b64 %[vt], %[j1c], %[lostC]\n\t"
        "v_cmp_gt_i64_e64 %[t2], 0, %[vt]\n\t"
        "s_xor_b64 %[sk], %[hmx], %[hmB]\n\t"
        "s_and_b64 %[sk], %[sk], %[t2]\n\t"
        "s_xor_b64 %[hmx], %[hmx], %[sk]\n\t"
        "s_lshl_b64 %[sk], -1, %[q]\n\t"
        "s_and_b64 %[t2], %[t2], %[sk]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %[hopv], %[hopv], %[hopB], %[t2]\n\t"
        "s_branch .Lh2_cont%=\n"
        ".Lh2_end%=:"
        : [stop] "=&s"(stop), [q] "+s"(q), [hits] "+s"(hits), [f] "+s"(f), [hv] "+s"(hv), [t] "=&s"(t), [hmx] "+s"(hmx),
          [lostC] "+s"(lostC), [hopv] "+v"(hopv), [sk] "=&s"(sk), [t2] "=&s"(t2), [vt] "=&v"(vt)
        : [hmB] "s"(hmB), [hopB] "v"(hopB), [j1c] "v"(j1c)
        : "scc");
#else
    hop_chain_pairs_ref(hmx, hmB, hopv, hopB, j1c, lostC, q, hits, f, hv, stop);
#endif
}

/*
 * A big block by several wavefronts (k4lz4_segments.hpp says who gets what): the encoder's state between two sequences is
 * its cursor and its hash table, and of the table only the entries that can still be used (at most 65 535 bytes back,
 * LL64.fast.cs:219-224).  A wave that starts `warm` bytes before a boundary with an empty table ("warm run") usually has
 * that very state when it crosses the boundary (experiments/speculative_segments: always, on text, with 384 KiB), and
 * whether it has can be checked exactly.  So the warm run writes nothing until its first match end at or behind the
 * boundary (its CUT), publishes cut + table there and encodes on from it; the run before it, the true one, stops at ITS
 * first match end at or behind the boundary, and if that is the published cut and every usable table entry is equal,
 * the two outputs put one behind the other are the block's encoding byte for byte -- if not, the block is encoded again
 * the plain way.  A round may end at any match end (that is what a match running out of the window does), which is all the
 * round-based encoder needs for it.
 */
constexpr uint32_t SEG_NONE = 0xffffffffu;
constexpr uint32_t SEG_ITEM_WORDS = 10u;          /* a segment's record (k4lz4_segments.hpp, SegItem) as the words the encoder kernels read and write */
constexpr int SEG_SNAP_DWORDS = 4096 + 16;          /* [0] cut + 1 (0: not there yet, SEG_NONE: this run never found one), [16..] the table */
constexpr uint32_t SEG_ITEMS_MAX = 8192u;           /* = SEG_MAX_ITEMS (k4lz4_segments.hpp): segments of all cut blocks of a launch, at most */
constexpr uint32_t SEG_HDR_DWORDS = 64u;            /* k4lz4_capi.hip lays a launch's segment scratch out as header (256 bytes), then the items */
constexpr uint32_t SEG_SPIN_MAX = 1u << 20;          /* polls (with s_sleep 8 between them: some tenths of a second) before a run stops waiting for the next one's cut */
struct SegRun {
    uint32_t begin;             /* where the run starts probing: 0, or the start of the warm-up */
    uint32_t emit_from;         /* 0: output from the start; else nothing is written before the first match end at or behind this position */
    uint32_t stop_at;           /* SEG_NONE: to the end of the block; else stop at the first match end at or behind this position ... */
    uint32_t *snap_pub;         /* where a warm run publishes its cut and table, or nullptr */
    const uint32_t *snap_chk;   /* ... if it is the cut published here, with an equal table; or nullptr */
    const uint32_t *resume;     /* nullptr, or the table published at position `begin`, a verified cut: the run goes on from there as the
                                 * true run would (cursor right behind a match), writing from its first sequence */
    uint32_t *fix;              /* where a run that stops at a cut the next run is NOT in step with (state 4) leaves its table, for a run that goes
                                 * on from there: the item's slot of SegArgs::tables (for a later segment's wave that IS its table: nothing to copy) */
    uint32_t spin_max;          /* polls before a run stops waiting for the next one's cut (0: SEG_SPIN_MAX; K4LZ4_SEG_SPIN_MAX, the tests' way
                                 * to force that exit) */
    uint32_t cut, stop, state;  /* results: first position of the output (the cut; 0), one past its last (verified cut, or U),
                                 * 1 stopped at a verified cut, 2 ran to the end of the block, 3 no use (no cut, no room, gave up waiting),
                                 * 4 stopped at a cut the next run is NOT in step with: the output stands if this run began in step, and its
                                 * table now lies in the item's own table slot (`fix`) for a run that goes on from `stop`; the next run's
                                 * snapshot is left as that run published it */
};

/*
 * LL64.LZ4_compress_generic for one block.  `ldsw`: ENCODE_LDS_DWORDS dwords of LDS owned by this
 * wave (16 KiB hash table, zeroed here = LZ4_initStream, LL.tools.cs:235-239; then the 1 KiB bit set
 * of the same-hash detection).  Returns bytes written, 0 when the output does not fit.
 *
 * One round looks at 64 probe positions -- normally the 64 consecutive positions from the cursor --
 * and resolves EVERY sequence that starts inside that window, not just the first:
 *   load     16 source bytes around each position (4 before, the 4 hashed ones, 8 after; issued at
 *            the end of the previous round, before its table commit and output stores); hash; look
 *            the hashes up in the table as it stood at the start of the round; load the 16 bytes
 *            around the 64 candidates; per lane: would this position hit, how far does the match
 *            extend (up to 4 back / 12 forward from registers).
 *   groups   lanes whose hashes collide inside the window are found through one LDS bit per hash
 *            value (atomic OR: the old value tells a lane that another one was there first); each
 *            such lane keeps the bit mask of its group.  A lane's candidate is the
 *            highest visited-or-future lane of its group below it (its bytes come from that lane's
 *            registers), else the table entry.
 *   hops     wave-uniform chain over the window, one v_readlane per sequence: first stop at or
 *            after the cursor -> the lane after that match (precomputed per lane) -> next cursor.
 *            A hop leaves the tight loop only when the match runs past the known bytes (wave-wide
 *            count in memory), ends the window or the block, or covers a lane that may be a later
 *            lane's candidate (then the lanes inside matches are derived from the hits so far and
 *            the affected lanes fall back to their next candidate -- for the usual pair groups a
 *            select between two precomputed alternatives).  The test of the position right after
 *            a match (LL64.fast.cs:393-463) is simply the probe of that lane.
 *   derive   from the hit lanes alone: lanes inside matches (never put, except the `ip-2` position
 *            of LL64.fast.cs:394), the cursor each hit was found from (-> literal run, backward
 *            extension LL64.fast.cs:237-242), the sequences' numbers -- all lane-parallel.
 *   commit   visited lanes write the table, one writer per hash (the highest visited of a group).
 *   emit     output positions from a DPP prefix sum over the hit lanes; each hit lane writes its
 *            token / offset / length bytes, every literal position of the window stores its own
 *            byte (the lane holds it already); literals pending from before the window, and
 *            multi-byte length fields, are moved by the whole wave.
 * After 64 misses the schedule's step grows (LL64.fast.cs:156-172); those rounds probe the strided
 * positions and stop at their first sequence.
 */
template <bool BYU16, bool PROF = false, bool X32 = false, bool PAIRS = true, bool MORE = false, bool N2 = true>
__device__ __forceinline__ int encode_fast_block(const uint8_t *src, int src_len, uint8_t *dst, int dst_cap,
                                                 uint32_t accel, uint32_t *ldsw, int lane, unsigned long long *pc = nullptr,
                                                 bool dry_arg = false, uint32_t *seq_count = nullptr, uint32_t *gtab = nullptr,
                                                 uint32_t *pace_words = nullptr, SegRun *sr = nullptr)
{
    /* (what comes out of *sr is the same in every lane; uni() says so to the compiler, which would otherwise refuse it as an
     * operand of the scalar chains below) */
    const uint32_t sr_emit_from = sr ? uni(sr->emit_from) : 0u, sr_stop_at = sr ? uni(sr->stop_at) : SEG_NONE, sr_begin = sr ? uni(sr->begin) : 0u;
    bool dry = dry_arg || sr_emit_from != 0u;      /* a warm run writes nothing before its cut */
    uint32_t cut_pos = sr_emit_from ? sr_emit_from : sr_stop_at;   /* the next match end at or behind this ends the round (outcome 3) */
    if (sr) { sr->cut = 0u; sr->stop = 0u; sr->state = 3u; }
    uint32_t sequences = 0;

    unsigned long long c_probe = 0, c_ext = 0, c_emit = 0, n_seq = 0, n_round = 0, n_dup = 0, n_rt3 = 0;
    unsigned long long c_s1 = 0, c_s2 = 0, c_s3 = 0, c_s4 = 0, n_cont = 0;
    prof_place<PROF>(pc, 8, lane);
    const unsigned long long t_begin = prof_now<PROF>();
    if ((uint32_t)src_len > (uint32_t)MAX_INPUT_SIZE) return 0;     /* LL64.fast.cs:90 */
    const bool limited = dst_cap < compress_bound(src_len);         /* :524 */
    const uint64_t olimit = (uint64_t)(dst_cap < 0 ? 0 : dst_cap);
    const uint32_t U = (uint32_t)src_len;
    typedef FastTable<BYU16 ? 1 : (X32 ? 2 : 0)> Table;
    Table tab;
    /* the table normally lives in LDS; `gtab` (16 KiB of global memory) lets more blocks run per CU */
    uint32_t *const tabmem = gtab ? gtab : ldsw;
    tab.t = (decltype(tab.t))tabmem;
    uint32_t *const seen = gtab ? ldsw : ldsw + 4096;       /* bit h: a lane of the current window hashed to h */
    uint2 *const rec = (uint2 *)(seen + ENCODE_SCRATCH_BYTES / 4);        /* the pending sequences */
    uint32_t *const pace_mine = seen + ENCODE_SCRATCH_BYTES / 4 + ENCODE_REC_DWORDS;
    if (K4_PACE) Pace::begin(pace_words, pace_mine, lane);
    constexpr uint32_t REC_SLOTS = (uint32_t)ENCODE_REC_SLOTS, REC_FLUSH_AT = REC_SLOTS - 16u;
    constexpr uint32_t SEEN_SHIFT = BYU16 ? 1u : 0u;        /* hash value -> bit of `seen` */
    const unsigned long long me = 1ull << lane, below_me = me - 1ull;

    const bool resumed = sr && uni(sr->resume != nullptr ? 1u : 0u) != 0u;
    if (resumed) {
#pragma unroll 2
        for (int k = lane; k < 4096; k += 64) tabmem[k] = sr->resume[k];
    }
    else
    for (int k = lane; k < 1024; k += 64) ((uint4 *)tabmem)[k] = make_uint4(0u, 0u, 0u, 0u);
    for (int k = lane; k < ENCODE_SCRATCH_BYTES / 4; k += 64) seen[k] = 0u;
    wave_sync();

    uint32_t anchor = 0;
    uint32_t op = 0;
    uint32_t rec_head = 0, rec_count = 0;      /* the pending sequences: ring slots rec_head .. rec_head + rec_count - 1 */
    uint32_t emitted_to = 0;                   /* end of the last sequence written out = start of the next one's literals */

    /* ---- write out up to 64 pending sequences, one per lane (:237-382).  The literal run of a sequence starts where the
     * one before it ended; backward extension (:237-242) from the 4 bytes the finding lane knew, the rare longer one
     * counted here.  Returns false when the output does not fit (:251-255, :346-350). */
    auto flush = [&]() -> bool {
        const uint32_t n = rec_count < 64u ? rec_count : 64u;
        lds_sync();
        const bool mine = (uint32_t)lane < n;
        const uint32_t slot = (rec_head + (uint32_t)lane) & (REC_SLOTS - 1u);
        const uint2 r = rec[slot];
        const uint32_t pos = r.x, cpos = r.x - (r.y & 0xffffu), binfo = (r.y >> 16) & 15u;
        uint32_t code = r.y >> 20;
        unsigned long long longer = ballot(mine && code == REC_CODE_MAX);     /* :326-329 once more, from where the record stops */
        while (longer) {
            const int g = ctz64(longer);
            longer &= longer - 1ull;
            const uint32_t p = readlane_u32(pos, g) + MINMATCH, m = readlane_u32(cpos, g) + MINMATCH;
            const uint32_t c = REC_CODE_MAX + wave_count(src + p + REC_CODE_MAX, src + m + REC_CODE_MAX, U - LASTLITERALS - p - REC_CODE_MAX, lane);
            if (lane == g) code = c;
        }
        const uint32_t end = mine ? pos + MINMATCH + code : 0u;
        const uint32_t prev = (uint32_t)__shfl_up((int)end, 1);
        const uint32_t ls = lane == 0 ? emitted_to : prev;
        const uint32_t lit0 = mine ? pos - ls : 0u;
        const uint32_t maxback = lit0 < cpos ? lit0 : cpos;                /* 0 right after a match */
        const uint32_t nb = binfo & 7u;
        uint32_t back = nb < maxback ? nb : maxback;
        unsigned long long slow = ballot(mine && (!(binfo & 8u) || nb == 4u) && back < maxback);
        while (slow) {
            const int g = ctz64(slow);
            slow &= slow - 1ull;
            const uint32_t p = readlane_u32(pos, g), match = readlane_u32(cpos, g), mb = readlane_u32(maxback, g);
            uint32_t b = readlane_u32(back, g);
            while (b < mb) {
                const uint32_t i = b + (uint32_t)lane;
                const bool eq = i < mb && src[p - 1u - i] == src[match - 1u - i];
                const unsigned long long ne2 = ~ballot(eq);
                const int run = ne2 ? ctz64(ne2) : 64;
                b += (uint32_t)run;
                if (run < 64) break;
            }
            if (lane == g) back = b;
        }
        const uint32_t ll = lit0 - back, mc = mine ? code + back : 0u;
        const bool short_run = mine && ll != 0u && ll <= LANE_COPY_MAX;
        const uint32_t lx = ll >= (uint32_t)RUN_MASK ? (ll - RUN_MASK) / 255u + 1u : 0u;
        const uint32_t mx = mc >= (uint32_t)ML_MASK ? (mc - ML_MASK) / 255u + 1u : 0u;
        const uint32_t sz = mine ? 1u + lx + ll + 2u + mx : 0u;
        const uint32_t incl = wave_inclusive_scan(sz);
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        const uint32_t o_tok = op + incl - sz;
        const uint32_t o_lit = o_tok + 1u + lx, o_off = o_lit + ll, o_mx = o_off + 2u;
        if (limited) {                                      /* :251-255, :346-350 */
            const bool fail = mine && ((uint64_t)o_tok + 1u + ll + (2 + 1 + LASTLITERALS) + ll / 255u > olimit ||
                                       (uint64_t)o_mx + (1 + LASTLITERALS) + (mc + 240u) / 255u > olimit);
            if (ballot(fail)) return false;
        }
        if (mine) {
            dst[o_tok] = (uint8_t)(((ll < (uint32_t)RUN_MASK ? ll : (uint32_t)RUN_MASK) << ML_BITS) |
                                   (mc < (uint32_t)ML_MASK ? mc : (uint32_t)ML_MASK));
            if (lx == 1u) dst[o_tok + 1u] = (uint8_t)(ll - RUN_MASK);
            ((U16u *)(dst + o_off))->v = (uint16_t)(pos - cpos);   /* :299-304 */
            if (mx == 1u) dst[o_mx] = (uint8_t)(mc - ML_MASK);
        }
        if (short_run) lane_copy32(dst + o_lit, src + ls, ll, U - ls);
        unsigned long long big = ballot(mine && (ll > LANE_COPY_MAX || lx > 1u || mx > 1u));
        while (big) {
            const int g = ctz64(big);
            big &= big - 1ull;
            const uint32_t g_ll = readlane_u32(ll, g), g_mc = readlane_u32(mc, g);
            const uint32_t g_tok = readlane_u32(o_tok, g);
            if (g_ll >= (uint32_t)RUN_MASK + 255u) emit_length_run(dst, g_tok + 1u, g_ll - RUN_MASK, lane);
            if (g_ll > LANE_COPY_MAX) wave_copy(dst + readlane_u32(o_lit, g), src + readlane_u32(ls, g), g_ll, lane);
            if (g_mc >= (uint32_t)ML_MASK + 255u) emit_length_run(dst, readlane_u32(o_mx, g), g_mc - ML_MASK, lane);
        }
        op += total;
        emitted_to = readlane_u32(end, (int)n - 1);
        rec_head += n;
        rec_count -= n;
        return true;
    };

    if (src_len >= MFLIMIT + 1) {                                   /* :117 */
        const uint32_t mflimit_plus_one = U - MFLIMIT + 1;
        const uint32_t matchlimit = U - LASTLITERALS;

        const uint32_t begin = sr_begin;      /* (a warm run: the block as if it began here) */
        anchor = begin;
        emitted_to = begin;
        if (lane == 0 && !resumed) tab.put(Table::hash(src + begin), begin);     /* :119-122 */
        uint32_t ip = resumed ? begin : begin + 1u;       /* cursor of a fresh round */
        uint32_t sbase = begin + 1u;    /* where the current search loop started (:466) */
        uint32_t flag_pos = cut_pos < mflimit_plus_one ? cut_pos : mflimit_plus_one;   /* hits whose match ends here or later leave the chain */
        uint32_t jbase = 0;    /* probes of the current search already done (0: fresh round) */
        bool test = resumed;   /* fresh round only: `ip` is the position right after a match (:393-463) */

        /* the 64 probe positions of a round and their source bytes; issued for the NEXT round before the
         * current one commits and emits, so that the loads fly during that work */
        uint32_t pos_n;
        bool valid_n;
        Around pa_n;
        /* MORE: the 16 source bytes after those (positions 12 .. 27 behind the probe), where the block has them: matches of
         * up to 28 bytes are then measured from registers, and only longer ones cost a trip to memory in the middle of the
         * chain (LL64.fast.cs:326-329).  Loaded unconditionally at a clamped address, like the others. */
        uint4 px_n = make_uint4(0u, 0u, 0u, 0u);
        bool more_n = false;
        auto prepare = [&]() {
            const bool fresh_n = jbase == 0u;
            const uint32_t sbase_n = fresh_n ? (test ? ip + 1u : ip) : sbase;
            const uint32_t shift_n = (fresh_n && test) ? 1u : 0u;
            const bool t0 = shift_n != 0u && lane == 0;
            const uint32_t j = jbase + (uint32_t)lane - (t0 ? 0u : shift_n);
            uint32_t npos;
            if (fresh_n && accel == 1u) {
                pos_n = t0 ? ip : sbase_n + j;
                npos = sbase_n + j + 1u;
            } else {
                pos_n = t0 ? ip : sbase_n + probe_offset(j, accel);
                npos = sbase_n + probe_offset(j + 1u, accel);
            }
            valid_n = t0 || (npos <= mflimit_plus_one && npos >= sbase_n);   /* :172 */
            /* unconditional (an invalid lane reads the block's first bytes and nobody looks at them): a load inside a
             * predicated region gets its result copied into the merged registers right there, and the copy waits for it --
             * the loads would not fly during commit and emission at all */
            pa_n = load_around(src, valid_n ? pos_n : 0u);
            if (MORE) {
                more_n = valid_n && U >= 28u && pos_n <= U - 28u;
                const U128u v = ld128u(src + (more_n ? pos_n + 12u : 0u));
                px_n = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
            }
        };
        prepare();

        for (;;) {
            /* ---------------- load: positions, hashes, candidates ---------------- */
            K4_PHASE("load");
            const unsigned long long t0 = prof_now<PROF>();
            const bool fresh = jbase == 0u;
            if (fresh) sbase = test ? ip + 1u : ip;
            const bool contig = fresh && accel == 1u;               /* lane l probes position ip + l */
            const uint32_t shift = (fresh && test) ? 1u : 0u;       /* lane 0 = the test probe */
            const uint32_t ip0 = ip;
#if K4_GTAB_DUTY
            /* A SIMD issues for its oldest wave first, and the global-table kernel is launched second: its waves lost every
             * arbitration to the LDS-table waves on the same SIMD and ran 1.2-1.6x longer per block than those (4.08 ms
             * against 3.52 for the two kernels of the bench batch, per-block time stamps in profiles/r15_stamp_*.txt).  With
             * the higher priority all the time they win every arbitration instead and the LDS-table kernel becomes the long
             * one (3.76 / 4.10 ms); taking it for three of every four 64-byte steps of the cursor -- a round's cursor is as
             * good as a coin here -- evens the two out (3.90 / 3.93 ms, +4.5 % on the encode call).  Only without the
             * late-blocks-first priorities (Pace, k4lz4_common.hpp: K4LZ4_NO_PACE), which do the same and more by measurement. */
            if (gtab && !(K4_PACE && pace_words)) { if (((ip0 >> 6) & 3u) < (uint32_t)K4_GTAB_DUTY) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
            const uint32_t pos = pos_n;
            const bool valid = valid_n;
            const Around pa = pa_n;
            const uint4 px = px_n;
            const bool more = MORE && more_n && U >= 16u;
            if (shift && lane == 0) {
                /* :394 -- the put of ip - 2.  Lane 0 probes ip itself and holds the 4 bytes before it: the bytes at
                 * ip - 2 come out of its registers, not out of another trip to memory at the top of every such round */
                const uint32_t seq2 = (pa.pre >> 16) | (pa.seq << 16), n2 = (pa.seq >> 16) | (pa.n0 << 16);
                tab.put(Table::hash_of(seq2, n2), ip - 2u);
            }
            wave_sync();
            if (PROF) { n_round++; if (!fresh) n_cont++; }
            /* (taken here, before the candidate fetch is issued: a first use of the probe's bytes after it would make the
             * compiler wait for every load in flight, the fetch included) */
            const uint32_t pn2 = (uint32_t)__shfl_down((int)pa.n1, 4);
            uint32_t h = 0, cand = 0;
            bool flagged = false;              /* another lane of this window has the same hash (at least one lane of every group sees it) */
            if (valid) {
                h = Table::hash_of(pa.seq, pa.n0);
                cand = tab.get(h);
                const uint32_t bit = h >> SEEN_SHIFT;       /* two hash values may share a bit: the groups below compare the hashes */
                flagged = ((atomicOr(&seen[bit >> 5], 1u << (bit & 31u)) >> (bit & 31u)) & 1u) != 0u;
            }
            const Around ca = load_around(src, cand);       /* the round's one dependent trip to memory: everything below that
                                                             * does not need the candidate bytes runs while it is under way */
            /* Four more bytes behind both (positions 12 .. 15 after the probe): a match of up to 15 bytes is then measured
             * from registers -- most of the "long" matches of text (LL64.fast.cs:326-329 costs a trip to memory in the middle
             * of the chain otherwise).  The probe's come out of the lane four positions on, the candidate's are one more
             * dword of the same trip; lanes without such a neighbour, and strided rounds, stay at 12 known bytes. */
            const uint32_t fwd_max = matchlimit - (pos + MINMATCH);
            const bool has2 = N2 && !MORE && contig && lane < 60 && valid && fwd_max > 8u && fwd_max < 0x80000000u;
            const uint32_t cn2 = ld32u(src + (has2 ? cand + 12u : 0u));      /* cand + 16 <= pos + 15 < U where fwd_max > 8 */
            uint4 cx = make_uint4(0u, 0u, 0u, 0u);
            if (MORE) {                                     /* (cand < pos, so cand + 28 <= U where pos + 28 <= U) */
                const U128u v = ld128u(src + (more ? cand + 12u : 0u));
                cx = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
            }

            /* ---------------- groups: lanes of the window with equal hashes ---------------- */
            K4_PHASE("groups");
            unsigned long long G = me;
            {
                unsigned long long fl = ballot(flagged);
                if (valid) seen[h >> (5u + SEEN_SHIFT)] = 0u;      /* every lane has recorded its hash by now: wipe the words this window touched */
                while (fl) {
                    const int j = ctz64(fl);
                    const uint32_t hj = __builtin_amdgcn_readlane(h, j);
                    const bool same = valid && h == hj;
                    const unsigned long long m = ballot(same);
                    if (same) G = m;
                    fl &= ~m;
                }
            }
            const unsigned long long dirty = ballot(G != me);

            /* ---------------- resolve: every sequence that starts in the window ---------------- */
            K4_PHASE("resolve");
            const unsigned long long t1 = prof_now<PROF>();
            const unsigned long long inv_m = ballot(!valid), preok_m = ballot(pa.pre_ok);
            /* the common group is a pair: the later lane's candidate is the earlier lane while that one counts
             * as visited, else the table entry -- both known up front, so losing a candidate is a select */
            const unsigned long long gb = G & below_me;
            const int j1 = gb ? 63 - (int)__clzll((long long)gb) : -1;
            const unsigned long long cand0 = ballot((G & ~(below_me | me)) != 0ull);
            /* now the candidate bytes */
            const bool hit_tab = valid && (BYU16 || cand + (uint32_t)DISTANCE_MAX >= pos) && ca.seq == pa.seq;
            const uint32_t info = extension_info(pa.pre, pa.n0, pa.n1, pa.pre_ok, ca.pre, ca.n0, ca.n1, ca.pre_ok, fwd_max, more, px, cx, has2, pn2, cn2);
            /* per lane: the candidate as the table and the visited positions of this round define it */
            uint32_t cpos = cand, cinfo = info;
            bool chit = hit_tab;
            unsigned long long skipped = 0;    /* lanes inside matches: never put into the table */
            unsigned long long hmx, cand_m;    /* lanes that stop the chain (hits, invalid lanes); lanes that may be a later lane's candidate */
            uint32_t epos;                     /* per lane: where its match would end */
            /* candidate lanes (lanes with a later lane in their group) that ended up inside a match: they were never put.
             * Every such lane is recorded when the chain covers it, so for candidates this mask says all `skipped` would */
            unsigned long long lost_cands = 0;
            auto candidates = [&]() {          /* general form: highest visited-or-future lane of the group below this one */
                const unsigned long long eff = G & below_me & ~lost_cands;
                const int j = eff ? 63 - (int)__clzll((long long)eff) : -1;
                const int sj = j & 63;
                const uint32_t vj = (uint32_t)__shfl((int)pa.seq, sj), pj = (uint32_t)__shfl((int)pa.pre, sj);
                const uint32_t nlo = (uint32_t)__shfl((int)pa.n0, sj), nhi = (uint32_t)__shfl((int)pa.n1, sj);
                const uint32_t posj = (uint32_t)__shfl((int)pos, sj);
                const bool okj = ((preok_m >> sj) & 1ull) != 0;
                if (j >= 0) {
                    chit = valid && vj == pa.seq;
                    cpos = posj;
                    cinfo = extension_info(pa.pre, pa.n0, pa.n1, pa.pre_ok, pj, nlo, nhi, okj, fwd_max);
                } else {
                    chit = hit_tab;
                    cpos = cand;
                    cinfo = info;
                }
            };
            uint32_t xcode = 0xffffffffu;      /* per lane: match length found by the slow forward count */
            uint32_t hopv;                     /* per lane, as a hit: bits 0-6 lane after its match (>= 64: outside the window),
                                                * 0x100 match runs past the known bytes, 0x200 block ends after it (:391),
                                                * 0x400 its match covers a lane that may be a later lane's candidate */
            /* a lane's hop word, were it a hit with a match of MINMATCH + c8 bytes */
            auto hop_word = [&](uint32_t c8, uint32_t long_flag) -> uint32_t {
                const uint32_t qn_full = (uint32_t)lane + MINMATCH + c8;
                const uint32_t qn = contig && qn_full < 127u ? qn_full : 127u;
                /* lanes lane+1 .. qn-1 except qn-2 are never visited if this lane is a hit */
                const unsigned long long inside = ~(below_me | me) & ((1ull << (qn & 63u)) - 1ull) & ~(1ull << ((qn - 2u) & 63u));
                const bool trig = qn < 64u && (inside & cand_m) != 0ull;
                return (valid ? 0u : 0x800u) | qn | long_flag | (pos + MINMATCH + c8 >= flag_pos ? 0x200u : 0u) | (trig ? 0x400u : 0u);
            };
            auto publish = [&]() {
                hmx = ballot(chit) | inv_m;
                const bool counted = xcode != 0xffffffffu;          /* this lane's long match has been measured already */
                const uint32_t c8 = counted ? xcode : (cinfo & EXT_FWD_MASK);
                epos = pos + MINMATCH + c8;
                /* a visited-or-future lane with a later lane in its group may be that lane's candidate */
                cand_m = cand0 & ~lost_cands;
                hopv = hop_word(c8, counted ? 0u : (cinfo & 0x100u));
            };
            /* from the hits so far: the lanes inside their matches (never visited) */
            auto derive = [&](unsigned long long hits_now) {
                const unsigned long long hb = hits_now & below_me;
                const int ph = hb ? 63 - (int)__clzll((long long)hb) : 0;
                const uint32_t qp = (uint32_t)__shfl((int)hopv, ph) & 127u;       /* lane after the match of the hit below */
                const bool in = hb != 0ull && (uint32_t)lane < qp;
                skipped = ballot(in && (uint32_t)lane + 2u != qp);
            };
            bool general = false;
            if (dirty) {
                candidates();
            }
            publish();
            /* the word of a pair lane that has fallen back to its table candidate; candidate lanes lost so far.
             * (cand_m only shrinks while the round goes on: a stale one costs a needless check, never a missed one) */
            /* A lane with ONE candidate in the window has both its words ready, so that losing the candidate is a select; a
             * lane with several stops the chain the moment it would go by its second word (0x1000) and the candidates are
             * then worked out in full (`general`) for the rest of the round. */
            const bool many = (gb & (gb - 1ull)) != 0ull;
            const uint32_t hop_tab = many ? 0x1040u : (dirty ? hop_word(info & EXT_FWD_MASK, info & 0x100u) : hopv);
            const unsigned long long hmB = dirty ? (ballot(hit_tab || many) | inv_m) : hmx;
            const uint32_t j1c = 63u - (uint32_t)(j1 >= 0 ? j1 : lane);
            K4_PHASE("hops");
            const unsigned long long ta = prof_now<PROF>();
            if (PROF) c_s1 += ta - t1;
            unsigned long long t_rec = 0;
            unsigned long long hits = 0;       /* lanes where a sequence's match starts (before backward extension) */
            uint32_t q = 0;                    /* lane of the cursor */
            int outcome;                       /* 0 window exhausted, 1 next round starts after a match, 2 block ends */
            auto go_general = [&]() {
                general = true;
                candidates();
                publish();
            };
            /* lanes f+1 .. q-1 except q-2 were never visited: a lane whose candidate is one of them falls back
             * to the next visited lane of its group (or to the table); q-2 is the put of :394 */
            auto lose = [&](unsigned long long sk) {
                if (!(sk & cand_m)) return;
                const unsigned long long tr0 = prof_now<PROF>();
                if (PROF) n_dup++;
                lost_cands |= sk;
                const bool lost = (long long)(lost_cands << j1c) < 0;
                if (general) {
                    if (ballot(lost) & (~0ull << q)) go_general();   /* lanes behind the cursor no longer matter */
                } else {
                    hmx ^= (hmx ^ hmB) & ballot(lost);
                    if (lost && (uint32_t)lane >= q) {              /* what the lanes below the cursor hold is final */
                        hopv = hop_tab;
                        chit = hit_tab;
                        cpos = cand;
                        cinfo = info;
                        epos = pos + MINMATCH + (info & EXT_FWD_MASK);
                    }
                }
                if (PROF) t_rec += prof_now<PROF>() - tr0;
            };
            for (;;) {
                /* plain hops: one readlane per sequence; everything else about the chain is derived afterwards.
                 * Lanes past the last probe position (:172) stop the chain like hits and carry flag 0x800. */
                uint32_t hv = 0;
                int f = 0;
                unsigned long long stop;
                if (general || !PAIRS) hop_chain(hmx, hopv, q, hits, f, hv, stop);
                else hop_chain_pairs(hmx, hmB, hopv, hop_tab, j1c, lost_cands, q, hits, f, hv, stop);
                if (!stop || (hv & 0x800u)) {
                    hits &= ~inv_m;
                    if (hits) anchor = ip0 + q;
                    outcome = stop ? 2 : 0;
                    break;
                }
                if (hv & 0x1000u) {                                 /* not a hit yet: a lane of a larger group that has lost its first candidate */
                    hits &= ~(1ull << f);
                    go_general();
                    continue;
                }
                /* hit f needs attention before the chain can go on */
                const uint32_t qn = hv & 127u;
                /* a lane the chain itself has switched to its table candidate still holds the first one in cpos / epos */
                const bool second = PAIRS && !general && ((hv & 0x100u) || qn == 127u) && (long long)(lost_cands << readlane_u32(j1c, f)) < 0;
                uint32_t e_end = qn < 127u ? ip0 + qn
                                           : (second ? readlane_u32(pos, f) + MINMATCH + (readlane_u32(info, f) & EXT_FWD_MASK) : readlane_u32(epos, f));
                if (hv & 0x100u) {                                  /* :326-329 beyond the 12 known bytes */
                    if (PROF) n_rt3++;
                    const uint32_t p = readlane_u32(pos, f);
                    const uint32_t match = second ? readlane_u32(cand, f) : readlane_u32(cpos, f);
                    /* the bytes already compared: 8, or 24 where the lane had them (a flagged lane's count sits at what it knew) */
                    const uint32_t kn = (second ? readlane_u32(info, f) : readlane_u32(cinfo, f)) & EXT_FWD_MASK;
                    const uint32_t code = kn + wave_count(src + p + MINMATCH + kn, src + match + MINMATCH + kn, matchlimit - (p + MINMATCH) - kn, lane);
                    e_end = p + MINMATCH + code;
                    const uint32_t qf = e_end - ip0;
                    if (lane == f) { xcode = code; hopv = (hopv & ~127u) | (contig && qf < 127u ? qf : 127u); }
                }
                anchor = e_end;
                if (e_end >= mflimit_plus_one) { outcome = 2; break; }   /* :391 */
                if (e_end >= cut_pos) { outcome = 3; break; }            /* a segment's cut: the round ends at this match end */
                if (!contig || e_end - ip0 >= 64u) { outcome = 1; break; }
                q = e_end - ip0;
                lose((~1ull << f) & ((1ull << q) - 1ull) & ~(1ull << (q - 2u)));
            }
            K4_PHASE("derive");
            if (PAIRS && !general && (long long)(lost_cands << j1c) < 0) {   /* the lanes the chain switched to their table candidates, for what follows */
                chit = hit_tab;
                cpos = cand;
                cinfo = info;
            }
            if (hits) derive(hits);
            unsigned long long I = ~skipped; /* lanes whose position has been put into the table this round */
            if (outcome == 3 && anchor - ip0 < 64u) I &= (1ull << (anchor - ip0)) - 1ull;    /* the lanes behind a cut were not visited */
            const bool q_test = shift != 0u || hits != 0ull;       /* the cursor is a position right after a match */
            const unsigned long long tb = prof_now<PROF>();
            if (PROF) { c_s2 += tb - ta - t_rec; c_s3 += t_rec; }
            const uint32_t k = (uint32_t)__popcll(hits);
            sequences += k;
            if (PROF) n_seq += k;

            /* ---------------- where the next round starts; its source loads go out now ---------------- */
            K4_PHASE("next");
            if (outcome == 1 || outcome == 3) {
                ip = anchor;
                test = true;
                jbase = 0;
            } else if (outcome == 0) {
                if (contig) {
                    const uint32_t qs = q + (q_test ? 1u : 0u);     /* lane where the running search started */
                    sbase = ip0 + qs;
                    jbase = 64u - qs;
                    test = false;
                    ip = sbase;
                } else {
                    jbase += 64u - shift;
                    test = false;
                }
            }
            prepare();                                      /* also when the block ends here: every lane is invalid then and reads offset 0 */
            if (K4_PACE && ((ip ^ ip0) >> PACE_STEP_LOG2) != 0u && outcome != 2)      /* (a segment's run: its own stretch of the block) */
                Pace::update(pace_words, pace_mine, ip - sr_begin, (sr_stop_at != SEG_NONE && sr_stop_at < U ? sr_stop_at : U) - sr_begin, lane);

            /* the k sequences of this round join the pending ones */
            K4_PHASE("records");
            const bool mine = ((hits >> lane) & 1ull) != 0ull;
            if (k && !dry) {
                if (mine) {
                    const uint32_t slot = (rec_head + rec_count + (uint32_t)__popcll(hits & below_me)) & (REC_SLOTS - 1u);
                    const uint32_t code = xcode != 0xffffffffu ? xcode : (cinfo & EXT_FWD_MASK);
                    const uint32_t binfo = ((cinfo >> EXT_BACK_SHIFT) & 7u) | ((cinfo & 0x200u) ? 8u : 0u);
                    rec[slot] = make_uint2(pos, (pos - cpos) | (binfo << 16) | ((code < REC_CODE_MAX ? code : REC_CODE_MAX) << 20));
                }
                rec_count += k;
            }
            /* ---------------- commit the visited positions, one writer per hash ---------------- */
            K4_PHASE("commit");
            const unsigned long long t2 = prof_now<PROF>();
            if (PROF) c_s4 += t2 - tb;
            if (outcome != 2) {
                if (((I >> lane) & 1ull) && (G & I & ~(below_me | me)) == 0ull) tab.put(h, pos);
            }
            __builtin_amdgcn_wave_barrier();    /* the next round's put of ip - 2 may hit one of these slots and comes second (a wave's
                                                 * LDS accesses execute in program order; this keeps the compiler to it) */

            /* ---------------- write sequences out once a wave-full of them is pending (:244-382) ---------------- */
            K4_PHASE("flush");
            if (rec_count >= REC_FLUSH_AT && !flush()) return 0;
            K4_PHASE("round-end");
            if (PROF) { const unsigned long long t3 = prof_now<PROF>(); c_probe += t1 - t0; c_ext += t2 - t1; c_emit += t3 - t2; }

            if (outcome == 3) {
                const uint32_t cut = anchor;        /* the cursor, right behind a match; the table holds every visited position before it */
                /* ... and position cut - 2, which the reference puts right behind a match (:394) and this encoder at the top of
                 * the round that follows, or with this round's other positions when the match ended inside the window: here,
                 * so that both runs hold it whichever way their windows fell (the round that follows puts it once more) */
                if (lane == 0) tab.put(Table::hash(src + cut - 2u), cut - 2u);
                wave_sync();
                if (dry) {
                    /* the warm run has reached its cut: publish it with the table, write from here on */
#pragma unroll 2
                    for (int k = lane; k < 4096; k += 64) sr->snap_pub[16 + k] = tabmem[k];
                    wave_sync();
                    if (lane == 0) agent_publish(sr->snap_pub, cut + 1u);
                    dry = false;
                    emitted_to = cut;
                    sr->cut = cut;
                    cut_pos = sr_stop_at;
                    flag_pos = cut_pos < mflimit_plus_one ? cut_pos : mflimit_plus_one;
                } else {
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
                            const uint32_t mine = tabmem[k], other = sr->snap_chk[16 + k];
                            differ = differ || (mine != other && !(cut - mine > (uint32_t)DISTANCE_MAX && cut - other > (uint32_t)DISTANCE_MAX));
                        }
                        same = ballot(differ) == 0ull;
                    }
                    /* Not in step?  If THIS run began in step (the join finds that out), what it has written up to `cut` is the
                     * block's encoding up to there and its table is the block's table there: it keeps its output and leaves the
                     * table in its item's slot (`fix`; the other run's snapshot stays as it is -- it says what THAT run began with,
                     * which whoever arrives at this cut later must still be able to check) for a run that goes on from this cut
                     * (k4_seg_join_kernel).  A run that gave up waiting has nothing to offer: the block is then encoded again the
                     * plain way. */
                    if (!same && (theirs == 0u || !sr->fix)) return 0;
                    while (rec_count) if (!flush()) return 0;
                    if (!same && sr->fix != tabmem) {
#pragma unroll 2
                        for (int k = lane; k < 4096; k += 64) sr->fix[k] = tabmem[k];
                        wave_sync();
                    }
                    sr->stop = cut;
                    sr->state = same ? 1u : 4u;
                    return (int)op;
                }
            }
            if (outcome == 2) break;
        }
    }

    if (sr && sr_emit_from != 0u && dry) {             /* a warm run that never found its cut: nothing to offer */
        if (lane == 0) agent_publish(sr->snap_pub, SEG_NONE);
        return 0;
    }
    while (rec_count) if (!flush()) return 0;

    /* ---- _last_literals (:469-503) ---- */
    {
        const uint32_t last_run = U - anchor;
        if (limited && (uint64_t)op + last_run + 1u + (last_run + 255u - RUN_MASK) / 255u > olimit) return 0;
        if (last_run >= (uint32_t)RUN_MASK) {
            if (!dry) {
                if (lane == 0) dst[op] = (uint8_t)(RUN_MASK << ML_BITS);
                emit_length_run(dst, op + 1u, last_run - RUN_MASK, lane);
            }
            op += 2u + (last_run - RUN_MASK) / 255u;
        } else {
            if (!dry && lane == 0) dst[op] = (uint8_t)(last_run << ML_BITS);
            op++;
        }
        if (!dry) wave_copy(dst + op, src + anchor, last_run, lane);
        op += last_run;
    }
    if (seq_count) *seq_count = sequences;
    if (sr) { sr->stop = U; sr->state = 2u; }
    if (PROF && pc && lane == 0) {
        pc[0] = prof_now<PROF>() - t_begin; pc[1] = c_probe; pc[2] = c_ext; pc[3] = c_emit;
        pc[4] = n_seq; pc[5] = n_round; pc[6] = n_dup; pc[7] = n_rt3;
        pc[11] = c_s1; pc[12] = c_s2; pc[13] = c_s3; pc[14] = c_s4; pc[15] = n_cont;
    }
    prof_place<PROF>(pc, 9, lane);
    return (int)op;
}

/* LL64.LZ4_compress_fast (LL64.fast.cs:517-576): table type by input size */
template <bool PAIRS = true, bool MORE = false, bool N2 = true>
__device__ __forceinline__ int compress_fast_block(const uint8_t *src, int src_len, uint8_t *dst, int dst_cap,
                                                   int accel, uint32_t *ldsw, int lane, uint32_t *gtab = nullptr, bool x32 = false,
                                                   uint32_t *pace = nullptr, SegRun *sr = nullptr)
{
    const uint32_t a = accel < 1 ? 1u : (accel > 65536 ? 65536u : (uint32_t)accel);
    if (src_len < LIMIT_64K) return encode_fast_block<true, false, false, PAIRS, MORE, N2>(src, src_len, dst, dst_cap, a, ldsw, lane, nullptr, false, nullptr, gtab, pace);   /* (never cut) */
    if (x32) return encode_fast_block<false, false, true, PAIRS, MORE, N2>(src, src_len, dst, dst_cap, a, ldsw, lane, nullptr, false, nullptr, gtab, pace, sr);
    return encode_fast_block<false, false, false, PAIRS, MORE, N2>(src, src_len, dst, dst_cap, a, ldsw, lane, nullptr, false, nullptr, gtab, pace, sr);
}

/* LZ4Codec.Encode mapping (LZ4Codec.cs:40-52) */
__device__ __forceinline__ int codec_encode_result(int src_len, int ret, int flags)
{
    if (flags & FLAG_RAW_RETURN) return ret;
    if (src_len <= 0) return 0;
    return ret <= 0 ? -1 : ret;
}

/*
 * Dispatch order.  Workgroups start in blockIdx order and a batch is as slow as its last block, so
 * the expensive blocks should start first (longest-processing-time-first).  The cost of a block is
 * estimated by running the encoder without output over its first COST_SAMPLE bytes and scaling
 * the sequence count to the block length; blocks are then bucketed by cost (k4_order_kernel).
 */
/* The sample: 256 bytes for the two-step encoder's batches (blocks under 64 KiB, all resident at once: the order only decides which
 * nine of a workgroup's sixteen start with their table in LDS, and the estimate is part of the timed call -- 512 bytes cost the bench
 * batch 1 % more than their better order brings), 512 for everything else (ragged batches of pickles: the split between the
 * kernels and the longest-first order hang on it; with 256 rank 0's share of configs[3] comes out at 96 or at 120 ms from one
 * run to the next, with 512 at 98-106, gpurun_out/r96). */
constexpr int COST_SAMPLE = 512, COST_SAMPLE_PARSE = 256;
constexpr int COST_BUCKETS = 64;

__device__ __forceinline__ uint32_t cost_bucket(unsigned long long cost)
{
    if (cost < 4ull) return (uint32_t)cost;
    const uint32_t l = 63u - (uint32_t)__clzll(cost);            /* 2 buckets per octave */
    const uint32_t bkt = 2u * l + (uint32_t)((cost >> (l - 1u)) & 1ull);
    return bkt < (uint32_t)COST_BUCKETS ? bkt : (uint32_t)COST_BUCKETS - 1u;
}

/* a.cost[b] = bucket of block b; a.hist[bucket]++.  mode 0: encoder sample, mode 1: by length */
__global__ __launch_bounds__(64) void k4_cost_kernel(BatchArgs a, int by_length, int sample_bytes = COST_SAMPLE)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[ENCODE_LDS_DWORDS];
    const int lane = lane_id();
    const long long b = (long long)blockIdx.x;
    const int src_len = a.srcLen[b];
    unsigned long long cost = 0;
    if (src_len > 0) {
        if (by_length || src_len <= 64) {
            cost = (unsigned long long)src_len;
        } else {
            const int sample = src_len < sample_bytes ? src_len : sample_bytes;
            uint32_t nseq = 0;
            (void)encode_fast_block<true>(a.src + a.srcOff[b], sample, nullptr, 0x7fffffff, 1u, tab, lane, nullptr, true, &nseq);
            /* ~3 probe positions per sequence-free stretch count too: base cost by length */
            cost = ((unsigned long long)(nseq * 8u + (uint32_t)sample / 16u) * (unsigned long long)src_len) / (unsigned long long)sample;
        }
    }
    if (lane == 0) {
        const uint32_t bkt = cost_bucket(cost);
        a.cost[b] = bkt;
        atomicAdd(&a.hist[bkt], 1u);
    }
}

/* what a block of cost bucket `bkt` costs, in the units cost_bucket() cut into buckets (the middle of the bucket) */
__device__ __forceinline__ unsigned long long bucket_cost(uint32_t bkt)
{
    if (bkt < 4u) return bkt ? (unsigned long long)bkt * 4ull : 1ull;
    const uint32_t l = bkt >> 1;
    return ((bkt & 1u) ? 7ull : 5ull) << (l - 1u);            /* 1.25 x 2^l / 1.75 x 2^l, times four */
}

/* order[] = block indices, most expensive bucket first.  Its first thread also says where the second encoder kernel's part of
 * that order begins (hist[2 * COST_BUCKETS]): the most expensive blocks that hold `first` hundredths of the batch's COST go
 * to the LDS-table kernel -- for blocks of one size that is the same share of their number, for a ragged batch a few big
 * ones --, but never fewer than `total` of them (one residency of that kernel), when the batch has that many.
 * `lds_max`: the number of slots the host launched the LDS-table kernel with -- the count decided here never exceeds it
 * (the rounding-up of a bucket's part could otherwise give one more, and that block would be encoded by neither kernel). */
__global__ __launch_bounds__(256) void k4_order_kernel(BatchArgs a, uint32_t lds_max = 0xffffffffu)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.first != 0u) {
        unsigned long long all = 0;
        for (uint32_t k = 0; k < (uint32_t)COST_BUCKETS; k++) all += (unsigned long long)a.hist[k] * bucket_cost(k);
        const unsigned long long want = all / 100ull * (unsigned long long)a.first;
        unsigned long long have = 0, cnt = 0;
        for (int k = COST_BUCKETS - 1; k >= 0 && have < want; k--) {
            const unsigned long long c = a.hist[k], w = bucket_cost((uint32_t)k);
            if (have + c * w <= want) { have += c * w; cnt += c; }
            else { const unsigned long long part = (want - have + w - 1ull) / w; cnt += part; have = want; }
        }
        const unsigned long long floor_n = (unsigned long long)a.total < (unsigned long long)a.n ? (unsigned long long)a.total : (unsigned long long)a.n;
        if (cnt < floor_n) cnt = floor_n;
        if (cnt > (unsigned long long)a.n) cnt = (unsigned long long)a.n;
        if (cnt > (unsigned long long)lds_max) cnt = (unsigned long long)lds_max;
        a.hist[2 * COST_BUCKETS] = (uint32_t)cnt;
    }
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= a.n) return;
    const uint32_t bkt = a.cost[b];
    uint32_t before = 0;
    for (uint32_t k = bkt + 1u; k < (uint32_t)COST_BUCKETS; k++) before += a.hist[k];
    const uint32_t pos = before + atomicAdd(&a.hist[COST_BUCKETS + bkt], 1u);
    a.order_out[pos] = (uint32_t)b;
}

constexpr int ENCODE_WAVES_PER_WG = 2;      /* blocks per workgroup; measured 1: 53.1, 2: 54.2, 4: 54.3 GiB/s on the bench batch */
/* the first segment of a block that is cut into segments (k4lz4_segments.hpp; SegItem's leading fields are read here as words) */
struct SegFirst { SegRun run; uint32_t cap; bool cut; };
__device__ __forceinline__ SegFirst seg_first_of(const BatchArgs &a, long long b)
{
    SegFirst f;
    f.cut = false; f.cap = 0u;
    f.run.begin = 0u; f.run.emit_from = 0u; f.run.stop_at = SEG_NONE; f.run.snap_pub = nullptr; f.run.snap_chk = nullptr; f.run.resume = nullptr;
    f.run.cut = 0u; f.run.stop = 0u; f.run.state = 3u; f.run.spin_max = 0u; f.run.fix = nullptr;
    if (!a.seg_first) return f;
    const int32_t it = (int32_t)uni((uint32_t)a.seg_first[b]);
    if (it < 0) return f;
    const uint32_t *w = (const uint32_t *)a.seg_items + SEG_ITEM_WORDS * (uint32_t)it;     /* block, k, nseg, start, next_start, warm_from, cut, stop, state, bytes */
    f.cut = true;
    f.run.spin_max = uni(((const uint32_t *)a.seg_items)[-(int)SEG_HDR_DWORDS + 3]);      /* SegHdr::spin_max: the header sits SEG_HDR_DWORDS in front of the items */
    f.run.stop_at = uni(w[4]);
    f.cap = f.run.stop_at;                                          /* its piece may not reach into the next segment's */
    f.run.snap_chk = a.seg_snaps + (size_t)(it + 1) * SEG_SNAP_DWORDS;
    /* SegArgs::tables lies right behind the launch's snapshots (SegHdr::max_items of them, header word 7) */
    f.run.fix = a.seg_snaps + (size_t)uni(((const uint32_t *)a.seg_items)[-(int)SEG_HDR_DWORDS + 7]) * SEG_SNAP_DWORDS + 4096ull * (unsigned long long)it;
    return f;
}
__device__ __forceinline__ void seg_first_done(const BatchArgs &a, long long b, const SegFirst &f, int ret, int lane)
{
    if (!f.cut || lane != 0) return;
    uint32_t *w = (uint32_t *)a.seg_items + SEG_ITEM_WORDS * (uint32_t)a.seg_first[b];
    w[6] = f.run.cut; w[7] = f.run.stop; w[8] = ret > 0 ? f.run.state : 3u; w[9] = (uint32_t)ret;
}

template <bool MORE, bool SEG = false>
__device__ __forceinline__ void encode_fast_kernel_body(const BatchArgs &a, uint32_t (*tabs)[ENCODE_LDS_DWORDS])
{
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const long long slot = (long long)blockIdx.x * ENCODE_WAVES_PER_WG + (long long)wave;
    if (slot >= a.n || (a.split && slot >= (long long)uni(*a.split))) return;      /* the rest is the other kernel's */
    uint32_t *tab = tabs[wave];
    const long long b = a.order ? (long long)uni(a.order[slot]) : slot;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    const uint8_t *src = a.src + a.srcOff[b];
    uint8_t *dst = a.dst + a.dstOff[b];
    int ret = 0;
    if (a.prof) { prof_place<true>(a.prof + PROF_STRIDE * b, 8, lane); if (lane == 0) a.prof[PROF_STRIDE * b + 11] = MORE ? 3u : 1u; }
    if (SEG) {
        SegFirst f = seg_first_of(a, b);
        const int c = cap < 0 ? 0 : (f.cut && (uint32_t)cap > f.cap ? (int)f.cap : cap);
        if (src_len > 0 || (a.flags & FLAG_RAW_RETURN))
            /* (always the address of f.run -- neutral fields when the block is not cut -- never a select between it and nullptr:
             * a pointer that may be null keeps the struct in scratch memory, one that cannot lets it live in registers) */
            ret = compress_fast_block<true, MORE>(src, src_len, dst, c, a.accel, tab, lane, nullptr, (a.flags & FLAG_X32) != 0, a.pace, &f.run);
        seg_first_done(a, b, f, ret, lane);
    } else
    if (src_len > 0 || (a.flags & FLAG_RAW_RETURN))
        ret = compress_fast_block<true, MORE>(src, src_len, dst, cap < 0 ? 0 : cap, a.accel, tab, lane, nullptr, (a.flags & FLAG_X32) != 0, a.pace);
    if (lane == 0) a.outLen[b] = codec_encode_result(src_len, ret, a.flags);
    if (a.prof) prof_place<true>(a.prof + PROF_STRIDE * b, 9, lane);
}

__global__ __launch_bounds__(64 * ENCODE_WAVES_PER_WG) __attribute__((amdgpu_num_vgpr(80))) void k4_encode_fast_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t tabs[ENCODE_WAVES_PER_WG][ENCODE_LDS_DWORDS];
    encode_fast_kernel_body<false>(a, tabs);
}

/* For batches that leave the chip half empty (at most 8 blocks per CU: every block's own latency is what counts, not the
 * instruction slots it shares): the same kernel knowing 28 bytes behind every probe and candidate instead of 12, so that
 * matches of up to 28 bytes are measured from registers.  +4.6 % on a 512-block batch; on the full 4096-block batch the
 * 15 extra instructions per round and four more VGPRs cost 1.8 %, which is why it is a kernel of its own. */
__global__ __launch_bounds__(64 * ENCODE_WAVES_PER_WG) void k4_encode_fast_more_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t tabs[ENCODE_WAVES_PER_WG][ENCODE_LDS_DWORDS];
    encode_fast_kernel_body<true>(a, tabs);
}

/* the same encoder with its hash table in global memory (a.gtab: 16 KiB per workgroup slot) and
 * only the output stage in LDS: twice as many blocks resident per CU, each a little slower */
/* waves_per_eu(6): at most 80 VGPRs.  Two LDS-table waves and four of these fit one SIMD's register file only
 * below that line; one register more costs 20 % of the batch rate (measured; round 3 once more:
 * waves_per_eu(5) -27 %).  That line is also why this kernel runs without the pair chain (its scalar registers: mr blocks
 * 9 % faster with it, osdb / samba 5 % slower, the call slower) and without the twelve known bytes (N2: one vector
 * register too many). */
template <bool SEG>
__device__ __forceinline__ void encode_fast_gtab_kernel_body(const BatchArgs &a, uint32_t (*stages)[ENCODE_STAGE_DWORDS])
{
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const long long slot = (long long)blockIdx.x * ENCODE_WAVES_PER_WG + (long long)wave;
    if (slot >= a.n) return;
    uint32_t *stage = stages[wave];
    /* its part of the order begins where the LDS-table kernel's ends (*split, on top of `first`) */
    const long long at = (long long)a.first + (a.split ? (long long)uni(*a.split) : 0ll) + slot;
    if (a.order && at >= (long long)a.total) return;
    const long long b = a.order ? (long long)uni(a.order[at]) : slot;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    int ret = 0;
    if (a.prof) { prof_place<true>(a.prof + PROF_STRIDE * b, 8, lane); if (lane == 0) a.prof[PROF_STRIDE * b + 11] = 2u; }
    if (SEG) {
        SegFirst f = seg_first_of(a, b);
        const int c = cap < 0 ? 0 : (f.cut && (uint32_t)cap > f.cap ? (int)f.cap : cap);
        if (src_len > 0 || (a.flags & FLAG_RAW_RETURN))
            ret = compress_fast_block<false, false, false>(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], c, a.accel, stage, lane,
                                             a.gtab + 4096ull * (unsigned long long)slot, (a.flags & FLAG_X32) != 0, a.pace, &f.run);
        seg_first_done(a, b, f, ret, lane);
    } else
    if (src_len > 0 || (a.flags & FLAG_RAW_RETURN))
        ret = compress_fast_block<false, false, false>(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, a.accel, stage, lane,
                                         a.gtab + 4096ull * (unsigned long long)slot, (a.flags & FLAG_X32) != 0, a.pace);
    if (lane == 0) a.outLen[b] = codec_encode_result(src_len, ret, a.flags);
    if (a.prof) prof_place<true>(a.prof + PROF_STRIDE * b, 9, lane);
}
__global__ __launch_bounds__(64 * ENCODE_WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(6, 6))) void k4_encode_fast_gtab_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t stages[ENCODE_WAVES_PER_WG][ENCODE_STAGE_DWORDS];
    encode_fast_gtab_kernel_body<false>(a, stages);
}
/* the two kernels once more for launches in which big blocks are cut into segments (k4lz4_segments.hpp): a cut block's first
 * segment comes their way like any block, with the rule where to stop; kernels of their own so that the others carry none of it */
__global__ __launch_bounds__(64 * ENCODE_WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(K4_SEG_WAVES_MIN, 6))) void k4_encode_fast_gtab_seg_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t stages[ENCODE_WAVES_PER_WG][ENCODE_STAGE_DWORDS];
    encode_fast_gtab_kernel_body<true>(a, stages);
}
__global__ __launch_bounds__(64 * ENCODE_WAVES_PER_WG) __attribute__((amdgpu_num_vgpr(88))) void k4_encode_fast_seg_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t tabs[ENCODE_WAVES_PER_WG][ENCODE_LDS_DWORDS];
    encode_fast_kernel_body<false, true>(a, tabs);
}

/* diagnostic twin (blocks < 65547 B only): per-phase cycle counters (a.prof, 8 per block) */
__global__ __launch_bounds__(64) void k4_encode_fast_prof_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[ENCODE_LDS_DWORDS];
    const int lane = lane_id();
    const long long b = a.order ? (long long)uni(a.order[blockIdx.x]) : (long long)blockIdx.x;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    int ret = 0;
    if (src_len > 0 && src_len < LIMIT_64K)
        ret = encode_fast_block<true, true>(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, 1u, tab, lane, a.prof + PROF_STRIDE * b,
                                            false, nullptr, a.gtab ? a.gtab + 4096ull * (unsigned long long)blockIdx.x : nullptr);
    if (lane == 0) a.outLen[b] = codec_encode_result(src_len, ret, a.flags);
}

}  // namespace k4
