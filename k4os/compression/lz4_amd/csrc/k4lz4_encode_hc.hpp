/*
 * k4lz4_encode_hc.hpp -- batched LZ4 HC block encoder for gfx950: hash chain (levels L03..L09) and the optimal
 * parser (L10..L12, LL64.high.cs:802-1122, at the end of this file).
 *
 * Replaces (for batches of independent blocks) the reference's
 *   LZ4Codec.Encode (level >= L03_HC)        src/K4os.Compression.LZ4/LZ4Codec.cs:48-51
 *   LLxx.LZ4_compress_HC                     Engine/LLxx.cs:94-103
 *   LL64.LZ4_compress_HC .. _fastReset       Engine/x64/LL64.high.cs:1336-1381
 *   LZ4HC_init_internal / clearTables        Engine/LL.high.cs:142-166
 *   LZ4HC_compress_generic / clTable         Engine/x64/LL64.high.cs:1124-1189
 *   LZ4HC_compress_hashChain                 Engine/x64/LL64.high.cs:512-800
 *   LZ4HC_InsertAndGetWiderMatch             Engine/x64/LL64.high.cs:70-383 (noDictCtx, no chainSwap;
 *                                            pattern analysis, :208-337, at level 9 = 256 attempts)
 *   LZ4HC_countPattern / reverseCountPattern Engine/x64/LL64.high.cs:37-68, Engine/LL.high.cs:232-254
 *   LZ4HC_Insert / LZ4HC_countBack           Engine/LL.high.cs:102-122,:216-230
 *   LZ4HC_encodeSequence                     Engine/x64/LL64.high.cs:435-510
 * with byte-identical output.
 *
 * Key observation: unlike the fast encoder, the HC tables do not depend on the parse.
 * LZ4HC_Insert enters EVERY position, in order, before it can be searched, so the chain of
 * candidates seen by a search at position p -- "previous position with the same 15-bit hash",
 * repeatedly -- is a pure function of the data.  The work is therefore split in two kernels:
 *
 *   k4_hc_chain_kernel   (data-parallel)  64 positions per step: hash, look up / update a per-block
 *                        hash table, resolve equal hashes inside the step in lane order -> prev[p];
 *                        then pointer-jump prev[] three times so that every position holds its
 *                        first four chain candidates in one 16-byte record cand[p][0..3].
 *   k4_hc_parse_kernel   (serial per block, one wavefront) the reference's match arbitration
 *                        (LL64.high.cs:553-749) with wave-uniform state.  A search loads one record
 *                        and evaluates its four candidates at once, 16 lanes per candidate (4 bytes
 *                        per lane per step forwards, bytewise backwards); the reference's
 *                        "first candidate in chain order that beats `longest`" rule is applied to
 *                        the four lengths.  Levels above 3 continue with the record of the last
 *                        candidate.
 *
 * Scratch (context-owned, device): per block a 128 KiB hash table (u32 x 32768, zero = empty with
 * positions stored +1; blocks of at most 64 KiB keep theirs in LDS), prev[U] (u32) and one 16-byte record per
 * position (its first four chain candidates as distances, and per candidate the forward / backward match
 * length, exact below HC_FLEN_CAP / HC_BLEN_CAP: k4_hc_cand_kernel), see k4_hc_layout_kernel.  With the
 * lengths precomputed, level 3 (4 attempts = exactly one record) needs no data compare at all in the common
 * case: a search is one record load, literal runs are skipped 64 positions per load.
 */
#pragma once
#include "k4lz4_common.hpp"

namespace k4 {

#ifndef K4_HC_PACE
#define K4_HC_PACE 1
#endif
constexpr int HC_HASH_LOG = 15;
constexpr uint32_t HC_NONE = 0xffffffffu;
constexpr int HC_OPTIMAL_ML = (ML_MASK - 1) + MINMATCH;   /* LL.types.cs:74 */

struct HcArgs {
    const uint8_t *src;
    const uint64_t *srcOff;
    const int32_t *srcLen;
    uint8_t *dst;
    const uint64_t *dstOff;
    const int32_t *dstCap;
    int32_t *outLen;
    long long n;
    int level;
    int flags;
    uint32_t *hash;            /* n x 32768, zeroed before k4_hc_chain_kernel */
    uint8_t *work;             /* prev[] and cand[] of every block */
    unsigned int posBase;      /* k4_hc_cand_kernel: first position covered by chunk 0 of this launch */
    unsigned int candChunks;   /* ... and how many chunks of HC_CAND_POS_PER_WG positions per block it covers (grid: 8 * chunks workgroups per eight blocks) */
    unsigned long long *workOff;   /* n + 2: byte offset of block i's work area, [n] = total, [n+1] = longest block (k4_hc_layout_kernel) */
    unsigned long long workCap;    /* bytes behind `work` when the launch was sized without asking the device (0 = sized from [n]) */
    unsigned int maxLen;           /* the longest block the launch was sized for (same case) */
    uint32_t *status;              /* the context's status word (k4lz4_common.hpp), or nullptr */
    uint32_t *pace;                /* the parse kernel's late-blocks-first slots (k4lz4_common.hpp, Pace), zeroed; or nullptr */
    unsigned int blockBase;        /* chain kernels: workgroup 0 is block blockBase (the two of them share a launch chunk) */
    long long nChain;              /* ... and this many blocks from there on are this kernel's */
    uint2 *recs;                   /* round 6, k4_hc_parse_kernel at level 3 on blocks of at most 64 KiB: PARSE_REC_STRIDE sequence records per block
                                    * (k4lz4_parse.hpp: x = where the match starts, y = offset | (length - MINMATCH) << 16) -- the parse only DECIDES,
                                    * the block's bytes are written from the records afterwards (emit_block<true>); nullptr: LZ4HC_encodeSequence
                                    * inside the parse loop as before */
};

/* a launch sized from a reservation (k4lz4_ctx_reserve_hc) whose batch turned out bigger: nothing is touched, every block
 * fails and the context's status word says why */
__device__ __forceinline__ bool hc_scratch_ok(const HcArgs &a)
{
    return a.workCap == 0ull || (a.workOff[a.n] <= a.workCap && a.workOff[a.n + 1] <= (unsigned long long)a.maxLen);
}

__device__ __forceinline__ uint32_t hc_hash(uint32_t v) { return (v * 2654435761u) >> (MINMATCH * 8 - HC_HASH_LOG); }
constexpr uint32_t HC_FLEN_CAP = 4 + 32;   /* precomputed match lengths are exact below this value */
constexpr uint32_t HC_BLEN_CAP = 32;       /* precomputed backward lengths are exact below this value */
constexpr uint32_t HC_WORK_PER_BYTE = 20u;   /* prev[] 4 + one 16-byte record (k4_hc_cand_kernel) */
__device__ __forceinline__ uint64_t hc_work_bytes(int len) { return len > 0 ? (((uint64_t)len + 3u) & ~3ull) * HC_WORK_PER_BYTE : 0u; }

/* exclusive scan of the per-block work sizes (one workgroup; n is at most a launch chunk) */
__global__ __launch_bounds__(256) void k4_hc_layout_kernel(HcArgs a)
{
    __shared__ unsigned long long part[256];
    __shared__ unsigned long long pmax[256];
    const int t = (int)threadIdx.x;
    const long long per = (a.n + 255) / 256;
    const long long lo = (long long)t * per, hi = lo + per < a.n ? lo + per : a.n;
    unsigned long long s = 0, mx = 0;
    for (long long i = lo; i < hi; i++) {
        s += hc_work_bytes(a.srcLen[i]);
        if (a.srcLen[i] > 0 && (unsigned long long)a.srcLen[i] > mx) mx = (unsigned long long)a.srcLen[i];
    }
    part[t] = s;
    pmax[t] = mx;
    __syncthreads();
    unsigned long long base = 0;
    for (int k = 0; k < t; k++) base += part[k];
    if (t == 0) {
        unsigned long long m = 0;
        for (int k = 0; k < 256; k++) m = pmax[k] > m ? pmax[k] : m;
        a.workOff[a.n + 1] = m;
    }
    for (long long i = lo; i < hi; i++) {
        a.workOff[i] = base;
        base += hc_work_bytes(a.srcLen[i]);
    }
    if (t == 255) {
        a.workOff[a.n] = base;
        if (a.workCap != 0ull && base > a.workCap) dev_status_raise(a.status, (uint32_t)DEV_STATUS_HC_SCRATCH);
    }
    if (t == 0 && a.workCap != 0ull && a.workOff[a.n + 1] > (unsigned long long)a.maxLen) dev_status_raise(a.status, (uint32_t)DEV_STATUS_HC_SCRATCH);
}

/* ---- kernel 1: chains ------------------------------------------------------------------- */
/* prev[p] = nearest earlier position with the same hash (HC_NONE if there is none).  64 positions per step; positions
 * of one step with equal hashes are rare (15-bit hash) and found through one LDS bit per hash value -- the atomic OR's
 * old value tells a lane that another one was there first -- so that only the groups that exist are walked.
 * TAB: the hash table, slot = position + 1 (0 = empty): uint32 in context scratch (any block length), or uint16 in LDS for
 * blocks of up to 64 KiB -- 4096 tables of 128 KiB in memory are 512 MiB of two-byte-at-a-time traffic past every cache,
 * 64 KiB of LDS per block is two blocks per CU and a look-up at LDS latency. */
/* PARTS > 1 (round 6, k4_hc_chain_part_kernel): the wave owns the hash values with h % PARTS == part and a table of 32768 / PARTS slots
 * indexed by h / PARTS; it looks at every position of the block (the hash of 64 positions is a load, a multiply and a shift) and
 * enters only its own -- PARTS waves per block side by side, each with a sixteenth of the table work and none of each other's. */
#ifndef K4_HC_CHAIN_AHEAD
#define K4_HC_CHAIN_AHEAD 16
#endif
/* `stage` (PARTS > 1): 2 x 64 * AHEAD dwords of LDS shared by the block's waves.  Each wave owns one position in PARTS, so its store of
 * prev[] is a handful of lanes per step -- 61.6 M store instructions and 4.6 GB of partial-sector writes per launch of the bench batch for
 * 1.07 GB of prev[].  The owners put their values into the stage instead; after AHEAD steps (a barrier: all waves of a block walk the
 * same positions) every wave writes 64 of the stage's positions out, whole lines.  Two stages in turn: the barrier behind the next
 * group of steps is also the one that says the previous write-out is over. */
template <typename TAB, int PARTS = 1, int AHEAD = 4>
__device__ __forceinline__ void hc_chain_block(const uint8_t *src, uint32_t U, TAB *tab, uint32_t *prev, uint32_t *seen, int lane, uint32_t part = 0u,
                                               uint32_t *stage = nullptr)
{
    static_assert((PARTS & (PARTS - 1)) == 0, "a power of two");
    const uint32_t npos = U - 3u;                          /* positions whose 4 bytes exist */
    const unsigned long long below_me = (1ull << lane) - 1ull, above_me = ~(below_me | (1ull << lane));
    /* The source bytes of a step are asked for AHEAD steps ahead (unconditionally, at a clamped position), into a register
     * of their own: the loop is unrolled AHEAD times so that no word has to be moved from one register to another while it
     * is still on its way -- such a move waits for the load, and a step would wait for memory after all. */
    uint32_t wq[AHEAD];
#pragma unroll
    for (uint32_t u = 0; u < (uint32_t)AHEAD; u++) wq[u] = ld32u(src + (64u * u + (uint32_t)lane < npos ? 64u * u + (uint32_t)lane : 0u));
    for (uint32_t q0 = 0; q0 < npos; q0 += 64u * (uint32_t)AHEAD) {
#pragma unroll
        for (uint32_t u = 0; u < (uint32_t)AHEAD; u++) {
            const uint32_t p0 = q0 + 64u * u;
            const uint32_t p = p0 + (uint32_t)lane;
            const uint32_t w = wq[u];
            wq[u] = ld32u(src + (p + 64u * (uint32_t)AHEAD < npos ? p + 64u * (uint32_t)AHEAD : 0u));
            const uint32_t hh = hc_hash(w);
            const bool act = p < npos && (PARTS == 1 || (hh & (uint32_t)(PARTS - 1)) == part);
            uint32_t h = 0, pr = HC_NONE;
            bool flagged = false;
            if (act) {
                h = PARTS == 1 ? hh : hh / (uint32_t)PARTS;
                const uint32_t t = tab[h];
                pr = t ? t - 1u : HC_NONE;
                flagged = ((atomicOr(&seen[h >> 5], 1u << (h & 31u)) >> (h & 31u)) & 1u) != 0u;
            }
            bool last = act;                               /* last lane of its hash inside this step */
            unsigned long long fl = __ballot(flagged);
            if (act) seen[h >> 5] = 0u;                     /* every lane has recorded its hash: wipe the words this step touched */
            while (fl) {
                const int j = ctz64(fl);
                const uint32_t hj = __builtin_amdgcn_readlane(h, j);
                const bool same = act && h == hj;
                const unsigned long long m = __ballot(same);
                if (same) {
                    const unsigned long long lower = m & below_me;
                    if (lower) pr = p0 + 63u - (uint32_t)__clzll((long long)lower);
                    last = (m & above_me) == 0ull;
                }
                fl &= ~m;
            }
            if (PARTS > 1 && stage) { if (act) stage[(q0 / (64u * (uint32_t)AHEAD) & 1u) * 64u * (uint32_t)AHEAD + 64u * u + (uint32_t)lane] = pr; }
            else if (act) prev[p] = pr;
            if (last) tab[h] = (TAB)(p + 1u);
            /* the next step's look-ups come after these puts: program order for a table in LDS (nothing waits for the
             * stores of prev[], which nobody reads here), wave_sync for one in memory */
            if (sizeof(TAB) == 2) lds_sync(); else wave_sync();
        }
        if (PARTS > 1 && stage) {
            __syncthreads();
            const uint32_t *st = stage + (q0 / (64u * (uint32_t)AHEAD) & 1u) * 64u * (uint32_t)AHEAD;
            for (uint32_t v = part; v < (uint32_t)AHEAD; v += (uint32_t)PARTS) {
                const uint32_t p = q0 + 64u * v + (uint32_t)lane;
                if (p < npos) prev[p] = st[64u * v + (uint32_t)lane];
            }
        }
    }
}

/* blocks per workgroup of the two chain kernels (round 6, measured with the parse kernels' change of shape, gpurun_out/r6j: the
 * table-in-memory kernel is the same with one or four waves per workgroup, 4.0 - 4.2 ms; the LDS-table kernel with two waves and
 * two 68 KiB tables per workgroup is SLOWER than with two one-wave workgroups per CU, 6.3 - 6.7 against 5.7 ms) */
#ifndef K4_HC_CHAIN_WAVES
#define K4_HC_CHAIN_WAVES 4
#endif
#ifndef K4_HC_CHAIN_LDS_WAVES
#define K4_HC_CHAIN_LDS_WAVES 1
#endif
constexpr int HC_CHAIN_WAVES_PER_WG = K4_HC_CHAIN_WAVES, HC_CHAIN_LDS_WAVES_PER_WG = K4_HC_CHAIN_LDS_WAVES;
__global__ __launch_bounds__(64 * HC_CHAIN_WAVES_PER_WG) void k4_hc_chain_kernel(HcArgs a)
{
    __shared__ uint32_t seen_all[HC_CHAIN_WAVES_PER_WG][(1u << HC_HASH_LOG) / 32u];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    uint32_t *seen = seen_all[wave];
    const long long slot = (long long)blockIdx.x * HC_CHAIN_WAVES_PER_WG + (long long)wave;
    if (slot >= a.nChain) return;
    const long long b = slot + (long long)a.blockBase;
    const int len = a.srcLen[b];
    if (len < MFLIMIT + 1 || !hc_scratch_ok(a)) return;    /* no search happens (LL64.high.cs:549) */
    for (uint32_t k = (uint32_t)lane; k < (1u << HC_HASH_LOG) / 32u; k += 64u) seen[k] = 0u;
    lds_sync();
    hc_chain_block<uint32_t>(a.src + a.srcOff[b], (uint32_t)len, a.hash + (size_t)b * (1u << HC_HASH_LOG), (uint32_t *)(a.work + a.workOff[b]), seen, lane);
}

/* every block of the launch is at most 64 KiB long (the host knows the longest): tables in LDS, no table memory touched */
__global__ __launch_bounds__(64 * HC_CHAIN_LDS_WAVES_PER_WG) void k4_hc_chain_lds_kernel(HcArgs a)
{
    __shared__ uint32_t seen_all[HC_CHAIN_LDS_WAVES_PER_WG][(1u << HC_HASH_LOG) / 32u];
    __shared__ uint16_t tab_all[HC_CHAIN_LDS_WAVES_PER_WG][1u << HC_HASH_LOG];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    uint32_t *seen = seen_all[wave];
    uint16_t *tab = tab_all[wave];
    const long long slot = (long long)blockIdx.x * HC_CHAIN_LDS_WAVES_PER_WG + (long long)wave;
    if (slot >= a.nChain) return;                          /* (this kernel's share of the launch chunk: blocks blockBase .. blockBase + nChain) */
    const long long b = slot + (long long)a.blockBase;
    const int len = a.srcLen[b];
    if (len < MFLIMIT + 1 || len > 65536 || !hc_scratch_ok(a)) return;
    for (uint32_t k = (uint32_t)lane; k < (1u << HC_HASH_LOG) / 32u; k += 64u) seen[k] = 0u;
    for (uint32_t k = (uint32_t)lane; k < (1u << HC_HASH_LOG) / 2u; k += 64u) ((uint32_t *)tab)[k] = 0u;
    lds_sync();
    hc_chain_block<uint16_t>(a.src + a.srcOff[b], (uint32_t)len, tab, (uint32_t *)(a.work + a.workOff[b]), seen, lane);
}

/* Round 6: HC_CHAIN_PARTS waves per block, each with its share of the hash values (hc_chain_block<.., PARTS>): per wave 32768 / PARTS
 * slots of 16 bits and as many bits, 68 KiB of LDS per block as before but several waves working in it instead of one -- two blocks
 * per CU where k4_hc_chain_lds_kernel had two waves (its step is a chain of LDS round trips with nothing to hide them behind).  Every
 * block of the launch is at most 64 KiB long.  Measured on the bench batch (profiles/r6_hc_ab.txt): one wave per block 5.7 ms (beside
 * the memory-table kernel with a third of the blocks), 16 / 8 / 4 parts 4.27 / 4.07 / 4.23 ms, with prev[] through the LDS stage 4.13 /
 * 3.74 / -; the sources asked for 4 / 8 / 16 steps ahead: no difference.  With sixteen parts the vector port is 83 % busy (every wave
 * hashes every position: 32 vector instructions per step), with fewer the step's two LDS round trips are what a wave waits for. */
#ifndef K4_HC_CHAIN_PARTS
#define K4_HC_CHAIN_PARTS 8
#endif
constexpr int HC_CHAIN_PARTS = K4_HC_CHAIN_PARTS;
__global__ __launch_bounds__(64 * HC_CHAIN_PARTS) void k4_hc_chain_part_kernel(HcArgs a)
{
    __shared__ uint32_t seen_all[HC_CHAIN_PARTS][(1u << HC_HASH_LOG) / HC_CHAIN_PARTS / 32u];
    __shared__ uint16_t tab_all[HC_CHAIN_PARTS][(1u << HC_HASH_LOG) / HC_CHAIN_PARTS];
    __shared__ uint32_t stage[2 * 64 * K4_HC_CHAIN_AHEAD];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    uint32_t *seen = seen_all[wave];
    uint16_t *tab = tab_all[wave];
    const long long b = (long long)blockIdx.x + (long long)a.blockBase;
    if ((long long)blockIdx.x >= a.nChain) return;
    const int len = a.srcLen[b];
    if (len < MFLIMIT + 1 || len > 65536 || !hc_scratch_ok(a)) return;
    for (uint32_t k = (uint32_t)lane; k < (1u << HC_HASH_LOG) / HC_CHAIN_PARTS / 32u; k += 64u) seen[k] = 0u;
    for (uint32_t k = (uint32_t)lane; k < (1u << HC_HASH_LOG) / HC_CHAIN_PARTS / 2u; k += 64u) ((uint32_t *)tab)[k] = 0u;
    lds_sync();
#ifdef K4_HC_CHAIN_NO_STAGE
    hc_chain_block<uint16_t, HC_CHAIN_PARTS, K4_HC_CHAIN_AHEAD>(a.src + a.srcOff[b], (uint32_t)len, tab, (uint32_t *)(a.work + a.workOff[b]), seen, lane, wave);
#else
    hc_chain_block<uint16_t, HC_CHAIN_PARTS, K4_HC_CHAIN_AHEAD>(a.src + a.srcOff[b], (uint32_t)len, tab, (uint32_t *)(a.work + a.workOff[b]), seen, lane, wave, stage);
#endif
}

/* ---- kernel 1b: candidates + forward lengths, every position independently ------------------ */
/* One 16-byte record per position (round 6; 32 bytes in three arrays before):
 *   x = d0 | d1 << 16, y = d2 | d3 << 16   the first four chain candidates as distances p - c (1 .. 65535; 0: the chain has ended --
 *                                          no earlier position with this hash, a chain step of 65535 or more, LL.high.cs:114, or a
 *                                          candidate more than 65535 back, below lowestMatchIndex, LL64.high.cs:87-88)
 *   z = fl0 | fl1 << 8 | fl2 << 16 | fl3 << 24   forward match lengths (0: the four bytes differ), exact below HC_FLEN_CAP
 *   w = bl0 | ...                                equal bytes before the two positions, exact below HC_BLEN_CAP */
constexpr int HC_CAND_POS_PER_WG = 1024;

__device__ __forceinline__ uint32_t hc_rec_dist(const uint4 &r, int k) { return k == 0 ? r.x & 0xffffu : k == 1 ? r.x >> 16 : k == 2 ? r.y & 0xffffu : r.y >> 16; }

/* Where k4_hc_cand_*'s bytes come from: the block in memory, or (blocks of at most 64 KiB, round 6) a copy of it in LDS -- a load at a
 * byte address of its own per lane costs a CU 90-180 cycles from the caches (scripts/ubench/gather_rate.hip), the same bytes as
 * aligned dwords out of LDS and a funnel shift a fraction of that. */
struct HcSrcMem {
    const uint8_t *s;
    __device__ __forceinline__ uint32_t byte(uint32_t off) const { return s[off]; }
    __device__ __forceinline__ uint32_t ld32(uint32_t off) const { return ld32u(s + off); }
    __device__ __forceinline__ uint64_t ld64(uint32_t off) const { return ld64u(s + off); }
    __device__ __forceinline__ U128u ld128(uint32_t off) const { return ld128u(s + off); }
};
struct HcSrcLds {
    const uint32_t *l;           /* LDS: byte i of the block is byte i of this array; readable up to the next multiple of 4 behind offset + 16 */
    __device__ __forceinline__ static uint32_t fun(uint32_t lo, uint32_t hi, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh); }
    __device__ __forceinline__ uint32_t byte(uint32_t off) const { return (l[off >> 2] >> (8u * (off & 3u))) & 0xffu; }
    __device__ __forceinline__ uint32_t ld32(uint32_t off) const { const uint32_t q = off >> 2, sh = 8u * (off & 3u); return fun(l[q], l[q + 1u], sh); }
    __device__ __forceinline__ uint64_t ld64(uint32_t off) const
    {
        const uint32_t q = off >> 2, sh = 8u * (off & 3u);
        const uint32_t a0 = l[q], a1 = l[q + 1u], a2 = l[q + 2u];
        return ((uint64_t)fun(a1, a2, sh) << 32) | fun(a0, a1, sh);
    }
    __device__ __forceinline__ U128u ld128(uint32_t off) const
    {
        const uint32_t q = off >> 2, sh = 8u * (off & 3u);
        const uint32_t a0 = l[q], a1 = l[q + 1u], a2 = l[q + 2u], a3 = l[q + 3u], a4 = l[q + 4u];
        U128u r;
        r.v[0] = fun(a0, a1, sh); r.v[1] = fun(a1, a2, sh); r.v[2] = fun(a2, a3, sh); r.v[3] = fun(a3, a4, sh);
        return r;
    }
};

/* equal bytes from a / b on, at most n (bytewise: block edges only) */
template <typename SRC> __device__ __forceinline__ uint32_t hc_count_fwd_bytes(const SRC &s, uint32_t a, uint32_t b, uint32_t n) { uint32_t i = 0; while (i < n && s.byte(a + i) == s.byte(b + i)) i++; return i; }
/* equal bytes before a / b, at most n */
template <typename SRC> __device__ __forceinline__ uint32_t hc_count_back_bytes(const SRC &s, uint32_t a, uint32_t b, uint32_t n) { uint32_t i = 0; while (i < n && s.byte(a - 1u - i) == s.byte(b - 1u - i)) i++; return i; }

/* One position's record from its four chain candidates c[] (HC_NONE: the chain has ended): the distances, per candidate the forward match
 * length a search at p may use (LL64.high.cs:87-88 lowest, :120 the 4-byte test, :126 LZ4_count up to matchlimit, capped at HC_FLEN_CAP)
 * and the equal bytes before the two positions (LZ4HC_countBack without its limits, capped at HC_BLEN_CAP).
 * A candidate is ONE 16-byte load: the sixteen bytes from four before it -- four bytes backward, the four of the :120 test, eight
 * forward -- where rounds 1-5 made three (4 + 8 + 8 bytes).  More only for the candidates whose four bytes backward or eight forward are
 * all equal.  The positions at a block's edges (the first four, the last eleven) count bytewise. */
template <typename SRC>
__device__ __forceinline__ uint4 hc_cand_record(const SRC &src, const uint32_t U, const uint32_t p, const uint32_t (&c)[4])
{
    const uint32_t matchlimit = U - LASTLITERALS;
    const uint32_t maxn = p + MINMATCH < matchlimit ? matchlimit - (p + MINMATCH) : 0u;
    const uint32_t lim = maxn < HC_FLEN_CAP - MINMATCH ? maxn : HC_FLEN_CAP - MINMATCH;
    const bool inner = p >= 4u && p + 12u <= U;          /* the sixteen bytes around p, and around every candidate from 4 on, are the block's */
    uint32_t o0 = 0, o1, o2 = 0, o3 = 0;                 /* bytes p-4 .. p-1, p .. p+3, p+4 .. p+11 */
    if (inner) { const U128u v = src.ld128(p - 4u); o0 = v.v[0]; o1 = v.v[1]; o2 = v.v[2]; o3 = v.v[3]; }
    else o1 = src.ld32(p);
    uint32_t d[4], fl[4], bl[4], blim[4];
    bool chain_ok = true, fo[4], bo[4];                  /* fo / bo: the forward / backward count of candidate k is not finished yet */
    U128u cv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        chain_ok = chain_ok && c[k] != HC_NONE && p - c[k] <= (uint32_t)DISTANCE_MAX;
        d[k] = chain_ok ? p - c[k] : 0u;
        cv[k].v[0] = 0u; cv[k].v[1] = 0u; cv[k].v[2] = 0u; cv[k].v[3] = 0u;
        if (chain_ok && inner && c[k] >= 4u) cv[k] = src.ld128(c[k] - 4u);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t f = 0, bk = 0;
        const uint32_t cc = c[k];
        fo[k] = false; bo[k] = false;
        blim[k] = cc < HC_BLEN_CAP ? cc : HC_BLEN_CAP;                         /* cc < p */
        if (d[k] != 0u) {
            if (inner && cc >= 4u) {
                if (cv[k].v[1] == o1) {
                    if (lim >= 8u) {
                        const uint64_t x = (((uint64_t)(o3 ^ cv[k].v[3])) << 32) | (uint64_t)(o2 ^ cv[k].v[2]);
                        if (x) f = MINMATCH + ((uint32_t)(__ffsll((unsigned long long)x) - 1) >> 3);
                        else { f = MINMATCH + 8u; fo[k] = lim > 8u; }
                    } else {
                        f = MINMATCH + hc_count_fwd_bytes(src, p + MINMATCH, cc + MINMATCH, lim);
                    }
                    const uint32_t xb = o0 ^ cv[k].v[0];                       /* blim >= 4 */
                    if (xb) bk = (uint32_t)__clz((int)xb) >> 3;
                    else { bk = 4u; bo[k] = blim[k] > 4u; }
                }
            } else if (src.ld32(cc) == o1) {
                f = MINMATCH + hc_count_fwd_bytes(src, p + MINMATCH, cc + MINMATCH, lim);
                bk = hc_count_back_bytes(src, p, cc, blim[k]);
            }
        }
        fl[k] = f;
        bl[k] = bk;
    }
    /* The counts that are not finished go on TOGETHER, eight bytes per step and candidate, forward and backward in one step: all
     * of a step's loads are issued before the first is looked at, so a position waits for memory once per step -- at most four
     * times (forward 8 / 16 / 24 bytes in, backward 4 / 12 / 20 / 28) -- where one candidate after the other, forward and then
     * backward, waited up to two dozen times.  What is left below eight bytes is one more 8-byte compare that overlaps the bytes
     * already counted (shifted out), not a byte loop. */
    for (uint32_t t = 0; t < 4u; t++) {
        if (!(fo[0] || fo[1] || fo[2] || fo[3] || bo[0] || bo[1] || bo[2] || bo[3])) break;      /* (per lane: the wave goes on while a lane has something open) */
        const uint32_t i = 8u + 8u * t, bb = 4u + 8u * t;
        const uint32_t rf = lim > i ? lim - i : 0u;                           /* (> 0 for a candidate that is open) */
        const uint32_t fa = rf >= 8u ? i : lim - 8u;
        uint64_t yf[4], yb[4];
        uint32_t rb[4];
        const uint64_t own_f = (fo[0] || fo[1] || fo[2] || fo[3]) ? src.ld64(p + MINMATCH + fa) : 0ull;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            yf[k] = fo[k] ? own_f ^ src.ld64(c[k] + MINMATCH + fa) : 0ull;
            rb[k] = blim[k] - bb;
            const uint32_t ba = rb[k] >= 8u ? bb + 8u : blim[k];
            yb[k] = bo[k] ? src.ld64(p - ba) ^ src.ld64(c[k] - ba) : 0ull;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (fo[k]) {
                if (rf >= 8u) {
                    if (yf[k]) { fl[k] += (uint32_t)(__ffsll((unsigned long long)yf[k]) - 1) >> 3; fo[k] = false; }
                    else { fl[k] += 8u; if (rf == 8u) fo[k] = false; }
                } else {
                    const uint64_t y = yf[k] >> (8u * (8u - rf));
                    fl[k] += y ? (uint32_t)(__ffsll((unsigned long long)y) - 1) >> 3 : rf;
                    fo[k] = false;
                }
            }
            if (bo[k]) {
                if (rb[k] >= 8u) {
                    if (yb[k]) { bl[k] += (uint32_t)__clzll((long long)yb[k]) >> 3; bo[k] = false; }
                    else { bl[k] += 8u; if (rb[k] == 8u) bo[k] = false; }
                } else {
                    const uint64_t y = yb[k] << (8u * (8u - rb[k]));
                    bl[k] += y ? (uint32_t)__clzll((long long)y) >> 3 : rb[k];
                    bo[k] = false;
                }
            }
        }
    }
    return make_uint4(d[0] | (d[1] << 16), d[2] | (d[3] << 16), fl[0] | (fl[1] << 8) | (fl[2] << 16) | (fl[3] << 24),
                      bl[0] | (bl[1] << 8) | (bl[2] << 16) | (bl[3] << 24));
}

#ifndef K4_HC_CAND_ATTR
#define K4_HC_CAND_ATTR
#endif
__global__ __launch_bounds__(256) K4_HC_CAND_ATTR void k4_hc_cand_kernel(HcArgs a)
{
    /* Which block, which 1024 positions of it.  A workgroup's loads go anywhere in the 64 KiB before its positions (prev[] and the
     * candidates' bytes), so the chunks of ONE block should run at the same time and behind the same L2: workgroups go to the
     * chip's eight XCDs in turn (workgroup w to XCD w % 8), so workgroup w takes block 8 * (w / (8 * chunks)) + w % 8 and chunk
     * (w / 8) % chunks -- every XCD works its way through one block of each group of eight, four or so of them in flight, ~1.3 MiB
     * of sources and prev[] against 4 MiB of L2.  (Rounds 1-5 ran chunk 0 of every block of the launch, then chunk 1, ...: the two
     * thousand workgroups in flight touched two thousand blocks, 650 MB, and every load of a candidate came from memory.) */
    const uint32_t chunks = a.candChunks;
    const uint32_t grp = blockIdx.x / (8u * chunks), rem = blockIdx.x % (8u * chunks);
    const long long b = (long long)grp * 8 + (long long)(rem & 7u);
    if (b >= a.n) return;
    const int len = a.srcLen[b];
    if (len < MFLIMIT + 1 || !hc_scratch_ok(a)) return;
    const uint32_t U = (uint32_t)len;
    const uint32_t npos = U - 3u;
    const uint32_t first = a.posBase + (rem >> 3) * (uint32_t)HC_CAND_POS_PER_WG;
    if (first >= npos) return;
    const HcSrcMem src{a.src + a.srcOff[b]};
    const uint32_t *prev = (const uint32_t *)(a.work + a.workOff[b]);
    uint4 *rec = (uint4 *)(prev + ((U + 3u) & ~3u));
    /* the first four chain candidates (a chain step of 65535 or more ends the walk: LL.high.cs:114 caps the delta, and such a candidate is
     * below lowestMatchIndex) */
    for (int j = 0; j < HC_CAND_POS_PER_WG / 256; j++) {
        const uint32_t p = first + threadIdx.x + 256u * (uint32_t)j;
        if (p >= npos) continue;
        uint32_t c[4];
        c[0] = prev[p];
#pragma unroll
        for (int k = 1; k < 4; k++) {
            const uint32_t q = c[k - 1] != HC_NONE ? prev[c[k - 1]] : HC_NONE;
            c[k] = (q != HC_NONE && c[k - 1] - q < (uint32_t)DISTANCE_MAX) ? q : HC_NONE;
        }
        rec[p] = hc_cand_record(src, U, p, c);
    }
}

/* Blocks of at most 64 KiB (round 6): the same records out of LDS, in two kernels with nothing but whole lines between them and memory.
 *   k4_hc_walk_lds_kernel   one workgroup of 16 waves per block, prev[] as 16-bit positions in 128 KiB of LDS: the three dependent
 *                           look-ups of a position's chain are LDS reads; the four distances go to the record's first two words.
 *   k4_hc_cand_lds_kernel   one workgroup of 16 waves per block, the block's bytes in 64 KiB of LDS (two blocks per CU): reads the
 *                           distances back, the candidates' bytes are aligned dwords out of LDS and a funnel shift (HcSrcLds). */
constexpr int HC_LDS_WAVES = 16;
__global__ __launch_bounds__(64 * HC_LDS_WAVES) void k4_hc_walk_lds_kernel(HcArgs a)
{
    __shared__ uint16_t pv[65536];
    const long long b = (long long)blockIdx.x;
    const int len = a.srcLen[b];
    if (len < MFLIMIT + 1 || len > 65536 || !hc_scratch_ok(a)) return;
    const uint32_t U = (uint32_t)len, npos = U - 3u;
    const uint32_t *prev = (const uint32_t *)(a.work + a.workOff[b]);
    uint4 *rec = (uint4 *)(prev + ((U + 3u) & ~3u));
    constexpr uint32_t T = 64u * HC_LDS_WAVES;
    /* (eight loads in flight per thread: one at a time, a block's 256 KiB of prev[] would be 64 trips to memory one after the other) */
    for (uint32_t base = threadIdx.x; base < npos; base += 8u * T) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) v[u] = base + u * T < npos ? prev[base + u * T] : HC_NONE;
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) if (base + u * T < npos) pv[base + u * T] = (uint16_t)v[u];      /* (HC_NONE: 0xffff, which is no position of such a block) */
    }
    __syncthreads();
    /* four positions per thread together: their three dependent look-ups each are LDS round trips that overlap */
    for (uint32_t base = threadIdx.x; base < npos; base += 4u * T) {
        uint32_t c[4], d01[4], d23[4];
        bool ok[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) { const uint32_t p = base + u * T; c[u] = p < npos ? pv[p] : 0xffffu; ok[u] = true; d01[u] = 0u; d23[u] = 0u; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                const uint32_t p = base + u * T;
                ok[u] = ok[u] && c[u] != 0xffffu;                /* (steps and distances of 65535 and more do not exist in a block of 64 KiB) */
                const uint32_t d = ok[u] ? p - c[u] : 0u;
                if (k < 2) d01[u] |= d << (16 * k); else d23[u] |= d << (16 * (k - 2));
                if (k < 3) c[u] = ok[u] ? pv[c[u]] : 0xffffu;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) { const uint32_t p = base + u * T; if (p < npos) rec[p] = make_uint4(d01[u], d23[u], 0u, 0u); }
    }
}

#ifndef K4_HC_CAND_LDS_ATTR
#define K4_HC_CAND_LDS_ATTR
#endif
#ifndef K4_HC_CAND_LDS_WAVES
#define K4_HC_CAND_LDS_WAVES 12
#endif
constexpr int HC_CAND_LDS_WAVES = K4_HC_CAND_LDS_WAVES;      /* (two workgroups per CU by their LDS: with ~80 VGPRs twelve waves each are what fits) */
__global__ __launch_bounds__(64 * HC_CAND_LDS_WAVES) K4_HC_CAND_LDS_ATTR void k4_hc_cand_lds_kernel(HcArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t ls[65536 / 4 + 8];
    const long long b = (long long)blockIdx.x;
    const int len = a.srcLen[b];
    if (len < MFLIMIT + 1 || len > 65536 || !hc_scratch_ok(a)) return;
    const uint32_t U = (uint32_t)len, npos = U - 3u;
    const uint8_t *g = a.src + a.srcOff[b];
    const uint32_t *prev = (const uint32_t *)(a.work + a.workOff[b]);
    uint4 *rec = (uint4 *)(prev + ((U + 3u) & ~3u));
    for (uint32_t v = threadIdx.x; 16u * v < U; v += 64u * HC_CAND_LDS_WAVES) {      /* the block, sixteen bytes per thread; its last bytes one by one */
        if (16u * v + 16u <= U) { const U128u w = ld128u(g + 16u * v); ls[4u * v] = w.v[0]; ls[4u * v + 1u] = w.v[1]; ls[4u * v + 2u] = w.v[2]; ls[4u * v + 3u] = w.v[3]; }
        else for (uint32_t q = 0; q < 4u; q++) { uint32_t w = 0; for (uint32_t r = 0; r < 4u; r++) if (16u * v + 4u * q + r < U) w |= (uint32_t)g[16u * v + 4u * q + r] << (8u * r); ls[4u * v + q] = w; }
    }
    if (threadIdx.x < 8u) ls[((U + 15u) & ~15u) / 4u + threadIdx.x] = 0u;       /* (read, never used: the dwords behind the last byte) */
    __syncthreads();
    const HcSrcLds src{ls};
    /* (the distances are read where they are used.  Asking for the next position's a step ahead -- `dn = rec[p + T]` at the top of the loop -- was
     * built and gave wrong records on the GPU for workgroups of 12 and 16 waves, the same ones every run, and right ones for 8 waves, for a build
     * limited to 64 VGPRs and under the emulator; a full s_waitcnt in front of its use changed nothing; with a trip count that is the same for every lane -- no lane leaves the loop before
     * another -- the same read-ahead gives the right records.  Not faster either way: gone.) */
    for (uint32_t p = threadIdx.x; p < npos; p += 64u * HC_CAND_LDS_WAVES) {
        const uint2 dd = ((const uint2 *)(rec + p))[0];
        uint32_t c[4];
        c[0] = (dd.x & 0xffffu) ? p - (dd.x & 0xffffu) : HC_NONE; c[1] = (dd.x >> 16) ? p - (dd.x >> 16) : HC_NONE;
        c[2] = (dd.y & 0xffffu) ? p - (dd.y & 0xffffu) : HC_NONE; c[3] = (dd.y >> 16) ? p - (dd.y >> 16) : HC_NONE;
        rec[p] = hc_cand_record(src, U, p, c);
    }
}

/* ---- kernel 2: parse -------------------------------------------------------------------- */

/* ---- level 9: pattern analysis (LL64.high.cs:208-337) ------------------------------------------
 * A candidate whose chain step is 1 sits inside a run of a repeating 1/2/4-byte pattern.  The reference
 * then measures the run around the candidate and jumps along it instead of walking it position by
 * position.  The walk becomes data dependent, so from the first such candidate on the search goes one
 * candidate at a time (wave-wide counts), exactly in the reference's order. */

/* bytes at src[from .. limit) that continue the 4-periodic `pattern` (phase 0 at `from`): LL64.high.cs:37-68 */
__device__ __forceinline__ uint32_t hc_count_pattern(const uint8_t *src, uint32_t from, uint32_t limit, uint32_t pattern, int lane)
{
    uint32_t done = 0;
    for (;;) {
        const uint32_t i = done + (uint32_t)lane;
        const bool eq = from + i < limit && src[from + i] == (uint8_t)(pattern >> (8u * (i & 3u)));
        const unsigned long long ne = ~__ballot(eq);
        const int run = ne ? ctz64(ne) : 64;
        done += (uint32_t)run;
        if (run < 64) return done;
    }
}

/* bytes before src[at] (down to position 0) that continue the pattern backwards: LL.high.cs:232-254 */
__device__ __forceinline__ uint32_t hc_reverse_count_pattern(const uint8_t *src, uint32_t at, uint32_t pattern, int lane)
{
    uint32_t done = 0;
    for (;;) {
        const uint32_t i = done + (uint32_t)lane;
        const bool eq = i < at && src[at - 1u - i] == (uint8_t)(pattern >> (8u * (3u - (i & 3u))));
        const unsigned long long ne = ~__ballot(eq);
        const int run = ne ? ctz64(ne) : 64;
        done += (uint32_t)run;
        if (run < 64) return done;
    }
}

struct HcMatch { int len; uint32_t mpos, spos; };          /* longest, *matchpos, *startpos */

/* the rest of a search, one candidate per step; entered right after candidate `mi` has been evaluated */
__device__ __forceinline__ void hc_search_serial(const uint8_t *src, const uint32_t *prev, uint32_t ip, uint32_t ilow, uint32_t matchlimit,
                                                 HcMatch &r, uint32_t mi, int attempts, int lane)
{
    const uint32_t pattern = uni(ld32u(src + ip));
    const uint32_t lowest = ip > (uint32_t)DISTANCE_MAX ? ip - (uint32_t)DISTANCE_MAX : 0u;
    const uint32_t look_back = ip - ilow;
    int repeat = 0;                                          /* 0 untested, 1 not a repeating pattern, 2 confirmed */
    uint32_t src_pattern_length = 0;
    for (;;) {
        /* chain step of candidate mi (LL.high.cs:114 caps it at 65535; such a step ends the walk) */
        uint32_t pv = uni(prev[mi]);
        uint32_t delta = pv == HC_NONE || mi - pv >= (uint32_t)DISTANCE_MAX ? 0u : mi - pv;   /* 0: the chain ends */
        bool jumped = false;
        if (delta == 1u) {                                   /* :208 (matchChainPos == 0) */
            const uint32_t cidx = mi - 1u;
            if (repeat == 0) {
                if (((pattern & 0xFFFFu) == (pattern >> 16)) && ((pattern & 0xFFu) == (pattern >> 24))) {
                    repeat = 2;
                    src_pattern_length = hc_count_pattern(src, ip + MINMATCH, matchlimit, pattern, lane) + MINMATCH;
                } else {
                    repeat = 1;
                }
            }
            if (repeat == 2 && cidx >= lowest && uni(ld32u(src + cidx)) == pattern) {
                const uint32_t fwd = hc_count_pattern(src, cidx + MINMATCH, matchlimit, pattern, lane) + MINMATCH;
                uint32_t back_len = hc_reverse_count_pattern(src, cidx, pattern, lane);
                {
                    const uint32_t a = cidx - back_len;
                    back_len = cidx - (a > lowest ? a : lowest);
                }
                const uint32_t cur_seg = back_len + fwd;
                jumped = true;
                if (cur_seg >= src_pattern_length && fwd <= src_pattern_length) {
                    mi = cidx + fwd - src_pattern_length;    /* the spot in the run with exactly the source's run ahead */
                } else {
                    mi = cidx - back_len;                    /* start of the run */
                    if (look_back == 0u) {
                        const uint32_t max_ml = cur_seg < src_pattern_length ? cur_seg : src_pattern_length;
                        if ((uint32_t)r.len < max_ml) {
                            if (ip - mi > (uint32_t)DISTANCE_MAX) return;
                            r.len = (int)max_ml; r.mpos = mi; r.spos = ip;
                        }
                        pv = uni(prev[mi]);
                        if (pv == HC_NONE || mi - pv >= (uint32_t)DISTANCE_MAX) return;
                        mi = pv;
                    }
                }
            }
        }
        if (!jumped) {
            if (delta == 0u) return;
            mi -= delta;
        }
        if (mi < lowest || attempts == 0) return;
        attempts--;
        if (uni(ld32u(src + mi)) == pattern) {               /* :120-133 */
            const uint32_t fwd = wave_count(src + ip + MINMATCH, src + mi + MINMATCH, matchlimit - (ip + MINMATCH), lane);
            uint32_t back = 0;
            const uint32_t maxb = look_back < mi ? look_back : mi;
            while (back < maxb) {
                const uint32_t i = back + (uint32_t)lane;
                const bool eq = i < maxb && src[ip - 1u - i] == src[mi - 1u - i];
                const unsigned long long ne = ~__ballot(eq);
                const int run = ne ? ctz64(ne) : 64;
                back += (uint32_t)run;
                if (run < 64) break;
            }
            const int ml = (int)(MINMATCH + fwd + back);
            if (ml > r.len) { r.len = ml; r.mpos = mi - back; r.spos = ip - back; }
        }
    }
}

/*
 * LZ4HC_InsertAndGetWiderMatch (LL64.high.cs:70-383) at position ip with low limit ilow, best
 * length so far `longest` (matchpos / startpos of the caller stay untouched unless it is beaten).
 * Four candidates per step, 16 lanes each.
 */
__device__ __forceinline__ HcMatch hc_search(const uint8_t *src, const uint4 *cand, uint32_t ip, uint32_t ilow,
                                             uint32_t matchlimit, int longest, uint32_t mpos, uint32_t spos,
                                             int max_attempts, int lane, const uint32_t *prev = nullptr, bool pa = false)
{
    HcMatch r;
    r.len = longest; r.mpos = mpos; r.spos = spos;
    const uint32_t pattern = uni(ld32u(src + ip));
    const uint32_t lowest = ip > (uint32_t)DISTANCE_MAX ? ip - (uint32_t)DISTANCE_MAX : 0u;   /* :87-88 */
    const uint32_t look_back = ip - ilow;
    const int grp = lane >> 4, sub = lane & 15;
    uint32_t rec_at = ip;
    int attempts = max_attempts;
    bool first_record = true;
    while (attempts > 0) {
        const uint4 rec = cand[rec_at];
        const uint32_t dist = hc_rec_dist(rec, grp);        /* the record's distances are from rec_at */
        uint32_t c = dist ? rec_at - dist : HC_NONE;
        /* the first candidate of a search may be exactly 65535 back; later chain steps may not */
        bool ok = c != HC_NONE && c >= lowest && grp < attempts;
        if (!first_record && grp == 0 && ok) ok = rec_at - c < (uint32_t)DISTANCE_MAX;
        /* candidates are only reachable through their predecessors */
        const unsigned long long okm = __ballot(ok);
        const bool ok0 = (okm & 1ull) != 0, ok1 = ok0 && ((okm >> 16) & 1ull) != 0, ok2 = ok1 && ((okm >> 32) & 1ull) != 0,
                   ok3 = ok2 && ((okm >> 48) & 1ull) != 0;
        ok = grp == 0 ? ok0 : grp == 1 ? ok1 : grp == 2 ? ok2 : ok3;
        int nvalid = (ok0 ? 1 : 0) + (ok1 ? 1 : 0) + (ok2 ? 1 : 0) + (ok3 ? 1 : 0);
        if (nvalid == 0) break;
        /* level 9: the first candidate whose chain step is 1 ends the four-at-a-time walk */
        int pa_at = -1;
        if (pa) {
            const uint32_t nd = grp < 3 ? hc_rec_dist(rec, grp + 1) : 0u;
            uint32_t nx = nd ? rec_at - nd : HC_NONE;
            if (grp == 3 && ok) nx = prev[c];
            const unsigned long long pm = __ballot(ok && sub == 0 && nx != HC_NONE && c - nx == 1u);
            if (pm) {
                pa_at = ctz64(pm) >> 4;
                if (nvalid > pa_at + 1) nvalid = pa_at + 1;
                ok = ok && grp <= pa_at;
            }
        }
        if (!ok) c = 0;
        const bool seq_ok = ok && ld32u(src + c) == pattern;                      /* :120 */
        /* forward: bytes ip+4.. vs c+4.., 64 per step and group */
        uint32_t fwd = 0;
        {
            const uint32_t maxn = matchlimit - (ip + MINMATCH);
            uint32_t done = 0;
            bool open = seq_ok;
            for (;;) {
                const uint32_t i = done + 4u * (uint32_t)sub;
                uint32_t neq = 4u;
                if (open) {
                    neq = 0;
                    if (i < maxn) {
                        const uint32_t x = ld32u(src + ip + MINMATCH + i) ^ ld32u(src + c + MINMATCH + i);
                        const uint32_t avail = maxn - i < 4u ? maxn - i : 4u;
                        const uint32_t e = x ? (uint32_t)(__ffs(x) - 1) >> 3 : 4u;
                        neq = e < avail ? e : avail;
                    }
                }
                const unsigned long long nf = __ballot(open && neq != 4u);
                const uint32_t gm = (uint32_t)(nf >> (16 * grp)) & 0xffffu;     /* this group's not-full lanes */
                const int fl = gm ? __ffs(gm) - 1 : 0;
                const uint32_t nq = (uint32_t)__shfl((int)neq, (lane & 48) + fl);
                if (open) {
                    if (gm) { fwd = done + 4u * (uint32_t)fl + nq; open = false; }
                    else done += 64u;
                }
                if (!__ballot(open)) break;
            }
        }
        /* backward: ip-1.. vs c-1.., bounded by ilow and the block start (:124-125, LL.high.cs:216-230) */
        uint32_t back = 0;
        if (look_back) {
            const uint32_t maxb = look_back < c ? look_back : c;
            uint32_t done = 0;
            bool open = seq_ok && maxb != 0u;
            for (;;) {
                const uint32_t i = done + (uint32_t)sub;
                const bool eq = open && i < maxb && src[ip - 1u - i] == src[c - 1u - i];
                const unsigned long long ne = __ballot(open && !eq);
                const uint32_t gm = (uint32_t)(ne >> (16 * grp)) & 0xffffu;
                if (open) {
                    if (gm) { back = done + (uint32_t)(__ffs(gm) - 1); open = false; }
                    else done += 16u;
                }
                if (!__ballot(open)) break;
            }
        }
        const int ml = seq_ok ? (int)(MINMATCH + fwd + back) : 0;                /* matchLength - back, back <= 0 */
        /* chain order: a candidate wins only if it beats everything before it (:128-133) */
        const int ml0 = (int)__builtin_amdgcn_readlane((uint32_t)ml, 0), ml1 = (int)__builtin_amdgcn_readlane((uint32_t)ml, 16),
                  ml2 = (int)__builtin_amdgcn_readlane((uint32_t)ml, 32), ml3 = (int)__builtin_amdgcn_readlane((uint32_t)ml, 48);
        const uint32_t c0 = __builtin_amdgcn_readlane(c, 0), c1 = __builtin_amdgcn_readlane(c, 16), c2 = __builtin_amdgcn_readlane(c, 32),
                       c3 = __builtin_amdgcn_readlane(c, 48);
        const uint32_t b0 = __builtin_amdgcn_readlane(back, 0), b1 = __builtin_amdgcn_readlane(back, 16),
                       b2 = __builtin_amdgcn_readlane(back, 32), b3 = __builtin_amdgcn_readlane(back, 48);
        if (ml0 > r.len) { r.len = ml0; r.mpos = c0 - b0; r.spos = ip - b0; }
        if (nvalid > 1 && ml1 > r.len) { r.len = ml1; r.mpos = c1 - b1; r.spos = ip - b1; }
        if (nvalid > 2 && ml2 > r.len) { r.len = ml2; r.mpos = c2 - b2; r.spos = ip - b2; }
        if (nvalid > 3 && ml3 > r.len) { r.len = ml3; r.mpos = c3 - b3; r.spos = ip - b3; }
        attempts -= nvalid;
        if (pa_at >= 0) {
            hc_search_serial(src, prev, ip, ilow, matchlimit, r, pa_at == 0 ? c0 : pa_at == 1 ? c1 : pa_at == 2 ? c2 : c3, attempts, lane);
            break;
        }
        if (nvalid < 4) break;
        rec_at = c3;                                        /* the chain continues behind the last candidate */
        first_record = false;
    }
    return r;
}

/* one position's precomputed record (k4_hc_cand_kernel), wave-uniform */
struct HcRec { uint32_t d01, d23, fl, bl; };

__device__ __forceinline__ HcRec hc_load_rec(const uint4 *cand, uint32_t p)
{
    const uint4 rc = cand[p];
    HcRec r;
    r.d01 = uni(rc.x); r.d23 = uni(rc.y); r.fl = uni(rc.z); r.bl = uni(rc.w);
    return r;
}

/* The records of 128 consecutive positions, two per lane: [base, base + 64) in rc, the 64 after them in rc2.  The second set is
 * asked for when the parse enters the first one and has a window's worth of work to arrive in; the searches of the three-match
 * arbitration look ahead by a match length at most, so they find their records here. */
struct HcWindow { uint32_t base; bool valid; uint4 rc, rc2; };

__device__ __forceinline__ HcRec hc_get_rec(const HcWindow &w, const uint4 *cand, uint32_t p)
{
    if (w.valid && p - w.base < 64u) {
        const int l = (int)(p - w.base);
        HcRec r;
        r.d01 = __builtin_amdgcn_readlane(w.rc.x, l); r.d23 = __builtin_amdgcn_readlane(w.rc.y, l);
        r.fl = __builtin_amdgcn_readlane(w.rc.z, l); r.bl = __builtin_amdgcn_readlane(w.rc.w, l);
        return r;
    }
    if (w.valid && p - w.base < 128u) {
        const int l = (int)(p - w.base - 64u);
        HcRec r;
        r.d01 = __builtin_amdgcn_readlane(w.rc2.x, l); r.d23 = __builtin_amdgcn_readlane(w.rc2.y, l);
        r.fl = __builtin_amdgcn_readlane(w.rc2.z, l); r.bl = __builtin_amdgcn_readlane(w.rc2.w, l);
        return r;
    }
    return hc_load_rec(cand, p);
}

/*
 * The same search when the whole chain walk is one record (level 3: 4 attempts): match lengths
 * come from the precomputed record, nothing is compared.  Falls back to hc_search when a length
 * sits at its cap and could be longer.
 */
__device__ __forceinline__ HcMatch hc_search_l3(const uint8_t *src, const uint4 *cand, const HcRec &rec, uint32_t ip, uint32_t ilow,
                                                uint32_t matchlimit, int longest, uint32_t mpos, uint32_t spos, int lane)
{
    /* The four candidates in four lanes (lane & 3: every quad computes the same, lanes 0..3 are the ones that are read).  The parse is
     * wave-uniform code and the scalar port -- one instruction per four cycles and SIMD -- is what it uses up (4.2 G scalar against
     * 0.3 G vector instructions per launch of the bench batch, gpurun_out/r6x_pmc); as scalar code this search was ~90 of them. */
    const uint32_t k = (uint32_t)lane & 3u;
    /* (a bit select, not `k & 2 ? rec.d23 : rec.d01`: the compiler turns a select between two neighbouring fields into a store of both and an
     * indexed load -- through scratch memory or LDS, a round trip in every search) */
    const uint32_t dd = rec.d01 ^ ((rec.d01 ^ rec.d23) & (0u - ((k >> 1) & 1u)));
    const uint32_t d = (dd >> (16u * (k & 1u))) & 0xffffu;
    uint32_t l = (rec.fl >> (8u * k)) & 0xffu;
    const uint32_t s = (rec.bl >> (8u * k)) & 0xffu;
    /* (a candidate without a forward length -- the chain has ended, or its four bytes differ -- is never looked at: its position may be anything) */
    const uint32_t c = ip - d;
    const uint32_t look_back = ip - ilow;
    if (__ballot(l == HC_FLEN_CAP) & 0xfull) {
        /* a forward length at its cap: the count goes on from there, 16 lanes per candidate and 64 bytes per step -- one
         * trip to memory for the usual match, where the general search would start from the candidate records again */
        const int grp = lane >> 4, sub = lane & 15;
        const uint32_t cg = (uint32_t)__shfl((int)c, grp), lg = (uint32_t)__shfl((int)l, grp);
        const uint32_t maxn = matchlimit - (ip + MINMATCH);
        bool open = lg == HC_FLEN_CAP;
        uint32_t done = HC_FLEN_CAP - MINMATCH, fwd = lg;
        for (;;) {
            const uint32_t i = done + 4u * (uint32_t)sub;
            uint32_t neq = 4u;
            if (open) {
                neq = 0;
                if (i < maxn) {
                    const uint32_t x = ld32u(src + ip + MINMATCH + i) ^ ld32u(src + cg + MINMATCH + i);
                    const uint32_t avail = maxn - i < 4u ? maxn - i : 4u;
                    const uint32_t e = x ? (uint32_t)(__ffs(x) - 1) >> 3 : 4u;
                    neq = e < avail ? e : avail;
                }
            }
            const unsigned long long nf = __ballot(open && neq != 4u);
            const uint32_t gm = (uint32_t)(nf >> (16 * grp)) & 0xffffu;     /* this group's not-full lanes */
            const int fl = gm ? __ffs(gm) - 1 : 0;
            const uint32_t nq = (uint32_t)__shfl((int)neq, (lane & 48) + fl);
            if (open) {
                if (gm) { fwd = MINMATCH + done + 4u * (uint32_t)fl + nq; open = false; }
                else done += 64u;
            }
            if (!__ballot(open)) break;
        }
        l = (uint32_t)__shfl((int)fwd, 16 * (int)k);
    }
    const uint32_t m = look_back < c ? look_back : c;       /* LZ4HC_countBack = min(equal bytes, ip - ilow, match - 0) */
    const uint32_t b = s < m ? s : m;
    if (__ballot(l != 0u && s == HC_BLEN_CAP && m > HC_BLEN_CAP) & 0xfull)
        return hc_search(src, cand, ip, ilow, matchlimit, longest, mpos, spos, 4, lane);
    /* the first candidate in chain order that has the greatest length wins, if that beats `longest` (:128-133) */
    uint32_t key = ((l ? l + b : 0u) << 2) | (3u - k);
    { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x111, 0xf, 0xf, true); key = key > o ? key : o; }    /* row_shr:1 */
    { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x112, 0xf, 0xf, true); key = key > o ? key : o; }    /* row_shr:2 */
    const uint32_t best = readlane_u32(key, 3);
    HcMatch r;
    r.len = longest; r.mpos = mpos; r.spos = spos;
    if ((int)(best >> 2) > longest) {
        const int kk = 3 - (int)(best & 3u);
        const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)b, kk);
        r.len = (int)(best >> 2);
        r.mpos = (uint32_t)__builtin_amdgcn_readlane((int)c, kk) - bb;
        r.spos = ip - bb;
    }
    return r;
}

/* LZ4HC_encodeSequence (LL64.high.cs:435-510); returns false on output overflow */
__device__ __forceinline__ bool hc_encode_sequence(const uint8_t *src, uint8_t *dst, uint32_t &ip, int64_t &op, uint32_t &anchor,
                                                   int match_length, uint32_t match, bool limited, int64_t oend, int lane,
                                                   uint32_t pf_anchor = HC_NONE, uint8_t pf_byte = 0)
{
    const uint32_t token_pos = (uint32_t)op;
    op++;
    uint32_t length = ip - anchor;
    if (limited && op + (int64_t)(length / 255u) + length + (2 + 1 + LASTLITERALS) > oend) return false;
    uint32_t token;
    if (length >= (uint32_t)RUN_MASK) {
        uint32_t len = length - RUN_MASK;
        token = (uint32_t)RUN_MASK << ML_BITS;
        const uint32_t nb = len / 255u;
        wave_fill(dst + op, 255, nb, lane);
        if (lane == 0) dst[op + nb] = (uint8_t)(len - nb * 255u);
        op += nb + 1u;
    } else {
        token = length << ML_BITS;
    }
    if (anchor == pf_anchor && length <= 64u) {            /* literal bytes were requested ahead of time */
        if ((uint32_t)lane < length) dst[op + lane] = pf_byte;
    } else {
        wave_copy(dst + op, src + anchor, length, lane);
    }
    op += length;
    if (lane == 0) {
        const uint32_t off = ip - match;
        dst[op] = (uint8_t)off;
        dst[op + 1] = (uint8_t)(off >> 8);
    }
    op += 2;
    length = (uint32_t)match_length - MINMATCH;
    if (limited && op + (int64_t)(length / 255u) + (1 + LASTLITERALS) > oend) return false;
    if (length >= (uint32_t)ML_MASK) {
        token += ML_MASK;
        length -= ML_MASK;
        const uint32_t nb = length / 255u;                 /* the 510-stepped loop writes the same bytes */
        wave_fill(dst + op, 255, nb, lane);
        if (lane == 0) dst[op + nb] = (uint8_t)(length - nb * 255u);
        op += nb + 1u;
    } else {
        token += length;
    }
    if (lane == 0) dst[token_pos] = (uint8_t)token;
    ip += (uint32_t)match_length;
    anchor = ip;
    op = (int64_t)uni((uint32_t)op);
    return true;
}

/* ---- levels 10..12: the optimal parser (LL64.high.cs:802-1122) ------------------------------------------
 * LZ4HC_FindLongerMatch = InsertAndGetWiderMatch with pattern analysis AND chain swap, no look-back: the walk
 * is data dependent from the first improvement on, so it goes one candidate at a time.  The price table of the
 * dynamic program (4096 + 3 positions) lives in LDS; its inner loops over match lengths run one length per lane. */
constexpr int HC_OPT_NUM = 1 << 12;                        /* LL.types.cs:75 */
constexpr int HC_OPT_ENTRIES = HC_OPT_NUM + 3;             /* + TRAILING_LITERALS */
constexpr int HC_OPT_LDS_DWORDS = HC_OPT_ENTRIES * 3 + 4;  /* price u32, litlen u32, (mlen | off << 16) u32 */

__device__ __forceinline__ uint32_t hc_delta(const uint32_t *prev, uint32_t p)       /* DELTANEXTU16(chainTable, p) */
{
    const uint32_t pv = uni(prev[p]);
    return pv == HC_NONE || p - pv >= (uint32_t)DISTANCE_MAX ? (uint32_t)DISTANCE_MAX : p - pv;
}

struct HcOptMatch { int len; int off; };

__device__ __forceinline__ HcOptMatch hc_find_longer_match(const uint8_t *src, const uint32_t *prev, uint32_t ip, uint32_t matchlimit,
                                                           int min_len, int nb_searches, int lane)
{
    HcOptMatch r; r.len = 0; r.off = 0;
    const uint32_t pattern = uni(ld32u(src + ip));
    const uint32_t lowest = ip > (uint32_t)DISTANCE_MAX ? ip - (uint32_t)DISTANCE_MAX : 0u;
    int longest = min_len;
    uint32_t mpos = 0;
    int attempts = nb_searches;
    int repeat = 0;
    uint32_t src_pattern_length = 0, chain_pos = 0;        /* matchChainPos (:95) */
    uint32_t mi = uni(prev[ip]);
    if (mi == HC_NONE) return r;
    while (mi >= lowest && attempts != 0) {
        attempts--;
        bool improved = false;
        if (uni(ld32u(src + mi)) == pattern) {              /* :120-133; the 16-bit pre-test only filters non-improvements */
            const int ml = (int)(MINMATCH + wave_count(src + ip + MINMATCH, src + mi + MINMATCH, matchlimit - (ip + MINMATCH), lane));
            if (ml > longest) { longest = ml; mpos = mi; improved = true; }
        }
        if (improved && mi + (uint32_t)longest <= ip) {     /* :172-206 chain swap */
            uint32_t dist_next = 1;
            const int end = longest - MINMATCH + 1;
            int step = 1, accel = 16;
            for (int pos = 0; pos < end; pos += step) {
                const uint32_t cd = hc_delta(prev, mi + (uint32_t)pos);
                step = accel++ >> 4;
                if (cd > dist_next) { dist_next = cd; chain_pos = (uint32_t)pos; accel = 16; }
            }
            if (dist_next > 1u) {
                if (dist_next > mi) break;
                mi -= dist_next;
                continue;
            }
        }
        {
            const uint32_t d0 = hc_delta(prev, mi);
            if (d0 == 1u && chain_pos == 0u) {              /* :208-337 pattern analysis */
                const uint32_t cidx = mi - 1u;
                if (repeat == 0) {
                    if (((pattern & 0xFFFFu) == (pattern >> 16)) && ((pattern & 0xFFu) == (pattern >> 24))) {
                        repeat = 2;
                        src_pattern_length = hc_count_pattern(src, ip + MINMATCH, matchlimit, pattern, lane) + MINMATCH;
                    } else {
                        repeat = 1;
                    }
                }
                if (repeat == 2 && cidx >= lowest && uni(ld32u(src + cidx)) == pattern) {
                    const uint32_t fwd = hc_count_pattern(src, cidx + MINMATCH, matchlimit, pattern, lane) + MINMATCH;
                    uint32_t back_len = hc_reverse_count_pattern(src, cidx, pattern, lane);
                    {
                        const uint32_t a = cidx - back_len;
                        back_len = cidx - (a > lowest ? a : lowest);
                    }
                    const uint32_t cur_seg = back_len + fwd;
                    if (cur_seg >= src_pattern_length && fwd <= src_pattern_length) {
                        mi = cidx + fwd - src_pattern_length;
                    } else {
                        mi = cidx - back_len;
                        const uint32_t max_ml = cur_seg < src_pattern_length ? cur_seg : src_pattern_length;
                        if ((uint32_t)longest < max_ml) {
                            if (ip - mi > (uint32_t)DISTANCE_MAX) break;
                            longest = (int)max_ml; mpos = mi;
                        }
                        const uint32_t d = hc_delta(prev, mi);
                        if (d > mi) break;
                        mi -= d;
                    }
                    continue;
                }
            }
        }
        {
            const uint32_t d = hc_delta(prev, mi + chain_pos);   /* :340 follow current chain */
            if (d > mi) break;
            mi -= d;
        }
    }
    if (longest <= min_len) return r;
    r.len = longest;
    r.off = (int)(ip - mpos);
    return r;
}

__device__ __forceinline__ int hc_literals_price(int litlen)                      /* LL.high.cs:267-274 */
{
    return litlen + (litlen >= (int)RUN_MASK ? 1 + (litlen - (int)RUN_MASK) / 255 : 0);
}
__device__ __forceinline__ int hc_sequence_price(int litlen, int mlen)            /* LL.high.cs:277-287 */
{
    return 3 + hc_literals_price(litlen) + (mlen >= (int)(ML_MASK + MINMATCH) ? 1 + (mlen - (int)(ML_MASK + MINMATCH)) / 255 : 0);
}

/* LZ4HC_compress_optimal for one block (limitedOutput / notLimited); returns bytes written, 0 = overflow */
__device__ __forceinline__ int hc_parse_block_opt(const uint8_t *src, int src_len, uint8_t *dst, int dst_cap, int level,
                                                  const uint32_t *prev, uint32_t *lds, int lane)
{
    if ((uint32_t)src_len > (uint32_t)MAX_INPUT_SIZE) return 0;
    const bool limited = dst_cap < compress_bound(src_len);
    const int64_t oend = dst_cap;
    const uint32_t U = (uint32_t)src_len;
    const int nb_searches = level <= 10 ? 96 : level == 11 ? 512 : 16384;         /* clTable :1134-1136; above 12: as 12 (:1160) */
    int sufficient_len = level <= 10 ? 64 : level == 11 ? 128 : HC_OPT_NUM;
    if (sufficient_len >= HC_OPT_NUM) sufficient_len = HC_OPT_NUM - 1;
    const bool full_update = level >= 12;
    uint32_t *o_price = lds, *o_litlen = lds + HC_OPT_ENTRIES, *o_ml_off = lds + 2 * HC_OPT_ENTRIES;
#define K4_OPT_SET(P, ML, OFF, LL, PR) do { o_price[(P)] = (uint32_t)(PR); o_litlen[(P)] = (uint32_t)(LL); o_ml_off[(P)] = (uint32_t)(ML) | ((uint32_t)(OFF) << 16); } while (0)
    uint32_t ip = 0, anchor = 0;
    int64_t op = 0;
    if (src_len >= MFLIMIT + 1) {
        const uint32_t mflimit = U - MFLIMIT, matchlimit = U - LASTLITERALS;
        while (ip <= mflimit) {
            const int llen = (int)(ip - anchor);
            const HcOptMatch first = hc_find_longer_match(src, prev, ip, matchlimit, MINMATCH - 1, nb_searches, lane);
            if (first.len == 0) { ip++; continue; }
            if (first.len > sufficient_len) {                       /* good enough: immediate encoding */
                if (!hc_encode_sequence(src, dst, ip, op, anchor, first.len, ip - (uint32_t)first.off, limited, oend, lane)) return 0;
                continue;
            }
            wave_sync();
            if (lane < MINMATCH) K4_OPT_SET(lane, 1, 0, llen + lane, hc_literals_price(llen + lane));
            for (int m = MINMATCH + lane; m <= first.len; m += 64) K4_OPT_SET(m, m, first.off, llen, hc_sequence_price(llen, m));
            int last_match_pos = first.len;
            wave_sync();
            {
                const uint32_t base_price = uni(o_price[last_match_pos]);
                if (lane >= 1 && lane <= 3) K4_OPT_SET(last_match_pos + lane, 1, 0, lane, base_price + (uint32_t)hc_literals_price(lane));
            }
            wave_sync();
            int best_mlen = 0, best_off = 0, cur;
            bool direct = false;
            for (cur = 1; cur < last_match_pos; cur++) {
                const uint32_t cur_ptr = ip + (uint32_t)cur;
                if (cur_ptr > mflimit) break;
                const int p_cur = (int)uni(o_price[cur]), p_next = (int)uni(o_price[cur + 1]);
                if (full_update) {
                    if (p_next <= p_cur && (int)uni(o_price[cur + MINMATCH]) < p_cur + 3) continue;
                } else {
                    if (p_next <= p_cur) continue;
                }
                const HcOptMatch nm = hc_find_longer_match(src, prev, cur_ptr, matchlimit, full_update ? MINMATCH - 1 : last_match_pos - cur,
                                                           nb_searches, lane);
                if (nm.len == 0) continue;
                if (nm.len > sufficient_len || nm.len + cur >= HC_OPT_NUM) {     /* immediate encoding */
                    best_mlen = nm.len; best_off = nm.off; last_match_pos = cur + 1;
                    direct = true;
                    break;
                }
                const int cur_litlen = (int)uni(o_litlen[cur]);
                const int cur_mlen = (int)(uni(o_ml_off[cur]) & 0xffffu);
                /* before the match: literals at the beginning */
                if (lane >= 1 && lane < MINMATCH) {
                    const int price = p_cur - hc_literals_price(cur_litlen) + hc_literals_price(cur_litlen + lane);
                    const int pos = cur + lane;
                    if (price < (int)o_price[pos]) K4_OPT_SET(pos, 1, 0, cur_litlen + lane, price);
                }
                wave_sync();
                /* prices using the match at position cur, one length per lane */
                {
                    int ll, base;
                    if (cur_mlen == 1) {
                        ll = cur_litlen;
                        base = cur > ll ? (int)uni(o_price[cur - ll]) : 0;
                    } else {
                        ll = 0;
                        base = p_cur;
                    }
                    const int lmp = last_match_pos;
                    bool took_last = false;
                    for (int ml = MINMATCH + lane; ml <= nm.len; ml += 64) {
                        const int pos = cur + ml;
                        const int price = base + hc_sequence_price(ll, ml);
                        if (pos > lmp + 3 || price <= (int)o_price[pos]) {
                            K4_OPT_SET(pos, ml, nm.off, ll, price);
                            if (ml == nm.len) took_last = true;
                        }
                    }
                    /* the last length of the match may extend the table (:993-995) */
                    if (__ballot(took_last) && lmp < cur + nm.len) last_match_pos = cur + nm.len;
                    wave_sync();
                }
                /* complete the following positions with literals */
                {
                    const uint32_t base_price = uni(o_price[last_match_pos]);
                    if (lane >= 1 && lane <= 3) K4_OPT_SET(last_match_pos + lane, 1, 0, lane, base_price + (uint32_t)hc_literals_price(lane));
                }
                wave_sync();
            }
            if (!direct) {
                const uint32_t mo = uni(o_ml_off[last_match_pos]);
                best_mlen = (int)(mo & 0xffffu);
                best_off = (int)(mo >> 16);
                cur = last_match_pos - best_mlen;
            }
            /* encode: walk the chosen path backwards, then emit it forwards (:1018-1059) */
            {
                int candidate_pos = cur, sel_ml = best_mlen, sel_off = best_off;
                for (;;) {
                    const uint32_t mo = uni(o_ml_off[candidate_pos]);
                    const int next_ml = (int)(mo & 0xffffu), next_off = (int)(mo >> 16);
                    wave_sync();
                    if (lane == 0) o_ml_off[candidate_pos] = (uint32_t)sel_ml | ((uint32_t)sel_off << 16);
                    wave_sync();
                    sel_ml = next_ml; sel_off = next_off;
                    if (next_ml > candidate_pos) break;
                    candidate_pos -= next_ml;
                }
            }
            {
                int r = 0;
                while (r < last_match_pos) {
                    const uint32_t mo = uni(o_ml_off[r]);
                    const int ml = (int)(mo & 0xffffu), offset = (int)(mo >> 16);
                    if (ml == 1) { ip++; r++; continue; }
                    r += ml;
                    if (!hc_encode_sequence(src, dst, ip, op, anchor, ml, ip - (uint32_t)offset, limited, oend, lane)) return 0;
                }
            }
        }
    }
#undef K4_OPT_SET
    /* _last_literals (:1062-1098) */
    {
        const uint32_t last_run = U - anchor;
        const uint32_t lit_length = (last_run + 255u - RUN_MASK) / 255u;
        if (limited && op + 1 + (int64_t)lit_length + (int64_t)last_run > oend) return 0;
        if (last_run >= (uint32_t)RUN_MASK) {
            const uint32_t acc = last_run - RUN_MASK;
            const uint32_t nb = acc / 255u;
            if (lane == 0) dst[op] = (uint8_t)(RUN_MASK << ML_BITS);
            wave_fill(dst + op + 1, 255, nb, lane);
            if (lane == 0) dst[op + 1 + nb] = (uint8_t)(acc - nb * 255u);
            op += 2 + nb;
        } else {
            if (lane == 0) dst[op] = (uint8_t)(last_run << ML_BITS);
            op++;
        }
        wave_copy(dst + op, src + anchor, last_run, lane);
        op += last_run;
    }
    return (int)op;
}

/*
 * Round 6: one block parsed by several waves.  The tables of the HC encoder do not depend on the parse (top of this file), so what
 * LZ4HC_compress_hashChain does from a cursor position on -- skip to the next position with a match, arbitrate, emit, go on behind
 * the last match (LL64.high.cs:553-749) -- is a function of that position and the data alone: two parses of one block that ever
 * stand at the same position with a first match in hand go on identically.  Wave j of a block's nseg waves starts at
 * hc_seg_start(j), as if a sequence began there, and writes its records into a slot of its own; every position where it finds a
 * first match (a SYNC POINT) inside its home range [start(j), start(j + 1)) it marks in an LDS bit set.  A wave that has left its
 * home range looks its own sync points up in that set and stops at the first one the wave at home there has marked too (in text the
 * cursors fall into step within a few sequences; a wave that never meets another's sync point simply parses on to the block's end,
 * which is the one-wave parse).  The block's sequences are then wave 0's records up to where it stopped, the records of the wave it
 * joined from that position on up to where THAT one stopped, and so on: wave 0 gathers them behind its own and writes the block
 * out (emit_block<true>).  The bytes are those of the one-wave parse by construction -- nothing is guessed, positions are compared.
 */
constexpr uint32_t HC_SEG_NONE = 0xffffffffu;
constexpr int HC_SEG_CTL = 5;                        /* LDS words per wave: [0] progress, [1] records, [2] the wave joined, [3] where, [4] done */
constexpr uint32_t HC_SEG_SPIN_MAX = 1u << 24;
/* where wave j's records begin inside a block's slots, in records: wave 0 may have to hold the whole block's (PARSE_REC_STRIDE), wave j > 0
 * at most those of the block's last (nseg - j) / nseg */
__host__ __device__ __forceinline__ constexpr uint32_t hc_seg_rec_cap(int nseg, int j) { return j == 0 ? PARSE_REC_STRIDE : (PARSE_REC_STRIDE * (uint32_t)(nseg - j) + (uint32_t)nseg - 1u) / (uint32_t)nseg + 64u; }
/* (a loop, not a recursion: as a recursive function it was CALLED from the kernels -- s_swappc and a stack -- where its arguments were constants) */
__host__ __device__ __forceinline__ constexpr uint32_t hc_seg_rec_off(int nseg, int j)
{
    uint32_t off = 0u;
    for (int i = 0; i < j; i++) off += hc_seg_rec_cap(nseg, i);
    return off;
}
struct HcSegs {
    uint32_t nseg, seg;       /* waves of this block, this wave's number */
    uint32_t *ctl;            /* LDS, HC_SEG_CTL words per wave of the block, zeroed */
    uint32_t *bits;           /* LDS, one bit per position from hc_seg_start(1) on, zeroed */
    uint2 *recs0;             /* wave j's records: recs0 + hc_seg_rec_off(nseg, j) */
    uint32_t *status;         /* the context's status word */
};
/* one bit per position from hc_seg_start(1) on, blocks of up to 64 KiB (the start is rounded down by up to 63) */
constexpr uint32_t hc_seg_bit_dwords(int nseg) { return 65536u / 32u - (65536u / (uint32_t)nseg) / 32u + 2u; }
__device__ __forceinline__ uint32_t hc_seg_start(uint32_t U, uint32_t nseg, uint32_t j) { return j >= nseg ? 0xffffffffu : ((U / nseg) * j) & ~63u; }
__device__ __forceinline__ uint32_t hc_lds_load(const uint32_t *p)
{
#ifndef K4_HOST_EMU
    return uni(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
#else
    return uni(*(const volatile uint32_t *)p);
#endif
}
/* (relaxed: LDS executes a wave's accesses in order, and the per-sync-point marks must not wait for the record stores) */
__device__ __forceinline__ void hc_lds_store(uint32_t *p, uint32_t v)
{
#ifndef K4_HOST_EMU
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    *(volatile uint32_t *)p = v;
#endif
}

/* clTable (LL64.high.cs:1124-1138), hash-chain levels */
__device__ __forceinline__ int hc_nb_searches(int level)
{
    if (level < 1) level = 9;
    return level <= 3 ? 4 : level == 4 ? 8 : level == 5 ? 16 : level == 6 ? 32 : level == 7 ? 64 : level == 8 ? 128 : 256;
}

/* LZ4HC_compress_hashChain (LL64.high.cs:512-800) for one block; returns bytes written, 0 = overflow */
/* REC: the sequences go into `recs` as 8-byte records and the bytes are written afterwards (emit_block<true>, k4lz4_parse.hpp) --
 * LZ4HC_encodeSequence (LL64.high.cs:435-510) is a pure function of (anchor, start, match, length) and the source, and inside this
 * serial loop it was 28 % of the kernel (wave-wide copies and fills with their own trips to memory, 22 more VGPRs, 59 more SGPR
 * spills: profiles/r6_hc_ab.txt).  Only where a match length fits the record: blocks of at most 64 KiB. */
template <bool L3, bool REC = false, bool SEG = false>
__device__ __forceinline__ int hc_parse_block(const uint8_t *src, int src_len, uint8_t *dst, int dst_cap, int level,
                                              const uint4 *cand, int lane,
                                              uint32_t *pace = nullptr, uint32_t *pace_mine = nullptr, uint2 *recs = nullptr, const HcSegs *sg = nullptr)
{
    static_assert(!SEG || (L3 && REC), "several waves per block: the level-3 parse with sequence records");
    uint32_t nrec = 0;
/* one sequence: literals [anchor, ip), a match of ML bytes at REF; the cursor and the anchor move behind it */
#define K4_HC_EMIT(ML, REF, PFA, PFB) \
    do { \
        if (REC) { \
            if (lane == 0) recs[nrec] = make_uint2(ip, (ip - (REF)) | ((uint32_t)((ML) - MINMATCH) << 16)); \
            nrec++; ip += (uint32_t)(ML); anchor = ip; \
        } else if (!hc_encode_sequence(src, dst, ip, op, anchor, (ML), (REF), limited, oend, lane, (PFA), (PFB))) return 0; \
    } while (0)
#define K4_HC_SEARCH(P, LOW, LONGEST, MPOS, SPOS) \
    (L3 ? hc_search_l3(src, cand, hc_get_rec(win, cand, (P)), (P), (LOW), matchlimit, (LONGEST), (MPOS), (SPOS), lane) \
        : hc_search(src, cand, (P), (LOW), matchlimit, (LONGEST), (MPOS), (SPOS), max_attempts, lane, prev, pattern_analysis))
    if ((uint32_t)src_len > (uint32_t)MAX_INPUT_SIZE) return 0;      /* :1153 */
    const bool limited = dst_cap < compress_bound(src_len);         /* :1348 */
    const int64_t oend = dst_cap;
    const int max_attempts = hc_nb_searches(level);
    const bool pattern_analysis = max_attempts > 128;               /* LZ4HC_compress_hashChain: patternAnalysis = (maxNbAttempts > 128), level 9 */
    const uint32_t *prev = (const uint32_t *)cand - (((uint32_t)(src_len > 0 ? src_len : 0) + 3u) & ~3u);
    const uint32_t U = (uint32_t)src_len;
    uint32_t ip = 0, anchor = 0;
    int64_t op = 0;
    HcWindow win;
    win.base = 0; win.valid = false;
    win.rc = make_uint4(0u, 0u, 0u, 0u); win.rc2 = make_uint4(0u, 0u, 0u, 0u);
    uint32_t home_hi = 0xffffffffu, seg_base = 0u, joined = HC_SEG_NONE, stop_e = U;
    bool left = false;
    if (SEG) {
        ip = anchor = hc_seg_start(U, sg->nseg, sg->seg);
        home_hi = hc_seg_start(U, sg->nseg, sg->seg + 1u);
        seg_base = hc_seg_start(U, sg->nseg, 1u);
    }

    if (src_len >= MFLIMIT + 1) {
        const uint32_t mflimit = U - MFLIMIT;
        const uint32_t matchlimit = U - LASTLITERALS;
        while (ip <= mflimit) {
          if (L3) {
            /* One pass of this loop = one window of 64 positions.  The parse normally walks from a window into the next: its
             * records are in the second set already and move up, and the set after it is asked for -- unconditionally, at a
             * clamped address (records of positions past the last searchable one are never looked at), so that the load is
             * not waited for where it is issued.  After a long match the window starts afresh at the cursor. */
            if (K4_HC_PACE && !SEG && pace && ((ip ^ win.base) >> 11) != 0u && ip >= 2048u) Pace::update<13>(pace, pace_mine, ip, U, lane);   /* late blocks first */
            if (win.valid && ip - win.base - 64u < 64u) {
                win.rc = win.rc2;
                win.base += 64u;
            } else {
                const uint32_t pos = ip + (uint32_t)lane < U - 4u ? ip + (uint32_t)lane : U - 4u;
                win.rc = cand[pos];
                win.base = ip;
            }
            win.valid = true;
            const uint32_t pos2 = win.base + 64u + (uint32_t)lane < U - 4u ? win.base + 64u + (uint32_t)lane : U - 4u;
            win.rc2 = cand[pos2];
          }
          do {      /* L3: the sequences that start in this window; other levels: one pass */
            uint32_t pf_anchor = HC_NONE;
            uint8_t pf_byte = 0;
            HcMatch m;
            if (L3) {
                /* positions without a 4-byte candidate match cannot start a sequence: literal runs
                 * are skipped a window at a time, and the record of the first position that can is
                 * already in registers */
                const uint32_t pos = win.base + (uint32_t)lane;
                if (!REC) {
                    if (anchor + (uint32_t)lane < U) pf_byte = src[anchor + (uint32_t)lane];
                    pf_anchor = anchor;
                }
                const unsigned long long hm = __ballot(pos >= ip && pos <= mflimit && win.rc.z != 0u);
                if (!hm) { ip = win.base + 64u; continue; }
                const int fz = ctz64(hm);
                ip = win.base + (uint32_t)fz;
                const HcRec rec = hc_get_rec(win, cand, ip);
                m = hc_search_l3(src, cand, rec, ip, ip, matchlimit, MINMATCH - 1, 0u, ip, lane);
            } else {
                m = hc_search(src, cand, ip, ip, matchlimit, MINMATCH - 1, 0u, ip, max_attempts, lane, prev, pattern_analysis);
            }
            int ml = m.len;
            uint32_t ref = m.mpos;
            if (ml < MINMATCH) { ip++; continue; }
            if (SEG) {      /* a sync point: the cursor at ip, a first match in hand */
                if (ip >= home_hi) {
                    if (!left) { left = true; if (lane == 0) hc_lds_store(sg->ctl + HC_SEG_CTL * sg->seg, 0xffffffffu); }   /* every sync point of the home range is marked */
                    uint32_t j = sg->seg + 1u;                                    /* the wave at home where the cursor stands */
                    while (ip >= hc_seg_start(U, sg->nseg, j + 1u)) j++;
                    if (ip <= hc_lds_load(sg->ctl + HC_SEG_CTL * j)) {             /* ... has been here */
                        const uint32_t w = hc_lds_load(sg->bits + ((ip - seg_base) >> 5));
                        if ((w >> ((ip - seg_base) & 31u)) & 1u) { joined = j; stop_e = ip; ip = U; break; }     /* ... with a first match in hand as well */
                    }
                } else if (sg->seg != 0u) {
                    if (lane == 0) {
                        atomicOr(sg->bits + ((ip - seg_base) >> 5), 1u << ((ip - seg_base) & 31u));
                        hc_lds_store(sg->ctl + HC_SEG_CTL * sg->seg, ip);
                    }
                }
            }
            uint32_t start0 = ip, ref0 = ref;
            int ml0 = ml;
            int ml2 = 0, ml3 = 0;
            uint32_t start2 = 0, ref2 = 0, start3 = 0, ref3 = 0;
            bool go_search2 = true;
            for (;;) {   /* one pass = _Search2 (if requested) followed by _Search3 rounds */
                if (go_search2) {
                    if (ip + (uint32_t)ml <= mflimit) {
                        const HcMatch m2 = K4_HC_SEARCH(ip + (uint32_t)ml - 2u, ip, ml, ref2, start2);
                        ml2 = m2.len; ref2 = m2.mpos; start2 = m2.spos;
                    } else {
                        ml2 = ml;
                    }
                    if (ml2 == ml) {                           /* no better match: encode ML1 */
                        K4_HC_EMIT(ml, ref, pf_anchor, pf_byte);
                        break;
                    }
                    if (start0 < ip) {
                        if (start2 < ip + (uint32_t)ml0) { ip = start0; ref = ref0; ml = ml0; }
                    }
                    if (start2 - ip < 3u) {                    /* first match too small: removed */
                        ml = ml2; ip = start2; ref = ref2;
                        continue;                              /* goto _Search2 */
                    }
                }
                go_search2 = false;
                /* _Search3 */
                if (start2 - ip < (uint32_t)HC_OPTIMAL_ML) {
                    int new_ml = ml;
                    if (new_ml > HC_OPTIMAL_ML) new_ml = HC_OPTIMAL_ML;
                    if (ip + (uint32_t)new_ml > start2 + (uint32_t)ml2 - MINMATCH) new_ml = (int)(start2 - ip) + ml2 - MINMATCH;
                    const int correction = new_ml - (int)(start2 - ip);
                    if (correction > 0) { start2 += (uint32_t)correction; ref2 += (uint32_t)correction; ml2 -= correction; }
                }
                if (start2 + (uint32_t)ml2 <= mflimit) {
                    const HcMatch m3 = K4_HC_SEARCH(start2 + (uint32_t)ml2 - 3u, start2, ml2, ref3, start3);
                    ml3 = m3.len; ref3 = m3.mpos; start3 = m3.spos;
                } else {
                    ml3 = ml2;
                }
                if (ml3 == ml2) {                              /* no better match: encode ML1 and ML2 */
                    if (start2 < ip + (uint32_t)ml) ml = (int)(start2 - ip);
                    K4_HC_EMIT(ml, ref, pf_anchor, pf_byte);
                    ip = start2;
                    K4_HC_EMIT(ml2, ref2, HC_NONE, (uint8_t)0);
                    break;
                }
                if (start3 < ip + (uint32_t)ml + 3u) {         /* not enough space for match 2: remove it */
                    if (start3 >= ip + (uint32_t)ml) {         /* write Seq1 now; Seq3 becomes Seq1 */
                        if (start2 < ip + (uint32_t)ml) {
                            const int correction = (int)(ip + (uint32_t)ml - start2);
                            start2 += (uint32_t)correction; ref2 += (uint32_t)correction; ml2 -= correction;
                            if (ml2 < MINMATCH) { start2 = start3; ref2 = ref3; ml2 = ml3; }
                        }
                        K4_HC_EMIT(ml, ref, pf_anchor, pf_byte);
                        ip = start3; ref = ref3; ml = ml3;
                        start0 = start2; ref0 = ref2; ml0 = ml2;
                        go_search2 = true;
                        continue;                              /* goto _Search2 */
                    }
                    start2 = start3; ref2 = ref3; ml2 = ml3;
                    continue;                                  /* goto _Search3 */
                }
                /* three ascending matches: write the first one */
                if (start2 < ip + (uint32_t)ml) {
                    if (start2 - ip < (uint32_t)HC_OPTIMAL_ML) {
                        if (ml > HC_OPTIMAL_ML) ml = HC_OPTIMAL_ML;
                        if (ip + (uint32_t)ml > start2 + (uint32_t)ml2 - MINMATCH) ml = (int)(start2 - ip) + ml2 - MINMATCH;
                        const int correction = ml - (int)(start2 - ip);
                        if (correction > 0) { start2 += (uint32_t)correction; ref2 += (uint32_t)correction; ml2 -= correction; }
                    } else {
                        ml = (int)(start2 - ip);
                    }
                }
                K4_HC_EMIT(ml, ref, pf_anchor, pf_byte);
                ip = start2; ref = ref2; ml = ml2;             /* ML2 becomes ML1, ML3 becomes ML2 */
                start2 = start3; ref2 = ref3; ml2 = ml3;
                /* goto _Search3 */
            }
          } while (L3 && ip <= mflimit && ip - win.base < 64u);
        }
    }
    if (REC && SEG) {
        uint32_t *mine = sg->ctl + HC_SEG_CTL * sg->seg;
        if (lane == 0) {
            hc_lds_store(mine + 1, nrec); hc_lds_store(mine + 2, joined); hc_lds_store(mine + 3, stop_e); hc_lds_store(mine + 0, 0xffffffffu);
#ifndef K4_HOST_EMU
            __hip_atomic_store(mine + 4, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);        /* (behind the record stores) */
#else
            *(volatile uint32_t *)(mine + 4) = 1u;
#endif
        }
        if (sg->seg != 0u) return 0;
        /* wave 0: the records of the waves it went through, behind its own */
        uint32_t total = nrec, j = joined, e = stop_e;
        while (j != HC_SEG_NONE) {
            const uint32_t *cj = sg->ctl + HC_SEG_CTL * j;
            uint32_t spin = 0;
            while (hc_lds_load(cj + 4) == 0u) {
                if (++spin >= HC_SEG_SPIN_MAX) { dev_status_raise(sg->status, (uint32_t)DEV_STATUS_PIPE_TIMEOUT); return 0; }
                __builtin_amdgcn_s_sleep(8);
            }
            const uint32_t nj = hc_lds_load(cj + 1);
            const uint2 *rj = sg->recs0 + (sg->nseg == 2u ? hc_seg_rec_off(2, 1) : j == 1u ? hc_seg_rec_off(4, 1) : j == 2u ? hc_seg_rec_off(4, 2) : hc_seg_rec_off(4, 3));
            uint32_t lo = 0, hi = nj;                    /* its first record at or behind e (records are in position order) */
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (uni(rj[mid].x) < e) lo = mid + 1u; else hi = mid;
            }
            for (uint32_t t = (uint32_t)lane; t < nj - lo; t += 64u) recs[total + t] = rj[lo + t];
            total += nj - lo;
            e = hc_lds_load(cj + 3);
            j = hc_lds_load(cj + 2);
        }
        wave_sync();
        return emit_block<true>(src, U, dst, dst_cap, recs, total, lane);
    }
    if (REC) {
        wave_sync();                 /* the records are this wave's own stores: in order with the loads that follow */
        return emit_block<true>(src, U, dst, dst_cap, recs, nrec, lane);
    }
    /* _last_literals (:751-787) */
    {
        const uint32_t last_run = U - anchor;
        const uint32_t lit_length = (last_run + 255u - RUN_MASK) / 255u;
        if (limited && op + 1 + (int64_t)lit_length + last_run > oend) return 0;
        if (last_run >= (uint32_t)RUN_MASK) {
            const uint32_t acc = last_run - RUN_MASK;
            const uint32_t nb = acc / 255u;
            if (lane == 0) dst[op] = (uint8_t)(RUN_MASK << ML_BITS);
            wave_fill(dst + op + 1, 255, nb, lane);
            if (lane == 0) dst[op + 1 + nb] = (uint8_t)(acc - nb * 255u);
            op += 2 + nb;
        } else {
            if (lane == 0) dst[op] = (uint8_t)(last_run << ML_BITS);
            op++;
        }
        wave_copy(dst + op, src + anchor, last_run, lane);
        op += last_run;
    }
    return (int)op;
#undef K4_HC_SEARCH
#undef K4_HC_EMIT
}

/* four blocks per workgroup (round 6: one-wave workgroups were placed so that the same 4096 blocks took 15.2 ms instead of 8.9 --
 * profiles/r6_hc_ab.txt) */
constexpr int HC_PARSE_WAVES_PER_WG = 4;
__global__ __launch_bounds__(64 * HC_PARSE_WAVES_PER_WG) void k4_hc_parse_kernel(HcArgs a)
{
    __shared__ uint32_t pace_all[HC_PARSE_WAVES_PER_WG][4];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    uint32_t *pace_mine = pace_all[wave];
    const long long b = (long long)blockIdx.x * HC_PARSE_WAVES_PER_WG + (long long)wave;
    if (b >= a.n) return;
    const int src_len = a.srcLen[b];
    if (K4_HC_PACE) Pace::begin(a.pace, pace_mine, lane);
    const int cap = a.dstCap[b];
    int ret = 0;
    if ((src_len > 0 || (a.flags & FLAG_RAW_RETURN)) && hc_scratch_ok(a)) {
        const uint32_t *prev = (const uint32_t *)(a.work + a.workOff[b]);
        const uint32_t al = ((uint32_t)(src_len > 0 ? src_len : 0) + 3u) & ~3u;
        const uint4 *cand = (const uint4 *)(prev + al);
        const uint8_t *s = a.src + a.srcOff[b];
        uint8_t *d = a.dst + a.dstOff[b];
        if (hc_nb_searches(a.level) <= 4) ret = hc_parse_block<true>(s, src_len, d, cap < 0 ? 0 : cap, a.level, cand, lane, a.pace, pace_mine);
        else ret = hc_parse_block<false>(s, src_len, d, cap < 0 ? 0 : cap, a.level, cand, lane);
    }
    if (lane == 0) {
        int r = ret;
        if (!(a.flags & FLAG_RAW_RETURN)) r = src_len <= 0 ? 0 : (ret <= 0 ? -1 : ret);   /* LZ4Codec.cs:45-51 */
        else if (!hc_scratch_ok(a)) r = HC_NO_SCRATCH;      /* "not encoded", which 0 would not say to the pickle envelope (raw fallback) */
        a.outLen[b] = r;
    }
}

/* level 3 on blocks of at most 64 KiB, sequences as records and the bytes behind the parse (HcArgs::recs): a kernel of its own so
 * that its registers are this form's and not the maximum over the three forms of k4_hc_parse_kernel */
constexpr int HC_REC_WAVES_PER_WG = HC_PARSE_WAVES_PER_WG;
__global__ __launch_bounds__(64 * HC_REC_WAVES_PER_WG) void k4_hc_parse_rec_kernel(HcArgs a)
{
    __shared__ uint32_t pace_all[HC_REC_WAVES_PER_WG][4];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    uint32_t *pace_mine = pace_all[wave];
    const long long b = (long long)blockIdx.x * HC_REC_WAVES_PER_WG + (long long)wave;
    if (b >= a.n) return;
    const int src_len = a.srcLen[b];
    if (K4_HC_PACE) Pace::begin(a.pace, pace_mine, lane);
    const int cap = a.dstCap[b];
    int ret = 0;
    if ((src_len > 0 || (a.flags & FLAG_RAW_RETURN)) && hc_scratch_ok(a)) {
        const uint32_t *prev = (const uint32_t *)(a.work + a.workOff[b]);
        const uint32_t al = ((uint32_t)(src_len > 0 ? src_len : 0) + 3u) & ~3u;
        const uint4 *cand = (const uint4 *)(prev + al);
        ret = hc_parse_block<true, true>(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, a.level, cand, lane, a.pace, pace_mine,
                                         a.recs + (unsigned long long)b * PARSE_REC_STRIDE);
    }
    if (lane == 0) {
        int r = ret;
        if (!(a.flags & FLAG_RAW_RETURN)) r = src_len <= 0 ? 0 : (ret <= 0 ? -1 : ret);   /* LZ4Codec.cs:45-51 */
        else if (!hc_scratch_ok(a)) r = HC_NO_SCRATCH;
        a.outLen[b] = r;
    }
}

/* ... and with NSEG waves per block (HcSegs above): eight waves per workgroup, 8 / NSEG blocks.  Blocks too short to be worth cutting
 * are parsed by their wave 0 alone. */
constexpr int HC_SEG_WAVES_PER_WG = 8;
#ifndef K4_HC_SEG_ATTR
#define K4_HC_SEG_ATTR
#endif
constexpr uint32_t HC_SEG_MIN_LEN = 8192u;
template <int NSEG>
__device__ __forceinline__ void hc_parse_seg_kernel_body(const HcArgs &a, uint32_t *lds)
{
    constexpr int BLOCKS = HC_SEG_WAVES_PER_WG / NSEG;
    constexpr uint32_t BIT_DWORDS = hc_seg_bit_dwords(NSEG);
    constexpr uint32_t PER_BLOCK = BIT_DWORDS + (uint32_t)(HC_SEG_CTL * NSEG);
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t blk = wave / (uint32_t)NSEG, seg = wave % (uint32_t)NSEG;
    for (uint32_t k = threadIdx.x; k < PER_BLOCK * (uint32_t)BLOCKS; k += 64u * HC_SEG_WAVES_PER_WG) lds[k] = 0u;
    __syncthreads();
    const long long b = (long long)blockIdx.x * BLOCKS + (long long)blk;
    if (b >= a.n) return;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    const bool cut = src_len >= (int)HC_SEG_MIN_LEN && src_len <= 65536;
    if (!cut && seg != 0u) return;
    int ret = 0;
    if ((src_len > 0 || (a.flags & FLAG_RAW_RETURN)) && hc_scratch_ok(a)) {
        const uint32_t *prev = (const uint32_t *)(a.work + a.workOff[b]);
        const uint32_t al = ((uint32_t)(src_len > 0 ? src_len : 0) + 3u) & ~3u;
        const uint4 *cand = (const uint4 *)(prev + al);
        uint2 *recs0 = a.recs + (unsigned long long)b * hc_seg_rec_off(NSEG, NSEG);
        if (cut) {
            HcSegs sg;
            sg.nseg = (uint32_t)NSEG; sg.seg = seg; sg.bits = lds + PER_BLOCK * blk; sg.ctl = sg.bits + BIT_DWORDS; sg.recs0 = recs0; sg.status = a.status;
            ret = hc_parse_block<true, true, true>(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, a.level, cand, lane, nullptr, nullptr,
                                                   recs0 + (seg == 0u ? 0u : seg == 1u ? hc_seg_rec_off(NSEG, 1) : seg == 2u ? hc_seg_rec_off(NSEG, 2) : hc_seg_rec_off(NSEG, 3)), &sg);
            if (seg != 0u) return;
        } else {
            ret = hc_parse_block<true, true>(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, a.level, cand, lane, nullptr, nullptr, recs0);
        }
    }
    if (lane == 0) {
        int r = ret;
        if (!(a.flags & FLAG_RAW_RETURN)) r = src_len <= 0 ? 0 : (ret <= 0 ? -1 : ret);   /* LZ4Codec.cs:45-51 */
        else if (!hc_scratch_ok(a)) r = HC_NO_SCRATCH;
        a.outLen[b] = r;
    }
}
__global__ __launch_bounds__(64 * HC_SEG_WAVES_PER_WG) K4_HC_SEG_ATTR void k4_hc_parse_seg2_kernel(HcArgs a)
{
    __shared__ uint32_t lds[(hc_seg_bit_dwords(2) + HC_SEG_CTL * 2u) * 4u];
    hc_parse_seg_kernel_body<2>(a, lds);
}
__global__ __launch_bounds__(64 * HC_SEG_WAVES_PER_WG) K4_HC_SEG_ATTR void k4_hc_parse_seg4_kernel(HcArgs a)
{
    __shared__ uint32_t lds[(hc_seg_bit_dwords(4) + HC_SEG_CTL * 4u) * 2u];
    hc_parse_seg_kernel_body<4>(a, lds);
}

/* levels 10..12: one wavefront per block, the price table in LDS (48 KiB: three blocks per CU) */
__global__ __launch_bounds__(64) void k4_hc_parse_opt_kernel(HcArgs a)
{
    __shared__ uint32_t opt_lds[HC_OPT_LDS_DWORDS];
    const int lane = lane_id();
    const long long b = (long long)blockIdx.x;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    int ret = 0;
    if ((src_len > 0 || (a.flags & FLAG_RAW_RETURN)) && hc_scratch_ok(a)) {
        const uint32_t *prev = (const uint32_t *)(a.work + a.workOff[b]);
        ret = hc_parse_block_opt(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, a.level, prev, opt_lds, lane);
    }
    if (lane == 0) {
        int r = ret;
        if (!(a.flags & FLAG_RAW_RETURN)) r = src_len <= 0 ? 0 : (ret <= 0 ? -1 : ret);   /* LZ4Codec.cs:45-51 */
        else if (!hc_scratch_ok(a)) r = HC_NO_SCRATCH;      /* "not encoded", which 0 would not say to the pickle envelope (raw fallback) */
        a.outLen[b] = r;
    }
}

}  // namespace k4
