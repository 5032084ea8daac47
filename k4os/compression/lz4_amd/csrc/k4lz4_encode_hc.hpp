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
 * positions stored +1), prev[U] (u32), cand[U][4] (u32), flen[U][4] and blen[U][4] (u16: forward /
 * backward match length of position p against its k-th candidate, exact below HC_FLEN_CAP /
 * HC_BLEN_CAP), see k4_hc_layout_kernel.  With the lengths precomputed, level 3 (4 attempts =
 * exactly one record) needs no data compare at all in the common case: a search is one record
 * load, literal runs are skipped 64 positions per load.
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
    unsigned int posBase;      /* k4_hc_cand_kernel: first position covered by blockIdx.y == 0 */
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
__device__ __forceinline__ uint64_t hc_work_bytes(int len) { return len > 0 ? (((uint64_t)len + 3u) & ~3ull) * 36u : 0u; }

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
template <typename TAB>
__device__ __forceinline__ void hc_chain_block(const uint8_t *src, uint32_t U, TAB *tab, uint32_t *prev, uint32_t *seen, int lane)
{
    const uint32_t npos = U - 3u;                          /* positions whose 4 bytes exist */
    const unsigned long long below_me = (1ull << lane) - 1ull, above_me = ~(below_me | (1ull << lane));
    /* The source bytes of a step are asked for four steps ahead (unconditionally, at a clamped position), into a register
     * of their own: the loop is unrolled four times so that no word has to be moved from one register to another while it
     * is still on its way -- such a move waits for the load, and a step would wait for memory after all. */
    uint32_t wq[4];
#pragma unroll
    for (uint32_t u = 0; u < 4u; u++) wq[u] = ld32u(src + (64u * u + (uint32_t)lane < npos ? 64u * u + (uint32_t)lane : 0u));
    for (uint32_t q0 = 0; q0 < npos; q0 += 256u) {
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            const uint32_t p0 = q0 + 64u * u;
            const uint32_t p = p0 + (uint32_t)lane;
            const bool act = p < npos;
            const uint32_t w = wq[u];
            wq[u] = ld32u(src + (p + 256u < npos ? p + 256u : 0u));
            uint32_t h = 0, pr = HC_NONE;
            bool flagged = false;
            if (act) {
                h = hc_hash(w);
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
            if (act) prev[p] = pr;
            if (last) tab[h] = (TAB)(p + 1u);
            /* the next step's look-ups come after these puts: program order for a table in LDS (nothing waits for the
             * stores of prev[], which nobody reads here), wave_sync for one in memory */
            if (sizeof(TAB) == 2) lds_sync(); else wave_sync();
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

/* ---- kernel 1b: candidates + forward lengths, every position independently ------------------ */
constexpr int HC_CAND_POS_PER_WG = 1024;
#ifndef K4_HC_CAND_GROUP
#define K4_HC_CAND_GROUP 1
#endif
__global__ __launch_bounds__(256) void k4_hc_cand_kernel(HcArgs a)
{
    const long long b = (long long)blockIdx.x;
    const int len = a.srcLen[b];
    if (len < MFLIMIT + 1 || !hc_scratch_ok(a)) return;
    const uint32_t U = (uint32_t)len;
    const uint32_t npos = U - 3u;
    const uint32_t first = a.posBase + (uint32_t)blockIdx.y * (uint32_t)HC_CAND_POS_PER_WG;
    if (first >= npos) return;
    const uint8_t *src = a.src + a.srcOff[b];
    const uint32_t *prev = (const uint32_t *)(a.work + a.workOff[b]);
    uint32_t *cand = (uint32_t *)prev + ((U + 3u) & ~3u);
    uint2 *flen = (uint2 *)(cand + 4u * ((U + 3u) & ~3u));
    uint2 *blen = flen + ((U + 3u) & ~3u);
    /* Per position: the first four chain candidates (a chain step of 65535 or more ends the walk: LL.high.cs:114 caps the delta, and such
     * a candidate is below lowestMatchIndex), per candidate the forward match length a search at p may use (:87-88 lowest, :120 the
     * 4-byte test, :126 LZ4_count up to matchlimit, capped at HC_FLEN_CAP) and the equal bytes before the two positions (LZ4HC_countBack
     * without its limits, capped at HC_BLEN_CAP).  Written so that K4_HC_CAND_GROUP of a thread's four positions go TOGETHER, every
     * step's loads -- prev[p], prev[c0], prev[c1], prev[c2], the candidates' first four bytes, their eight bytes forward and backward
     * -- issued for all of them before the first is used.  Round 6 measured what that is worth: nothing -- groups of 1 / 2 / 4: 11.07 /
     * 11.24 / 13.96 ms for the bench batch (46 / 73 / 131 VGPRs; the one-position loop of rounds 1-5: 11.58).  The kernel is not
     * waiting for its chains: an unaligned eight-byte read at an address of its own per lane costs what it costs wherever it is
     * issued (experiments/hc_cand_lds), and more in flight per wave only takes waves away. */
    constexpr int PP = K4_HC_CAND_GROUP;                  /* positions of a thread that go together (of HC_CAND_POS_PER_WG / 256 = 4) */
    static_assert((HC_CAND_POS_PER_WG / 256) % PP == 0, "groups divide a thread's positions");
    const uint32_t matchlimit = U - LASTLITERALS;
  for (int grp = 0; grp < HC_CAND_POS_PER_WG / 256 / PP; grp++) {
    uint32_t pos[PP], c[PP][4], seq[PP], lim[PP];
    uint64_t pf0[PP], pb0[PP];
    bool in[PP];
#pragma unroll
    for (int j = 0; j < PP; j++) {
        pos[j] = first + threadIdx.x + 256u * (uint32_t)(grp * PP + j);
        in[j] = pos[j] < npos;
        const uint32_t p = in[j] ? pos[j] : 0u;
        c[j][0] = in[j] ? prev[p] : HC_NONE;
        seq[j] = ld32u(src + p);
        const uint32_t maxn = p + MINMATCH < matchlimit ? matchlimit - (p + MINMATCH) : 0u;
        lim[j] = maxn < HC_FLEN_CAP - MINMATCH ? maxn : HC_FLEN_CAP - MINMATCH;
        /* the position's own eight bytes behind its first four and before it: read once, not once per candidate */
        pf0[j] = (in[j] && lim[j] >= 8u) ? ld64u(src + p + MINMATCH) : 0ull;
        pb0[j] = (in[j] && p >= 8u) ? ld64u(src + p - 8u) : 0ull;
    }
#pragma unroll
    for (int k = 1; k < 4; k++) {
        uint32_t q[PP];
#pragma unroll
        for (int j = 0; j < PP; j++) q[j] = c[j][k - 1] != HC_NONE ? prev[c[j][k - 1]] : HC_NONE;
#pragma unroll
        for (int j = 0; j < PP; j++) c[j][k] = (q[j] != HC_NONE && c[j][k - 1] - q[j] < (uint32_t)DISTANCE_MAX) ? q[j] : HC_NONE;
    }
    /* which candidates a search at p may use, and their first four bytes */
    bool ok[PP][4];
    uint32_t cseq[PP][4];
#pragma unroll
    for (int j = 0; j < PP; j++) {
        const uint32_t lowest = pos[j] > (uint32_t)DISTANCE_MAX ? pos[j] - (uint32_t)DISTANCE_MAX : 0u;
        bool chain_ok = in[j];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            chain_ok = chain_ok && c[j][k] != HC_NONE && c[j][k] >= lowest;
            ok[j][k] = chain_ok;
            cseq[j][k] = chain_ok ? ld32u(src + c[j][k]) : 0u;
        }
    }
    /* the first eight bytes forward and backward of every candidate that has the four */
    uint64_t cf[PP][4], cb[PP][4];
#pragma unroll
    for (int j = 0; j < PP; j++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ok[j][k] = ok[j][k] && cseq[j][k] == seq[j];
            const uint32_t cc = c[j][k];
            cf[j][k] = (ok[j][k] && lim[j] >= 8u) ? ld64u(src + cc + MINMATCH) : 0ull;
            cb[j][k] = (ok[j][k] && cc >= 8u) ? ld64u(src + cc - 8u) : 0ull;
        }
#pragma unroll
    for (int j = 0; j < PP; j++) {
        uint32_t fl[4], bl[4];
        const uint32_t p = pos[j];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t cc = c[j][k];
            uint32_t f = 0, bk = 0;
            if (ok[j][k]) {
                const uint32_t l = lim[j];
                uint32_t i = 0;
                bool more = true;                       /* the count is not finished by the first eight bytes */
                if (l >= 8u) {
                    const uint64_t x = pf0[j] ^ cf[j][k];
                    if (x) { i = (uint32_t)(__ffsll((unsigned long long)x) - 1) >> 3; more = false; }
                    else i = 8u;
                }
                if (more) {
                    while (i + 8u <= l) {
                        const uint64_t x = ld64u(src + p + MINMATCH + i) ^ ld64u(src + cc + MINMATCH + i);
                        if (x) { i += (uint32_t)(__ffsll((unsigned long long)x) - 1) >> 3; break; }
                        i += 8u;
                    }
                    if (i + 8u > l) while (i < l && src[p + MINMATCH + i] == src[cc + MINMATCH + i]) i++;
                }
                f = MINMATCH + i;
                /* equal bytes before the two positions */
                const uint32_t blim = cc < HC_BLEN_CAP ? cc : HC_BLEN_CAP;       /* cc < p */
                more = true;
                if (blim >= 8u) {
                    const uint64_t x = pb0[j] ^ cb[j][k];
                    if (x) { bk = (uint32_t)__clzll((unsigned long long)x) >> 3; more = false; }
                    else bk = 8u;
                }
                if (more) {
                    while (bk + 8u <= blim) {
                        const uint64_t x = ld64u(src + p - 8u - bk) ^ ld64u(src + cc - 8u - bk);
                        if (x) { bk += (uint32_t)__clzll((unsigned long long)x) >> 3; break; }
                        bk += 8u;
                    }
                    if (bk + 8u > blim) while (bk < blim && src[p - 1u - bk] == src[cc - 1u - bk]) bk++;
                }
            }
            fl[k] = f;
            bl[k] = bk;
        }
        if (in[j]) {
            ((uint4 *)cand)[p] = make_uint4(c[j][0], c[j][1], c[j][2], c[j][3]);
            flen[p] = make_uint2(fl[0] | (fl[1] << 16), fl[2] | (fl[3] << 16));
            blen[p] = make_uint2(bl[0] | (bl[1] << 16), bl[2] | (bl[3] << 16));
        }
    }
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
__device__ __forceinline__ HcMatch hc_search(const uint8_t *src, const uint32_t *cand, uint32_t ip, uint32_t ilow,
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
        const uint4 rec = ((const uint4 *)cand)[rec_at];
        uint32_t c = grp == 0 ? rec.x : grp == 1 ? rec.y : grp == 2 ? rec.z : rec.w;
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
            uint32_t nx = grp == 0 ? rec.y : grp == 1 ? rec.z : grp == 2 ? rec.w : HC_NONE;
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

/* one position's precomputed record, wave-uniform */
struct HcRec { uint32_t c0, c1, c2, c3, fl01, fl23, bl01, bl23; };

__device__ __forceinline__ HcRec hc_load_rec(const uint32_t *cand, const uint2 *flen, const uint2 *blen, uint32_t p)
{
    const uint4 rc = ((const uint4 *)cand)[p];
    const uint2 f = flen[p], b = blen[p];
    HcRec r;
    r.c0 = uni(rc.x); r.c1 = uni(rc.y); r.c2 = uni(rc.z); r.c3 = uni(rc.w);
    r.fl01 = uni(f.x); r.fl23 = uni(f.y); r.bl01 = uni(b.x); r.bl23 = uni(b.y);
    return r;
}

/* The records of 128 consecutive positions, two per lane: [base, base + 64) in rc / f / b, the 64 after them in rc2 / f2 /
 * b2.  The second set is asked for when the parse enters the first one and has a window's worth of work to arrive in;
 * the searches of the three-match arbitration look ahead by a match length at most, so they find their records here. */
struct HcWindow { uint32_t base; bool valid; uint4 rc; uint2 f, b; uint4 rc2; uint2 f2, b2; };

__device__ __forceinline__ HcRec hc_get_rec(const HcWindow &w, const uint32_t *cand, const uint2 *flen, const uint2 *blen, uint32_t p)
{
    if (w.valid && p - w.base < 64u) {
        const int l = (int)(p - w.base);
        HcRec r;
        r.c0 = __builtin_amdgcn_readlane(w.rc.x, l); r.c1 = __builtin_amdgcn_readlane(w.rc.y, l);
        r.c2 = __builtin_amdgcn_readlane(w.rc.z, l); r.c3 = __builtin_amdgcn_readlane(w.rc.w, l);
        r.fl01 = __builtin_amdgcn_readlane(w.f.x, l); r.fl23 = __builtin_amdgcn_readlane(w.f.y, l);
        r.bl01 = __builtin_amdgcn_readlane(w.b.x, l); r.bl23 = __builtin_amdgcn_readlane(w.b.y, l);
        return r;
    }
    if (w.valid && p - w.base < 128u) {
        const int l = (int)(p - w.base - 64u);
        HcRec r;
        r.c0 = __builtin_amdgcn_readlane(w.rc2.x, l); r.c1 = __builtin_amdgcn_readlane(w.rc2.y, l);
        r.c2 = __builtin_amdgcn_readlane(w.rc2.z, l); r.c3 = __builtin_amdgcn_readlane(w.rc2.w, l);
        r.fl01 = __builtin_amdgcn_readlane(w.f2.x, l); r.fl23 = __builtin_amdgcn_readlane(w.f2.y, l);
        r.bl01 = __builtin_amdgcn_readlane(w.b2.x, l); r.bl23 = __builtin_amdgcn_readlane(w.b2.y, l);
        return r;
    }
    return hc_load_rec(cand, flen, blen, p);
}

/*
 * The same search when the whole chain walk is one record (level 3: 4 attempts): match lengths
 * come from the precomputed record, nothing is compared.  Falls back to hc_search when a length
 * sits at its cap and could be longer.
 */
__device__ __forceinline__ HcMatch hc_search_l3(const uint8_t *src, const uint32_t *cand, const HcRec &rec, uint32_t ip, uint32_t ilow,
                                                uint32_t matchlimit, int longest, uint32_t mpos, uint32_t spos, int lane)
{
    uint32_t l0 = rec.fl01 & 0xffffu, l1 = rec.fl01 >> 16, l2 = rec.fl23 & 0xffffu, l3 = rec.fl23 >> 16;
    const uint32_t look_back = ip - ilow;
    if (l0 == HC_FLEN_CAP || l1 == HC_FLEN_CAP || l2 == HC_FLEN_CAP || l3 == HC_FLEN_CAP) {
        /* a forward length at its cap: the count goes on from there, 16 lanes per candidate and 64 bytes per step -- one
         * trip to memory for the usual match, where the general search would start from the candidate records again */
        const int grp = lane >> 4, sub = lane & 15;
        const uint32_t c = grp == 0 ? rec.c0 : grp == 1 ? rec.c1 : grp == 2 ? rec.c2 : rec.c3;
        const uint32_t l = grp == 0 ? l0 : grp == 1 ? l1 : grp == 2 ? l2 : l3;
        const uint32_t maxn = matchlimit - (ip + MINMATCH);
        bool open = l == HC_FLEN_CAP;
        uint32_t done = HC_FLEN_CAP - MINMATCH, fwd = l;
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
                if (gm) { fwd = MINMATCH + done + 4u * (uint32_t)fl + nq; open = false; }
                else done += 64u;
            }
            if (!__ballot(open)) break;
        }
        l0 = readlane_u32(fwd, 0); l1 = readlane_u32(fwd, 16); l2 = readlane_u32(fwd, 32); l3 = readlane_u32(fwd, 48);
    }
    bool slow = false;
    uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    if (look_back) {                                        /* LZ4HC_countBack = min(equal bytes, ip - ilow, match - 0) */
        const uint32_t s0 = rec.bl01 & 0xffffu, s1 = rec.bl01 >> 16, s2 = rec.bl23 & 0xffffu, s3 = rec.bl23 >> 16;
        const uint32_t m0 = look_back < rec.c0 ? look_back : rec.c0, m1 = look_back < rec.c1 ? look_back : rec.c1,
                       m2 = look_back < rec.c2 ? look_back : rec.c2, m3 = look_back < rec.c3 ? look_back : rec.c3;
        b0 = s0 < m0 ? s0 : m0; b1 = s1 < m1 ? s1 : m1; b2 = s2 < m2 ? s2 : m2; b3 = s3 < m3 ? s3 : m3;
        slow = slow || (l0 && s0 == HC_BLEN_CAP && m0 > HC_BLEN_CAP) || (l1 && s1 == HC_BLEN_CAP && m1 > HC_BLEN_CAP) ||
               (l2 && s2 == HC_BLEN_CAP && m2 > HC_BLEN_CAP) || (l3 && s3 == HC_BLEN_CAP && m3 > HC_BLEN_CAP);
    }
    if (slow) return hc_search(src, cand, ip, ilow, matchlimit, longest, mpos, spos, 4, lane);
    HcMatch r;
    r.len = longest; r.mpos = mpos; r.spos = spos;
    const int m0 = l0 ? (int)(l0 + b0) : 0, m1 = l1 ? (int)(l1 + b1) : 0, m2 = l2 ? (int)(l2 + b2) : 0, m3 = l3 ? (int)(l3 + b3) : 0;
    if (m0 > r.len) { r.len = m0; r.mpos = rec.c0 - b0; r.spos = ip - b0; }
    if (m1 > r.len) { r.len = m1; r.mpos = rec.c1 - b1; r.spos = ip - b1; }
    if (m2 > r.len) { r.len = m2; r.mpos = rec.c2 - b2; r.spos = ip - b2; }
    if (m3 > r.len) { r.len = m3; r.mpos = rec.c3 - b3; r.spos = ip - b3; }
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
template <bool L3, bool REC = false>
__device__ __forceinline__ int hc_parse_block(const uint8_t *src, int src_len, uint8_t *dst, int dst_cap, int level,
                                              const uint32_t *cand, const uint2 *flen, const uint2 *blen, int lane,
                                              uint32_t *pace = nullptr, uint32_t *pace_mine = nullptr, uint2 *recs = nullptr)
{
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
    (L3 ? hc_search_l3(src, cand, hc_get_rec(win, cand, flen, blen, (P)), (P), (LOW), matchlimit, (LONGEST), (MPOS), (SPOS), lane) \
        : hc_search(src, cand, (P), (LOW), matchlimit, (LONGEST), (MPOS), (SPOS), max_attempts, lane, prev, pattern_analysis))
    if ((uint32_t)src_len > (uint32_t)MAX_INPUT_SIZE) return 0;      /* :1153 */
    const bool limited = dst_cap < compress_bound(src_len);         /* :1348 */
    const int64_t oend = dst_cap;
    const int max_attempts = hc_nb_searches(level);
    const bool pattern_analysis = max_attempts > 128;               /* LZ4HC_compress_hashChain: patternAnalysis = (maxNbAttempts > 128), level 9 */
    const uint32_t *prev = cand - (((uint32_t)(src_len > 0 ? src_len : 0) + 3u) & ~3u);
    const uint32_t U = (uint32_t)src_len;
    uint32_t ip = 0, anchor = 0;
    int64_t op = 0;
    HcWindow win;
    win.base = 0; win.valid = false;
    win.rc = make_uint4(0u, 0u, 0u, 0u); win.f = make_uint2(0u, 0u); win.b = make_uint2(0u, 0u);
    win.rc2 = make_uint4(0u, 0u, 0u, 0u); win.f2 = make_uint2(0u, 0u); win.b2 = make_uint2(0u, 0u);

    if (src_len >= MFLIMIT + 1) {
        const uint32_t mflimit = U - MFLIMIT;
        const uint32_t matchlimit = U - LASTLITERALS;
        while (ip <= mflimit) {
          if (L3) {
            /* One pass of this loop = one window of 64 positions.  The parse normally walks from a window into the next: its
             * records are in the second set already and move up, and the set after it is asked for -- unconditionally, at a
             * clamped address (records of positions past the last searchable one are never looked at), so that the load is
             * not waited for where it is issued.  After a long match the window starts afresh at the cursor. */
            if (K4_HC_PACE && pace && ((ip ^ win.base) >> 11) != 0u && ip >= 2048u) Pace::update<13>(pace, pace_mine, ip, U, lane);   /* late blocks first */
            if (win.valid && ip - win.base - 64u < 64u) {
                win.rc = win.rc2; win.f = win.f2; win.b = win.b2;
                win.base += 64u;
            } else {
                const uint32_t pos = ip + (uint32_t)lane < U - 4u ? ip + (uint32_t)lane : U - 4u;
                win.rc = ((const uint4 *)cand)[pos]; win.f = flen[pos]; win.b = blen[pos];
                win.base = ip;
            }
            win.valid = true;
            const uint32_t pos2 = win.base + 64u + (uint32_t)lane < U - 4u ? win.base + 64u + (uint32_t)lane : U - 4u;
            win.rc2 = ((const uint4 *)cand)[pos2]; win.f2 = flen[pos2]; win.b2 = blen[pos2];
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
                const unsigned long long hm = __ballot(pos >= ip && pos <= mflimit && (win.f.x | win.f.y) != 0u);
                if (!hm) { ip = win.base + 64u; continue; }
                const int fz = ctz64(hm);
                ip = win.base + (uint32_t)fz;
                const HcRec rec = hc_get_rec(win, cand, flen, blen, ip);
                m = hc_search_l3(src, cand, rec, ip, ip, matchlimit, MINMATCH - 1, 0u, ip, lane);
            } else {
                m = hc_search(src, cand, ip, ip, matchlimit, MINMATCH - 1, 0u, ip, max_attempts, lane, prev, pattern_analysis);
            }
            int ml = m.len;
            uint32_t ref = m.mpos;
            if (ml < MINMATCH) { ip++; continue; }
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
        const uint32_t *cand = prev + al;
        const uint2 *flen = (const uint2 *)(cand + 4u * al);
        const uint2 *blen = flen + al;
        const uint8_t *s = a.src + a.srcOff[b];
        uint8_t *d = a.dst + a.dstOff[b];
        if (hc_nb_searches(a.level) <= 4) ret = hc_parse_block<true>(s, src_len, d, cap < 0 ? 0 : cap, a.level, cand, flen, blen, lane, a.pace, pace_mine);
        else ret = hc_parse_block<false>(s, src_len, d, cap < 0 ? 0 : cap, a.level, cand, flen, blen, lane);
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
        const uint32_t *cand = prev + al;
        const uint2 *flen = (const uint2 *)(cand + 4u * al);
        const uint2 *blen = flen + al;
        ret = hc_parse_block<true, true>(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, a.level, cand, flen, blen, lane, a.pace, pace_mine,
                                         a.recs + (unsigned long long)b * PARSE_REC_STRIDE);
    }
    if (lane == 0) {
        int r = ret;
        if (!(a.flags & FLAG_RAW_RETURN)) r = src_len <= 0 ? 0 : (ret <= 0 ? -1 : ret);   /* LZ4Codec.cs:45-51 */
        else if (!hc_scratch_ok(a)) r = HC_NO_SCRATCH;
        a.outLen[b] = r;
    }
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
