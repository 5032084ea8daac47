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
 * parallel".  What the wavefront parallelises instead:
 *
 *   search   the next 64 probe positions of the skip schedule (LL64.fast.cs:156-234) are hashed,
 *            looked up (16 KiB table in LDS) and compared at once, one per lane.  The serial
 *            semantics "a probe sees the puts of every earlier probe" are restored inside the
 *            wave: a lane whose hash equals that of an earlier lane takes that lane's position as
 *            its candidate (found with shuffles over the window that can still matter), and only
 *            the puts of lanes up to the first hit are committed, in lane order.
 *   extend   backward extension and LZ4_count compare 64 / 256 bytes per step with a ballot.
 *   emit     literal runs move 1 KiB per wave instruction; 255-runs are wave fills.
 *
 * Table slots hold positions relative to the block start (currentOffset == 0, empty slot == 0).
 */
#pragma once
#include "k4lz4_common.hpp"

namespace k4 {

template <bool BYU16> struct FastTable;

template <> struct FastTable<true> {   /* byU16: 8192 x u16, hash4 >> 19 (LL.tools.cs:46-51) */
    uint16_t *t;
    __device__ __forceinline__ static uint32_t hash(const uint8_t *p) { return (ld32u(p) * 2654435761u) >> (32 - 13); }
    __device__ __forceinline__ uint32_t get(uint32_t h) const { return t[h]; }
    __device__ __forceinline__ void put(uint32_t h, uint32_t pos) const { t[h] = (uint16_t)pos; }
};
template <> struct FastTable<false> {  /* byU32: 4096 x u32, hash5 (LL.tools.cs:53-58, LL64.tools.cs:135-143) */
    uint32_t *t;
    __device__ __forceinline__ static uint32_t hash(const uint8_t *p)
    {
        return (uint32_t)(((ld64u(p) << 24) * 889523592379ull) >> (64 - 12));
    }
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
        const unsigned long long notfull = __ballot(neq != 4u);
        if (!notfull) { done += 256u; continue; }
        const int fl = ctz64(notfull);
        return done + 4u * (uint32_t)fl + __shfl(neq, fl);
    }
}

/* length field tail: `rem` encoded as 255-run + final byte (LL64.fast.cs:262-272,:365-381,:484-495) */
__device__ __forceinline__ uint32_t emit_length_run(uint8_t *dst, uint32_t op, uint32_t rem, int lane)
{
    const uint32_t nb = rem / 255u;
    wave_fill(dst + op, 255, nb, lane);
    if (lane == 0) dst[op + nb] = (uint8_t)(rem - nb * 255u);
    return op + nb + 1u;
}

/*
 * LL64.LZ4_compress_generic for one block.  `tab` is this wave's 16 KiB LDS table (zeroed here:
 * LZ4_initStream, LL.tools.cs:235-239).  Returns bytes written, 0 when the output does not fit.
 *
 * Every iteration of the main loop produces one sequence with three dependent memory round trips:
 *   (1) the 4 (8) source bytes of up to 64 probe positions        -> hashes -> LDS table lookups
 *   (2) the 4 bytes at the 64 candidates                          -> first hit
 *   (3) backward bytes + up to 256 forward bytes of both sides, and the literal bytes
 * The reference's "test the position right after a match" step (LL64.fast.cs:393-463) rides in
 * lane 0 of the next probe round instead of being a round trip of its own.
 */
template <bool BYU16, bool PROF = false>
__device__ __forceinline__ int encode_fast_block(const uint8_t *src, int src_len, uint8_t *dst, int dst_cap,
                                                 uint32_t accel, uint32_t *tabw, int lane, unsigned long long *pc = nullptr)
{
    unsigned long long c_probe = 0, c_ext = 0, c_emit = 0, n_seq = 0, n_round = 0, n_dup = 0, n_win = 0;
    prof_place<PROF>(pc, 8, lane);
    const unsigned long long t_begin = prof_now<PROF>();
    if ((uint32_t)src_len > (uint32_t)MAX_INPUT_SIZE) return 0;     /* LL64.fast.cs:90 */
    const bool limited = dst_cap < compress_bound(src_len);         /* :524 */
    const int64_t olimit = dst_cap;
    const uint32_t U = (uint32_t)src_len;
    FastTable<BYU16> tab;
    tab.t = (decltype(tab.t))tabw;

    for (int k = lane; k < 1024; k += 64) ((uint4 *)tabw)[k] = make_uint4(0u, 0u, 0u, 0u);
    wave_sync();

    uint32_t anchor = 0;
    int64_t op = 0;

    if (src_len >= MFLIMIT + 1) {                                   /* :117 */
        const uint32_t mflimit_plus_one = U - MFLIMIT + 1;
        const uint32_t matchlimit = U - LASTLITERALS;

        if (lane == 0) tab.put(FastTable<BYU16>::hash(src), 0);     /* :119-122 */
        uint32_t ip = 1;
        bool test = false;   /* true: `ip` is the position right after a match (:393-463) */

        for (;;) {
            /* ---------------- probe rounds ---------------- */
            const unsigned long long t0 = prof_now<PROF>();
            if (test && lane == 0) tab.put(FastTable<BYU16>::hash(src + ip - 2), ip - 2u);   /* :394 */
            wave_sync();
            const uint32_t sbase = test ? ip + 1u : ip;             /* where the search loop starts (:466) */
            uint32_t jbase = 0;
            uint32_t shift = test ? 1u : 0u;                        /* lane 0 of the first round = the test probe */
            uint32_t match = 0;
            bool found = false, test_hit = false;
            for (;;) {
                const bool is_test = shift != 0u && lane == 0;
                const uint32_t j = jbase + (uint32_t)lane - (is_test ? 0u : shift);
                const uint32_t pos = is_test ? ip : sbase + probe_offset(j, accel);
                const uint32_t npos = sbase + probe_offset(j + 1u, accel);
                const bool valid = is_test || (npos <= mflimit_plus_one && npos >= sbase);   /* :172 */
                uint32_t seq = 0, h = 0, cand = 0;
                if (valid) {
                    seq = ld32u(src + pos);
                    h = FastTable<BYU16>::hash(src + pos);
                    cand = tab.get(h);
                }
                bool hit = valid && (BYU16 || cand + (uint32_t)DISTANCE_MAX >= pos) && ld32u(src + cand) == seq;
                const unsigned long long vmask = __ballot(valid);
                const unsigned long long stop0 = __ballot(hit || !valid);
                const int W = stop0 ? ctz64(stop0) + 1 : 64;
                if (PROF) { n_round++; n_win += (unsigned long long)W; }

                /* in-window duplicates: a later lane must see the earlier lane's put */
                uint32_t pk = 0, rank = 0;
                for (int d = 1; d < W; d++) {
                    const uint32_t hk = __shfl_up(h, (unsigned)d);
                    if (valid && lane >= d && lane < W && hk == h) {
                        rank++;
                        if (pk == 0) pk = (uint32_t)d;
                    }
                }
                if (__ballot(pk != 0)) {
                    if (PROF) n_dup++;
                    const uint32_t ppos = __shfl(pos, lane - (int)pk);
                    if (pk != 0) {
                        cand = ppos;
                        hit = (BYU16 || cand + (uint32_t)DISTANCE_MAX >= pos) && ld32u(src + cand) == seq;
                    }
                }
                const unsigned long long stop = __ballot(lane < W && (hit || !valid));
                const int f = stop ? ctz64(stop) : W;
                const bool fvalid = stop ? ((vmask >> f) & 1ull) != 0 : false;
                const int ncommit = stop ? f + (fvalid ? 1 : 0) : W;

                /* commit puts of lanes < ncommit in lane order (:213, :420) */
                for (uint32_t r = 0;; r++) {
                    if (!__ballot(lane < ncommit && rank >= r)) break;
                    if (lane < ncommit && rank == r) tab.put(h, pos);
                    wave_sync();
                }
                if (!stop) {
                    jbase += (uint32_t)W - shift;
                    shift = 0;
                    continue;
                }
                if (!fvalid) break;                                 /* -> _last_literals */
                test_hit = shift != 0u && f == 0;
                ip = __shfl(pos, f);
                match = __shfl(cand, f);
                found = true;
                break;
            }
            if (!found) break;

            /* ---------------- extension: one round trip for both directions + literal bytes ---------------- */
            const unsigned long long t1 = prof_now<PROF>();
            const uint32_t lit0 = test_hit ? 0u : ip - anchor;
            const uint32_t maxback = test_hit ? 0u : (lit0 < match ? lit0 : match);
            const bool beq = (uint32_t)lane < maxback && src[ip - 1u - (uint32_t)lane] == src[match - 1u - (uint32_t)lane];
            const uint8_t litbyte = (uint32_t)lane < lit0 && lit0 <= 64u ? src[anchor + (uint32_t)lane] : (uint8_t)0;
            const uint32_t fwd_max = matchlimit - (ip + MINMATCH);
            uint32_t neq = 0;
            {
                const uint32_t i = 4u * (uint32_t)lane;
                if (i < fwd_max) {
                    const uint32_t x = ld32u(src + ip + MINMATCH + i) ^ ld32u(src + match + MINMATCH + i);
                    const uint32_t avail = fwd_max - i < 4u ? fwd_max - i : 4u;
                    const uint32_t e = x ? (uint32_t)(__ffs(x) - 1) >> 3 : 4u;
                    neq = e < avail ? e : avail;
                }
            }
            /* backward (:237-242) */
            uint32_t back = 0;
            if (maxback) {
                const unsigned long long ne = ~__ballot(beq);
                back = (uint32_t)(ne ? ctz64(ne) : 64);
                if (back == 64u) {
                    while (back < maxback) {
                        const uint32_t i = back + (uint32_t)lane;
                        const bool eq = i < maxback && src[ip - 1u - i] == src[match - 1u - i];
                        const unsigned long long ne2 = ~__ballot(eq);
                        const int run = ne2 ? ctz64(ne2) : 64;
                        back += (uint32_t)run;
                        if (run < 64) break;
                    }
                }
            }
            /* forward (:326-329); counting from ip+4 of the un-extended position, see header */
            uint32_t code;
            {
                const unsigned long long notfull = __ballot(neq != 4u);
                if (notfull) {
                    const int fl = ctz64(notfull);
                    code = 4u * (uint32_t)fl + __shfl(neq, fl);
                } else {
                    code = 256u + wave_count(src + ip + MINMATCH + 256u, src + match + MINMATCH + 256u, fwd_max - 256u, lane);
                }
            }
            const uint32_t ip_end = ip + MINMATCH + code;
            ip -= back;
            match -= back;
            code += back;

            const unsigned long long t2 = prof_now<PROF>();
            /* ---------------- emit (:244-382) ---------------- */
            uint32_t token_pos = (uint32_t)op;
            uint32_t token = 0;
            op++;
            if (!test_hit) {
                const uint32_t lit = ip - anchor;
                if (limited && op + lit + (2 + 1 + LASTLITERALS) + lit / 255u > olimit) return 0;
                if (lit >= (uint32_t)RUN_MASK) {
                    token = (uint32_t)RUN_MASK << ML_BITS;
                    op = emit_length_run(dst, (uint32_t)op, lit - RUN_MASK, lane);
                } else {
                    token = lit << ML_BITS;
                }
                if (lit0 <= 64u) {
                    if ((uint32_t)lane < lit) dst[op + lane] = litbyte;
                } else {
                    wave_copy(dst + op, src + anchor, lit, lane);
                }
                op += lit;
            }
            if (lane == 0) {                                        /* offset (:299-304) */
                const uint32_t off = ip - match;
                dst[op] = (uint8_t)off;
                dst[op + 1] = (uint8_t)(off >> 8);
            }
            op += 2;
            if (limited && op + (1 + LASTLITERALS) + (code + 240u) / 255u > olimit) return 0;
            if (code >= (uint32_t)ML_MASK) {
                token += ML_MASK;
                op = emit_length_run(dst, (uint32_t)op, code - ML_MASK, lane);
            } else {
                token += code;
            }
            if (lane == 0) dst[token_pos] = (uint8_t)token;
            ip = ip_end;
            anchor = ip;
            if (PROF) { const unsigned long long t3 = prof_now<PROF>(); c_probe += t1 - t0; c_ext += t2 - t1; c_emit += t3 - t2; n_seq++; }
            if (ip >= mflimit_plus_one) break;                      /* :391 */
            test = true;
        }
    }

    /* ---- _last_literals (:469-503) ---- */
    {
        const uint32_t last_run = U - anchor;
        if (limited && op + last_run + 1 + (last_run + 255u - RUN_MASK) / 255u > olimit) return 0;
        if (last_run >= (uint32_t)RUN_MASK) {
            if (lane == 0) dst[op] = (uint8_t)(RUN_MASK << ML_BITS);
            op = emit_length_run(dst, (uint32_t)op + 1u, last_run - RUN_MASK, lane);
        } else {
            if (lane == 0) dst[op] = (uint8_t)(last_run << ML_BITS);
            op++;
        }
        wave_copy(dst + op, src + anchor, last_run, lane);
        op += last_run;
    }
    if (PROF && pc && lane == 0) {
        pc[0] = prof_now<PROF>() - t_begin; pc[1] = c_probe; pc[2] = c_ext; pc[3] = c_emit;
        pc[4] = n_seq; pc[5] = n_round; pc[6] = n_dup; pc[7] = n_win;
    }
    prof_place<PROF>(pc, 9, lane);
    return (int)op;
}

/* LL64.LZ4_compress_fast (LL64.fast.cs:517-576): table type by input size */
__device__ __forceinline__ int compress_fast_block(const uint8_t *src, int src_len, uint8_t *dst, int dst_cap,
                                                   int accel, uint32_t *tabw, int lane)
{
    const uint32_t a = accel < 1 ? 1u : (accel > 65536 ? 65536u : (uint32_t)accel);
    if (src_len < LIMIT_64K) return encode_fast_block<true>(src, src_len, dst, dst_cap, a, tabw, lane);
    return encode_fast_block<false>(src, src_len, dst, dst_cap, a, tabw, lane);
}

/* LZ4Codec.Encode mapping (LZ4Codec.cs:40-52) */
__device__ __forceinline__ int codec_encode_result(int src_len, int ret, int flags)
{
    if (flags & FLAG_RAW_RETURN) return ret;
    if (src_len <= 0) return 0;
    return ret <= 0 ? -1 : ret;
}

__global__ __launch_bounds__(64) void k4_encode_fast_kernel(BatchArgs a)
{
    __shared__ uint32_t tab[4096];
    const int lane = lane_id();
    const long long b = (long long)blockIdx.x;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    const uint8_t *src = a.src + a.srcOff[b];
    uint8_t *dst = a.dst + a.dstOff[b];
    int ret = 0;
    if (src_len > 0 || (a.flags & FLAG_RAW_RETURN)) ret = compress_fast_block(src, src_len, dst, cap < 0 ? 0 : cap, a.accel, tab, lane);
    if (lane == 0) a.outLen[b] = codec_encode_result(src_len, ret, a.flags);
}

/* diagnostic twin (blocks < 65547 B only): per-phase cycle counters (a.prof, 8 per block) */
__global__ __launch_bounds__(64) void k4_encode_fast_prof_kernel(BatchArgs a)
{
    __shared__ uint32_t tab[4096];
    const int lane = lane_id();
    const long long b = (long long)blockIdx.x;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    int ret = 0;
    if (src_len > 0 && src_len < LIMIT_64K)
        ret = encode_fast_block<true, true>(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, 1u, tab, lane, a.prof + PROF_STRIDE * b);
    if (lane == 0) a.outLen[b] = codec_encode_result(src_len, ret, a.flags);
}

}  // namespace k4
