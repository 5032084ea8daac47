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
    /* same hash from bytes already in registers: seq = bytes p..p+3, next = bytes p+4..p+11 */
    __device__ __forceinline__ static uint32_t hash_of(uint32_t seq, uint64_t) { return (seq * 2654435761u) >> (32 - 13); }
    __device__ __forceinline__ uint32_t get(uint32_t h) const { return t[h]; }
    __device__ __forceinline__ void put(uint32_t h, uint32_t pos) const { t[h] = (uint16_t)pos; }
};
template <> struct FastTable<false> {  /* byU32: 4096 x u32, hash5 (LL.tools.cs:53-58, LL64.tools.cs:135-143) */
    uint32_t *t;
    __device__ __forceinline__ static uint32_t hash(const uint8_t *p)
    {
        return (uint32_t)(((ld64u(p) << 24) * 889523592379ull) >> (64 - 12));
    }
    __device__ __forceinline__ static uint32_t hash_of(uint32_t seq, uint64_t next)
    {
        return (uint32_t)((((next << 32) | seq) << 24) * 889523592379ull >> (64 - 12));
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
        return done + 4u * (uint32_t)fl + __builtin_amdgcn_readlane(neq, fl);
    }
}

constexpr int ENCODE_STAGE_BYTES = 2048;
constexpr int ENCODE_LDS_DWORDS = 4096 + ENCODE_STAGE_BYTES / 4;   /* hash table + output stage */

/* Output staging: the compressed stream is assembled in LDS and written to HBM in 16 B per lane
 * bursts.  Besides coalescing the byte-granular token/offset/length stores, this keeps the wave's
 * in-order memory queue free of stores, so the probe loads of the next sequence are not held up
 * behind write acknowledgements. */
struct OutStage {
    uint8_t *lds;      /* ENCODE_STAGE_BYTES, 16-byte aligned */
    uint8_t *dst;      /* block output in global memory */
    uint32_t base;     /* output position held by lds[0] */
    bool dry;          /* cost estimation run: nothing leaves the stage */

    __device__ __forceinline__ void flush_to(uint32_t op, int lane)
    {
        const uint32_t n = op - base;
        if (dry) { base = uni(op); return; }
        wave_sync();
        for (uint32_t k = 16u * (uint32_t)lane; k < n; k += 1024u) {
            if (k + 16u <= n) {
                const uint4 v = *(const uint4 *)(lds + k);
                U128u o;
                o.v[0] = v.x; o.v[1] = v.y; o.v[2] = v.z; o.v[3] = v.w;
                st128u(dst + base + k, o);
            } else {
                for (uint32_t t = k; t < n; t++) dst[base + t] = lds[t];
            }
        }
        wave_sync();
        base = uni(op);
    }
    /* make room for `need` more staged bytes at output position op */
    __device__ __forceinline__ void reserve(uint32_t op, uint32_t need, int lane)
    {
        if (op - base + need > (uint32_t)ENCODE_STAGE_BYTES) flush_to(op, lane);
    }
    __device__ __forceinline__ uint8_t *at(uint32_t op) const { return lds + (op - base); }
};

/* length field tail: `rem` encoded as 255-run + final byte (LL64.fast.cs:262-272,:365-381,:484-495) */
__device__ __forceinline__ uint32_t emit_length_run(OutStage &st, uint32_t op, uint32_t rem, int lane)
{
    const uint32_t nb = rem / 255u;
    if (nb > 256u) {                       /* multi-KiB run: straight to global memory */
        st.flush_to(op, lane);
        if (!st.dry) {
            wave_fill(st.dst + op, 255, nb, lane);
            if (lane == 0) st.dst[op + nb] = (uint8_t)(rem - nb * 255u);
        }
        st.base = uni(op + nb + 1u);
        return uni(op + nb + 1u);
    }
    st.reserve(op, nb + 1u, lane);
    wave_fill(st.at(op), 255, nb, lane);
    if (lane == 0) *st.at(op + nb) = (uint8_t)(rem - nb * 255u);
    return uni(op + nb + 1u);
}

/* the 16 source bytes around position p: 4 before, the 4 compared ones, 8 after */
struct Around {
    uint32_t pre, seq;
    uint64_t next;
    bool pre_ok;
};
__device__ __forceinline__ Around load_around(const uint8_t *src, uint32_t p)
{
    Around a;
    a.pre_ok = p >= 4u;
    if (a.pre_ok) {
        const U128u v = ld128u(src + p - 4u);
        a.pre = v.v[0]; a.seq = v.v[1]; a.next = ((uint64_t)v.v[3] << 32) | v.v[2];
    } else {
        a.pre = 0; a.seq = ld32u(src + p); a.next = ld64u(src + p + 4u);
    }
    return a;
}

/*
 * LL64.LZ4_compress_generic for one block.  `ldsw`: ENCODE_LDS_DWORDS dwords of LDS owned by this
 * wave (16 KiB hash table, zeroed here = LZ4_initStream, LL.tools.cs:235-239; then the output
 * stage).  Returns bytes written, 0 when the output does not fit.
 *
 * One iteration of the main loop produces one sequence with two dependent memory round trips:
 *   (1) 16 source bytes around each of up to 64 probe positions (4 before, the 4 hashed ones, 8
 *       after) and the pending literal bytes                      -> hashes -> LDS table lookups
 *   (2) the 16 bytes around the 64 candidates                     -> first hit, and from the hit
 *       lane's registers the backward extension (up to 4) and the match length (up to 12)
 * Longer extensions take a third round trip (wave_count).  The reference's "test the position
 * right after a match" step (LL64.fast.cs:393-463) rides in lane 0 of the next probe round.
 */
template <bool BYU16, bool PROF = false>
__device__ __forceinline__ int encode_fast_block(const uint8_t *src, int src_len, uint8_t *dst, int dst_cap,
                                                 uint32_t accel, uint32_t *ldsw, int lane, unsigned long long *pc = nullptr,
                                                 bool dry = false, uint32_t *seq_count = nullptr, uint32_t *gtab = nullptr)
{
    uint32_t sequences = 0;
    unsigned long long c_probe = 0, c_ext = 0, c_emit = 0, n_seq = 0, n_round = 0, n_dup = 0, n_rt3 = 0;
    unsigned long long c_s1 = 0, c_s2 = 0, c_s3 = 0, c_s4 = 0;
    prof_place<PROF>(pc, 8, lane);
    const unsigned long long t_begin = prof_now<PROF>();
    if ((uint32_t)src_len > (uint32_t)MAX_INPUT_SIZE) return 0;     /* LL64.fast.cs:90 */
    const bool limited = dst_cap < compress_bound(src_len);         /* :524 */
    const int64_t olimit = dst_cap;
    const uint32_t U = (uint32_t)src_len;
    FastTable<BYU16> tab;
    /* the table normally lives in LDS; `gtab` (16 KiB of global memory) lets more blocks run per CU */
    uint32_t *const tabmem = gtab ? gtab : ldsw;
    tab.t = (decltype(tab.t))tabmem;
    OutStage st;
    st.lds = (uint8_t *)(gtab ? ldsw : ldsw + 4096);
    st.dst = dst;
    st.base = 0;
    st.dry = dry;

    for (int k = lane; k < 1024; k += 64) ((uint4 *)tabmem)[k] = make_uint4(0u, 0u, 0u, 0u);
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
            const uint32_t sbase = test ? ip + 1u : ip;             /* where the search loop starts (:466) */
            uint32_t shift = test ? 1u : 0u;                        /* lane 0 of the first round = the test probe */
            /* first round: positions do not depend on the table, so their source bytes (and the
             * bytes hashed for the ip-2 put) are requested before the table is touched */
            uint32_t pos, npos;
            {
                const bool is_test0 = shift != 0u && lane == 0;
                const uint32_t j0 = (uint32_t)lane - (is_test0 ? 0u : shift);
                if (accel == 1u) {                                  /* first 64 probes of the schedule: step 1 */
                    pos = is_test0 ? ip : sbase + j0;
                    npos = sbase + j0 + 1u;
                } else {
                    pos = is_test0 ? ip : sbase + probe_offset(j0, accel);
                    npos = sbase + probe_offset(j0 + 1u, accel);
                }
            }
            bool valid = (shift != 0u && lane == 0) || (npos <= mflimit_plus_one && npos >= sbase);   /* :172 */
            Around pa, ca;
            pa.pre = 0; pa.seq = 0; pa.next = 0; pa.pre_ok = false;
            if (valid) pa = load_around(src, pos);
            const uint8_t litbyte = anchor + (uint32_t)lane < U ? src[anchor + (uint32_t)lane] : (uint8_t)0;
            if (test) {
                const uint32_t h2 = FastTable<BYU16>::hash(src + ip - 2);
                if (lane == 0) tab.put(h2, ip - 2u);                /* :394 */
            }
            wave_sync();
            unsigned long long ts = prof_now<PROF>();
            if (PROF) c_s1 += ts - t0;
            uint32_t jbase = 0;
            uint32_t match = 0;
            bool found = false, test_hit = false;
            int f = 0;
            for (bool first = true;; first = false) {
                if (!first) {
                    const uint32_t j = jbase + (uint32_t)lane;
                    pos = sbase + probe_offset(j, accel);
                    npos = sbase + probe_offset(j + 1u, accel);
                    valid = npos <= mflimit_plus_one && npos >= sbase;
                    pa.pre = 0; pa.seq = 0; pa.next = 0; pa.pre_ok = false;
                    if (valid) pa = load_around(src, pos);
                }
                uint32_t h = 0, cand = 0;
                if (valid) {
                    h = FastTable<BYU16>::hash_of(pa.seq, pa.next);
                    cand = tab.get(h);
                }
                if (PROF) { const unsigned long long tn = prof_now<PROF>(); c_s2 += tn - ts; ts = tn; }
                ca = load_around(src, cand);
                bool hit = valid && (BYU16 || cand + (uint32_t)DISTANCE_MAX >= pos) && ca.seq == pa.seq;
                const unsigned long long vmask = __ballot(valid);
                const unsigned long long stop0 = __ballot(hit || !valid);
                if (PROF) { const unsigned long long tn = prof_now<PROF>(); c_s3 += tn - ts; ts = tn; }
                const int W = stop0 ? ctz64(stop0) + 1 : 64;
                if (PROF) n_round++;

                /* in-window duplicates: a later lane must see the earlier lane's put */
                uint32_t pk = 0, rank = 0;
                for (int d = 1; d < W; d++) {
                    const uint32_t hk = __shfl_up(h, (unsigned)d);
                    if (valid && lane >= d && lane < W && hk == h) {
                        rank++;
                        if (pk == 0) pk = (uint32_t)d;
                    }
                }
                const unsigned long long dupmask = __ballot(pk != 0);
                if (dupmask) {
                    if (PROF) n_dup++;
                    const uint32_t ppos = __shfl(pos, lane - (int)pk);
                    if (pk != 0) {
                        cand = ppos;
                        ca = load_around(src, cand);
                        hit = (BYU16 || cand + (uint32_t)DISTANCE_MAX >= pos) && ca.seq == pa.seq;
                    }
                }
                const unsigned long long stop = __ballot(lane < W && (hit || !valid));
                f = stop ? ctz64(stop) : W;
                const bool fvalid = stop ? ((vmask >> f) & 1ull) != 0 : false;
                const int ncommit = stop ? f + (fvalid ? 1 : 0) : W;

                /* commit puts of lanes < ncommit in lane order (:213, :420) */
                if ((dupmask & ((ncommit >= 64 ? 0ull : (1ull << ncommit)) - 1ull)) == 0) {
                    if (lane < ncommit) tab.put(h, pos);
                    wave_sync();
                } else {
                    for (uint32_t r = 0;; r++) {
                        if (!__ballot(lane < ncommit && rank >= r)) break;
                        if (lane < ncommit && rank == r) tab.put(h, pos);
                        wave_sync();
                    }
                }
                if (PROF) { const unsigned long long tn = prof_now<PROF>(); c_s4 += tn - ts; ts = tn; }
                if (!stop) {
                    jbase += (uint32_t)W - shift;
                    shift = 0;
                    continue;
                }
                if (!fvalid) break;                                 /* -> _last_literals */
                test_hit = shift != 0u && f == 0;
                ip = __builtin_amdgcn_readlane(pos, f);
                match = __builtin_amdgcn_readlane(cand, f);
                found = true;
                break;
            }
            if (!found) break;

            /* ---------------- extension from the hit lane's registers ---------------- */
            const unsigned long long t1 = prof_now<PROF>();
            const uint32_t lit0 = test_hit ? 0u : ip - anchor;
            const uint32_t maxback = test_hit ? 0u : (lit0 < match ? lit0 : match);
            const uint32_t fwd_max = matchlimit - (ip + MINMATCH);
            uint32_t back = 0, code;
            {
                const uint32_t a_pre = __builtin_amdgcn_readlane(pa.pre, f), b_pre = __builtin_amdgcn_readlane(ca.pre, f);
                const uint32_t a_lo = __builtin_amdgcn_readlane((uint32_t)pa.next, f), a_hi = __builtin_amdgcn_readlane((uint32_t)(pa.next >> 32), f);
                const uint32_t b_lo = __builtin_amdgcn_readlane((uint32_t)ca.next, f), b_hi = __builtin_amdgcn_readlane((uint32_t)(ca.next >> 32), f);
                const bool pre_ok = (__ballot(pa.pre_ok && ca.pre_ok) >> f) & 1ull;
                /* forward (:326-329): bytes ip+4.. vs match+4.. */
                const uint64_t x = (((uint64_t)a_hi << 32) | a_lo) ^ (((uint64_t)b_hi << 32) | b_lo);
                const uint32_t e = x ? (uint32_t)(__ffsll((unsigned long long)x) - 1) >> 3 : 8u;
                if (e == 8u && fwd_max > 8u) {
                    if (PROF) n_rt3++;
                    code = 8u + wave_count(src + ip + MINMATCH + 8u, src + match + MINMATCH + 8u, fwd_max - 8u, lane);
                } else {
                    code = e < fwd_max ? e : fwd_max;
                }
                /* backward (:237-242): bytes ip-1, ip-2, .. vs match-1, .. */
                if (maxback) {
                    uint32_t nb = 0;
                    if (pre_ok) {
                        const uint32_t y = a_pre ^ b_pre;
                        nb = y ? (uint32_t)__clz(y) >> 3 : 4u;
                        back = nb < maxback ? nb : maxback;
                    }
                    if ((!pre_ok || nb == 4u) && back < maxback) {
                        if (PROF) n_rt3++;
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
            }
            const uint32_t ip_end = ip + MINMATCH + code;
            ip -= back;
            match -= back;
            code += back;

            /* ---------------- emit into the LDS stage (:244-382) ---------------- */
            const unsigned long long t2 = prof_now<PROF>();
            st.reserve((uint32_t)op, 1u + 64u + 2u, lane);
            uint32_t token_pos = (uint32_t)op;
            uint32_t token = 0;
            op++;
            if (!test_hit) {
                const uint32_t lit = ip - anchor;
                if (limited && op + lit + (2 + 1 + LASTLITERALS) + lit / 255u > olimit) return 0;
                if (lit >= (uint32_t)RUN_MASK) {
                    token = (uint32_t)RUN_MASK << ML_BITS;
                    op = emit_length_run(st, (uint32_t)op, lit - RUN_MASK, lane);
                } else {
                    token = lit << ML_BITS;
                }
                if (lit <= 64u) {
                    st.reserve((uint32_t)op, 64u + 2u, lane);
                    if ((uint32_t)lane < lit) *st.at((uint32_t)op + (uint32_t)lane) = litbyte;
                } else {
                    st.flush_to((uint32_t)op, lane);
                    if (!dry) wave_copy(dst + op, src + anchor, lit, lane);
                    st.base = (uint32_t)op + lit;
                }
                op += lit;
            }
            if (token_pos < st.base) {                              /* token already left the stage */
                st.flush_to((uint32_t)op, lane);
            }
            st.reserve((uint32_t)op, 2u, lane);
            if (lane == 0) {                                        /* offset (:299-304) */
                const uint32_t off = ip - match;
                uint8_t *o = st.at((uint32_t)op);
                o[0] = (uint8_t)off;
                o[1] = (uint8_t)(off >> 8);
            }
            op += 2;
            if (limited && op + (1 + LASTLITERALS) + (code + 240u) / 255u > olimit) return 0;
            if (code >= (uint32_t)ML_MASK) {
                token += ML_MASK;
                op = emit_length_run(st, (uint32_t)op, code - ML_MASK, lane);
            } else {
                token += code;
            }
            if (lane == 0) {
                if (token_pos >= st.base) *st.at(token_pos) = (uint8_t)token;
                else if (!dry) dst[token_pos] = (uint8_t)token;
            }
            ip = ip_end;
            anchor = ip;
            sequences++;
            if (PROF) { const unsigned long long t3 = prof_now<PROF>(); c_probe += t1 - t0; c_ext += t2 - t1; c_emit += t3 - t2; n_seq++; }
            if (ip >= mflimit_plus_one) break;                      /* :391 */
            test = true;
        }
    }

    /* ---- _last_literals (:469-503) ---- */
    {
        const uint32_t last_run = U - anchor;
        if (limited && op + last_run + 1 + (last_run + 255u - RUN_MASK) / 255u > olimit) return 0;
        st.reserve((uint32_t)op, 1u, lane);
        if (last_run >= (uint32_t)RUN_MASK) {
            if (lane == 0) *st.at((uint32_t)op) = (uint8_t)(RUN_MASK << ML_BITS);
            op = emit_length_run(st, (uint32_t)op + 1u, last_run - RUN_MASK, lane);
        } else {
            if (lane == 0) *st.at((uint32_t)op) = (uint8_t)(last_run << ML_BITS);
            op++;
        }
        st.flush_to((uint32_t)op, lane);
        if (!dry) wave_copy(dst + op, src + anchor, last_run, lane);
        op += last_run;
    }
    if (seq_count) *seq_count = sequences;
    if (PROF && pc && lane == 0) {
        pc[0] = prof_now<PROF>() - t_begin; pc[1] = c_probe; pc[2] = c_ext; pc[3] = c_emit;
        pc[4] = n_seq; pc[5] = n_round; pc[6] = n_dup; pc[7] = n_rt3;
        pc[11] = c_s1; pc[12] = c_s2; pc[13] = c_s3; pc[14] = c_s4;
    }
    prof_place<PROF>(pc, 9, lane);
    return (int)op;
}

/* LL64.LZ4_compress_fast (LL64.fast.cs:517-576): table type by input size */
__device__ __forceinline__ int compress_fast_block(const uint8_t *src, int src_len, uint8_t *dst, int dst_cap,
                                                   int accel, uint32_t *ldsw, int lane, uint32_t *gtab = nullptr)
{
    const uint32_t a = accel < 1 ? 1u : (accel > 65536 ? 65536u : (uint32_t)accel);
    if (src_len < LIMIT_64K) return encode_fast_block<true>(src, src_len, dst, dst_cap, a, ldsw, lane, nullptr, false, nullptr, gtab);
    return encode_fast_block<false>(src, src_len, dst, dst_cap, a, ldsw, lane, nullptr, false, nullptr, gtab);
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
constexpr int COST_SAMPLE = 2048;
constexpr int COST_BUCKETS = 64;

__device__ __forceinline__ uint32_t cost_bucket(unsigned long long cost)
{
    if (cost < 4ull) return (uint32_t)cost;
    const uint32_t l = 63u - (uint32_t)__clzll(cost);            /* 2 buckets per octave */
    const uint32_t bkt = 2u * l + (uint32_t)((cost >> (l - 1u)) & 1ull);
    return bkt < (uint32_t)COST_BUCKETS ? bkt : (uint32_t)COST_BUCKETS - 1u;
}

/* a.cost[b] = bucket of block b; a.hist[bucket]++.  mode 0: encoder sample, mode 1: by length */
__global__ __launch_bounds__(64) void k4_cost_kernel(BatchArgs a, int by_length)
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
            const int sample = src_len < COST_SAMPLE ? src_len : COST_SAMPLE;
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

/* order[] = block indices, most expensive bucket first */
__global__ __launch_bounds__(256) void k4_order_kernel(BatchArgs a)
{
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= a.n) return;
    const uint32_t bkt = a.cost[b];
    uint32_t before = 0;
    for (uint32_t k = bkt + 1u; k < (uint32_t)COST_BUCKETS; k++) before += a.hist[k];
    const uint32_t pos = before + atomicAdd(&a.hist[COST_BUCKETS + bkt], 1u);
    a.order_out[pos] = (uint32_t)b;
}

__global__ __launch_bounds__(64) void k4_encode_fast_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[ENCODE_LDS_DWORDS];
    const int lane = lane_id();
    const long long b = a.order ? (long long)uni(a.order[blockIdx.x]) : (long long)blockIdx.x;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    const uint8_t *src = a.src + a.srcOff[b];
    uint8_t *dst = a.dst + a.dstOff[b];
    int ret = 0;
    if (src_len > 0 || (a.flags & FLAG_RAW_RETURN)) ret = compress_fast_block(src, src_len, dst, cap < 0 ? 0 : cap, a.accel, tab, lane);
    if (lane == 0) a.outLen[b] = codec_encode_result(src_len, ret, a.flags);
}

/* the same encoder with its hash table in global memory (a.gtab: 16 KiB per workgroup slot) and
 * only the output stage in LDS: twice as many blocks resident per CU, each a little slower */
__global__ __launch_bounds__(64) void k4_encode_fast_gtab_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t stage[ENCODE_STAGE_BYTES / 4];
    const int lane = lane_id();
    const long long b = a.order ? (long long)uni(a.order[blockIdx.x]) : (long long)blockIdx.x;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    int ret = 0;
    if (src_len > 0 || (a.flags & FLAG_RAW_RETURN))
        ret = compress_fast_block(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, a.accel, stage, lane,
                                  a.gtab + 4096ull * (unsigned long long)blockIdx.x);
    if (lane == 0) a.outLen[b] = codec_encode_result(src_len, ret, a.flags);
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
        ret = encode_fast_block<true, true>(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, 1u, tab, lane, a.prof + PROF_STRIDE * b);
    if (lane == 0) a.outLen[b] = codec_encode_result(src_len, ret, a.flags);
}

}  // namespace k4
