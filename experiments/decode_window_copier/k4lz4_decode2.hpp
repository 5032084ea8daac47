/*
 * k4lz4_decode2.hpp -- the pair decoder's second generation (round 6): window parser + byte-parallel copier.
 *
 * Same job and same reference as k4lz4_decode.hpp (LL64.LZ4_decompress_safe, Engine/x64/LL64.dec.cs:123-477: accept / reject,
 * error position and bytes), same pairing (one wavefront parses a block, its partner writes it, a queue in LDS between them),
 * but both halves are organised by BYTES instead of by sequences, because the blocks that set a launch's length are the ones
 * with ~7 output bytes per sequence:
 *
 *   PARSE   W windows of 64 stream bytes per round.  Every lane of every window assumes that a token starts at its byte and works
 *           out that hypothetical sequence (the hypotheses and their rules are k4lz4_decode.hpp's, unchanged).  The first
 *           generation then walked the true chain with one v_readlane per sequence (~65 cycles each, 15 in a row for a window of
 *           text).  Here every lane learns by pointer doubling -- five ds_bpermute steps, all lanes and all windows at once --
 *           the SET of token lanes its own chain visits inside its window (a 64-bit mask) and the lane it leaves through; the
 *           serial part is then one hop per WINDOW: read the mask of the entry lane, read where its chain leaves, that is the next
 *           window's entry.  Output positions come from one prefix sum per window; the position-dependent rules (LL64.dec.cs:191,
 *           :247,:338,:427-433) are checked on the chosen lanes as before, and the first sequence that is not plain -- 270 and more
 *           literals, a match of 274 and more, offset 0, the neighbourhood of either end, anything malformed -- goes through the
 *           scalar parser that follows the reference check by check.  What the parser hands on is 16 bytes per sequence: where
 *           its output begins, where its literals lie in the stream, how many they are, the offset.  (The match length is the
 *           distance to the next record.)
 *   COPY    64 output bytes per step, one per lane, whatever the number of sequences they belong to.  A lane finds its record
 *           (the records that start inside the window mark their first byte in a 64-byte LDS array; a ballot of the marks and
 *           a population count give every lane its record's index), and its byte is then either a literal (stream, global
 *           memory), or a match byte whose source was written long ago (global memory: everything below the flushed mark), or
 *           recently (the last 2 KiB of output live in an LDS ring), or inside this very window -- those are resolved by
 *           pointer doubling over the lanes, which is also what makes an overlapping match (offset < length, LL64.dec.cs:408-450:
 *           byte-serial semantics) the same code as any other.  The loads from global memory are issued PD windows ahead of their
 *           use.  The ring leaves for memory 1 KiB at a time, 16 bytes per lane.
 *           A step costs the same ~45 wave instructions for 64 bytes of text (9 sequences) and for 64 bytes of one long literal
 *           run; the first generation spent ~16 per sequence.
 *
 * Not here: partial decoding, dictionaries (the first generation keeps those arms), blocks whose parse needs them.
 */
#pragma once
#include "k4lz4_decode.hpp"

namespace k4 {

#ifndef K4_X_WINDOWS
#define K4_X_WINDOWS 2
#endif
#ifndef K4_X_DEPTH
#define K4_X_DEPTH 2
#endif
constexpr int XQ = 256;                          /* records in a pair's queue (power of two) */
constexpr uint32_t XQ_MASK = XQ - 1;
constexpr int XHIST = 2048;                      /* bytes of output history in LDS (power of two) */
constexpr int XFLUSH = 1024;                     /* ... of which this many leave for memory at a time */
/* a pair's LDS, in dwords: the parser's stream ring, the control words, the queue (records, and their first word once more as
 * an array of its own: what the copier scans), the history ring, the marks */
constexpr int X_PIPE = RING_DWORDS, X_QUEUE = X_PIPE + 16, X_QOUT = X_QUEUE + 4 * XQ, X_HIST = X_QOUT + XQ,
              X_FLAGS = X_HIST + XHIST / 4 + 16, X_PAIR_DWORDS = X_FLAGS + 16;
static_assert((X_QUEUE * 4) % 16 == 0 && (X_HIST * 4) % 16 == 0 && (X_PAIR_DWORDS * 4) % 16 == 0, "16-byte LDS accesses");
/* control words: [0] records published  [1] records the copier is done with  [2] the block is parsed  [3] its result
 * [6..7] the context's status word (pipe_init) */

/* orders this wave's LDS accesses for the COMPILER only: the hardware executes a wave's LDS instructions in order, so a store
 * by one lane is seen by a later load of another lane without a wait in between (the emulator's lanes need the rendezvous) */
__device__ __forceinline__ void lds_order()
{
#ifndef K4_HOST_EMU
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#else
    wave_sync();
#endif
}

/* a control word the partner polls (no ordering asked for: LDS executes this wave's accesses in order) */
__device__ __forceinline__ void xpipe_post(uint32_t *p, uint32_t v)
{
#ifndef K4_HOST_EMU
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    *(volatile uint32_t *)p = v;
#endif
}

/* poll pipe[which] until it is >= want; returns its value, or 0xffffffff after PIPE_SPIN_MAX polls (status raised) */
__device__ __forceinline__ uint32_t xpipe_wait(const uint32_t *pipe, int which, uint32_t want)
{
    for (uint32_t spin = 0; spin < PIPE_SPIN_MAX; spin++) {
        const uint32_t v = pipe_load(pipe + which);
        if (v >= want) return v;
        __builtin_amdgcn_s_sleep(1);
    }
    dev_status_raise((uint32_t *)(uintptr_t)((unsigned long long)pipe[6] | ((unsigned long long)pipe[7] << 32)), (uint32_t)DEV_STATUS_PIPE_TIMEOUT);
    return 0xffffffffu;
}

/* the two degenerate calls of LL64.dec.cs:162-172, the same for both waves of a pair; true: *ret is the block's result */
__device__ __forceinline__ bool xdecode_trivial(const uint8_t *in, int src_size, int out_size, int lane, int *ret)
{
    if (out_size == 0) {                                   /* LL64.dec.cs:162-168 */
        if (src_size == 1) {
            const uint32_t b = uni(lane == 0 ? (uint32_t)in[0] : 0u);
            *ret = b == 0 ? 0 : -1;
        } else {
            *ret = -1;
        }
        return true;
    }
    if (src_size <= 0) { *ret = -1; return true; }         /* :172 */
    return false;
}

/* per-lane words of a hypothesis: A = offset | match length << 16;
 * B = literals (9 bits) | two-byte header << 9 | leaves the shortcut's match stage << 11 | usable << 12 | lane of the next token << 16 */
constexpr uint32_t XB_G = 1u << 9, XB_GENERAL = 1u << 11, XB_USABLE = 1u << 12;

/*
 * PARSE (the pair's first wave).  Returns the block's result as LL64.LZ4_decompress_safe would: bytes written, or
 * -(input position) - 1.  The copier is told through the control words.
 */
template <int W, bool PROF = false>
__device__ __forceinline__ int xparse_block(const uint8_t *in, int src_size, int out_size, int lane, uint32_t *lds, unsigned long long *pc = nullptr)
{
    unsigned long long c_room = 0, n_round = 0, n_scalar = 0, c_hyp = 0, c_dbl = 0, c_ent = 0, c_rules = 0, c_scalar = 0;
    const unsigned long long t_begin = PROF ? (unsigned long long)__builtin_readcyclecounter() : 0ull;
    auto now = [&]() -> unsigned long long { return PROF ? (unsigned long long)__builtin_readcyclecounter() : 0ull; };
    uint32_t *pipe = lds + X_PIPE;
    uint4 *queue = (uint4 *)(lds + X_QUEUE);
    uint32_t *qout = lds + X_QOUT;
    {
        int r;
        if (xdecode_trivial(in, src_size, out_size, lane, &r)) return r;
    }
    StreamRing win;
    win.init(lds, in, (uint32_t)src_size, lane);

    const int64_t iend = src_size;
    const int64_t oend = out_size;
    const int64_t shortiend = iend - 14 - 2;               /* :152 */
    const int64_t shortoend = oend - 14 - 18;              /* :153 */
    const uint32_t iendu = (uint32_t)iend, oendu = (uint32_t)oend;
    int64_t ip = 0, op = 0;
    uint32_t head = 0, tail_seen = 0;                      /* records published; what the copier was last seen to be done with */
    const uint32_t bit_lo = lane < 32 ? 1u << lane : 0u, bit_hi = lane >= 32 ? 1u << (lane - 32) : 0u;

    /* room for n more records and the end mark behind them */
    auto room = [&](uint32_t n) -> bool {
        const uint32_t need = head + n + 1u;
        if (need - tail_seen <= (uint32_t)XQ) return true;
        const unsigned long long t0 = now();
        tail_seen = xpipe_wait(pipe, 1, need - (uint32_t)XQ);
        if (PROF) c_room += now() - t0;
        return tail_seen != 0xffffffffu;
    };
    /* the records up to `head` are complete and the output continues at `op`: tell the copier */
    auto publish = [&]() {
        lds_order();
        if (lane == 0) {
            qout[head & XQ_MASK] = (uint32_t)op;
            xpipe_post(pipe + 0, head);
        }
        lds_order();
    };
    auto finish = [&](int result) -> int {
        publish();
        if (lane == 0) {
            pipe[3] = (uint32_t)result;
            xpipe_post(pipe + 2, 1u);
        }
        lds_order();
        if (PROF && pc && lane == 0) {
            pc[0] = now() - t_begin; pc[1] = c_room; pc[2] = n_round; pc[3] = n_scalar; pc[4] = head; pc[5] = c_hyp; pc[6] = c_dbl; pc[7] = c_ent;
            pc[11] = c_rules; pc[12] = c_scalar;
        }
        return result;
    };
    /* LZ4_readVLE, a wave-full of bytes at a time: k4lz4_decode.hpp */
    auto read_vle = [&](uint32_t &length, int64_t lencheck) -> bool {
        for (;;) {
            win.ensure((uint32_t)ip + win.a0, lane);
            const uint32_t b = win.read4((uint32_t)ip + win.a0 + (uint32_t)lane) & 0xffu;
            const unsigned long long nz = ballot(b != 255u);
            uint32_t n = nz ? (uint32_t)ctz64(nz) + 1u : 64u;
            const int64_t avail = lencheck - ip;
            if (avail < (int64_t)n) n = avail > 1 ? (uint32_t)avail : 1u;
            const uint32_t last = readlane_u32(b, (int)n - 1);
            length += 255u * (n - 1u) + last;
            ip += n;
            if (ip >= lencheck) return true;
            if (last != 255u) return false;
        }
    };

    for (;;) {
        const uint32_t ipu = (uint32_t)ip, opu = (uint32_t)op;
        bool scalar = true;
        if (ipu + (uint32_t)RUN_MASK + 1u < iendu) {       /* some hypothesis can be usable */
            K4_PHASE("x-hyp");
            const unsigned long long tp0 = now();
            if (PROF) n_round++;
            win.ensure_ahead<(64 * W + 384) / 4>(ipu + win.a0, lane);
            uint32_t A[W], B[W], Mlo[W], Mhi[W], J[W];
#pragma unroll
            for (int k = 0; k < W; k++) {
                /* the hypotheses of k4lz4_decode.hpp's speculative round, for the window that begins at ip + 64 k */
                const uint32_t pl = ipu + 64u * (uint32_t)k + (uint32_t)lane;
                const uint32_t q = pl + win.a0;
                const uint32_t t4 = win.read4(q);
                const uint32_t L0 = (t4 >> 4) & 15u;
                const uint32_t M = t4 & 15u;
                const bool cls_g = L0 == RUN_MASK;
                const uint32_t ext_l = (t4 >> 8) & 0xffu;
                const uint32_t L = cls_g ? L0 + ext_l : L0;
                const uint32_t hdr = cls_g ? 2u : 1u;
                bool fast = cls_g ? (pl + (uint32_t)RUN_MASK + 1u < iendu && ext_l != 255u && pl + hdr + L + (2u + 1u + LASTLITERALS) <= iendu)
                                  : pl + 1u + 14u + 2u < iendu;
                const uint32_t o4 = win.read4(q + hdr + L);
                const uint32_t offset = o4 & 0xffffu;
                const bool m_ext = M == ML_MASK;
                const uint32_t ext_m = (o4 >> 16) & 0xffu;
                const uint32_t fwd = hdr + L + (m_ext ? 3u : 2u);                 /* stream bytes of the sequence */
                const uint32_t mlen = M + MINMATCH + (m_ext ? ext_m : 0u);
                fast = fast && offset != 0u && (!m_ext || (ext_m != 255u && pl + fwd + (LASTLITERALS - 1u) < iendu));
                const bool general = cls_g || m_ext || offset < 8u;
                const uint32_t nxt = (uint32_t)lane + fwd;
                A[k] = offset | (mlen << 16);
                B[k] = L | (cls_g ? XB_G : 0u) | (general ? XB_GENERAL : 0u) | (fast ? XB_USABLE : 0u) | (nxt << 16);
                Mlo[k] = fast ? bit_lo : 0u;
                Mhi[k] = fast ? bit_hi : 0u;
                J[k] = fast && nxt < 64u ? nxt : (uint32_t)lane;
            }
            K4_PHASE("x-double");
            const unsigned long long tp1 = now();
            /* M(l) = the usable token lanes on l's chain inside its window, J(l) = the lane that chain ends on (a usable token whose
             * successor lies beyond the window, or one that is not usable): M |= M[J], J = J[J], five times -- a sequence is at
             * least 3 stream bytes, so a window holds at most 22 of them */
#pragma unroll
            for (int r = 0; r < 5; r++) {
#pragma unroll
                for (int k = 0; k < W; k++) {
                    const int a = (int)(J[k] << 2);
                    Mlo[k] |= (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)Mlo[k]);
                    Mhi[k] |= (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)Mhi[k]);
                    J[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)J[k]);
                }
            }
            K4_PHASE("x-entries");
            const unsigned long long tp2 = now();
            /* one hop per window: which lanes are real, where the chain goes on */
            uint32_t Tlo[W], Thi[W];
            uint32_t pos = 0;                               /* relative to ip: the next token */
            bool stop = false;                              /* ... which the scalar parser has to take */
#pragma unroll
            for (int k = 0; k < W; k++) {
                Tlo[k] = 0u; Thi[k] = 0u;
                if (!stop && pos < 64u * (uint32_t)(k + 1)) {
                    const int e = (int)(pos - 64u * (uint32_t)k);
                    Tlo[k] = readlane_u32(Mlo[k], e);
                    Thi[k] = readlane_u32(Mhi[k], e);
                    const uint32_t t = readlane_u32(J[k], e);
                    const uint32_t bt = readlane_u32(B[k], (int)t);
                    if (bt & XB_USABLE) pos = 64u * (uint32_t)k + (bt >> 16);
                    else { stop = true; pos = 64u * (uint32_t)k + t; }
                }
            }
            K4_PHASE("x-rules");
            const unsigned long long tp3 = now();
            uint32_t nmax = 0;
#pragma unroll
            for (int k = 0; k < W; k++) nmax += (uint32_t)__popc(Tlo[k]) + (uint32_t)__popc(Thi[k]);
            if (nmax != 0u) {
                if (!room(nmax)) return PIPE_TIMEOUT;
                uint32_t opk = opu, nrec = 0;
                bool cut = false;
#pragma unroll
                for (int k = 0; k < W; k++) {
                    unsigned long long T = ((unsigned long long)Thi[k] << 32) | Tlo[k];
                    if (cut || T == 0ull) continue;
                    const uint32_t pl = ipu + 64u * (uint32_t)k + (uint32_t)lane;
                    const uint32_t L = B[k] & 511u, offset = A[k] & 0xffffu, mlen = A[k] >> 16;
                    const bool cls_g = (B[k] & XB_G) != 0u, general = (B[k] & XB_GENERAL) != 0u;
                    const uint32_t outlen = L + mlen;
                    bool in_t = ((T >> lane) & 1ull) != 0;
                    const uint32_t incl = wave_inclusive_scan(in_t ? outlen : 0u);
                    const uint32_t v_o = opk + (incl - (in_t ? outlen : 0u));
                    const uint32_t total = readlane_u32(incl, 63);
                    /* the position-dependent rules on the chosen sequences, as in k4lz4_decode.hpp: the shortcut needs op <= shortoend
                     * (:191), a 15+ literal run cpy <= oend - MFLIMIT (:247); the offset must stay inside the output (:338); sequences
                     * that left the shortcut also obey the end-of-block rule (:427-433).  The first sequence that fails, and
                     * everything after it, is left to the scalar parser. */
                    const uint32_t mdst_l = v_o + L;
                    unsigned long long bad;
                    if (opk + total + 64u <= oendu) {
                        bad = ballot(in_t && offset > mdst_l);
                    } else {
                        bad = ballot(in_t && ((cls_g ? mdst_l + (uint32_t)MFLIMIT > oendu : v_o + 14u + 18u > oendu) || offset > mdst_l ||
                                                (general && mdst_l + mlen + (uint32_t)MATCH_SAFEGUARD > oendu)));
                    }
                    uint32_t adv = total;
                    if (bad) {
                        const int b = ctz64(bad);
                        T &= (1ull << b) - 1ull;
                        adv = readlane_u32(incl, b) - readlane_u32(outlen, b);
                        in_t = ((T >> lane) & 1ull) != 0;
                        pos = 64u * (uint32_t)k + (uint32_t)b;
                        stop = true;
                        cut = true;
                    }
                    if (T) {
                        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(T >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)T, 0u));
                        if (in_t) {
                            const uint32_t slot = (head + nrec + below) & XQ_MASK;
                            queue[slot] = make_uint4(v_o, pl + (cls_g ? 2u : 1u), L, offset);
                            qout[slot] = v_o;
                        }
                        nrec += (uint32_t)__popcll(T);
                    }
                    opk += adv;
                }
                head += nrec;
                ip += pos;
                op = (int64_t)opk;
                if (nrec) publish();
            }
            scalar = stop || nmax == 0u;
            if (PROF) { const unsigned long long tp4 = now(); c_hyp += tp1 - tp0; c_dbl += tp2 - tp1; c_ent += tp3 - tp2; c_rules += tp4 - tp3; }
        }
        if (!scalar) continue;

        /* ---- scalar parser: one sequence, the reference's order of checks (k4lz4_decode.hpp, without the partial / dictionary arms) ---- */
        K4_PHASE("x-scalar");
        const unsigned long long ts0 = now();
        if (PROF) n_scalar++;
        int err = 0;
        uint32_t w = win.fetch((uint32_t)ip, lane);
        const uint32_t token = w & 0xffu;
        ip++;
        uint32_t length = token >> ML_BITS;
        uint32_t offset = 0;
        int64_t match = 0;
        uint32_t s_lpos = 0, s_llen = 0, s_out = 0, s_moff = 0, adv = 0;
        bool last = false, need_match = true;
        do {
            if (length != RUN_MASK && ip < shortiend && op <= shortoend) {   /* :191-225 */
                s_lpos = (uint32_t)ip; s_llen = length; s_out = (uint32_t)op;
                op += length;
                ip += length;
                const uint32_t ow = length <= 1 ? (w >> (8u * (1u + length))) : win.fetch((uint32_t)ip, lane);
                offset = ow & 0xffffu;
                ip += 2;
                match = op - (int64_t)offset;
                length = token & ML_MASK;
                if (length != ML_MASK && offset >= 8u && match >= 0) {   /* :213 */
                    s_moff = offset; adv = length + MINMATCH;
                    need_match = false;
                }
            } else {
                if (length == RUN_MASK) {                      /* :228-243, LL.tools.cs:165-193 */
                    const int64_t lencheck = iend - RUN_MASK;
                    if (ip >= lencheck) { err = (int)(-ip) - 1; break; }   /* initial_error */
                    (void)read_vle(length, lencheck);          /* loop_error: not fatal here */
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
                    if (read_vle(length, lencheck)) { err = (int)(-ip) - 1; break; }
                }
                length += MINMATCH;
                if (match < 0) { err = (int)(-ip) - 1; break; }   /* :338 */
                const int64_t cpy = op + (int64_t)length;
                if (cpy > oend - MATCH_SAFEGUARD && cpy > oend - LASTLITERALS) { err = (int)(-ip) - 1; break; }  /* :427-433 */
                s_moff = offset;                               /* 0 (hostile): those output bytes stay what they are (:408-418) */
                adv = length;
            }
        } while (false);
        if (err) return finish(err);
        if (s_llen + adv != 0u) {
            if (!room(1u)) return PIPE_TIMEOUT;
            if (lane == 0) {
                queue[head & XQ_MASK] = make_uint4(s_out, s_lpos, s_llen, s_moff);
                qout[head & XQ_MASK] = s_out;
            }
            head++;
        }
        op += adv;
        if (last) return finish((int)op);
        publish();
        if (PROF) c_scalar += now() - ts0;
    }
}

/*
 * COPY (the pair's second wave): the block's bytes, 64 at a time.  Returns the block's result (the parser's).
 * PD = how many windows the loads from global memory are issued ahead of their use: a window's look-up (A) and its completion (B)
 * are PD steps apart, the windows in between wait in PD + 1 register slots that are addressed statically (the loop is unrolled
 * PD + 1 times -- moving a slot's registers would wait for its load).
 */
template <int PD, bool PROF = false>
__device__ __forceinline__ int xcopy_block(const uint8_t *in, int src_size, uint8_t *out, int out_size, int lane, uint32_t *lds, unsigned long long *pc = nullptr)
{
    uint32_t *pipe = lds + X_PIPE;
    const uint4 *queue = (const uint4 *)(lds + X_QUEUE);
    const uint32_t *qout = lds + X_QOUT;
    uint8_t *hist = (uint8_t *)(lds + X_HIST);
    uint8_t *flags = (uint8_t *)(lds + X_FLAGS);
    {
        int r;
        if (xdecode_trivial(in, src_size, out_size, lane, &r)) return r;
    }
    constexpr uint32_t RESOLVED = 0xffffffffu;
    constexpr int NS = PD + 1;
    uint32_t pv[NS], ps[NS], plim[NS];                      /* windows on their way: the byte, where it comes from if not loaded, how many bytes */
#pragma unroll
    for (int i = 0; i < NS; i++) { pv[i] = 0u; ps[i] = RESOLVED; plim[i] = 0u; }
    uint32_t kcov = 0xffffffffu;                           /* index of the last record that begins below baseA */
    uint32_t baseA = 0, baseB = 0, F = 0;                  /* next window to look up / to finish; bytes below F are in memory */
    uint32_t head = 0, avail = 0;                          /* records published as last seen, and the output position they reach */
    bool done = false, more = true, go = true;
    int result = 0;
    uint32_t inflight = 0, spin = 0;
    unsigned long long c_wait = 0, n_win = 0, n_poll = 0, n_inw = 0, n_idle = 0;
    const unsigned long long t_begin = PROF ? (unsigned long long)__builtin_readcyclecounter() : 0ull;

    while (go) {
#pragma unroll
        for (int j = 0; j < NS; j++) {
            if (!(more || inflight)) { go = false; break; }
            const int sa = j, sb = (j + 1) % NS;
            /* ---------------- A: records and sources of window [baseA, baseA + 64) ---------------- */
            uint32_t lim = 0;
            if (more) {
                const unsigned long long tw = PROF ? (unsigned long long)__builtin_readcyclecounter() : 0ull;
                while (!done && avail < baseA + 64u) {
                    const uint32_t d = pipe_load(pipe + 2);
                    head = pipe_load(pipe + 0);
                    avail = uni(qout[head & XQ_MASK]);
                    if (PROF) n_poll++;
                    if (d) {
                        done = true;
                        result = (int)uni(pipe[3]);
                    } else if (avail < baseA + 64u) {
                        if (inflight) break;                /* finish an older window meanwhile */
                        if (++spin >= PIPE_SPIN_MAX) {
                            dev_status_raise((uint32_t *)(uintptr_t)((unsigned long long)pipe[6] | ((unsigned long long)pipe[7] << 32)), (uint32_t)DEV_STATUS_PIPE_TIMEOUT);
                            return PIPE_TIMEOUT;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if (PROF) c_wait += (unsigned long long)__builtin_readcyclecounter() - tw;
                if (done && result < 0) return result;      /* (what a failed block leaves in its slot is not defined) */
                if (avail >= baseA + 64u) lim = 64u;
                else if (done) { lim = avail - baseA; more = false; }
            }
            if (lim != 0u) {
                K4_PHASE("x-copy-a");
                spin = 0;
                const uint32_t p = baseA + (uint32_t)lane;
                if (lane < 16) ((uint32_t *)flags)[lane] = 0u;
                lds_order();
                const uint32_t ci = kcov + 1u + (uint32_t)lane;
                const bool cv = (int32_t)(head - ci) > 0;
                const uint32_t co = cv ? qout[ci & XQ_MASK] - baseA : 64u;
                if (co < 64u) flags[co] = 1;
                lds_order();
                const unsigned long long S = ballot(flags[lane] != 0);
                const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(S >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)S, 0u)) + (uint32_t)((S >> lane) & 1ull);
                const uint4 rec = queue[(kcov + r) & XQ_MASK];
                kcov += (uint32_t)__popcll(S);
                lds_order();
                /* the records below kcov are not looked at again */
                if (lane == 0) xpipe_post(pipe + 1, kcov == 0xffffffffu ? 0u : kcov);
                const uint32_t rel = p - rec.x;
                const bool valid = (uint32_t)lane < lim;
                const bool lit = rel < rec.z;
                const uint32_t src = p - rec.w;
                const bool far = !lit && (src < F || rec.w == 0u);             /* offset 0: the byte that is there already */
                /* one load for the whole window: literals from the stream, old match sources from the output, the other lanes
                 * a byte that is not used (the first of the output) */
                const uint8_t *g = (valid && lit) ? in + (rec.y + rel) : out + ((valid && far) ? src : 0u);
                pv[sa] = (uint32_t)*g;
                ps[sa] = (valid && !lit && !far) ? src : RESOLVED;
                plim[sa] = lim;
                baseA += 64u;
                inflight++;
                if (PROF) n_win++;
            } else if (PROF) n_idle++;
            /* ---------------- B: the oldest window on its way ---------------- */
            if (plim[sb] != 0u) {
                K4_PHASE("x-copy-b");
                uint32_t val = pv[sb];
                const uint32_t src = ps[sb];
                const bool valid = (uint32_t)lane < plim[sb];
                const bool inw = src != RESOLVED && src >= baseB;
                if (src != RESOLVED && !inw) val = hist[src & (uint32_t)(XHIST - 1)];
                if (ballot(inw)) {
                    /* sources inside the window: follow them to a lane that has its byte */
                    uint32_t ptr = inw ? src - baseB : (uint32_t)lane;
                    for (int it = 0; it < 6; it++) {
                        const uint32_t nx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ptr << 2), (int)ptr);
                        const bool same = nx == ptr;
                        ptr = nx;
                        if (!ballot(!same)) break;
                    }
                    val = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ptr << 2), (int)val);
                    if (PROF) n_inw++;
                }
                if (valid) hist[(baseB + (uint32_t)lane) & (uint32_t)(XHIST - 1)] = (uint8_t)val;
                baseB += plim[sb];
                plim[sb] = 0u;
                inflight--;
                lds_order();
                while (baseB - F >= (uint32_t)XFLUSH) {
                    const uint4 v = *(const uint4 *)(hist + ((F + 16u * (uint32_t)lane) & (uint32_t)(XHIST - 1)));
                    U128u o;
                    o.v[0] = v.x; o.v[1] = v.y; o.v[2] = v.z; o.v[3] = v.w;
                    st128u(out + F + 16u * (uint32_t)lane, o);
                    F += (uint32_t)XFLUSH;
                }
            }
        }
    }
    /* what is left in the ring: 16 bytes per lane, then the last bytes one by one */
    {
        const uint32_t n = baseB - F;
        for (uint32_t k = 16u * (uint32_t)lane; k + 16u <= n; k += 1024u) {
            const uint4 v = *(const uint4 *)(hist + ((F + k) & (uint32_t)(XHIST - 1)));
            U128u o;
            o.v[0] = v.x; o.v[1] = v.y; o.v[2] = v.z; o.v[3] = v.w;
            st128u(out + F + k, o);
        }
        const uint32_t t0 = n & ~15u;
        if ((uint32_t)lane < (n & 15u)) out[F + t0 + (uint32_t)lane] = hist[(F + t0 + (uint32_t)lane) & (uint32_t)(XHIST - 1)];
    }
    if (PROF && pc && lane == 0) {
        pc[0] = (unsigned long long)__builtin_readcyclecounter() - t_begin; pc[1] = c_wait; pc[2] = n_win; pc[3] = n_poll; pc[4] = n_inw; pc[5] = n_idle;
    }
    return result;
}

constexpr int XPAIRS_PER_WG = 2;

/* may a batch go through this decoder?  (plain LZ4_decompress_safe: no dictionary, not partial) */
__device__ __forceinline__ void xdecode_kernel_body(const BatchArgs &a, uint32_t (*lds)[X_PAIR_DWORDS])
{
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    /* odd workgroups swap the roles, so that a SIMD hosts parsing and copying waves alike */
    const uint32_t pair = wave >> 1, role = (wave ^ blockIdx.x) & 1u;
    const long long slot = (long long)blockIdx.x * XPAIRS_PER_WG + (long long)pair;
    uint32_t *mine = lds[pair];
    if (role == 0) {
        pipe_init(mine + X_PIPE, a.status, lane);
        if (lane == 0) mine[X_QOUT] = 0u;                   /* the end mark of an empty queue */
    }
    __syncthreads();
    if (slot >= a.n) return;
    const long long b = a.order ? (long long)uni(a.order[slot]) : slot;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    const uint8_t *in = a.src + a.srcOff[b];
    uint8_t *out = a.dst + a.dstOff[b];
    const bool run = src_len > 0 || (a.flags & FLAG_RAW_RETURN);
    if (role == 0) {
        if (a.prof) prof_place<true>(a.prof + PROF_STRIDE * b, 8, lane);
        if (run) xparse_block<K4_X_WINDOWS>(in, src_len, cap < 0 ? 0 : cap, lane, mine);
    } else {
        int ret = 0;
        if (run) ret = xcopy_block<K4_X_DEPTH>(in, src_len, out, cap < 0 ? 0 : cap, lane, mine);
        if (lane == 0) a.outLen[b] = codec_decode_result(src_len, ret, a.flags);
        if (a.prof) prof_place<true>(a.prof + PROF_STRIDE * b, 9, lane);
    }
}

__global__ __launch_bounds__(128 * XPAIRS_PER_WG) __attribute__((amdgpu_waves_per_eu(8, 8))) void k4_decode_x_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[XPAIRS_PER_WG][X_PAIR_DWORDS];
    xdecode_kernel_body(a, lds);
}

/* diagnostic twin: 32 cycle / event counters per block, the parsing wave's in [0, 16), the copying wave's in [16, 32) (scripts/x_probe.py) */
__global__ __launch_bounds__(128 * XPAIRS_PER_WG) __attribute__((amdgpu_waves_per_eu(8, 8))) void k4_decode_x_prof_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[XPAIRS_PER_WG][X_PAIR_DWORDS];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t pair = wave >> 1, role = (wave ^ blockIdx.x) & 1u;
    const long long b = (long long)blockIdx.x * XPAIRS_PER_WG + (long long)pair;
    uint32_t *mine = lds[pair];
    if (role == 0) {
        pipe_init(mine + X_PIPE, a.status, lane);
        if (lane == 0) mine[X_QOUT] = 0u;
    }
    __syncthreads();
    if (b >= a.n) return;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    const uint8_t *in = a.src + a.srcOff[b];
    uint8_t *out = a.dst + a.dstOff[b];
    if (src_len <= 0) return;
    if (role == 0) {
        xparse_block<K4_X_WINDOWS, true>(in, src_len, cap < 0 ? 0 : cap, lane, mine, a.prof + 2 * PROF_STRIDE * b);
    } else {
        const int ret = xcopy_block<K4_X_DEPTH, true>(in, src_len, out, cap < 0 ? 0 : cap, lane, mine, a.prof + 2 * PROF_STRIDE * b + PROF_STRIDE);
        if (lane == 0) a.outLen[b] = codec_decode_result(src_len, ret, a.flags);
    }
}

}  // namespace k4
