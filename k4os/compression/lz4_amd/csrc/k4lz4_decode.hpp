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
 * on malformed input); how the work is organised is entirely different.  A block is decoded in
 * batches of up to 64 sequences, each batch in three phases:
 *
 *   PARSE     The compressed stream sits in a 1 KiB per-wave LDS ring that is refilled with
 *             coalesced dword loads one chunk ahead of use.  Token parsing is speculative and
 *             lane-parallel: in one round every lane i assumes that a token starts at stream byte
 *             ip + i and decodes that hypothetical sequence (literal length, offset, match length,
 *             where the next token would start).  Lane 0's hypothesis is true; following the
 *             `next` links from lane 0 with v_readlane (a handful of scalar instructions per hop)
 *             picks out the real sequences among the 64 hypotheses and assigns their output
 *             positions.  Only sequences that need more than that (15+ literals, multi-byte match
 *             length, block end, any malformed input) go through the scalar parser, which follows
 *             the reference line by line.  Each real sequence's (literal position, literal length,
 *             output position, offset, match length) is compacted into lane k of the batch.
 *   LITERALS  every lane moves the literal run of its own sequence (all loads before the first
 *             store: one memory round trip for the whole batch); long runs are moved by the whole
 *             wave, 16 B per lane.
 *   MATCHES   a match may only be copied once every earlier match that writes into its source
 *             range is finished.  Destination ranges are sorted by lane, so each lane finds the
 *             lane interval it depends on with one binary search per batch; then, round by round,
 *             all matches without unfinished dependencies copy in parallel.  Overlapping matches
 *             (offset < length) and long ones are moved by the whole wave; overlapping copies read
 *             the already-final first period, so no lane depends on a byte of the same instruction.
 *
 * The match source is the block's own earlier output in HBM/L2; this wave's stores and later loads
 * to it are ordered by program order (wave_sync() pins the compiler).
 */
#pragma once
#include "k4lz4_common.hpp"

namespace k4 {

#ifndef K4_DEC_STAGE
#define K4_DEC_STAGE 2048
#endif
#ifndef K4_DEC_SOFT
#define K4_DEC_SOFT (K4_DEC_STAGE - 512)
#endif
#ifndef K4_DEC_WAVE_AT
#define K4_DEC_WAVE_AT 128
#endif
/* late blocks first for the decoder pairs (k4lz4_common.hpp, Pace): a report every 2^STEP compressed bytes, epochs of 2^EPOCH
 * ticks of 10 ns, priority tiers 1/DEN apart.  Measured on the bench batch: 245 GiB/s without, 265 / 281 / 290 / 287 with steps of
 * 4 / 2 / 1 / 0.5 KiB (epochs to match), 281 / 290 / 287 / 277 with tiers of 1/16 / 1/32 / 1/64 / 1/128; with the parsing wave
 * alone taking the priority (the first attempt) 241; with the copying wave one level below its block's 263, with the parsing wave
 * one level above it 278: the two belong at the same level. */
#ifndef K4_DEC_PACE
#define K4_DEC_PACE 1
#endif
#ifndef K4_DEC_PACE_STEP
#define K4_DEC_PACE_STEP 10
#endif
#ifndef K4_DEC_PACE_EPOCH
#define K4_DEC_PACE_EPOCH 10
#endif
#ifndef K4_DEC_PACE_DEN
#define K4_DEC_PACE_DEN 32u
#endif
constexpr int DECODE_STAGE_BYTES = K4_DEC_STAGE;          /* a batch's output, kept in LDS while its matches resolve */
constexpr int DECODE_LDS_DWORDS = RING_DWORDS + 5 * 64 + (DECODE_STAGE_BYTES + 64) / 4;   /* ring + 5 descriptor arrays + stage */
constexpr int MAX_SEQ_PER_ROUND = 22;            /* 64 hypotheses, >= 3 stream bytes per sequence */
constexpr int DECODE_BATCH_SOFT_BYTES = K4_DEC_SOFT;   /* PARSE stops adding to a batch beyond this many output bytes */

/* PARSE's serial part: from hypothesis 0 follow the `next` links while the hypotheses are usable and stay inside
 * the 64-lane window; T collects the real sequences, idx ends on the first position not taken.  A lane's word:
 * bits 0-5 next lane if the chain goes on from here, else the lane itself; bit 7 the chain ends here (hypothesis
 * unusable, or next token outside the window), bit 8 usable, bits 9.. next.  Scalar ISA by hand, and without a branch
 * in the loop-carried path: s_bitset1 and v_readlane take the lane from the low six bits of the word just read, so
 * one v_readlane feeds the next directly, and a lane where the chain ends points at itself, so hopping on is
 * harmless -- 24 hops (a window holds at most 22 sequences) are laid out straight, with an exit test after 8 and 16.
 * The compiler's loop has 14 instructions and two branches per sequence.  The lane the chain stops on is marked before it
 * is known to be usable and unmarked afterwards if it was not. */
__device__ __forceinline__ uint32_t token_word(bool fast, uint32_t next, int lane)
{
    return (fast && next < 64u ? next : (0x80u | (uint32_t)lane)) | (fast ? 0x100u : 0u) | (next << 9);
}
__device__ __forceinline__ void follow_tokens_ref(uint32_t word, unsigned long long &T, uint32_t &idx)
{   /* what the ISA below does, in C: the emulator build runs this, k4_chain_selftest_kernel compares the two on the GPU */
    T = 0;
    idx = 0;
    while (idx < 64u) {
        const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane(word, (int)idx);
        if (!(pk & 0x100u)) break;
        T |= 1ull << idx;
        idx = pk >> 9;
    }
}
__device__ __forceinline__ void follow_tokens1(uint32_t word, unsigned long long &T, uint32_t &idx);
__device__ __forceinline__ void follow_tokens(uint32_t word, unsigned long long &T, uint32_t &idx, bool two = true)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (!two) { follow_tokens1(word, T, idx); return; }
    /* Two sequences per hop: every lane first learns where its successor points (one ds_bpermute), and the chain follows those
     * double links -- the word a hop reads carries the lane after next in its low six bits (the next v_readlane's lane select, as
     * before) and the lane in between in bits 24-29, which a shift hands to a second s_bitset1.  Lanes where the chain ends point
     * at themselves, so both links of such a lane are the lane itself and hopping on stays harmless; the word read last is the
     * end lane's own either way.  The three scalar instructions between two v_readlane are three of the four wait states the
     * lane select needs.  scripts/ubench/hop_chain.hip, 24 sequences at 1 / 4 / 8 waves per SIMD: 1016 / 1045 / 1375 cycles
     * one link at a time, 592 / 711 / 1054 this way, the bpermute included.  In the pair decoder (A/B on one box, bench batch cut
     * to 512 / 1024 / 2048 / 3072 / 4096 blocks): +10 / +9.7 / +7 / +1.6 / -3 % -- with every wave slot of the chip taken, the LDS
     * round trip costs more than the shorter chain saves -- so the host side launches the kernels built with it (k4_decode_pair2_kernel, k4_unpickle_pair2_kernel) up to 12 blocks per CU,
     * and follow_tokens1 below is the form for a full chip. */
    const uint32_t n1 = word & 63u;
    const uint32_t w1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(n1 << 2), (int)word);
    const uint32_t word2 = (word & ~63u) | (w1 & 63u) | (n1 << 24);
    uint32_t pk = 0, mid = 0;
    T = 0;
#define K4_HOP "s_bitset1_b64 %[T], %[pk]\n\ts_lshr_b32 %[mid], %[pk], 24\n\ts_bitset1_b64 %[T], %[mid]\n\ts_nop 0\n\tv_readlane_b32 %[pk], %[word], %[pk]\n\t"
#define K4_HOP4 K4_HOP K4_HOP K4_HOP K4_HOP
    asm volatile(
        K4_HOP4
        "s_bitcmp1_b32 %[pk], 7\n\t"
        "s_cbranch_scc1 .Ltok_end%=\n\t"
        K4_HOP4
        "s_bitcmp1_b32 %[pk], 7\n\t"
        "s_cbranch_scc1 .Ltok_end%=\n\t"
        K4_HOP4
        ".Ltok_end%=:"
        : [T] "+s"(T), [pk] "+s"(pk), [mid] "+s"(mid)
        : [word] "v"(word2)
        : "scc");
#undef K4_HOP4
#undef K4_HOP
    const uint32_t last = 63u - (uint32_t)__builtin_clzll(T);
    if (pk & 0x100u) {
        idx = (pk >> 9) & 0x7fffu;                          /* (bits 24-29: the link in between) */
    } else {
        T &= ~(1ull << last);
        idx = last;
    }
#else
    (void)two;
    follow_tokens_ref(word, T, idx);
#endif
}
/* one link per hop: s_bitset1 and v_readlane both take the lane from the low six bits of the word just read */
__device__ __forceinline__ void follow_tokens1(uint32_t word, unsigned long long &T, uint32_t &idx)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t pk = 0;
    T = 0;
#define K4_HOP "s_bitset1_b64 %[T], %[pk]\n\ts_nop 2\n\tv_readlane_b32 %[pk], %[word], %[pk]\n\t"
#define K4_HOP8 K4_HOP K4_HOP K4_HOP K4_HOP K4_HOP K4_HOP K4_HOP K4_HOP
    asm volatile(
        K4_HOP8
        "s_bitcmp1_b32 %[pk], 7\n\t"
        "s_cbranch_scc1 .Ltok_end%=\n\t"
        K4_HOP8
        "s_bitcmp1_b32 %[pk], 7\n\t"
        "s_cbranch_scc1 .Ltok_end%=\n\t"
        K4_HOP8
        ".Ltok_end%=:"
        : [T] "+s"(T), [pk] "+s"(pk)
        : [word] "v"(word)
        : "scc");
#undef K4_HOP8
#undef K4_HOP
    const uint32_t last = 63u - (uint32_t)__builtin_clzll(T);
    if (pk & 0x100u) {
        idx = pk >> 9;
    } else {
        T &= ~(1ull << last);
        idx = last;
    }
#else
    follow_tokens_ref(word, T, idx);
#endif
}

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

/* Dictionary of a block (LZ4Codec.Decode(..., dictionary), LL64.LZ4_decompress_safe_usingDict,
 * LL64.dec.cs:523-546): mode 0 none, 1 prefix (the dictionary sits immediately before the output:
 * `size` bytes, 65536 when it is 64 KiB-1 or more -- withPrefix64k), 2 external.
 * `end` = one past the last dictionary byte (mode 1: == out). */
struct DecodeDict { const uint8_t *end; uint32_t size; int mode; };

/* match that starts before the block: `from_dict` bytes come out of the dictionary, the rest from
 * the start of the output with the usual replicating semantics (LL64.dec.cs:342-378) */
__device__ __forceinline__ void wave_dict_copy(uint8_t *out, const uint8_t *dict_end, uint32_t op, uint32_t from_dict, uint32_t len, int lane)
{
    const uint32_t n1 = len < from_dict ? len : from_dict;
    wave_sync();
    wave_copy(out + op, dict_end - from_dict, n1, lane);
    if (len > n1) wave_match_copy(out, op + n1, op + n1, len - n1, lane);
}

/* inclusive running maximum over the 64 lanes (values >= 0), DPP like wave_inclusive_scan */
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t x)
{
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));   /* row_shr:1 */
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));   /* row_shr:2 */
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));   /* row_shr:4 */
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));   /* row_shr:8 */
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));  /* row_bcast:15 -> rows 1,3 */
    x = mx(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));  /* row_bcast:31 -> rows 2,3 */
    return x;
}

/* Items longer than a lane moves at once (literal runs, match sources from before the batch), all of them together:
 * lane l's item is `x` bytes (0 = none, else > 32) at src + soff, to go to stage + doff.  The items are cut into 16-byte
 * pieces, the pieces are numbered through (prefix sum) and dealt out 64 at a time, so one load instruction of the wave
 * moves a KiB of them whatever their number.  Which item a piece belongs to: every item marks the slot of its first piece
 * (`own`, 64 dwords of LDS), a running maximum fills the slots in between.  An item's last piece is its last 16 bytes
 * (it overlaps the piece before it), so nothing outside [src + soff, + x) is read and nothing outside the item written. */
__device__ __forceinline__ void long_pieces(uint8_t *stg, uint32_t *own, const uint8_t *src, uint32_t soff, uint32_t x, uint32_t doff, int lane)
{
    const uint32_t np = (x + 15u) >> 4;
    const uint32_t incl = wave_inclusive_scan(np);
    const uint32_t total = readlane_u32(incl, 63), base = incl - np;
    uint32_t carry = 0;                                     /* the item (lane + 1) that owns the last piece of the pass before */
    for (uint32_t p0 = 0; p0 < total; p0 += 64u) {
        lds_sync();
        own[lane] = 0u;
        lds_sync();
        if (np != 0u && base - p0 < 64u) own[base - p0] = (uint32_t)lane + 1u;
        lds_sync();
        uint32_t it = wave_inclusive_max(own[lane]);
        if (it == 0u) it = carry;
        carry = readlane_u32(it, 63);
        const bool act = p0 + (uint32_t)lane < total;
        const int sl = (int)((it - 1u) & 63u);
        const uint32_t i_soff = (uint32_t)__shfl((int)soff, sl), i_x = (uint32_t)__shfl((int)x, sl);
        const uint32_t i_doff = (uint32_t)__shfl((int)doff, sl), i_base = (uint32_t)__shfl((int)base, sl);
        if (act) {
            uint32_t o = 16u * (p0 + (uint32_t)lane - i_base);
            if (o + 16u > i_x) o = i_x - 16u;
            const U128u v = ld128u(src + i_soff + o);
            uint8_t *d = stg + i_doff + o;
            ((U64u *)d)->v = ((uint64_t)v.v[1] << 32) | v.v[0];
            ((U64u *)(d + 8))->v = ((uint64_t)v.v[3] << 32) | v.v[2];
        }
    }
    lds_sync();
}

/* stage[dst ..) = stage[dst - offset ..) for n bytes with the byte-serial semantics of LL64.dec.cs:408-450, by the whole
 * wave: eight bytes per lane and pass, a pass never reaching into what it writes (periods of 8 bytes and more); one
 * byte per lane for shorter periods, every lane reading the first period, which is final. */
__device__ __forceinline__ void stage_wave_copy(uint8_t *stg, uint32_t dst, uint32_t offset, uint32_t n, int lane)
{
    uint8_t *d = stg + dst;
    const uint8_t *m = d - offset;
    lds_sync();
    if (offset >= 8u) {
        const uint32_t per = offset < 512u ? offset : 512u;          /* bytes per pass */
        for (uint32_t k0 = 0; k0 < n; k0 += per) {
            const uint32_t len = n - k0 < per ? n - k0 : per;        /* this pass: [k0, k0 + len), len >= 1 */
            uint32_t k = 8u * (uint32_t)lane;
            const bool act = k < len;
            if (len >= 8u) {
                if (k + 8u > len) k = len - 8u;                      /* the last word of the pass: its last 8 bytes */
                if (act) {
                    const uint64_t v = ld64u(m + k0 + k);
                    ((U64u *)(d + k0 + k))->v = v;
                }
            } else if ((uint32_t)lane < len) {
                d[k0 + (uint32_t)lane] = m[k0 + (uint32_t)lane];
            }
            lds_sync();
        }
    } else {
        const uint32_t chunk = (64u / offset) * offset;
        const uint32_t r = (uint32_t)lane % offset;
        for (uint32_t k0 = 0; k0 < n; k0 += chunk) {
            const uint32_t k = k0 + (uint32_t)lane;
            if ((uint32_t)lane < chunk && k < n) d[k] = m[r];
        }
        lds_sync();
    }
}

/*
 * Two wavefronts per block (k4_decode_pair_kernel): a decoder wave spends ~85 % of its cycles waiting on its own
 * dependent chains, so PARSE (wave A) and LITERALS/MATCHES (wave B) of one block run side by side, batch k+1 being
 * parsed while batch k is copied.  The waves share a PIPE_SLOTS-deep queue of batch descriptors in LDS:
 *   pipe[0] head  batches published by A      pipe[1] tail  batches taken by B
 *   pipe[8 + 8 * slot ..]  nseq, output position of the batch, its byte count, last-batch flag, the block's result
 *   pipe[PIPE_DESC + 320 * slot ..]  the five descriptor arrays
 *   pipe[PIPE_SCRATCH ..]  B's two sort arrays, pipe[PIPE_STAGE ..] B's output stage
 * LDS executes in order, so "write the batch, wait for the writes, then advance head" is a release; the reader polls
 * with s_sleep between attempts and gives up (block fails) after PIPE_SPIN_MAX polls instead of hanging.
 */
constexpr int PIPE_SLOTS = 4;                     /* batches in flight between the two waves (power of two) */
constexpr int PIPE_DESC = 8 + 8 * PIPE_SLOTS, PIPE_SCRATCH = PIPE_DESC + PIPE_SLOTS * 320, PIPE_STAGE = PIPE_SCRATCH + 128;
constexpr int PIPE_DWORDS = PIPE_STAGE + (DECODE_STAGE_BYTES + 64) / 4;
constexpr int DECODE_PAIR_LDS_DWORDS = RING_DWORDS + PIPE_DWORDS;
constexpr uint32_t PIPE_SPIN_MAX = 1u << 24;
constexpr int PIPE_TIMEOUT = -0x7ffffff0;

__device__ __forceinline__ uint32_t pipe_load(const uint32_t *p)
{
#ifndef K4_HOST_EMU
    return uni(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
#else
    return uni(*(const volatile uint32_t *)p);             /* one lane's view for the whole wave */
#endif
}
__device__ __forceinline__ void pipe_store(uint32_t *p, uint32_t v, int lane)
{
    lds_sync();
#ifndef K4_HOST_EMU
    if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    if (lane == 0) *(volatile uint32_t *)p = v;
#endif
    lds_sync();
}
/* the pair's queue keeps the address of its context's status word in pipe[6..7] (written by pipe_init) */
__device__ __forceinline__ void pipe_init(uint32_t *pipe, uint32_t *status, int lane)
{
    if (lane < 6) pipe[lane] = 0u;                          /* head, tail */
    if (lane == 6) pipe[6] = (uint32_t)(uintptr_t)status;
    if (lane == 7) pipe[7] = (uint32_t)((unsigned long long)(uintptr_t)status >> 32);
}
/* poll until pipe[which] >= want (counters only grow); false after PIPE_SPIN_MAX polls */
__device__ __forceinline__ bool pipe_wait(const uint32_t *pipe, int which, uint32_t want)
{
    for (uint32_t spin = 0; spin < PIPE_SPIN_MAX; spin++) {
        if (pipe_load(pipe + which) >= want) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    /* reported at call level, to the context that launched this kernel: not this block's fault */
    dev_status_raise((uint32_t *)(uintptr_t)((unsigned long long)pipe[6] | ((unsigned long long)pipe[7] << 32)), (uint32_t)DEV_STATUS_PIPE_TIMEOUT);
    return false;
}

/*
 * Decode one block.  Returns what LL64.LZ4_decompress_safe returns: the number of bytes written,
 * or -(input position) - 1 when the stream is malformed (LL64.dec.cs:465).
 * `lds`: DECODE_LDS_DWORDS dwords of LDS owned by this wave.
 */
/* ROLE 0: one wave does everything.  ROLE 1: PARSE only, batches go into the pair's queue (`pipe`).  ROLE 2: takes
 * batches from the queue and does LITERALS / MATCHES; returns the block's result. */
template <bool PROF = false, int ROLE = 0, bool HOP2 = false>
__device__ __forceinline__ int decode_block(const uint8_t *in, int src_size, uint8_t *out, int out_size, int lane,
                                            uint32_t *lds, unsigned long long *pc = nullptr, bool partial = false,
                                            DecodeDict dict = DecodeDict{nullptr, 0u, 0}, uint32_t *pipe = nullptr,
                                            uint32_t *seq = nullptr, uint32_t *pace = nullptr)
{
    /* lowPrefix relative to out (<= 0), the size used by the offset check (:149,:338) */
    const int64_t low_prefix = dict.mode == 1 ? -(int64_t)dict.size : 0;
    const int64_t chk_size = dict.mode == 2 ? (int64_t)dict.size : 0;
    const bool check_offset = chk_size < 65536;
    const bool prefix64 = dict.mode == 1 && dict.size == 65536u;
    unsigned long long c_parse = 0, c_lit = 0, c_match = 0, n_batch = 0, n_round = 0, n_seq = 0, n_slow = 0;
    unsigned long long c_hyp = 0, c_chain = 0, c_rules = 0, c_slots = 0, n_spec = 0;   /* PARSE split: speculative rounds */
    unsigned long long c_wait = 0, c_search = 0, c_rounds = 0, c_flush = 0;               /* pair kernel: queue waits; the copying wave's staged batches */
    if (ROLE == 0) prof_place<PROF>(pc, 8, lane);
    const unsigned long long t_begin = prof_now<PROF>();
    if (out_size == 0) {                                   /* LL64.dec.cs:162-168 */
        if (partial) return 0;
        if (src_size == 1) {
            uint32_t b = uni(lane == 0 ? (uint32_t)in[0] : 0u);
            return b == 0 ? 0 : -1;
        }
        return -1;
    }
    if (src_size <= 0) return -1;                          /* :172 */

    StreamRing win;
    if (ROLE != 2) win.init(lds, in, (uint32_t)src_size, lane);
    uint32_t *d_lpos = ROLE == 0 ? lds + RING_DWORDS : pipe + PIPE_DESC, *d_llen = d_lpos + 64, *d_out = d_llen + 64,
             *d_moff = d_out + 64, *d_mlen = d_moff + 64;
    /* MATCHES sorts destination ranges in two arrays: the descriptor arrays themselves when one wave does it all */
    uint32_t *w_out = ROLE == 0 ? d_out : pipe + PIPE_SCRATCH, *w_end = ROLE == 0 ? d_llen : pipe + PIPE_SCRATCH + 64;
    uint8_t *const stage_base = ROLE == 0 ? (uint8_t *)(d_mlen + 64) : (uint8_t *)(pipe + PIPE_STAGE);
    /* ROLE 1/2: batches published / taken so far.  A pair that decodes several blocks one after the other (linked frames)
     * keeps counting in *seq, so the parsing wave can be blocks ahead of the copying one */
    uint32_t batch_no = (ROLE != 0 && seq) ? *seq : 0u;

    const int64_t iend = src_size;
    const int64_t oend = out_size;
    const int64_t shortiend = iend - 14 - 2;               /* :152 */
    const int64_t shortoend = oend - 14 - 18;              /* :153 */
    int64_t ip = 0, op = 0;
    /* LZ4_readVLE (LL.tools.cs:165-193: read a byte, ip++, add it, stop when ip reaches `lencheck` or the byte was not 255), a
     * wave-full of bytes at a time: lane l looks at stream byte ip + l, the first byte that is not 255 ends the field.  A run of
     * 4 096 literals is 17 such bytes, a 4 MiB one 16 448: one step (or 257) instead of as many trips through the ring.  The bytes
     * consumed and the sum (mod 2^32, like the reference's uint) are the scalar loop's: a step never reads past lencheck - 1
     * except for the one byte the loop reads before it looks.  Returns true when ip reached lencheck (the caller decides what
     * that means: nothing for a literal length, an error for a match length). */
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
        /* ======================= PARSE ======================= */
        const unsigned long long t0 = prof_now<PROF>();
        int64_t op_batch = op;
        const uint32_t ip_batch = (uint32_t)ip;
        int nseq = 0;
        int err = 0;
        bool done = false;
        bool gap = false;      /* a sequence of the batch leaves output bytes as they are (offset 0 in a hostile stream: the reference
                                * copies them onto themselves, LL64.dec.cs:408-418) -- such a batch is not assembled in the stage */
        uint32_t *meta = nullptr;
        if (ROLE != 0) {
            const uint32_t slot = batch_no & (uint32_t)(PIPE_SLOTS - 1);
            meta = pipe + 8 + 8 * slot;
            d_lpos = pipe + PIPE_DESC + 320 * slot; d_llen = d_lpos + 64; d_out = d_llen + 64; d_moff = d_out + 64; d_mlen = d_moff + 64;
            if (ROLE == 1) {                                /* the slot is free once the other wave has taken batch_no - PIPE_SLOTS */
                if (batch_no >= (uint32_t)PIPE_SLOTS && !pipe_wait(pipe, 1, batch_no + 1u - (uint32_t)PIPE_SLOTS)) return PIPE_TIMEOUT;
                if (PROF) c_wait += prof_now<PROF>() - t0;
            } else {
                if (!pipe_wait(pipe, 0, batch_no + 1u)) return PIPE_TIMEOUT;
                if (PROF) c_wait += prof_now<PROF>() - t0;
                nseq = (int)uni(meta[0]);
                op_batch = (int64_t)uni(meta[1]);
                op = op_batch + (int64_t)uni(meta[2]);
                done = uni(meta[3]) != 0u;
                err = (int)uni(meta[4]);                    /* the block's result, valid with `done` */
                gap = uni(meta[5]) != 0u;
#if K4_DEC_PACE
                if (pace) { const uint32_t lv = uni(pipe[3]); if (lv) Pace::set_level((int)lv - 1); }
#endif
            }
        }
        /* a batch is closed when it may not take another round's sequences, or when its output nears what the stage holds */
        while (ROLE != 2 && nseq <= 64 - MAX_SEQ_PER_ROUND && op - op_batch <= (int64_t)DECODE_BATCH_SOFT_BYTES && !done) {
            /* ---- speculative round: 64 hypotheses "a token starts at ip + lane" ----
             * All positions of this part are below 2^31 (sizes are `int`), a hypothesis adds at most a few hundred: the conditions
             * of LL64.dec.cs are written as sums in uint32 (ip + lane + k < iend instead of lane < iend - k - ip), which costs a
             * third of the instructions their 64-bit forms did, and both extension bytes are taken without a branch (among 64
             * hypotheses some lane always has them). */
            const uint32_t ipu = (uint32_t)ip, opu = (uint32_t)op, iendu = (uint32_t)iend, oendu = (uint32_t)oend;
            if (ipu + (uint32_t)RUN_MASK + 1u < iendu) {   /* some hypothesis can be usable: lim > 0 || iend - RUN_MASK - 1 - ip > 0 */
                K4_PHASE("spec-hyp");
                const unsigned long long tp0 = prof_now<PROF>();
                win.ensure(ipu + win.a0, lane);
                const uint32_t pl = ipu + (uint32_t)lane;           /* stream position of this lane's hypothetical token */
                const uint32_t q = pl + win.a0;
                const uint32_t t4 = win.read4(q);
                const uint32_t L0 = (t4 >> 4) & 15u;
                const uint32_t M = t4 & 15u;
                /* class S: the shortcut (:191-225), literal length in the token: ip + lane + 1 < shortiend.
                 * class G: 15 + one extension byte of literals -> the general literal path (:228-315): lane < iend - RUN_MASK - 1 - ip,
                 * the extension byte not 255, and the run leaves room for offset + a last sequence (:247) */
                const bool cls_g = L0 == RUN_MASK;
                const uint32_t ext_l = (t4 >> 8) & 0xffu;
                const uint32_t L = cls_g ? L0 + ext_l : L0;
                const uint32_t hdr = cls_g ? 2u : 1u;             /* token (+ literal-length extension) bytes */
                bool fast = cls_g ? (pl + (uint32_t)RUN_MASK + 1u < iendu && ext_l != 255u && pl + hdr + L + (2u + 1u + LASTLITERALS) <= iendu)
                                  : pl + 1u + 14u + 2u < iendu;
                const uint32_t q2 = q + hdr + L;
                const uint32_t o4 = win.read4(q2);
                const uint32_t offset = o4 & 0xffffu;
                /* where the match-length field ends and the next token starts, relative to ip; one extension byte (:326-334):
                 * the byte after it must stay below iend - LASTLITERALS + 1 */
                const bool m_ext = M == ML_MASK;
                const uint32_t ext_m = (o4 >> 16) & 0xffu;
                const uint32_t next = (uint32_t)lane + hdr + L + (m_ext ? 3u : 2u);
                const uint32_t mlen = M + MINMATCH + (m_ext ? ext_m : 0u);
                fast = fast && offset != 0u && (!m_ext || (ext_m != 255u && ipu + next + (LASTLITERALS - 1u) < iendu));
                const bool general = cls_g || m_ext || offset < 8u;   /* not the shortcut's match stage */
                const uint32_t outlen = L + mlen;
                const uint32_t packed = token_word(fast, next, lane);

                /* follow the true chain from hypothesis 0: one v_readlane per real sequence */
                unsigned long long T = 0;
                uint32_t idx = 0;
                K4_PHASE("spec-chain");
                const unsigned long long tp1 = prof_now<PROF>();
                follow_tokens(packed, T, idx, HOP2);
                K4_PHASE("spec-scan-rules");
                const unsigned long long tp2 = prof_now<PROF>();
                /* output position of every chosen sequence: prefix sum of the chosen lengths */
                bool in_t = ((T >> lane) & 1ull) != 0;
                const uint32_t incl = wave_inclusive_scan(in_t ? outlen : 0u);
                const uint32_t v_o = opu + (incl - (in_t ? outlen : 0u));
                const uint32_t total = readlane_u32(incl, 63);
                /* position-dependent rules on the chosen sequences: the shortcut needs
                 * op <= shortoend (:191), a 15+ literal run cpy <= oend - MFLIMIT (:247); the offset
                 * must stay inside the output (:338); sequences
                 * that left the shortcut also obey the end-of-block rule (:427-433).  The first
                 * sequence that fails, and everything after it, is left to the scalar parser. */
                const uint32_t mdst_l = v_o + L;
                unsigned long long bad;
                if (opu + total + 64u <= oendu) {
                    /* every chosen sequence ends at least 64 bytes before the end of the output: the three
                     * end-of-block rules hold for all of them, only the offset can be wrong */
                    bad = ballot(in_t && offset > mdst_l);
                } else {
                    bad = ballot(in_t && ((cls_g ? mdst_l + (uint32_t)MFLIMIT > oendu : v_o + 14u + 18u > oendu) || offset > mdst_l ||
                                            (general && mdst_l + mlen + (uint32_t)MATCH_SAFEGUARD > oendu)));
                }
                uint32_t adv_op = total;
                if (bad) {
                    const int b = ctz64(bad);
                    T &= (1ull << b) - 1ull;
                    idx = (uint32_t)b;
                    adv_op = readlane_u32(incl, b) - readlane_u32(outlen, b);
                    in_t = ((T >> lane) & 1ull) != 0;
                }
                K4_PHASE("spec-slots");
                const unsigned long long tp3 = prof_now<PROF>();
                if (PROF) { c_hyp += tp1 - tp0; c_chain += tp2 - tp1; c_rules += tp3 - tp2; n_spec++; }
                if (T) {
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(T >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)T, 0u));
                    if (in_t) {
                        const uint32_t slot = (uint32_t)nseq + below;
                        d_lpos[slot] = pl + hdr;
                        d_llen[slot] = L;
                        d_out[slot] = v_o;
                        d_moff[slot] = offset;
                        d_mlen[slot] = mlen;
                    }
                    nseq += __popcll(T);
                    ip += idx;
                    op += adv_op;
                    if (PROF) c_slots += prof_now<PROF>() - tp3;
                    continue;
                }
            }

            /* ---- scalar parser: one sequence, the reference's order of checks ---- */
            K4_PHASE("scalar-parser");
            if (PROF) n_slow++;
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
                if (length != ML_MASK && offset >= 8u && (prefix64 || match >= low_prefix)) {   /* :213 */
                    s_moff = offset; s_mlen = length + MINMATCH; adv = s_mlen;
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
                    if (partial) {                             /* :250-270: stop early, never past oend */
                        if (ip + (int64_t)length > iend - (2 + 1 + LASTLITERALS) && ip + (int64_t)length != iend) { err = (int)(-ip) - 1; break; }
                        bool at_end = cpy == oend;
                        if (cpy > oend) { length = (uint32_t)(oend - op); s_llen = length; at_end = true; }
                        ip += length;
                        op += length;
                        if (at_end || ip == iend) {
                            last = true;
                            need_match = false;
                        } else {                               /* :303: keep going with the match */
                            offset = win.fetch((uint32_t)ip, lane) & 0xffffu;
                            ip += 2;
                            match = op - (int64_t)offset;
                            length = token & ML_MASK;
                        }
                    } else {
                        if (ip + (int64_t)length != iend || cpy > oend) { err = (int)(-ip) - 1; break; }
                        ip += length;
                        op += length;
                        last = true;
                        need_match = false;
                    }
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
                if (check_offset && match + chk_size < low_prefix) { err = (int)(-ip) - 1; break; }   /* :338 */
                const int64_t cpy = op + (int64_t)length;
                if (dict.mode == 2 && match < 0) {                 /* :342-378 match starts in the external dictionary */
                    if (cpy > oend - LASTLITERALS) {
                        if (!partial) { err = (int)(-ip) - 1; break; }
                        if ((int64_t)length > oend - op) length = (uint32_t)(oend - op);
                    }
                    s_moff = offset;
                    s_mlen = length;
                    adv = length;
                } else if (partial && cpy > oend - MATCH_SAFEGUARD) {     /* :387-406: truncated final match */
                    const uint32_t mlen = (int64_t)length < oend - op ? length : (uint32_t)(oend - op);
                    s_moff = offset;
                    s_mlen = offset != 0u ? mlen : 0u;
                    adv = mlen;
                    gap = gap || (offset == 0u && mlen != 0u);
                    if (op + (int64_t)mlen == oend) last = true;
                } else {
                    if (cpy > oend - MATCH_SAFEGUARD && cpy > oend - LASTLITERALS) { err = (int)(-ip) - 1; break; }  /* :427-433 */
                    s_moff = offset;
                    s_mlen = offset != 0u ? length : 0u;       /* offset 0 (hostile): output left as is */
                    adv = length;
                    gap = gap || offset == 0u;
                }
            }
            if (lane == 0) {
                d_lpos[nseq] = s_lpos; d_llen[nseq] = s_llen; d_out[nseq] = s_out; d_moff[nseq] = s_moff; d_mlen[nseq] = s_mlen;
            }
            nseq++;
            op += adv;
            if (last) done = true;
        }
        K4_PHASE("batch-end");
        if (ROLE == 1) {                                    /* publish the batch (or the failure) and go on parsing */
            if (lane == 0) {
                meta[0] = err ? 0u : (uint32_t)nseq;
                meta[1] = (uint32_t)op_batch;
                meta[2] = err ? 0u : (uint32_t)(op - op_batch);
                meta[3] = (err || done) ? 1u : 0u;
                meta[4] = (uint32_t)(err ? err : (int)op);
                meta[5] = gap ? 1u : 0u;
            }
            pipe_store(pipe + 0, batch_no + 1u, lane);
            batch_no++;
#if K4_DEC_PACE
            /* late blocks first (k4lz4_common.hpp, Pace): every K4_DEC_PACE_STEP compressed bytes the parsing wave takes its block's
             * priority and leaves it in pipe[3] for the copying wave */
            if (pace && (((uint32_t)ip ^ ip_batch) >> K4_DEC_PACE_STEP) != 0u && !done && !err) {
                const int level = Pace::update<K4_DEC_PACE_EPOCH, K4_DEC_PACE_DEN>(pace, pipe + 2, (uint32_t)ip, (uint32_t)src_size, lane);
                if (level >= 0 && lane == 0) pipe[3] = (uint32_t)level + 1u;
            }
#endif
            if (PROF) { c_parse += prof_now<PROF>() - t0; n_batch++; n_seq += (unsigned long long)nseq; }
            if (err || done) {
                if (seq) *seq = batch_no;
                if (PROF && pc && lane == 0) {
                    pc[0] = prof_now<PROF>() - t_begin; pc[1] = c_wait; pc[2] = c_parse; pc[4] = n_batch; pc[6] = n_seq; pc[7] = n_slow;
                    pc[11] = c_hyp; pc[12] = c_chain; pc[13] = c_rules; pc[14] = c_slots; pc[15] = n_spec;
                }
                return err ? err : (int)op;
            }
            continue;
        }
        if (ROLE == 0 && err) return err;
        wave_sync();
        const bool mine = lane < nseq;
        const uint32_t v_lpos = mine ? d_lpos[lane] : 0u;
        const uint32_t v_llen = mine ? d_llen[lane] : 0u;
        const uint32_t v_out = mine ? d_out[lane] : 0u;
        const uint32_t v_moff = mine ? d_moff[lane] : 0u;
        const uint32_t v_mlen = mine ? d_mlen[lane] : 0u;
        if (ROLE == 2) {                                    /* the descriptors are in registers: the slot may be refilled */
            pipe_store(pipe + 1, batch_no + 1u, lane);
            batch_no++;
        }
        const unsigned long long t1 = prof_now<PROF>();

        const uint32_t o0 = (uint32_t)op_batch, T = (uint32_t)(op - op_batch);
        const bool has_m = mine && v_mlen != 0u;
        /* matches that start before the block (dictionary) keep the general path below */
        const unsigned long long neg_m = ballot(has_m && v_moff > v_out + v_llen);
        unsigned long long t2 = t1;
        if (T <= (uint32_t)DECODE_STAGE_BYTES && neg_m == 0ull && !gap) {
            /* ======================= STAGED: the batch's output lives in LDS while it is assembled ==========
             * All global loads of the batch go out first -- literals from the compressed stream, and the
             * match sources that lie before the batch's output (they are final).  Matches that read this
             * batch's own output take it from the LDS stage, so the dependency rounds never wait for global
             * memory; the finished batch is then written out in one coalesced pass.
             * A lane moves up to 32 bytes at a time.  Longer literal runs and longer sources from before the batch are cut
             * into 16-byte pieces that are dealt out to the lanes -- one load instruction then serves all the long
             * items of a batch (`long_pieces`); a longer copy inside the stage, or one that overlaps its own output
             * (offset < length: the reference's byte-serial semantics, LL64.dec.cs:408-450), takes several rounds of its lane,
             * each round a piece that lies entirely behind its source, or -- periods under 8 bytes, copies over 128 bytes --
             * is made by the whole wave (`stage_wave_copy`). */
            uint8_t *stg = stage_base;
            const uint32_t mdst = v_out + v_llen, mend = mdst + v_mlen, msrc = mdst - v_moff;
            const uint32_t before = has_m && msrc < o0 ? (o0 - msrc < v_mlen ? o0 - msrc : v_mlen) : 0u;   /* source bytes before the batch */
            const bool lit_long = mine && v_llen > LANE_COPY_MAX, bef_long = before > LANE_COPY_MAX;
            LaneRun L, M;
            lane_run_load(L, in + v_lpos, (mine && !lit_long) ? v_llen : 0u, (uint32_t)src_size - v_lpos);
            lane_run_load(M, out + msrc, bef_long ? 0u : before, (uint32_t)out_size - msrc);
            /* dependencies among the matches of the batch, as below */
            const uint32_t send = msrc + v_mlen < mdst ? msrc + v_mlen : mdst;
            lds_sync();
            w_out[lane] = mine ? mdst : 0xffffffffu;
            w_end[lane] = mine ? mend : 0xffffffffu;
            lds_sync();
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (uint32_t step = 32; step != 0; step >>= 1) {
                if (w_end[lo + step - 1u] <= msrc) lo += step;
                if (w_out[hi + step - 1u] < send) hi += step;
            }
            const uint32_t hi_c = hi < (uint32_t)lane ? hi : (uint32_t)lane;
            unsigned long long deps = 0;
            const bool later = has_m && before < v_mlen;            /* some source bytes are this batch's output */
            if (later && lo < hi_c) deps = ((hi_c >= 64u ? 0ull : (1ull << hi_c)) - 1ull) & ~((1ull << lo) - 1ull);
            if (PROF) c_search += prof_now<PROF>() - t1;
            if (mine && v_llen != 0u && !lit_long) lane_run_store(stg + (v_out - o0), L, v_llen);
            if (before != 0u && !bef_long) lane_run_store(stg + (mdst - o0), M, before);
            if (ballot(lit_long)) long_pieces(stg, w_out, in, v_lpos, lit_long ? v_llen : 0u, v_out - o0, lane);
            if (ballot(bef_long)) long_pieces(stg, w_out, out, msrc, bef_long ? before : 0u, mdst - o0, lane);
            if (PROF) t2 = prof_now<PROF>();
            /* the part of every match that comes out of the stage: stage[src_s ..) -> stage[dst_s ..), `n` bytes */
            const uint32_t n = later ? v_mlen - before : 0u;
            const uint32_t dst_s = mdst + before - o0, src_s = msrc + before - o0;
            const bool overlap = later && v_moff < v_mlen;
            const bool by_wave = later && (n > (uint32_t)K4_DEC_WAVE_AT || (overlap && v_moff < 8u));
            const uint32_t piece = overlap && v_moff < LANE_COPY_MAX ? v_moff : LANE_COPY_MAX;    /* what one round of the lane may move */
            uint32_t moved = 0;
            unsigned long long pend = ballot(later);
            while (pend) {
                if (PROF) n_round++;
                const bool ready = ((pend >> lane) & 1ull) != 0 && (deps & pend) == 0;
                unsigned long long fin = ballot(ready && by_wave);
                lds_sync();
                if (ready && !by_wave) {
                    const uint32_t c = n - moved < piece ? n - moved : piece;
                    lane_move32_slack(stg + dst_s + moved, stg + src_s + moved, c);   /* the stage has slack behind it */
                    moved += c;
                }
                for (unsigned long long big = fin; big; big &= big - 1ull) {
                    const int g = ctz64(big);
                    stage_wave_copy(stg, readlane_u32(dst_s, g), readlane_u32(v_moff, g), readlane_u32(n, g), lane);
                }
                fin |= ballot(ready && !by_wave && moved == n);
                pend &= ~fin;
            }
            lds_sync();
            const unsigned long long tc = prof_now<PROF>();
            if (PROF) c_rounds += tc - t2;
            /* the finished batch leaves the stage in one pass, 16 bytes per lane */
            for (uint32_t k = 16u * (uint32_t)lane; k + 16u <= T; k += 1024u) {
                const uint4 v = *(const uint4 *)(stg + k);
                U128u o;
                o.v[0] = v.x; o.v[1] = v.y; o.v[2] = v.z; o.v[3] = v.w;
                st128u(out + o0 + k, o);
            }
            if (T & 15u) {                                          /* the last, partial 16 bytes: once more as the batch's last 16 */
                if (T >= 16u) {
                    if (lane == 0) st128u(out + o0 + T - 16u, ld128u(stg + T - 16u));
                } else if ((uint32_t)lane < T) {
                    out[o0 + (uint32_t)lane] = stg[lane];
                }
            }
            if (PROF) c_flush += prof_now<PROF>() - tc;
        } else {
        /* ======================= LITERALS ======================= */
        {
            if (mine && v_llen != 0u && v_llen <= LANE_COPY_MAX)
                lane_copy32(out + v_out, in + v_lpos, v_llen, (uint32_t)src_size - v_lpos);
            unsigned long long big = ballot(mine && v_llen > LANE_COPY_MAX);
            while (big) {
                const int f = ctz64(big);
                big &= big - 1;
                wave_copy(out + __builtin_amdgcn_readlane(v_out, f), in + __builtin_amdgcn_readlane(v_lpos, f),
                          __builtin_amdgcn_readlane(v_llen, f), lane);
            }
        }
        if (PROF) t2 = prof_now<PROF>();

        /* ======================= MATCHES ======================= */
        {
            const bool has = mine && v_mlen != 0u;
            const uint32_t mdst = v_out + v_llen;
            const uint32_t mend = mdst + v_mlen;
            const uint32_t msrc = mdst - v_moff;
            const uint32_t send = msrc + v_mlen < mdst ? msrc + v_mlen : mdst;   /* source bytes below own output */
            /* destination ranges sorted by lane: publish [mdst, mend) (sentinel for idle lanes) */
            wave_sync();
            w_out[lane] = mine ? mdst : 0xffffffffu;
            w_end[lane] = mine ? mend : 0xffffffffu;
            wave_sync();
            /* first lane whose match ends above msrc, first lane whose match starts at/after send */
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (uint32_t step = 32; step != 0; step >>= 1) {
                if (w_end[lo + step - 1u] <= msrc) lo += step;
                if (w_out[hi + step - 1u] < send) hi += step;
            }
            /* dependencies: lanes [lo, hi) below this lane */
            const uint32_t hi_c = hi < (uint32_t)lane ? hi : (uint32_t)lane;
            unsigned long long deps = 0;
            if (has && lo < hi_c) deps = ((hi_c >= 64u ? 0ull : (1ull << hi_c)) - 1ull) & ~((1ull << lo) - 1ull);
            /* a match that starts before the block (dictionary) waits for everything below it and is
             * moved by the whole wave */
            const bool neg = has && v_moff > mdst;
            if (neg) deps = lane == 0 ? 0ull : ((1ull << lane) - 1ull);
            const unsigned long long negmask = ballot(neg);
            const bool coop = v_mlen > LANE_COPY_MAX || v_moff < v_mlen || neg;
            unsigned long long pend = ballot(has);
            while (pend) {
                if (PROF) n_round++;
                const bool ready = ((pend >> lane) & 1ull) != 0 && (deps & pend) == 0;
                const unsigned long long rmask = ballot(ready);
                wave_sync();
                if (ready && !coop) lane_copy32(out + mdst, out + msrc, v_mlen, (uint32_t)out_size - msrc);
                unsigned long long big = ballot(ready && coop);
                while (big) {
                    const int g = ctz64(big);
                    big &= big - 1;
                    const uint32_t g_dst = __builtin_amdgcn_readlane(mdst, g), g_off = __builtin_amdgcn_readlane(v_moff, g),
                                   g_len = __builtin_amdgcn_readlane(v_mlen, g);
                    if ((negmask >> g) & 1ull) wave_dict_copy(out, dict.end, g_dst, g_off - g_dst, g_len, lane);
                    else wave_match_copy(out, g_dst, g_off, g_len, lane);
                }
                pend &= ~rmask;
            }
            wave_sync();
        }
        }
        if (PROF) {
            const unsigned long long t3 = prof_now<PROF>();
            c_parse += t1 - t0; c_lit += t2 - t1; c_match += t3 - t2; n_batch++; n_seq += (unsigned long long)nseq;
        }
        if (ROLE == 2 && done) {                            /* the result the parsing wave arrived at */
            if (seq) *seq = batch_no;
            if (PROF && pc && lane == 0) {
                pc[0] = prof_now<PROF>() - t_begin; pc[1] = c_wait; pc[2] = c_parse; pc[3] = c_search; pc[4] = c_lit; pc[5] = c_rounds;
                pc[6] = c_flush; pc[7] = c_match; pc[8] = n_batch; pc[9] = n_round; pc[10] = n_seq;
            }
            return err;
        }
        if (done) break;
    }
    if (PROF && pc && lane == 0) {
        pc[0] = prof_now<PROF>() - t_begin; pc[1] = c_parse; pc[2] = c_lit; pc[3] = c_match;
        pc[4] = n_batch; pc[5] = n_round; pc[6] = n_seq; pc[7] = n_slow;
        pc[11] = c_hyp; pc[12] = c_chain; pc[13] = c_rules; pc[14] = c_slots; pc[15] = n_spec;
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

constexpr int DECODE_WAVES_PER_WG = 2;       /* measured on 1 M x 4 KiB blocks: 4 waves per workgroup 286 GiB/s, 2 waves 297 */

/* LL64.LZ4_decompress_safe_usingDict (LL64.dec.cs:523-546): no dictionary / prefix / external */
__device__ __forceinline__ DecodeDict block_dict(const BatchArgs &a, long long b, const uint8_t *out)
{
    DecodeDict d{nullptr, 0u, 0};
    if (!a.dict || !a.dictLen) return d;
    const int len = a.dictLen[b];
    if (len <= 0) return d;
    const uint8_t *p = a.dict + a.dictOff[b];
    const bool prefix = a.dictMode ? a.dictMode[b] == 1 : p + len == out;
    d.end = p + len;
    d.size = prefix ? (len >= 65535 ? 65536u : (uint32_t)len) : (uint32_t)len;
    d.mode = prefix ? 1 : 2;
    return d;
}

__device__ __forceinline__ void decode_kernel_body(const BatchArgs &a, uint32_t (*lds)[DECODE_LDS_DWORDS])
{
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const long long slot = (long long)blockIdx.x * DECODE_WAVES_PER_WG + (long long)wave;
    if (slot >= a.n) return;
    const long long b = a.order ? (long long)uni(a.order[slot]) : slot;    /* K4LZ4_FLAG_REORDER: longest first */
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    const uint8_t *in = a.src + a.srcOff[b];
    uint8_t *out = a.dst + a.dstOff[b];
    int ret = 0;
    if (src_len > 0 || (a.flags & FLAG_RAW_RETURN)) ret = decode_block(in, src_len, out, cap < 0 ? 0 : cap, lane, lds[wave], nullptr, (a.flags & FLAG_PARTIAL) != 0,
                           block_dict(a, b, out));
    if (lane == 0) a.outLen[b] = codec_decode_result(src_len, ret, a.flags);
}

/* the batch fits on the chip at once: a block's latency is what counts, the compiler may use the registers it likes */
__global__ __launch_bounds__(64 * DECODE_WAVES_PER_WG) void k4_decode_kernel(BatchArgs a)
{
    __shared__ uint32_t lds[DECODE_WAVES_PER_WG][DECODE_LDS_DWORDS];
    decode_kernel_body(a, lds);
}

/* ... and with at most half as many blocks as the chip has wave slots, two waves per block: wave 2p parses block p of
 * the workgroup, wave 2p+1 copies (see the queue above).  8 waves per SIMD need <= 64 VGPRs. */
#ifndef K4_DEC_PAIRS
#define K4_DEC_PAIRS 2
#endif
constexpr int DECODE_PAIRS_PER_WG = K4_DEC_PAIRS;     /* measured: 4 pairs per workgroup 233 GiB/s on the bench batch, 2 pairs 242, 1 pair 223; with the late-blocks-first priorities 260 / 290 / 265 */
/* HOP2: the parsing wave follows the token chain two links at a time (follow_tokens) -- the kernel for launches that leave wave slots free */
template <bool HOP2>
__device__ __forceinline__ void decode_pair_kernel_body(const BatchArgs &a, uint32_t (*lds)[DECODE_PAIR_LDS_DWORDS])
{
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    /* odd workgroups swap the roles, so that a SIMD hosts parsing and copying waves alike */
    const uint32_t pair = wave >> 1, role = (wave ^ blockIdx.x) & 1u;
    const long long slot = (long long)blockIdx.x * DECODE_PAIRS_PER_WG + (long long)pair;
    uint32_t *ring = lds[pair], *pipe = lds[pair] + RING_DWORDS;
    if (role == 0) pipe_init(pipe, a.status, lane);
    __syncthreads();
    if (slot >= a.n) return;
    const long long b = a.order ? (long long)uni(a.order[slot]) : slot;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    const uint8_t *in = a.src + a.srcOff[b];
    uint8_t *out = a.dst + a.dstOff[b];
    const bool run = src_len > 0 || (a.flags & FLAG_RAW_RETURN);
    const bool partial = (a.flags & FLAG_PARTIAL) != 0;
    const DecodeDict dict = block_dict(a, b, out);
    if (role == 0) {
        if (a.prof) prof_place<true>(a.prof + PROF_STRIDE * b, 8, lane);
        if (K4_DEC_PACE) Pace::begin(a.pace, pipe + 2, lane);
        if (run) decode_block<false, 1, HOP2>(in, src_len, out, cap < 0 ? 0 : cap, lane, ring, nullptr, partial, dict, pipe, nullptr, a.pace);
    } else {
        int ret = 0;
        if (run) ret = decode_block<false, 2>(in, src_len, out, cap < 0 ? 0 : cap, lane, ring, nullptr, partial, dict, pipe, nullptr, a.pace);
        if (lane == 0) a.outLen[b] = codec_decode_result(src_len, ret, a.flags);
        if (a.prof) prof_place<true>(a.prof + PROF_STRIDE * b, 9, lane);
    }
}
__global__ __launch_bounds__(128 * DECODE_PAIRS_PER_WG) __attribute__((amdgpu_waves_per_eu(8, 8))) void k4_decode_pair_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[DECODE_PAIRS_PER_WG][DECODE_PAIR_LDS_DWORDS];
    decode_pair_kernel_body<false>(a, lds);
}
__global__ __launch_bounds__(128 * DECODE_PAIRS_PER_WG) __attribute__((amdgpu_waves_per_eu(8, 8))) void k4_decode_pair2_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[DECODE_PAIRS_PER_WG][DECODE_PAIR_LDS_DWORDS];
    decode_pair_kernel_body<true>(a, lds);
}

/* diagnostic twin of the pair kernel: 32 counters per block, the parsing wave's in [0, 16), the copying wave's in [16, 32) */
__global__ __launch_bounds__(128 * DECODE_PAIRS_PER_WG) void k4_decode_pair_prof_kernel(BatchArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[DECODE_PAIRS_PER_WG][DECODE_PAIR_LDS_DWORDS];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t pair = wave >> 1, role = (wave ^ blockIdx.x) & 1u;
    const long long b = (long long)blockIdx.x * DECODE_PAIRS_PER_WG + (long long)pair;
    uint32_t *ring = lds[pair], *pipe = lds[pair] + RING_DWORDS;
    if (role == 0) pipe_init(pipe, a.status, lane);
    __syncthreads();
    if (b >= a.n) return;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    const uint8_t *in = a.src + a.srcOff[b];
    uint8_t *out = a.dst + a.dstOff[b];
    const DecodeDict dict{nullptr, 0u, 0};
    if (src_len <= 0) return;
    if (role == 0) {
        decode_block<true, 1>(in, src_len, out, cap < 0 ? 0 : cap, lane, ring, a.prof + 2 * PROF_STRIDE * b, false, dict, pipe);
    } else {
        const int ret = decode_block<true, 2>(in, src_len, out, cap < 0 ? 0 : cap, lane, ring, a.prof + 2 * PROF_STRIDE * b + PROF_STRIDE, false, dict, pipe);
        if (lane == 0) a.outLen[b] = codec_decode_result(src_len, ret, a.flags);
    }
}

/* many more blocks than the chip holds: throughput counts, so one more wave per SIMD (<= 72 VGPRs) is worth the
 * few spills (measured on 1 M x 4 KiB: +6.5 %; on the 4096 x 64 KiB batch the other variant is 3 % faster) */
__global__ __launch_bounds__(64 * DECODE_WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(7, 7))) void k4_decode_dense_kernel(BatchArgs a)
{
    __shared__ uint32_t lds[DECODE_WAVES_PER_WG][DECODE_LDS_DWORDS];
    decode_kernel_body(a, lds);
}

/* diagnostic twin: same decode with per-phase cycle counters (a.prof, PROF_STRIDE per block) */
__global__ __launch_bounds__(64 * DECODE_WAVES_PER_WG) void k4_decode_prof_kernel(BatchArgs a)
{
    __shared__ uint32_t lds[DECODE_WAVES_PER_WG][DECODE_LDS_DWORDS];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const long long b = (long long)blockIdx.x * DECODE_WAVES_PER_WG + (long long)wave;
    if (b >= a.n) return;
    const int src_len = a.srcLen[b];
    const int cap = a.dstCap[b];
    int ret = 0;
    if (src_len > 0) ret = decode_block<true>(a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], cap < 0 ? 0 : cap, lane, lds[wave], a.prof + PROF_STRIDE * b);
    if (lane == 0) a.outLen[b] = codec_decode_result(src_len, ret, a.flags);
}

}  // namespace k4
