/*
 * k4lz4_common.hpp -- shared device-side helpers for the gfx950 LZ4 block kernels.
 *
 * Execution model used throughout: ONE WAVEFRONT (64 lanes) OWNS ONE LZ4 BLOCK.  All control
 * state of a block (input/output cursors, token fields) is wave-uniform and lives in SGPRs; the
 * 64 lanes are used for the data-parallel pieces: coalesced byte moves, 64 speculative
 * hash-probes at once, match-length counting, run-length fills.  Cross-lane traffic goes through
 * ballot / readlane / shuffles, or through LDS/global memory separated by wave_sync().
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace k4 {

/* Block-format constants: reference src/K4os.Compression.LZ4/Engine/LL.types.cs:18-78 */
enum : int {
    MINMATCH = 4,
    LASTLITERALS = 5,
    MFLIMIT = 12,
    MATCH_SAFEGUARD = 12,
    ML_BITS = 4,
    ML_MASK = 15,
    RUN_MASK = 15,
    DISTANCE_MAX = 65535,
    SKIP_TRIGGER = 6,
    LIMIT_64K = 65536 + (MFLIMIT - 1),
    MAX_INPUT_SIZE = 0x7E000000,
};

/* per-batch flags (k4lz4.h) */
enum : int {
    FLAG_RAW_RETURN = 1,  /* outLen = LLxx-level return instead of the LZ4Codec mapping */
    FLAG_PICKLE_WRITER = 2,
    FLAG_X32 = 128,        /* fast encoder: LZ4Codec.Enforce32 -- the 32-bit engine's hash for inputs of 64 KiB and more */
    FLAG_PARTIAL = 32,     /* decode: LZ4_decompress_safe_partial semantics, dstCap = target size */
};

struct BatchArgs {
    const uint8_t *src;
    const uint64_t *srcOff;
    const int32_t *srcLen;
    uint8_t *dst;
    const uint64_t *dstOff;
    const int32_t *dstCap;
    int32_t *outLen;
    long long n;
    int level;   /* encode: LZ4Level; */
    int accel;   /* fast encoder acceleration (LZ4Codec always passes 1) */
    int flags;
    unsigned long long *prof;  /* diagnostics: PROF_STRIDE counters per block, or nullptr */
    const uint32_t *order;     /* dispatch order (block index per workgroup slot), or nullptr */
    uint32_t *cost;            /* scheduling scratch: cost bucket per block */
    uint32_t *hist;            /* scheduling scratch: 2 x COST_BUCKETS counters (zeroed) */
    uint32_t *order_out;       /* scheduling scratch: the order being built */
    uint32_t *gtab;            /* k4_encode_fast_gtab_kernel: 4096 dwords of table per workgroup */
    const uint8_t *dict;       /* decode with dictionaries: packed dictionaries, or nullptr */
    const uint64_t *dictOff;
    const int32_t *dictLen;
    const signed char *dictMode;   /* optional: 1 = prefix semantics, 2 = external (host staging); nullptr = by address */
    uint32_t *status;              /* the context's status word (DEV_STATUS_* bits, raised with dev_status_raise), or nullptr */
    uint32_t *pace;                /* three zeroed words shared by the waves of a launch (Pace below), or nullptr */
    const uint32_t *split;         /* the two encoder kernels of one batch: where in the dispatch order the second one's part begins
                                    * (hist[2 * COST_BUCKETS], written by k4_order_kernel), or nullptr: `first` alone says */
    uint32_t first;                /* global-table kernel: its slot 0 is entry first (+ *split) of the order */
    uint32_t total;                /* ... and the order has this many entries */
    const int32_t *seg_first;      /* big blocks cut into segments (k4lz4_segments.hpp): per block its first segment's record, or -1; or nullptr */
    void *seg_items;               /* ... the records (SegItem) */
    uint32_t *seg_snaps;           /* ... and where the segments' runs publish their cuts */
};

struct __attribute__((packed, aligned(1))) U16u { uint16_t v; };
struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };
struct __attribute__((packed, aligned(1))) U64u { uint64_t v; };
struct __attribute__((packed, aligned(1))) U128u { uint32_t v[4]; };

__device__ __forceinline__ uint32_t ld32u(const uint8_t *p) { return ((const U32u *)p)->v; }
__device__ __forceinline__ uint64_t ld64u(const uint8_t *p) { return ((const U64u *)p)->v; }
__device__ __forceinline__ U128u ld128u(const uint8_t *p) { return *(const U128u *)p; }
__device__ __forceinline__ void st128u(uint8_t *p, U128u v) { *(U128u *)p = v; }
/* the same load with the non-temporal hint (the L2 keeps such lines shortest): the fast encoder's candidate bytes under K4_NT_CAND, an A/B build */
typedef uint32_t k4_v4u __attribute__((ext_vector_type(4)));
typedef k4_v4u __attribute__((aligned(1))) k4_v4u_u;
__device__ __forceinline__ U128u ld128u_nt(const uint8_t *p)
{
#ifndef K4_HOST_EMU
    const k4_v4u v = __builtin_nontemporal_load((const k4_v4u_u *)p);
    U128u r; r.v[0] = v.x; r.v[1] = v.y; r.v[2] = v.z; r.v[3] = v.w;
    return r;
#else
    return *(const U128u *)p;
#endif
}

/* write-once / read-once data (the two-step encoder's records, the encoded bytes) with the non-temporal hint, so that they do not
 * push the source lines the candidate fetches come back for out of the L2 and the Infinity Cache (K4_NT_RECS / K4_NT_OUT: A/B builds) */
typedef uint64_t __attribute__((aligned(1))) k4_u64u;
typedef uint32_t __attribute__((aligned(1))) k4_u32u;
typedef uint16_t __attribute__((aligned(1))) k4_u16u;
__device__ __forceinline__ void st64u_out(uint8_t *p, uint64_t v)
{
#if defined(K4_NT_OUT) && !defined(K4_HOST_EMU)
    __builtin_nontemporal_store(v, (k4_u64u *)p);
#else
    ((U64u *)p)->v = v;
#endif
}
__device__ __forceinline__ void st32u_out(uint8_t *p, uint32_t v)
{
#if defined(K4_NT_OUT) && !defined(K4_HOST_EMU)
    __builtin_nontemporal_store(v, (k4_u32u *)p);
#else
    ((U32u *)p)->v = v;
#endif
}
__device__ __forceinline__ void st16u_out(uint8_t *p, uint16_t v)
{
#if defined(K4_NT_OUT) && !defined(K4_HOST_EMU)
    __builtin_nontemporal_store(v, (k4_u16u *)p);
#else
    ((U16u *)p)->v = v;
#endif
}
__device__ __forceinline__ void st8_out(uint8_t *p, uint8_t v)
{
#if defined(K4_NT_OUT) && !defined(K4_HOST_EMU)
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

/* Orders this wave's earlier LDS/global accesses (any lane) before its later ones (any lane).
 * The hardware executes a wave's memory instructions in order; this pins the compiler. */
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* scripts/isa_phase_table.py builds with -DK4_PHASE_MARKS and counts the instructions between these comments in the ISA listing
 * (a marker is an empty volatile asm that clobbers memory: it pins the phases' order in THAT build only; normal builds have none) */
#ifdef K4_PHASE_MARKS
#define K4_PHASE(name) asm volatile("; k4phase " name ::: "memory")
#else
#define K4_PHASE(name) ((void)0)
#endif

constexpr int PROF_STRIDE = 16;

/* Call-level trouble that is not a property of any block's data: bit 0 = a wave of a decoder pair gave up waiting for
 * its partner (scheduling, never corrupt input), bit 1 = the HC scratch reserved for an asynchronous call was too small.
 * The word belongs to the CONTEXT whose call launched the kernel (every args struct carries its address), so that one
 * context's trouble is never reported to -- or cleared by -- another context on the same device.  The host reads and
 * clears it whenever it synchronises (k4lz4_synchronize, every host-pointer call) and reports all the bits it finds;
 * the affected blocks' outLen only say "failed". */
enum : uint32_t { DEV_STATUS_PIPE_TIMEOUT = 1u, DEV_STATUS_HC_SCRATCH = 2u };
constexpr int HC_NO_SCRATCH = -0x7ffffff1;   /* LLxx-level result of an HC block that was not encoded for want of scratch (<= 0: failure) */
__device__ __forceinline__ void dev_status_raise(uint32_t *status, uint32_t bits)
{
    if (status) atomicOr(status, bits);
}

/* placement record of the diagnostic kernels: [8] start, [9] end (100 MHz real-time counter), [10] HW_ID */
template <bool PROF> __device__ __forceinline__ void prof_place(unsigned long long *pc, int slot, int lane)
{
    if (!PROF || !pc) return;
#ifndef K4_HOST_EMU
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    if (lane == 0) { pc[slot] = t; if (slot == 8) pc[10] = hw; }
#else
    (void)slot; (void)lane;
#endif
}

/* phase timestamps for the diagnostic kernels: drain this wave's memory traffic, then read the
 * shader clock, so that a phase's cycles include its own memory waits */
template <bool PROF> __device__ __forceinline__ unsigned long long prof_now()
{
    if (!PROF) return 0ull;
#if !defined(K4_HOST_EMU) && !defined(K4_PROF_NODRAIN)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    return (unsigned long long)__builtin_readcyclecounter();
}

/* The lanes where `p` holds.  HIP's __ballot takes an int: the predicate is widened into a vector register and compared
 * with zero again (two vector instructions per ballot where the predicate was a lane mask already); the builtin takes it as it is. */
__device__ __forceinline__ unsigned long long ballot(bool p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(p);
#else
    return __ballot(p ? 1 : 0);
#endif
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
/* the builtin returns int: widening its result directly would sign-extend */
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int ctz64(unsigned long long m) { return __ffsll(m) - 1; }

/* a word in memory that one wave publishes and waves of other workgroups read (release / acquire at agent scope) */
__device__ __forceinline__ void agent_publish(uint32_t *p, uint32_t v)
{
#ifndef K4_HOST_EMU
    __threadfence();
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
    __atomic_store_n(p, v, __ATOMIC_RELEASE);
#endif
}
__device__ __forceinline__ void agent_acquire()             /* what other workgroups published before the word just seen may be read plainly now */
{
#ifndef K4_HOST_EMU
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
#endif
}
__device__ __forceinline__ uint32_t agent_peek(const uint32_t *p)
{
#ifndef K4_HOST_EMU
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
#else
    return __atomic_load_n(p, __ATOMIC_ACQUIRE);
#endif
}

/*
 * Late blocks first.  A launch is as long as its last block, all blocks of a batch are resident from the start, and what a
 * wave gets of its SIMD's issue slots is decided by priority, then age -- luck of placement, which spreads the finishing
 * times of equally expensive blocks by +-10 %.  Priorities only arbitrate between the waves of ONE SIMD, so that is the
 * scope: every wave estimates, each time its block has advanced another PACE_STEP bytes, when the block will be done at the
 * pace so far (F = start + elapsed * total / done on the 100 MHz real-time counter, relative to the first start seen on
 * this SIMD), reports it to the SIMD's slot (atomic max into the word of the running EPOCH, tagged with the epoch number so
 * that a new epoch's reports supersede the old ones), reads what the epoch before arrived at, and takes an issue priority
 * by how close it is to that: on every SIMD the block that would finish last runs first.  One round trip to the L2 per
 * PACE_STEP bytes of a block (a slot is touched by the handful of waves of one SIMD: no contention; one word for the whole
 * launch was tried first and cost 35 %); nothing depends on it but timing.  Measured on the bench batch (4096 x 64 KiB, both
 * encoder kernels side by side): 62.2 -> 63.3 GiB/s over giving the younger kernel's waves the priority three steps in four;
 * tiers at 1/32 of the latest estimate, a report every 2 KiB, epochs of 164 us (1/16 and 1/8 tiers, 4 KiB steps measured
 * 62.9 / 61.8 / 63.0).  The pair decoders use it with BOTH waves of a pair at the block's priority (the parsing wave decides, the
 * copying wave reads the level from the pair's queue): 245 -> 290 GiB/s on the bench batch (k4lz4_decode.hpp, K4_DEC_PACE; with
 * the parsing wave alone at the priority it had lost 2 %).
 *   slot (16 bytes, index from XCC_ID and HW_ID: se, sh, cu, simd):  [0], [1]  epoch << 20 | F in 80 ns units, even / odd
 *   epochs      [2]  ~(earliest start on this SIMD)
 */
#ifndef K4_PACE_DEN
#define K4_PACE_DEN 32u
#endif
#ifndef K4_PACE_STEP
#define K4_PACE_STEP 11
#endif
#ifndef K4_PACE_EPOCH
#define K4_PACE_EPOCH 14
#endif
constexpr uint32_t PACE_STEP_LOG2 = K4_PACE_STEP, PACE_EPOCH_LOG2 = K4_PACE_EPOCH;          /* the encoders' (a 64 KiB block takes milliseconds) */
constexpr uint32_t PACE_SLOTS = 8192, PACE_BYTES = PACE_SLOTS * 16;
struct Pace {
    __device__ __forceinline__ static uint32_t now()
    {
#ifndef K4_HOST_EMU
        return (uint32_t)__builtin_amdgcn_s_memrealtime();
#else
        return 0u;
#endif
    }
    __device__ __forceinline__ static uint32_t *slot_of(uint32_t *base)
    {
#ifndef K4_HOST_EMU
        const uint32_t hw = __builtin_amdgcn_s_getreg((16 << 11) | 4);          /* HW_ID[15:0]: wave, simd, pipe, cu, sh, se */
        const uint32_t xcc = __builtin_amdgcn_s_getreg((4 << 11) | 20);         /* XCC_ID[3:0] */
        const uint32_t idx = ((hw >> 4) & 3u) | (((hw >> 8) & 0xffu) << 2) | ((xcc & 7u) << 10);
        return base + 4u * idx;
#else
        return base;
#endif
    }
    /* `mine`: a dword of LDS this wave keeps (its start time lives there, not in a register the round loop is short of) */
    __device__ __forceinline__ static void begin(uint32_t *base, uint32_t *mine, int lane)
    {
#ifndef K4_HOST_EMU
        if (!base) return;
        const uint32_t t = now();
        if (lane == 0) {
            *mine = t;
            atomicMax(slot_of(base) + 2, ~t);
        }
#else
        (void)base; (void)mine; (void)lane;
#endif
    }
    /* `done` of `total` units of the block are behind this wave (0 < done <= total) */
    __device__ __forceinline__ static void set_level(int level)
    {
#ifndef K4_HOST_EMU
        if (level >= 3) __builtin_amdgcn_s_setprio(3);
        else if (level == 2) __builtin_amdgcn_s_setprio(2);
        else if (level == 1) __builtin_amdgcn_s_setprio(1);
        else if (level == 0) __builtin_amdgcn_s_setprio(0);
#else
        (void)level;
#endif
    }
    /* returns the priority taken (0..3), or -1 when there was nothing to compare with yet */
    template <uint32_t EPOCH_LOG2 = PACE_EPOCH_LOG2, uint32_t DEN = K4_PACE_DEN>
    __device__ __forceinline__ static int update(uint32_t *base, const uint32_t *mine, uint32_t done, uint32_t total, int lane)
    {
#ifndef K4_HOST_EMU
        if (!base) return -1;
        uint32_t *w = slot_of(base);
        uint32_t ref = 0, t0inv = 0, cur = 0;
        if (lane == 0) {
            cur = __hip_atomic_load(w + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ref = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t0inv = __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint32_t t_start = uni(*mine);
        const uint32_t w0 = uni(cur), w1 = uni(ref), t0i = uni(t0inv), t0 = t0i ? ~t0i : t_start;
        const uint32_t t = now();
        const float own = (float)(t - t_start) * ((float)total / (float)done);
        const uint32_t own_u = own < 4.0e9f ? (uint32_t)own : 4000000000u;
        const uint32_t f = (t_start - t0) + own_u;                      /* ticks from the first start on this SIMD to this block's end */
        const uint32_t fs = (f >> 3) < 0xfffffu ? (f >> 3) : 0xfffffu;
        const uint32_t epoch = ((t - t0) >> EPOCH_LOG2) & 0xfffu;
        if (lane == 0) atomicMax(w + (epoch & 1u), (epoch << 20) | fs);
        const uint32_t before = (epoch & 1u) ? w0 : w1;                 /* the other word: the epoch before, if anybody reported in it */
        if ((before >> 20) != ((epoch - 1u) & 0xfffu)) return -1;
        const uint32_t last = before & 0xfffffu;
        const int level = fs * DEN >= last * (DEN - 1u) ? 3 : fs * DEN >= last * (DEN - 2u) ? 2 : fs * DEN >= last * (DEN - 3u) ? 1 : 0;
        set_level(level);
        return level;
#else
        (void)base; (void)mine; (void)done; (void)total; (void)lane;
        return -1;
#endif
    }
};


/* Inclusive prefix sum over the 64 lanes with DPP row shifts / row broadcasts (no LDS):
 * Hillis-Steele inside each row of 16, then row 15 -> row 1/3 and lane 31 -> rows 2,3. */
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   /* row_shr:1 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);   /* row_shr:2 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   /* row_shr:4 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);   /* row_shr:8 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  /* row_bcast:15 -> rows 1,3 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  /* row_bcast:31 -> rows 2,3 */
    return x;
}

/* LL.tools.cs:38-40 */
__device__ __forceinline__ int compress_bound(int n)
{
    return n > MAX_INPUT_SIZE ? 0 : n + n / 255 + 16;
}

/* Wave-cooperative copy of n bytes, regions must not overlap.  Long runs move 16 B per lane
 * (1 KiB per wave instruction, unaligned dwordx4), the rest one byte per lane. */
__device__ __forceinline__ void wave_copy(uint8_t *d, const uint8_t *s, uint32_t n, int lane)
{
    uint32_t done = 0;
    if (n >= 128) {
        uint32_t nv = n >> 4;
        for (uint32_t v = (uint32_t)lane; v < nv; v += 64) st128u(d + 16ull * v, ld128u(s + 16ull * v));
        done = nv << 4;
    }
    for (uint32_t k = done + (uint32_t)lane; k < n; k += 64) d[k] = s[k];
}

/* move n bytes down to a lower address that may lie inside them (d < s): ascending 1 KiB steps, every step loaded by all the
 * lanes before any of them stores */
__device__ __forceinline__ void wave_shift_down(uint8_t *d, const uint8_t *s, uint32_t n, int lane)
{
    for (uint32_t k0 = 0; k0 < n; k0 += 1024u) {
        const uint32_t k = k0 + 16u * (uint32_t)lane;
        U128u v = {{0, 0, 0, 0}};
        uint8_t tail[16];
        const bool full = k + 16u <= n;
        if (full) {
            v = ld128u(s + k);
        } else {
            for (uint32_t i = 0; i < 16u; i++) tail[i] = (k + i < n) ? s[k + i] : (uint8_t)0;
        }
        wave_sync();
        if (full) {
            st128u(d + k, v);
        } else {
            for (uint32_t i = 0; i < 16u; i++) if (k + i < n) d[k + i] = tail[i];
        }
        wave_sync();
    }
}

/* n bytes of value `b` */
__device__ __forceinline__ void wave_fill(uint8_t *d, uint8_t b, uint32_t n, int lane)
{
    for (uint32_t k = (uint32_t)lane; k < n; k += 64) d[k] = b;
}

constexpr uint32_t LANE_COPY_MAX = 32;

/* up to 7 bytes at p (fewer than 8 readable): little-endian assemble without reading past them */
__device__ __forceinline__ uint64_t load_tail(const uint8_t *p, uint32_t avail)
{
    uint64_t v = 0;
    for (uint32_t i = 0; i < 8u && i < avail; i++) v |= (uint64_t)p[i] << (8u * i);
    return v;
}

/* A run of up to 32 bytes held by one lane as its full 8-byte chunks plus a word with its LAST 8 bytes
 * (runs under 8 bytes: v[0] only).  Storing it then needs no byte-granular tail: the last word is written over
 * the end of the last full chunk (the same bytes twice), a 4..7 byte run as two overlapping dwords.  Nothing
 * outside [d, d + len) is touched.  `readable` = bytes that may be read from s on (>= len). */
struct LaneRun { uint64_t v[4]; uint64_t last; };

__device__ __forceinline__ void lane_run_load(LaneRun &r, const uint8_t *s, uint32_t len, uint32_t readable)
{
    r.v[0] = 0;
    if (len != 0u) r.v[0] = readable >= 8u ? ld64u(s) : load_tail(s, readable);
#pragma unroll
    for (uint32_t c = 1; c < 4u; c++) r.v[c] = 8u * c + 8u <= len ? ld64u(s + 8u * c) : 0ull;
    r.last = len >= 8u ? ld64u(s + len - 8u) : 0ull;
}

__device__ __forceinline__ void lane_run_store(uint8_t *d, const LaneRun &r, uint32_t len)
{
    if (len >= 8u) {
        ((U64u *)d)->v = r.v[0];
        if (len >= 16u) ((U64u *)(d + 8))->v = r.v[1];
        if (len >= 24u) ((U64u *)(d + 16))->v = r.v[2];
        if (len >= 32u) ((U64u *)(d + 24))->v = r.v[3];
        ((U64u *)(d + len - 8u))->v = r.last;
    } else if (len >= 4u) {
        ((U32u *)d)->v = (uint32_t)r.v[0];
        ((U32u *)(d + len - 4u))->v = (uint32_t)(r.v[0] >> (8u * (len - 4u)));
    } else {
        if (len & 2u) ((U16u *)d)->v = (uint16_t)r.v[0];
        if (len & 1u) d[len - 1u] = (uint8_t)(r.v[0] >> (8u * (len - 1u)));
    }
}

/* One lane moves n <= 32 bytes between two places that do not overlap and where 8 bytes may be READ past
 * either end of the source (an LDS stage with slack): the same chunks-plus-last-word scheme as LaneRun with the
 * loads placed where they are needed, so that the common lengths (4..16) pass through two predicated regions. */
__device__ __forceinline__ void lane_move32_slack(uint8_t *d, const uint8_t *s, uint32_t n)
{
    const uint64_t v0 = ld64u(s);
    if (n >= 8u) {
        const uint64_t vt = ld64u(s + n - 8u);
        ((U64u *)d)->v = v0;
        if (n > 16u) {
            ((U64u *)(d + 8))->v = ld64u(s + 8);
            if (n > 24u) ((U64u *)(d + 16))->v = ld64u(s + 16);
        }
        ((U64u *)(d + n - 8u))->v = vt;
    } else if (n >= 4u) {
        ((U32u *)d)->v = (uint32_t)v0;
        ((U32u *)(d + n - 4u))->v = (uint32_t)(v0 >> (8u * (n - 4u)));
    } else {
        if (n & 2u) ((U16u *)d)->v = (uint16_t)v0;
        if (n & 1u) d[n - 1u] = (uint8_t)(v0 >> (8u * (n - 1u)));
    }
}

/* Orders this wave's LDS accesses only (LDS executes a wave's accesses in order; this pins the
 * compiler) -- unlike wave_sync() it never waits for global stores to be acknowledged. */
__device__ __forceinline__ void lds_sync()
{
#ifndef K4_HOST_EMU
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
#else
    wave_sync();
#endif
}

/* Per-lane copy of len <= 32 bytes, regions must not overlap.  All (up to four) 8-byte loads are
 * issued before the first store, so a lane pays one memory round trip; the stores write exactly
 * len bytes.  `readable` = bytes that may be read starting at s (>= len). */
__device__ __forceinline__ void lane_copy32(uint8_t *d, const uint8_t *s, uint32_t len, uint32_t readable)
{
    LaneRun r;
    lane_run_load(r, s, len, readable);
    lane_run_store(d, r, len);
}

/*
 * A sliding window over a byte stream in global memory, kept in a per-wave LDS ring and refilled
 * with coalesced dword loads one 256-byte chunk ahead of use.  Used for the compressed stream in
 * the decoder and for the source bytes around the cursor in the encoder: lanes can then read any
 * byte position near the cursor at LDS latency instead of issuing 64 unaligned global requests.
 */
constexpr int RING_DWORDS = 256;                 /* 1 KiB of stream per wave */
struct StreamRing {
    uint32_t *ring;        /* LDS, RING_DWORDS dwords, slot = dword index & (RING_DWORDS-1) */
    const uint32_t *base;  /* dword-aligned address at or below the first stream byte */
    uint32_t a0;           /* misalignment of the stream start: 0..3 */
    uint32_t ndw;          /* dwords that contain stream bytes */
    uint32_t rlo, rhi;     /* dwords [rlo, rhi) are in the ring (rhi multiple of 64, rhi - rlo <= RING_DWORDS) */
    uint32_t pf;           /* lane l: dword rhi + l, loaded ahead of need */

    __device__ __forceinline__ uint32_t load(uint32_t dw) const { return dw < ndw ? base[dw] : 0u; }

    __device__ __forceinline__ void init(uint32_t *lds, const uint8_t *in, uint32_t len, int lane)
    {
        ring = lds;
        a0 = (uint32_t)((uintptr_t)in & 3u);
        base = (const uint32_t *)(in - a0);
        ndw = (a0 + len + 3u) >> 2;
        ring[lane] = load((uint32_t)lane);
        ring[64 + lane] = load(64u + (uint32_t)lane);
        rlo = 0u;
        rhi = 128u;
        pf = load(rhi + (uint32_t)lane);
        wave_sync();
    }

    /* make the ring cover [q, q + 96) (aligned byte positions) and run one chunk ahead */
    __device__ __forceinline__ void ensure(uint32_t q, int lane)
    {
        const uint32_t d = q >> 2;
        if (d + 24u > rhi + 64u) {               /* jumped past what is loaded or in flight */
            wave_sync();
            rhi = d & ~63u;
            rlo = rhi;
            ring[(rhi + (uint32_t)lane) & (RING_DWORDS - 1)] = load(rhi + (uint32_t)lane);
            ring[(rhi + 64u + (uint32_t)lane) & (RING_DWORDS - 1)] = load(rhi + 64u + (uint32_t)lane);
            rhi += 128u;
            pf = load(rhi + (uint32_t)lane);
            wave_sync();
            return;
        }
        while (rhi < d + 128u && rhi < ndw + 64u) {
            wave_sync();
            ring[(rhi + (uint32_t)lane) & (RING_DWORDS - 1)] = pf;
            rhi += 64u;
            if (rhi - rlo > (uint32_t)RING_DWORDS) rlo = rhi - (uint32_t)RING_DWORDS;
            pf = load(rhi + (uint32_t)lane);
            wave_sync();
        }
    }

    /* encoder use: keep [qlo, q + 512) readable where the stream has it; after a far jump the ring
     * restarts at qlo.  Aligned byte positions. */
    __device__ __forceinline__ void ensure_from(uint32_t qlo, uint32_t q, int lane)
    {
        const uint32_t d = q >> 2;
        if (d + 24u > rhi + 64u || (qlo >> 2) < rlo) ensure(qlo, lane);   /* restart at qlo (or no-op) */
        ensure(q, lane);
    }
    /* first / one-past-last aligned byte position readable */
    __device__ __forceinline__ uint32_t lo_byte() const { return rlo << 2; }
    __device__ __forceinline__ uint32_t hi_byte() const { return rhi << 2; }

    /* 4 stream bytes at aligned byte position q (per lane) */
    __device__ __forceinline__ uint32_t read4(uint32_t q) const
    {
        const uint32_t d = q >> 2;
        const uint32_t lo = ring[d & (RING_DWORDS - 1)];
        const uint32_t hi = ring[(d + 1u) & (RING_DWORDS - 1)];
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> ((q & 3u) * 8u));
    }

    /* scalar parser: the 4 stream bytes at wave-uniform stream position p */
    __device__ __forceinline__ uint32_t fetch(uint32_t p, int lane)
    {
        ensure(p + a0, lane);
        return uni(read4(p + a0));
    }
};

}  // namespace k4
