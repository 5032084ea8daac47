/*
 * tests/emu/hip/hip_runtime.h -- HOST-SIDE WAVE EMULATOR (test infrastructure only).
 *
 * This is NOT a compatibility layer of the product.  The product's kernels
 * (k4os/compression/lz4_amd/csrc/ *.hpp) are written as native HIP for gfx950 and are only
 * ever shipped compiled by hipcc.  Because the build container has no GPU, the test suite
 * additionally compiles the *same kernel source* with g++ against this fake <hip/hip_runtime.h>
 * so that the kernels' control logic (token parsing, speculative probe batches, cross-lane
 * ballots/shuffles, LDS hand-offs) can be exercised on the CPU and compared with the oracle
 * before any GPU minute is spent.  Every lane of a workgroup is a cooperative fiber on one OS
 * thread; a wave-level collective (ballot / shfl / readlane / wave barrier) or __syncthreads()
 * is a rendezvous of the participating fibers.  Between rendezvous points lanes run one after
 * another, so cross-lane communication through LDS/global memory must be separated by a
 * wave barrier in the kernel source -- which is also what the hardware memory model requires.
 *
 * Nothing under k4os/ includes or links this file.
 */
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define K4_HOST_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __constant__ static const

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(uint32_t a, uint32_t b) { return uint2{a, b}; }
typedef struct ihipStream_t *hipStream_t;

namespace k4emu {
enum { WAVE = 64 };
struct Lane;
struct LaneIds { dim3 tid, bid, bdim, gdim; };
LaneIds &ids();
/* rendezvous of the 64 lanes of the calling lane's wave; each posts `in`, gets all 64 back */
const uint64_t *wave_exchange(uint64_t in);
void block_barrier();
int lane_id();
/* run `fn(arg)` for every thread of a grid; blocks are spread over `threads` OS threads */
void launch(dim3 grid, dim3 block, void (*fn)(void *), void *arg, int threads);
template <class F> static void launch_fn(dim3 grid, dim3 block, F f, int threads = 0) {
    launch(grid, block, [](void *p) { (*(F *)p)(); }, &f, threads);
}
}  // namespace k4emu

#define threadIdx (k4emu::ids().tid)
#define blockIdx (k4emu::ids().bid)
#define blockDim (k4emu::ids().bdim)
#define gridDim (k4emu::ids().gdim)
#define warpSize 64

static inline void __syncthreads() { k4emu::block_barrier(); }

static inline unsigned long long __ballot(int pred) {
    const uint64_t *v = k4emu::wave_exchange(pred ? 1 : 0);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) m |= (unsigned long long)(v[i] & 1) << i;
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) { return __ballot(pred) == ~0ull; }

template <class T> static inline T k4emu_shfl_idx(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl of <= 8 bytes");
    uint64_t in = 0; memcpy(&in, &v, sizeof(T));
    const uint64_t *a = k4emu::wave_exchange(in);
    T out; memcpy(&out, &a[src & 63], sizeof(T));
    return out;
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int l = k4emu::lane_id();
    int s = (l & ~(width - 1)) | (src & (width - 1));
    return k4emu_shfl_idx(v, s);
}
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int l = k4emu::lane_id();
    int s = l - (int)d;
    if (s < (l & ~(width - 1))) s = l;
    return k4emu_shfl_idx(v, s);
}
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = k4emu::lane_id();
    int s = l + (int)d;
    if (s > (l | (width - 1))) s = l;
    return k4emu_shfl_idx(v, s);
}
template <class T> static inline T __shfl_xor(T v, int m, int width = 64) {
    int l = k4emu::lane_id();
    (void)width;
    return k4emu_shfl_idx(v, l ^ m);
}

static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }

static inline int __builtin_amdgcn_readlane(int v, int lane) { return (int)k4emu_shfl_idx((uint32_t)v, lane); }   /* int, as the real builtin: widening sign-extends */
static inline int __builtin_amdgcn_ds_bpermute(int addr, int data) { return (int)k4emu_shfl_idx((uint32_t)data, (addr >> 2) & 63); }
static inline uint32_t __builtin_amdgcn_readfirstlane(uint32_t v) { return k4emu_shfl_idx(v, 0); }
static inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t mask, uint32_t add) {
    int l = k4emu::lane_id();
    uint32_t below = l >= 32 ? 0xffffffffu : ((1u << l) - 1u);
    return add + (uint32_t)__builtin_popcount(mask & below);
}
static inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t mask, uint32_t add) {
    int l = k4emu::lane_id();
    uint32_t below = l <= 32 ? 0u : ((1u << (l - 32)) - 1u);
    return add + (uint32_t)__builtin_popcount(mask & below);
}
/* v_mov_b32 with a DPP control word: the subset the kernels use (row_shr:1..15, row_bcast:15/31) */
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    const uint64_t *v = k4emu::wave_exchange((uint64_t)(uint32_t)src);
    const int l = k4emu::lane_id();
    const int row = l >> 4, bank = (l & 15) >> 2;
    if (!((row_mask >> row) & 1) || !((bank_mask >> bank) & 1)) return old;
    int srcl = -1;
    if (ctrl >= 0x111 && ctrl <= 0x11f) {
        const int n = ctrl - 0x110;
        if ((l & 15) >= n) srcl = l - n;
    } else if (ctrl == 0x138) {          /* wave_shr:1 */
        if (l >= 1) srcl = l - 1;
    } else if (ctrl == 0x142) {
        if (row >= 1) srcl = row * 16 - 1;
    } else if (ctrl == 0x143) {
        if (row >= 2) srcl = 31;
    } else {
        __builtin_trap();
    }
    if (srcl < 0) return bound_ctrl ? 0 : old;
    return (int)(uint32_t)v[srcl];
}
static inline void __builtin_amdgcn_wave_barrier() { (void)k4emu::wave_exchange(0); }
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_readcyclecounter() 0ull
#define __builtin_amdgcn_s_memrealtime() 0ull
#define __builtin_amdgcn_s_getreg(x) 0u
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_s_setprio(n) ((void)0)
#define __builtin_amdgcn_sched_barrier(n) ((void)0)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))

static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAnd(unsigned *p, unsigned v) { return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicCAS(unsigned *p, unsigned cmp, unsigned v) { __atomic_compare_exchange_n(p, &cmp, v, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMax(unsigned *p, unsigned v) {
    unsigned o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
static inline unsigned atomicMin(unsigned *p, unsigned v) {
    unsigned o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o > v && !__atomic_compare_exchange_n(p, &o, v, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
template <class T> static inline T min(T a, T b) { return a < b ? a : b; }
template <class T> static inline T max(T a, T b) { return a > b ? a : b; }
