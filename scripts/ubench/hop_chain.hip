// Micro-benchmark: what one hop of the decoder's token chain costs (gfx950).
// hipcc --offload-arch=gfx950 -O3 hop_chain.hip -o hop_chain && ./hop_chain
// A: the shipped form (k4lz4_decode.hpp, follow_tokens): s_bitset1 + s_nop 2 + v_readlane, the lane select of one
//    v_readlane being the result of the one before.
// B: the 64 link bytes moved into 16 SGPRs first (3 DPP moves + 3 shift-ors + 16 v_readlane), then the chain in the
//    scalar unit alone: s_lshr / s_mov m0 / s_movrels / s_lshr / s_and per hop.
// Both with W waves per SIMD running the same thing (W = 1, 4, 8), cycles of s_memtime per hop.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define HOPS 24

__device__ __forceinline__ unsigned long long memtime() { return __builtin_readcyclecounter(); }

__global__ void hop_readlane(uint32_t iters, unsigned long long *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t word = (lane + 1u + (lane * 7u) % 3u) & 63u;          // a chain that never ends
    unsigned long long T = 0, acc = 0;
    const unsigned long long t0 = memtime();
    for (uint32_t i = 0; i < iters; i++) {
        uint32_t pk = i & 63u;
        T = 0;
#define K4_HOP "s_bitset1_b64 %[T], %[pk]\n\ts_nop 2\n\tv_readlane_b32 %[pk], %[word], %[pk]\n\t"
#define K4_HOP8 K4_HOP K4_HOP K4_HOP K4_HOP K4_HOP K4_HOP K4_HOP K4_HOP
        asm volatile(K4_HOP8 K4_HOP8 K4_HOP8 : [T] "+s"(T), [pk] "+s"(pk) : [word] "v"(word) : "scc");
        acc += T + pk;
        word ^= (uint32_t)(acc & 0u);
    }
    const unsigned long long t1 = memtime();
    if (lane == 0) { out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = t1 - t0; out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + 1] = acc; }
}

template <bool CHAIN>
__global__ void hop_sgpr(uint32_t iters, unsigned long long *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t word = (lane + 1u + (lane * 7u) % 3u) & 63u;
    unsigned long long T = 0, acc = 0;
    const unsigned long long t0 = memtime();
    for (uint32_t i = 0; i < iters; i++) {
        uint32_t pk = i & 63u;
        T = 0;
        // bytes of four neighbouring lanes into the first lane of each quad
        const uint32_t b1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)word, 0x55, 0xf, 0xf, true);   // quad_perm [1,1,1,1]
        const uint32_t b2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)word, 0xaa, 0xf, 0xf, true);   // [2,2,2,2]
        const uint32_t b3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)word, 0xff, 0xf, 0xf, true);   // [3,3,3,3]
        const uint32_t quad = (word & 0xffu) | ((b1 & 0xffu) << 8) | ((b2 & 0xffu) << 16) | (b3 << 24);
        asm volatile(
            "v_readlane_b32 s64, %[q], 0\n\tv_readlane_b32 s65, %[q], 4\n\tv_readlane_b32 s66, %[q], 8\n\tv_readlane_b32 s67, %[q], 12\n\t"
            "v_readlane_b32 s68, %[q], 16\n\tv_readlane_b32 s69, %[q], 20\n\tv_readlane_b32 s70, %[q], 24\n\tv_readlane_b32 s71, %[q], 28\n\t"
            "v_readlane_b32 s72, %[q], 32\n\tv_readlane_b32 s73, %[q], 36\n\tv_readlane_b32 s74, %[q], 40\n\tv_readlane_b32 s75, %[q], 44\n\t"
            "v_readlane_b32 s76, %[q], 48\n\tv_readlane_b32 s77, %[q], 52\n\tv_readlane_b32 s78, %[q], 56\n\tv_readlane_b32 s79, %[q], 60\n\t"
            "s_nop 3\n\t"
#define S_HOP "s_bitset1_b64 %[T], %[pk]\n\ts_lshr_b32 s80, %[pk], 2\n\ts_lshl_b32 s81, %[pk], 3\n\ts_mov_b32 m0, s80\n\ts_and_b32 s81, s81, 24\n\ts_movrels_b32 s82, s64\n\ts_lshr_b32 s82, s82, s81\n\ts_and_b32 %[pk], s82, 63\n\t"
#define S_HOP8 S_HOP S_HOP S_HOP S_HOP S_HOP S_HOP S_HOP S_HOP
            : [T] "+s"(T), [pk] "+s"(pk)
            : [q] "v"(quad)
            : "scc", "m0", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82");
        if (CHAIN)
            asm volatile(S_HOP8 S_HOP8 S_HOP8
                : [T] "+s"(T), [pk] "+s"(pk)
                :
                : "scc", "m0", "s80", "s81", "s82");
        acc += T + pk;
        word ^= (uint32_t)(acc & 0u);
    }
    const unsigned long long t1 = memtime();
    if (lane == 0) { out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = t1 - t0; out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + 1] = acc; }
}

// C: two sequences per hop: every lane also learns where its successor points (one ds_bpermute), the chain follows those
//    double links and marks the lane in between from a second field of the word just read.
template <bool CHAIN>
__global__ void hop_double(uint32_t iters, unsigned long long *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t word = (lane + 1u + (lane * 7u) % 3u) & 63u;
    unsigned long long T = 0, acc = 0;
    const unsigned long long t0 = memtime();
    for (uint32_t i = 0; i < iters; i++) {
        uint32_t pk = i & 63u, t = 0;
        T = 0;
        const uint32_t n1 = word & 63u;
        const uint32_t w1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(n1 << 2), (int)word);
        const uint32_t word2 = (word & ~63u) | (w1 & 63u) | (n1 << 24);
#define D_HOP "s_bitset1_b64 %[T], %[pk]\n\ts_lshr_b32 %[t], %[pk], 24\n\ts_bitset1_b64 %[T], %[t]\n\ts_nop 0\n\tv_readlane_b32 %[pk], %[word], %[pk]\n\t"
#define D_HOP4 D_HOP D_HOP D_HOP D_HOP
        if (CHAIN)
            asm volatile(D_HOP4 D_HOP4 D_HOP4 : [T] "+s"(T), [pk] "+s"(pk), [t] "+s"(t) : [word] "v"(word2) : "scc");
        else
            asm volatile("v_readlane_b32 %[pk], %[word], %[pk]" : [pk] "+s"(pk) : [word] "v"(word2));
        acc += T + pk;
        word ^= (uint32_t)(acc & 0u);
    }
    const unsigned long long t1 = memtime();
    if (lane == 0) { out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2] = t1 - t0; out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + 1] = acc; }
}

template <typename K>
static double run(K kern, int waves_per_simd)
{
    unsigned long long *d; hipMalloc(&d, 1 << 20);
    const uint32_t iters = 4000;
    const int wg = 256 * waves_per_simd;            // 256 CUs x 4 SIMDs: one workgroup of 4 waves per CU and "wave per SIMD"
    hipLaunchKernelGGL(kern, dim3(wg), dim3(256), 0, 0, iters, d);
    hipDeviceSynchronize();
    static unsigned long long h[1 << 16]; hipMemcpy(h, d, (size_t)wg * 4 * 16, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < wg * 4; i++) s += (double)h[2 * i];
    hipFree(d);
    return s / (wg * 4) / iters;
}
int main()
{
    for (int w : {1, 4, 8}) {
        const double a = run(hop_readlane, w), b0 = run(hop_sgpr<false>, w), b1 = run(hop_sgpr<true>, w);
        const double c0 = run(hop_double<false>, w), c1 = run(hop_double<true>, w);
        printf("%d waves/SIMD: double links: set-up + one read %6.1f, with 12 double hops %6.1f (the same %d sequences)\n", w, c0, c1, HOPS);
        printf("%d waves/SIMD: readlane chain %6.1f ticks per %d hops = %5.1f per hop | SGPR table: set-up %6.1f, with chain %6.1f = %5.1f per hop\n",
               w, a, HOPS, a / HOPS, b0, b1, (b1 - b0) / HOPS);
    }
    return 0;
}
