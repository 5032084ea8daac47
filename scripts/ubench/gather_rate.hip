// Micro-benchmark: the RATE at which a CU takes wave-wide loads (32 waves per CU, eight independent loads in flight per wave), by
// bytes per lane (4 / 8 / 16), by address pattern (consecutive byte positions; lanes in lines of their own, aligned or at any byte)
// and by the size of the window the addresses fall into (64 KiB per workgroup: L1 / L2 hits; 256 MiB: misses).
// Prints CU cycles per wave-load.  hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate && ./gather_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };
struct __attribute__((packed, aligned(1))) U64u { uint64_t v; };
struct __attribute__((packed, aligned(1))) U128u { uint32_t v[4]; };

template <int BYTES, int PATTERN>       // PATTERN 0: consecutive bytes (lane l at base + l), 1: scattered aligned to BYTES, 2: scattered at any byte,
                                        // 3: as 2 with 16 of the 64 lanes active, 4: as 2 with the lanes' addresses within 256 bytes of each other
__global__ __launch_bounds__(256) void k(const uint8_t *src, uint32_t window, uint32_t iters, uint32_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wg_base = (uint32_t)(((unsigned long long)blockIdx.x * 2654435761ull) % (256u << 20)) & ~0xffffu;   // this workgroup's window inside 256 MiB + slack
    uint32_t state = blockIdx.x * 977u + threadIdx.x * 31u + 7u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; i++) {
        uint32_t a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            state = state * 1664525u + 1013904223u;
            uint32_t off;
            if (PATTERN == 0) off = ((state >> 8) % (window - 1024u)) + lane;                    // wave-uniform base is not enforced: per lane nearly the same -> use lane-independent part
            else off = (state >> 8) % (window - 64u);
            if (PATTERN == 0) { uint32_t s2 = __builtin_amdgcn_readfirstlane(state); off = ((s2 >> 8) % (window - 1024u)) + lane; }
            if (PATTERN == 1) off &= ~(uint32_t)(BYTES - 1);
            if (PATTERN == 4) { uint32_t s2 = __builtin_amdgcn_readfirstlane(state); off = ((s2 >> 8) % (window - 1024u)) + ((state >> 12) & 255u); }
            a[u] = wg_base + off;
        }
        if (PATTERN != 3 || lane < 16u)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (BYTES == 4) acc ^= ((const U32u *)(src + a[u]))->v;
            else if (BYTES == 8) { const uint64_t v = ((const U64u *)(src + a[u]))->v; acc ^= (uint32_t)v ^ (uint32_t)(v >> 32); }
            else { const U128u v = *(const U128u *)(src + a[u]); acc ^= v.v[0] ^ v.v[1] ^ v.v[2] ^ v.v[3]; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int BYTES, int PATTERN>
static void run(const uint8_t *d, uint32_t *o, uint32_t window, const char *name, int cus)
{
    const uint32_t iters = 200;
    const int wgs = cus * 8;                       // 8 workgroups of 4 waves per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BYTES, PATTERN>), dim3(wgs), dim3(256), 0, 0, d, window, 20u, o);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<BYTES, PATTERN>), dim3(wgs), dim3(256), 0, 0, d, window, iters, o);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double loads_per_cu = (double)wgs * 4 * iters * 8 / cus;
    printf("%2d B/lane %-28s window %9u B : %7.1f CU cycles per wave-load (at 2.4 GHz), %6.1f GB/s per CU\n", BYTES, name, window,
           ms * 1e-3 * 2.4e9 / loads_per_cu, loads_per_cu * 64 * BYTES / (ms * 1e-3) / 1e9);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, (size_t)(256u << 20) + (1u << 20) + (256u << 20)); hipMemset(d, 7, (size_t)(256u << 20) + (1u << 20) + (256u << 20)); hipMalloc(&o, 64);
    printf("%d CUs\n", cus);
    for (uint32_t window : {65536u, 256u << 20}) {
        run<4, 0>(d, o, window, "consecutive bytes", cus); run<8, 0>(d, o, window, "consecutive bytes", cus); run<16, 0>(d, o, window, "consecutive bytes", cus);
        run<4, 1>(d, o, window, "scattered, aligned", cus); run<8, 1>(d, o, window, "scattered, aligned", cus); run<16, 1>(d, o, window, "scattered, aligned", cus);
        run<4, 2>(d, o, window, "scattered, any byte", cus); run<8, 2>(d, o, window, "scattered, any byte", cus); run<16, 2>(d, o, window, "scattered, any byte", cus);
        run<8, 3>(d, o, window, "scattered, 16 lanes active", cus); run<16, 3>(d, o, window, "scattered, 16 lanes active", cus);
        run<8, 4>(d, o, window, "lanes within 256 bytes", cus); run<16, 4>(d, o, window, "lanes within 256 bytes", cus);
    }
    return 0;
}
