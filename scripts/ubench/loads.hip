// Micro-benchmark: cost of one wave-wide 16-byte-per-lane load in the access patterns the encoder uses.
// hipcc --offload-arch=gfx950 -O3 loads.hip -o loads && ./loads
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
struct __attribute__((packed, aligned(1))) U128u { uint32_t v[4]; };
struct __attribute__((aligned(4))) U128a4 { uint32_t v[4]; };

template <int MODE>
__global__ void k(const uint8_t* src, uint32_t n, uint32_t iters, unsigned long long* out)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t pos = (blockIdx.x * 7919u + 13u) % (n - 4096);
    uint32_t acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (uint32_t i = 0; i < iters; i++) {
        uint32_t p;
        if (MODE < 3) p = pos + lane;                                        // consecutive byte positions (probe side)
        else p = (pos * 2654435761u + lane * 40503u) % (n - 64);             // scattered (candidate side)
        uint32_t x;
        if (MODE == 0 || MODE == 3) { U128u v = *(const U128u*)(src + p); x = v.v[0] ^ v.v[1] ^ v.v[2] ^ v.v[3]; }          // unaligned dwordx4
        else if (MODE == 1 || MODE == 4) { const uint8_t* q = src + (p & ~3u); U128a4 v = *(const U128a4*)q; uint32_t e = *(const uint32_t*)(q + 16);
                                           x = v.v[0] ^ v.v[1] ^ v.v[2] ^ v.v[3] ^ e; }                                       // dword-aligned x4 + x1
        else { x = *(const uint32_t*)(src + ((pos & ~3u) + 4 * lane)); }                                                       // coalesced aligned dword per lane
        acc += x;
        pos = (pos + 9 + (x & 7)) % (n - 4096);                               // dependent chain
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = acc; }
}

int main()
{
    const uint32_t n = 64u << 20, iters = 2000;
    uint8_t* d; unsigned long long* o;
    hipMalloc(&d, n); hipMemset(d, 7, n); hipMalloc(&o, 4096 * 16);
    const char* names[] = {"consecutive, unaligned dwordx4", "consecutive, dword-aligned x4+x1", "coalesced aligned dword/lane",
                           "scattered, unaligned dwordx4", "scattered, dword-aligned x4+x1"};
    for (int waves = 1; waves <= 2048; waves *= 2048) {
        for (int m = 0; m < 5; m++) {
            for (int rep = 0; rep < 2; rep++) {
                switch (m) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, d, n, iters, o); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, d, n, iters, o); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, d, n, iters, o); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(waves), dim3(64), 0, 0, d, n, iters, o); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(waves), dim3(64), 0, 0, d, n, iters, o); break;
                }
                hipDeviceSynchronize();
            }
            unsigned long long h[2]; hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
            printf("waves %4d  %-36s %7.1f cycles per dependent load step\n", waves, names[m], (double)h[0] / iters);
        }
    }
    return 0;
}
