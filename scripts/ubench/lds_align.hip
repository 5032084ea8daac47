// Micro-benchmark: cost of per-lane LDS accesses at aligned / misaligned byte addresses (gfx950).
// hipcc --offload-arch=gfx950 -O3 lds_align.hip -o lds_align && ./lds_align
// One wave per workgroup; each iteration is a dependent read (the next address comes from the data read), so the
// number is latency + throughput of one access as a decoder lane sees it.  Also: the same for stores followed by a read.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };
struct __attribute__((packed, aligned(1))) U64u { uint64_t v; };

template <int BYTES, int MIS, int ACTIVE>
__global__ void k(uint32_t iters, unsigned long long* out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192 + 64];
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t i = lane; i < (8192 + 64) / 4; i += 64) ((uint32_t*)lds)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a = lane * BYTES * 3u;                 // spread over the banks
    uint64_t acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (lane < ACTIVE) {
        for (uint32_t i = 0; i < iters; i++) {
            const uint32_t addr = ((a & 4095u) & ~(uint32_t)(BYTES - 1)) + MIS;
            uint64_t x;
            if (BYTES == 4) x = ((const U32u*)(lds + addr))->v; else x = ((const U64u*)(lds + addr))->v;
            acc += x;
            ((U64u*)(lds + 4096 + ((addr * 5u) & 4088u) + MIS))->v = acc;   // a store per iteration as well
            a += (uint32_t)x & 0xff8u;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = acc; }
}

template <int BYTES, int MIS, int ACTIVE>
static void run(const char* name)
{
    unsigned long long* d; hipMalloc(&d, 16 * 1024);
    const uint32_t iters = 20000;
    hipLaunchKernelGGL((k<BYTES, MIS, ACTIVE>), dim3(256), dim3(64), 0, 0, iters, d);
    hipDeviceSynchronize();
    unsigned long long h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; i++) s += (double)h[2 * i];
    printf("%-44s %7.1f cycles (counter ticks) per read+store\n", name, s / 256 / iters);
    hipFree(d);
}
int main()
{
    run<4, 0, 64>("b32 aligned, 64 lanes");
    run<4, 1, 64>("b32 misaligned by 1, 64 lanes");
    run<8, 0, 64>("b64 aligned, 64 lanes");
    run<8, 1, 64>("b64 misaligned by 1, 64 lanes");
    run<8, 4, 64>("b64 misaligned by 4, 64 lanes");
    run<8, 0, 8>("b64 aligned, 8 lanes");
    run<8, 1, 8>("b64 misaligned by 1, 8 lanes");
    run<8, 4, 8>("b64 misaligned by 4, 8 lanes");
    return 0;
}
