#include "hip/hip_runtime.h"
#include <stdio.h>
__global__ void k(uint32_t *out) {
    __shared__ uint32_t tab[64];
    int lane = threadIdx.x & 63;
    tab[lane] = lane * 3;
    __builtin_amdgcn_wave_barrier();
    uint32_t v = tab[63 - lane];
    unsigned long long m = __ballot(v & 1);
    uint32_t s = __shfl_up(v, 1);
    uint32_t r = __builtin_amdgcn_readlane(v, 5);
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v + (uint32_t)__popcll(m) + s + r;
}
int main() {
    static uint32_t out[4 * 128];
    k4emu::launch_fn(dim3(4), dim3(128), [&] { k(out); });
    unsigned sum = 0;
    for (unsigned i = 0; i < 512; i++) sum += out[i];
    printf("sum=%u out[1]=%u\n", sum, out[1]);
    return 0;
}
