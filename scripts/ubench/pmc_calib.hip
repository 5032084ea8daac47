// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on kernels whose HBM byte counts are known exactly, in the access
// patterns of the LZ4 kernels (MI355X_MICROARCH.md: only the wide coalesced read is calibrated there -- x2 -- "calibrate on a
// known byte count in your own access pattern").  Buffers are larger than the 256 MiB Infinity Cache and touched once.
//   calib_stream_read16   every lane reads 16 B, coalesced (1 GiB)            -> FETCH_SIZE should report 1 GiB
//   calib_scatter_read16  every lane reads 16 B at its own 64 B-aligned line    -> 64 B lines fetched for 16 B used: 4x the useful bytes
//   calib_stream_write16  every lane writes 16 B, coalesced (1 GiB)
//   calib_scatter_write2  every lane writes 2 B into its own 64 B line (the global hash table's access)
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/pmc_calib.hip -o scripts/ubench/pmc_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#pragma clang diagnostic ignored "-Wunused-result"

constexpr size_t N16 = (size_t)1 << 26;   // 64 Mi x 16 B = 1 GiB

__global__ void calib_stream_read16(const uint4 *src, uint4 *sink)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 v = src[i];
    if (v.x == 0x12345678u && v.y == 0x9abcdef0u) sink[0] = v;   // never true: keeps the load
}
__global__ void calib_scatter_read16(const uint4 *src, uint4 *sink)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // i < N16 / 4: one 16-byte piece per 64-byte line
    const size_t line = (i * 2654435761ull) & (N16 / 4 - 1);            // a permutation of the lines (odd multiplier)
    const uint4 v = src[line * 4 + (i & 3)];
    if (v.x == 0x12345678u && v.y == 0x9abcdef0u) sink[0] = v;
}
__global__ void calib_stream_write16(uint4 *dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    dst[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ void calib_scatter_write2(uint16_t *dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // i < N16 / 4
    const size_t line = (i * 2654435761ull) & (N16 / 4 - 1);
    dst[line * 32 + (i & 31)] = (uint16_t)i;
}

int main()
{
    uint4 *a, *sink;
    hipMalloc(&a, N16 * 16);
    hipMalloc(&sink, 64);
    hipMemset(a, 1, N16 * 16);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        calib_stream_read16<<<N16 / 256, 256>>>(a, sink);
        calib_scatter_read16<<<N16 / 4 / 256, 256>>>(a, sink);
        calib_stream_write16<<<N16 / 256, 256>>>(a);
        calib_scatter_write2<<<N16 / 4 / 256, 256>>>((uint16_t *)a);
        hipDeviceSynchronize();
    }
    printf("known bytes: stream_read16 %zu, scatter_read16 useful %zu / lines %zu, stream_write16 %zu, scatter_write2 useful %zu / lines %zu\n",
           N16 * 16, N16 / 4 * 16, N16 / 4 * 64, N16 * 16, N16 / 4 * 2, N16 / 4 * 64);
    return 0;
}
