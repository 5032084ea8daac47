/*
 * k4lz4_xxh32.hpp -- batched xxHash32 for gfx950 (frame layer).
 *
 * The reference takes XXH32 from NuGet K4os.Hash.xxHash 1.0.8 (Streams/K4os.Compression.LZ4.Streams.csproj:15;
 * call sites Streams/Frames/LZ4FrameWriter.cs:100 (header byte), :162-182 (block and content checksums),
 * Streams/Internal/Stash.cs:149-150), always with seed 0.  The algorithm is the published xxHash32: four
 * 32-bit accumulators over 16-byte stripes, each accumulator a serial multiply-rotate chain.
 *
 * One buffer cannot be split (the chains are serial), so a buffer is owned by FOUR lanes -- one per
 * accumulator, together they read 16 contiguous bytes per step -- and a wavefront hashes 16 buffers
 * at a time.  Block checksums of a frame (thousands of 64 KiB..4 MiB blocks) fill the chip; a single
 * content checksum over one long stream runs at the speed of its one chain (about 1 GB/s).
 */
#pragma once
#include "k4lz4_common.hpp"

namespace k4 {

constexpr uint32_t XXH_P1 = 2654435761u, XXH_P2 = 2246822519u, XXH_P3 = 3266489917u, XXH_P4 = 668265263u,
                   XXH_P5 = 374761393u;

struct HashArgs {
    const uint8_t *data;
    const uint64_t *off;
    const uint64_t *len;
    uint32_t *out;
    long long n;
    uint32_t seed;
};

__device__ __forceinline__ uint32_t xxh_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ uint32_t xxh_round(uint32_t acc, uint32_t x) { return xxh_rotl(acc + x * XXH_P2, 13) * XXH_P1; }

constexpr int XXH_THREADS = 256;                 /* 64 buffers per workgroup */

__global__ __launch_bounds__(XXH_THREADS) void k4_xxh32_kernel(HashArgs a)
{
    const long long t = (long long)blockIdx.x * XXH_THREADS + threadIdx.x;
    const long long g = t >> 2;                  /* buffer */
    const int c = (int)(t & 3);                  /* accumulator */
    const int lane = lane_id();
    const bool live = g < a.n;
    const uint8_t *p = live ? a.data + a.off[g] : a.data;
    const uint64_t len = live ? a.len[g] : 0ull;
    const uint64_t stripes = len >> 4;
    uint32_t v = a.seed + (c == 0 ? XXH_P1 + XXH_P2 : c == 1 ? XXH_P2 : c == 2 ? 0u : 0u - XXH_P1);
    const uint8_t *q = p + 4 * c;
    uint64_t s = 0;
    if (stripes >= 16) {
        /* long buffers: eight loads per chain always in flight while the previous eight are folded in */
        uint32_t x[8], y[8];
#pragma unroll
        for (int k = 0; k < 8; k++) x[k] = ld32u(q + 16 * k);
        q += 128;
        for (s = 8; s + 8 <= stripes; s += 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) y[k] = ld32u(q + 16 * k);
            q += 128;
#pragma unroll
            for (int k = 0; k < 8; k++) v = xxh_round(v, x[k]);
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = y[k];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) v = xxh_round(v, x[k]);
    }
    for (; s < stripes; s++) {
        v = xxh_round(v, ld32u(q));
        q += 16;
    }
    const int base = lane & ~3;
    const uint32_t v1 = (uint32_t)__shfl((int)v, base), v2 = (uint32_t)__shfl((int)v, base + 1),
                   v3 = (uint32_t)__shfl((int)v, base + 2), v4 = (uint32_t)__shfl((int)v, base + 3);
    if (live && c == 0) {
        uint32_t h = stripes ? xxh_rotl(v1, 1) + xxh_rotl(v2, 7) + xxh_rotl(v3, 12) + xxh_rotl(v4, 18) : a.seed + XXH_P5;
        h += (uint32_t)len;
        const uint8_t *r = p + (stripes << 4), *end = p + len;
        while (r + 4 <= end) { h = xxh_rotl(h + ld32u(r) * XXH_P3, 17) * XXH_P4; r += 4; }
        while (r < end) { h = xxh_rotl(h + (uint32_t)(*r) * XXH_P5, 11) * XXH_P1; r++; }
        h ^= h >> 15; h *= XXH_P2; h ^= h >> 13; h *= XXH_P3; h ^= h >> 16;
        a.out[g] = h;
    }
}

}  // namespace k4
