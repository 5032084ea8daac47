// Micro-benchmark of parse_window (k4lz4_decode_parse.hpp) alone: one wave per block walks the whole compressed
// stream window by window (no copying), returns the number of tokens found and cycles spent per phase.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I k4os/compression/lz4_amd/csrc scripts/ubench/parse_only.hip -o scripts/ubench/parse_only
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-result"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "k4lz4_decode_parse.hpp"

template <bool PROF>
__global__ __launch_bounds__(64) void parse_only(const uint8_t *src, const uint64_t *off, const int32_t *len, uint32_t *ntok, unsigned long long *cyc, int n)
{
    __shared__ uint32_t lds[k4::PARSE_LDS_DWORDS];
    const int lane = k4::lane_id();
    const int b = blockIdx.x;
    if (b >= n) return;
    const uint8_t *in = src + off[b];
    const uint32_t iend = (uint32_t)len[b];
    k4::ParseWin win;
    win.init(lds, in, iend, lane);
    uint32_t ip = 0, total = 0, rounds = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (ip + 16 < iend) {
        uint32_t end_ip;
        const uint32_t k = k4::parse_window<PROF>(win, ip, iend - 16, lane, lds, end_ip, cyc + 16 * b + 1);
        rounds++;
        if (k == 0) break;
        total += k;
        ip = end_ip;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { ntok[2 * b] = total; ntok[2 * b + 1] = rounds; cyc[16 * b] = t1 - t0; }
}

int main(int argc, char **argv)
{
    // input file: packed blocks produced by scripts/ubench/parse_only_input.py: [n:u32][len:u32 * n][bytes...]
    FILE *f = fopen(argc > 1 ? argv[1] : "/tmp/parse_only.bin", "rb");
    if (!f) { printf("no input\n"); return 1; }
    uint32_t n; if (fread(&n, 4, 1, f) != 1) return 1;
    std::vector<int32_t> len(n); if (fread(len.data(), 4, n, f) != n) return 1;
    std::vector<uint64_t> off(n); uint64_t tot = 0; for (uint32_t i = 0; i < n; i++) { off[i] = tot; tot += (len[i] + 15) & ~15; }
    std::vector<uint8_t> buf(tot + 64);
    for (uint32_t i = 0; i < n; i++) if (fread(buf.data() + off[i], 1, len[i], f) != (size_t)len[i]) return 1;
    int reps = argc > 2 ? atoi(argv[2]) : 1;   // tile the blocks `reps` times to fill the chip
    uint32_t N = n * reps;
    std::vector<uint64_t> offN(N); std::vector<int32_t> lenN(N);
    for (uint32_t i = 0; i < N; i++) { offN[i] = off[i % n]; lenN[i] = len[i % n]; }
    uint8_t *d_src; uint64_t *d_off; int32_t *d_len; uint32_t *d_tok; unsigned long long *d_cyc;
    hipMalloc(&d_src, buf.size()); hipMalloc(&d_off, N * 8); hipMalloc(&d_len, N * 4); hipMalloc(&d_tok, N * 8); hipMalloc(&d_cyc, N * 128); hipMemset(d_cyc, 0, N * 128);
    hipMemcpy(d_src, buf.data(), buf.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_off, offN.data(), N * 8, hipMemcpyHostToDevice); hipMemcpy(d_len, lenN.data(), N * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; it++) {
        hipEventRecord(e0);
        if (it == 2) hipMemset(d_cyc, 0, N * 128);
        if (it < 2) parse_only<false><<<N, 64>>>(d_src, d_off, d_len, d_tok, d_cyc, (int)N);
        else parse_only<true><<<N, 64>>>(d_src, d_off, d_len, d_tok, d_cyc, (int)N);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %u blocks, %.3f ms\n", it, N, ms);
    }
    std::vector<uint32_t> tok(2 * N); std::vector<unsigned long long> cyc(16 * N);
    hipMemcpy(tok.data(), d_tok, N * 8, hipMemcpyDeviceToHost); hipMemcpy(cyc.data(), d_cyc, N * 128, hipMemcpyDeviceToHost);
    for (uint32_t i = 0; i < n; i++) {
        const unsigned long long *c = &cyc[16 * i];
        printf("block %u: C=%d tokens=%u windows=%u cycles=%llu (instrumented) cycles/token=%.1f | cover %llu main %llu ext %llu walk %llu list %llu | trips main %llu ext %llu hops %llu serial %llu list %llu\n",
               i, len[i], tok[2 * i], tok[2 * i + 1], c[0], (double)c[0] / (tok[2 * i] ? tok[2 * i] : 1), c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], c[10]);
    }
    return 0;
}
