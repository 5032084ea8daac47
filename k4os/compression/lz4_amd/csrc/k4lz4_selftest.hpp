/*
 * k4lz4_selftest.hpp -- the hand-written scalar chains against their C twins, on the device.
 *
 * follow_tokens (decoder), hop_chain and hop_chain_pairs (fast encoder) are inline ISA; the wave emulator runs their
 * C twins (*_ref).  This kernel runs BOTH on the GPU over pseudo-random but well-formed inputs (every hop word points
 * to a higher lane or carries an exit flag, so every chain ends) and counts the rounds in which any output differs:
 * res[0] follow_tokens, res[1] hop_chain, res[2] hop_chain_pairs.  Diagnostic (k4lz4_selftest_chains); no product
 * path calls it.
 */
#pragma once
#include "k4lz4_decode.hpp"
#include "k4lz4_encode_fast.hpp"

namespace k4 {

__device__ __forceinline__ uint32_t st_rand(uint32_t &s)
{
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;     /* xorshift32 */
    return s;
}

/* a hop word of lane `lane` as the encoder would build it: lane after the match (127 = outside the window), flags */
__device__ __forceinline__ uint32_t st_hop_word(uint32_t r, int lane, bool valid)
{
    const uint32_t qn_full = (uint32_t)lane + MINMATCH + (r % 21u);
    const uint32_t qn = qn_full < 64u ? qn_full : 127u;
    uint32_t w = qn | (valid ? 0u : 0x800u);
    if (((r >> 8) & 7u) == 0u) w |= 0x100u;
    if (((r >> 11) & 31u) == 0u) w |= 0x200u;
    if (qn < 64u && ((r >> 16) % 5u) == 0u) w |= 0x400u;
    return w;
}

__global__ __launch_bounds__(64) void k4_chain_selftest_kernel(uint32_t seed, int rounds, uint32_t *res)
{
    const int lane = lane_id();
    uint32_t s = seed * 2654435761u + (uint32_t)blockIdx.x * 40503u + (uint32_t)lane * 9973u + 1u;
    uint32_t su = uni(seed * 747796405u + (uint32_t)blockIdx.x * 2891336453u + 7u);      /* wave-uniform stream */
    uint32_t bad_tok = 0, bad_hop = 0, bad_pair = 0;
    for (int it = 0; it < rounds; it++) {
        /* ---- follow_tokens ---- */
        {
            const uint32_t r = st_rand(s);
            const bool fast = (r & 7u) != 0u;
            const uint32_t next = (uint32_t)lane + 3u + ((r >> 3) % 12u);
            const uint32_t word = token_word(fast, next, lane);
            unsigned long long Ta, Tb;
            uint32_t ia, ib;
            unsigned long long Tc;
            uint32_t ic;
            follow_tokens(word, Ta, ia, true);                  /* two links per hop */
            follow_tokens1(word, Tc, ic);                       /* one */
            follow_tokens_ref(word, Tb, ib);
            if (Ta != Tb || ia != ib || Tc != Tb || ic != ib) bad_tok++;
        }
        const uint32_t ru = uni(st_rand(su));
        const uint32_t nvalid = 64u - (ru % 3u == 0u ? (ru >> 4) % 20u : 0u);          /* lanes past the last probe position */
        const bool valid = (uint32_t)lane < nvalid;
        const unsigned long long inv_m = __ballot(!valid);
        /* ---- hop_chain ---- */
        {
            const uint32_t r = st_rand(s);
            const uint32_t hopv = st_hop_word(r, lane, valid);
            const unsigned long long hmx = __ballot(valid && ((r >> 24) % 3u) == 0u) | inv_m;
            uint32_t qa = (ru >> 8) % 9u, qb = qa, hva = 0, hvb = 0;
            unsigned long long ha = 0, hb = 0, sa, sb;
            int fa = 0, fb = 0;
            hop_chain(hmx, hopv, qa, ha, fa, hva, sa);
            hop_chain_ref(hmx, hopv, qb, hb, fb, hvb, sb);
            if (qa != qb || ha != hb || fa != fb || hva != hvb || sa != sb) bad_hop++;
        }
        /* ---- hop_chain_pairs ---- */
        {
            const uint32_t r = st_rand(s), r2 = st_rand(s);
            const bool many = valid && (r2 % 11u) == 0u;
            uint32_t hopa = st_hop_word(r, lane, valid), hopb = hopa;
            const uint32_t hopB = many ? 0x1040u : st_hop_word(r2, lane, valid);
            const bool later = lane > 0 && ((r2 >> 20) & 3u) == 0u;                     /* has a candidate lane below it */
            const uint32_t j1c = 63u - (later ? (r2 >> 8) % (uint32_t)lane : (uint32_t)lane);
            const unsigned long long hm0 = __ballot(valid && ((r >> 24) % 3u) == 0u) | inv_m;
            const unsigned long long hmB = __ballot(valid && (((r2 >> 24) % 3u) == 0u || many)) | inv_m;
            unsigned long long hma = hm0, hmb = hm0, la = 0, lb = 0, ha = 0, hb = 0, sa, sb;
            uint32_t qa = (ru >> 12) % 9u, qb = qa, hva = 0, hvb = 0;
            int fa = 0, fb = 0;
            hop_chain_pairs(hma, hmB, hopa, hopB, j1c, la, qa, ha, fa, hva, sa);
            hop_chain_pairs_ref(hmb, hmB, hopb, hopB, j1c, lb, qb, hb, fb, hvb, sb);
            const bool words_differ = __ballot(hopa != hopb) != 0ull;
            if (qa != qb || ha != hb || fa != fb || hva != hvb || sa != sb || hma != hmb || la != lb || words_differ) bad_pair++;
        }
    }
    if (lane == 0) {
        if (bad_tok) atomicAdd(&res[0], bad_tok);
        if (bad_hop) atomicAdd(&res[1], bad_hop);
        if (bad_pair) atomicAdd(&res[2], bad_pair);
    }
}

}  // namespace k4
