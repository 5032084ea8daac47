/*
 * k4lz4_oracle_hc.c -- CPU oracle, HC (hash-chain) encoder.  TEST INFRASTRUCTURE ONLY.
 * Placeholder until the L03..L09 restatement lands.
 */
#include <stdint.h>
#define K4O_API __attribute__((visibility("default")))
K4O_API int k4o_compress_hc(const uint8_t *src, uint8_t *dst, int src_len, int dst_cap, int level)
{
    (void)src; (void)dst; (void)src_len; (void)dst_cap; (void)level;
    return 0;
}
