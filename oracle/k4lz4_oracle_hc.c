/*
 * k4lz4_oracle_hc.c -- CPU oracle, HC (hash-chain) encoder, levels 3..9.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the semantics of the reference's managed engine for one independent
 * block (noDictCtx, no external dictionary, favorCompressionRatio):
 *   LZ4_compress_HC -> _extStateHC -> _fastReset      Engine/x64/LL64.high.cs:1336-1381
 *   LZ4_initStreamHC / LZ4HC_init_internal / clearTables  Engine/LL.high.cs:56-69,:142-166
 *   LZ4HC_compress_generic(_internal), clTable         Engine/x64/LL64.high.cs:1124-1189
 *   LZ4HC_compress_hashChain                           Engine/x64/LL64.high.cs:512-800
 *   LZ4HC_InsertAndGetWiderMatch (+ pattern analysis)  Engine/x64/LL64.high.cs:70-383
 *   LZ4HC_Insert, LZ4HC_countBack, pattern helpers     Engine/LL.high.cs:91-122,:209-264,
 *                                                      Engine/x64/LL64.high.cs:37-68
 *   LZ4HC_encodeSequence                               Engine/x64/LL64.high.cs:435-510
 * Levels 10..12 (LZ4HC_compress_optimal, :802-1122) are outside the hot path and
 * return 0 here.
 *
 * Parity pin: byte equality with the system liblz4.so.1 (1.9.3) LZ4_compress_HC on the
 * reproducible fixtures (tests/test_oracle_pins.py); the reference's HC goldens
 * (ChecksumBlockTests.cs:125-172) need the Silesia corpus, which is not available offline.
 * Positions are block offsets; `index = offset + 65536` reproduces the reference's index space
 * (startingOffset 64 KB, LL.high.cs:158-165) where the arithmetic depends on it.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define K4O_API __attribute__((visibility("default")))

enum {
    MINMATCH = 4, LASTLITERALS = 5, MFLIMIT = 12, ML_BITS = 4, ML_MASK = 15, RUN_MASK = 15,
    DISTANCE_MAX = 65535, OPTIMAL_ML = (ML_MASK - 1) + MINMATCH, MAX_INPUT_SIZE = 0x7E000000,
    HASH_LOG = 15, HASH_SIZE = 1 << HASH_LOG, MAXD = 1 << 16, START = 65536
};

typedef struct {
    uint32_t hash[HASH_SIZE];   /* LL.types.high.cs:35 */
    uint16_t chain[MAXD];       /* :36 */
    uint32_t next_to_update;
    const uint8_t *base;        /* base + index = byte; base = src - START */
    const uint8_t *src;
} hc_t;

static inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t hash_ptr(const uint8_t *p) { return (rd32(p) * 2654435761u) >> (MINMATCH * 8 - HASH_LOG); }

/* LL64.tools.cs:86-133 */
static uint32_t count(const uint8_t *a, const uint8_t *b, const uint8_t *alimit)
{
    const uint8_t *start = a;
    while (a + 8 <= alimit) {
        uint64_t d = rd64(a) ^ rd64(b);
        if (d) return (uint32_t)(a - start) + (uint32_t)(__builtin_ctzll(d) >> 3);
        a += 8; b += 8;
    }
    while (a < alimit && *a == *b) { a++; b++; }
    return (uint32_t)(a - start);
}

/* LL.high.cs:102-122 */
static void hc_insert(hc_t *c, uint32_t target)
{
    uint32_t idx = c->next_to_update;
    while (idx < target) {
        uint32_t h = hash_ptr(c->base + idx);
        uint32_t delta = idx - c->hash[h];
        if (delta > DISTANCE_MAX) delta = DISTANCE_MAX;
        c->chain[(uint16_t)idx] = (uint16_t)delta;
        c->hash[h] = idx;
        idx++;
    }
    c->next_to_update = target;
}

/* LL.high.cs:216-230 */
static int count_back(const uint8_t *ip, const uint8_t *match, const uint8_t *imin, const uint8_t *mmin)
{
    int back = 0;
    int64_t m1 = imin - ip, m2 = mmin - match;
    int min = (int)(m1 > m2 ? m1 : m2);
    while (back > min && ip[back - 1] == match[back - 1]) back--;
    return back;
}

/* LL64.high.cs:37-68 */
static uint32_t count_pattern(const uint8_t *ip, const uint8_t *iend, uint32_t pattern32)
{
    const uint8_t *start = ip;
    uint64_t pattern = pattern32;
    pattern |= pattern << 32;
    while (ip < iend - 7) {
        uint64_t diff = rd64(ip) ^ pattern;
        if (diff == 0) { ip += 8; continue; }
        ip += __builtin_ctzll(diff) >> 3;
        return (uint32_t)(ip - start);
    }
    uint64_t pb = pattern;
    while (ip < iend && *ip == (uint8_t)pb) { ip++; pb >>= 8; }
    return (uint32_t)(ip - start);
}

/* LL.high.cs:232-254 */
static uint32_t reverse_count_pattern(const uint8_t *ip, const uint8_t *ilow, uint32_t pattern)
{
    const uint8_t *start = ip;
    while (ip >= ilow + 4) {
        if (rd32(ip - 4) != pattern) break;
        ip -= 4;
    }
    {
        uint8_t bytes[4];
        memcpy(bytes, &pattern, 4);
        int k = 3;
        while (ip > ilow) {
            if (ip[-1] != bytes[k]) break;
            ip--; k--;
        }
    }
    return (uint32_t)(start - ip);
}

/* LL.high.cs:209-214 */
static int protect_dict_end(uint32_t dict_limit, uint32_t match_index) { return (uint32_t)((dict_limit - 1) - match_index) >= 3; }

/*
 * LL64.high.cs:70-383 for dict == noDictCtx, favorCompressionRatio, chainSwap == false,
 * dictLimit == lowLimit == START (every candidate lies in the current prefix).
 * ip / ilow / ihigh are pointers into src; returns `longest`, updates *matchpos / *startpos.
 */
static int insert_and_get_wider_match(hc_t *c, const uint8_t *ip, const uint8_t *ilow, const uint8_t *ihigh,
                                      int longest, const uint8_t **matchpos, const uint8_t **startpos,
                                      int max_attempts, int pattern_analysis)
{
    const uint8_t *base = c->base;
    const uint32_t dict_limit = START;
    const uint8_t *low_prefix = base + dict_limit;
    const uint32_t ip_index = (uint32_t)(ip - base);
    const uint32_t lowest = (START + (DISTANCE_MAX + 1) > ip_index) ? START : ip_index - DISTANCE_MAX;
    const int look_back = (int)(ip - ilow);
    int attempts = max_attempts;
    const uint32_t pattern = rd32(ip);
    uint32_t match_index;
    int repeat = 0;  /* 0 untested, 1 not, 2 confirmed */
    uint32_t src_pattern_length = 0;

    hc_insert(c, ip_index);
    match_index = c->hash[hash_ptr(ip)];

    while (match_index >= lowest && attempts != 0) {
        int match_length = 0;
        attempts--;
        {
            const uint8_t *mp = base + match_index;
            if (rd16(ilow + longest - 1) == rd16(mp - look_back + longest - 1)) {
                if (rd32(mp) == pattern) {
                    int back = look_back != 0 ? count_back(ip, mp, ilow, low_prefix) : 0;
                    match_length = MINMATCH + (int)count(ip + MINMATCH, mp + MINMATCH, ihigh);
                    match_length -= back;
                    if (match_length > longest) {
                        longest = match_length;
                        *matchpos = mp + back;
                        *startpos = ip + back;
                    }
                }
            }
        }
        {
            uint32_t dist_next = c->chain[(uint16_t)match_index];
            if (pattern_analysis && dist_next == 1) {          /* :208-337 (matchChainPos == 0) */
                uint32_t cand_idx = match_index - 1;
                if (repeat == 0) {
                    if (((pattern & 0xFFFF) == (pattern >> 16)) & ((pattern & 0xFF) == (pattern >> 24))) {
                        repeat = 2;
                        src_pattern_length = count_pattern(ip + 4, ihigh, pattern) + 4;
                    } else {
                        repeat = 1;
                    }
                }
                if (repeat == 2 && cand_idx >= lowest && protect_dict_end(dict_limit, cand_idx)) {
                    const uint8_t *mp = base + cand_idx;
                    if (rd32(mp) == pattern) {
                        uint32_t fwd = count_pattern(mp + 4, ihigh, pattern) + 4;
                        uint32_t back_len = reverse_count_pattern(mp, low_prefix, pattern);
                        uint32_t cur_seg;
                        {
                            uint32_t a = cand_idx - back_len;
                            uint32_t mx = a > lowest ? a : lowest;
                            back_len = cand_idx - mx;
                        }
                        cur_seg = back_len + fwd;
                        if (cur_seg >= src_pattern_length && fwd <= src_pattern_length) {
                            uint32_t nmi = cand_idx + fwd - src_pattern_length;
                            if (protect_dict_end(dict_limit, nmi)) match_index = nmi;
                            else match_index = dict_limit;
                        } else {
                            uint32_t nmi = cand_idx - back_len;
                            if (!protect_dict_end(dict_limit, nmi)) {
                                match_index = dict_limit;
                            } else {
                                match_index = nmi;
                                if (look_back == 0) {
                                    uint32_t max_ml = cur_seg < src_pattern_length ? cur_seg : src_pattern_length;
                                    if ((uint32_t)longest < max_ml) {
                                        if ((uint32_t)(ip - base) - match_index > DISTANCE_MAX) break;
                                        longest = (int)max_ml;
                                        *matchpos = base + match_index;
                                        *startpos = ip;
                                    }
                                    {
                                        uint32_t d = c->chain[(uint16_t)match_index];
                                        if (d > match_index) break;
                                        match_index -= d;
                                    }
                                }
                            }
                        }
                        continue;
                    }
                }
            }
        }
        match_index -= c->chain[(uint16_t)match_index];
    }
    return longest;
}

/* LL64.high.cs:435-510; returns 1 on output overflow */
static int encode_sequence(const uint8_t **ip, uint8_t **op, const uint8_t **anchor, int match_length,
                           const uint8_t *match, int limited, uint8_t *oend)
{
    size_t length;
    uint8_t *token = (*op)++;
    length = (size_t)(*ip - *anchor);
    if (limited && (*op + (length / 255) + length + (2 + 1 + LASTLITERALS)) > oend) return 1;
    if (length >= RUN_MASK) {
        size_t len = length - RUN_MASK;
        *token = (uint8_t)(RUN_MASK << ML_BITS);
        for (; len >= 255; len -= 255) *(*op)++ = 255;
        *(*op)++ = (uint8_t)len;
    } else {
        *token = (uint8_t)(length << ML_BITS);
    }
    memcpy(*op, *anchor, length);
    *op += length;
    {
        uint16_t off = (uint16_t)(*ip - match);
        memcpy(*op, &off, 2);
        *op += 2;
    }
    length = (size_t)match_length - MINMATCH;
    if (limited && (*op + (length / 255) + (1 + LASTLITERALS) > oend)) return 1;
    if (length >= ML_MASK) {
        *token += ML_MASK;
        length -= ML_MASK;
        for (; length >= 510; length -= 510) { *(*op)++ = 255; *(*op)++ = 255; }
        if (length >= 255) { length -= 255; *(*op)++ = 255; }
        *(*op)++ = (uint8_t)length;
    } else {
        *token += (uint8_t)length;
    }
    *ip += match_length;
    *anchor = *ip;
    return 0;
}

/* LL64.high.cs:512-800 with limit in {notLimited, limitedOutput} */
static int compress_hash_chain(hc_t *ctx, const uint8_t *source, uint8_t *dest, int input_size, int max_output,
                               int max_attempts, int limited)
{
    const int pattern_analysis = max_attempts > 128;
    const uint8_t *ip = source, *anchor = ip;
    const uint8_t *iend = ip + input_size;
    const uint8_t *mflimit = iend - MFLIMIT;
    const uint8_t *matchlimit = iend - LASTLITERALS;
    uint8_t *op = dest;
    uint8_t *oend = op + max_output;
    int ml0, ml, ml2, ml3;
    const uint8_t *start0, *ref0, *ref = NULL, *start2 = NULL, *ref2 = NULL, *start3 = NULL, *ref3 = NULL;

    if (input_size < MFLIMIT + 1) goto last_literals;

    while (ip <= mflimit) {
        {
            const uint8_t *useless = ip;
            ml = insert_and_get_wider_match(ctx, ip, ip, matchlimit, MINMATCH - 1, &ref, &useless, max_attempts,
                                            pattern_analysis);
        }
        if (ml < MINMATCH) { ip++; continue; }
        start0 = ip; ref0 = ref; ml0 = ml;

    search2:
        if (ip + ml <= mflimit)
            ml2 = insert_and_get_wider_match(ctx, ip + ml - 2, ip + 0, matchlimit, ml, &ref2, &start2, max_attempts,
                                             pattern_analysis);
        else
            ml2 = ml;

        if (ml2 == ml) {
            if (encode_sequence(&ip, &op, &anchor, ml, ref, limited, oend)) return 0;
            continue;
        }
        if (start0 < ip) {
            if (start2 < ip + ml0) { ip = start0; ref = ref0; ml = ml0; }
        }
        if ((start2 - ip) < 3) { ml = ml2; ip = start2; ref = ref2; goto search2; }

    search3:
        if ((start2 - ip) < OPTIMAL_ML) {
            int correction;
            int new_ml = ml;
            if (new_ml > OPTIMAL_ML) new_ml = OPTIMAL_ML;
            if (ip + new_ml > start2 + ml2 - MINMATCH) new_ml = (int)(start2 - ip) + ml2 - MINMATCH;
            correction = new_ml - (int)(start2 - ip);
            if (correction > 0) { start2 += correction; ref2 += correction; ml2 -= correction; }
        }
        if (start2 + ml2 <= mflimit)
            ml3 = insert_and_get_wider_match(ctx, start2 + ml2 - 3, start2, matchlimit, ml2, &ref3, &start3,
                                             max_attempts, pattern_analysis);
        else
            ml3 = ml2;

        if (ml3 == ml2) {
            if (start2 < ip + ml) ml = (int)(start2 - ip);
            if (encode_sequence(&ip, &op, &anchor, ml, ref, limited, oend)) return 0;
            ip = start2;
            if (encode_sequence(&ip, &op, &anchor, ml2, ref2, limited, oend)) return 0;
            continue;
        }
        if (start3 < ip + ml + 3) {
            if (start3 >= (ip + ml)) {
                if (start2 < ip + ml) {
                    int correction = (int)(ip + ml - start2);
                    start2 += correction; ref2 += correction; ml2 -= correction;
                    if (ml2 < MINMATCH) { start2 = start3; ref2 = ref3; ml2 = ml3; }
                }
                if (encode_sequence(&ip, &op, &anchor, ml, ref, limited, oend)) return 0;
                ip = start3; ref = ref3; ml = ml3;
                start0 = start2; ref0 = ref2; ml0 = ml2;
                goto search2;
            }
            start2 = start3; ref2 = ref3; ml2 = ml3;
            goto search3;
        }
        if (start2 < ip + ml) {
            if ((start2 - ip) < OPTIMAL_ML) {
                int correction;
                if (ml > OPTIMAL_ML) ml = OPTIMAL_ML;
                if (ip + ml > start2 + ml2 - MINMATCH) ml = (int)(start2 - ip) + ml2 - MINMATCH;
                correction = ml - (int)(start2 - ip);
                if (correction > 0) { start2 += correction; ref2 += correction; ml2 -= correction; }
            } else {
                ml = (int)(start2 - ip);
            }
        }
        if (encode_sequence(&ip, &op, &anchor, ml, ref, limited, oend)) return 0;
        ip = start2; ref = ref2; ml = ml2;
        start2 = start3; ref2 = ref3; ml2 = ml3;
        goto search3;
    }

last_literals:
    {
        size_t last_run = (size_t)(iend - anchor);
        size_t lit_length = (last_run + 255 - RUN_MASK) / 255;
        size_t total = 1 + lit_length + last_run;
        if (limited && (op + total > oend)) return 0;
        if (last_run >= RUN_MASK) {
            size_t acc = last_run - RUN_MASK;
            *op++ = (uint8_t)(RUN_MASK << ML_BITS);
            for (; acc >= 255; acc -= 255) *op++ = 255;
            *op++ = (uint8_t)acc;
        } else {
            *op++ = (uint8_t)(last_run << ML_BITS);
        }
        memcpy(op, anchor, last_run);
        op += last_run;
    }
    return (int)(op - dest);
}

/* clTable nbSearches for the hash-chain levels (LL64.high.cs:1124-1138) */
static int nb_searches(int level)
{
    static const int t[10] = {2, 2, 2, 4, 8, 16, 32, 64, 128, 256};
    return t[level];
}

/* LL64.LZ4_compress_HC (LL64.high.cs:1367-1381) for levels <= 9 */
K4O_API int k4o_compress_hc(const uint8_t *src, uint8_t *dst, int src_len, int dst_cap, int level)
{
    if ((uint32_t)src_len > (uint32_t)MAX_INPUT_SIZE) return 0;          /* :1153 */
    if (level < 1) level = 9;                                           /* LZ4HC_CLEVEL_DEFAULT */
    if (level > 12) level = 12;
    if (level >= 10) return 0;                                          /* optimal parser: not restated */
    hc_t *ctx = (hc_t *)malloc(sizeof(hc_t));
    if (!ctx) return 0;
    memset(ctx->hash, 0, sizeof ctx->hash);                             /* LZ4HC_clearTables */
    memset(ctx->chain, 0xFF, sizeof ctx->chain);
    ctx->next_to_update = START;                                        /* LZ4HC_init_internal */
    ctx->base = src - START;
    ctx->src = src;
    int bound = src_len > MAX_INPUT_SIZE ? 0 : src_len + src_len / 255 + 16;
    int limited = dst_cap < bound;
    int r = compress_hash_chain(ctx, src, dst, src_len, dst_cap, nb_searches(level), limited);
    free(ctx);
    return r;
}
