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
 * Levels 10..12 (LZ4HC_compress_optimal, :802-1122, with LZ4HC_FindLongerMatch and the chain swap :172-206) are
 * restated further down in this file as well.
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
static int insert_and_get_wider_match_x(hc_t *c, const uint8_t *ip, const uint8_t *ilow, const uint8_t *ihigh,
                                        int longest, const uint8_t **matchpos, const uint8_t **startpos,
                                        int max_attempts, int pattern_analysis, int chain_swap)
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
    uint32_t match_chain_pos = 0;                                   /* LL64.high.cs:95 */

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
        if (chain_swap && match_length == longest) {           /* :172-206 better match => select a better chain */
            if (match_index + (uint32_t)longest <= ip_index) {
                const int k_trigger = 4;
                uint32_t distance_to_next = 1;
                const int end = longest - MINMATCH + 1;
                int step = 1, accel = 1 << k_trigger, pos;
                for (pos = 0; pos < end; pos += step) {
                    uint32_t cd = c->chain[(uint16_t)(match_index + (uint32_t)pos)];
                    step = (accel++ >> k_trigger);
                    if (cd > distance_to_next) {
                        distance_to_next = cd;
                        match_chain_pos = (uint32_t)pos;
                        accel = 1 << k_trigger;
                    }
                }
                if (distance_to_next > 1) {
                    if (distance_to_next > match_index) break;
                    match_index -= distance_to_next;
                    continue;
                }
            }
        }
        {
            uint32_t dist_next = c->chain[(uint16_t)match_index];
            if (pattern_analysis && dist_next == 1 && match_chain_pos == 0) {   /* :208-337 */
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
        match_index -= c->chain[(uint16_t)(match_index + match_chain_pos)];   /* :340 follow current chain */
    }
    return longest;
}

static int insert_and_get_wider_match(hc_t *c, const uint8_t *ip, const uint8_t *ilow, const uint8_t *ihigh,
                                      int longest, const uint8_t **matchpos, const uint8_t **startpos,
                                      int max_attempts, int pattern_analysis)
{
    return insert_and_get_wider_match_x(c, ip, ilow, ihigh, longest, matchpos, startpos, max_attempts, pattern_analysis, 0);
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

/* ---- optimal parser, levels 10..12 (LL64.high.cs:802-1122; LL.high.cs:267-287 prices; :404-445 FindLongerMatch) ---- */
#define OPT_NUM (1 << 12)
#define TRAILING_LITERALS 3
typedef struct { int price, off, mlen, litlen; } opt_t;

static int literals_price(int litlen)
{
    int price = litlen;
    if (litlen >= (int)RUN_MASK) price += 1 + ((litlen - (int)RUN_MASK) / 255);
    return price;
}
static int sequence_price(int litlen, int mlen)
{
    int price = 1 + 2;
    price += literals_price(litlen);
    if (mlen >= (int)(ML_MASK + MINMATCH)) price += 1 + ((mlen - (int)(ML_MASK + MINMATCH)) / 255);
    return price;
}

/* LZ4HC_FindLongerMatch with favorCompressionRatio: pattern analysis and chain swap on, no look-back */
static void find_longer_match(hc_t *c, const uint8_t *ip, const uint8_t *ihigh, int min_len, int nb_searches, int *len, int *off)
{
    const uint8_t *mp = NULL, *sp = ip;
    int ml = insert_and_get_wider_match_x(c, ip, ip, ihigh, min_len, &mp, &sp, nb_searches, 1, 1);
    *len = 0; *off = 0;
    if (ml <= min_len) return;
    *len = ml;
    *off = (int)(ip - mp);
}

static int compress_optimal(hc_t *ctx, const uint8_t *source, uint8_t *dst, int src_size, int dst_capacity, int nb_searches,
                            size_t sufficient_len, int limited, int full_update)
{
    opt_t *opt = (opt_t *)malloc(sizeof(opt_t) * (OPT_NUM + TRAILING_LITERALS));
    const uint8_t *ip = source, *anchor = ip;
    const uint8_t *iend = ip + src_size, *mflimit = iend - MFLIMIT, *matchlimit = iend - LASTLITERALS;
    uint8_t *op = dst, *oend = op + dst_capacity;
    int result = 0;
    if (!opt) return 0;
    if (sufficient_len >= OPT_NUM) sufficient_len = OPT_NUM - 1;

    while (ip <= mflimit) {
        int llen = (int)(ip - anchor);
        int best_mlen, best_off, cur, last_match_pos = 0;
        int first_len, first_off;
        find_longer_match(ctx, ip, matchlimit, MINMATCH - 1, nb_searches, &first_len, &first_off);
        if (first_len == 0) { ip++; continue; }
        if ((size_t)first_len > sufficient_len) {
            if (encode_sequence(&ip, &op, &anchor, first_len, ip - first_off, limited, oend)) goto overflow;
            continue;
        }
        for (int r = 0; r < MINMATCH; r++) {
            opt[r].mlen = 1; opt[r].off = 0; opt[r].litlen = llen + r; opt[r].price = literals_price(llen + r);
        }
        for (int mlen = MINMATCH; mlen <= first_len; mlen++) {
            opt[mlen].mlen = mlen; opt[mlen].off = first_off; opt[mlen].litlen = llen; opt[mlen].price = sequence_price(llen, mlen);
        }
        last_match_pos = first_len;
        for (int a = 1; a <= TRAILING_LITERALS; a++) {
            opt[last_match_pos + a].mlen = 1; opt[last_match_pos + a].off = 0; opt[last_match_pos + a].litlen = a;
            opt[last_match_pos + a].price = opt[last_match_pos].price + literals_price(a);
        }
        for (cur = 1; cur < last_match_pos; cur++) {
            const uint8_t *cur_ptr = ip + cur;
            int new_len, new_off;
            if (cur_ptr > mflimit) break;
            if (full_update) {
                if ((opt[cur + 1].price <= opt[cur].price) && (opt[cur + MINMATCH].price < opt[cur].price + 3)) continue;
            } else {
                if (opt[cur + 1].price <= opt[cur].price) continue;
            }
            if (full_update) find_longer_match(ctx, cur_ptr, matchlimit, MINMATCH - 1, nb_searches, &new_len, &new_off);
            else find_longer_match(ctx, cur_ptr, matchlimit, last_match_pos - cur, nb_searches, &new_len, &new_off);
            if (new_len == 0) continue;
            if (((size_t)new_len > sufficient_len) || (new_len + cur >= OPT_NUM)) {
                best_mlen = new_len; best_off = new_off; last_match_pos = cur + 1;
                goto encode;
            }
            {
                int base_litlen = opt[cur].litlen;
                for (int litlen = 1; litlen < MINMATCH; litlen++) {
                    int price = opt[cur].price - literals_price(base_litlen) + literals_price(base_litlen + litlen);
                    int pos = cur + litlen;
                    if (price < opt[pos].price) {
                        opt[pos].mlen = 1; opt[pos].off = 0; opt[pos].litlen = base_litlen + litlen; opt[pos].price = price;
                    }
                }
            }
            {
                int match_ml = new_len;
                for (int ml = MINMATCH; ml <= match_ml; ml++) {
                    int pos = cur + ml, price, ll;
                    if (opt[cur].mlen == 1) {
                        ll = opt[cur].litlen;
                        price = ((cur > ll) ? opt[cur - ll].price : 0) + sequence_price(ll, ml);
                    } else {
                        ll = 0;
                        price = opt[cur].price + sequence_price(0, ml);
                    }
                    if (pos > last_match_pos + TRAILING_LITERALS || price <= opt[pos].price) {
                        if ((ml == match_ml) && (last_match_pos < pos)) last_match_pos = pos;
                        opt[pos].mlen = ml; opt[pos].off = new_off; opt[pos].litlen = ll; opt[pos].price = price;
                    }
                }
            }
            for (int a = 1; a <= TRAILING_LITERALS; a++) {
                opt[last_match_pos + a].mlen = 1; opt[last_match_pos + a].off = 0; opt[last_match_pos + a].litlen = a;
                opt[last_match_pos + a].price = opt[last_match_pos].price + literals_price(a);
            }
        }
        best_mlen = opt[last_match_pos].mlen;
        best_off = opt[last_match_pos].off;
        cur = last_match_pos - best_mlen;
    encode:
        {
            int candidate_pos = cur, sel_ml = best_mlen, sel_off = best_off;
            for (;;) {
                int next_ml = opt[candidate_pos].mlen, next_off = opt[candidate_pos].off;
                opt[candidate_pos].mlen = sel_ml; opt[candidate_pos].off = sel_off;
                sel_ml = next_ml; sel_off = next_off;
                if (next_ml > candidate_pos) break;
                candidate_pos -= next_ml;
            }
        }
        {
            int r = 0;
            while (r < last_match_pos) {
                int ml = opt[r].mlen, offset = opt[r].off;
                if (ml == 1) { ip++; r++; continue; }
                r += ml;
                if (encode_sequence(&ip, &op, &anchor, ml, ip - offset, limited, oend)) goto overflow;
            }
        }
    }
    {   /* _last_literals (:1062-1098), limitedOutput / notLimited only */
        size_t last_run = (size_t)(iend - anchor);
        size_t lit_length = (last_run + 255 - RUN_MASK) / 255;
        size_t total = 1 + lit_length + last_run;
        if (limited && (op + total > oend)) goto overflow;
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
    result = (int)(op - dst);
overflow:
    free(opt);
    return result;
}

/* LL64.LZ4_compress_HC (LL64.high.cs:1367-1381): clTable (:1124-1138) picks hash chain (3..9) or the optimal parser (10..12) */
K4O_API int k4o_compress_hc(const uint8_t *src, uint8_t *dst, int src_len, int dst_cap, int level)
{
    if ((uint32_t)src_len > (uint32_t)MAX_INPUT_SIZE) return 0;          /* :1153 */
    if (level < 1) level = 9;                                           /* LZ4HC_CLEVEL_DEFAULT */
    if (level > 12) level = 12;
    hc_t *ctx = (hc_t *)malloc(sizeof(hc_t));
    if (!ctx) return 0;
    memset(ctx->hash, 0, sizeof ctx->hash);                             /* LZ4HC_clearTables */
    memset(ctx->chain, 0xFF, sizeof ctx->chain);
    ctx->next_to_update = START;                                        /* LZ4HC_init_internal */
    ctx->base = src - START;
    ctx->src = src;
    int bound = src_len > MAX_INPUT_SIZE ? 0 : src_len + src_len / 255 + 16;
    int limited = dst_cap < bound;
    int r;
    if (level >= 10) {
        static const int nbs[3] = {96, 512, 16384};
        static const size_t tgt[3] = {64, 128, OPT_NUM};
        r = compress_optimal(ctx, src, dst, src_len, dst_cap, nbs[level - 10], tgt[level - 10], limited, level == 12);
    } else {
        r = compress_hash_chain(ctx, src, dst, src_len, dst_cap, nb_searches(level), limited);
    }
    free(ctx);
    return r;
}
