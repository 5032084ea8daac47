/*
 * k4lz4_oracle.c -- CPU oracle for the K4os.Compression.LZ4 block hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product (libk4lz4.so, the
 * k4os.compression.lz4_amd package) links, loads or calls this file.  It is the
 * checker used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * What it is: a plain-C restatement of the *semantics* of the reference's managed
 * LL64 engine for the arms on the hot path:
 *   - LZ4_compress_fast / LZ4_compress_generic, arms noDict + {notLimited,
 *     limitedOutput} + {byU16, byU32/hash5}, acceleration >= 1
 *         reference: src/K4os.Compression.LZ4/Engine/x64/LL64.fast.cs:34-576
 *                    Engine/x64/LL64.tools.cs:86-153, Engine/LL.tools.cs:38-148,235-239
 *                    Engine/LL.types.cs:18-78
 *   - LZ4_decompress_safe (endOnInputSize, full, noDict) plus the partial / dictionary
 *     variants of LZ4_decompress_generic
 *         reference: Engine/x64/LL64.dec.cs:123-556, Engine/LL.tools.cs:165-193
 *   - LZ4Codec.Encode / Decode return-value mapping
 *         reference: src/K4os.Compression.LZ4/LZ4Codec.cs:30-52,104-115
 *   - LZ4Pickler envelope V0 (Pickle / Unpickle)
 *         reference: LZ4Pickler.pickle.cs:51-228, LZ4Pickler.unpickle.cs:39-158
 *
 * Parity pin (round 4): PINNED TO THE REFERENCE ITSELF.  oracle/make_ref.py respells the reference's own
 * engine files (Engine/x64/LL64.*.cs, Engine/x32/LL32.*.cs, Engine/LL.*.cs, Internal/Mem*.cs) as C++ and g++
 * compiles them into oracle/_ref/libk4ref.so; tests/test_ref_pins.py compares every entry point of this file
 * with it byte for byte -- all 4096 bench blocks at L00, 384 at L03, 216 at L09/L10/L12, 200 configs[3]
 * messages through both engines, 1 500 mutated streams (return value incl. error position), partial and
 * dictionary arms, the Enforce32 arm against LL32.  Older pins stay in tests/test_oracle_pins.py: the in-repo
 * known-answer fixture assets/issue64 (tests/golden/issue64_*.bin), the probe values of SURVEY.md 8c, and byte
 * equality with the system liblz4.so.1 (1.9.3).  The LZ4Codec / LZ4Pickler mappings below follow managed-array
 * code the translator does not take; they are pinned by reading and by the reference's reproducible tests.
 * Output bytes are exact; unlike the reference's wildcopy helpers this restatement never stores beyond the
 * returned length (the overshoot is unobservable: SURVEY.md 8a row a10).  Not run here: the reference's recorded
 * encode outputs (ChecksumBlockTests.cs:14-50,125-172: whole Silesia files; no corpus in this image,
 * tests/tools/fetch_silesia.md).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>

#define K4O_API __attribute__((visibility("default")))

enum {
    MINMATCH = 4,
    WILDCOPYLENGTH = 8,
    LASTLITERALS = 5,
    MFLIMIT = 12,
    MATCH_SAFEGUARD_DISTANCE = 2 * WILDCOPYLENGTH - MINMATCH, /* 12 */
    LZ4_MIN_LENGTH = MFLIMIT + 1,
    ML_BITS = 4,
    ML_MASK = 15,
    RUN_MASK = 15,
    DISTANCE_MAX = 65535,
    SKIP_TRIGGER = 6,
    LIMIT_64K = 65536 + (MFLIMIT - 1), /* LL.types.cs:77 */
    MAX_INPUT_SIZE = 0x7E000000
};

static inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* LL.tools.cs:38-40 */
K4O_API int k4o_compress_bound(int n)
{
    return n > MAX_INPUT_SIZE ? 0 : n + n / 255 + 16;
}

/* ------------------------------------------------------------------------------------
 * Fast encoder
 * ---------------------------------------------------------------------------------- */

/* LL.tools.cs:46-51 (hash4, byU16 => 13 bits) */
static inline uint32_t hash_seq4_u16(uint32_t seq) { return (seq * 2654435761u) >> (32 - 13); }
/* LL32 (Enforce32): LZ4_hashPosition has no hash5 arm (x32/LL32.tools.cs:141-148), so a byU32 table is indexed by
 * LZ4_hash4 with LZ4_HASHLOG = 12 bits (LL.tools.cs:46-51) */
static inline uint32_t hash_seq4_u32(uint32_t seq) { return (seq * 2654435761u) >> (32 - 12); }
/* LL.tools.cs:53-58 (hash5 on 64-bit, byU32 => 12 bits), LL64.tools.cs:135-143 */
static inline uint32_t hash_seq5_u32(uint64_t seq)
{
    return (uint32_t)(((seq << 24) * 889523592379ULL) >> (64 - 12));
}

/* LL64.tools.cs:86-133 -- number of equal bytes at a/b, a bounded by alimit.
 * (The reference steps 8/4/2/1 bytes; the value it returns is exactly this.) */
static inline uint32_t common_length(const uint8_t *a, const uint8_t *b, const uint8_t *alimit)
{
    const uint8_t *start = a;
    while (a + 8 <= alimit) {
        uint64_t diff = rd64(a) ^ rd64(b);
        if (diff) return (uint32_t)(a - start) + (uint32_t)(__builtin_ctzll(diff) >> 3);
        a += 8; b += 8;
    }
    while (a < alimit && *a == *b) { a++; b++; }
    return (uint32_t)(a - start);
}

typedef struct {
    uint32_t words[4096]; /* 16 KiB: 8192 x u16 or 4096 x u32 (LL.types.cs:29-39) */
} fast_table_t;

/*
 * LL64.fast.cs:34-513 for dict == noDict, dictIssue == noDictIssue.
 * `limited` selects the limitedOutput arm (cap checks at :247-251, :332-334, :472-473).
 * `by_u16` selects tableType byU16 (13-bit hash4, u16 slots) vs byU32 (12-bit hash5).
 * Offsets (not pointers) are used for all bounds arithmetic.
 */
static int fast_generic(fast_table_t *tbl, const uint8_t *src, uint8_t *dst, int src_len,
                        int dst_cap, int limited, int by_u16, int accel, int x32)
{
    uint16_t *t16 = (uint16_t *)tbl->words;
    uint32_t *t32 = tbl->words;
    const int64_t iend = src_len;
    const int64_t mflimit_plus_one = iend - MFLIMIT + 1;
    const int64_t matchlimit = iend - LASTLITERALS;
    const int64_t olimit = dst_cap;
    int64_t ip = 0, anchor = 0, op = 0;
    uint32_t forward_h;

    if ((uint32_t)src_len > (uint32_t)MAX_INPUT_SIZE) return 0;          /* :90 */
    if (by_u16 && src_len >= LIMIT_64K) return 0;                        /* :92 */
    if (src_len < LZ4_MIN_LENGTH) goto last_literals;                    /* :117 */

#define HASH_AT(pos) (by_u16 ? hash_seq4_u16(rd32(src + (pos))) : x32 ? hash_seq4_u32(rd32(src + (pos))) : hash_seq5_u32(rd64(src + (pos))))
#define TGET(h) (by_u16 ? (uint32_t)t16[h] : t32[h])
#define TPUT(h, v) do { if (by_u16) t16[h] = (uint16_t)(v); else t32[h] = (uint32_t)(v); } while (0)

    TPUT(HASH_AT(0), 0);                                                 /* :120 first byte */
    ip = 1;
    forward_h = HASH_AT(ip);

    for (;;) {
        int64_t match;
        int64_t token;
        int zero_literal_entry = 0;

        /* :156-234 search loop */
        {
            int64_t forward_ip = ip;
            int step = 1;
            int search_nb = accel << SKIP_TRIGGER;
            for (;;) {
                uint32_t h = forward_h;
                uint32_t current = (uint32_t)forward_ip;
                uint32_t match_index = TGET(h);
                ip = forward_ip;
                forward_ip += step;
                step = (search_nb++ >> SKIP_TRIGGER);
                if (forward_ip > mflimit_plus_one) goto last_literals;   /* :172 */
                match = match_index;
                forward_h = HASH_AT(forward_ip);
                TPUT(h, current);
                if (!by_u16 && match_index + DISTANCE_MAX < current) continue; /* :219-224 */
                if (rd32(src + match) == rd32(src + ip)) break;          /* :228 */
            }
        }

        /* :237-242 backward extension */
        while (ip > anchor && match > 0 && src[ip - 1] == src[match - 1]) { ip--; match--; }

        /* :244-272 token + literals */
        {
            uint32_t lit = (uint32_t)(ip - anchor);
            token = op++;
            if (limited && op + lit + (2 + 1 + LASTLITERALS) + lit / 255 > olimit) return 0;
            if (lit >= RUN_MASK) {
                int len = (int)(lit - RUN_MASK);
                dst[token] = (uint8_t)(RUN_MASK << ML_BITS);
                for (; len >= 255; len -= 255) dst[op++] = 255;
                dst[op++] = (uint8_t)len;
            } else {
                dst[token] = (uint8_t)(lit << ML_BITS);
            }
            memcpy(dst + op, src + anchor, lit);
            op += lit;
        }

    next_match:
        /* :299-304 offset */
        dst[op] = (uint8_t)((ip - match) & 0xff);
        dst[op + 1] = (uint8_t)(((ip - match) >> 8) & 0xff);
        op += 2;

        /* :326-382 match length */
        {
            uint32_t code = common_length(src + ip + MINMATCH, src + match + MINMATCH, src + matchlimit);
            ip += (int64_t)code + MINMATCH;
            if (limited && op + (1 + LASTLITERALS) + (code + 240) / 255 > olimit) return 0;
            if (code >= ML_MASK) {
                dst[token] += ML_MASK;
                code -= ML_MASK;
                while (code >= 255) { dst[op++] = 255; code -= 255; }
                dst[op++] = (uint8_t)code;
            } else {
                dst[token] += (uint8_t)code;
            }
        }
        (void)zero_literal_entry;

        anchor = ip;
        if (ip >= mflimit_plus_one) break;                               /* :391 */

        TPUT(HASH_AT(ip - 2), (uint32_t)(ip - 2));                       /* :394 */

        /* :410-463 test next position */
        {
            uint32_t h = HASH_AT(ip);
            uint32_t current = (uint32_t)ip;
            uint32_t match_index = TGET(h);
            match = match_index;
            TPUT(h, current);
            if ((by_u16 || match_index + DISTANCE_MAX >= current)
                && rd32(src + match) == rd32(src + ip)) {
                token = op++;
                dst[token] = 0;
                zero_literal_entry = 1;
                goto next_match;
            }
        }
        forward_h = HASH_AT(++ip);                                       /* :466 */
    }

last_literals:
    /* :469-503 */
    {
        uint32_t last_run = (uint32_t)(iend - anchor);
        if (limited && op + last_run + 1 + ((last_run + 255 - RUN_MASK) / 255) > olimit) return 0;
        if (last_run >= RUN_MASK) {
            uint32_t acc = last_run - RUN_MASK;
            dst[op++] = (uint8_t)(RUN_MASK << ML_BITS);
            for (; acc >= 255; acc -= 255) dst[op++] = 255;
            dst[op++] = (uint8_t)acc;
        } else {
            dst[op++] = (uint8_t)(last_run << ML_BITS);
        }
        memcpy(dst + op, src + anchor, last_run);
        op += last_run;
    }
    return (int)op;
#undef HASH_AT
#undef TGET
#undef TPUT
}

/* LL64.fast.cs:517-576 (LZ4_compress_fast_extState + LZ4_compress_fast) */
K4O_API int k4o_compress_fast(const uint8_t *src, uint8_t *dst, int src_len, int dst_cap, int accel)
{
    fast_table_t tbl;
    memset(&tbl, 0, sizeof tbl);                                         /* LZ4_initStream */
    if (accel < 1) accel = 1;
    int limited = !(dst_cap >= k4o_compress_bound(src_len));
    int by_u16 = src_len < LIMIT_64K;
    return fast_generic(&tbl, src, dst, src_len, limited ? dst_cap : 0, limited, by_u16, accel, 0);
}

/* LL32.LZ4_compress_fast as LZ4Codec.Enforce32 = true runs it in a 64-bit process (x32/LL32.fast.cs:517-576): the table
 * type for >= 64 KiB inputs is still byU32 (`sizeof(void*) < 8` is false, :543-545), only the hash differs.
 * Pinned to LL32 compiled here (oracle/_ref, k4ref_compress_fast_x32): tests/test_ref_pins.py. */
K4O_API int k4o_compress_fast_x32(const uint8_t *src, uint8_t *dst, int src_len, int dst_cap, int accel)
{
    fast_table_t tbl;
    memset(&tbl, 0, sizeof tbl);
    if (accel < 1) accel = 1;
    int limited = !(dst_cap >= k4o_compress_bound(src_len));
    int by_u16 = src_len < LIMIT_64K;
    return fast_generic(&tbl, src, dst, src_len, limited ? dst_cap : 0, limited, by_u16, accel, 1);
}

/* ------------------------------------------------------------------------------------
 * Decoder
 * ---------------------------------------------------------------------------------- */

/* LL.tools.cs:165-193.  err: 0 ok, -1 initial_error, -2 loop_error */
static uint32_t read_vle(const uint8_t *src, int64_t *ip, int64_t lencheck, int loop_check,
                         int initial_check, int *err)
{
    uint32_t length = 0, s;
    if (initial_check && *ip >= lencheck) { *err = -1; return length; }
    do {
        s = src[*ip];
        (*ip)++;
        length += s;
        if (loop_check && *ip >= lencheck) { *err = -2; return length; }
    } while (s == 255);
    return length;
}

/*
 * LL64.dec.cs:123-467 with endOnInput == true (all "safe" entry points).
 *   partial      : earlyEnd_directive.partial
 *   prefix       : bytes readable immediately before dst (lowPrefix = dst - prefix)
 *   with_prefix64: dict == withPrefix64k (the shortcut skips its lowPrefix test, :213)
 *   dict/dict_len: external dictionary (dict == usingExtDict) or NULL
 * Returns bytes written, or -(input position) - 1 on malformed input.
 * Offsets relative to dst may go negative down to -prefix.
 */
static int decode_generic(const uint8_t *src, uint8_t *dst, int src_size, int out_size,
                          int partial, int64_t prefix, int with_prefix64,
                          const uint8_t *dict, uint32_t dict_size)
{
    if (src == NULL) return -1;
    const int64_t iend = src_size;
    const int64_t oend = out_size;
    const int64_t low_prefix = -prefix;
    const int use_ext_dict = (dict != NULL);
    const int check_offset = dict_size < 65536u;                         /* :149 */
    const int64_t shortiend = iend - 14 - 2;                             /* :152 */
    const int64_t shortoend = oend - 14 - 18;                            /* :153 */
    int64_t ip = 0, op = 0, cpy, match;
    uint32_t offset, token, length;

    if (out_size == 0) {                                                 /* :162-168 */
        if (partial) return 0;
        return (src_size == 1 && src[0] == 0) ? 0 : -1;
    }
    if (src_size == 0) return -1;                                        /* :172 */

    for (;;) {
        token = src[ip++];
        length = token >> ML_BITS;

        /* :191-225 two-stage shortcut */
        if (length != RUN_MASK && ip < shortiend && op <= shortoend) {
            memmove(dst + op, src + ip, length);   /* reference copies 16, keeps `length` */
            op += length; ip += length;
            length = token & ML_MASK;
            offset = rd16(src + ip); ip += 2;
            match = op - (int64_t)offset;
            if (length != ML_MASK && offset >= 8 && (with_prefix64 || match >= low_prefix)) {
                /* Copy18 then op += length + MINMATCH; offset >= 8 => plain forward copy */
                uint32_t n = length + MINMATCH;
                for (uint32_t i = 0; i < n; i++) dst[op + i] = dst[match + i];
                op += n;
                continue;
            }
            goto copy_match;
        }

        /* :228-243 literal length */
        if (length == RUN_MASK) {
            int err = 0;
            length += read_vle(src, &ip, iend - RUN_MASK, 1, 1, &err);
            if (err == -1) goto output_error;
            /* pointer-overflow checks (:234-242) cannot fire with 64-bit offsets */
        }

        /* :246-315 copy literals */
        cpy = op + (int64_t)length;
        if (cpy > oend - MFLIMIT || ip + (int64_t)length > iend - (2 + 1 + LASTLITERALS)) {
            if (partial) {
                if (ip + (int64_t)length > iend - (2 + 1 + LASTLITERALS)
                    && ip + (int64_t)length != iend) goto output_error;
                if (cpy > oend) { cpy = oend; length = (uint32_t)(oend - op); }
            } else {
                if (ip + (int64_t)length != iend || cpy > oend) goto output_error;
            }
            memmove(dst + op, src + ip, length);
            ip += length; op += length;
            if (!partial || cpy == oend || ip == iend) break;
        } else {
            memcpy(dst + op, src + ip, length);
            ip += length; op = cpy;
        }

        /* :318-323 */
        offset = rd16(src + ip); ip += 2;
        match = op - (int64_t)offset;
        length = token & ML_MASK;

    copy_match:
        if (length == ML_MASK) {                                         /* :326-334 */
            int err = 0;
            length += read_vle(src, &ip, iend - LASTLITERALS + 1, 1, 0, &err);
            if (err != 0) goto output_error;
        }
        length += MINMATCH;

        if (check_offset && match + (int64_t)dict_size < low_prefix) goto output_error; /* :338 */

        if (use_ext_dict && match < low_prefix) {                        /* :342-378 */
            if (op + (int64_t)length > oend - LASTLITERALS) {
                if (partial) { if ((int64_t)length > oend - op) length = (uint32_t)(oend - op); }
                else goto output_error;
            }
            int64_t from_dict = low_prefix - match;
            if ((int64_t)length <= from_dict) {
                memmove(dst + op, dict + dict_size - from_dict, length);
                op += length;
            } else {
                int64_t rest = (int64_t)length - from_dict;
                memcpy(dst + op, dict + dict_size - from_dict, (size_t)from_dict);
                op += from_dict;
                for (int64_t i = 0; i < rest; i++) dst[op + i] = dst[low_prefix + i];
                op += rest;
            }
            continue;
        }

        cpy = op + (int64_t)length;
        if (partial && cpy > oend - MATCH_SAFEGUARD_DISTANCE) {          /* :387-406 */
            int64_t mlen = (int64_t)length < oend - op ? (int64_t)length : oend - op;
            for (int64_t i = 0; i < mlen; i++) dst[op + i] = dst[match + i];
            op += mlen;
            if (op == oend) break;
            continue;
        }
        /* :408-450 -- byte-serial copy is the defined result of the 8/4-byte stepping for
         * offset >= 1; for offset == 0 (hostile input) the reference's output is
         * unspecified (it replicates whatever dst held) -- we leave dst as it is. */
        if (cpy > oend - MATCH_SAFEGUARD_DISTANCE) {
            if (cpy > oend - LASTLITERALS) goto output_error;            /* :430-433 */
        }
        if (offset != 0)
            for (int64_t i = 0; i < (int64_t)length; i++) dst[op + i] = dst[match + i];
        op = cpy;
    }
    return (int)op;                                                      /* :456 */

output_error:
    return (int)(-ip) - 1;                                               /* :465 */
}

/* LL64.dec.cs:469-477 */
K4O_API int k4o_decompress_safe(const uint8_t *src, uint8_t *dst, int src_len, int dst_cap)
{
    return decode_generic(src, dst, src_len, dst_cap, 0, 0, 0, NULL, 0);
}

/* LL64.dec.cs:548-556 */
K4O_API int k4o_decompress_safe_partial(const uint8_t *src, uint8_t *dst, int src_len,
                                        int target_size, int dst_cap)
{
    uint32_t m = (uint32_t)target_size < (uint32_t)dst_cap ? (uint32_t)target_size : (uint32_t)dst_cap;
    return decode_generic(src, dst, src_len, (int)m, 1, 0, 0, NULL, 0);
}

/* LL64.dec.cs:523-546 */
K4O_API int k4o_decompress_safe_using_dict(const uint8_t *src, uint8_t *dst, int src_len,
                                           int dst_cap, const uint8_t *dict, int dict_len)
{
    if (dict_len == 0) return k4o_decompress_safe(src, dst, src_len, dst_cap);
    if (dict + dict_len == dst) {
        if (dict_len >= 65536 - 1)
            return decode_generic(src, dst, src_len, dst_cap, 0, 65536, 1, NULL, 0);
        return decode_generic(src, dst, src_len, dst_cap, 0, dict_len, 0, NULL, 0);
    }
    return decode_generic(src, dst, src_len, dst_cap, 0, 0, 0, dict, (uint32_t)dict_len);
}

/* ------------------------------------------------------------------------------------
 * LZ4Codec return mapping (LZ4Codec.cs:40-52, :104-115)
 * ---------------------------------------------------------------------------------- */
K4O_API int k4o_compress_hc(const uint8_t *src, uint8_t *dst, int src_len, int dst_cap, int level);

K4O_API int k4o_codec_encode(const uint8_t *src, int src_len, uint8_t *dst, int dst_cap, int level)
{
    if (src_len <= 0) return 0;
    int n = level < 3 ? k4o_compress_fast(src, dst, src_len, dst_cap, 1)
                      : k4o_compress_hc(src, dst, src_len, dst_cap, level);
    return n <= 0 ? -1 : n;
}

K4O_API int k4o_codec_decode(const uint8_t *src, int src_len, uint8_t *dst, int dst_cap)
{
    if (src_len <= 0) return 0;
    int n = k4o_decompress_safe(src, dst, src_len, dst_cap);
    return n <= 0 ? -1 : n;
}

/* ------------------------------------------------------------------------------------
 * LZ4Pickler envelope V0
 * ---------------------------------------------------------------------------------- */

/* pickle.cs:225-226 */
static int effective_size_of(int value) { return (value > 0xffff || value < 0) ? 4 : (value > 0xff ? 2 : 1); }

K4O_API int k4o_pickle_bound(int src_len) { return src_len <= 0 ? 0 : 1 + 4 + src_len; }

/*
 * The V0 header alone, for a block of src_len bytes whose LZ4 block came out as enc_len bytes (pickle.cs:85-105 array path,
 * :128-148 writer path; header byte :221-222, width :224-228, little-endian diff :214-219).  Writes 1..5 bytes, returns how many.
 * Pinned to the reference's own helpers over the (src_len, enc_len) plane by tests/test_ref_pins.py.
 */
K4O_API int k4o_pickle_header(int src_len, int enc_len, int writer_mode, uint8_t *dst)
{
    if (enc_len <= 0 || enc_len >= src_len) { dst[0] = 0; return 1; }   /* :85, :135, :189-194 */
    int diff = src_len - enc_len;
    int size_of_diff = writer_mode ? effective_size_of(src_len) : effective_size_of(diff);   /* :128,:161-165 / :97,:174-179 */
    int code = size_of_diff == 4 ? 3 : size_of_diff;                     /* :228 */
    dst[0] = (uint8_t)((0 & 7) | ((code & 3) << 6));                     /* :221-222 */
    for (int i = 0; i < size_of_diff; i++) dst[1 + i] = (uint8_t)((uint32_t)diff >> (8 * i));
    return 1 + size_of_diff;
}

/*
 * pickle.cs:51-106 (array/span path, writer_mode == 0) and :113-158 (IBufferWriter path,
 * writer_mode == 1: header width chosen from the source length, :129,:161-165).
 * `scratch` must hold max(src_len, 1024) bytes.  Returns envelope bytes written to dst
 * (dst must hold k4o_pickle_bound(src_len)); 0 for empty input.
 */
K4O_API int k4o_pickle(const uint8_t *src, int src_len, uint8_t *dst, uint8_t *scratch,
                       int level, int writer_mode)
{
    if (src_len <= 0) return 0;
    int cap = writer_mode ? src_len : (src_len <= 1024 ? 1024 : src_len);
    int enc = k4o_codec_encode(src, src_len, scratch, cap, level);
    if (enc <= 0 || enc >= src_len) {                                    /* :85, :135 */
        dst[0] = 0;
        memcpy(dst + 1, src, (size_t)src_len);
        return 1 + src_len;
    }
    int size_of_diff = k4o_pickle_header(src_len, enc, writer_mode, dst) - 1;
    memcpy(dst + 1 + size_of_diff, scratch, (size_t)enc);
    return 1 + size_of_diff + enc;
}

/* unpickle.cs:131-148.  Returns 0 and fills the header fields, or <0 if corrupted. */
K4O_API int k4o_unpickle_header(const uint8_t *src, int src_len, int *data_offset,
                                int *result_len, int *compressed)
{
    if (src_len <= 0) return -1;
    if ((src[0] & 7) != 0) return -2;                                    /* version */
    int code = (src[0] >> 6) & 3;
    int size_of_diff = code == 3 ? 4 : code;
    int off = 1 + size_of_diff;
    int data_len = src_len - off;
    if (data_len < 0) return -3;
    uint32_t diff = 0;
    for (int i = 0; i < size_of_diff; i++) diff |= (uint32_t)src[1 + i] << (8 * i);
    *data_offset = off;
    *result_len = (int)((uint32_t)data_len + diff);
    *compressed = diff != 0;
    return 0;
}

/*
 * unpickle.cs:100-129 (Unpickle(source, output) + UnpickleCore).
 * Returns the unpickled size (== dst_len), 0 for empty input, <0 when the reference would
 * throw InvalidDataException.
 */
K4O_API int k4o_unpickle(const uint8_t *src, int src_len, uint8_t *dst, int dst_len)
{
    int off, expect, compressed;
    if (src_len == 0) return 0;
    int rc = k4o_unpickle_header(src, src_len, &off, &expect, &compressed);
    if (rc < 0) return rc;
    if (dst_len != expect) return -4;
    if (!compressed) { memcpy(dst, src + off, (size_t)(src_len - off)); return expect; }
    int n = k4o_codec_decode(src + off, src_len - off, dst, dst_len);
    if (n != expect) return -5;
    return n;
}

/* ------------------------------------------------------------------------------------
 * Threaded batch drivers (CPU baseline for bench.py; static partition over T threads)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int op; /* 0 encode, 1 decode */
    const uint8_t *src; const uint64_t *src_off; const int32_t *src_len;
    uint8_t *dst; const uint64_t *dst_off; const int32_t *dst_cap;
    int32_t *out_len; int64_t lo, hi; int level;
} batch_job_t;

static void *batch_worker(void *arg)
{
    batch_job_t *j = (batch_job_t *)arg;
    for (int64_t i = j->lo; i < j->hi; i++) {
        const uint8_t *s = j->src + j->src_off[i];
        uint8_t *d = j->dst + j->dst_off[i];
        j->out_len[i] = j->op == 0 ? k4o_codec_encode(s, j->src_len[i], d, j->dst_cap[i], j->level)
                                   : k4o_codec_decode(s, j->src_len[i], d, j->dst_cap[i]);
    }
    return NULL;
}

static int run_batch(int op, const uint8_t *src, const uint64_t *src_off, const int32_t *src_len,
                     uint8_t *dst, const uint64_t *dst_off, const int32_t *dst_cap,
                     int32_t *out_len, int64_t n, int level, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t tid[256];
    batch_job_t jobs[256];
    for (int t = 0; t < threads; t++) {
        jobs[t] = (batch_job_t){op, src, src_off, src_len, dst, dst_off, dst_cap, out_len,
                                n * t / threads, n * (t + 1) / threads, level};
        if (threads == 1) batch_worker(&jobs[t]);
        else if (pthread_create(&tid[t], NULL, batch_worker, &jobs[t]) != 0) return -1;
    }
    if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    return 0;
}

K4O_API int k4o_encode_batch(const uint8_t *src, const uint64_t *src_off, const int32_t *src_len,
                             uint8_t *dst, const uint64_t *dst_off, const int32_t *dst_cap,
                             int32_t *out_len, int64_t n, int level, int threads)
{
    return run_batch(0, src, src_off, src_len, dst, dst_off, dst_cap, out_len, n, level, threads);
}

K4O_API int k4o_decode_batch(const uint8_t *src, const uint64_t *src_off, const int32_t *src_len,
                             uint8_t *dst, const uint64_t *dst_off, const int32_t *dst_cap,
                             int32_t *out_len, int64_t n, int threads)
{
    return run_batch(1, src, src_off, src_len, dst, dst_off, dst_cap, out_len, n, 0, threads);
}

/* Number of sequences (tokens) in a VALID block: what bench.py divides the decoder's instruction counters by.
 * Walks the token chain of LL64.dec.cs:175-336 without copying; -1 on anything that does not parse to the end. */
K4O_API int64_t k4o_count_sequences(const uint8_t *src, int src_len)
{
    const uint8_t *ip = src, *iend = src + src_len;
    int64_t n = 0;
    while (ip < iend) {
        unsigned token = *ip++;
        size_t len = token >> 4;
        n++;
        if (len == 15) { unsigned s; do { if (ip >= iend) return -1; s = *ip++; len += s; } while (s == 255); }
        if ((size_t)(iend - ip) < len) return -1;
        ip += len;
        if (ip == iend) return n;            /* last sequence: literals only */
        if (iend - ip < 2) return -1;
        ip += 2;
        if ((token & 15) == 15) { unsigned s; do { if (ip >= iend) return -1; s = *ip++; } while (s == 255); }
    }
    return -1;
}

/* Adler32 exactly as the reference's test helper computes it (src/TestHelpers/Tools.cs:29-44):
 * used to compare against the golden (length, adler32) rows of ChecksumBlockTests.cs. */
K4O_API uint32_t k4o_adler32(const uint8_t *data, int64_t len)
{
    uint32_t a = 1, b = 0;
    while (len > 0) {
        int64_t n = len < 5552 ? len : 5552;
        for (int64_t i = 0; i < n; i++) { a += data[i]; b += a; }
        a %= 65521u; b %= 65521u;
        data += n; len -= n;
    }
    return (b << 16) | a;
}
