/*
 * k4lz4_oracle_frame.c -- TEST INFRASTRUCTURE (same rules as k4lz4_oracle.c: only tests/, smoke()
 * and bench.py's cpu_baseline leg may use it).
 *
 * CPU restatement of the frame layer around the block codec:
 *   XXH32                     third-party: NuGet K4os.Hash.xxHash 1.0.8 (Streams.csproj:15), NOT under
 *                             /root/reference.  Restated from the published xxHash32 specification
 *                             (seed 0 at every call site: Frames/LZ4FrameWriter.cs:100,:162-182,
 *                             Internal/Stash.cs:149-150).  PINNED against the python `xxhash` package
 *                             (bindings of the xxHash reference implementation) in tests/test_frame_pins.py
 *                             and, through whole frames, against the system liblz4's LZ4F_* functions.
 *   LZ4EncoderBase.Encode     Encoders/LZ4EncoderBase.cs:66-88  (allowCopy: encoded >= length -> raw, -length)
 *   frame writer              Streams/Frames/LZ4FrameWriter.cs:57-108 (header), :159-189 (length code,
 *                             checksums, size code), LZ4FrameWriter.async.cs:15-27,:75-90 (block, tail)
 *   frame reader              Streams/Frames/LZ4FrameReader.async.cs:52-136
 *   LZ4BlockDecoder.Decode    Encoders/LZ4BlockDecoder.cs:39-56; chained blocks: LZ4ChainDecoder ->
 *                             LZ4_decompress_safe_continue == decode with the previous 64 KiB as prefix
 */
#include <stdint.h>
#include <string.h>

#define K4O_API __attribute__((visibility("default")))

int k4o_codec_encode(const uint8_t *src, int src_len, uint8_t *dst, int dst_cap, int level);
int k4o_compress_bound(int n);
int k4o_decompress_safe(const uint8_t *src, uint8_t *dst, int src_len, int dst_cap);
int k4o_decompress_safe_using_dict(const uint8_t *src, uint8_t *dst, int src_len, int dst_cap, const uint8_t *dict, int dict_len);

#define P1 2654435761u
#define P2 2246822519u
#define P3 3266489917u
#define P4 668265263u
#define P5 374761393u

static uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static void wr32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

/* xxHash32 (specification: stripes of 16 bytes into four lanes, then 4-byte and 1-byte tails, avalanche) */
K4O_API uint32_t k4o_xxh32(const uint8_t *p, int64_t len, uint32_t seed)
{
    const uint8_t *end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        const uint8_t *limit = end - 16;
        do {
            v1 = rotl(v1 + rd32(p) * P2, 13) * P1;
            v2 = rotl(v2 + rd32(p + 4) * P2, 13) * P1;
            v3 = rotl(v3 + rd32(p + 8) * P2, 13) * P1;
            v4 = rotl(v4 + rd32(p + 12) * P2, 13) * P1;
            p += 16;
        } while (p <= limit);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    } else {
        h = seed + P5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) { h = rotl(h + rd32(p) * P3, 17) * P4; p += 4; }
    while (p < end) { h = rotl(h + (uint32_t)(*p) * P5, 11) * P1; p++; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}

static int size_code(int block_size)   /* LZ4FrameWriter.cs:184-189 */
{
    return block_size <= (64 << 10) ? 4 : block_size <= (256 << 10) ? 5 : block_size <= (1 << 20) ? 6 : block_size <= (4 << 20) ? 7 : -1;
}
static int code_size(int code)         /* LZ4FrameReader.cs MaxBlockSize */
{
    switch (code) { case 7: return 4 << 20; case 6: return 1 << 20; case 5: return 256 << 10; case 4: return 64 << 10; default: return 64 << 10; }
}

/* worst-case frame size for `len` content bytes */
K4O_API int64_t k4o_frame_bound(int64_t len, int block_size)
{
    const int64_t nblk = (len + block_size - 1) / block_size;
    return 7 + nblk * 8 + len + 8;
}

/* One frame of independent blocks, as LZ4FrameWriter with ChainBlocks=false writes it.
 * scratch: k4o_compress_bound(block_size) bytes.  Returns the frame length or -1. */
K4O_API int64_t k4o_frame_encode(const uint8_t *src, int64_t len, uint8_t *dst, int64_t cap, int block_size, int level,
                                 int block_checksum, int content_checksum, uint8_t *scratch)
{
    const int code = size_code(block_size);
    if (code < 0 || cap < k4o_frame_bound(len, block_size)) return -1;
    uint8_t *op = dst;
    wr32(op, 0x184D2204u); op += 4;
    const int flg = (1 << 6) | (1 << 5) | ((block_checksum ? 1 : 0) << 4) | ((content_checksum ? 1 : 0) << 2);
    op[0] = (uint8_t)flg; op[1] = (uint8_t)(code << 4);
    op[2] = (uint8_t)(k4o_xxh32(op, 2, 0) >> 8);
    op += 3;
    const int bound = k4o_compress_bound(block_size);
    for (int64_t pos = 0; pos < len; pos += block_size) {
        const int n = (int)(len - pos < block_size ? len - pos : block_size);
        int enc = k4o_codec_encode(src + pos, n, scratch, bound, level);
        if (enc <= 0) return -1;                                  /* LZ4EncoderBase.cs:75-77 */
        const uint8_t *payload = scratch;
        uint32_t lencode = (uint32_t)enc;
        if (enc >= n) { payload = src + pos; enc = n; lencode = (uint32_t)n | 0x80000000u; }   /* allowCopy */
        wr32(op, lencode); op += 4;
        memcpy(op, payload, (size_t)enc); op += enc;
        if (block_checksum) { wr32(op, k4o_xxh32(payload, enc, 0)); op += 4; }
    }
    wr32(op, 0); op += 4;
    if (content_checksum) { wr32(op, k4o_xxh32(src, len, 0)); op += 4; }
    return op - dst;
}

/* Parse and decode one frame.  Returns content bytes, or: -1 bad magic, -2 version, -3 header checksum,
 * -4 truncated, -5 block checksum, -6 block does not decode, -7 content checksum, -8 content size mismatch,
 * -9 target too small, -10 unsupported (dictionary id).  *consumed = frame bytes read. */
K4O_API int64_t k4o_frame_decode(const uint8_t *src, int64_t len, uint8_t *dst, int64_t cap, int64_t *consumed)
{
    const uint8_t *ip = src, *end = src + len;
    if (len < 7) return -4;
    if (rd32(ip) != 0x184D2204u) return -1;
    ip += 4;
    const uint8_t *hdr = ip;
    const int flg = ip[0], bd = ip[1];
    ip += 2;
    if (((flg >> 6) & 0x11) != 1) return -2;                      /* LZ4FrameReader.async.cs:72-75 (mask 0x11 as written there) */
    const int chaining = ((flg >> 5) & 1) == 0, bsum = (flg >> 4) & 1, has_size = (flg >> 3) & 1, csum = (flg >> 2) & 1, has_dict = flg & 1;
    int64_t content_size = -1;
    if (has_size) { if (end - ip < 8) return -4; content_size = (int64_t)rd32(ip) | ((int64_t)rd32(ip + 4) << 32); ip += 8; }
    if (has_dict) { if (end - ip < 4) return -4; ip += 4; }
    if (end - ip < 1) return -4;
    if ((uint8_t)(k4o_xxh32(hdr, ip - hdr, 0) >> 8) != *ip) return -3;
    ip++;
    if (has_dict) return -10;
    const int block_size = code_size((bd >> 4) & 7);
    uint8_t *op = dst;
    for (;;) {
        if (end - ip < 4) return -4;
        uint32_t lc = rd32(ip); ip += 4;
        if (lc == 0) break;
        const int raw = (lc & 0x80000000u) != 0;
        const int64_t n = lc & 0x7fffffffu;
        if (end - ip < n + (bsum ? 4 : 0)) return -4;
        if (bsum && rd32(ip + n) != k4o_xxh32(ip, n, 0)) return -5;
        int64_t room = dst + cap - op;
        if (raw) {
            if (n > room) return -9;
            memcpy(op, ip, (size_t)n); op += n;
        } else {
            int want = (int)(room < block_size ? room : block_size);
            int64_t hist = op - dst;
            int d = chaining && hist > 0
                ? k4o_decompress_safe_using_dict(ip, op, (int)n, want, op - (hist < 65536 ? hist : 65536), (int)(hist < 65536 ? hist : 65536))
                : k4o_decompress_safe(ip, op, (int)n, want);
            if (d < 0) return room < block_size ? -9 : -6;
            op += d;
        }
        ip += n + (bsum ? 4 : 0);
    }
    if (csum) { if (end - ip < 4) return -4; if (rd32(ip) != k4o_xxh32(dst, op - dst, 0)) return -7; ip += 4; }
    if (content_size >= 0 && content_size != op - dst) return -8;
    if (consumed) *consumed = ip - src;
    return op - dst;
}
