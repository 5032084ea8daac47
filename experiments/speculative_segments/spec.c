// feasibility probe: does an LZ4 fast encoder (byU32, hash5) started WARM bytes before a boundary with an empty table
// fall into step with the true run by the boundary?  Build: gcc -O2 -shared -fPIC -o libspec.so spec.c
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
enum { MINMATCH = 4, LASTLITERALS = 5, MFLIMIT = 12, DISTANCE_MAX = 65535, SKIP_TRIGGER = 6 };
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t h5(uint64_t seq) { return (uint32_t)(((seq << 24) * 889523592379ull) >> (64 - 12)); }
static uint32_t clen(const uint8_t *a, const uint8_t *b, const uint8_t *lim) { const uint8_t *s = a; while (a < lim && *a == *b) { a++; b++; } return (uint32_t)(a - s); }
// runs from `start` (table as given) and calls back at every match end; stops when cb returns nonzero or at the end.
typedef int (*cb_t)(void *u, int64_t pos, uint32_t *tab);
static void run(const uint8_t *src, int64_t n, int64_t start, uint32_t *t, cb_t cb, void *u)
{
    const int64_t mfl1 = n - MFLIMIT + 1, matchlimit = n - LASTLITERALS;
    int64_t ip = start, anchor = start;
    if (start == 0) { t[h5(rd64(src))] = 0; ip = 1; }
    uint32_t fh = h5(rd64(src + ip));
    for (;;) {
        int64_t match;
        { int64_t fip = ip; int step = 1, nb = 1 << SKIP_TRIGGER;
          for (;;) { uint32_t h = fh, cur = (uint32_t)fip, mi = t[h]; ip = fip; fip += step; step = nb++ >> SKIP_TRIGGER;
                     if (fip > mfl1) return; match = mi; fh = h5(rd64(src + fip)); t[h] = cur;
                     if (mi + DISTANCE_MAX < cur) continue; if (rd32(src + match) == rd32(src + ip)) break; } }
        while (ip > anchor && match > 0 && src[ip - 1] == src[match - 1]) { ip--; match--; }
    next:
        ip += clen(src + ip + MINMATCH, src + match + MINMATCH, src + matchlimit) + MINMATCH;
        anchor = ip;
        if (ip >= mfl1) return;
        if (cb(u, ip, t)) return;
        t[h5(rd64(src + ip - 2))] = (uint32_t)(ip - 2);
        { uint32_t h = h5(rd64(src + ip)), cur = (uint32_t)ip, mi = t[h]; match = mi; t[h] = cur;
          if (mi + DISTANCE_MAX >= cur && rd32(src + match) == rd32(src + ip)) goto next; }
        fh = h5(rd64(src + ++ip));
    }
}
struct warm { int64_t B, Q; uint32_t snap[4096]; };
static int cb_warm(void *u, int64_t pos, uint32_t *t) { struct warm *w = u; if (pos >= w->B) { w->Q = pos; memcpy(w->snap, t, sizeof w->snap); return 1; } return 0; }
struct truth { struct warm *w; int nb; int *ok; int64_t *land; };
static int equiv(const uint32_t *a, const uint32_t *b, int64_t Q)
{
    for (int i = 0; i < 4096; i++) {
        const int la = (int64_t)a[i] + DISTANCE_MAX >= Q, lb = (int64_t)b[i] + DISTANCE_MAX >= Q;
        if (la != lb || (la && a[i] != b[i])) return 0;
    }
    return 1;
}
static int cb_true(void *u, int64_t pos, uint32_t *t)
{
    struct truth *T = u;
    for (int k = 0; k < T->nb; k++) {
        struct warm *w = &T->w[k];
        if (T->ok[k] == -1 && w->Q >= 0 && pos >= w->Q) { T->land[k] = pos; T->ok[k] = (pos == w->Q) ? (equiv(t, w->snap, pos) ? 1 : 2) : 0; }
    }
    return 0;
}
// boundaries B_k = seg * k (k = 1..); returns per boundary: 1 in step, 0 true run did not stop at Q, 2 stopped there with another table
int probe(const uint8_t *src, int64_t n, int64_t seg, int64_t warmup, int *ok, int64_t *Q, int64_t *land, int maxb)
{
    int nb = 0;
    struct warm *w = calloc(maxb, sizeof *w);
    uint32_t *t = malloc(4096 * 4);
    for (int64_t B = seg; B + seg / 2 < n && nb < maxb; B += seg, nb++) {
        memset(t, 0, 4096 * 4);
        w[nb].B = B; w[nb].Q = -1;
        run(src, n, B - warmup, t, cb_warm, &w[nb]);
        ok[nb] = -1; Q[nb] = w[nb].Q; land[nb] = -1;
    }
    struct truth T = { w, nb, ok, land };
    memset(t, 0, 4096 * 4);
    run(src, n, 0, t, cb_true, &T);
    free(t); free(w);
    return nb;
}
