/*
 * k4lz4_segments.hpp -- a big block by several wavefronts (fast levels).
 *
 * One wavefront encodes about 20 MB/s, so a 4 MiB message is 165 ms long whatever else the batch holds, and a ragged batch
 * (BASELINE configs[3]: LZ4Pickler over messages of 1 KiB .. 4 MiB, LZ4Pickler.pickle.cs:51-106 -> LL64.fast.cs:526-544, the
 * byU32 arm) is as slow as its biggest message.  k4lz4_encode_fast.hpp (SegRun) says why a block can be cut: a wave that
 * starts SEG_WARM bytes before a boundary with an empty table is, at its first match end behind the boundary, in the state
 * the wave before it arrives in -- and that is checked, table entry by table entry, before the two outputs are joined; a block
 * with a boundary that does not verify is encoded again by one wave (k4_seg_join_kernel), so the bytes are the reference's
 * either way.
 *
 *   k4_seg_plan_kernel     which blocks are cut (length >= seg_min, room for the pieces in their own output slot), into how
 *                          many segments; the segments' records (SegItem), the work list of the segments behind the first
 *   k4_encode_seg_kernel   one wave per such segment (hash table in memory, like the global-table kernel): warm-up, cut,
 *                          output into the block's slot at the segment's own offset (a piece never outgrows its span: a
 *                          segment that does not shrink gives up), stop at the next segment's verified cut
 *   the ordinary kernels   take the first segment of a cut block like any block, with the stop rule (their *_seg twins)
 *   k4_seg_join_kernel     per cut block: the pieces up to the first boundary that did not verify are moved down next to each
 *                          other; if that was all of them, their total is the block's length; otherwise one wave encodes the
 *                          rest from the last verified cut (its table was published there), or the whole block if there is none
 */
#pragma once
#include "k4lz4_encode_fast.hpp"
#include "k4lz4_parse.hpp"

namespace k4 {

constexpr int SEG_MAX_ITEMS = (int)SEG_ITEMS_MAX;   /* segments of all cut blocks of a launch */
constexpr int SEG_MAX_BLOCKS = 4096;         /* cut blocks of a launch */

struct SegItem {
    uint32_t block;                          /* the block this is a segment of */
    uint32_t k, nseg;
    uint32_t start, next_start, warm_from;   /* [start, next_start) is its range (next_start = SEG_NONE: to the end); the run starts at warm_from */
    uint32_t cut, stop, state;               /* SegRun's results */
    int32_t bytes;                           /* what it wrote at slot + start (at the slot's beginning for segment 0) */
};

static_assert(sizeof(SegItem) == 4u * SEG_ITEM_WORDS, "seg_first_of / seg_first_done address SegItem's fields as words 4 (next_start) and 6..9 (results)");

struct SegHdr {
    uint32_t n_items, n_work, n_blocks, spin_max;    /* spin_max: SegRun::spin_max for every run of the launch (word 3: seg_first_of reads it as such) */
    uint32_t n_resumed, n_resume_stops, n_plain;     /* k4_seg_join_kernel's runs: begun from a cut; of those, stopped at a verified boundary; whole blocks again */
    uint32_t max_items;                              /* word 7: what this launch's arrays hold (<= SEG_MAX_ITEMS); the table slots begin max_items snapshots behind `snaps` */
};

struct SegArgs {
    SegHdr *hdr;
    SegItem *items;              /* SEG_MAX_ITEMS */
    uint32_t *work;              /* SEG_MAX_ITEMS: item indices for k4_encode_seg_kernel, a block's later segments first */
    uint32_t *blocks;            /* SEG_MAX_BLOCKS: the cut blocks */
    int32_t *first;              /* per block of the batch: its segment 0's item, or -1 */
    uint32_t *snaps;             /* SEG_MAX_ITEMS x SEG_SNAP_DWORDS */
    uint32_t *tables;            /* SEG_MAX_ITEMS x 4096, one slot per ITEM, right behind `snaps` (seg_first_of counts on that): the hash table of the
                                  * wave that runs a later segment -- what it holds when the run ends is the table at the run's stop --, and for
                                  * a first segment the place where its run leaves its table when the next segment is not in step (SegRun::fix) */
    uint32_t seg_min, seg_target, seg_warm;
    uint32_t seg_target_max;     /* the segment size grows with the batch up to this (0 or <= seg_target: fixed size), see k4_seg_plan_kernel */
    uint32_t spin_max;           /* SegRun::spin_max (0: the default) */
    uint32_t max_items;          /* items the arrays of this launch hold: SEG_MAX_ITEMS, or fewer when the host knows the lengths (a batch of
                                  * thirteen big messages then takes 3 MB of snapshots and tables instead of 269) */
    uint32_t seg_div;            /* a block is cut only if it is longer than the batch's bytes / seg_div: a wave encodes ~25 MB/s, the whole chip
                                  * ~2 500 times that, so shorter blocks are over before the batch is and cutting them only adds their warm-ups */
};

/* One workgroup.  Every thread takes a contiguous range of the blocks; what a block gets (items, places in the work list, its
 * number among the cut blocks) follows from the counts of the ranges before it, so the plan does not depend on timing, and
 * when the tables are full the blocks behind simply stay whole. */
__global__ __launch_bounds__(256) void k4_seg_plan_kernel(BatchArgs a, SegArgs g)
{
    __shared__ uint32_t items_of[256], blocks_of[256];
    __shared__ unsigned long long bytes_of[256];
    const int t = (int)threadIdx.x;
    const long long per = (a.n + 255) / 256;
    const long long lo = (long long)t * per, hi = lo + per < a.n ? lo + per : a.n;
    unsigned long long mine = 0;
    for (long long b = lo; b < hi; b++) mine += a.srcLen[b] > 0 ? (unsigned long long)a.srcLen[b] : 0ull;
    bytes_of[t] = mine;
    __syncthreads();
    unsigned long long all = 0;
    for (int k = 0; k < 256; k++) all += bytes_of[k];
    const unsigned long long by_share = g.seg_div ? all / (unsigned long long)g.seg_div : 0ull;
    const uint32_t min_len = by_share > (unsigned long long)g.seg_min ? (by_share > 0x7fffffffull ? 0x7fffffffu : (uint32_t)by_share) : g.seg_min;
    /* The segment size goes by the same budget: what a wave encodes while the chip does the whole batch, less the warm-up a
     * later segment's wave runs first -- a small batch is as long as its longest serial run, so short segments (seg_target);
     * a batch that fills the chip is as long as its work, and every segment costs a warm-up's worth of that, so long ones
     * (up to seg_target_max).  Rank 0's share of configs[3], 6.3 GB: 125 ms with 768 KiB segments, 111 with 1.1 - 1.25 MiB,
     * 120 with 1.5 MiB; half of it (3.1 GB): 74 / 81 / 93 ms with 768 KiB / 1 MiB / 1.25 MiB. */
    uint32_t target = g.seg_target;
    if (g.seg_target_max > g.seg_target && by_share > (unsigned long long)g.seg_warm + (unsigned long long)g.seg_target) {
        const unsigned long long t = by_share - (unsigned long long)g.seg_warm;
        target = t < (unsigned long long)g.seg_target_max ? (uint32_t)t : g.seg_target_max;
    }
    auto segments_of = [&](long long b) -> uint32_t {
        const int U = a.srcLen[b], cap = a.dstCap[b];
        if (U < (int)min_len || U < LIMIT_64K || cap < U - 1) return 0u;
        const uint32_t nseg = ((uint32_t)U + target - 1u) / target;
        return nseg >= 2u ? nseg : 0u;
    };
    uint32_t ni = 0, nb = 0;
    for (long long b = lo; b < hi; b++) { const uint32_t k = segments_of(b); ni += k; nb += k ? 1u : 0u; }
    items_of[t] = ni; blocks_of[t] = nb;
    if (t == 0) { g.hdr->n_items = 0u; g.hdr->n_work = 0u; g.hdr->n_blocks = 0u; g.hdr->spin_max = g.spin_max; g.hdr->n_resumed = 0u; g.hdr->n_resume_stops = 0u; g.hdr->n_plain = 0u; g.hdr->max_items = g.max_items; }
    __syncthreads();
    uint32_t base = 0, bi = 0;
    for (int k = 0; k < t; k++) { base += items_of[k]; bi += blocks_of[k]; }
    for (long long b = lo; b < hi; b++) {
        const uint32_t nseg = segments_of(b);
        int32_t at = -1;
        if (nseg && base + nseg <= g.max_items && bi < (uint32_t)SEG_MAX_BLOCKS) {
            const uint32_t U = (uint32_t)a.srcLen[b];
            /* the first segment is the longer one: its wave has no warm-up to run first, so the runs of a block -- warm-up and
             * segment -- come out about equally long (the warm-up counted for at most half a segment) */
            const uint32_t w = g.seg_warm < U / (2u * nseg) ? g.seg_warm : U / (2u * nseg);
            const uint32_t later = ((U - w) / nseg) & ~63u;     /* (rounded down: the first segment takes the remainder, whatever nseg is) */
            const uint32_t first = U - (nseg - 1u) * later;
            const uint32_t w0 = base - bi;                  /* every cut block before this one took one item more than places in the work list */
            at = (int32_t)base;
            g.blocks[bi] = (uint32_t)b;
            for (uint32_t k = 0; k < nseg; k++) {
                SegItem it;
                it.block = (uint32_t)b; it.k = k; it.nseg = nseg;
                it.start = k ? first + (k - 1u) * later : 0u;
                it.next_start = k + 1u < nseg ? first + k * later : SEG_NONE;
                it.warm_from = it.start > g.seg_warm ? it.start - g.seg_warm : 0u;
                it.cut = 0u; it.stop = 0u; it.state = 3u; it.bytes = 0;
                g.items[base + k] = it;
                g.snaps[(size_t)(base + k) * SEG_SNAP_DWORDS] = 0u;
                if (k) g.work[w0 + (nseg - 1u - k)] = base + k;      /* later segments first: a wave never waits for one behind it */
            }
            atomicMax(&g.hdr->n_items, base + nseg);
            atomicMax(&g.hdr->n_work, w0 + nseg - 1u);
            atomicMax(&g.hdr->n_blocks, bi + 1u);
        }
        if (nseg) { base += nseg; bi++; }
        g.first[b] = at;
    }
}

/* the run of item `it` of a cut block: what encode_fast_block needs to know */
__device__ __forceinline__ SegRun seg_run_of(const SegArgs &g, uint32_t it, const SegItem &s)
{
    SegRun r;
    r.begin = s.k ? s.warm_from : 0u;
    r.emit_from = s.k ? s.start : 0u;
    r.stop_at = s.next_start;
    r.snap_pub = s.k ? g.snaps + (size_t)it * SEG_SNAP_DWORDS : nullptr;
    r.snap_chk = s.next_start != SEG_NONE ? g.snaps + (size_t)(it + 1u) * SEG_SNAP_DWORDS : nullptr;
    r.resume = nullptr;
    r.fix = g.tables + 4096ull * (unsigned long long)it;
    r.spin_max = g.spin_max;
    r.cut = 0u; r.stop = 0u; r.state = 3u;
    return r;
}

/* the segments behind the first: a wave each, table in memory */
__global__ __launch_bounds__(64 * ENCODE_WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(K4_SEG_WAVES_MIN, 6))) void k4_encode_seg_kernel(BatchArgs a, SegArgs g)
{
    __shared__ __attribute__((aligned(16))) uint32_t stages[ENCODE_WAVES_PER_WG][ENCODE_STAGE_DWORDS];
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t w = blockIdx.x * (uint32_t)ENCODE_WAVES_PER_WG + wave;
    const uint32_t n_work = uni(g.hdr->n_work);
    if (w >= n_work) return;
    const uint32_t it = uni(g.work[w]);
    const SegItem s = g.items[it];
    const long long b = (long long)s.block;
    const int U = a.srcLen[b];
    SegRun r = seg_run_of(g, it, s);
    /* its piece lies at `start` of the block's slot and may reach neither the next segment's place nor the end of the slot
     * (the plan accepts cap == U - 1: the last piece must not store to dst[cap]) */
    const uint32_t slot_cap = a.dstCap[b] < 0 ? 0u : (uint32_t)a.dstCap[b];
    uint32_t end = s.next_start != SEG_NONE ? s.next_start : (uint32_t)U;
    if (end > slot_cap) end = slot_cap;
    if (end <= s.start) {            /* no room for this piece: say so to the run before it, which waits for this one's cut */
        if (lane == 0) { agent_publish(r.snap_pub, SEG_NONE); g.items[it].cut = 0u; g.items[it].stop = 0u; g.items[it].state = 3u; g.items[it].bytes = 0; }
        return;
    }
    const int ret = compress_fast_block<false, false, false>(a.src + a.srcOff[b], U, a.dst + a.dstOff[b] + s.start, (int)(end - s.start), a.accel, stages[wave], lane,
                                                             g.tables + 4096ull * (unsigned long long)it, (a.flags & FLAG_X32) != 0, a.pace, &r);
    if (lane == 0) {
        g.items[it].cut = r.cut; g.items[it].stop = r.stop; g.items[it].state = ret > 0 ? r.state : 3u; g.items[it].bytes = ret;
    }
}

/*
 * Round 6: the same runs by the two-step encoder (k4lz4_parse.hpp, parse_block<.., SEG>) -- ONE persistent launch, a workgroup of sixteen
 * waves per CU (nine tables in LDS, the others in memory until one becomes free), for every block of 65 547 bytes and more of the batch:
 * the later segments of the cut blocks (the work list, a block's last segment first: a run never waits for one that has not been
 * taken), then the blocks themselves in cost order -- a cut block's first segment with the stop rule, the others whole.  The waves
 * with LDS tables take from that end, the others from the cheap end of the order.  Blocks below 65 547 bytes are k4_parse_kernel's
 * (launched before this one: it marks whose every block is).  What a run leaves behind -- SegItem's cut / stop / state / bytes, the
 * published snapshots, the table of a cut that did not verify -- is what k4_encode_seg_kernel and the *_seg twins leave, so
 * k4_seg_join_kernel joins the pieces of either.
 */
template <int TT>
__device__ __forceinline__ void parse_seg_kernel_body(const BatchArgs &a, const ParseArgs &p, const SegArgs &g, uint32_t *lds)
{
    const int lane = lane_id();
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint32_t waves = blockDim.x >> 6;
    const uint32_t lds_tables = waves < (uint32_t)PARSE_LDS_TABLES ? waves : (uint32_t)PARSE_LDS_TABLES;
    uint32_t *seen = lds + 4096u * (uint32_t)PARSE_LDS_TABLES + (uint32_t)PARSE_SEEN_DWORDS * wave;
    const bool in_lds = wave < lds_tables;
    if (p.migrate) {
        if (threadIdx.x == 0) lds[PARSE_LDS_DWORDS - 1] = 0u;
        __syncthreads();
    }
    const uint32_t n_work = uni(g.hdr->n_work);
    if (n_work == 0u && p.nbig && uni(*(volatile uint32_t *)p.nbig) == 0u) return;    /* no block of 65 547 bytes or more in this batch */
    const uint32_t total = n_work + (uint32_t)a.n;
    for (;;) {
        uint32_t t = 0u;
        if (lane == 0) {
            t = atomicAdd(p.queue, 1u);
            if (t < total) t = in_lds ? atomicAdd(p.queue + 1, 1u) : total - 1u - atomicAdd(p.queue + 2, 1u);
            else t = 0xffffffffu;
        }
        t = uni(t);
        if (t == 0xffffffffu) {
            if (p.migrate && in_lds && lane == 0) atomicOr(lds + PARSE_LDS_DWORDS - 1, 1u << wave);
            return;
        }
        if (t < n_work) {
            /* a later segment of a cut block */
            const uint32_t it = uni(g.work[t]);
            SegItem s = g.items[it];
            /* (what comes out of memory is the same in every lane; uni() says so to the compiler, which wants the block's length, the
             * cursors and everything the scalar chains take in scalar registers) */
            s.block = uni(s.block); s.k = uni(s.k); s.nseg = uni(s.nseg); s.start = uni(s.start); s.next_start = uni(s.next_start); s.warm_from = uni(s.warm_from);
            const long long b = (long long)s.block;
            const int U = (int)uni((uint32_t)a.srcLen[b]);
            SegRun r = seg_run_of(g, it, s);
            /* its piece lies at `start` of the block's slot and may reach neither the next segment's place nor the end of the slot */
            const int dcap = (int)uni((uint32_t)a.dstCap[b]);
            const uint32_t slot_cap = dcap < 0 ? 0u : (uint32_t)dcap;
            uint32_t end = s.next_start != SEG_NONE ? s.next_start : (uint32_t)U;
            if (end > slot_cap) end = slot_cap;
            if (end <= s.start) {            /* no room for this piece: say so to the run before it, which waits for this one's cut */
                if (lane == 0) { agent_publish(r.snap_pub, SEG_NONE); g.items[it].cut = 0u; g.items[it].stop = 0u; g.items[it].state = 3u; g.items[it].bytes = 0; }
                continue;
            }
            uint32_t n = 0u;
            const int ret = parse_one<1, TT, true>(a, p, lds, seen, in_lds, wave, waves, lane, b, a.src + a.srcOff[b], U, a.dst + a.dstOff[b] + s.start, (int)(end - s.start), &r, &n);
            if (lane == 0) { g.items[it].cut = r.cut; g.items[it].stop = r.stop; g.items[it].state = ret > 0 ? r.state : 3u; g.items[it].bytes = ret; }
        } else {
            const long long idx = (long long)(t - n_work);
            const long long b = a.order ? (long long)uni(a.order[idx]) : idx;
            const int src_len = (int)uni((uint32_t)a.srcLen[b]);
            if (src_len < LIMIT_64K || a.accel != 1) continue;         /* k4_parse_kernel's, or the one-kernel encoder's */
            SegFirst f = seg_first_of(a, b);     /* (neutral fields when the block is not cut: begin 0, no cut, stop at the end) */
            const int cap = (int)uni((uint32_t)a.dstCap[b]);
            const int c = cap < 0 ? 0 : (f.cut && (uint32_t)cap > f.cap ? (int)f.cap : cap);
            uint32_t n = 0u;
            const int ret = parse_one<1, TT, true>(a, p, lds, seen, in_lds, wave, waves, lane, b, a.src + a.srcOff[b], src_len, a.dst + a.dstOff[b], c, &f.run, &n);
            seg_first_done(a, b, f, ret, lane);
            if (lane == 0) {
                p.meta[2ull * (unsigned long long)b] = n;
                a.outLen[b] = codec_encode_result(src_len, ret, a.flags);
            }
        }
    }
}

__global__ __launch_bounds__(64 * PARSE_MAX_WAVES) void k4_parse_seg_kernel(BatchArgs a, ParseArgs p, SegArgs g)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[PARSE_LDS_DWORDS];
    if (a.flags & FLAG_X32) parse_seg_kernel_body<2>(a, p, g, lds);
    else parse_seg_kernel_body<0>(a, p, g, lds);
}

/* Per cut block, after every encoder kernel of the launch: join or encode again.  outLen gets what the block's encoder call
 * returns (LLxx level with FLAG_RAW_RETURN, else the LZ4Codec mapping). */
__global__ __launch_bounds__(64) void k4_seg_join_kernel(BatchArgs a, SegArgs g)
{
    __shared__ __attribute__((aligned(16))) uint32_t tab[ENCODE_LDS_DWORDS];
    const int lane = lane_id();
    const uint32_t n_blocks = uni(g.hdr->n_blocks);
    if (blockIdx.x >= n_blocks) return;
    const long long b = (long long)uni(g.blocks[blockIdx.x]);
    const int32_t base = g.first[b];
    if (base < 0) return;
    const int U = a.srcLen[b], cap = a.dstCap[b];
    const uint8_t *src = a.src + a.srcOff[b];
    uint8_t *dst = a.dst + a.dstOff[b];
    const uint32_t nseg = uni(g.items[base].nseg);
    /* Piece by piece.  A piece stands if it begins where the encoding so far ends (`at`) and stopped at a cut (state 1: the next
     * piece is in step there; state 4: it is not, and this piece left the table of the cut in ITS OWN slot of SegArgs::tables -- `fix`; the
     * next piece's snapshot stays as that piece published it) or at the
     * end of the block (state 2).  Where a piece does not stand, or the boundary behind it did not verify, ONE wave goes on from
     * the last cut with the table that lies there -- and stops at the next boundary whose piece IS in step (the same check the
     * pieces' own runs make), so that the pieces behind a bad boundary are not encoded again when they are good. */
    uint32_t out = 0, at = 0u, k = 0;
    bool whole = false, failed = false, resume = false;
    const uint32_t *resume_tab = nullptr;          /* the table at `at` for a run that goes on from there */
    while (k < nseg && !whole && !failed) {
        if (!resume) {
            const SegItem s = g.items[(uint32_t)base + k];
            const uint32_t st = uni(s.state), nb = uni((uint32_t)(s.bytes > 0 ? s.bytes : 0));
            const bool stands = nb != 0u && uni(s.cut) == at && (st == 1u || st == 2u || st == 4u) && (long long)out + nb <= (long long)cap &&
                                (k != 0u || out == 0u);
            if (stands) {
                const uint32_t start = uni(s.start);
                wave_sync();
                if (k != 0u && out != start) wave_shift_down(dst + out, dst + start, nb, lane);     /* it lies at or behind where it belongs */
                out += nb;
                at = uni(s.stop);
                if (st == 2u) { whole = true; break; }
                resume = st == 4u;            /* the next piece is not in step: the table at `at` is in this piece's slot */
                resume_tab = g.tables + 4096ull * (unsigned long long)((uint32_t)base + k);
                k++;
                continue;
            }
            if (k == 0u) { failed = true; break; }        /* nothing stands: the plain way */
            resume = true;                                /* the run before verified piece k's published table at `at`: go on from it */
            resume_tab = g.snaps + (size_t)((uint32_t)base + k) * SEG_SNAP_DWORDS + 16u;
        }
        /* one wave from `at` through piece k's range, behind the pieces that stand */
        wave_sync();
        const bool has_next = k + 1u < nseg;
        const uint32_t room_end = has_next ? uni(g.items[(uint32_t)base + k + 1u].start) : (uint32_t)(cap < 0 ? 0 : cap);   /* not into the next piece's place */
        SegRun r;
        r.begin = at; r.emit_from = 0u; r.snap_pub = nullptr;
        r.stop_at = has_next ? uni(g.items[(uint32_t)base + k].next_start) : SEG_NONE;
        r.snap_chk = has_next ? g.snaps + (size_t)((uint32_t)base + k + 1u) * SEG_SNAP_DWORDS : nullptr;
        r.resume = resume_tab;
        r.fix = g.tables + 4096ull * (unsigned long long)((uint32_t)base + k);      /* piece k does not stand: its slot is free */
        r.spin_max = 1u;                 /* every piece's run is over by now: what it published is there, what is not never comes */
        r.cut = 0u; r.stop = 0u; r.state = 3u;
        int more = 0;
        if (room_end > out && (long long)room_end <= (long long)cap)
            more = compress_fast_block<true, false>(src, U, dst + out, (int)(room_end - out), a.accel, tab, lane, nullptr, (a.flags & FLAG_X32) != 0, nullptr, &r);
        if (lane == 0) { atomicAdd(&g.hdr->n_resumed, 1u); if (more > 0 && r.state == 1u) atomicAdd(&g.hdr->n_resume_stops, 1u); }
        if (more <= 0 || r.state == 3u) { failed = true; break; }
        out += (uint32_t)more;
        at = r.stop;
        if (r.state == 2u) { whole = true; break; }
        resume = r.state == 4u;
        resume_tab = r.fix;
        k++;
    }
    int ret;
    if (whole && !failed) {
        ret = (int)out;
    } else {
        wave_sync();
        if (lane == 0) atomicAdd(&g.hdr->n_plain, 1u);
        ret = compress_fast_block<true, false>(src, U, dst, cap < 0 ? 0 : cap, a.accel, tab, lane, nullptr, (a.flags & FLAG_X32) != 0);
    }
    if (lane == 0) a.outLen[b] = codec_encode_result(U, ret, a.flags);
}

}  // namespace k4
