/* tests/emu/emu_kernels.cpp -- compiles the product's kernel headers against the host wave
 * emulator and exposes them with the same batch shape as the C-ABI.  Test infrastructure only. */
#include "hip/hip_runtime.h"
#include "k4lz4_decode.hpp"
#include "k4lz4_encode_fast.hpp"
#include "k4lz4_parse.hpp"
#include "k4lz4_pickle.hpp"
#include "k4lz4_segments.hpp"
#include "k4lz4_encode_hc.hpp"
#include "k4lz4_frame.hpp"
#include "k4lz4_xxh32.hpp"
#include <vector>

extern "C" {

int k4emu_decode_batch(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                       const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n, int flags,
                       int threads)
{
    k4::BatchArgs a{src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, 0, 1, flags, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (n <= 0) return 0;
    unsigned grid = (unsigned)((n + k4::DECODE_WAVES_PER_WG - 1) / k4::DECODE_WAVES_PER_WG);
    k4emu::launch_fn(dim3(grid), dim3(64 * k4::DECODE_WAVES_PER_WG), [=] { k4::k4_decode_kernel(a); }, threads);
    return 0;
}

/* the two-waves-per-block decoder (parse wave + copy wave, LDS queue between them) */
int k4emu_decode_pair_batch(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                            const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n, int flags,
                            const uint8_t *dict, const uint64_t *dictOff, const int32_t *dictLen, int threads)
{
    k4::BatchArgs a{};
    a.src = src; a.srcOff = srcOff; a.srcLen = srcLen; a.dst = dst; a.dstOff = dstOff; a.dstCap = dstCap;
    a.outLen = outLen; a.n = n; a.accel = 1; a.flags = flags; a.dict = dict; a.dictOff = dictOff; a.dictLen = dictLen;
    if (n <= 0) return 0;
    unsigned grid = (unsigned)((n + k4::DECODE_PAIRS_PER_WG - 1) / k4::DECODE_PAIRS_PER_WG);
    k4emu::launch_fn(dim3(grid), dim3(128 * k4::DECODE_PAIRS_PER_WG), [=] { k4::k4_decode_pair_kernel(a); }, threads);
    return 0;
}

/* the pair kernel's instrumented twin: counters[32 * i ..] (parsing wave, copying wave; see k4_decode_pair_prof_kernel) */
int k4emu_decode_pair_prof_batch(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                                 const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n,
                                 unsigned long long *counters, int threads)
{
    k4::BatchArgs a{};
    a.src = src; a.srcOff = srcOff; a.srcLen = srcLen; a.dst = dst; a.dstOff = dstOff; a.dstCap = dstCap;
    a.outLen = outLen; a.n = n; a.accel = 1; a.prof = counters;
    if (n <= 0) return 0;
    unsigned grid = (unsigned)((n + k4::DECODE_PAIRS_PER_WG - 1) / k4::DECODE_PAIRS_PER_WG);
    k4emu::launch_fn(dim3(grid), dim3(128 * k4::DECODE_PAIRS_PER_WG), [=] { k4::k4_decode_pair_prof_kernel(a); }, threads);
    return 0;
}

int k4emu_decode_dict_batch(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                            const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n, int flags,
                            const uint8_t *dict, const uint64_t *dictOff, const int32_t *dictLen, int threads)
{
    k4::BatchArgs a{};
    a.src = src; a.srcOff = srcOff; a.srcLen = srcLen; a.dst = dst; a.dstOff = dstOff; a.dstCap = dstCap;
    a.outLen = outLen; a.n = n; a.accel = 1; a.flags = flags; a.dict = dict; a.dictOff = dictOff; a.dictLen = dictLen;
    if (n <= 0) return 0;
    unsigned grid = (unsigned)((n + k4::DECODE_WAVES_PER_WG - 1) / k4::DECODE_WAVES_PER_WG);
    k4emu::launch_fn(dim3(grid), dim3(64 * k4::DECODE_WAVES_PER_WG), [=] { k4::k4_decode_kernel(a); }, threads);
    return 0;
}

int k4emu_encode_batch(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                       const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n, int level,
                       int accel, int flags, int threads)
{
    k4::BatchArgs a{src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, level, accel, flags, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (n <= 0) return 0;
    k4emu::launch_fn(dim3((unsigned)((n + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)), dim3(64 * k4::ENCODE_WAVES_PER_WG),
                     [=] { k4::k4_encode_fast_kernel(a); }, threads);
    return 0;
}

/* the variant the launcher picks for half-empty batches: 28 known bytes behind every probe and candidate */
int k4emu_encode_more_batch(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                            const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n, int level,
                            int accel, int flags, int threads)
{
    k4::BatchArgs a{src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, level, accel, flags, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (n <= 0) return 0;
    k4emu::launch_fn(dim3((unsigned)((n + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)), dim3(64 * k4::ENCODE_WAVES_PER_WG),
                     [=] { k4::k4_encode_fast_more_kernel(a); }, threads);
    return 0;
}

/* the same blocks through the variant with its hash tables in (here: host) memory, 16 KiB per block */
int k4emu_encode_gtab_batch(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                            const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n, int level,
                            int accel, int flags, int threads)
{
    k4::BatchArgs a{src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, level, accel, flags, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (n <= 0) return 0;
    const unsigned grid = (unsigned)((n + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG);
    std::vector<uint32_t> tables((size_t)grid * k4::ENCODE_WAVES_PER_WG * 4096u);
    a.gtab = tables.data();
    k4emu::launch_fn(dim3(grid), dim3(64 * k4::ENCODE_WAVES_PER_WG), [=] { k4::k4_encode_fast_gtab_kernel(a); }, threads);
    return 0;
}


/* the two-kernel fast encoder (k4lz4_parse.hpp): parse -> emit, then the blocks the parse left alone.  K sub-windows per round,
 * `waves` blocks per workgroup (waves beyond PARSE_LDS_TABLES keep their table in memory), optional dispatch order. */
} /* extern C */
template <int K> static void emu_parse_launch(const k4::BatchArgs &a, const k4::ParseArgs &p, unsigned waves, int threads)
{
    k4emu::launch_fn(dim3(p.nwg), dim3(64 * waves), [=] {
        __shared__ __attribute__((aligned(16))) uint32_t lds[k4::PARSE_LDS_DWORDS];
        k4::parse_kernel_body<K>(a, p, lds);
    }, threads);
}
static void emu_parse_big_launch(const k4::BatchArgs &a, const k4::ParseArgs &p, unsigned waves, int threads)
{
    k4emu::launch_fn(dim3(p.nwg), dim3(64 * waves), [=] { k4::k4_parse_big_kernel(a, p); }, threads);
}
extern "C" {
int k4emu_encode_parse_batch(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                             const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n, int accel, int flags,
                             int K, int waves, const uint32_t *order, uint32_t *nseq_out, int threads)
{
    /* K + 16: the parsing waves write their blocks out themselves (ParseArgs::inline_emit) and k4_emit_kernel is not launched;
     * K + 32: fewer waves than blocks, every wave takes the next block of the order when it is done with one (ParseArgs::queue) */
    const bool inline_emit = (K & 16) != 0, use_queue = (K & 32) != 0, migrate = (K & 64) != 0;     /* K + 64: ParseArgs::migrate */
    const bool slot_recs = (K & 128) != 0 && inline_emit;                                             /* K + 128: ParseArgs::slot_recs */
    K &= 15;
    if (n <= 0) return 0;
    k4::BatchArgs a{};
    a.src = src; a.srcOff = srcOff; a.srcLen = srcLen; a.dst = dst; a.dstOff = dstOff; a.dstCap = dstCap; a.outLen = outLen;
    a.n = n; a.accel = accel; a.flags = flags; a.order = order;
    if (waves < 1) waves = 1;
    if (waves > k4::PARSE_MAX_WAVES) waves = k4::PARSE_MAX_WAVES;
    k4::ParseArgs p{};
    p.nwg = (uint32_t)((n + waves - 1) / waves);
    if (use_queue && p.nwg > 2) p.nwg = 2;              /* two workgroups' waves share the whole batch */
    /* filled with a mark: what lies behind a block's last record must still carry it afterwards (nothing is written that is not counted) */
    const size_t rec_slots = slot_recs ? (size_t)p.nwg * waves : (size_t)n;
    std::vector<uint2> recs(rec_slots * k4::PARSE_REC_STRIDE, uint2{0xA5A5A5A5u, 0x5A5A5A5Au});
    p.slot_recs = slot_recs ? 1u : 0u;
    std::vector<uint32_t> meta((size_t)n * 2, 0x12345678u), gtab((size_t)p.nwg * k4::PARSE_MAX_WAVES * 4096u, 0xdeadbeefu);
    p.recs = recs.data(); p.meta = meta.data(); p.gtab = gtab.data();
    p.inline_emit = inline_emit ? 1u : 0u;
    p.migrate = migrate ? 1u : 0u;
    std::vector<uint32_t> q(8, 0u);
    p.big = inline_emit ? 1u : 0u;                      /* blocks of 65 547 bytes and more: k4_parse_big_kernel behind the first launch */
    p.nbig = q.data() + 3;
    std::vector<uint32_t> ident;
    if (use_queue) {
        p.queue = q.data();
        if (p.nwg > 2) p.nwg = 2;                       /* two workgroups' waves share the whole batch */
        if (!a.order) { ident.resize((size_t)n); for (long long i = 0; i < n; i++) ident[(size_t)i] = (uint32_t)i; a.order = ident.data(); }
    }
    if (K == 1) emu_parse_launch<1>(a, p, (unsigned)waves, threads);
    else if (K == 2) emu_parse_launch<2>(a, p, (unsigned)waves, threads);
    else if (K == 3) emu_parse_launch<3>(a, p, (unsigned)waves, threads);
    else emu_parse_launch<4>(a, p, (unsigned)waves, threads);
    if (p.big) {
        k4::ParseArgs pb = p;
        if (use_queue) pb.queue = q.data() + 4;
        emu_parse_big_launch(a, pb, (unsigned)waves, threads);
    }
    if (!inline_emit) k4emu::launch_fn(dim3((unsigned)((n + k4::EMIT_WAVES_PER_WG - 1) / k4::EMIT_WAVES_PER_WG)), dim3(64 * k4::EMIT_WAVES_PER_WG), [=] { k4::k4_emit_kernel(a, p); }, threads);
    k4emu::launch_fn(dim3((unsigned)((n + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)), dim3(64 * k4::ENCODE_WAVES_PER_WG), [=] { k4::k4_encode_fast_rest_kernel(a, p); }, threads);
    if (nseq_out) for (long long i = 0; i < n; i++) nseq_out[i] = meta[(size_t)i * 2];
    if (!slot_recs)                                     /* per-block slots: exactly the counted records were written */
        for (long long i = 0; i < n; i++) {
            const uint32_t m = meta[(size_t)i * 2];
            if (m == k4::PARSE_REST || m == k4::PARSE_BIG || srcLen[i] >= k4::LIMIT_64K) continue;
            if (m > k4::PARSE_REC_STRIDE) return -2;
            for (size_t r = m; r < k4::PARSE_REC_STRIDE; r++)
                if (recs[(size_t)i * k4::PARSE_REC_STRIDE + r].x != 0xA5A5A5A5u || recs[(size_t)i * k4::PARSE_REC_STRIDE + r].y != 0x5A5A5A5Au) return -3;
        }
    return 0;
}

/* the two-kernel encoder's own cost estimate (a parse without output over the block's first bytes) and order */
int k4emu_porder(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, long long n, uint32_t *cost, uint32_t *order, int threads)
{
    k4::BatchArgs a{};
    a.src = src; a.srcOff = srcOff; a.srcLen = srcLen; a.n = n; a.accel = 1;
    std::vector<uint32_t> hist(2 * k4::PCOST_BUCKETS, 0u);
    a.cost = cost; a.hist = hist.data(); a.order_out = order;
    if (n <= 0) return 0;
    k4emu::launch_fn(dim3((unsigned)((n + k4::PCOST_WAVES_PER_WG - 1) / k4::PCOST_WAVES_PER_WG)), dim3(64 * k4::PCOST_WAVES_PER_WG), [=] { k4::k4_pcost_kernel(a); }, threads);
    k4emu::launch_fn(dim3((unsigned)((n + 255) / 256)), dim3(256), [=] { k4::k4_porder_kernel(a); }, 1);
    return 0;
}

/* dispatch-order kernels: cost estimate (dry encoder run over a sample, or by length) + bucket order */
int k4emu_order(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, long long n, int by_length,
                uint32_t *cost, uint32_t *hist, uint32_t *order, int threads)
{
    k4::BatchArgs a{};
    a.src = src; a.srcOff = srcOff; a.srcLen = srcLen; a.n = n; a.accel = 1;
    a.cost = cost; a.hist = hist; a.order_out = order;
    if (n <= 0) return 0;
    k4emu::launch_fn(dim3((unsigned)n), dim3(64), [=] { k4::k4_cost_kernel(a, by_length); }, threads);
    k4emu::launch_fn(dim3((unsigned)((n + 255) / 256)), dim3(256), [=] { k4::k4_order_kernel(a); }, 1);
    return 0;
}

/* HC: layout + chain + parse kernels, scratch on the host heap */
int k4emu_encode_hc_batch(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                          const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n, int level,
                          int flags, int threads)
{
    if (n <= 0) return 0;
    std::vector<unsigned long long> off((size_t)n + 2);
    k4::HcArgs a{src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, level, flags, nullptr, nullptr, 0u, 0u, off.data()};
    k4emu::launch_fn(dim3(1), dim3(256), [=] { k4::k4_hc_layout_kernel(a); }, 1);
    std::vector<uint32_t> hash((size_t)n << k4::HC_HASH_LOG, 0u);
    std::vector<uint8_t> work((size_t)off[(size_t)n] + 64);
    a.hash = hash.data();
    a.work = work.data();
    a.nChain = n;
    /* flags bit 29 (the emulator's own): the chains by k4_hc_chain_lds_kernel (rounds 3-5) instead of sixteen waves per block */
    if (off[(size_t)n + 1] <= 65536 && !(flags & (1 << 29))) k4emu::launch_fn(dim3((unsigned)n), dim3(64 * k4::HC_CHAIN_PARTS), [=] { k4::k4_hc_chain_part_kernel(a); }, threads);
    else if (off[(size_t)n + 1] <= 65536) k4emu::launch_fn(dim3((unsigned)((n + k4::HC_CHAIN_LDS_WAVES_PER_WG - 1) / k4::HC_CHAIN_LDS_WAVES_PER_WG)), dim3(64 * k4::HC_CHAIN_LDS_WAVES_PER_WG), [=] { k4::k4_hc_chain_lds_kernel(a); }, threads);   /* as the launcher chooses */
    else k4emu::launch_fn(dim3((unsigned)((n + k4::HC_CHAIN_WAVES_PER_WG - 1) / k4::HC_CHAIN_WAVES_PER_WG)), dim3(64 * k4::HC_CHAIN_WAVES_PER_WG), [=] { k4::k4_hc_chain_kernel(a); }, threads);
    /* flags bit 26 (the emulator's own): the candidate records by k4_hc_cand_kernel (from memory) also where every block is at most 64 KiB */
    if (off[(size_t)n + 1] >= 13 && off[(size_t)n + 1] <= 65536 && level < 10 && !(flags & (1 << 26))) {
        k4emu::launch_fn(dim3((unsigned)n), dim3(64 * k4::HC_LDS_WAVES), [=] { k4::k4_hc_walk_lds_kernel(a); }, threads);
        k4emu::launch_fn(dim3((unsigned)n), dim3(64 * k4::HC_CAND_LDS_WAVES), [=] { k4::k4_hc_cand_lds_kernel(a); }, threads);
    } else if (off[(size_t)n + 1] >= 13 && level < 10) {
        const unsigned gy = (unsigned)((off[(size_t)n + 1] + k4::HC_CAND_POS_PER_WG - 1) / k4::HC_CAND_POS_PER_WG);
        a.candChunks = gy;
        k4emu::launch_fn(dim3((unsigned)((n + 7) / 8) * 8u * gy), dim3(256), [=] { k4::k4_hc_cand_kernel(a); }, threads);
    }
    /* flags bit 30 (the emulator's own): level 3 with sequence records (HcArgs::recs), as the launcher runs blocks of at most 64 KiB */
    std::vector<uint2> recs;
    /* flags bits 27-28 (the emulator's own, with bit 30): 1 / 2 -- two / four waves per block (HcSegs) */
    const int nseg = 1 << ((flags >> 27) & 3);
    if ((flags & (1 << 30)) && level <= 3 && off[(size_t)n + 1] <= 65536) { recs.resize((size_t)n * (nseg == 4 ? k4::hc_seg_rec_off(4, 4) : nseg == 2 ? k4::hc_seg_rec_off(2, 2) : k4::PARSE_REC_STRIDE)); a.recs = recs.data(); }
    a.flags = flags & ~((1 << 30) | (1 << 29) | (3 << 27) | (1 << 26));
    if (level >= 10) k4emu::launch_fn(dim3((unsigned)n), dim3(64), [=] { k4::k4_hc_parse_opt_kernel(a); }, threads);
    else if (a.recs && nseg == 4) k4emu::launch_fn(dim3((unsigned)((n + 1) / 2)), dim3(64 * k4::HC_SEG_WAVES_PER_WG), [=] { k4::k4_hc_parse_seg4_kernel(a); }, threads);
    else if (a.recs && nseg == 2) k4emu::launch_fn(dim3((unsigned)((n + 3) / 4)), dim3(64 * k4::HC_SEG_WAVES_PER_WG), [=] { k4::k4_hc_parse_seg2_kernel(a); }, threads);
    else if (a.recs) k4emu::launch_fn(dim3((unsigned)((n + k4::HC_REC_WAVES_PER_WG - 1) / k4::HC_REC_WAVES_PER_WG)), dim3(64 * k4::HC_REC_WAVES_PER_WG), [=] { k4::k4_hc_parse_rec_kernel(a); }, threads);
    else k4emu::launch_fn(dim3((unsigned)((n + k4::HC_PARSE_WAVES_PER_WG - 1) / k4::HC_PARSE_WAVES_PER_WG)), dim3(64 * k4::HC_PARSE_WAVES_PER_WG), [=] { k4::k4_hc_parse_kernel(a); }, threads);
    return 0;
}

int k4emu_pickle_batch(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                       const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n, int level,
                       int flags, int threads)
{
    k4::BatchArgs a{src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, level, 1, flags, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (n <= 0) return 0;
    k4emu::launch_fn(dim3((unsigned)n), dim3(64), [=] { k4::k4_pickle_kernel(a); }, threads);
    return 0;
}

/* Fast-level pickles the way the launcher sends a batch that may hold big messages: slots prepared, big messages cut into
 * segments (plan on the "device"), their later segments by k4_encode_seg_kernel, everything else -- first segments included --
 * by the LDS-table kernel's segment twin, pieces joined, envelopes closed.  stats[0..2] = cut blocks, segments, blocks whose
 * every boundary verified (the others were encoded again by the join kernel). */
int k4emu_pickle_seg_batch(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                           const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n, int flags,
                           unsigned seg_min, unsigned seg_target, unsigned seg_warm, unsigned *stats, int threads)
{
    if (n <= 0) return 0;
    std::vector<uint64_t> encOff((size_t)n);
    std::vector<int32_t> encCap((size_t)n), encLen((size_t)n, 0), first((size_t)n, -1);
    k4::BatchArgs a{};
    a.src = src; a.srcOff = srcOff; a.srcLen = srcLen; a.dst = dst; a.dstOff = dstOff; a.dstCap = dstCap; a.outLen = outLen;
    a.n = n; a.accel = 1; a.flags = flags;
    uint64_t *eo = encOff.data(); int32_t *ec = encCap.data();
    k4emu::launch_fn(dim3((unsigned)((n + 255) / 256)), dim3(256), [=] { k4::k4_pickle_prep_kernel(a, eo, ec); }, 1);
    k4::BatchArgs e = a;
    e.dstOff = encOff.data(); e.dstCap = encCap.data(); e.outLen = encLen.data(); e.flags = (flags & k4::FLAG_X32) | k4::FLAG_RAW_RETURN;
    /* header and items in one piece of memory, the header SEG_HDR_DWORDS in front of the items, as k4lz4_capi.hip lays them out
     * (seg_first_of finds SegHdr::spin_max there) */
    std::vector<uint32_t> segmem((size_t)k4::SEG_HDR_DWORDS + (size_t)k4::SEG_MAX_ITEMS * k4::SEG_ITEM_WORDS, 0u);
    k4::SegHdr &hdr = *(k4::SegHdr *)segmem.data();
    k4::SegItem *items = (k4::SegItem *)(segmem.data() + k4::SEG_HDR_DWORDS);
    std::vector<uint32_t> work((size_t)k4::SEG_MAX_ITEMS), blocks((size_t)k4::SEG_MAX_BLOCKS);
    k4::SegArgs g{};
    g.hdr = &hdr; g.items = items; g.work = work.data(); g.blocks = blocks.data(); g.first = first.data();
    g.seg_min = seg_min; g.seg_target = seg_target; g.seg_warm = seg_warm; g.seg_div = 0u; g.max_items = 600u;      /* (fewer than SEG_MAX_ITEMS: the table slots then begin 600 snapshots in) */
    /* snapshots and, right behind them, one table slot per item (seg_first_of counts on that layout) */
    std::vector<uint32_t> snaps;
    snaps.assign((size_t)g.max_items * k4::SEG_SNAP_DWORDS + 64, 0u);       /* the plan only touches the flags */
    g.snaps = snaps.data();
    k4emu::launch_fn(dim3(1), dim3(256), [=] { k4::k4_seg_plan_kernel(e, g); }, 1);
    snaps.resize((size_t)g.max_items * k4::SEG_SNAP_DWORDS + (size_t)(hdr.n_items + 2u) * 4096u, 0u);
    g.snaps = snaps.data();
    g.tables = snaps.data() + (size_t)g.max_items * k4::SEG_SNAP_DWORDS;
    e.seg_first = g.first; e.seg_items = g.items; e.seg_snaps = g.snaps;
    const bool two_step = (flags & (1 << 30)) != 0;         /* the emulator's own flag: the runs by the two-step encoder (k4_parse_kernel for the blocks
                                                             * below 65 547 bytes, k4_parse_seg_kernel for the others and the later segments), round 6 */
    e.flags &= ~(1 << 30); a.flags &= ~(1 << 30);
    std::vector<uint2> recs;
    std::vector<uint32_t> meta, gtab, q(8, 0u), ident;
    if (two_step) {
        const unsigned waves = (unsigned)((flags >> 24) & 31) ? (unsigned)((flags >> 24) & 31) : 16u;      /* bits 24-28: waves per workgroup (0: sixteen) */
        e.flags &= ~(31 << 24); a.flags &= ~(31 << 24);
        k4::ParseArgs p{};
        p.nwg = 1u;                                         /* one persistent workgroup takes everything (the emulator runs workgroups one after the other) */
        recs.resize((size_t)p.nwg * waves * k4::PARSE_REC_STRIDE);
        meta.assign((size_t)n * 2, 0x12345678u); gtab.assign((size_t)p.nwg * k4::PARSE_MAX_WAVES * 4096u, 0xdeadbeefu);
        ident.resize((size_t)n); for (long long i = 0; i < n; i++) ident[(size_t)i] = (uint32_t)i;
        e.order = ident.data();
        p.recs = recs.data(); p.meta = meta.data(); p.gtab = gtab.data();
        p.inline_emit = 1u; p.slot_recs = 1u; p.migrate = 1u; p.big = 1u; p.queue = q.data(); p.nbig = q.data() + 3;
        k4emu::launch_fn(dim3(p.nwg), dim3(64 * waves), [=] { k4::k4_parse_kernel(e, p); }, 1);
        k4::ParseArgs pb = p;
        pb.queue = q.data() + 4;
        k4emu::launch_fn(dim3(p.nwg), dim3(64 * waves), [=] { k4::k4_parse_seg_kernel(e, pb, g); }, 1);
        k4emu::launch_fn(dim3((unsigned)((n + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)), dim3(64 * k4::ENCODE_WAVES_PER_WG), [=] { k4::k4_encode_fast_rest_kernel(e, p); }, threads);
    } else {
    if (hdr.n_work)
        k4emu::launch_fn(dim3((hdr.n_work + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG), dim3(64 * k4::ENCODE_WAVES_PER_WG), [=] { k4::k4_encode_seg_kernel(e, g); }, 1);
    k4emu::launch_fn(dim3((unsigned)((n + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)), dim3(64 * k4::ENCODE_WAVES_PER_WG), [=] { k4::k4_encode_fast_seg_kernel(e); }, threads);
    }
    if (stats) {
        stats[0] = hdr.n_blocks; stats[1] = hdr.n_items; stats[2] = 0; stats[3] = 0;
        for (uint32_t i = 0; i < hdr.n_items; i++) stats[3] += items[i].state == 4u && items[i].bytes > 0 ? 1u : 0u;    /* pieces that kept their output behind a boundary that did not verify */
        for (uint32_t i = 0; i < hdr.n_blocks; i++) {
            const int32_t base = first[blocks[i]];
            bool ok = true; uint32_t at = 0;
            for (uint32_t k = 0; k < items[base].nseg && ok; k++) {
                const k4::SegItem &s = items[base + k];
                ok = s.bytes > 0 && s.cut == at && (k + 1 < s.nseg ? s.state == 1u : s.state == 2u);      /* (state 4 pieces stand too, but the block is then not "joined as planned") */
                at = s.stop;
            }
            stats[2] += ok ? 1u : 0u;
        }
    }
    if (hdr.n_blocks) k4emu::launch_fn(dim3(hdr.n_blocks), dim3(64), [=] { k4::k4_seg_join_kernel(e, g); }, threads);
    if (stats) { stats[4] = hdr.n_resumed; stats[5] = hdr.n_resume_stops; stats[6] = hdr.n_plain; }
    const int32_t *el = encLen.data();
    k4emu::launch_fn(dim3((unsigned)((n + k4::PICKLE_FINISH_WAVES_PER_WG - 1) / k4::PICKLE_FINISH_WAVES_PER_WG)), dim3(64 * k4::PICKLE_FINISH_WAVES_PER_WG), [=] { k4::k4_pickle_finish_kernel(a, el); }, threads);
    return 0;
}

int k4emu_unpickle_batch(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                         const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n, int flags,
                         int threads)
{
    k4::BatchArgs a{src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, 0, 1, flags, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (n <= 0) return 0;
    unsigned grid = (unsigned)((n + k4::DECODE_WAVES_PER_WG - 1) / k4::DECODE_WAVES_PER_WG);
    k4emu::launch_fn(dim3(grid), dim3(64 * k4::DECODE_WAVES_PER_WG), [=] { k4::k4_unpickle_kernel(a); }, threads);
    return 0;
}

int k4emu_unpickle_pair_batch(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                              const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n, int flags,
                              int threads)
{
    k4::BatchArgs a{};
    a.src = src; a.srcOff = srcOff; a.srcLen = srcLen; a.dst = dst; a.dstOff = dstOff; a.dstCap = dstCap;
    a.outLen = outLen; a.n = n; a.accel = 1; a.flags = flags;
    if (n <= 0) return 0;
    unsigned grid = (unsigned)((n + k4::DECODE_PAIRS_PER_WG - 1) / k4::DECODE_PAIRS_PER_WG);
    k4emu::launch_fn(dim3(grid), dim3(128 * k4::DECODE_PAIRS_PER_WG), [=] { k4::k4_unpickle_pair_kernel(a); }, threads);
    return 0;
}

int k4emu_unpickle_sizes(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, int32_t *outLen,
                         long long n, int threads)
{
    k4::BatchArgs a{src, srcOff, srcLen, nullptr, nullptr, nullptr, outLen, n, 0, 1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (n <= 0) return 0;
    k4emu::launch_fn(dim3((unsigned)((n + 255) / 256)), dim3(256), [=] { k4::k4_unpickle_sizes_kernel(a); }, threads);
    return 0;
}

/* unit test of the per-lane run copies (k4lz4_common.hpp): lane l moves len[l] bytes src+soff[l] -> dst+doff[l];
 * mode 0 lane_copy32 (any memory), 1 lane_move32_slack (8 readable bytes past either end of the source) */
int k4emu_lane_copy(const uint8_t *src, const uint32_t *soff, uint8_t *dst, const uint32_t *doff, const uint32_t *len,
                    uint32_t src_size, int mode)
{
    k4emu::launch_fn(dim3(1), dim3(64), [=] {
        const int lane = k4::lane_id();
        if (mode == 0) k4::lane_copy32(dst + doff[lane], src + soff[lane], len[lane], src_size - soff[lane]);
        else if (len[lane]) k4::lane_move32_slack(dst + doff[lane], src + soff[lane], len[lane]);
    }, 1);
    return 0;
}

int k4emu_xxh32_batch(const uint8_t *data, const uint64_t *off, const uint64_t *len, uint32_t *out, long long n,
                      uint32_t seed, int threads)
{
    if (n <= 0) return 0;
    k4::HashArgs a{data, off, len, out, n, seed};
    k4emu::launch_fn(dim3((unsigned)((n * 4 + k4::XXH_THREADS - 1) / k4::XXH_THREADS)), dim3(k4::XXH_THREADS),
                     [=] { k4::k4_xxh32_kernel(a); }, threads);
    return 0;
}

int k4emu_allow_copy(const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                     const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, long long n, int threads)
{
    if (n <= 0) return 0;
    k4::BatchArgs a{};
    a.src = src; a.srcOff = srcOff; a.srcLen = srcLen; a.dst = dst; a.dstOff = dstOff; a.dstCap = dstCap; a.outLen = outLen; a.n = n;
    k4emu::launch_fn(dim3((unsigned)((n + 3) / 4)), dim3(256), [=] { k4::k4_allow_copy_kernel(a); }, threads);
    return 0;
}

int k4emu_decode_chain_batch(const uint8_t *src, const uint64_t *blkOff, const uint32_t *blkLen, const uint64_t *firstBlk,
                             const uint32_t *nBlk, const int32_t *blockSize, const uint8_t *chained, uint8_t *dst,
                             const uint64_t *dstOff, const uint64_t *dstCap, long long *outLen, long long n, int threads)
{
    if (n <= 0) return 0;
    k4::ChainArgs a{src, blkOff, blkLen, firstBlk, nBlk, blockSize, chained, dst, dstOff, dstCap, outLen, n};
    unsigned grid = (unsigned)((n + k4::DECODE_WAVES_PER_WG - 1) / k4::DECODE_WAVES_PER_WG);
    k4emu::launch_fn(dim3(grid), dim3(64 * k4::DECODE_WAVES_PER_WG), [=] { k4::k4_decode_chain_kernel(a); }, threads);
    return 0;
}

int k4emu_decode_chain_pair_batch(const uint8_t *src, const uint64_t *blkOff, const uint32_t *blkLen, const uint64_t *firstBlk,
                                  const uint32_t *nBlk, const int32_t *blockSize, const uint8_t *chained, uint8_t *dst,
                                  const uint64_t *dstOff, const uint64_t *dstCap, long long *outLen, long long n, int threads)
{
    if (n <= 0) return 0;
    k4::ChainArgs a{src, blkOff, blkLen, firstBlk, nBlk, blockSize, chained, dst, dstOff, dstCap, outLen, n};
    unsigned grid = (unsigned)((n + k4::DECODE_PAIRS_PER_WG - 1) / k4::DECODE_PAIRS_PER_WG);
    k4emu::launch_fn(dim3(grid), dim3(128 * k4::DECODE_PAIRS_PER_WG), [=] { k4::k4_decode_chain_pair_kernel(a); }, threads);
    return 0;
}
}
