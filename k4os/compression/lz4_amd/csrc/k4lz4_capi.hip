/*
 * k4lz4_capi.hip -- host side of libk4lz4.so (see include/k4lz4.h for the contract).
 *
 * The only compute path is the gfx950 kernels in this directory; there is no CPU fallback.
 * Host-pointer batch calls stage through device buffers owned by the context:
 *   H2D  the source span + the four metadata vectors
 *   run  one kernel launch per batch (one wavefront per block)
 *   D2H  outLen, then the produced bytes (compact device layout -> pinned staging -> the
 *        caller's slots, copying exactly outLen[i] bytes so dst[ret..cap) stays untouched,
 *        SpanTests.cs:36-44).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <memory>
#include <new>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include "../../../../include/k4lz4.h"
#include "k4lz4_decode.hpp"
#include "k4lz4_encode_fast.hpp"
#include "k4lz4_parse.hpp"
#include "k4lz4_pickle.hpp"
#include "k4lz4_segments.hpp"
#include "k4lz4_encode_hc.hpp"
#include "k4lz4_frame.hpp"
#include "k4lz4_xxh32.hpp"
#include "k4lz4_selftest.hpp"

constexpr int MAX_PARTS = 8;      /* a big host-pointer call is staged, run and brought back in up to this many parts */

struct k4lz4_ctx {
    int device = -1;
    std::string error;
    hipStream_t stream = nullptr;   /* used by the host-pointer calls */
    int accel = 1;                  /* fast-encoder acceleration of the next launch (LLxx-level calls only) */
    unsigned long long *prof = nullptr;   /* diagnostic counters of the next launch (k4lz4_profile_batch_device) */
    bool prof_pair = false;               /* ... of the two-waves-per-block decoder (32 counters per block) */
    bool prof_stamp = false;              /* ... or only start / end / placement of every block, written by the ordinary kernels */
    /* grow-only device / pinned scratch for the host-pointer calls */
    uint8_t *d_src = nullptr; size_t d_src_cap = 0;
    uint8_t *d_dst = nullptr; size_t d_dst_cap = 0;
    uint8_t *d_meta = nullptr; size_t d_meta_cap = 0;
    uint8_t *h_stage = nullptr; size_t h_stage_cap = 0;
    uint8_t *d_sched = nullptr; size_t d_sched_cap = 0;   /* dispatch-order scratch: cost[n], order[n], counters */
    int cu_count = 256;
    uint8_t *d_dict = nullptr; size_t d_dict_cap = 0;         /* host-pointer decode with dictionaries: staged dictionaries + their metadata */
    uint8_t *d_gtab = nullptr; size_t d_gtab_cap = 0;         /* fast encoder: hash tables of the blocks encoded without an LDS table */
    hipStream_t aux = nullptr;                                /* second queue: those blocks run beside the LDS-table kernel */
    hipStream_t aux2 = nullptr;                               /* third queue: the later segments of blocks cut into segments (k4lz4_segments.hpp) */
    hipEvent_t ev_join2 = nullptr;
    uint8_t *d_seg = nullptr; size_t d_seg_cap = 0;           /* segment records, work list, snapshots, tables */
    uint8_t *d_seg_first = nullptr; size_t d_seg_first_cap = 0;   /* per block: its first segment's record or -1 */
    bool use_segments = true;                                 /* K4LZ4_NO_SEGMENTS */
    uint32_t seg_min = 1024u << 10, seg_target = 640u << 10, seg_warm = 384u << 10;   /* K4LZ4_SEG_MIN / _TARGET / _WARM (bytes) */
    uint32_t seg_target_max = 1408u << 10;                    /* K4LZ4_SEG_TARGET_MAX; K4LZ4_SEG_TARGET alone fixes the size (1152 KiB until round 6: the two-step encoder's runs are faster, gpurun_out/r6o) */
    uint32_t seg_spin_max = 0;                                /* K4LZ4_SEG_SPIN_MAX: polls a run waits for its successor's cut (0: SEG_SPIN_MAX); tests force the exit with 1 */
    uint32_t seg_div = 3500;                                  /* K4LZ4_SEG_DIV: blocks shorter than the batch's bytes / this stay whole */
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    uint8_t *d_hc_hash = nullptr; size_t d_hc_hash_cap = 0;   /* HC: per-block hash tables of one launch chunk */
    uint8_t *d_hc_work = nullptr; size_t d_hc_work_cap = 0;   /* HC: prev[] / cand[] of one launch chunk */
    uint8_t *d_hc_meta = nullptr; size_t d_hc_meta_cap = 0;   /* HC: work offsets, pickle slots */
    uint8_t *d_pk_meta = nullptr; size_t d_pk_meta_cap = 0;   /* fast-level pickles through the encoder kernels: encoder slots and results */
    int pickle_split_min = 512;           /* K4LZ4_PICKLE_SPLIT_MIN: batches of more messages than this go that way */
    /* The scratch above is shared by all calls on this context but ordered only by the stream a call runs on: the end of
     * every launch is recorded here and the next launch on a DIFFERENT stream waits for it before it touches the scratch
     * (calls on one context are serialised across streams). */
    hipEvent_t ev_busy = nullptr;
    hipStream_t last_stream = nullptr;
    bool busy = false;
    /* k4lz4_ctx_reserve_hc: HC levels sized without asking the device (0 = not reserved) */
    uint64_t hc_res_total = 0; uint32_t hc_res_longest = 0;
    /* host-pointer calls of some size stage through two pinned buffers per direction, filled / emptied by a few threads while
     * the DMA engine moves the previous chunk; d_pack takes the produced bytes packed next to each other */
    uint8_t *h_in[2] = {nullptr, nullptr}; size_t h_in_cap[2] = {0, 0};
    uint8_t *h_out[2] = {nullptr, nullptr}; size_t h_out_cap[2] = {0, 0};
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    uint8_t *d_pack = nullptr; size_t d_pack_cap = 0;
    /* big host-pointer calls run as two halves: the second half's bytes go up (copy queue) while the first half's kernels
     * run, the first half's results come down while the second half's kernels run; results' sizes via a pinned array */
    hipStream_t copyq = nullptr;
    hipStream_t dlq = nullptr;          /* ... and the way back has a queue (and a thread, and helper threads) of its own: up and down at once */
    hipEvent_t ev_up[MAX_PARTS] = {}, ev_len[MAX_PARTS] = {};
    uint8_t *h_len = nullptr; size_t h_len_cap = 0;
    struct Pool *pool = nullptr, *pool_dl = nullptr;   /* helper threads of the upload side, of the download side */
    bool pool_failed = false, pool_dl_failed = false;
    int stage_threads = 7;      /* helper threads of the staging copies (K4LZ4_STAGE_THREADS - 1, read once at creation) */
    /* diagnostic switches, read once at creation: K4LZ4_SPLIT_PCT (1..100, share of an encode batch on the LDS-table kernel),
     * K4LZ4_NO_PAIR (decode with one wave per block) */
    uint32_t *d_pace = nullptr;     /* the per-SIMD slots of the late-blocks-first priorities (k4lz4_common.hpp, Pace) */
    uint32_t *d_status = nullptr;   /* this context's status word: DEV_STATUS_* bits raised by its kernels (k4lz4_common.hpp) */
    int split_pct = -1;
    bool no_pair = false;
    int direct_span_pct = 200;            /* K4LZ4_DIRECT_SPAN_PCT: a registered source goes up as it lies while its span is at most this share of its blocks' bytes */
    bool no_direct = false;               /* K4LZ4_NO_DIRECT: registered host memory is staged like any other */
    int lds_floor_per_cu = 8;             /* K4LZ4_LDS_FLOOR: blocks per CU the LDS-table kernel gets at least */
    int cost_pct = 48;                    /* K4LZ4_COST_PCT: the LDS-table kernel's share of a batch's estimated cost (k4_order_kernel) */
    int dec_parts = 4, dec_parts_direct = 8;   /* K4LZ4_DEC_PARTS, K4LZ4_DEC_PARTS_DIRECT: parts of a big decode-like host-pointer call (2..MAX_PARTS); with a registered destination */
    int hop2_max_per_cu = 12;             /* K4LZ4_HOP2_MAX: pair decoders follow the token chain two links per hop in launches of up to this many blocks per CU */
    bool hc_cand_lds = true;              /* K4LZ4_HC_CAND_MEM unsets it: blocks of at most 64 KiB get their candidate records by k4_hc_cand_kernel (from memory) */
    int hc_cand_dynlds = 0;               /* K4LZ4_HC_CAND_DYNLDS (a measurement switch): bytes of unused LDS per workgroup of k4_hc_cand_kernel, i.e. fewer of them per CU */
    int hc_segs = 0;                      /* K4LZ4_HC_SEGS = 1 / 2 / 4: waves per block of the level-3 parse (0: by the batch's size) */
    bool hc_chain_parts = true;           /* K4LZ4_HC_CHAIN_OLD unsets it: blocks of at most 64 KiB build their chains with sixteen waves per block (k4_hc_chain_part_kernel) */
    int hc_mem_pct = 33;                  /* K4LZ4_HC_MEM_PCT: share of an HC chunk whose chains are built with the table in memory, beside the LDS-table kernel */
    int pace_min_per_cu = 6;              /* K4LZ4_PACE_MIN: batches of more blocks per CU than this use the priorities (measured: 8 per CU +3 % encode, +7 % decode; 4 per CU -1 %, -4 %) */
    bool use_pace = true;                 /* K4LZ4_NO_PACE: without the late-blocks-first priorities */
    bool prof_gtab = false;               /* K4LZ4_PROF_GTAB: the instrumented encoder keeps its table in global memory */
    uint8_t *d_parse = nullptr; size_t d_parse_cap = 0;       /* two-kernel fast encoder (k4lz4_parse.hpp): records, per-block counts, tables of the waves without an LDS table */
    bool use_parse = true;                /* K4LZ4_NO_PARSE: fast-level batches go to the one-kernel encoders as before */
    bool parse_queue = false;             /* K4LZ4_PARSE_QUEUE */
    bool hc_records = true;               /* K4LZ4_NO_HC_RECORDS: level 3 writes its sequences out inside the parse loop (rounds 1-5) */
    bool parse_seg = true;                /* K4LZ4_NO_PARSE_SEG: ragged batches (pickles, K4LZ4_FLAG_SEGMENTS) through the one-kernel encoders and their *_seg twins (rounds 3-5) */
    bool parse_big = true;                /* K4LZ4_NO_PARSE_BIG: blocks of 65 547 bytes and more stay with the one-kernel encoder (round 5) */
    bool parse_persist = true;            /* K4LZ4_NO_PERSIST: batches beyond one residency in launches of one residency each (round 5) instead of one persistent launch */
    bool parse_inline_emit = true;        /* K4LZ4_NO_INLINE_EMIT: the blocks' bytes by k4_emit_kernel behind the parse instead of by the parsing waves themselves */
    bool parse_migrate = true;            /* K4LZ4_NO_MIGRATE: blocks whose table lives in memory stay there (k4lz4_parse.hpp, ParseCtl) */
    bool parse_pcost = false;             /* K4LZ4_PCOST: the parse's own cost estimate (k4_pcost_kernel) orders the blocks; measured: costs more than it gains */
    int parse_waves = 16;                 /* K4LZ4_PARSE_WAVES: blocks per workgroup (= per CU) of the parse kernel, at most PARSE_MAX_WAVES */
    bool trace = false;         /* K4LZ4_TRACE: host-pointer calls print where their time went (stderr) */
};

/* a few helper threads for the staging copies of big host-pointer calls (memcpy between the caller's pageable memory and
 * the pinned buffers is what limits such calls, not PCIe) */
struct Pool {
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::function<void(int)> job;
    int parts = 0, next = 0, pending = 0;
    unsigned long long gen = 0;
    bool stop = false;
    explicit Pool(int nthreads)
    {
        for (int t = 0; t < nthreads; t++) workers.emplace_back([this] { run(); });
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> g(m); stop = true; }
        cv_work.notify_all();
        for (auto &w : workers) w.join();
    }
    void run()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_work.wait(lk, [this] { return stop || next < parts; });
            if (stop) return;
            const int i = next++;
            lk.unlock();
            job(i);
            lk.lock();
            if (--pending == 0) cv_done.notify_all();
        }
    }
    /* f(0) .. f(parts - 1), the caller takes part as well */
    void parallel(int nparts, const std::function<void(int)> &f)
    {
        if (nparts <= 1 || workers.empty()) { for (int i = 0; i < nparts; i++) f(i); return; }
        std::unique_lock<std::mutex> lk(m);
        job = f; parts = nparts; next = 0; pending = nparts; gen++;
        cv_work.notify_all();
        while (next < parts) {
            const int i = next++;
            lk.unlock();
            f(i);
            lk.lock();
            if (--pending == 0) cv_done.notify_all();
        }
        cv_done.wait(lk, [this] { return pending == 0; });
        parts = 0;
    }
};

namespace {

thread_local std::string tl_error;
thread_local int tl_status = K4LZ4_OK;

std::mutex g_err_mu;   /* a host-pointer call's download thread and its staging thread may both fail at the same moment */
int fail(k4lz4_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) { std::lock_guard<std::mutex> g(g_err_mu); ctx->error = msg; }
    tl_error = msg;
    return code;
}

int hip_fail(k4lz4_ctx *ctx, hipError_t e, const char *what)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return fail(ctx, K4LZ4_E_HIP, buf);
}

#define K4_HIP(ctx, call)                                                   \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call);              \
    } while (0)

enum Kind { KIND_ENCODE, KIND_DECODE, KIND_PICKLE, KIND_UNPICKLE };
constexpr int FLAG_SEGMENTS_OK = 1 << 20;   /* launch_inner, fast encode: big blocks may be cut into segments (k4lz4_segments.hpp); not part of the API */

/* LL.Enforce32 (Engine/LL.tools.cs:19-27): a process-wide switch in the reference, so here too */
std::atomic<int> g_enforce32{0};

int check_level(k4lz4_ctx *ctx, int level)
{
    (void)ctx; (void)level;   /* < L03_HC -> fast (LZ4Codec.cs:48); L03..L09 hash chain; L10..L12 optimal parser; above: as L12 (LL64.high.cs:1160) */
    return K4LZ4_OK;
}

int grow(k4lz4_ctx *ctx, uint8_t **p, size_t *cap, size_t need, bool pinned);

/* After a synchronisation of the call's stream: did a kernel launched through THIS context report call-level trouble?
 * Reads and clears the context's own word -- only this context's kernels write it and they have finished, so nothing can
 * land between the read and the clear -- and reports every bit that is set. */
int take_device_status(k4lz4_ctx *ctx)
{
    uint32_t v = 0;
    if (!ctx->d_status) return K4LZ4_OK;
    if (hipMemcpy(&v, ctx->d_status, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return K4LZ4_OK; }
    if (v == 0) return K4LZ4_OK;
    (void)hipMemset(ctx->d_status, 0, sizeof v);
    const bool nomem = (v & k4::DEV_STATUS_HC_SCRATCH) != 0, timeout = (v & k4::DEV_STATUS_PIPE_TIMEOUT) != 0;
    std::string msg;
    if (nomem) msg += "HC scratch reserved with k4lz4_ctx_reserve_hc was too small for the batch: its blocks were not encoded";
    if (timeout) msg += std::string(nomem ? "; " : "") + "a wave gave up waiting for its partner wave (a decoder pair, or the waves that parse one block at HC level 3: scheduling time-out, not corrupt data): the affected blocks report failure";
    return fail(ctx, nomem ? K4LZ4_E_NOMEM : K4LZ4_E_HIP, msg.c_str());
}

/* HC levels: layout -> (sync for the scratch size) -> hash-table clear -> chain kernel -> parse kernel.
 * `pickle`: the same around the LZ4Pickler envelope (encoder slot = envelope + 5, cap U - 1). */
/* hostLen: the blocks' lengths when the caller has them on the host (host-pointer calls), else nullptr */
int launch_hc(k4lz4_ctx *ctx, bool pickle, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
              const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n, int level, int flags,
              hipStream_t stream, const int32_t *hostLen)
{
    const int64_t chunk_max = 4096;   /* 128 KiB of hash table per block in flight */
    for (int64_t first = 0; first < n; first += chunk_max) {
        const int64_t cnt = std::min<int64_t>(chunk_max, n - first);
        /* scratch that has to grow is freed first: what still runs on it must be finished (same-stream reuse needs no wait) */
        const size_t need_hash = (size_t)cnt << (k4::HC_HASH_LOG + 2), need_meta = (size_t)(cnt + 2) * 8 + (size_t)cnt * 16 + 64;
        if (need_hash > ctx->d_hc_hash_cap || need_meta > ctx->d_hc_meta_cap) K4_HIP(ctx, hipStreamSynchronize(stream));
        int rc = grow(ctx, &ctx->d_hc_hash, &ctx->d_hc_hash_cap, need_hash, false);
        if (rc != K4LZ4_OK) return rc;
        rc = grow(ctx, &ctx->d_hc_meta, &ctx->d_hc_meta_cap, need_meta, false);
        if (rc != K4LZ4_OK) return rc;
        unsigned long long *d_woff = (unsigned long long *)ctx->d_hc_meta;
        uint64_t *d_encoff = (uint64_t *)(d_woff + cnt + 2);
        int32_t *d_enccap = (int32_t *)(d_encoff + cnt);
        int32_t *d_enclen = d_enccap + cnt;
        k4::HcArgs h{};
        h.status = ctx->d_status;
        h.src = src; h.srcOff = srcOff + first; h.srcLen = srcLen + first;
        h.dst = dst; h.dstOff = dstOff + first; h.dstCap = dstCap + first; h.outLen = outLen + first;
        h.n = cnt; h.level = level; h.flags = flags;
        h.hash = (uint32_t *)ctx->d_hc_hash; h.workOff = d_woff;
        k4::BatchArgs a{};
        a.src = src; a.srcOff = h.srcOff; a.srcLen = h.srcLen; a.dst = dst; a.dstOff = h.dstOff; a.dstCap = h.dstCap;
        a.outLen = h.outLen; a.n = cnt; a.level = level; a.accel = 1; a.flags = flags;
        if (pickle) {
            hipLaunchKernelGGL(k4::k4_pickle_prep_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, a, d_encoff, d_enccap);
            h.dstOff = d_encoff; h.dstCap = d_enccap; h.outLen = d_enclen; h.flags = K4LZ4_FLAG_RAW_RETURN;
        }
        /* How much work area, and how long the longest block is: from the host's copy of the lengths, from the
         * reservation (k4lz4_ctx_reserve_hc), or -- the only case that synchronises -- from the device. */
        unsigned long long tail[2] = {0, 0};   /* total work bytes, longest block */
        bool ask_device = false;
        if (hostLen) {
            for (int64_t i = 0; i < cnt; i++) {
                const int32_t len = hostLen[first + i];
                if (len > 0) {
                    tail[0] += (((unsigned long long)len + 3u) & ~3ull) * k4::HC_WORK_PER_BYTE;
                    tail[1] = std::max<unsigned long long>(tail[1], (unsigned long long)len);
                }
            }
        } else if (ctx->hc_res_total) {
            tail[0] = (ctx->hc_res_total + 4u * (unsigned long long)cnt) * k4::HC_WORK_PER_BYTE;
            tail[1] = ctx->hc_res_longest;
            h.workCap = tail[0]; h.maxLen = ctx->hc_res_longest;
        } else {
            ask_device = true;
        }
        if (!ask_device && (size_t)tail[0] + 256 > ctx->d_hc_work_cap) K4_HIP(ctx, hipStreamSynchronize(stream));
        if (!ask_device) {
            rc = grow(ctx, &ctx->d_hc_work, &ctx->d_hc_work_cap, (size_t)tail[0] + 256, false);
            if (rc != K4LZ4_OK) return rc;
        }
        hipLaunchKernelGGL(k4::k4_hc_layout_kernel, dim3(1), dim3(256), 0, stream, h);
        if (ask_device) {
            K4_HIP(ctx, hipMemcpyAsync(tail, d_woff + cnt, 16, hipMemcpyDeviceToHost, stream));
            K4_HIP(ctx, hipStreamSynchronize(stream));
            rc = grow(ctx, &ctx->d_hc_work, &ctx->d_hc_work_cap, (size_t)tail[0] + 256, false);
            if (rc != K4LZ4_OK) return rc;
        }
        h.work = ctx->d_hc_work;
        if (tail[1] <= 65536 && ctx->hc_chain_parts) {
            /* no block over 64 KiB: eight waves per block, each with an eighth of the hash values and of the table (round 6) */
            h.nChain = cnt;
            hipLaunchKernelGGL(k4::k4_hc_chain_part_kernel, dim3((unsigned)cnt), dim3(64 * k4::HC_CHAIN_PARTS), 0, stream, h);
        } else if (tail[1] <= 65536) {
            /* (rounds 3-5, K4LZ4_HC_CHAIN_OLD) no block over 64 KiB (known from the host lengths, the reservation or the device): hash tables in LDS -- two blocks
             * per CU, a wave each, which leaves the CU's other wave slots empty; so the last third of a big chunk goes through the
             * table-in-memory kernel on the second queue at the same time (its tables: 128 KiB per block, cleared here) */
            const int64_t n_mem = cnt >= 8 * (int64_t)ctx->cu_count ? cnt * ctx->hc_mem_pct / 100 : 0;
            const int64_t n_lds = cnt - n_mem;
            if (n_mem > 0) {
                K4_HIP(ctx, hipEventRecord(ctx->ev_fork, stream));
                K4_HIP(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
                K4_HIP(ctx, hipMemsetAsync(ctx->d_hc_hash + ((size_t)n_lds << (k4::HC_HASH_LOG + 2)), 0, (size_t)n_mem << (k4::HC_HASH_LOG + 2), ctx->aux));
                k4::HcArgs hm = h;
                hm.blockBase = (unsigned)n_lds; hm.nChain = n_mem;
                hipLaunchKernelGGL(k4::k4_hc_chain_kernel, dim3((unsigned)((n_mem + k4::HC_CHAIN_WAVES_PER_WG - 1) / k4::HC_CHAIN_WAVES_PER_WG)), dim3(64 * k4::HC_CHAIN_WAVES_PER_WG), 0, ctx->aux, hm);
                K4_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
            }
            h.nChain = n_lds;
            hipLaunchKernelGGL(k4::k4_hc_chain_lds_kernel, dim3((unsigned)((n_lds + k4::HC_CHAIN_LDS_WAVES_PER_WG - 1) / k4::HC_CHAIN_LDS_WAVES_PER_WG)), dim3(64 * k4::HC_CHAIN_LDS_WAVES_PER_WG), 0, stream, h);
            if (n_mem > 0) K4_HIP(ctx, hipStreamWaitEvent(stream, ctx->ev_join, 0));
        } else {
            K4_HIP(ctx, hipMemsetAsync(ctx->d_hc_hash, 0, (size_t)cnt << (k4::HC_HASH_LOG + 2), stream));
            h.nChain = cnt;
            hipLaunchKernelGGL(k4::k4_hc_chain_kernel, dim3((unsigned)((cnt + k4::HC_CHAIN_WAVES_PER_WG - 1) / k4::HC_CHAIN_WAVES_PER_WG)), dim3(64 * k4::HC_CHAIN_WAVES_PER_WG), 0, stream, h);
        }
        const bool optimal = level >= K4LZ4_L10_OPT;              /* clTable (LL64.high.cs:1124-1138): lz4opt strategy */
        if (tail[1] >= 13 && tail[1] <= 65536 && !optimal && ctx->hc_cand_lds) {
            /* no block over 64 KiB: the candidates' records out of LDS (k4_hc_walk_lds_kernel, k4_hc_cand_lds_kernel; round 6) */
            hipLaunchKernelGGL(k4::k4_hc_walk_lds_kernel, dim3((unsigned)cnt), dim3(64 * k4::HC_LDS_WAVES), 0, stream, h);
            hipLaunchKernelGGL(k4::k4_hc_cand_lds_kernel, dim3((unsigned)cnt), dim3(64 * k4::HC_CAND_LDS_WAVES), 0, stream, h);
        } else if (tail[1] >= 13 && !optimal) {
            const unsigned gy = (unsigned)((tail[1] + k4::HC_CAND_POS_PER_WG - 1) / k4::HC_CAND_POS_PER_WG);
            const unsigned groups = (unsigned)((cnt + 7) / 8);                     /* eight blocks, one per XCD (k4_hc_cand_kernel) */
            const unsigned per = std::max(1u, (1u << 30) / (8u * groups));         /* chunks per launch: the grid stays below 2^31 workgroups */
            for (unsigned y0 = 0; y0 < gy; y0 += per) {
                k4::HcArgs hy = h;
                hy.posBase = y0 * (unsigned)k4::HC_CAND_POS_PER_WG;
                hy.candChunks = std::min(per, gy - y0);
                hipLaunchKernelGGL(k4::k4_hc_cand_kernel, dim3(groups * 8u * hy.candChunks), dim3(256), (size_t)ctx->hc_cand_dynlds, stream, hy);
            }
        }
        const hipStream_t pstream = stream;
        if (optimal) hipLaunchKernelGGL(k4::k4_hc_parse_opt_kernel, dim3((unsigned)cnt), dim3(64), 0, stream, h);
        else {
            /* level 3 on blocks of at most 64 KiB: the parse writes 8-byte sequence records (the fast encoder's scratch, a slot per
             * block) and the same wave turns them into bytes afterwards; without room for them LZ4HC_encodeSequence stays in the loop */
            /* ... by two or four waves per block while the chip has the wave slots for them (k4lz4_encode_hc.hpp, HcSegs; round 6) */
            int nseg = 1;
            if (level <= K4LZ4_L03_HC && tail[1] <= 65536 && ctx->hc_records) {
                const int64_t slots = 32 * (int64_t)ctx->cu_count;
                nseg = ctx->hc_segs > 0 ? ctx->hc_segs : (cnt * 2 <= slots ? 4 : 1);      /* (measured, profiles/r6_hc_ab.txt: four waves per block beat two even when they take turns -- 4096 blocks: 9.5 against 12.2-13.3 ms, one wave 13.9) */
                if (tail[1] < k4::HC_SEG_MIN_LEN) nseg = 1;
                for (;;) {
                    const size_t need = (size_t)cnt * (nseg == 4 ? k4::hc_seg_rec_off(4, 4) : nseg == 2 ? k4::hc_seg_rec_off(2, 2) : k4::PARSE_REC_STRIDE) * sizeof(uint2);
                    if (need > ctx->d_parse_cap) K4_HIP(ctx, hipStreamSynchronize(stream));
                    if (grow(ctx, &ctx->d_parse, &ctx->d_parse_cap, need, false) == K4LZ4_OK) { h.recs = (uint2 *)ctx->d_parse; break; }
                    { std::lock_guard<std::mutex> g(g_err_mu); ctx->error.clear(); }
                    if (nseg == 1) break;
                    nseg = 1;
                }
            }
            if (ctx->use_pace && ctx->d_pace && cnt > 8 * (int64_t)ctx->cu_count) {
                h.pace = ctx->d_pace;
                K4_HIP(ctx, hipMemsetAsync(h.pace, 0, k4::PACE_BYTES, pstream));
            }
            if (h.recs && nseg == 4) hipLaunchKernelGGL(k4::k4_hc_parse_seg4_kernel, dim3((unsigned)((cnt + 1) / 2)), dim3(64 * k4::HC_SEG_WAVES_PER_WG), 0, pstream, h);
            else if (h.recs && nseg == 2) hipLaunchKernelGGL(k4::k4_hc_parse_seg2_kernel, dim3((unsigned)((cnt + 3) / 4)), dim3(64 * k4::HC_SEG_WAVES_PER_WG), 0, pstream, h);
            else if (h.recs) hipLaunchKernelGGL(k4::k4_hc_parse_rec_kernel, dim3((unsigned)((cnt + k4::HC_REC_WAVES_PER_WG - 1) / k4::HC_REC_WAVES_PER_WG)), dim3(64 * k4::HC_REC_WAVES_PER_WG), 0, pstream, h);
            else hipLaunchKernelGGL(k4::k4_hc_parse_kernel, dim3((unsigned)((cnt + k4::HC_PARSE_WAVES_PER_WG - 1) / k4::HC_PARSE_WAVES_PER_WG)), dim3(64 * k4::HC_PARSE_WAVES_PER_WG), 0, pstream, h);
        }
        if (pickle) hipLaunchKernelGGL(k4::k4_pickle_finish_kernel, dim3((unsigned)((cnt + k4::PICKLE_FINISH_WAVES_PER_WG - 1) / k4::PICKLE_FINISH_WAVES_PER_WG)), dim3(64 * k4::PICKLE_FINISH_WAVES_PER_WG), 0, stream, a, d_enclen);
        K4_HIP(ctx, hipGetLastError());
    }
    return K4LZ4_OK;
}

/* per-block dictionaries for decode (device pointers) */
struct DictArgs {
    const uint8_t *dict;
    const uint64_t *off;
    const int32_t *len;
    const signed char *mode;
};

/* how many slots the LDS-table encoder kernel is launched with for a batch of cnt blocks: a fixed share of the number
 * (K4LZ4_SPLIT_PCT), or the most the device-side split by cost (k4_order_kernel, cost_pct hundredths of the cost, at least one
 * residency) can ask for -- that kernel clamps its count to this value, so a block is never left to neither kernel */
static int64_t lds_share(const k4lz4_ctx *ctx, int64_t cnt)
{
    const int64_t lds_slots = 8 * (int64_t)ctx->cu_count;
    if (ctx->split_pct > 0) return std::min<int64_t>(cnt, std::max<int64_t>(1, cnt * ctx->split_pct / 100));
    const int64_t pct = std::max<int64_t>(48, std::min<int64_t>(ctx->cost_pct, 100));
    return std::min<int64_t>(cnt, std::max<int64_t>(lds_slots, (cnt * pct + 99) / 100));
}

/* enqueue the kernels for n blocks; all pointers are device pointers */
int launch_inner(k4lz4_ctx *ctx, Kind kind, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                 const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n, int level, int flags,
                 hipStream_t stream, const DictArgs *dd, const int32_t *hostLen);

int launch(k4lz4_ctx *ctx, Kind kind, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
           const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n, int level, int flags,
           hipStream_t stream, const DictArgs *dd = nullptr, const int32_t *hostLen = nullptr)
{
    if (n == 0) return K4LZ4_OK;
    /* the context's scratch (dispatch order, global hash tables, HC work areas) may still be in use by a call that was
     * enqueued on another stream */
    if (ctx->busy && ctx->last_stream != stream) K4_HIP(ctx, hipStreamWaitEvent(stream, ctx->ev_busy, 0));
    const int rc = launch_inner(ctx, kind, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, level, flags, stream, dd, hostLen);
    if (hipEventRecord(ctx->ev_busy, stream) == hipSuccess) { ctx->busy = true; ctx->last_stream = stream; }
    else (void)hipGetLastError();
    return rc;
}

int launch_inner(k4lz4_ctx *ctx, Kind kind, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                 const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n, int level, int flags,
                 hipStream_t stream, const DictArgs *dd, const int32_t *hostLen)
{
    int64_t chunk_max = 1 << 24;   /* blocks per launch (grid.x * blockDim.x must stay < 2^32) */
#ifdef K4_PARSE_PROF
    const bool parse_prof_ok = true;
#else
    const bool parse_prof_ok = false;
#endif
    const bool encode_like = kind == KIND_ENCODE || kind == KIND_PICKLE;
    /* Fast-level LZ4Codec.Encode batches beyond what is resident at once (16 blocks per CU: the parse kernel's one workgroup per CU,
     * nine waves with their table in LDS and seven that move into a table as one becomes free): one persistent launch (round 6,
     * below); with K4LZ4_NO_PERSIST, or through the one-kernel encoders of rounds 1-4, equal parts of at most one residency each --
     * a block takes its ~2-3 ms whatever the batch, so those launches are best full.  Not for batches that may be ragged (pickles,
     * K4LZ4_FLAG_SEGMENTS): those need their one cost-ordered launch. */
    const bool parse_path = kind == KIND_ENCODE && level < K4LZ4_L03_HC && ctx->use_parse && ctx->accel == 1 && (!ctx->prof || parse_prof_ok || ctx->prof_stamp) &&
                            !(flags & (FLAG_SEGMENTS_OK | K4LZ4_FLAG_SEGMENTS | K4LZ4_FLAG_NO_SPLIT));
    /* round 6: the two-step encoder takes a batch beyond one residency in ONE persistent launch -- one workgroup of sixteen waves per
     * CU, every wave takes the next block of the cost order when it is done with one (the waves with LDS tables from the expensive
     * end, the others from the cheap end), records in per-wave slots -- instead of launches of one residency each with their own tail */
    /* round 6: ragged batches -- pickles, K4LZ4_FLAG_SEGMENTS -- through the two-step encoder as well: the blocks below 65 547 bytes by
     * k4_parse_kernel, the others (and the later segments of the cut ones) by k4_parse_seg_kernel, the join as before */
    const bool parse_seg_path = kind == KIND_ENCODE && level < K4LZ4_L03_HC && ctx->use_parse && ctx->parse_big && ctx->parse_seg && ctx->parse_inline_emit &&
                                ctx->accel == 1 && !ctx->prof && (flags & (FLAG_SEGMENTS_OK | K4LZ4_FLAG_SEGMENTS)) != 0 &&
                                !(flags & (K4LZ4_FLAG_NO_SPLIT | K4LZ4_FLAG_NO_REORDER | K4LZ4_FLAG_ALLOW_COPY));
    const bool parse_persistent = parse_path && ctx->parse_persist && ctx->parse_inline_emit && !ctx->prof && n > (int64_t)k4::PARSE_MAX_WAVES * (int64_t)ctx->cu_count &&
                                  !(flags & K4LZ4_FLAG_NO_REORDER);
    if (kind == KIND_ENCODE && level < K4LZ4_L03_HC && !(flags & (FLAG_SEGMENTS_OK | K4LZ4_FLAG_SEGMENTS | K4LZ4_FLAG_NO_SPLIT)) &&
        ctx->split_pct <= 0 && !ctx->prof && n > 16 * (int64_t)ctx->cu_count && !parse_persistent) {
        const int64_t parts = (n + 16 * (int64_t)ctx->cu_count - 1) / (16 * (int64_t)ctx->cu_count);
        chunk_max = (n + parts - 1) / parts;
    }
    if (parse_path && !parse_persistent) chunk_max = std::min<int64_t>(chunk_max, (int64_t)k4::PARSE_MAX_WAVES * (int64_t)ctx->cu_count);      /* (one residency) */
    if (encode_like && level >= K4LZ4_L03_HC) {
        const int rc = launch_hc(ctx, kind == KIND_PICKLE, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, level, flags, stream, hostLen);
        if (rc != K4LZ4_OK || kind != KIND_ENCODE || !(flags & K4LZ4_FLAG_ALLOW_COPY)) return rc;
        for (int64_t first = 0; first < n; first += chunk_max) {
            const int64_t cnt = std::min<int64_t>(chunk_max, n - first);
            k4::BatchArgs a{};
            a.src = src; a.srcOff = srcOff + first; a.srcLen = srcLen + first;
            a.dst = dst; a.dstOff = dstOff + first; a.dstCap = dstCap + first; a.outLen = outLen + first; a.n = cnt;
            hipLaunchKernelGGL(k4::k4_allow_copy_kernel, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, stream, a);
        }
        K4_HIP(ctx, hipGetLastError());
        return K4LZ4_OK;
    }
    /* (a small batch without a message big enough to be cut into segments keeps the one-kernel pickle path: the segment
     * machinery costs a plan kernel, a 4096-workgroup launch on a second queue, a join, and -- on a context's first such
     * call -- 269 MB of scratch; lengths the host does not know count as "may be big") */
    bool may_cut = ctx->use_segments;
    if (may_cut && hostLen && n <= (int64_t)ctx->pickle_split_min) {
        may_cut = false;
        for (int64_t i = 0; i < n && !may_cut; i++) may_cut = (uint32_t)hostLen[i] >= ctx->seg_min;
    }
    if (kind == KIND_PICKLE && (n > (int64_t)ctx->pickle_split_min || may_cut) && !ctx->prof) {
        /* Fast-level pickles of a batch go the encoders' way: slots prepared (block = envelope + 5, cap U - 1, exactly what
         * k4_pickle_kernel hands its encoder), the batch encoded by the two encoder kernels side by side -- the expensive
         * messages with their tables in LDS, the others with tables in memory, which puts a ragged batch on all the
         * chip's wave slots instead of the nine per CU that have room for a table in LDS --, envelopes closed afterwards. */
        for (int64_t first = 0; first < n; first += chunk_max) {
            const int64_t cnt = std::min<int64_t>(chunk_max, n - first);
            const size_t need_meta = (size_t)cnt * 16 + 64;
            if (need_meta > ctx->d_pk_meta_cap) K4_HIP(ctx, hipStreamSynchronize(stream));
            int rc = grow(ctx, &ctx->d_pk_meta, &ctx->d_pk_meta_cap, need_meta, false);
            if (rc != K4LZ4_OK) return rc;
            uint64_t *d_encoff = (uint64_t *)ctx->d_pk_meta;
            int32_t *d_enccap = (int32_t *)(d_encoff + cnt);
            int32_t *d_enclen = d_enccap + cnt;
            k4::BatchArgs a{};
            a.src = src; a.srcOff = srcOff + first; a.srcLen = srcLen + first; a.dst = dst; a.dstOff = dstOff + first;
            a.dstCap = dstCap + first; a.outLen = outLen + first; a.n = cnt; a.level = level; a.accel = 1;
            a.flags = flags | (g_enforce32.load(std::memory_order_relaxed) ? K4LZ4_FLAG_X32 : 0);
            hipLaunchKernelGGL(k4::k4_pickle_prep_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, a, d_encoff, d_enccap);
            rc = launch_inner(ctx, KIND_ENCODE, src, srcOff + first, srcLen + first, dst, d_encoff, d_enccap, d_enclen, cnt, level,
                              (flags & (K4LZ4_FLAG_NO_REORDER | K4LZ4_FLAG_X32)) | K4LZ4_FLAG_RAW_RETURN | FLAG_SEGMENTS_OK, stream, nullptr, hostLen ? hostLen + first : nullptr);
            if (rc != K4LZ4_OK) return rc;
            hipLaunchKernelGGL(k4::k4_pickle_finish_kernel, dim3((unsigned)((cnt + k4::PICKLE_FINISH_WAVES_PER_WG - 1) / k4::PICKLE_FINISH_WAVES_PER_WG)), dim3(64 * k4::PICKLE_FINISH_WAVES_PER_WG), 0, stream, a, d_enclen);
            K4_HIP(ctx, hipGetLastError());
        }
        return K4LZ4_OK;
    }
    /* cost-ordered dispatch (most expensive blocks first): encoders by default, decoders on request */
    const bool reorder = n > 1 &&
                         (encode_like ? !(flags & K4LZ4_FLAG_NO_REORDER)
                                      : ((flags & K4LZ4_FLAG_REORDER) != 0 ||
                                         /* pickles are ragged by nature: start the long ones first unless told not to */
                                         (kind == KIND_UNPICKLE && n > 64 && !(flags & K4LZ4_FLAG_NO_REORDER))));
    uint32_t *d_cost = nullptr, *d_order = nullptr, *d_hist = nullptr;
    if (reorder) {
        const size_t cnt_max = (size_t)std::min<int64_t>(chunk_max, n);
        const size_t hist_words = std::max<size_t>(2 * (size_t)k4::PCOST_BUCKETS, 2 * (size_t)k4::COST_BUCKETS + 16);
        const size_t need = cnt_max * 8 + hist_words * 4 + 64;
        if (need > ctx->d_sched_cap) K4_HIP(ctx, hipStreamSynchronize(stream));   /* scratch may still be in use */
        int rc = grow(ctx, &ctx->d_sched, &ctx->d_sched_cap, need, false);
        if (rc != K4LZ4_OK) return rc;
        d_hist = (uint32_t *)ctx->d_sched;
        d_cost = d_hist + hist_words;     /* [2 * COST_BUCKETS]: where the second encoder kernel's part of the order begins */
        d_order = d_cost + cnt_max;
    }
    for (int64_t first = 0; first < n; first += chunk_max) {
        const int64_t cnt = std::min<int64_t>(chunk_max, n - first);
        k4::BatchArgs a{};
        a.src = src; a.srcOff = srcOff + first; a.srcLen = srcLen + first;
        a.dst = dst; a.dstOff = dstOff + first; a.dstCap = dstCap + first;
        a.outLen = outLen + first; a.n = cnt; a.level = level; a.accel = ctx->accel;
        a.flags = flags | (g_enforce32.load(std::memory_order_relaxed) ? K4LZ4_FLAG_X32 : 0);
        /* pair decoders: the token chain two links at a time while the launch leaves wave slots free (k4lz4_decode.hpp, follow_tokens) */
        const bool hop2 = cnt <= (int64_t)ctx->hop2_max_per_cu * (int64_t)ctx->cu_count;
        a.prof = ctx->prof ? ctx->prof + k4::PROF_STRIDE * first : nullptr;
        a.status = ctx->d_status;
        if (((kind == KIND_ENCODE && !parse_path) || (K4_DEC_PACE && (kind == KIND_DECODE || kind == KIND_UNPICKLE))) && ctx->use_pace && ctx->d_pace && cnt > (int64_t)ctx->pace_min_per_cu * (int64_t)ctx->cu_count) {   /* k4lz4_common.hpp, Pace: more than two blocks per SIMD */
            a.pace = ctx->d_pace;
            K4_HIP(ctx, hipMemsetAsync(a.pace, 0, k4::PACE_BYTES, stream));
        }
        /* big blocks in several segments (k4lz4_segments.hpp): the plan is made on the device, the later segments run on a queue of
         * their own beside the ordinary kernels (which take a cut block's first segment), the pieces are joined afterwards */
        k4::SegArgs sg{};
        bool seg = false;
        if (kind == KIND_ENCODE && (flags & (FLAG_SEGMENTS_OK | K4LZ4_FLAG_SEGMENTS)) && ctx->use_segments && !a.prof && !(flags & K4LZ4_FLAG_ALLOW_COPY)) {
            static_assert(k4::SEG_HDR_DWORDS * 4 == 256, "seg_first_of finds the header in front of the items");
            /* how many segments the plan can come to: every item costs a 16 KiB snapshot and a 16 KiB table slot, so where the
             * host knows the lengths the arrays are sized by them (an upper bound: the plan's segments are never shorter than
             * seg_target, its threshold never below seg_min) instead of by the most a launch may have */
            size_t max_items = (size_t)k4::SEG_MAX_ITEMS;
            if (hostLen) {
                size_t bound = 0;
                for (int64_t i = 0; i < cnt && bound < max_items; i++) {
                    const int32_t u = hostLen[first + i];
                    if (u > 0 && (uint32_t)u >= ctx->seg_min) bound += ((size_t)u + ctx->seg_target - 1) / ctx->seg_target;
                }
                max_items = std::min(max_items, std::max<size_t>(bound, 2));
            }
            const size_t o_items = 256, o_work = o_items + max_items * sizeof(k4::SegItem);
            const size_t o_blocks = o_work + max_items * 4, o_snaps = (o_blocks + (size_t)k4::SEG_MAX_BLOCKS * 4 + 255) & ~(size_t)255;
            const size_t o_tables = o_snaps + max_items * k4::SEG_SNAP_DWORDS * 4, total = o_tables + max_items * 16384;
            if (total > ctx->d_seg_cap || (size_t)cnt * 4 > ctx->d_seg_first_cap) {
                K4_HIP(ctx, hipStreamSynchronize(stream));
                K4_HIP(ctx, hipStreamSynchronize(ctx->aux2));
            }
            int rc = grow(ctx, &ctx->d_seg, &ctx->d_seg_cap, total, false);
            if (rc == K4LZ4_OK) rc = grow(ctx, &ctx->d_seg_first, &ctx->d_seg_first_cap, (size_t)cnt * 4, false);
            if (rc != K4LZ4_OK) return rc;
            sg.hdr = (k4::SegHdr *)ctx->d_seg; sg.items = (k4::SegItem *)(ctx->d_seg + o_items); sg.work = (uint32_t *)(ctx->d_seg + o_work);
            sg.blocks = (uint32_t *)(ctx->d_seg + o_blocks); sg.snaps = (uint32_t *)(ctx->d_seg + o_snaps); sg.tables = (uint32_t *)(ctx->d_seg + o_tables);
            sg.first = (int32_t *)ctx->d_seg_first;
            sg.seg_min = ctx->seg_min; sg.seg_target = ctx->seg_target; sg.seg_warm = ctx->seg_warm; sg.seg_div = ctx->seg_div; sg.seg_target_max = ctx->seg_target_max; sg.spin_max = ctx->seg_spin_max; sg.max_items = (uint32_t)max_items;
            hipLaunchKernelGGL(k4::k4_seg_plan_kernel, dim3(1), dim3(256), 0, stream, a, sg);
            a.seg_first = sg.first; a.seg_items = sg.items; a.seg_snaps = sg.snaps;
            seg = true;
        }
        /* the later segments: on their queue, before the ordinary kernels are launched (their waves are the ones others wait for) */
        auto segments_start = [&]() -> int {
            if (!seg) return K4LZ4_OK;
            K4_HIP(ctx, hipEventRecord(ctx->ev_fork, stream));
            K4_HIP(ctx, hipStreamWaitEvent(ctx->aux2, ctx->ev_fork, 0));
            hipLaunchKernelGGL(k4::k4_encode_seg_kernel, dim3((unsigned)((sg.max_items + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)), dim3(64 * k4::ENCODE_WAVES_PER_WG), 0, ctx->aux2, a, sg);
            K4_HIP(ctx, hipEventRecord(ctx->ev_join2, ctx->aux2));
            return K4LZ4_OK;
        };
        auto segments_join = [&]() -> int {
            if (!seg) return K4LZ4_OK;
            K4_HIP(ctx, hipStreamWaitEvent(stream, ctx->ev_join2, 0));
            hipLaunchKernelGGL(k4::k4_seg_join_kernel, dim3((unsigned)std::min<int64_t>(k4::SEG_MAX_BLOCKS, cnt)), dim3(64), 0, stream, a, sg);
            return K4LZ4_OK;
        };
        if (dd && dd->dict) {
            a.dict = dd->dict; a.dictOff = dd->off + first; a.dictLen = dd->len + first;
            a.dictMode = dd->mode ? dd->mode + first : nullptr;
        }
        if (reorder && parse_path && kind == KIND_ENCODE && ctx->parse_pcost) {
            /* the two-kernel encoder's own estimate and order (k4lz4_parse.hpp, k4_pcost_kernel) */
            a.cost = d_cost; a.hist = d_hist; a.order_out = d_order;
            K4_HIP(ctx, hipMemsetAsync(d_hist, 0, 2 * k4::PCOST_BUCKETS * 4, stream));
            hipLaunchKernelGGL(k4::k4_pcost_kernel, dim3((unsigned)((cnt + k4::PCOST_WAVES_PER_WG - 1) / k4::PCOST_WAVES_PER_WG)), dim3(64 * k4::PCOST_WAVES_PER_WG), 0, stream, a);
            hipLaunchKernelGGL(k4::k4_porder_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, a);
            a.order = d_order;
        }
        else if (reorder) {
            a.cost = d_cost; a.hist = d_hist; a.order_out = d_order;
            K4_HIP(ctx, hipMemsetAsync(d_hist, 0, (2 * k4::COST_BUCKETS + 16) * 4, stream));
            hipLaunchKernelGGL(k4::k4_cost_kernel, dim3((unsigned)cnt), dim3(64), 0, stream, a, encode_like ? 0 : 1,
                               kind == KIND_ENCODE && parse_path ? k4::COST_SAMPLE_PARSE : k4::COST_SAMPLE);
            /* the split between the two encoder kernels by cost (see k4_order_kernel): 48 % of it, at least one residency of
             * the LDS-table kernel; K4LZ4_SPLIT_PCT fixes a share of the NUMBER of blocks instead */
            const bool two_kernels = kind == KIND_ENCODE && !a.prof && cnt > 512 && !(flags & K4LZ4_FLAG_NO_SPLIT) &&
                                     cnt > 8 * (int64_t)ctx->cu_count && ctx->split_pct <= 0;
            k4::BatchArgs ao = a;
            ao.first = two_kernels ? (uint32_t)ctx->cost_pct : 0u;
            ao.total = (uint32_t)((int64_t)ctx->lds_floor_per_cu * (int64_t)ctx->cu_count);
            hipLaunchKernelGGL(k4::k4_order_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, ao, (uint32_t)lds_share(ctx, cnt));
            a.order = d_order;
            if (two_kernels) a.split = d_hist + 2 * k4::COST_BUCKETS;
        }
        const unsigned wg4 = (unsigned)((cnt + k4::DECODE_WAVES_PER_WG - 1) / k4::DECODE_WAVES_PER_WG);
        /* LZ4Codec.Encode's case -- acceleration 1, no segments -- goes through the two-kernel encoder (k4lz4_parse.hpp): which
         * sequences (one wavefront per block, 64 x K positions per round, 16 blocks per workgroup = per CU dealt from the cost
         * order, the nine most expensive of a workgroup with their table in LDS), then their bytes (a throughput kernel), then
         * whatever block the parse left alone (65 547 bytes and more, very short ones) by the one-kernel encoder. */
        bool parse_here = kind == KIND_ENCODE && (parse_path || (parse_seg_path && (a.order || cnt == 1)));
        /* blocks of 65 547 bytes and more (byU32 tables) go through k4_parse_big_kernel behind the first launch -- where the parsing
         * waves write their blocks out themselves (a big block has more sequences than a record slot holds) */
        const bool big_ok = ctx->parse_big && ctx->parse_inline_emit;
        bool any_small = true, any_big = big_ok;
        if (parse_here && hostLen) {             /* nothing for the parse kernels in this part (all blocks too short): no scratch, no launch */
            any_small = false; any_big = false;
            for (int64_t i = 0; i < cnt && !(any_small && (any_big || !big_ok)); i++) {
                const int32_t len = hostLen[first + i];
                if (len >= (int32_t)k4::PARSE_MIN_LEN && len < k4::LIMIT_64K) any_small = true;
                else if (len >= k4::LIMIT_64K && big_ok) any_big = true;
            }
            parse_here = any_small || any_big;
        }
        int64_t waves = 0, nwg = 0, nwg_seg = 0;
        const bool seg_two_step = seg && parse_here && parse_seg_path;      /* (seg: the plan has been made above) */
        bool queue = false, slot_recs = false;
        size_t o_meta = 0, o_gtab = 0;
        if (parse_here) {
            waves = std::max<int64_t>(1, std::min<int64_t>(ctx->parse_waves, (cnt + ctx->cu_count - 1) / ctx->cu_count));
            /* K4LZ4_PARSE_QUEUE, and every batch beyond one residency: one workgroup per CU at most, every wave takes the next block of
             * the cost order when it is done with one */
            queue = (ctx->parse_queue || parse_persistent || parse_seg_path) && a.order && cnt > waves * (int64_t)ctx->cu_count;
            nwg = queue ? (int64_t)ctx->cu_count : (cnt + waves - 1) / waves;
            /* the segments' launch: always sixteen waves per workgroup, one workgroup per CU at most, everything from its queue */
            if (seg_two_step) nwg_seg = std::min<int64_t>(ctx->cu_count, (cnt + (int64_t)sg.max_items + k4::PARSE_MAX_WAVES - 1) / k4::PARSE_MAX_WAVES);
            /* records: a slot per wave of the launch where the parsing waves write their blocks out themselves, else one per block */
            slot_recs = ctx->parse_inline_emit;
            const size_t rec_slots = slot_recs ? std::max((size_t)nwg * (size_t)waves, (size_t)nwg_seg * k4::PARSE_MAX_WAVES) : (size_t)cnt;
            o_meta = rec_slots * k4::PARSE_REC_STRIDE * sizeof(uint2); o_gtab = (o_meta + (size_t)cnt * 8 + 64 + 255) & ~(size_t)255;
            const size_t need = o_gtab + std::max(waves > k4::PARSE_LDS_TABLES ? (size_t)nwg * k4::PARSE_MAX_WAVES * 16384 : 0, (size_t)nwg_seg * k4::PARSE_MAX_WAVES * 16384);
            if (need > ctx->d_parse_cap) K4_HIP(ctx, hipStreamSynchronize(stream));
            const int rcp = grow(ctx, &ctx->d_parse, &ctx->d_parse_cap, need, false);
            /* (no room for the records: the one-kernel encoders below need next to none -- the same bytes, a third slower) */
            if (rcp != K4LZ4_OK) { std::lock_guard<std::mutex> g(g_err_mu); ctx->error.clear(); parse_here = false; }
        }
        if (parse_here) {
            k4::ParseArgs pa{};
            pa.recs = (uint2 *)ctx->d_parse; pa.meta = (uint32_t *)(ctx->d_parse + o_meta); pa.gtab = (uint32_t *)(ctx->d_parse + o_gtab);
            pa.nwg = (uint32_t)nwg;
            pa.inline_emit = ctx->parse_inline_emit ? 1u : 0u;
            pa.slot_recs = slot_recs ? 1u : 0u;
            pa.migrate = ctx->parse_migrate ? 1u : 0u;
            pa.big = any_big ? 1u : 0u;
            /* eight words behind the counts: the two launches' queues (three words each) and, word 3, the number of blocks the first
             * launch leaves to the second */
            K4_HIP(ctx, hipMemsetAsync(pa.meta + 2 * cnt, 0, 32, stream));
            if (queue) pa.queue = pa.meta + 2 * cnt;
            pa.nbig = pa.meta + 2 * cnt + 3;
            /* (the first launch also says whose every block is -- PARSE_BIG / PARSE_REST --, so it runs even without a block of its own) */
            hipLaunchKernelGGL(k4::k4_parse_kernel, dim3((unsigned)nwg), dim3((unsigned)(64 * waves)), 0, stream, a, pa);
            if (seg_two_step) {
                k4::ParseArgs pb = pa;
                pb.queue = pa.meta + 2 * cnt + 4;
                pb.nwg = (uint32_t)nwg_seg;
                hipLaunchKernelGGL(k4::k4_parse_seg_kernel, dim3((unsigned)nwg_seg), dim3(64 * k4::PARSE_MAX_WAVES), 0, stream, a, pb, sg);
            } else if (any_big) {
                k4::ParseArgs pb = pa;
                if (queue) pb.queue = pa.queue + 4;
                hipLaunchKernelGGL(k4::k4_parse_big_kernel, dim3((unsigned)nwg), dim3((unsigned)(64 * waves)), 0, stream, a, pb);
            }
            if (!pa.inline_emit)
                hipLaunchKernelGGL(k4::k4_emit_kernel, dim3((unsigned)((cnt + k4::EMIT_WAVES_PER_WG - 1) / k4::EMIT_WAVES_PER_WG)), dim3(64 * k4::EMIT_WAVES_PER_WG), 0, stream, a, pa);
            bool rest = true;          /* (where the host knows the lengths it knows whether there is anything left) */
            if (hostLen) {
                rest = false;
                for (int64_t i = 0; i < cnt && !rest; i++) rest = hostLen[first + i] < (int32_t)k4::PARSE_MIN_LEN || (hostLen[first + i] >= k4::LIMIT_64K && !any_big);
            }
            if (rest)
                hipLaunchKernelGGL(k4::k4_encode_fast_rest_kernel, dim3((unsigned)((cnt + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)), dim3(64 * k4::ENCODE_WAVES_PER_WG), 0, stream, a, pa);
            if (seg_two_step)            /* the cut blocks' pieces: joined, or encoded again (k4lz4_segments.hpp) */
                hipLaunchKernelGGL(k4::k4_seg_join_kernel, dim3((unsigned)std::min<int64_t>(k4::SEG_MAX_BLOCKS, cnt)), dim3(64), 0, stream, a, sg);
            if (flags & K4LZ4_FLAG_ALLOW_COPY)
                hipLaunchKernelGGL(k4::k4_allow_copy_kernel, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, stream, a);
            K4_HIP(ctx, hipGetLastError());
            continue;
        }
        switch (kind) {
        case KIND_ENCODE:
            if (a.prof && !ctx->prof_stamp) {
                if (ctx->prof_gtab) {    /* diagnostics: the instrumented encoder with its table in global memory, like the global-table kernel */
                    if ((size_t)cnt * 16384 > ctx->d_gtab_cap) {
                        K4_HIP(ctx, hipStreamSynchronize(ctx->aux));
                        int rc2 = grow(ctx, &ctx->d_gtab, &ctx->d_gtab_cap, (size_t)cnt * 16384, false);
                        if (rc2 != K4LZ4_OK) return rc2;
                    }
                    a.gtab = (uint32_t *)ctx->d_gtab;
                }
                hipLaunchKernelGGL(k4::k4_encode_fast_prof_kernel, dim3((unsigned)cnt), dim3(64), 0, stream, a);
            }
            else if (reorder && cnt > 512 && !(flags & K4LZ4_FLAG_NO_SPLIT)) {
                /* Only 8 blocks per CU fit with their hash table in LDS.  The most expensive blocks
                 * (front of the dispatch order) take those slots; the others are encoded at the same
                 * time on a second queue by the global-memory-table variant of the kernel. */
                /* measured on MI355X (profiles/r02_split_sweep.txt): best when the LDS-table kernel gets one full
                 * residency of the chip (8 blocks per CU) or about 48 % of a larger batch; one block more
                 * than a residency starts a second pass and costs 20 % */
                const int64_t lds_slots = 8 * (int64_t)ctx->cu_count;
                const int64_t n_lds = lds_share(ctx, cnt);
                /* (with the split decided on the device, a.split: n_lds is the most the LDS-table kernel can get -- the share of
                 * the cost is never a larger share of the number -- and the other kernel is sized for the least) */
                const int64_t n_g = cnt - (a.split ? std::min<int64_t>(cnt, (int64_t)ctx->lds_floor_per_cu * (int64_t)ctx->cu_count) : n_lds);
                const int64_t gchunk = 8192;
                if (n_g > 0 && (size_t)std::min(n_g, gchunk) * 16384 > ctx->d_gtab_cap) {
                    K4_HIP(ctx, hipStreamSynchronize(ctx->aux));
                    int rc2 = grow(ctx, &ctx->d_gtab, &ctx->d_gtab_cap, (size_t)std::min(n_g, gchunk) * 16384, false);
                    if (rc2 != K4LZ4_OK) return rc2;
                }
                { const int rcs = segments_start(); if (rcs != K4LZ4_OK) return rcs; }
                K4_HIP(ctx, hipEventRecord(ctx->ev_fork, stream));
                K4_HIP(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
                for (int64_t g0 = 0; g0 < n_g; g0 += gchunk) {
                    k4::BatchArgs ag = a;
                    ag.first = (uint32_t)((a.split ? 0 : n_lds) + g0);
                    ag.total = (uint32_t)cnt;
                    ag.gtab = (uint32_t *)ctx->d_gtab;
                    ag.n = std::min(gchunk, n_g - g0);
                    if (seg) hipLaunchKernelGGL(k4::k4_encode_fast_gtab_seg_kernel, dim3((unsigned)((ag.n + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)),
                                                dim3(64 * k4::ENCODE_WAVES_PER_WG), 0, ctx->aux, ag);
                    else hipLaunchKernelGGL(k4::k4_encode_fast_gtab_kernel, dim3((unsigned)((ag.n + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)),
                                       dim3(64 * k4::ENCODE_WAVES_PER_WG), 0, ctx->aux, ag);
                }
                K4_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
                {
                    k4::BatchArgs al = a;
                    al.n = n_lds;
                    if (seg)
                        hipLaunchKernelGGL(k4::k4_encode_fast_seg_kernel, dim3((unsigned)((n_lds + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)),
                                           dim3(64 * k4::ENCODE_WAVES_PER_WG), 0, stream, al);
                    else if (n_g == 0 && cnt <= lds_slots)
                        hipLaunchKernelGGL(k4::k4_encode_fast_more_kernel, dim3((unsigned)((n_lds + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)),
                                           dim3(64 * k4::ENCODE_WAVES_PER_WG), 0, stream, al);
                    else
                        hipLaunchKernelGGL(k4::k4_encode_fast_kernel, dim3((unsigned)((n_lds + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)),
                                           dim3(64 * k4::ENCODE_WAVES_PER_WG), 0, stream, al);
                }
                K4_HIP(ctx, hipStreamWaitEvent(stream, ctx->ev_join, 0));
                { const int rcs = segments_join(); if (rcs != K4LZ4_OK) return rcs; }
            }
            else if (seg) {
                { const int rcs = segments_start(); if (rcs != K4LZ4_OK) return rcs; }
                hipLaunchKernelGGL(k4::k4_encode_fast_seg_kernel, dim3((unsigned)((cnt + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)),
                                   dim3(64 * k4::ENCODE_WAVES_PER_WG), 0, stream, a);
                { const int rcs = segments_join(); if (rcs != K4LZ4_OK) return rcs; }
            }
            else if (cnt <= 8 * (int64_t)ctx->cu_count)       /* a half-empty chip: the variant that buys latency with instructions */
                hipLaunchKernelGGL(k4::k4_encode_fast_more_kernel, dim3((unsigned)((cnt + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)),
                                   dim3(64 * k4::ENCODE_WAVES_PER_WG), 0, stream, a);
            else hipLaunchKernelGGL(k4::k4_encode_fast_kernel, dim3((unsigned)((cnt + k4::ENCODE_WAVES_PER_WG - 1) / k4::ENCODE_WAVES_PER_WG)),
                                    dim3(64 * k4::ENCODE_WAVES_PER_WG), 0, stream, a);
            break;
        case KIND_DECODE:
            if (a.prof && ctx->prof_pair && !ctx->prof_stamp) {
                hipLaunchKernelGGL(k4::k4_decode_pair_prof_kernel, dim3((unsigned)((cnt + k4::DECODE_PAIRS_PER_WG - 1) / k4::DECODE_PAIRS_PER_WG)),
                                   dim3(128 * k4::DECODE_PAIRS_PER_WG), 0, stream, a);
            }
            else if (a.prof && !ctx->prof_stamp) hipLaunchKernelGGL(k4::k4_decode_prof_kernel, dim3(wg4), dim3(64 * k4::DECODE_WAVES_PER_WG), 0, stream, a);
            else if (cnt > 24 * (int64_t)ctx->cu_count)   /* more blocks than can be resident (6 waves x 4 SIMDs per CU) */
                hipLaunchKernelGGL(k4::k4_decode_dense_kernel, dim3(wg4), dim3(64 * k4::DECODE_WAVES_PER_WG), 0, stream, a);
            else if (cnt <= 16 * (int64_t)ctx->cu_count && !ctx->no_pair) {
                /* at most half the chip's wave slots (8 per SIMD at 64 VGPRs) are needed: two waves per block, one
                 * parsing ahead of the one that copies */
                hipLaunchKernelGGL(hop2 ? k4::k4_decode_pair2_kernel : k4::k4_decode_pair_kernel, dim3((unsigned)((cnt + k4::DECODE_PAIRS_PER_WG - 1) / k4::DECODE_PAIRS_PER_WG)),
                                   dim3(128 * k4::DECODE_PAIRS_PER_WG), 0, stream, a);
            }
            else hipLaunchKernelGGL(k4::k4_decode_kernel, dim3(wg4), dim3(64 * k4::DECODE_WAVES_PER_WG), 0, stream, a);
            break;
        case KIND_PICKLE:
            hipLaunchKernelGGL(k4::k4_pickle_kernel, dim3((unsigned)cnt), dim3(64), 0, stream, a);
            break;
        case KIND_UNPICKLE:
            if (cnt <= 64 * (int64_t)ctx->cu_count && !ctx->no_pair) {
                hipLaunchKernelGGL(hop2 ? k4::k4_unpickle_pair2_kernel : k4::k4_unpickle_pair_kernel, dim3((unsigned)((cnt + k4::DECODE_PAIRS_PER_WG - 1) / k4::DECODE_PAIRS_PER_WG)),
                                   dim3(128 * k4::DECODE_PAIRS_PER_WG), 0, stream, a);
            }
            else hipLaunchKernelGGL(k4::k4_unpickle_kernel, dim3(wg4), dim3(64 * k4::DECODE_WAVES_PER_WG), 0, stream, a);
            break;
        }
        if (kind == KIND_ENCODE && (flags & K4LZ4_FLAG_ALLOW_COPY))
            hipLaunchKernelGGL(k4::k4_allow_copy_kernel, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, stream, a);   /* 4 blocks per workgroup */
        K4_HIP(ctx, hipGetLastError());
    }
    return K4LZ4_OK;
}

int grow(k4lz4_ctx *ctx, uint8_t **p, size_t *cap, size_t need, bool pinned)
{
    if (need <= *cap) return K4LZ4_OK;
    const size_t want = std::max(need, *cap + *cap / 2);
    if (*p) {
        if (pinned) (void)hipHostFree(*p); else (void)hipFree(*p);
        *p = nullptr; *cap = 0;
    }
    hipError_t e = pinned ? hipHostMalloc((void **)p, want, hipHostMallocDefault) : hipMalloc((void **)p, want);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(ctx, K4LZ4_E_NOMEM, "out of device/pinned memory"); }
    *cap = want;
    return K4LZ4_OK;
}

int check_batch_args(k4lz4_ctx *ctx, const void *src, const void *srcOff, const void *srcLen, const void *dst,
                     const void *dstOff, const void *dstCap, const void *outLen, int64_t n)
{
    if (!ctx) return fail(nullptr, K4LZ4_E_ARG, "ctx is NULL");
    if (n < 0) return fail(ctx, K4LZ4_E_ARG, "negative block count");
    if (n > 0 && (!src || !srcOff || !srcLen || !dst || !dstOff || !dstCap || !outLen))
        return fail(ctx, K4LZ4_E_ARG, "NULL batch pointer");
    return K4LZ4_OK;
}

constexpr size_t STAGE_CHUNK = (size_t)16 << 20;

Pool *pool_of(k4lz4_ctx *ctx)
{
    if (!ctx->pool && !ctx->pool_failed) {
        const unsigned hw = std::thread::hardware_concurrency();
        const int nthreads = (int)std::min<unsigned>((unsigned)ctx->stage_threads, hw > 2 ? hw / 2 - 1 : 0u);
        /* std::thread's constructor throws when the system has no thread to give: nothing may leave a C entry point that way,
         * and the copies work without helpers (pool == nullptr: the calling thread does them alone) */
        try { ctx->pool = new Pool(nthreads); } catch (...) { ctx->pool = nullptr; ctx->pool_failed = true; }
    }
    return ctx->pool;
}

Pool *pool_dl_of(k4lz4_ctx *ctx)
{
    if (!ctx->pool_dl && !ctx->pool_dl_failed) {
        const unsigned hw = std::thread::hardware_concurrency();
        const int nthreads = (int)std::min<unsigned>((unsigned)ctx->stage_threads, hw > 2 ? hw / 2 - 1 : 0u);
        try { ctx->pool_dl = new Pool(nthreads); } catch (...) { ctx->pool_dl = nullptr; ctx->pool_dl_failed = true; }
    }
    return ctx->pool_dl;
}

/* memcpy split over the helper threads */
void parallel_copy(k4lz4_ctx *ctx, uint8_t *d, const uint8_t *s, size_t nbytes)
{
    Pool *p = pool_of(ctx);
    const int parts = p ? (int)std::min<size_t>(p->workers.size() + 1, std::max<size_t>(1, nbytes >> 20)) : 1;
    if (parts <= 1) { memcpy(d, s, nbytes); return; }
    /* (the parts' size rounded UP before it is rounded to 64: with nbytes / parts a multiple of 64 and a remainder left, `parts`
     * pieces of the rounded-down size stop up to parts - 1 bytes short of the end -- the last bytes of an upload's last chunk
     * stayed what the pinned buffer held before; found by tests/tools/gpu_stress_encode.py in round 5) */
    const size_t per = (((nbytes + (size_t)parts - 1) / (size_t)parts) + 63) & ~(size_t)63;
    p->parallel(parts, [=](int i) {
        const size_t lo = per * (size_t)i;
        if (lo < nbytes) memcpy(d + lo, s + lo, std::min(per, nbytes - lo));
    });
}

/* caller's (pageable) memory -> device: small transfers directly, big ones through two pinned buffers so that the copy into
 * the buffer for chunk k+1 runs while the DMA engine moves chunk k */
int staged_upload(k4lz4_ctx *ctx, uint8_t *d_dst, const uint8_t *h_src, size_t nbytes, hipStream_t st)
{
    if (nbytes < 2 * STAGE_CHUNK) {
        K4_HIP(ctx, hipMemcpyAsync(d_dst, h_src, nbytes, hipMemcpyHostToDevice, st));
        return K4LZ4_OK;
    }
    int rc;
    for (int b = 0; b < 2; b++)
        if ((rc = grow(ctx, &ctx->h_in[b], &ctx->h_in_cap[b], STAGE_CHUNK, true)) != K4LZ4_OK) return rc;
    bool used[2] = {true, true};            /* an earlier upload of this call may still be reading the buffers */
    int b = 0;
    for (size_t pos = 0; pos < nbytes; pos += STAGE_CHUNK, b ^= 1) {
        const size_t len = std::min(STAGE_CHUNK, nbytes - pos);
        if (used[b]) K4_HIP(ctx, hipEventSynchronize(ctx->ev_in[b]));   /* the DMA out of this buffer two chunks ago */
        parallel_copy(ctx, ctx->h_in[b], h_src + pos, len);
        K4_HIP(ctx, hipMemcpyAsync(d_dst + pos, ctx->h_in[b], len, hipMemcpyHostToDevice, st));
        K4_HIP(ctx, hipEventRecord(ctx->ev_in[b], st));
        used[b] = true;
    }
    return K4LZ4_OK;
}

/* The same for blocks that lie far apart in the caller's memory (compressed blocks in slots of worst-case size: the
 * span is 1.7 times the bytes): only the blocks travel, packed next to each other at 16-byte steps -- the copy into the
 * pinned buffer is being made anyway.  packed[i] = where block i goes on the device (ascending, 16-byte steps, given);
 * blocks first .. last-1. */
int staged_upload_packed(k4lz4_ctx *ctx, uint8_t *d_dst, const uint8_t *h_base, const uint64_t *srcOff, const int32_t *srcLen,
                         int64_t first_block, int64_t last_block, const uint64_t *packed, hipStream_t st)
{
    int rc;
    for (int b = 0; b < 2; b++)
        if ((rc = grow(ctx, &ctx->h_in[b], &ctx->h_in_cap[b], STAGE_CHUNK, true)) != K4LZ4_OK) return rc;
    bool used[2] = {true, true};            /* an earlier upload of this call may still be reading the buffers */
    int b = 0;
    for (int64_t first = first_block; first < last_block; b ^= 1) {
        const uint64_t base = packed[first];             /* device offset of the chunk being filled */
        uint64_t fill = 0;
        int64_t last = first;
        for (; last < last_block; last++) {
            const uint64_t len = srcLen[last] > 0 ? (((uint64_t)srcLen[last] + 15u) & ~(uint64_t)15u) : 0u;
            if (fill + len > STAGE_CHUNK) break;
            fill += len;
        }
        if (used[b]) K4_HIP(ctx, hipEventSynchronize(ctx->ev_in[b]));   /* the DMA out of this buffer two chunks ago */
        uint8_t *buf = ctx->h_in[b];
        const int64_t cnt = last - first;
        Pool *p = pool_of(ctx);
        const int parts = p && fill >= ((uint64_t)2 << 20) ? (int)std::min<int64_t>((int64_t)p->workers.size() + 1, cnt) : 1;
        auto body = [&, parts](int part) {
            const int64_t a = first + cnt * part / parts, e = first + cnt * (part + 1) / parts;
            for (int64_t i = a; i < e; i++)
                if (srcLen[i] > 0) memcpy(buf + (packed[i] - base), h_base + srcOff[i], (size_t)srcLen[i]);
        };
        if (parts <= 1) body(0); else p->parallel(parts, body);
        if (fill) K4_HIP(ctx, hipMemcpyAsync(d_dst + base, buf, (size_t)fill, hipMemcpyHostToDevice, st));
        K4_HIP(ctx, hipEventRecord(ctx->ev_in[b], st));
        used[b] = true;
        first = last;
    }
    return K4LZ4_OK;
}

/* device -> the caller's slots: block i's stored[i] bytes sit at d_from + from_off[i]; chunks of whole blocks come over
 * into a pinned buffer while the previous chunk is scattered into the slots by the helper threads */
int staged_download(k4lz4_ctx *ctx, uint8_t *dst, const uint64_t *dstOff, const uint8_t *d_from, const uint64_t *from_off,
                    const int32_t *stored, int64_t n, hipStream_t st, Pool *p)
{
    struct Chunk { int64_t first, last; uint64_t lo, hi; };
    std::vector<Chunk> chunks;
    size_t biggest = 0;
    for (int64_t i = 0; i < n;) {
        while (i < n && stored[i] <= 0) i++;
        if (i >= n) break;
        Chunk c{i, i, from_off[i], from_off[i] + (uint64_t)stored[i]};
        for (int64_t j = i + 1; j < n; j++) {
            if (stored[j] <= 0) { c.last = j; continue; }
            const uint64_t hi = from_off[j] + (uint64_t)stored[j];
            if (hi - c.lo > STAGE_CHUNK) break;
            c.last = j; c.hi = hi;
        }
        biggest = std::max(biggest, (size_t)(c.hi - c.lo));
        chunks.push_back(c);
        i = c.last + 1;
    }
    if (chunks.empty()) return K4LZ4_OK;
    int rc;
    for (int b = 0; b < 2; b++)
        if ((rc = grow(ctx, &ctx->h_out[b], &ctx->h_out_cap[b], biggest + 64, true)) != K4LZ4_OK) return rc;
    auto scatter = [&](const Chunk &c, const uint8_t *buf) {
        const int64_t cnt = c.last - c.first + 1;
        const int parts = p && (c.hi - c.lo) >= ((uint64_t)2 << 20) ? (int)std::min<int64_t>((int64_t)p->workers.size() + 1, cnt) : 1;
        auto body = [&, parts](int part) {
            const int64_t a = c.first + cnt * part / parts, e = c.first + cnt * (part + 1) / parts;
            for (int64_t i = a; i < e; i++)
                if (stored[i] > 0) memcpy(dst + dstOff[i], buf + (from_off[i] - c.lo), (size_t)stored[i]);
        };
        if (parts <= 1) body(0); else p->parallel(parts, body);
    };
    for (size_t k = 0; k < chunks.size(); k++) {
        const int b = (int)(k & 1);
        K4_HIP(ctx, hipMemcpyAsync(ctx->h_out[b], d_from + chunks[k].lo, (size_t)(chunks[k].hi - chunks[k].lo), hipMemcpyDeviceToHost, st));
        K4_HIP(ctx, hipEventRecord(ctx->ev_out[b], st));
        if (k > 0) {                                      /* the previous chunk has arrived by now, or soon: scatter it while this one flies */
            K4_HIP(ctx, hipEventSynchronize(ctx->ev_out[b ^ 1]));
            scatter(chunks[k - 1], ctx->h_out[b ^ 1]);
        }
    }
    const int lastb = (int)((chunks.size() - 1) & 1);
    K4_HIP(ctx, hipEventSynchronize(ctx->ev_out[lastb]));
    scatter(chunks.back(), ctx->h_out[lastb]);
    return K4LZ4_OK;
}

/* Host memory the caller has page-locked through k4lz4_host_register: the DMA engine reads and writes it directly, so a
 * host-pointer call whose source or destination lies inside such a range skips the pinned staging buffers on that side. */
std::mutex g_reg_mu;
struct RegRange { uintptr_t first; size_t second; bool pinned; };      /* pinned: hipHostRegister has succeeded -- until then the range only blocks overlapping registrations */
std::vector<RegRange> g_reg;   /* [start, bytes), guarded by g_reg_mu */

bool registered(const void *p, size_t bytes)
{
    if (!p || !bytes) return false;
    const uintptr_t a = (uintptr_t)p;
    std::lock_guard<std::mutex> g(g_reg_mu);
    for (const auto &r : g_reg)
        if (r.pinned && a >= r.first && a - r.first <= r.second && bytes <= r.second - (a - r.first)) return true;
    return false;
}

constexpr size_t DIRECT_RUNS_MAX = 64;

/* device -> registered caller memory without the stop in a pinned buffer: blocks whose stored bytes follow one another both on
 * the device and in the caller's memory travel as one copy (a decoded batch in adjacent slots: one copy per part).  More than
 * DIRECT_RUNS_MAX such runs (compressed blocks in worst-case slots: every block its own) are cheaper through the buffers:
 * returns false and has enqueued nothing.  Exactly stored[i] bytes land in slot i, as on the staged way. */
bool direct_download(k4lz4_ctx *ctx, uint8_t *dst, const uint64_t *dstOff, const uint8_t *d_from, const uint64_t *from_off,
                     const int32_t *stored, int64_t n, hipStream_t st, int *rc)
{
    struct Run { uint64_t host, dev, len; };
    std::vector<Run> runs;
    for (int64_t i = 0; i < n; i++) {
        if (stored[i] <= 0) continue;
        if (!runs.empty() && runs.back().host + runs.back().len == dstOff[i] && runs.back().dev + runs.back().len == from_off[i])
            runs.back().len += (uint64_t)stored[i];
        else {
            if (runs.size() == DIRECT_RUNS_MAX) return false;
            runs.push_back(Run{dstOff[i], from_off[i], (uint64_t)stored[i]});
        }
    }
    *rc = K4LZ4_OK;
    for (const Run &r : runs) {
        const hipError_t e = hipMemcpyAsync(dst + r.host, d_from + r.dev, (size_t)r.len, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) { *rc = hip_fail(ctx, e, "hipMemcpyAsync"); break; }
    }
    return true;
}

int run_host_inner(k4lz4_ctx *ctx, Kind kind, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                   uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n, int level,
                   int flags, const DictArgs *hd);

/* host-pointer batch: stage, run, scatter.  A call that fails part-way has copies and kernels in flight on the context's
 * two queues that use its staging buffers and scratch: they are drained -- and a status bit they may have raised is taken
 * with them -- before the error is returned, so that the next call on the context starts from nothing. */
int run_host(k4lz4_ctx *ctx, Kind kind, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
             uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n, int level,
             int flags, const DictArgs *hd = nullptr)
{
    const int rc = run_host_inner(ctx, kind, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, level, flags, hd);
    if (rc != K4LZ4_OK && ctx) {
        const std::string why = ctx->error;
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        if (ctx->copyq) (void)hipStreamSynchronize(ctx->copyq);
        if (ctx->dlq) (void)hipStreamSynchronize(ctx->dlq);
        if (ctx->aux) (void)hipStreamSynchronize(ctx->aux);
        if (ctx->aux2) (void)hipStreamSynchronize(ctx->aux2);
        (void)hipGetLastError();
        (void)take_device_status(ctx);
        ctx->error = why;
        tl_error = why;
    }
    return rc;
}

int run_host_inner(k4lz4_ctx *ctx, Kind kind, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                   uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n, int level,
                   int flags, const DictArgs *hd)
{
    int rc = check_batch_args(ctx, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n);
    if (rc != K4LZ4_OK) return rc;
    if (kind == KIND_ENCODE || kind == KIND_PICKLE) {
        rc = check_level(ctx, level);
        if (rc != K4LZ4_OK) return rc;
    }
    if (n == 0) return K4LZ4_OK;
    K4_HIP(ctx, hipSetDevice(ctx->device));
    const auto tr0 = std::chrono::steady_clock::now();
    auto lap = [&, last = tr0](const char *what) mutable {
        if (!ctx->trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[k4lz4 trace] %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    };

    /* source span and compact destination layout */
    uint64_t lo = UINT64_MAX, hi = 0;
    std::vector<uint64_t> h_soff((size_t)n), h_doff((size_t)n);
    std::vector<int32_t> h_cap((size_t)n);
    uint64_t dtotal = 0;
    uint64_t packed_bytes = 0, longest = 0;
    bool ascending = true;                  /* the blocks lie one after the other in the caller's memory, in index order */
    for (int64_t i = 0; i < n; i++) {
        const int32_t len = srcLen[i];
        if (len > 0) {
            if (hi != 0 && srcOff[i] < hi) ascending = false;
            lo = std::min(lo, srcOff[i]);
            hi = std::max(hi, srcOff[i] + (uint64_t)len);
            packed_bytes += ((uint64_t)len + 15u) & ~(uint64_t)15u;
            longest = std::max<uint64_t>(longest, (uint64_t)len);
        }
        h_cap[(size_t)i] = dstCap[i] < 0 ? 0 : dstCap[i];
        h_doff[(size_t)i] = dtotal;
        dtotal += ((uint64_t)h_cap[(size_t)i] + 15u) & ~(uint64_t)15u;
    }
    if (lo == UINT64_MAX) { lo = 0; hi = 0; }
    const size_t span = (size_t)(hi - lo);
    /* blocks that fill less than 7/8 of their span (and each fit a staging chunk) travel packed */
    /* a registered source goes up as it lies (its span, straight from the caller's pages) unless that is over twice the bytes */
    const bool src_reg = span > 0 && !ctx->no_direct && registered(src + lo, span) && (uint64_t)span * 100u <= (uint64_t)ctx->direct_span_pct * packed_bytes;
    const bool packed = !src_reg && span >= 2 * STAGE_CHUNK && packed_bytes + (packed_bytes >> 3) < span && longest <= STAGE_CHUNK;
    {
        uint64_t at = 0;
        for (int64_t i = 0; i < n; i++) {
            const uint64_t len = srcLen[i] > 0 ? (uint64_t)srcLen[i] : 0u;
            h_soff[(size_t)i] = packed ? at : (len ? srcOff[i] - lo : 0);
            at += (len + 15u) & ~(uint64_t)15u;
        }
    }
    /* Two halves (of about equal source bytes) when the call is big: see k4lz4_ctx::copyq.  HC levels and dictionaries keep
     * to one part (their launches size scratch and synchronise on their own). */
    const uint64_t up_bytes = packed ? packed_bytes : (uint64_t)span;
    const bool hc = (kind == KIND_ENCODE || kind == KIND_PICKLE) && level >= K4LZ4_L03_HC;
    /* Encoders: two parts (a part's kernels last as long as their slowest block whatever their number, and the parts' kernels
     * run one after the other).  Decoders, whose kernels are short: four, so that more of the way up and of the way down
     * overlap. */
    bool dst_reg = false;                   /* the caller's slots lie in registered memory: whole runs of them come back in one copy each */
    if (!ctx->no_direct) {
        uint64_t dlo = UINT64_MAX, dhi = 0;
        for (int64_t i = 0; i < n; i++)
            if (h_cap[(size_t)i] > 0) { dlo = std::min(dlo, dstOff[i]); dhi = std::max(dhi, dstOff[i] + (uint64_t)h_cap[(size_t)i]); }
        dst_reg = dlo != UINT64_MAX && registered(dst + dlo, (size_t)(dhi - dlo));
    }
    int nparts = 1;
    int64_t part_lo[MAX_PARTS], part_hi[MAX_PARTS];
    for (int q = 0; q < MAX_PARTS; q++) { part_lo[q] = q ? n : 0; part_hi[q] = n; }
    if (n >= 1024 && up_bytes >= 4 * STAGE_CHUNK && (packed || ascending) && !hc && !(hd && hd->dict)) {
        const bool decode_like = kind == KIND_DECODE || kind == KIND_UNPICKLE;
        /* (a registered destination: no scatter thread to keep fed, so the shorter the first and the last part the better -- eight
         * parts 36.9 GiB/s, six 35.9, four 34.4 on the bench batch; through the buffers eight parts are slower than four, 27 against 29) */
        const int want = decode_like && n >= 2048 && up_bytes >= 8 * STAGE_CHUNK ? (dst_reg && n >= 4096 ? ctx->dec_parts_direct : ctx->dec_parts) : 2;
        int64_t cuts[MAX_PARTS + 1];
        for (int q = 0; q <= MAX_PARTS; q++) cuts[q] = q ? n : 0;
        uint64_t acc = 0;
        int k = 1;
        for (int64_t i = 0; i < n && k < want; i++) {
            acc += srcLen[i] > 0 ? (((uint64_t)srcLen[i] + 15u) & ~(uint64_t)15u) : 0u;
            if (acc >= packed_bytes * (uint64_t)k / (uint64_t)want) cuts[k++] = i + 1;   /* (the blocks' own bytes: a span has its gaps) */
        }
        bool ok = k == want;
        cuts[want] = n;
        for (int q = 0; q < want && ok; q++) ok = cuts[q + 1] - cuts[q] >= 256;
        if (ok) {
            nparts = want;
            for (int q = 0; q < want; q++) { part_lo[q] = cuts[q]; part_hi[q] = cuts[q + 1]; }
        }
    }
    const size_t meta_bytes = (size_t)n * (8 + 4 + 8 + 4 + 4 + 8);
    if ((rc = grow(ctx, &ctx->d_src, &ctx->d_src_cap, span + 64, false)) != K4LZ4_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_dst, &ctx->d_dst_cap, (size_t)dtotal + 64, false)) != K4LZ4_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_meta, &ctx->d_meta_cap, meta_bytes + 64, false)) != K4LZ4_OK) return rc;
    if ((rc = grow(ctx, &ctx->h_len, &ctx->h_len_cap, (size_t)n * 4 + 64, true)) != K4LZ4_OK) return rc;

    uint64_t *d_soff = (uint64_t *)ctx->d_meta;
    uint64_t *d_doff = d_soff + n;
    uint64_t *d_poff = d_doff + n;
    int32_t *d_slen = (int32_t *)(d_poff + n);
    int32_t *d_cap = d_slen + n;
    int32_t *d_out = d_cap + n;
    int32_t *h_len = (int32_t *)ctx->h_len;
    hipStream_t st = ctx->stream;
    hipStream_t cq = nparts > 1 ? ctx->copyq : st;          /* one part: everything in order on the one queue, as ever */
    K4_HIP(ctx, hipMemcpyAsync(d_soff, h_soff.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    K4_HIP(ctx, hipMemcpyAsync(d_doff, h_doff.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    K4_HIP(ctx, hipMemcpyAsync(d_slen, srcLen, (size_t)n * 4, hipMemcpyHostToDevice, st));
    K4_HIP(ctx, hipMemcpyAsync(d_cap, h_cap.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    DictArgs ddev{nullptr, nullptr, nullptr, nullptr};
    std::vector<uint64_t> h_dictoff;
    std::vector<signed char> h_mode;
    if (hd && hd->dict) {
        /* dictionaries: stage their span; prefix-vs-external is decided on the HOST addresses
         * (LL64.dec.cs:531-541: dictStart + dictSize == dest) because staging breaks adjacency */
        uint64_t dlo = UINT64_MAX, dhi = 0;
        h_dictoff.assign((size_t)n, 0);
        h_mode.assign((size_t)n, 0);
        for (int64_t i = 0; i < n; i++) {
            const int32_t len = hd->len[i];
            if (len <= 0) continue;
            dlo = std::min(dlo, hd->off[i]);
            dhi = std::max(dhi, hd->off[i] + (uint64_t)len);
            h_mode[(size_t)i] = hd->dict + hd->off[i] + len == dst + dstOff[i] ? 1 : 2;
        }
        if (dlo != UINT64_MAX) {
            for (int64_t i = 0; i < n; i++) h_dictoff[(size_t)i] = hd->len[i] > 0 ? hd->off[i] - dlo : 0;
            const size_t dspan = (size_t)(dhi - dlo);
            const size_t need = dspan + 64 + (size_t)n * (8 + 4 + 1) + 64;
            if ((rc = grow(ctx, &ctx->d_dict, &ctx->d_dict_cap, need, false)) != K4LZ4_OK) return rc;
            uint8_t *base = ctx->d_dict;
            uint64_t *d_dictoff = (uint64_t *)(base + ((dspan + 63) & ~(size_t)63));
            int32_t *d_dictlen = (int32_t *)(d_dictoff + n);
            signed char *d_mode = (signed char *)(d_dictlen + n);
            K4_HIP(ctx, hipMemcpyAsync(base, hd->dict + dlo, dspan, hipMemcpyHostToDevice, st));
            K4_HIP(ctx, hipMemcpyAsync(d_dictoff, h_dictoff.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
            K4_HIP(ctx, hipMemcpyAsync(d_dictlen, hd->len, (size_t)n * 4, hipMemcpyHostToDevice, st));
            K4_HIP(ctx, hipMemcpyAsync(d_mode, h_mode.data(), (size_t)n, hipMemcpyHostToDevice, st));
            ddev = DictArgs{base, d_dictoff, d_dictlen, d_mode};
        }
    }
    lap("prepare");

    /* ---- The way back, part by part.  What each block produced is packed next to each other on the device (when that saves a
     * tenth or more of the transfer) and comes back through the pinned buffers in chunks cut at block boundaries; exactly
     * outLen[i] bytes land in each caller slot.  With more than one part this runs on a thread of its own (own queue, own
     * pinned buffers, own helper threads) while this thread stages and launches the later parts: PCIe carries both ways at
     * once, and so do the host's copies.  `launched` says how many parts have had their kernels and their ev_len enqueued. ---- */
    const bool raw_negative = (flags & K4LZ4_FLAG_ALLOW_COPY) && kind == KIND_ENCODE;
    std::vector<uint64_t> h_poff((size_t)n);
    std::vector<int32_t> stored((size_t)n);
    if (nparts > 1 && (rc = grow(ctx, &ctx->d_pack, &ctx->d_pack_cap, (size_t)dtotal + 64, false)) != K4LZ4_OK) return rc;
    hipStream_t dq = nparts > 1 ? ctx->dlq : st;
    std::mutex dm;
    std::condition_variable dcv;
    int launched = 0;                       /* guarded by dm */
    bool abandon = false;                   /* the staging side failed: the way back stops waiting */
    std::string dl_error;
    auto download_part = [&](int p, uint64_t &pack_base) -> int {
        const int64_t b0 = part_lo[p], b1 = part_hi[p], cnt = b1 - b0;
        K4_HIP(ctx, hipEventSynchronize(ctx->ev_len[p]));
        memcpy(outLen + b0, h_len + b0, (size_t)cnt * 4);
        uint64_t used = 0, cap_total = 0;
        for (int64_t i = b0; i < b1; i++) {
            const int32_t got = outLen[i];
            int32_t sv = (got < 0 && raw_negative) ? -got : got;   /* raw blocks come back as -length */
            if (sv < 0 || sv > h_cap[(size_t)i]) sv = 0;
            stored[(size_t)i] = sv;
            h_poff[(size_t)i] = pack_base + used;
            used += ((uint64_t)sv + 15u) & ~(uint64_t)15u;
            cap_total += ((uint64_t)h_cap[(size_t)i] + 15u) & ~(uint64_t)15u;
        }
        const uint8_t *d_from = ctx->d_dst;
        const uint64_t *from_off = h_doff.data() + b0;
        if (dq != st) K4_HIP(ctx, hipStreamWaitEvent(dq, ctx->ev_len[p], 0));    /* this part's kernels are through */
        if (cnt > 1 && used + (used >> 3) < cap_total) {
            int rc2;
            if (nparts == 1 && (rc2 = grow(ctx, &ctx->d_pack, &ctx->d_pack_cap, (size_t)dtotal + 64, false)) != K4LZ4_OK) return rc2;
            K4_HIP(ctx, hipMemcpyAsync(d_poff + b0, h_poff.data() + b0, (size_t)cnt * 8, hipMemcpyHostToDevice, dq));
            hipLaunchKernelGGL(k4::k4_compact_kernel, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, dq, ctx->d_dst, d_doff + b0, d_out + b0,
                               ctx->d_pack, d_poff + b0, (long long)cnt, raw_negative ? 1 : 0);
            K4_HIP(ctx, hipGetLastError());
            d_from = ctx->d_pack;
            from_off = h_poff.data() + b0;
        }
        pack_base += used;
        if (dst_reg) {
            int rc3 = K4LZ4_OK;
            if (direct_download(ctx, dst, dstOff + b0, d_from, from_off, stored.data() + b0, cnt, dq, &rc3)) return rc3;
        }
        return staged_download(ctx, dst, dstOff + b0, d_from, from_off, stored.data() + b0, cnt, dq, nparts > 1 ? pool_dl_of(ctx) : pool_of(ctx));
    };
    int dl_rc = K4LZ4_OK;
    auto download_all = [&]() {
        try {      /* it may run as a thread: nothing may escape (std::terminate), and its message is its own (tl_error) */
            (void)hipSetDevice(ctx->device);
            uint64_t pack_base = 0;
            for (int p = 0; p < nparts; p++) {
                {
                    std::unique_lock<std::mutex> lk(dm);
                    dcv.wait(lk, [&] { return launched > p || abandon; });
                    if (launched <= p) return;
                }
                const int r = download_part(p, pack_base);
                if (r != K4LZ4_OK) { dl_rc = r; dl_error = tl_error; return; }
            }
            if (hipStreamSynchronize(dq) != hipSuccess) { (void)hipGetLastError(); dl_rc = K4LZ4_E_HIP; dl_error = "download queue failed"; }
        } catch (const std::bad_alloc &) { dl_rc = K4LZ4_E_NOMEM; dl_error = "download: out of host memory";
        } catch (...) { dl_rc = K4LZ4_E_HIP; dl_error = "download: unexpected exception"; }
    };
    std::thread dl_thread;
    bool threaded = false;
    if (nparts > 1) {
        try { dl_thread = std::thread(download_all); threaded = true; } catch (...) { threaded = false; }   /* no thread to be had: in line, below */
    }
    auto finish = [&](int up_rc) -> int {   /* every way out of the staging loop comes through here */
        if (threaded) {
            { std::lock_guard<std::mutex> g(dm); if (up_rc != K4LZ4_OK) abandon = true; }
            dcv.notify_all();
            dl_thread.join();
        } else if (up_rc == K4LZ4_OK) {
            { std::lock_guard<std::mutex> g(dm); launched = nparts; }
            download_all();
        }
        if (up_rc != K4LZ4_OK) return up_rc;
        if (dl_rc != K4LZ4_OK) return fail(ctx, dl_rc, dl_error);
        return K4LZ4_OK;
    };

    /* ---- up and launch, part by part: the host stages part k + 1 while part k's kernels run ---- */
    for (int p = 0; p < nparts; p++) {
        const int64_t b0 = part_lo[p], b1 = part_hi[p], cnt = b1 - b0;
        if (packed) {
            if ((rc = staged_upload_packed(ctx, ctx->d_src, src, srcOff, srcLen, b0, b1, h_soff.data(), cq)) != K4LZ4_OK) return finish(rc);
        } else {
            /* ascending blocks (or one part): this part's stretch of the span */
            uint64_t a = UINT64_MAX, e = 0;
            for (int64_t i = b0; i < b1; i++)
                if (srcLen[i] > 0) { a = std::min(a, h_soff[(size_t)i]); e = std::max(e, h_soff[(size_t)i] + (uint64_t)srcLen[i]); }
            if (nparts == 1) { a = 0; e = span; }
            if (a != UINT64_MAX && e > a) {
                if (src_reg) {
                    const hipError_t ue = hipMemcpyAsync(ctx->d_src + a, src + lo + a, (size_t)(e - a), hipMemcpyHostToDevice, cq);
                    if (ue != hipSuccess) return finish(hip_fail(ctx, ue, "hipMemcpyAsync"));
                } else if ((rc = staged_upload(ctx, ctx->d_src + a, src + lo + a, (size_t)(e - a), cq)) != K4LZ4_OK) return finish(rc);
            }
        }
        hipError_t he = hipSuccess;
        if (cq != st) {
            he = hipEventRecord(ctx->ev_up[p], cq);
            if (he == hipSuccess) he = hipStreamWaitEvent(st, ctx->ev_up[p], 0);
            if (he != hipSuccess) return finish(hip_fail(ctx, he, "hipEventRecord"));
        }
        DictArgs dpart = ddev;
        /* The kernels' target is the context's staging buffer, and a hostile stream with a match offset of 0 "copies output bytes onto
         * themselves" (LL64.dec.cs:408-418): those bytes stay what the target held.  Decode-like calls therefore start from a zeroed
         * target (this part's range of it; a memset on the device, ~0.1 ms per 256 MB beside the milliseconds of PCIe), so that such
         * bytes are zeros and never what an earlier call of this context left there.  Not in the kernels: carrying the
         * case through decode_block cost the pair kernel five spilled VGPRs and the single-wave kernels a wave per SIMD. */
        if (kind != KIND_ENCODE && kind != KIND_PICKLE) {
            const uint64_t lo_d = h_doff[(size_t)b0];
            const uint64_t hi_d = b1 < n ? h_doff[(size_t)b1] : dtotal;
            if (hi_d > lo_d) { const hipError_t me = hipMemsetAsync(ctx->d_dst + lo_d, 0, (size_t)(hi_d - lo_d), st); if (me != hipSuccess) return finish(hip_fail(ctx, me, "hipMemsetAsync")); }
        }
        rc = launch(ctx, kind, ctx->d_src, d_soff + b0, d_slen + b0, ctx->d_dst, d_doff + b0, d_cap + b0, d_out + b0, cnt, level, flags, st,
                    &dpart, srcLen + b0);
        if (rc != K4LZ4_OK) return finish(rc);
        he = hipMemcpyAsync(h_len + b0, d_out + b0, (size_t)cnt * 4, hipMemcpyDeviceToHost, st);
        if (he == hipSuccess) he = hipEventRecord(ctx->ev_len[p], st);
        if (he != hipSuccess) return finish(hip_fail(ctx, he, "hipMemcpyAsync"));
        { std::lock_guard<std::mutex> g(dm); launched = p + 1; }
        dcv.notify_all();
        lap(p == 0 ? "up+launch" : "up+launch+");
    }
    rc = finish(K4LZ4_OK);
    lap("down");
    if (rc != K4LZ4_OK) return rc;
    K4_HIP(ctx, hipStreamSynchronize(st));
    if (cq != st) K4_HIP(ctx, hipStreamSynchronize(cq));
    return take_device_status(ctx);
}

int run_device(k4lz4_ctx *ctx, Kind kind, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
               uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n, int level,
               int flags, void *stream, const DictArgs *dd = nullptr)
{
    int rc = check_batch_args(ctx, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n);
    if (rc != K4LZ4_OK) return rc;
    if (kind == KIND_ENCODE || kind == KIND_PICKLE) {
        rc = check_level(ctx, level);
        if (rc != K4LZ4_OK) return rc;
    }
    if (n == 0) return K4LZ4_OK;
    K4_HIP(ctx, hipSetDevice(ctx->device));
    return launch(ctx, kind, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, level, flags, (hipStream_t)stream, dd);
}

struct CtxDeleter { void operator()(k4lz4_ctx *c) const { k4lz4_ctx_destroy(c); } };
thread_local std::unique_ptr<k4lz4_ctx, CtxDeleter> tl_ctx;

k4lz4_ctx *implicit_ctx()
{
    if (!tl_ctx) {
        k4lz4_ctx *c = nullptr;
        if (k4lz4_ctx_create(&c, -1) != K4LZ4_OK) return nullptr;
        tl_ctx.reset(c);
    }
    return tl_ctx.get();
}

/* batch of one with LLxx-level returns */
int single(Kind kind, const uint8_t *src, uint8_t *dst, int srcLen, int dstCap, int level, int accel = 1, int extra_flags = 0,
           const uint8_t *dict = nullptr, int dictLen = 0)
{
    tl_status = K4LZ4_OK;
    k4lz4_ctx *ctx = implicit_ctx();
    if (!ctx) { tl_status = K4LZ4_E_NO_DEVICE; return kind == KIND_DECODE ? -1 : 0; }
    if (!src || !dst) { tl_status = fail(ctx, K4LZ4_E_ARG, "NULL buffer"); return kind == KIND_DECODE ? -1 : 0; }
    const uint64_t off = 0;
    int32_t slen = srcLen, cap = dstCap, out = 0;
    ctx->accel = accel < 1 ? 1 : accel;
    int32_t dlen = dictLen;
    const DictArgs hd{dict, &off, &dlen, nullptr};
    const int rc = run_host(ctx, kind, src, &off, &slen, dst, &off, &cap, &out, 1, level, K4LZ4_FLAG_RAW_RETURN | extra_flags,
                            dict && dictLen > 0 ? &hd : nullptr);
    ctx->accel = 1;
    tl_status = rc;
    if (rc != K4LZ4_OK) return kind == KIND_DECODE ? -1 : 0;
    return out;
}

}  // namespace

extern "C" {

int k4lz4_version(void) { return K4LZ4_VERSION; }

void k4lz4_set_enforce32(int on) { g_enforce32.store(on ? 1 : 0, std::memory_order_relaxed); }
int k4lz4_get_enforce32(void) { return g_enforce32.load(std::memory_order_relaxed); }

int k4lz4_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int k4lz4_ctx_create(k4lz4_ctx **out, int device)
{
    if (!out) return fail(nullptr, K4LZ4_E_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, K4LZ4_E_NO_DEVICE, "no HIP device visible (libk4lz4 has no CPU fallback)");
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); device = 0; }
    }
    if (device >= n) return fail(nullptr, K4LZ4_E_ARG, "device index out of range");
    hipDeviceProp_t prop;
    K4_HIP(nullptr, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, K4LZ4_E_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", libk4lz4 is built for gfx950 only");
    k4lz4_ctx *ctx = new (std::nothrow) k4lz4_ctx();
    if (!ctx) return fail(nullptr, K4LZ4_E_NOMEM, "out of host memory");
    ctx->device = device;
    ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_status, 64);
    if (e == hipSuccess) e = hipMemset(ctx->d_status, 0, 64);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_pace, k4::PACE_BYTES);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->copyq, hipStreamNonBlocking);
    if (e == hipSuccess) {
        /* the later segments of cut blocks are the longest jobs of their launch and others wait for their cuts: the queue they
         * are dispatched from comes first when the chip is full (K4LZ4_SEG_PRIO=0: an ordinary queue) */
        int least = 0, greatest = 0;
        const char *pe = getenv("K4LZ4_SEG_PRIO");
        if ((!pe || atoi(pe) != 0) && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
            e = hipStreamCreateWithPriority(&ctx->aux2, hipStreamNonBlocking, greatest);
        else { (void)hipGetLastError(); e = hipStreamCreateWithFlags(&ctx->aux2, hipStreamNonBlocking); }
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_join2, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_busy, hipEventDisableTiming);
    for (int b = 0; b < 2 && e == hipSuccess; b++) {
        e = hipEventCreateWithFlags(&ctx->ev_in[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_out[b], hipEventDisableTiming);
    }
    for (int b = 0; b < MAX_PARTS && e == hipSuccess; b++) {
        e = hipEventCreateWithFlags(&ctx->ev_up[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_len[b], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->dlq, hipStreamNonBlocking);
    if (const char *pct = getenv("K4LZ4_SPLIT_PCT")) { const int v = atoi(pct); ctx->split_pct = v < 1 ? 1 : (v > 100 ? 100 : v); }
    ctx->no_pair = getenv("K4LZ4_NO_PAIR") != nullptr;
    ctx->prof_gtab = getenv("K4LZ4_PROF_GTAB") != nullptr;
    ctx->use_pace = getenv("K4LZ4_NO_PACE") == nullptr;
    if (const char *e = getenv("K4LZ4_PACE_MIN")) ctx->pace_min_per_cu = std::max(0, atoi(e));
    if (const char *e = getenv("K4LZ4_DIRECT_SPAN_PCT")) ctx->direct_span_pct = std::max(100, atoi(e));
    if (const char *e = getenv("K4LZ4_NO_DIRECT")) ctx->no_direct = atoi(e) != 0;
    if (const char *e = getenv("K4LZ4_LDS_FLOOR")) ctx->lds_floor_per_cu = std::max(0, std::min(8, atoi(e)));
    if (const char *e = getenv("K4LZ4_COST_PCT")) ctx->cost_pct = std::max(1, std::min(99, atoi(e)));
    if (const char *e = getenv("K4LZ4_DEC_PARTS")) ctx->dec_parts = std::max(2, std::min(MAX_PARTS, atoi(e)));
    if (const char *e = getenv("K4LZ4_DEC_PARTS_DIRECT")) ctx->dec_parts_direct = std::max(2, std::min(MAX_PARTS, atoi(e)));
    if (const char *e = getenv("K4LZ4_HOP2_MAX")) ctx->hop2_max_per_cu = std::max(0, atoi(e));
    ctx->hc_chain_parts = getenv("K4LZ4_HC_CHAIN_OLD") == nullptr;
    ctx->hc_cand_lds = getenv("K4LZ4_HC_CAND_MEM") == nullptr;
    if (const char *e = getenv("K4LZ4_HC_CAND_DYNLDS")) ctx->hc_cand_dynlds = std::max(0, std::min(65536, atoi(e)));
    if (const char *e = getenv("K4LZ4_HC_SEGS")) { const int v = atoi(e); ctx->hc_segs = v >= 4 ? 4 : (v >= 2 ? 2 : (v == 1 ? 1 : 0)); }
    if (const char *e = getenv("K4LZ4_HC_MEM_PCT")) ctx->hc_mem_pct = std::max(0, std::min(100, atoi(e)));
    ctx->use_segments = getenv("K4LZ4_NO_SEGMENTS") == nullptr;
    if (const char *e = getenv("K4LZ4_SEG_MIN")) ctx->seg_min = (uint32_t)std::max(65536 + 4096, atoi(e));
    if (const char *e = getenv("K4LZ4_SEG_TARGET")) ctx->seg_target = ctx->seg_target_max = (uint32_t)std::max(8192, atoi(e));
    if (const char *e = getenv("K4LZ4_SEG_TARGET_MAX")) ctx->seg_target_max = (uint32_t)std::max(8192, atoi(e));
    if (const char *e = getenv("K4LZ4_SEG_WARM")) ctx->seg_warm = (uint32_t)std::max(0, atoi(e));
    if (const char *e = getenv("K4LZ4_SEG_SPIN_MAX")) ctx->seg_spin_max = (uint32_t)std::max(0, atoi(e));
    if (const char *e = getenv("K4LZ4_SEG_DIV")) ctx->seg_div = (uint32_t)std::max(0, atoi(e));
    if (const char *e = getenv("K4LZ4_PICKLE_SPLIT_MIN")) ctx->pickle_split_min = std::max(0, atoi(e));
    if (const char *e = getenv("K4LZ4_STAGE_THREADS")) ctx->stage_threads = std::max(0, std::min(63, atoi(e) - 1));
    ctx->use_parse = getenv("K4LZ4_NO_PARSE") == nullptr;
    ctx->parse_queue = getenv("K4LZ4_PARSE_QUEUE") != nullptr;
    ctx->parse_persist = getenv("K4LZ4_NO_PERSIST") == nullptr;
    ctx->parse_big = getenv("K4LZ4_NO_PARSE_BIG") == nullptr;
    ctx->parse_seg = getenv("K4LZ4_NO_PARSE_SEG") == nullptr;
    ctx->hc_records = getenv("K4LZ4_NO_HC_RECORDS") == nullptr;
    ctx->parse_pcost = getenv("K4LZ4_PCOST") != nullptr;
    ctx->parse_migrate = getenv("K4LZ4_NO_MIGRATE") == nullptr;
    ctx->parse_inline_emit = getenv("K4LZ4_NO_INLINE_EMIT") == nullptr;
    if (const char *e = getenv("K4LZ4_PARSE_WAVES")) ctx->parse_waves = std::max(1, std::min(k4::PARSE_MAX_WAVES, atoi(e)));
    ctx->trace = getenv("K4LZ4_TRACE") != nullptr;
    if (e != hipSuccess) { delete ctx; return hip_fail(nullptr, e, "hipStreamCreate"); }
    *out = ctx;
    return K4LZ4_OK;
}

void k4lz4_ctx_destroy(k4lz4_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); }
    if (ctx->aux2) { (void)hipStreamSynchronize(ctx->aux2); (void)hipStreamDestroy(ctx->aux2); }
    if (ctx->ev_join2) (void)hipEventDestroy(ctx->ev_join2);
    if (ctx->d_seg) (void)hipFree(ctx->d_seg);
    if (ctx->d_seg_first) (void)hipFree(ctx->d_seg_first);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_busy) (void)hipEventDestroy(ctx->ev_busy);
    for (int b = 0; b < 2; b++) {
        if (ctx->ev_in[b]) (void)hipEventDestroy(ctx->ev_in[b]);
        if (ctx->ev_out[b]) (void)hipEventDestroy(ctx->ev_out[b]);
        if (ctx->h_in[b]) (void)hipHostFree(ctx->h_in[b]);
        if (ctx->h_out[b]) (void)hipHostFree(ctx->h_out[b]);
    }
    if (ctx->d_pack) (void)hipFree(ctx->d_pack);
    if (ctx->h_len) (void)hipHostFree(ctx->h_len);
    for (int b = 0; b < MAX_PARTS; b++) {
        if (ctx->ev_up[b]) (void)hipEventDestroy(ctx->ev_up[b]);
        if (ctx->ev_len[b]) (void)hipEventDestroy(ctx->ev_len[b]);
    }
    if (ctx->copyq) (void)hipStreamDestroy(ctx->copyq);
    if (ctx->dlq) (void)hipStreamDestroy(ctx->dlq);
    delete ctx->pool;
    delete ctx->pool_dl;
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->d_pace) (void)hipFree(ctx->d_pace);
    if (ctx->d_gtab) (void)hipFree(ctx->d_gtab);
    if (ctx->d_parse) (void)hipFree(ctx->d_parse);
    if (ctx->d_dict) (void)hipFree(ctx->d_dict);
    if (ctx->d_src) (void)hipFree(ctx->d_src);
    if (ctx->d_dst) (void)hipFree(ctx->d_dst);
    if (ctx->d_meta) (void)hipFree(ctx->d_meta);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->d_sched) (void)hipFree(ctx->d_sched);
    if (ctx->d_hc_hash) (void)hipFree(ctx->d_hc_hash);
    if (ctx->d_hc_work) (void)hipFree(ctx->d_hc_work);
    if (ctx->d_hc_meta) (void)hipFree(ctx->d_hc_meta);
    if (ctx->d_pk_meta) (void)hipFree(ctx->d_pk_meta);
    delete ctx;
}

const char *k4lz4_last_error(const k4lz4_ctx *ctx) { return ctx ? ctx->error.c_str() : tl_error.c_str(); }

/* measured floors of a device call (profiles/r5_block_count_scaling.txt: a 64 KiB text block alone 2.0 ms to encode -- 2.9 ms with the
 * one-kernel encoder of rounds 1-4 --, 0.55 ms to decode; r18_stamp.txt: 8.7 ms at HC level 3): the time of ONE block on its wavefront, scaled with the block length (a wavefront's
 * time per block is linear in its length), never below the launch + synchronisation cost of a call */
int64_t k4lz4_recommended_min_batch(int kind, int32_t blockBytes, double hostGiBs)
{
    if (kind < 0 || kind > 2 || blockBytes <= 0) return K4LZ4_E_ARG;
    static const double floor_ms_64k[3] = {2.0, 0.55, 2.9};      /* (HC level 3: 8.7 until round 6; profiles/r6_hc_small_batches.txt) */
    static const double box_host_GiBs[3] = {32.0, 35.0, 2.2};
    const double host = hostGiBs > 0.0 ? hostGiBs : box_host_GiBs[kind];
    double floor_ms = floor_ms_64k[kind] * (double)blockBytes / 65536.0;
    /* (fast encode of blocks of 65 547 bytes and more -- byU32 tables -- is the one-kernel encoder's: its floor is the older, higher one) */
    if (kind == 0 && blockBytes >= k4::LIMIT_64K) floor_ms = 2.9 * (double)blockBytes / 65536.0;
    if (floor_ms < 0.05) floor_ms = 0.05;
    const double blocks = floor_ms * 1e-3 * host * 1073741824.0 / (double)blockBytes;
    return blocks < 1.0 ? 1 : (int64_t)(blocks + 0.999);
}

int k4lz4_ctx_device(const k4lz4_ctx *ctx) { return ctx ? ctx->device : -1; }

int k4lz4_synchronize(k4lz4_ctx *ctx, void *stream)
{
    if (!ctx) return fail(nullptr, K4LZ4_E_ARG, "ctx is NULL");
    K4_HIP(ctx, hipSetDevice(ctx->device));
    K4_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return take_device_status(ctx);
}

int k4lz4_host_register(void *ptr, size_t bytes)
{
    if (!ptr || !bytes) { tl_error = "k4lz4_host_register: empty range"; return K4LZ4_E_ARG; }
    /* one lock scope for check + insert: the range is entered before the (slow) pinning so that a second thread registering an
     * overlapping range is refused; if pinning fails the entry is taken out again */
    const uintptr_t a = (uintptr_t)ptr;
    try {
        std::lock_guard<std::mutex> g(g_reg_mu);
        for (const auto &r : g_reg)
            if (a < r.first + r.second && r.first < a + bytes) { tl_error = "k4lz4_host_register: overlaps a registered range"; return K4LZ4_E_ARG; }
        g_reg.push_back(RegRange{a, bytes, false});
    } catch (const std::bad_alloc &) { tl_error = "k4lz4_host_register: out of host memory"; return K4LZ4_E_NOMEM; }
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> g(g_reg_mu);
            auto it = std::find_if(g_reg.begin(), g_reg.end(), [&](const RegRange &r) { return r.first == a && r.second == bytes && !r.pinned; });
            if (it != g_reg.end()) g_reg.erase(it);
        }
        tl_error = std::string("hipHostRegister: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? K4LZ4_E_NOMEM : K4LZ4_E_HIP;
    }
    {   /* only now do host-pointer calls treat the range as page-locked, and only now can it be unregistered */
        std::lock_guard<std::mutex> g(g_reg_mu);
        auto it = std::find_if(g_reg.begin(), g_reg.end(), [&](const RegRange &r) { return r.first == a && r.second == bytes && !r.pinned; });
        if (it != g_reg.end()) it->pinned = true;
    }
    return K4LZ4_OK;
}

int k4lz4_host_unregister(void *ptr)
{
    {
        std::lock_guard<std::mutex> g(g_reg_mu);
        auto it = std::find_if(g_reg.begin(), g_reg.end(), [&](const RegRange &r) { return r.first == (uintptr_t)ptr && r.pinned; });      /* (a range still being pinned is not there yet) */
        if (it == g_reg.end()) { tl_error = "k4lz4_host_unregister: not a registered range"; return K4LZ4_E_ARG; }
        g_reg.erase(it);
    }
    const hipError_t e = hipHostUnregister(ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); tl_error = std::string("hipHostUnregister: ") + hipGetErrorString(e); return K4LZ4_E_HIP; }
    return K4LZ4_OK;
}

int k4lz4_ctx_reserve_hc(k4lz4_ctx *ctx, int64_t totalSrcBytes, int32_t longestBlock)
{
    if (!ctx) return fail(nullptr, K4LZ4_E_ARG, "ctx is NULL");
    if (totalSrcBytes < 0 || longestBlock < 0) return fail(ctx, K4LZ4_E_ARG, "negative reservation");
    ctx->hc_res_total = (uint64_t)totalSrcBytes;
    ctx->hc_res_longest = (uint32_t)longestBlock;
    return K4LZ4_OK;
}

int k4lz4_selftest_chains(k4lz4_ctx *ctx, int waves, int rounds, uint32_t seed, uint32_t mismatches[3])
{
    if (!ctx || !mismatches || waves <= 0 || rounds <= 0) return fail(ctx, K4LZ4_E_ARG, "selftest: bad arguments");
    K4_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t *d_res = nullptr;
    K4_HIP(ctx, hipMalloc((void **)&d_res, 3 * sizeof(uint32_t)));
    hipError_t e = hipMemsetAsync(d_res, 0, 3 * sizeof(uint32_t), ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k4::k4_chain_selftest_kernel, dim3((unsigned)waves), dim3(64), 0, ctx->stream, seed, rounds, d_res);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(mismatches, d_res, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_res);
    K4_HIP(ctx, e);
    return K4LZ4_OK;
}

int k4lz4_last_status(void) { return tl_status; }

int k4lz4_compress_bound(int n) { return n > 0x7E000000 ? 0 : n + n / 255 + 16; }

int k4lz4_compress_fast(const uint8_t *src, uint8_t *dst, int srcLen, int dstCap, int acceleration)
{
    return single(KIND_ENCODE, src, dst, srcLen, dstCap, K4LZ4_L00_FAST, acceleration);
}

int k4lz4_compress_hc(const uint8_t *src, uint8_t *dst, int srcLen, int dstCap, int level)
{
    /* LZ4Level has no value between L00_FAST and L03_HC and LZ4Codec never passes one (LZ4Codec.cs:48-50); the upstream
     * meaning of levels below 3 (default 9 / two attempts, LL64.high.cs:1155-1160) is not implemented: refuse, do not guess */
    if (level < K4LZ4_L03_HC) {
        tl_status = fail(nullptr, K4LZ4_E_ARG, "k4lz4_compress_hc: level must be L03_HC (3) or higher");
        return 0;
    }
    return single(KIND_ENCODE, src, dst, srcLen, dstCap, level);
}

int k4lz4_decompress_safe(const uint8_t *src, uint8_t *dst, int srcLen, int dstCap)
{
    return single(KIND_DECODE, src, dst, srcLen, dstCap, 0);
}

int k4lz4_decompress_safe_partial(const uint8_t *src, uint8_t *dst, int srcLen, int targetLen)
{
    return single(KIND_DECODE, src, dst, srcLen, targetLen, 0, 1, K4LZ4_FLAG_PARTIAL);
}

int k4lz4_encode_batch(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                       const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n, int level, int flags)
{
    return run_host(ctx, KIND_ENCODE, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, level, flags);
}

int k4lz4_decompress_safe_using_dict(const uint8_t *src, uint8_t *dst, int srcLen, int dstCap, const uint8_t *dict, int dictLen)
{
    return single(KIND_DECODE, src, dst, srcLen, dstCap, 0, 1, 0, dict, dictLen);
}

int k4lz4_decode_dict_batch(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                            const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n, int flags,
                            const uint8_t *dict, const uint64_t *dictOff, const int32_t *dictLen)
{
    if (dict && n > 0 && (!dictOff || !dictLen)) return fail(ctx, K4LZ4_E_ARG, "NULL dictionary offsets/lengths");
    const DictArgs hd{dict, dictOff, dictLen, nullptr};
    return run_host(ctx, KIND_DECODE, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, 0, flags, dict ? &hd : nullptr);
}

int k4lz4_decode_dict_batch_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                                   uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n,
                                   int flags, const uint8_t *dict, const uint64_t *dictOff, const int32_t *dictLen,
                                   void *stream)
{
    if (dict && n > 0 && (!dictOff || !dictLen)) return fail(ctx, K4LZ4_E_ARG, "NULL dictionary offsets/lengths");
    const DictArgs dd{dict, dictOff, dictLen, nullptr};
    return run_device(ctx, KIND_DECODE, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, 0, flags, stream, dict ? &dd : nullptr);
}

int k4lz4_decode_batch(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                       const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n, int flags)
{
    return run_host(ctx, KIND_DECODE, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, 0, flags);
}

int k4lz4_encode_batch_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                              uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n,
                              int level, int flags, void *stream)
{
    return run_device(ctx, KIND_ENCODE, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, level, flags, stream);
}

int k4lz4_decode_batch_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                              uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n,
                              int flags, void *stream)
{
    return run_device(ctx, KIND_DECODE, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, 0, flags, stream);
}

int k4lz4_pickle_bound(int srcLen) { return srcLen <= 0 ? 0 : 1 + 4 + srcLen; }

int k4lz4_unpickle_size(const uint8_t *p, int len)
{
    /* LZ4Pickler.unpickle.cs:131-148 */
    if (len == 0) return 0;
    if (!p || len < 0) return -1;
    if ((p[0] & 7) != 0) return -1;
    const int code = (p[0] >> 6) & 3;
    const int sod = code == 3 ? 4 : code;
    const int data_len = len - 1 - sod;
    if (data_len < 0) return -1;
    uint32_t diff = 0;
    for (int i = 0; i < sod; i++) diff |= (uint32_t)p[1 + i] << (8 * i);
    const int r = (int)((uint32_t)data_len + diff);
    return r < 0 ? -1 : r;
}

int k4lz4_pickle_batch(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                       const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n, int level, int flags)
{
    return run_host(ctx, KIND_PICKLE, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, level, flags);
}

int k4lz4_unpickle_batch(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen, uint8_t *dst,
                         const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n, int flags)
{
    return run_host(ctx, KIND_UNPICKLE, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, 0, flags);
}

int k4lz4_pickle_batch_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                              uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n,
                              int level, int flags, void *stream)
{
    return run_device(ctx, KIND_PICKLE, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, level, flags, stream);
}

int k4lz4_unpickle_batch_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                                uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap, int32_t *outLen, int64_t n,
                                int flags, void *stream)
{
    return run_device(ctx, KIND_UNPICKLE, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n, 0, flags, stream);
}

int k4lz4_profile_batch_device(k4lz4_ctx *ctx, int decode, const uint8_t *src, const uint64_t *srcOff,
                               const int32_t *srcLen, uint8_t *dst, const uint64_t *dstOff, const int32_t *dstCap,
                               int32_t *outLen, int64_t n, uint64_t *counters, void *stream)
{
    if (!ctx || !counters) return fail(ctx, K4LZ4_E_ARG, "bad argument");
    ctx->prof = (unsigned long long *)counters;
    ctx->prof_pair = decode == 2;        /* 2: the two-waves-per-block decoder, 32 counters per block (parsing wave, copying wave) */
    ctx->prof_stamp = decode >= 4;       /* 4 / 5: the ordinary encode / decode kernels, which only record [8] start, [9] end, [10] HW_ID, [11] kernel */
    const int rc = run_device(ctx, (decode && decode != 4) ? KIND_DECODE : KIND_ENCODE, src, srcOff, srcLen, dst, dstOff, dstCap, outLen, n,
                              K4LZ4_L00_FAST, 0, stream);
    ctx->prof = nullptr;
    ctx->prof_stamp = false;
    return rc;
}

/* ---- frame layer -------------------------------------------------------------------------------- */
int k4lz4_xxh32_batch_device(k4lz4_ctx *ctx, const uint8_t *data, const uint64_t *off, const uint64_t *len,
                             uint32_t *out, int64_t n, uint32_t seed, void *stream)
{
    if (!ctx) return fail(nullptr, K4LZ4_E_ARG, "ctx is NULL");
    if (n < 0 || (n > 0 && (!data || !off || !len || !out))) return fail(ctx, K4LZ4_E_ARG, "bad argument");
    if (n == 0) return K4LZ4_OK;
    K4_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t per_launch = (int64_t)1 << 24;
    for (int64_t first = 0; first < n; first += per_launch) {
        const int64_t cnt = std::min<int64_t>(per_launch, n - first);
        k4::HashArgs a{data, off + first, len + first, out + first, cnt, seed};
        hipLaunchKernelGGL(k4::k4_xxh32_kernel, dim3((unsigned)((cnt * 4 + k4::XXH_THREADS - 1) / k4::XXH_THREADS)),
                           dim3(k4::XXH_THREADS), 0, (hipStream_t)stream, a);
    }
    K4_HIP(ctx, hipGetLastError());
    return K4LZ4_OK;
}

int k4lz4_xxh32_batch(k4lz4_ctx *ctx, const uint8_t *data, const uint64_t *off, const uint64_t *len, uint32_t *out,
                      int64_t n, uint32_t seed)
{
    if (!ctx) return fail(nullptr, K4LZ4_E_ARG, "ctx is NULL");
    if (n < 0 || (n > 0 && (!data || !off || !len || !out))) return fail(ctx, K4LZ4_E_ARG, "bad argument");
    if (n == 0) return K4LZ4_OK;
    K4_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t lo = UINT64_MAX, hi = 0;
    for (int64_t i = 0; i < n; i++)
        if (len[i]) { lo = std::min(lo, off[i]); hi = std::max(hi, off[i] + len[i]); }
    if (lo == UINT64_MAX) { lo = 0; hi = 0; }
    std::vector<uint64_t> h_off((size_t)n);
    for (int64_t i = 0; i < n; i++) h_off[(size_t)i] = len[i] ? off[i] - lo : 0;
    const size_t span = (size_t)(hi - lo);
    int rc;
    if ((rc = grow(ctx, &ctx->d_src, &ctx->d_src_cap, span + 64, false)) != K4LZ4_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_meta, &ctx->d_meta_cap, (size_t)n * 20 + 64, false)) != K4LZ4_OK) return rc;
    uint64_t *d_off = (uint64_t *)ctx->d_meta, *d_len = d_off + n;
    uint32_t *d_out = (uint32_t *)(d_len + n);
    hipStream_t st = ctx->stream;
    if (span) K4_HIP(ctx, hipMemcpyAsync(ctx->d_src, data + lo, span, hipMemcpyHostToDevice, st));
    K4_HIP(ctx, hipMemcpyAsync(d_off, h_off.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    K4_HIP(ctx, hipMemcpyAsync(d_len, len, (size_t)n * 8, hipMemcpyHostToDevice, st));
    rc = k4lz4_xxh32_batch_device(ctx, ctx->d_src, d_off, d_len, d_out, n, seed, st);
    if (rc != K4LZ4_OK) return rc;
    K4_HIP(ctx, hipMemcpyAsync(out, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    K4_HIP(ctx, hipStreamSynchronize(st));
    return K4LZ4_OK;
}

int k4lz4_decode_chain_batch_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *blkOff, const uint32_t *blkLen,
                                    const uint64_t *firstBlk, const uint32_t *nBlk, const int32_t *blockSize,
                                    const uint8_t *chained, uint8_t *dst, const uint64_t *dstOff, const uint64_t *dstCap,
                                    int64_t *outLen, int64_t nStreams, void *stream)
{
    if (!ctx) return fail(nullptr, K4LZ4_E_ARG, "ctx is NULL");
    if (nStreams < 0 || (nStreams > 0 && (!src || !blkOff || !blkLen || !firstBlk || !nBlk || !blockSize || !chained || !dst ||
                                          !dstOff || !dstCap || !outLen)))
        return fail(ctx, K4LZ4_E_ARG, "bad argument");
    if (nStreams == 0) return K4LZ4_OK;
    K4_HIP(ctx, hipSetDevice(ctx->device));
    k4::ChainArgs a{src, blkOff, blkLen, firstBlk, nBlk, blockSize, chained, dst, dstOff, dstCap, (long long *)outLen, nStreams, ctx->d_status};
    const unsigned grid = (unsigned)((nStreams + k4::DECODE_WAVES_PER_WG - 1) / k4::DECODE_WAVES_PER_WG);
    if (nStreams <= 16 * (int64_t)ctx->cu_count && !ctx->no_pair) {  /* room for two waves per stream */
        hipLaunchKernelGGL(k4::k4_decode_chain_pair_kernel, dim3((unsigned)((nStreams + k4::DECODE_PAIRS_PER_WG - 1) / k4::DECODE_PAIRS_PER_WG)),
                           dim3(128 * k4::DECODE_PAIRS_PER_WG), 0, (hipStream_t)stream, a);
    } else
        hipLaunchKernelGGL(k4::k4_decode_chain_kernel, dim3(grid), dim3(64 * k4::DECODE_WAVES_PER_WG), 0, (hipStream_t)stream, a);
    K4_HIP(ctx, hipGetLastError());
    return K4LZ4_OK;
}

int k4lz4_decode_chain_batch(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *blkOff, const uint32_t *blkLen,
                             int64_t nBlocks, const uint64_t *firstBlk, const uint32_t *nBlk, const int32_t *blockSize,
                             const uint8_t *chained, uint8_t *dst, const uint64_t *dstOff, const uint64_t *dstCap,
                             int64_t *outLen, int64_t nStreams)
{
    if (!ctx) return fail(nullptr, K4LZ4_E_ARG, "ctx is NULL");
    if (nStreams < 0 || nBlocks < 0 ||
        (nStreams > 0 && (!src || !firstBlk || !nBlk || !blockSize || !chained || !dst || !dstOff || !dstCap || !outLen)) ||
        (nBlocks > 0 && (!blkOff || !blkLen)))
        return fail(ctx, K4LZ4_E_ARG, "bad argument");
    if (nStreams == 0) return K4LZ4_OK;
    K4_HIP(ctx, hipSetDevice(ctx->device));
    uint64_t lo = UINT64_MAX, hi = 0;
    for (int64_t i = 0; i < nBlocks; i++) {
        const uint64_t l = blkLen[i] & 0x7fffffffu;
        if (l) { lo = std::min(lo, blkOff[i]); hi = std::max(hi, blkOff[i] + l); }
    }
    if (lo == UINT64_MAX) { lo = 0; hi = 0; }
    std::vector<uint64_t> h_boff((size_t)std::max<int64_t>(nBlocks, 1)), h_doff((size_t)nStreams);
    for (int64_t i = 0; i < nBlocks; i++) h_boff[(size_t)i] = (blkLen[i] & 0x7fffffffu) ? blkOff[i] - lo : 0;
    uint64_t dtotal = 0;
    for (int64_t i = 0; i < nStreams; i++) {
        if (firstBlk[i] + nBlk[i] > (uint64_t)nBlocks) return fail(ctx, K4LZ4_E_ARG, "stream refers to blocks outside the block table");
        h_doff[(size_t)i] = dtotal;
        dtotal += (dstCap[i] + 15u) & ~(uint64_t)15u;
    }
    const size_t span = (size_t)(hi - lo);
    const size_t meta = (size_t)nBlocks * 12 + (size_t)nStreams * (8 + 4 + 4 + 1 + 8 + 8 + 8) + 256;
    int rc;
    if ((rc = grow(ctx, &ctx->d_src, &ctx->d_src_cap, span + 64, false)) != K4LZ4_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_dst, &ctx->d_dst_cap, (size_t)dtotal + 64, false)) != K4LZ4_OK) return rc;
    if ((rc = grow(ctx, &ctx->d_meta, &ctx->d_meta_cap, meta, false)) != K4LZ4_OK) return rc;
    if ((rc = grow(ctx, &ctx->h_stage, &ctx->h_stage_cap, (size_t)dtotal + 64, true)) != K4LZ4_OK) return rc;
    uint8_t *m = ctx->d_meta;
    uint64_t *d_boff = (uint64_t *)m;            m += (size_t)nBlocks * 8;
    uint64_t *d_first = (uint64_t *)m;           m += (size_t)nStreams * 8;
    uint64_t *d_doff = (uint64_t *)m;            m += (size_t)nStreams * 8;
    uint64_t *d_dcap = (uint64_t *)m;            m += (size_t)nStreams * 8;
    int64_t *d_out = (int64_t *)m;               m += (size_t)nStreams * 8;
    uint32_t *d_blen = (uint32_t *)m;            m += (size_t)nBlocks * 4;
    uint32_t *d_nblk = (uint32_t *)m;            m += (size_t)nStreams * 4;
    int32_t *d_bsize = (int32_t *)m;             m += (size_t)nStreams * 4;
    uint8_t *d_chained = m;
    hipStream_t st = ctx->stream;
    if (span) K4_HIP(ctx, hipMemcpyAsync(ctx->d_src, src + lo, span, hipMemcpyHostToDevice, st));
    if (nBlocks) {
        K4_HIP(ctx, hipMemcpyAsync(d_boff, h_boff.data(), (size_t)nBlocks * 8, hipMemcpyHostToDevice, st));
        K4_HIP(ctx, hipMemcpyAsync(d_blen, blkLen, (size_t)nBlocks * 4, hipMemcpyHostToDevice, st));
    }
    K4_HIP(ctx, hipMemcpyAsync(d_first, firstBlk, (size_t)nStreams * 8, hipMemcpyHostToDevice, st));
    K4_HIP(ctx, hipMemcpyAsync(d_doff, h_doff.data(), (size_t)nStreams * 8, hipMemcpyHostToDevice, st));
    K4_HIP(ctx, hipMemcpyAsync(d_dcap, dstCap, (size_t)nStreams * 8, hipMemcpyHostToDevice, st));
    K4_HIP(ctx, hipMemcpyAsync(d_nblk, nBlk, (size_t)nStreams * 4, hipMemcpyHostToDevice, st));
    K4_HIP(ctx, hipMemcpyAsync(d_bsize, blockSize, (size_t)nStreams * 4, hipMemcpyHostToDevice, st));
    K4_HIP(ctx, hipMemcpyAsync(d_chained, chained, (size_t)nStreams, hipMemcpyHostToDevice, st));
    K4_HIP(ctx, hipMemsetAsync(ctx->d_dst, 0, (size_t)dtotal, st));       /* (offset-0 matches of a hostile stream leave zeros, not an earlier call's bytes: see run_host_inner) */
    rc = k4lz4_decode_chain_batch_device(ctx, ctx->d_src, d_boff, d_blen, d_first, d_nblk, d_bsize, d_chained, ctx->d_dst, d_doff,
                                         d_dcap, d_out, nStreams, st);
    if (rc != K4LZ4_OK) return rc;
    K4_HIP(ctx, hipMemcpyAsync(outLen, d_out, (size_t)nStreams * 8, hipMemcpyDeviceToHost, st));
    if (dtotal) K4_HIP(ctx, hipMemcpyAsync(ctx->h_stage, ctx->d_dst, (size_t)dtotal, hipMemcpyDeviceToHost, st));
    K4_HIP(ctx, hipStreamSynchronize(st));
    if ((rc = take_device_status(ctx)) != K4LZ4_OK) return rc;
    for (int64_t i = 0; i < nStreams; i++)
        if (outLen[i] > 0 && (uint64_t)outLen[i] <= dstCap[i]) memcpy(dst + dstOff[i], ctx->h_stage + h_doff[(size_t)i], (size_t)outLen[i]);
    return K4LZ4_OK;
}

int k4lz4_frame_assemble_device(k4lz4_ctx *ctx, const uint8_t *arena, const uint64_t *slotOff, const int32_t *outLen,
                                const uint32_t *blkSum, const uint64_t *recOff, int64_t nBlocks, const uint8_t *hdr,
                                const uint32_t *hdrLen, const uint32_t *hdrSum, const uint64_t *frameOff,
                                const uint64_t *tailOff, const uint32_t *contentSum, uint8_t *frames, uint64_t *frameLen,
                                int64_t nFrames, void *stream)
{
    if (!ctx) return fail(nullptr, K4LZ4_E_ARG, "ctx is NULL");
    if (nBlocks < 0 || nFrames < 0 || (nBlocks > 0 && (!arena || !slotOff || !outLen || !recOff)) ||
        (nFrames > 0 && (!hdr || !hdrLen || !hdrSum || !frameOff || !tailOff || !frames || !frameLen)))
        return fail(ctx, K4LZ4_E_ARG, "bad argument");
    K4_HIP(ctx, hipSetDevice(ctx->device));
    if (nBlocks > 0) {
        k4::FrameBlocksArgs a{arena, slotOff, outLen, blkSum, recOff, frames, nBlocks};
        hipLaunchKernelGGL(k4::k4_frame_blocks_kernel, dim3((unsigned)((nBlocks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    }
    if (nFrames > 0) {
        k4::FrameEdgesArgs e{hdr, hdrLen, hdrSum, frameOff, tailOff, contentSum, frames, frameLen, nFrames};
        hipLaunchKernelGGL(k4::k4_frame_edges_kernel, dim3((unsigned)((nFrames + 255) / 256)), dim3(256), 0, (hipStream_t)stream, e);
    }
    K4_HIP(ctx, hipGetLastError());
    return K4LZ4_OK;
}

int k4lz4_unpickle_sizes_device(k4lz4_ctx *ctx, const uint8_t *src, const uint64_t *srcOff, const int32_t *srcLen,
                                int32_t *outLen, int64_t n, void *stream)
{
    if (!ctx) return fail(nullptr, K4LZ4_E_ARG, "ctx is NULL");
    if (n < 0 || (n > 0 && (!src || !srcOff || !srcLen || !outLen))) return fail(ctx, K4LZ4_E_ARG, "bad argument");
    if (n == 0) return K4LZ4_OK;
    K4_HIP(ctx, hipSetDevice(ctx->device));
    k4::BatchArgs a{};
    a.src = src; a.srcOff = srcOff; a.srcLen = srcLen; a.outLen = outLen; a.n = n;
    hipLaunchKernelGGL(k4::k4_unpickle_sizes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    K4_HIP(ctx, hipGetLastError());
    return K4LZ4_OK;
}

}  // extern "C"
