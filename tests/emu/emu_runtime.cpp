/* tests/emu/emu_runtime.cpp -- fiber scheduler behind the fake <hip/hip_runtime.h>.
 * Test infrastructure only (see the header).  x86-64 System V only. */
#include "hip/hip_runtime.h"
#include <stdlib.h>
#include <stdio.h>
#include <sys/mman.h>
#include <thread>
#include <vector>
#include <atomic>

extern "C" void k4emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl k4emu_switch
.type k4emu_switch,@function
k4emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

namespace k4emu {

enum { STACK_BYTES = 256 * 1024 };

struct Lane {
    void *sp;
    char *stack;
    LaneIds ids;
    int index;      /* linear thread index in block */
    bool done;
};

struct WaveSync {
    uint64_t slot[2][WAVE];
    int arrived;
    unsigned gen;
};

struct Block {
    std::vector<Lane> lanes;
    std::vector<WaveSync> waves;
    int nthreads;
    int block_arrived;
    unsigned block_gen;
    int live;
    void *sched_sp;
    Lane *cur;
    void (*fn)(void *);
    void *arg;
};

static thread_local Block *tl_block = nullptr;

LaneIds &ids() { return tl_block->cur->ids; }
int lane_id() { return tl_block->cur->index & (WAVE - 1); }

static void yield_to_scheduler() {
    Block *b = tl_block;
    Lane *me = b->cur;
    k4emu_switch(&me->sp, b->sched_sp);
}

const uint64_t *wave_exchange(uint64_t in) {
    Block *b = tl_block;
    Lane *me = b->cur;
    int w = me->index / WAVE, l = me->index % WAVE;
    WaveSync &ws = b->waves[w];
    unsigned g = ws.gen;
    int wave_lanes = b->nthreads - w * WAVE;
    if (wave_lanes > WAVE) wave_lanes = WAVE;
    ws.slot[g & 1][l] = in;
    if (++ws.arrived == wave_lanes) {
        ws.arrived = 0;
        for (int i = wave_lanes; i < WAVE; i++) ws.slot[g & 1][i] = 0;
        ws.gen = g + 1;
    } else {
        while (ws.gen == g) yield_to_scheduler();
    }
    return ws.slot[g & 1];
}

void block_barrier() {
    Block *b = tl_block;
    unsigned g = b->block_gen;
    if (++b->block_arrived == b->nthreads) {
        b->block_arrived = 0;
        b->block_gen = g + 1;
    } else {
        while (b->block_gen == g) yield_to_scheduler();
    }
}

static void lane_entry() {
    Block *b = tl_block;
    Lane *me = b->cur;
    b->fn(b->arg);
    me->done = true;
    b->live--;
    /* a lane that returns while wave-mates still wait would deadlock the rendezvous:
     * kernels must exit wave-uniformly (which they also must on hardware for ballots). */
    k4emu_switch(&me->sp, b->sched_sp);
    abort();
}

static void run_block(Block &b, dim3 bid, dim3 grid, dim3 bdim) {
    tl_block = &b;
    b.block_arrived = 0;
    b.block_gen = 0;
    b.live = b.nthreads;
    for (auto &w : b.waves) { w.arrived = 0; w.gen = 0; }
    for (int i = 0; i < b.nthreads; i++) {
        Lane &L = b.lanes[i];
        L.index = i;
        L.done = false;
        L.ids.tid = dim3(i, 0, 0);
        L.ids.bid = bid;
        L.ids.bdim = bdim;
        L.ids.gdim = grid;
        uintptr_t top = ((uintptr_t)(L.stack + STACK_BYTES)) & ~(uintptr_t)15;
        void **sp = (void **)(top - 64);     /* sp % 16 == 0 */
        for (int k = 0; k < 6; k++) sp[k] = nullptr;
        sp[6] = (void *)&lane_entry;          /* return address consumed by `ret` */
        sp[7] = nullptr;
        L.sp = sp;
    }
    long spins = 0;
    while (b.live > 0) {
        int before = b.live;
        for (int i = 0; i < b.nthreads; i++) {
            Lane &L = b.lanes[i];
            if (L.done) continue;
            b.cur = &L;
            k4emu_switch(&b.sched_sp, L.sp);
        }
        if (b.live == before && ++spins > 200000000L) {
            fprintf(stderr, "k4emu: no lane finished after many rounds (divergent collective / deadlock?)\n");
            abort();
        }
    }
    tl_block = nullptr;
}

void launch(dim3 grid, dim3 block, void (*fn)(void *), void *arg, int threads) {
    int nthreads = (int)(block.x * block.y * block.z);
    long nblocks = (long)grid.x * grid.y * grid.z;
    if (threads <= 0) {
        const char *e = getenv("K4EMU_THREADS");
        threads = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        if (threads < 1) threads = 1;
    }
    if (threads > nblocks) threads = (int)nblocks;
    std::atomic<long> next(0);
    auto worker = [&]() {
        Block b;
        b.nthreads = nthreads;
        b.fn = fn;
        b.arg = arg;
        b.lanes.resize(nthreads);
        b.waves.resize((nthreads + WAVE - 1) / WAVE);
        char *stacks = (char *)mmap(nullptr, (size_t)STACK_BYTES * nthreads, PROT_READ | PROT_WRITE,
                                    MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stacks == MAP_FAILED) { perror("mmap"); abort(); }
        for (int i = 0; i < nthreads; i++) b.lanes[i].stack = stacks + (size_t)STACK_BYTES * i;
        for (;;) {
            long i = next.fetch_add(1);
            if (i >= nblocks) break;
            dim3 bid((unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((long)grid.x * grid.y)));
            run_block(b, bid, grid, block);
        }
        munmap(stacks, (size_t)STACK_BYTES * nthreads);
    };
    if (threads <= 1) { worker(); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(worker);
    for (auto &t : pool) t.join();
}

}  // namespace k4emu
