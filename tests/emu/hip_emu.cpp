// TEST INFRASTRUCTURE ONLY -- scheduler of the fiber-based SIMT emulator (see hip_emu.h).
#include "hip_emu.h"

#include <sys/mman.h>

namespace emu {

Fiber* cur = nullptr;
dim3 g_blockIdx, g_blockDim, g_gridDim;
char* g_dyn_smem = nullptr;
WaveBuf* g_wave_bufs = nullptr;

static ucontext_t sched_ctx;
static void (*g_entry)(void*) = nullptr;
static void* g_arg = nullptr;
static constexpr size_t kStack = 256 * 1024;
static constexpr size_t kMaxSmem = 160 * 1024;

void yield(int state) {
    cur->state = state;
    swapcontext(&cur->ctx, &sched_ctx);
}

struct PendingDma {
    const void* src;
    void* dst;
    int bytes;
};
static std::vector<std::vector<PendingDma>> g_dma;  // per thread, oldest first
static int g_glds_defer = 0;  // re-read from the environment at every launch (run_grid)
static int glds_defer() { return g_glds_defer; }
void glds(const void* src, void* dst, int bytes) {
    if (!glds_defer()) {
        std::memcpy(dst, src, size_t(bytes));
        return;
    }
    if (g_dma.size() <= size_t(cur->linear)) g_dma.resize(cur->linear + 1);
    g_dma[cur->linear].push_back({src, dst, bytes});
}
void wait_vm(int keep) {
    if (g_dma.size() <= size_t(cur->linear)) return;
    auto& q = g_dma[cur->linear];
    while (q.size() > size_t(keep)) {
        std::memcpy(q.front().dst, q.front().src, size_t(q.front().bytes));
        q.erase(q.begin());
    }
}

static void trampoline() {
    g_entry(g_arg);
    wait_vm(0);
    cur->state = DONE;
    swapcontext(&cur->ctx, &sched_ctx);
}

static std::vector<Fiber> fibers;
static std::vector<char*> stacks;

bool lane_alive(int linear) { return size_t(linear) < fibers.size() && fibers[linear].state != DONE; }

static void ensure_fibers(size_t n) {
    while (stacks.size() < n) {
        void* p = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) {
            perror("emu: mmap stack");
            abort();
        }
        stacks.push_back(static_cast<char*>(p));
    }
    fibers.resize(n);
}

static void run_block(dim3 block) {
    const size_t n = size_t(block.x) * block.y * block.z;
    ensure_fibers(n);
    const size_t nwaves = (n + 63) / 64;
    static std::vector<WaveBuf> wb;
    if (wb.size() < nwaves) wb.resize(nwaves);
    std::memset(wb.data(), 0, nwaves * sizeof(WaveBuf));
    g_wave_bufs = wb.data();
    std::memset(g_dyn_smem, 0xFF, kMaxSmem);  // LDS is uninitialised on hardware: poison with bf16 NaNs
    for (size_t i = 0; i < n; ++i) {
        Fiber& f = fibers[i];
        f.stack = stacks[i];
        f.state = READY;
        f.linear = int(i);
        f.tid = dim3(unsigned(i % block.x), unsigned((i / block.x) % block.y), unsigned(i / (size_t(block.x) * block.y)));
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &sched_ctx;
        makecontext(&f.ctx, trampoline, 0);
    }
    for (;;) {
        bool ran = false;
        for (size_t i = 0; i < n; ++i) {
            if (fibers[i].state == READY) {
                cur = &fibers[i];
                swapcontext(&sched_ctx, &cur->ctx);
                ran = true;
            }
        }
        size_t done = 0, at_block = 0;
        for (size_t i = 0; i < n; ++i) {
            done += fibers[i].state == DONE;
            at_block += fibers[i].state == WAIT_BLOCK;
        }
        if (done == n) break;
        bool released = false;
        if (at_block && at_block + done == n) {
            for (size_t i = 0; i < n; ++i)
                if (fibers[i].state == WAIT_BLOCK) fibers[i].state = READY;
            released = true;
        }
        for (size_t w = 0; w < nwaves; ++w) {
            size_t lo = w * 64, hi = lo + 64 < n ? lo + 64 : n, ww = 0, dd = 0;
            for (size_t i = lo; i < hi; ++i) {
                ww += fibers[i].state == WAIT_WAVE;
                dd += fibers[i].state == DONE;
            }
            if (ww && ww + dd == hi - lo) {
                for (size_t i = lo; i < hi; ++i)
                    if (fibers[i].state == WAIT_WAVE) fibers[i].state = READY;
                released = true;
            }
        }
        if (!ran && !released) {
            fprintf(stderr, "emu: DEADLOCK in block (%u,%u,%u): divergent barrier / wave collective\n", g_blockIdx.x,
                    g_blockIdx.y, g_blockIdx.z);
            for (size_t i = 0; i < n; ++i)
                if (fibers[i].state != DONE) {
                    fprintf(stderr, "  first stuck thread %zu state %d\n", i, fibers[i].state);
                    break;
                }
            abort();
        }
    }
}

void run_grid(dim3 grid, dim3 block, size_t shmem, void (*entry)(void*), void* arg) {
    if (shmem > kMaxSmem) {
        fprintf(stderr, "emu: dynamic LDS %zu > 160 KiB\n", shmem);
        abort();
    }
    if (!g_dyn_smem) g_dyn_smem = static_cast<char*>(aligned_alloc(256, kMaxSmem));
    const char* defer = std::getenv("ARIA_EMU_GLDS_DEFER");
    g_glds_defer = defer && defer[0] == '1';
    g_entry = entry;
    g_arg = arg;
    g_gridDim = grid;
    g_blockDim = block;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                g_blockIdx = dim3(x, y, z);
                run_block(block);
            }
}

}  // namespace emu
