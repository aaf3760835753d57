// TEST INFRASTRUCTURE ONLY -- a tiny SIMT emulator for the HIP kernels in aria_amd/csrc.
//
// The build container has no GPU and GPU time is scarce, so the kernels are written against
// aria_amd/csrc/aria_device.h, which maps a small device vocabulary either to the gfx950
// builtins (product build, hipcc) or to this emulator (-DARIA_EMU, host clang++).  The emulator
// runs every thread of a workgroup as a ucontext fiber on ONE OS thread, strictly sequentially:
// a fiber runs until it reaches __syncthreads() or a wave collective (shuffle / ballot / MFMA)
// and then yields.  That schedule is adversarial for missing barriers (thread 0 runs a whole
// phase before thread 1 starts), so LDS races show up as wrong results, deterministically.
//
// Wave collectives follow the gfx950 lane layouts given in /opt/skills/guides (wave = 64 lanes;
// mfma_f32_32x32x16_bf16: A lane l holds A[l&31][8*(l>>5)..+7], B lane l holds B[8*(l>>5)..+7][l&31],
// C reg r of lane l is C[(r&3)+8*(r>>2)+4*(l>>5)][l&31]).  Those layouts are re-verified on real
// hardware by tests/test_gpu_probes.py.
//
// This file is never part of libaria_hip.so and nothing under aria_amd/ python imports the
// emulated library.
#pragma once
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <utility>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {

enum State { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    int state = READY;
    dim3 tid;
    int linear = 0;
};

struct WaveBuf {
    alignas(16) unsigned char slot[64][128];
};

extern Fiber* cur;
extern dim3 g_blockIdx, g_blockDim, g_gridDim;
extern char* g_dyn_smem;
extern WaveBuf* g_wave_bufs;

void yield(int state);
void run_grid(dim3 grid, dim3 block, size_t shmem, void (*entry)(void*), void* arg);

inline void syncthreads() { yield(WAIT_BLOCK); }
inline void wave_sync() { yield(WAIT_WAVE); }
inline int lane() { return cur->linear & 63; }
inline WaveBuf& wbuf() { return g_wave_bufs[cur->linear >> 6]; }

template <class T>
inline T shfl(T v, int src) {
    static_assert(sizeof(T) <= 128, "");
    WaveBuf& w = wbuf();
    std::memcpy(w.slot[lane()], &v, sizeof(T));
    wave_sync();
    T r;
    std::memcpy(&r, w.slot[src & 63], sizeof(T));
    wave_sync();
    return r;
}

bool lane_alive(int linear);

// LDS-DMA model.  Default: the 16 bytes land at once (adversarial for WAR: a slot restaged too early is clobbered before its
// last reader runs).  ARIA_EMU_GLDS_DEFER=1: they land only when a vmcnt wait (or the end of the thread) forces them to
// (adversarial for RAW: a read that is not covered by a counted wait + barrier sees the 0xFF poison).
void glds(const void* src, void* dst, int bytes = 16);
void wait_vm(int keep);

inline unsigned long long ballot(bool p) {
    WaveBuf& w = wbuf();
    w.slot[lane()][0] = p ? 1 : 0;
    wave_sync();
    unsigned long long m = 0;
    const int base = cur->linear & ~63;
    for (int i = 0; i < 64; ++i)
        if (lane_alive(base + i) && w.slot[i][0]) m |= 1ull << i;
    wave_sync();
    return m;
}

template <class F, class... Args>
struct LaunchPack {
    F f;
    std::tuple<Args...> args;
};

template <class F, class... Args>
void launch_entry(void* p) {
    auto* lp = static_cast<LaunchPack<F, Args...>*>(p);
    std::apply(lp->f, lp->args);
}

template <class... KArgs, class... Args>
void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, Args... args) {
    using F = void (*)(KArgs...);
    LaunchPack<F, KArgs...> lp{kernel, std::tuple<KArgs...>(static_cast<KArgs>(args)...)};
    run_grid(grid, block, shmem, &launch_entry<F, KArgs...>, &lp);
}

}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)
