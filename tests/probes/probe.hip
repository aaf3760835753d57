// Hardware-semantics probes -- TEST INFRASTRUCTURE, built into tests/probes/libaria_probe.so (never into the product library: round 3 had
// them in libaria_hip.so's ABI).  tests/test_gpu_probes.py re-verifies on a real MI355X the lane layouts tests/emu/hip_emu.h assumes;
// tools/probes/l2_atomics.py measures fp32 atomic rates.
#include <hip/hip_runtime.h>
#include <cstdint>
#define ARIA_OK 0
#define ARIA_ERR_INVALID 1
#define ARIA_ERR_UNSUPPORTED 3
#define ARIA_ERR_LAUNCH 4
static int aria_check_launch() { return hipGetLastError() == hipSuccess ? ARIA_OK : ARIA_ERR_LAUNCH; }

#ifndef ARIA_EMU
namespace {
typedef short s16x4v __attribute__((ext_vector_type(4)));
// LDS holds lds[i] = i (u16).  mode 0: lane address = lane*8 bytes; mode 1: lane address = (lane&15)*128 + (lane>>4)*8 bytes.
__global__ void probe_tr16_kernel(unsigned short* out, int mode) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    const int off = mode == 0 ? l * 4 : ((l & 15) * 64 + (l >> 4) * 4);
    s16x4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(lds + off));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

// ---- L2-scope fp32 atomics (the single-pass attention backward accumulates dQ with them) -----------------------------------------------
// Every workgroup (256 threads) adds 1.0f `iters` times to every float of ITS region; a wave instruction covers 2 rows of 32 consecutive
// floats (128 bytes each, `row_stride` floats apart) -- the access shape of an MFMA accumulator tile added to a row-major fp32 matrix.
//   region of workgroup b:  region_mode 0: b & 7 (= the XCD under round-robin placement: each region is touched by ONE XCD only)
//                           region_mode 1: b (private region)      region_mode 2: 0 (everybody, all XCDs)
//   scope 0: no sc bits (performed in the issuing XCD's L2)   scope 1: sc1 (agent scope)
// xcc[b] = HW_REG_XCC_ID of the workgroup (verifies the placement assumption).
__global__ __launch_bounds__(256) void probe_atomic_kernel(float* buf, int* xcc, long long region_floats, int region_mode, int scope,
                                                           int iters, int row_stride) {
    const int b = blockIdx.x, t = threadIdx.x, l = t & 63, w = t >> 6;
    if (t == 0) xcc[b] = int(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)));  // HW_REG_XCC_ID, bits [3:0]
    const long long region = region_mode == 0 ? (b & 7) : region_mode == 1 ? b : 0;
    float* base = buf + region * region_floats;
    const long long rows = region_floats / row_stride;  // rows of `row_stride` floats; the first 128 floats of a row are used
    for (int it = 0; it < iters; ++it)
        for (long long r0 = 2 * w; r0 < rows; r0 += 8)
            for (int c0 = 0; c0 < 128; c0 += 32) {
                float* p = base + (r0 + (l >> 5)) * row_stride + c0 + (l & 31);
                const float one = 1.0f;
                if (scope == 0)
                    asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(one) : "memory");
                else
                    asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(one) : "memory");
            }
}

// ---- what the cross-workgroup synchronisation of a one-launch decode schedule costs (tools/probes/stream_sync_costs.py) ----------------
// Every workgroup (256 threads, `nblocks` of them) does only the selected pieces; mode bits:
//   1  ticket: thread 0, RETURNING atomic add on ONE word, broadcast through LDS + barrier     64: the word is cnt[16 * (b & 7)] (one per XCD)
//   2  completion: thread 0, atomic add on cnt[1024 + 16 * ((b / 512) & 63)] at the end (a new word every 512 workgroups)   128: non-returning
//   4  every wave: buffer_wbl2 sc1 + s_waitcnt vmcnt(0)        8  every wave: buffer_inv sc1
//   16 wave 0: one sc1 poll load of cnt[2048] + wait           32 two bare s_barriers
//   256 every wave: one 2-byte sc1 store to a private slot of `sink` + s_waitcnt vmcnt(0)
__global__ __launch_bounds__(256) void probe_sync_kernel(int* cnt, unsigned short* sink, int mode) {
    __shared__ int s_t;
    const int b = blockIdx.x, t = threadIdx.x, w = t >> 6;
    int ticket = b;
    if (mode & 1) {
        if (t == 0) s_t = __hip_atomic_fetch_add(cnt + ((mode & 64) ? 16 * (b & 7) : 0), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        ticket = s_t;
    }
    if ((mode & 16) && w == 0) {
        const int v = __hip_atomic_load(cnt + 2048, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == 0x7fffffff) sink[0] = 1;
    }
    if (mode & 32) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
    }
    if (mode & 8) asm volatile("buffer_inv sc1" ::: "memory");
    if (mode & 256) {
        if ((t & 63) == 0) __hip_atomic_store(sink + 64 + (((long long)ticket * 4 + w) & 0xfffff), (unsigned short)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (mode & 4) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
    if (mode & 2) {
        __syncthreads();
        int* c = cnt + 1024 + 16 * ((b / 512) & 63);
        if (t == 0) {
            if (mode & 128)
                __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                s_t = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Two workgroups (0 and `partner`; everybody else leaves) bounce a word `iters` times: A writes 2i + 1 and waits for 2i + 2, B the reverse.
// how 0: sc1 store / sc1 load     1: atomic exchange-add / sc1 load     2: plain store + buffer_wbl2 sc1 / buffer_inv sc1 + plain load
// Polls are bounded (a lost update ends the run with out[1] = 1 instead of hanging the GPU).  out[0] = round trips completed.
__global__ __launch_bounds__(64) void probe_pingpong_kernel(int* flag, int* out, int iters, int partner, int how) {
    const int b = blockIdx.x;
    if ((b != 0 && b != partner) || threadIdx.x != 0) return;
    const bool A = b == 0;
    int done = 0;
    for (int i = 0; i < iters; ++i) {
        const int mine = A ? 2 * i + 1 : 2 * i + 2, theirs = A ? 2 * i + 2 : 2 * i + 1;
        for (int phase = 0; phase < 2; ++phase) {
            if ((phase == 0) == A) {  // A writes first, B waits first
                if (how == 0)
                    __hip_atomic_store(flag, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (how == 1)
                    __hip_atomic_exchange(flag, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else {
                    *(volatile int*)flag = mine;
                    asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
                }
            } else {
                int spins = 0, v;
                do {
                    if (how == 2) {
                        asm volatile("buffer_inv sc1" ::: "memory");
                        v = *(volatile int*)flag;
                    } else
                        v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while (v != theirs && ++spins < (1 << 20));
                if (v != theirs) {
                    out[1] = 1;
                    out[0] = done;
                    return;
                }
            }
        }
        ++done;
    }
    if (A) out[0] = done;
}

// ---- how fast a GEMV-shaped read streams as a function of the bytes a CU keeps in flight (tools/probes/gemv_stream_rate.py) ----------------
// A wave owns R + RL consecutive rows of 5120 bytes (Aria's D = 2560 bf16): R rows land in registers (16-byte non-temporal loads, lane l takes
// chunks l + 64 i -- the decode GEMVs' access shape), RL rows land in LDS by LDS-DMA (no registers).  Everything is requested before the first
// use; the "use" is an xor fold written only if it has an impossible value.  Workgroups per CU are set from outside by the dynamic LDS size.
typedef uint32_t pu32x4 __attribute__((ext_vector_type(4)));
template <int R, int RL>
__global__ __launch_bounds__(256) void probe_stream_kernel(const uint16_t* W, long long nrows, uint32_t* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NC = 5;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long row0 = ((long long)blockIdx.x * 4 + w) * (R + RL);
    if (row0 + R + RL > nrows) return;
    pu32x4 a[R > 0 ? R : 1][NC];
    char* lds = smem + w * (RL > 0 ? RL : 1) * 5120;
#pragma unroll
    for (int r = 0; r < RL; ++r)
#pragma unroll
        for (int i = 0; i < NC; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + (row0 + R + r) * 2560 + (l + 64 * i) * 8),
                                             (__attribute__((address_space(3))) void*)(lds + r * 5120 + i * 1024), 16, 0, 2 /* nt */);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < NC; ++i)
            a[r][i] = __builtin_nontemporal_load(reinterpret_cast<const pu32x4*>(W + (row0 + r) * 2560 + (l + 64 * i) * 8));
    __builtin_amdgcn_sched_barrier(0);  // every request is out before the first use (the compiler would otherwise fold as it goes, a few loads at a time)
    uint32_t x = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < NC; ++i) x ^= a[r][i][0] ^ a[r][i][1] ^ a[r][i][2] ^ a[r][i][3];
    if (RL > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < RL; ++r)
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const pu32x4 v = *reinterpret_cast<const pu32x4*>(lds + r * 5120 + i * 1024 + l * 16);
                x ^= v[0] ^ v[1] ^ v[2] ^ v[3];
            }
    }
    if (x == 0x9e3779b9u) out[blockIdx.x & 1023] = x;
}

// ---- which accesses make one workgroup's data visible to a workgroup on another XCD INSIDE a launch (tools/probes/xcd_visibility.py) ------
// Workgroups 0 (writer) and `partner` (reader) take turns `iters` times over the SAME 64-word buffer (so the reader's XCD has the previous
// round's lines in its L2): the writer fills buf[l] = round (one word per lane), makes it visible, bumps `flag`; the reader waits for the
// flag, reads the buffer, counts the words that are not `round`, answers through `flag2`.
//   wmode 0: plain stores + s_waitcnt            1: sc1 stores + s_waitcnt           2: plain stores + buffer_wbl2 sc1 + s_waitcnt
//   rmode 0: plain loads      1: sc1 loads       2: buffer_inv sc1, then plain loads      3: sc0 sc1 loads (system scope)
// out[0] = rounds completed, out[1] = stale words seen, out[2] = 1 if a wait timed out.
__global__ __launch_bounds__(64) void probe_visibility_kernel(int* buf, int* flag, int* flag2, int* out, int iters, int partner, int wmode, int rmode) {
    const int b = blockIdx.x, l = threadIdx.x;
    if (b != 0 && b != partner) return;
    int stale = 0;
    for (int i = 1; i <= iters; ++i) {
        if (b == 0) {
            if (wmode == 1)
                __hip_atomic_store(buf + l, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                *(volatile int*)(buf + l) = i;
            if (wmode == 2)
                asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (l == 0) __hip_atomic_store(flag, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(flag2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != i && ++spins < (1 << 20)) {}
            if (spins >= (1 << 20)) {
                if (l == 0) out[2] = 1;
                return;
            }
        } else {
            int spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != i && ++spins < (1 << 20)) {}
            if (spins >= (1 << 20)) {
                if (l == 0) out[2] = 1;
                return;
            }
            int v;
            if (rmode == 1)
                v = __hip_atomic_load(buf + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (rmode == 3)
                v = __hip_atomic_load(buf + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else {
                if (rmode == 2) asm volatile("buffer_inv sc1" ::: "memory");
                v = *(volatile int*)(buf + l);
            }
            stale += v != i;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (l == 0) __hip_atomic_store(flag2, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (b == partner) {
        atomicAdd(out + 1, stale);
        if (l == 0) out[0] = iters;
    }
}
}  // namespace
#endif

extern "C" int aria_probe_visibility(int* buf, int* flag, int* flag2, int* out, int iters, int partner, int wmode, int rmode, void* stream) {
#ifdef ARIA_EMU
    (void)buf; (void)flag; (void)flag2; (void)out; (void)iters; (void)partner; (void)wmode; (void)rmode; (void)stream;
    return ARIA_ERR_UNSUPPORTED;
#else
    if (!buf || !flag || !flag2 || !out || iters <= 0 || partner <= 0) return ARIA_ERR_INVALID;
    hipLaunchKernelGGL(probe_visibility_kernel, dim3(unsigned(partner + 1)), dim3(64), 0, static_cast<hipStream_t>(stream), buf, flag, flag2, out,
                       iters, partner, wmode, rmode);
    return aria_check_launch();
#endif
}

// rows in registers R (0, 2, 4, 8) + rows through LDS RL (0, 2, 4); lds_bytes sets the workgroups a CU can hold (and must cover 4 * RL * 5120)
extern "C" int aria_probe_stream(const void* W, int64_t nrows, int R, int RL, int64_t lds_bytes, void* out, void* stream) {
#ifdef ARIA_EMU
    (void)W; (void)nrows; (void)R; (void)RL; (void)lds_bytes; (void)out; (void)stream;
    return ARIA_ERR_UNSUPPORTED;
#else
    if (!W || !out || nrows <= 0 || lds_bytes < int64_t(4) * RL * 5120 || lds_bytes > 160 * 1024) return ARIA_ERR_INVALID;
    const int64_t per_block = int64_t(4) * (R + RL);
    if (per_block <= 0) return ARIA_ERR_INVALID;
    const dim3 grid(unsigned(nrows / per_block)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint16_t* w = static_cast<const uint16_t*>(W);
    uint32_t* o = static_cast<uint32_t*>(out);
#define PS(r, rl)                                                                                               \
    if (R == r && RL == rl) {                                                                                   \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_stream_kernel<r, rl>),                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));                  \
        hipLaunchKernelGGL((probe_stream_kernel<r, rl>), grid, block, size_t(lds_bytes), st, w, (long long)nrows, o); \
        return aria_check_launch();                                                                             \
    }
    PS(2, 0) PS(4, 0) PS(8, 0) PS(4, 2) PS(4, 4) PS(2, 2) PS(0, 4) PS(8, 4) PS(6, 0)
#undef PS
    return ARIA_ERR_UNSUPPORTED;
#endif
}

extern "C" int aria_probe_sync(int* cnt, unsigned short* sink, int64_t nblocks, int mode, void* stream) {
#ifdef ARIA_EMU
    (void)cnt; (void)sink; (void)nblocks; (void)mode; (void)stream;
    return ARIA_ERR_UNSUPPORTED;
#else
    if (!cnt || !sink || nblocks <= 0) return ARIA_ERR_INVALID;
    hipLaunchKernelGGL(probe_sync_kernel, dim3(unsigned(nblocks)), dim3(256), 0, static_cast<hipStream_t>(stream), cnt, sink, mode);
    return aria_check_launch();
#endif
}

extern "C" int aria_probe_pingpong(int* flag, int* out, int iters, int partner, int how, void* stream) {
#ifdef ARIA_EMU
    (void)flag; (void)out; (void)iters; (void)partner; (void)how; (void)stream;
    return ARIA_ERR_UNSUPPORTED;
#else
    if (!flag || !out || iters <= 0 || partner <= 0) return ARIA_ERR_INVALID;
    hipLaunchKernelGGL(probe_pingpong_kernel, dim3(unsigned(partner + 1)), dim3(64), 0, static_cast<hipStream_t>(stream), flag, out, iters, partner, how);
    return aria_check_launch();
#endif
}

extern "C" int aria_probe_atomic(float* buf, int* xcc, int64_t nblocks, int64_t region_floats, int region_mode, int scope, int iters,
                                 int row_stride, void* stream) {
#ifdef ARIA_EMU
    (void)buf; (void)xcc; (void)nblocks; (void)region_floats; (void)region_mode; (void)scope; (void)iters; (void)row_stride; (void)stream;
    return ARIA_ERR_UNSUPPORTED;
#else
    if (!buf || !xcc || nblocks <= 0 || row_stride < 128 || region_floats % row_stride) return ARIA_ERR_INVALID;
    hipLaunchKernelGGL(probe_atomic_kernel, dim3(unsigned(nblocks)), dim3(256), 0, static_cast<hipStream_t>(stream), buf, xcc,
                       (long long)region_floats, region_mode, scope, iters, row_stride);
    return aria_check_launch();
#endif
}

extern "C" int aria_probe_tr16(void* out, int mode, void* stream) {
#ifdef ARIA_EMU
    (void)out; (void)mode; (void)stream;
    return ARIA_ERR_UNSUPPORTED;
#else
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<unsigned short*>(out), mode);
    return aria_check_launch();
#endif
}
