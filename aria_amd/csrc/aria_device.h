// Device vocabulary shared by every kernel in aria_amd/csrc (gfx950 / CDNA4 only).
//
// Product build (hipcc --offload-arch=gfx950): thin inline wrappers over the gfx950 builtins.
// Test build (-DARIA_EMU, host clang++): the same names implemented on tests/emu/hip_emu.h so the
// kernels' index arithmetic and barrier placement can be exercised on a machine without a GPU.
// The emulated library is test infrastructure; the Python package never loads it.
#pragma once
#include <cstdint>

#ifdef ARIA_EMU
#include "hip_emu.h"
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define ARIA_SMEM_STATIC static
#define ARIA_DYN_SMEM(name) char* name = emu::g_dyn_smem
#define ARIA_LAUNCH(kernel, grid, block, shmem, stream, ...) emu::launch(kernel, grid, block, shmem, __VA_ARGS__)
typedef void* hipStream_t;
#else
#include <hip/hip_runtime.h>
#define ARIA_SMEM_STATIC __shared__
#define ARIA_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define ARIA_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, shmem, static_cast<hipStream_t>(stream), __VA_ARGS__)
#endif

// internal helper shared by the translation units (common.hip): ARIA_OK, or ARIA_ERR_LAUNCH when the launch just enqueued failed.  NOT part
// of the C ABI: hidden, so libaria_hip.so exports the extern "C" entry points of include/aria_hip.h and nothing else of ours.
__attribute__((visibility("hidden"))) int aria_check_launch();

#ifdef ARIA_EMU
#include <algorithm>
#include <cmath>
inline float __expf(float x) { return std::exp(x); }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
using std::max;
using std::min;
#endif

namespace ad {

typedef uint16_t bf16_t;  // storage type: raw bfloat16 bits
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- bf16 <-> f32 (round to nearest even; identical to torch's conversion) ----
__device__ __forceinline__ float bf2f(bf16_t h) { return __builtin_bit_cast(float, uint32_t(h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
#ifdef ARIA_EMU
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return bf16_t((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return bf16_t(u >> 16);
#else
    return __builtin_bit_cast(bf16_t, static_cast<__bf16>(f));  // v_cvt_pk_bf16_f32 (RNE)
#endif
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
#ifdef ARIA_EMU
    return uint32_t(f2bf(lo)) | (uint32_t(f2bf(hi)) << 16);
#else
    // ONE v_cvt_pk_bf16_f32 (two separate conversions + shift + or cost 4 VALU slots; the softmax / epilogue code packs thousands)
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
#endif
}
__device__ __forceinline__ float bflo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
// c + a.lo * b.lo + a.hi * b.hi on packed bf16 pairs, fp32 accumulate (v_dot2c_f32_bf16: no unpacking of either operand)
__device__ __forceinline__ float dot2bf(uint32_t a, uint32_t b, float c) {
#ifdef ARIA_EMU
    return c + bflo(a) * bflo(b) + bfhi(a) * bfhi(b);
#else
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
#endif
}
// round an fp32 value through bf16 (mirrors a bf16 tensor op whose result is materialised)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// ---- wave (64 lanes) collectives ----
#ifdef ARIA_EMU
__device__ __forceinline__ void sync() { emu::syncthreads(); }
template <class T>
__device__ __forceinline__ T shfl(T v, int src) { return emu::shfl(v, src); }
template <class T>
__device__ __forceinline__ T shfl_xor(T v, int mask) { return emu::shfl(v, emu::lane() ^ mask); }
__device__ __forceinline__ float xor1(float v) { return emu::shfl(v, emu::lane() ^ 1); }
__device__ __forceinline__ int wave_incl_scan(int v) {
    for (int d = 1; d < 64; d <<= 1) {
        const int u = emu::shfl(v, (emu::lane() - d) & 63);
        if (emu::lane() >= d) v += u;
    }
    return v;
}
__device__ __forceinline__ int wave_bcast(int v, int lane) { return emu::shfl(v, lane); }
__device__ __forceinline__ int read_lane(int v, int lane) { return emu::shfl(v, lane); }
__device__ __forceinline__ float wave_max_bcast(float v) {
    for (int m = 32; m >= 1; m >>= 1) v = std::fmax(v, emu::shfl(v, emu::lane() ^ m));
    return v;
}
template <int N>
__device__ __forceinline__ float group_sum(float v) {  // sum over aligned groups of N = 8 / 16 lanes, every lane gets it
    for (int d = 1; d < N; d <<= 1) v += emu::shfl(v, emu::lane() ^ d);
    return v;
}
__device__ __forceinline__ float wave_sum_bcast(float v) {  // (the hardware form adds in another order: callers compare with a tolerance)
    for (int m = 32; m >= 1; m >>= 1) v += emu::shfl(v, emu::lane() ^ m);
    return v;
}
__device__ __forceinline__ unsigned long long ballot(bool p) { return emu::ballot(p); }
__device__ __forceinline__ int lane_id() { return emu::lane(); }
__device__ __forceinline__ int first_lane(int v) { return emu::shfl(v, __builtin_ctzll(emu::ballot(true))); }
__device__ __forceinline__ void setprio(int) {}
template <class T>
__device__ __forceinline__ T atomic_add(T* p, T v) { T o = *p; *p = o + v; return o; }
#else
__device__ __forceinline__ void sync() { __syncthreads(); }
template <class T>
__device__ __forceinline__ T shfl(T v, int src) { return __shfl(v, src, 64); }
template <class T>
__device__ __forceinline__ T shfl_xor(T v, int mask) { return __shfl_xor(v, mask, 64); }
// the value lane l ^ 1 holds: ONE DPP move on the vector ALU (quad_perm [1,0,3,2]).  __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt
// lgkmcnt(0) -- a round trip through the LDS per call; 64 of them in a row were 4 of the 6 us a GEMM tile spent packing its accumulators
// (profiles/r02_gemm_tile_timeline.md)
__device__ __forceinline__ float xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
// inclusive prefix sum over the wave's 64 lanes on the vector ALU: four DPP row shifts (scan inside each row of 16) and the two row
// broadcasts of gfx9 (lane 15 -> next row, lane 31 -> rows 2, 3) -- 6 adds instead of 6 ds_bpermute round trips through the LDS
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return v;
}
// the value a wave-uniform lane holds.  (v_readlane_b32 would avoid the ds_bpermute, but the scalar results it produces made the v4 GEMM
// kernels spill 200-366 VGPRs in their hot loop -- scalar register pressure pushed back into vector registers -- for no measurable gain:
// the five broadcasts of a grouped tile lookup hide under the prologue's DMA wait.)
__device__ __forceinline__ int wave_bcast(int v, int lane) { return __shfl(v, lane, 64); }
// the value of lane `lane` (a compile-time constant or otherwise wave-uniform) as a SCALAR: v_readlane_b32, no LDS round trip -- for
// values that become addresses of wave-uniform rows
__device__ __forceinline__ int read_lane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
// sum over aligned groups of N = 8 or 16 lanes (one DPP row or half of it), every lane of the group gets it: quad_perm [1,0,3,2] and
// [2,3,0,1], then row_half_mirror (lane i <-> 7 - i: the other quad of the half row -- every lane of a quad already holds the quad's sum)
// and row_mirror (i <-> 15 - i: the other half).  Each step adds the same two partial sums the xor butterfly adds (commuted), so the result
// equals `for d: v += shfl_xor(v, d)` bit for bit -- on the vector ALU instead of log2(N) ds_bpermute round trips through the LDS.
template <int N>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(N == 8 || N == 16, "one DPP row or half of it");
#define ARIA_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    ARIA_DPP_ADD(0xB1);   // quad_perm:[1,0,3,2]
    ARIA_DPP_ADD(0x4E);   // quad_perm:[2,3,0,1]
    ARIA_DPP_ADD(0x141);  // row_half_mirror
    if (N == 16) ARIA_DPP_ADD(0x140);  // row_mirror
#undef ARIA_DPP_ADD
    return v;
}
// sum over the wave's 64 lanes as a wave-uniform value on the vector ALU: the DPP ladder of wave_incl_scan (prefix sums inside each row of
// 16, the two row broadcasts), lane 63 read back as a scalar -- 6 VALU operations instead of the butterfly's 6 ds_bpermute round trips.
// The fp32 summation ORDER differs from wave_sum's butterfly: for reductions whose consumers compare with a tolerance (the decode GEMVs).
__device__ __forceinline__ float wave_sum_bcast(float v) {
#define ARIA_DPP_ADD0(ctrl, rows) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rows, 0xF, false))
    ARIA_DPP_ADD0(0x111, 0xF);  // row_shr:1
    ARIA_DPP_ADD0(0x112, 0xF);  // row_shr:2
    ARIA_DPP_ADD0(0x114, 0xF);  // row_shr:4
    ARIA_DPP_ADD0(0x118, 0xF);  // row_shr:8  -> lane 15 of every row holds the row's sum
    ARIA_DPP_ADD0(0x142, 0xA);  // row_bcast:15 into rows 1 and 3
    ARIA_DPP_ADD0(0x143, 0xC);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
#undef ARIA_DPP_ADD0
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// maximum over the wave's 64 lanes as a wave-uniform value, on the vector ALU: the DPP ladder of wave_incl_scan (row shifts inside each
// row of 16, then the two row broadcasts) with max instead of add, lane 63 read back as a scalar -- 6 VALU operations instead of 6
// ds_bpermute round trips through the LDS (max is exact, so the result equals the butterfly's bit for bit)
__device__ __forceinline__ float wave_max_bcast(float v) {
    const int ninf = int(0xff800000u);
#define ARIA_DPP_MAX(ctrl, rows) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ninf, __builtin_bit_cast(int, v), ctrl, rows, 0xF, false)))
    ARIA_DPP_MAX(0x111, 0xF);  // row_shr:1
    ARIA_DPP_MAX(0x112, 0xF);  // row_shr:2
    ARIA_DPP_MAX(0x114, 0xF);  // row_shr:4
    ARIA_DPP_MAX(0x118, 0xF);  // row_shr:8  -> lane 15 of every row holds the row's maximum
    ARIA_DPP_MAX(0x142, 0xA);  // row_bcast:15 into rows 1 and 3
    ARIA_DPP_MAX(0x143, 0xC);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
#undef ARIA_DPP_MAX
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ unsigned long long ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int first_lane(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void setprio(int) {}
template <class T>
__device__ __forceinline__ T atomic_add(T* p, T v) { return atomicAdd(p, v); }
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
    return v;
}

// ---- MFMA (matrix cores) ----
// D = A(32x16) * B(16x32) + C, bf16 inputs, f32 accumulate.
//   A fragment: lane l holds A[l & 31][8 * (l >> 5) + 0..7]
//   B fragment: lane l holds B[8 * (l >> 5) + 0..7][l & 31]
//   C/D:        reg r of lane l is C[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31]
__device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c) {
#ifdef ARIA_EMU
    struct AB { s16x8 a, b; };
    AB mine{a, b};
    emu::WaveBuf& w = emu::wbuf();
    std::memcpy(w.slot[emu::lane()], &mine, sizeof(AB));
    emu::wave_sync();
    const int l = emu::lane();
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = 0.f;
        for (int k = 0; k < 16; ++k) {
            AB ra, rb;
            std::memcpy(&ra, w.slot[row + 32 * (k >> 3)], sizeof(AB));
            std::memcpy(&rb, w.slot[col + 32 * (k >> 3)], sizeof(AB));
            acc += bf2f(bf16_t(ra.a[k & 7])) * bf2f(bf16_t(rb.b[k & 7]));
        }
        d[r] = c[r] + acc;
    }
    emu::wave_sync();
    return d;
#else
    typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b), c, 0, 0, 0);
#endif
}

// D = A(16x32) * B(32x16) + C.
//   A fragment: lane l holds A[l & 15][8 * (l >> 4) + 0..7];  B: B[8 * (l >> 4) + 0..7][l & 15]
//   C/D: reg r of lane l is C[4 * (l >> 4) + r][l & 15]
__device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c) {
#ifdef ARIA_EMU
    struct AB { s16x8 a, b; };
    AB mine{a, b};
    emu::WaveBuf& w = emu::wbuf();
    std::memcpy(w.slot[emu::lane()], &mine, sizeof(AB));
    emu::wave_sync();
    const int l = emu::lane();
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        float acc = 0.f;
        for (int k = 0; k < 32; ++k) {
            AB ra, rb;
            std::memcpy(&ra, w.slot[row + 16 * (k >> 3)], sizeof(AB));
            std::memcpy(&rb, w.slot[col + 16 * (k >> 3)], sizeof(AB));
            acc += bf2f(bf16_t(ra.a[k & 7])) * bf2f(bf16_t(rb.b[k & 7]));
        }
        d[r] = c[r] + acc;
    }
    emu::wave_sync();
    return d;
#else
    typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, a), __builtin_bit_cast(bfx8, b), c, 0, 0, 0);
#endif
}

// low 32 bits of the product of two values below 2^24 (v_mul_u32_u24: full rate; v_mul_lo_u32 is quarter rate)
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) {
#ifdef ARIA_EMU
    return a * b;
#else
    return __umul24(a, b);
#endif
}

// ---- 16-byte global / LDS access helpers ----
__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ u32x4 zero16() { return u32x4{0u, 0u, 0u, 0u}; }
// streaming 16-byte store (non-temporal: the line is not kept in the L2 for a reader that will not come from this XCD)
__device__ __forceinline__ void st16_stream(void* p, u32x4 v) {
#ifdef ARIA_EMU
    *reinterpret_cast<u32x4*>(p) = v;
#else
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
#endif
}

// ---- LDS-DMA (global_load_lds_dwordx4) and the explicit counters / barriers that pipeline it ----
// Every lane fetches 16 bytes from ITS OWN global address; the wave's 1 KiB lands lane-linearly at lds_wave_base + 16 * lane
// (lds_wave_base must be wave-uniform: it travels in M0).  Completion is tracked by vmcnt, in issue order.
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
#ifdef ARIA_EMU
    emu::glds(g, static_cast<char*>(lds_wave_base) + 16 * emu::lane());
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g),
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0);
#endif
}
// glds16 at agent scope (sc1): the source was written by another workgroup of the same launch, possibly on another XCD
__device__ __forceinline__ void glds16_agent(const void* g, void* lds_wave_base) {
#ifdef ARIA_EMU
    emu::glds(g, static_cast<char*>(lds_wave_base) + 16 * emu::lane());
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g),
                                     (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 16 /* sc1 */);
#endif
}
// The same DMA issued as the kernel's OWN instruction (inline assembly): hipcc cannot tell a later LDS read from the piece's pending LDS
// write and puts `s_waitcnt vmcnt(0)` in front of every LDS read that follows a __builtin_amdgcn_global_load_lds -- a complete drain of
// the prefetch at its first consumer.  Invisible to the compiler's counters, the piece stays in flight until the caller's wait_vm<N>().
// (Compiler-issued loads stay correct next to it: a counted wait only ever waits longer when more operations are outstanding.)
__device__ __forceinline__ void glds16_raw(const void* g, void* lds_wave_base) {
#ifdef ARIA_EMU
    emu::glds(g, static_cast<char*>(lds_wave_base) + 16 * emu::lane());
#else
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    const uint32_t lds = __builtin_amdgcn_readfirstlane(uint32_t(reinterpret_cast<uintptr_t>(lds_wave_base)));
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
#pragma clang diagnostic pop
#endif
}
// 4 bytes per lane (global_load_lds_dword): the wave's 256 bytes land lane-linearly at lds_wave_base + 4 * lane.  Same counter, same order.
__device__ __forceinline__ void glds4_raw(const void* g, void* lds_wave_base) {
#ifdef ARIA_EMU
    emu::glds(g, static_cast<char*>(lds_wave_base) + 4 * emu::lane(), 4);
#else
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    const uint32_t lds = __builtin_amdgcn_readfirstlane(uint32_t(reinterpret_cast<uintptr_t>(lds_wave_base)));
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
#pragma clang diagnostic pop
#endif
}
// wait until at most N of this wave's VMEM operations (LDS-DMA pieces included) are still outstanding
template <int N>
__device__ __forceinline__ void wait_vm() {
#ifdef ARIA_EMU
    emu::wait_vm(N);
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ void wait_lds() {  // all of this wave's LDS reads/writes have completed
#ifndef ARIA_EMU
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
// bare s_barrier: unlike __syncthreads() it carries no fence, so LDS-DMA pieces in flight stay in flight across it
__device__ __forceinline__ void raw_barrier() {
#ifdef ARIA_EMU
    emu::syncthreads();
#else
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#endif
}
// all lanes of the wave have executed what precedes (hardware: a wave runs in lockstep and its LDS operations complete in order, so this
// is only a compiler scheduling barrier; the emulator runs lanes as fibers and really synchronises them)
__device__ __forceinline__ void wave_barrier() {
#ifdef ARIA_EMU
    emu::wave_sync();
#else
    __builtin_amdgcn_wave_barrier();
#endif
}
template <int P>
__device__ __forceinline__ void wave_prio() {
#ifndef ARIA_EMU
    __builtin_amdgcn_s_setprio(P);
#endif
}
// Make the compiler finish the load that produced `v` HERE (an empty asm that reads the register): a value fetched by a global load in
// front of a loop and first used inside it otherwise gets its `s_waitcnt vmcnt(0)` placed at that first use -- inside the loop, where on
// every later iteration it drains the loop's own prefetch loads (seen in the attention kernels: the wave sat out the whole global-load
// latency of the NEXT key tile in front of its first MFMA of every tile).
template <class T>
__device__ __forceinline__ void settle(const T& v) {
#ifndef ARIA_EMU
    asm volatile("" ::"v"(v));
#endif
}
// The opposite of settle(): keep the compiler from touching `v` (hoisting arithmetic on it, and with it the wait for the load that
// produced it) before this point -- the value passes through an empty asm that "modifies" it.
template <class T>
__device__ __forceinline__ void hold(T& v) {
#ifndef ARIA_EMU
    asm volatile("" : "+v"(v));
#endif
}
__device__ __forceinline__ void sched_fence() {
#ifndef ARIA_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }

// gelu_pytorch_tanh (ViT MLP activation, transformers ACT2FN["gelu_pytorch_tanh"]); shared by the stand-alone kernel (vit.hip) and
// the GEMM epilogues so that the fused and the unfused path round identically
// (0.5 x (1 + tanh u) = x / (1 + 2^(-2 u log2 e)): one v_exp_f32 and one v_rcp_f32, both good to 1 ulp, instead of libdevice's branching
// tanhf -- ~50 instructions per value, 11 us of a 256 x 256 tile's epilogue in the ViT fc1 GEMM.)
__device__ __forceinline__ float gelu_tanh(float x);

// 2^x straight on v_exp_f32 (no denormal range fix-up: results below 2^-126 flush to 0, which is what softmax wants)
__device__ __forceinline__ float exp2_fast(float x) {
#ifdef ARIA_EMU
    return std::exp2(x);
#else
    return __builtin_amdgcn_exp2f(x);
#endif
}
__device__ __forceinline__ float rcp_fast(float x) {
#ifdef ARIA_EMU
    return 1.f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
// x * sigmoid(x) (SiLU): x / (1 + e^-x) on v_exp_f32 + v_rcp_f32.  ONE definition for the fused GEMM epilogue, the stand-alone SwiGLU
// kernels and the decode path, so that all of them round identically.
__device__ __forceinline__ float silu_fast(float a) { return a * rcp_fast(1.f + exp2_fast(-1.4426950408889634f * a)); }
// ---- r06: the two activation formulas whose evaluation sits in GEMM epilogues are written ON PAIRS of values, as explicit operation sequences
// (every fused multiply-add spelled out, no sum of products left to the compiler's contraction): the compiler then emits v_pk_mul / v_pk_fma /
// v_pk_add for everything but the two quarter-rate transcendentals, and every caller -- fused epilogue, stand-alone kernel, the scalar wrappers
// below -- executes the same IEEE operations in the same order, i.e. rounds identically.  (The scalar forms left to the auto-vectorizer packed
// only part of the work: 36 full-rate instructions per four GELU values against 22 now; the ViT fc1 launch spends ~10 us of a ~42 us tile in
// this epilogue, the SwiGLU backward ~15 of ~83.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) {
#ifdef ARIA_EMU
    return f32x2{std::fma(a.x, b.x, c.x), std::fma(a.y, b.y, c.y)};
#else
    return __builtin_elementwise_fma(a, b, c);
#endif
}
__device__ __forceinline__ f32x2 exp2_fast2(f32x2 x) { return f32x2{exp2_fast(x.x), exp2_fast(x.y)}; }
__device__ __forceinline__ f32x2 rcp_fast2(f32x2 x) { return f32x2{rcp_fast(x.x), rcp_fast(x.y)}; }
__device__ __forceinline__ f32x2 rbf2(f32x2 v) {   // both values rounded to bf16 (one v_cvt_pk_bf16_f32, two unpacks)
    const uint32_t w = pack2bf(v.x, v.y);
    return f32x2{bflo(w), bfhi(w)};
}
// Backward of y = bf16(silu(a)) * b (GroupedMLP's glu, moe_lm.py:505-507): d_a = g b silu'(a), d_b = g bf16(silu(a)), with silu(a) = a sig
// evaluated as silu_fast does (so bf16(silu(a)) here IS the value the forward multiplied by); silu'(a) = sig (1 + a (1 - sig)).  ONE definition
// for the stand-alone kernel (moe.hip) and the GEMM epilogue that absorbs it (gemm3.hip, VER 5).
__device__ __forceinline__ void swiglu_bwd_pair(f32x2 a, f32x2 b, f32x2 g, f32x2& da, f32x2& db) {
    const f32x2 one = {1.f, 1.f};
    const f32x2 sig = rcp_fast2(exp2_fast2(a * -1.4426950408889634f) + one);
    db = g * rbf2(a * sig);
    da = (g * b) * (sig * fma2(a, one - sig, one));
}
__device__ __forceinline__ void swiglu_bwd_elem(float a, float b, float g, float& da, float& db) {
    f32x2 va, vb;
    swiglu_bwd_pair(f32x2{a, a}, f32x2{b, b}, f32x2{g, g}, va, vb);
    da = va.x, db = vb.x;
}
// gelu_pytorch_tanh: x / (1 + 2^(-k2 (x + 0.044715 x^3))), the cubic as fma(0.044715 x, x x, x)
__device__ __forceinline__ f32x2 gelu_tanh2(f32x2 x) {
    const float k2 = 2.f * 0.7978845608028654f * 1.4426950408889634f;  // 2 sqrt(2/pi) log2(e)
    const f32x2 one = {1.f, 1.f};
    const f32x2 u = fma2(x * 0.044715f, x * x, x);
    return x * rcp_fast2(exp2_fast2(u * -k2) + one);
}
__device__ __forceinline__ float gelu_tanh(float x) { return gelu_tanh2(f32x2{x, x}).x; }

// ds_read_b64_tr_b16: every lane passes the LDS address of 4 consecutive bf16 (8-byte aligned); within each 16-lane group
// lane q receives element (q & 3) of the four lanes 4j + (q >> 2), j = 0..3 (semantics verified on hardware by
// tests/test_gpu_probes.py).  Used to read row-major [k][n] tiles as k-contiguous MFMA fragments.
__device__ __forceinline__ s16x4 ds_read_tr16(const bf16_t* p) {
#ifdef ARIA_EMU
    emu::WaveBuf& w = emu::wbuf();
    const int l = emu::lane();
    std::memcpy(w.slot[l], &p, sizeof(p));
    emu::wave_sync();
    s16x4 r;
    const int base = l & ~15, q = l & 15;
    for (int j = 0; j < 4; ++j) {
        const bf16_t* src;
        std::memcpy(&src, w.slot[base + 4 * j + (q >> 2)], sizeof(src));
        r[j] = short(src[q & 3]);
    }
    emu::wave_sync();
    return r;
#else
    typedef short v4s __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p));
#endif
}

// TopKRouter.routing on the E logits of one token, exactly as route_kernel (moe.hip) -- ONE definition for the decode engine (decode.hip) and the fused router launch (moe.hip router_fused_kernel): k rounds of arg-max with ties to the lowest expert
// id, softmax over the selected logits in fp32, scores cast to bf16.  One wave; the logits come from a regular (multi-workgroup) GEMV --
// a single workgroup reading the whole 320 KB gate matrix cost 16 us per layer.
// returns the expert id of slot `want` (wave-uniform); lanes < k also get (score, id) of their own slot
__device__ __forceinline__ int route_one_token(const bf16_t* logits, int E, int k, int l, int want, float& my_score, int& my_idx) {
    float val[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) val[i] = (l + 64 * i < E) ? bf2f(logits[l + 64 * i]) : -INFINITY;
    float top[8];
    int topi = -1, wanted = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        top[j] = -INFINITY;
        if (j < k) {
            // round j: the largest remaining logit, ties to the LOWEST expert id.  The maximum comes from a DPP ladder (vector ALU) and the
            // id from ballots over the four id-ordered slots -- every workgroup of the up-projection runs this in front of its first weight
            // load, and the butterfly form (two ds_bpermute round trips per step, 6 steps, k rounds) was ~2 us of pure latency there
            const float m = wave_max_bcast(fmaxf(fmaxf(val[0], val[1]), fmaxf(val[2], val[3])));
            int bi = -1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned long long hit = ballot(val[i] == m);
                if (bi < 0 && hit) bi = __builtin_ctzll(hit) + 64 * i;
            }
            if (bi < 0) bi = 0;  // (NaN logits: no lane compares equal -- stay in range)
            top[j] = m;
            if (l == j) topi = bi;
            if (j == want) wanted = bi;
            if ((bi & 63) == l) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i == (bi >> 6)) val[i] = -INFINITY;
            }
        }
    }
    const float mx = top[0];
    float den = 0.f, mine = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < k) den += expf(top[j] - mx);
        if (j == l) mine = top[j];
    }
    my_score = expf(mine - mx) / den;
    my_idx = topi;
    return wanted;
}

}  // namespace ad

