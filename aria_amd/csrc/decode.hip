// Single-token decode engine for the gptfast surface (gptfast/model.py:178-234 Transformer.forward with one new token;
// ConditionalFeedForward's T < 50 path :318-325; KVCache.update :67-93): BASELINE.md's only published numbers are the reference's
// generate() tokens/s, and one token is 7.7 GB of weights against ~0.4 GF per GB -- an HBM-streaming problem, not an MFMA one.
//
// The tile GEMMs spend a token on ~24 launches per layer from Python (16.7 ms/token, launch-bound) and, with one valid row in a
// 128-row tile, keep only N/128 workgroups busy.  Here:
//   * every projection is a GEMV: a wave owns R weight rows, lanes stride the reduction in 16-byte chunks (fully coalesced row reads),
//     fp32 accumulate, wave reduce -- N / (4 R) workgroups, so even N = 2560 fills the chip;
//   * RMSNorm is folded into the consumer GEMV (each wave re-normalises the 2560-vector it needs anyway, with the same lane <-> chunk
//     mapping and reduction order as rmsnorm_fwd_kernel: identical rstd), residual adds into the GEMV epilogue, SwiGLU into the
//     up-projection pair, RoPE + KV-cache write into one kernel, the six routed experts are indexed on the device (no gather of weights);
//   * ONE C call walks all layers and enqueues 6 launches per layer back to back (no Python, no allocation, no host sync): the
//     position lives on the device, so the same enqueue sequence is valid for every token.  Per layer: qkv | attention (RoPE + cache
//     write inside) | wo + residual | router logits + shared up-projection | routed up-projection (top-k inside) | every
//     down-projection + combine + residual;
//   * what a kernel does in front of its first weight byte is kept short: rows are requested before the activation vector is fetched and
//     normalised, and the wave reductions on the way (routing maximum, per-key score, row sums) run on DPP instead of LDS round trips --
//     at ~6 TB/s of streaming a layer's 258 MB take 43 us, and every microsecond of fixed cost per kernel is 2 % of that;
//   * sampling (gptfast/generate.py:35-58) is one launch as well (aria_sample_topk).
// Rounding points mirror the tile path (GEMM outputs, norm, SwiGLU, residual adds are each rounded to bf16 where the reference
// materialises a bf16 tensor); only the fp32 summation ORDER inside a dot product differs from the MFMA kernels.
#include "aria_device.h"
#include "aria_hip.h"
#include <cmath>
#include <cstdlib>

namespace {
using namespace ad;

__device__ __forceinline__ float silu(float a) { return silu_fast(a); }

// x[K] (bf16) -> this lane's chunks c = l + 64 i (i < NC) as packed bf16 pairs, optionally RMS-normalised exactly like
// rmsnorm_fwd_kernel (norm.hip: same lane <-> chunk mapping and reduction order, so the same rstd).  Chunks past K read as zeros;
// no branches, so all loads are in flight together.
template <int NC>
__device__ __forceinline__ void load_vector(u32x4 (&xv)[NC], const bf16_t* x, const bf16_t* norm_w, float eps, int K, int l) {
    const int nch = K >> 3;
    u32x4 wv[NC];
    const bf16_t* nw = norm_w ? norm_w : x;  // (no branch around the loads: with one, the compiler fetched chunk after chunk, a wait in between)
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int cc = min(l + 64 * i, nch - 1);
        xv[i] = ld16(x + cc * 8);
        wv[i] = ld16(nw + cc * 8);
    }
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[i][q] = (l + 64 * i < nch) ? xv[i][q] : 0u;
    if (!norm_w) return;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v0 = bflo(xv[i][q]), v1 = bfhi(xv[i][q]);
            ss += v0 * v0 + v1 * v1;
        }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / float(K) + eps);
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            xv[i][q] = pack2bf(bflo(wv[i][q]) * rbf(bflo(xv[i][q]) * r), bfhi(wv[i][q]) * rbf(bfhi(xv[i][q]) * r));
}

// 16 bytes of a WEIGHT stream: every byte is read once per token by one CU, so the load is marked non-temporal (it does not displace
// what the L2 / MALL hold for re-use: the guide's "nt-weights" row -- issued -> landed -18 % on this chip's decode GEMVs).
// -DARIA_DECODE_NT=0: plain loads (A/B builds, tools/gpu_r4_s5.sh).
#ifndef ARIA_DECODE_NT
#define ARIA_DECODE_NT 1
#endif
__device__ __forceinline__ u32x4 ldw16(const bf16_t* p) {
#if defined(ARIA_EMU) || !ARIA_DECODE_NT
    return ld16(p);
#else
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#endif
}
// R consecutive weight rows as 16-byte chunks c = l + 64 i per lane.  Rows past the end are clamped (their results are never stored), no
// branches: all R * NC loads of a wave are in flight together.  Kernels issue these BEFORE they fetch and normalise the vector: the vector
// (L2 hits + a wave reduction + rsqrt, ~1.5 us of dependent latency) then hides under the HBM latency of the rows instead of in front of it.
template <int R, int NC>
__device__ __forceinline__ void load_rows(u32x4 (&a)[R][NC], const bf16_t* w, long long ldw, int row0, int nrows, int K, int l) {
    const int nch = K >> 3;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bf16_t* row = w + (long long)min(row0 + r, nrows - 1) * ldw;
#pragma unroll
        for (int i = 0; i < NC; ++i) a[r][i] = ldw16(row + min(l + 64 * i, nch - 1) * 8);
    }
}
// dot products of the loaded rows with the lane-distributed vector (chunks past K read as zeros there); every lane returns the full sums
template <int R, int NC>
__device__ __forceinline__ void dot_loaded(float (&acc)[R], const u32x4 (&a)[R][NC], const u32x4 (&xv)[NC]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) s = dot2bf(a[r][i][q], xv[i][q], s);
        acc[r] = wave_sum_bcast(s);  // (DPP ladder: no LDS round trips at the tail of every GEMV wave)
    }
}
template <int R, int NC>
__device__ __forceinline__ void dot_rows(float (&acc)[R], const bf16_t* w, long long ldw, int row0, int nrows, const u32x4 (&xv)[NC],
                                         int K, int l) {
    u32x4 a[R][NC];
    load_rows<R, NC>(a, w, ldw, row0, nrows, K, l);
    dot_loaded<R, NC>(acc, a, xv);
}

// y[n] = bf16(W[n,:] . xn) (+ residual[n], added to the ROUNDED product like the stand-alone add kernel)
template <int R, int NC>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* W, long long ldw, const bf16_t* x, const bf16_t* norm_w, float eps, int K,
                                                   int N, const bf16_t* residual, bf16_t* y) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + w) * R;
    if (row0 >= N) return;
    u32x4 xv[NC], a[R][NC];
    float acc[R];
    load_rows<R, NC>(a, W, ldw, row0, N, K, l);
    load_vector<NC>(xv, x, norm_w, eps, K, l);
    dot_loaded<R, NC>(acc, a, xv);
    if (l == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (row0 + r < N) y[row0 + r] = residual ? f2bf(bf2f(residual[row0 + r]) + rbf(acc[r])) : f2bf(acc[r]);
    }
}

// Up-projection pair + SwiGLU for the k routed experts (e_j = idx[j]) AND the shared expert in one launch: grid.y = k + ns, the shared
// expert's [ns*I, D] matrices are ns further "experts" of I rows each, so act rows k .. k+ns-1 are its activation vector of length ns*I.
//   act[j][n] = bf16( bf16(silu(bf16(W1[n,:] . xn))) * bf16(W3[n,:] . xn) )
// `logits` (E router logits of the token): every wave derives the routing itself (one load + ~40 shuffles: cheaper than a launch of
// its own); workgroup (0, 0) publishes scores / idx for the down-projection and the combine.
template <int R, int NC>
__global__ __launch_bounds__(256) void expert_up_kernel(const bf16_t* W1, const bf16_t* W3, const bf16_t* S1, const bf16_t* S3,
                                                        const bf16_t* logits, int E, bf16_t* scores, int32_t* idx, int k, const bf16_t* x,
                                                        const bf16_t* norm_w, float eps, int K, int I, bf16_t* act) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, j = blockIdx.y;
    const int row0 = (blockIdx.x * 4 + w) * R;
    if (row0 >= I) return;
    float my_score;
    int my_idx;
    const int e = route_one_token(logits, E, k, l, j, my_score, my_idx);
    if (blockIdx.x == 0 && j == 0 && w == 0 && l < k) {
        scores[l] = f2bf(my_score);
        idx[l] = my_idx;
    }
    const long long stride = (long long)I * K;
    const bf16_t* w1 = j < k ? W1 + (long long)e * stride : S1 + (long long)(j - k) * stride;
    const bf16_t* w3 = j < k ? W3 + (long long)e * stride : S3 + (long long)(j - k) * stride;
    u32x4 xv[NC];
    float a1[R], a3[R];
    load_vector<NC>(xv, x, norm_w, eps, K, l);
    dot_rows<R, NC>(a1, w1, K, row0, I, xv, K, l);
    dot_rows<R, NC>(a3, w3, K, row0, I, xv, K, l);
    if (l == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (row0 + r < I) act[(long long)j * I + row0 + r] = f2bf(rbf(silu(rbf(a1[r]))) * rbf(a3[r]));
    }
}

// Down-projections: out[j] = bf16(W2[e_j] . act[j]) for the routed experts (reduction I, NCI chunks per lane) and
// out[k] = bf16(S2 . act[k..]) for the shared expert (reduction ns*I, NCS chunks per lane), grid.y = k + 1
template <int R, int NCI, int NCS>
__global__ __launch_bounds__(256) void expert_down_kernel(const bf16_t* W2, const bf16_t* S2, const int32_t* idx, int k, int ns,
                                                          const bf16_t* act, int I, int N, bf16_t* out) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, j = blockIdx.y;
    const int row0 = (blockIdx.x * 4 + w) * R;
    if (row0 >= N) return;
    float acc[R];
    if (j < k) {  // block-uniform
        u32x4 xv[NCI];
        load_vector<NCI>(xv, act + (long long)j * I, nullptr, 0.f, I, l);
        dot_rows<R, NCI>(acc, W2 + (long long)idx[j] * N * I, I, row0, N, xv, I, l);
    } else {
        u32x4 xv[NCS];
        load_vector<NCS>(xv, act + (long long)j * I, nullptr, 0.f, ns * I, l);
        dot_rows<R, NCS>(acc, S2, (long long)ns * I, row0, N, xv, ns * I, l);
    }
    if (l == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (row0 + r < N) out[(long long)j * N + row0 + r] = f2bf(acc[r]);
    }
}

// ---- fused schedule (6 launches per layer instead of 7; ARIA_DECODE_FUSE=0 keeps the 7-launch one for A/B and as the fallback for
// widths the fused down-projection has no instantiation for) ------------------------------------------------------------------------
// (a) Router logits AND the shared expert's up-projection pair + SwiGLU in ONE launch.  The router GEMV alone is 320 KB of gate matrix
//     on 8 workgroups: pure launch + ramp latency in front of the one kernel that needs its result.  The shared expert (a quarter of
//     the layer's up-projection bytes) needs no routing, so its rows ride along: blocks [0, nrb) compute the E logits exactly as
//     gemv_kernel<2, NC> does (same rows per wave, same summation order -> the same bits), the rest the shared activation vector exactly
//     as expert_up_kernel's j >= k slices do.  The routed up-projection (expert_up_kernel, grid.y = k) follows and reads the logits.
template <int R, int NC>
__global__ __launch_bounds__(256) void router_shared_up_kernel(const bf16_t* gate, int E, bf16_t* logits, const bf16_t* S1, const bf16_t* S3,
                                                               int rows_s, const bf16_t* x, const bf16_t* norm_w, float eps, int K,
                                                               bf16_t* act_s, int nrb) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32x4 xv[NC];
    if (int(blockIdx.x) < nrb) {  // block-uniform
        const int row0 = (blockIdx.x * 4 + w) * 2;
        if (row0 >= E) return;
        float acc[2];
        u32x4 a[2][NC];
        load_rows<2, NC>(a, gate, K, row0, E, K, l);
        load_vector<NC>(xv, x, norm_w, eps, K, l);
        dot_loaded<2, NC>(acc, a, xv);
        if (l == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (row0 + r < E) logits[row0 + r] = f2bf(acc[r]);
        }
        return;
    }
    const int row0 = ((int(blockIdx.x) - nrb) * 4 + w) * R;
    if (row0 >= rows_s) return;
    float a1[R], a3[R];
    u32x4 r1[R][NC], r3[R][NC];
    load_rows<R, NC>(r1, S1, K, row0, rows_s, K, l);
    load_rows<R, NC>(r3, S3, K, row0, rows_s, K, l);
    load_vector<NC>(xv, x, norm_w, eps, K, l);
    dot_loaded<R, NC>(a1, r1, xv);
    dot_loaded<R, NC>(a3, r3, xv);
    if (l == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (row0 + r < rows_s) act_s[row0 + r] = f2bf(rbf(silu(rbf(a1[r]))) * rbf(a3[r]));
    }
}

// (b) All k + 1 down-projections of ONE output row per wave, and the combine (token_unpermutation + shared add + residual,
//     combine_kernel) as the epilogue: the k + 1 partial rows never visit HBM and the layer loses a launch.
//     The workgroup's four waves share the activation vectors, kept in LDS as lane-chunk images (chunk c of expert j at
//     [j][c], zeros past the end of the reduction) -- a lane's operand is one conflict-free ds_read_b128 at the point of use instead of
//     (k NCI + NCS) x 4 live registers; every weight row chunk of the wave's row (k NCI + NCS 16-byte loads per lane, 31 at Aria's
//     widths) is issued before the first use.  Per-expert dot products run in dot_rows' order and the epilogue rounds where
//     expert_down_kernel / combine_kernel materialise bf16, so the hidden state equals the 7-launch schedule's bit for bit.
constexpr int DOWN_KMAX = 6;  // routed experts the fused form holds weight rows for (Aria: top-6); wider routing takes the 7-launch schedule
__device__ const uint32_t decode_zero_page[64] = {};  // 256 zero bytes: LDS-DMA source of the image chunks past the end of a reduction
template <int NCI, int NCS>
__global__ __launch_bounds__(256) void expert_down_combine_kernel(const bf16_t* W2, const bf16_t* S2, const int32_t* idx, const bf16_t* scores,
                                                                  int k, int ns, const bf16_t* act, int I, int N, const bf16_t* h,
                                                                  bf16_t* out) {
    ARIA_DYN_SMEM(smem);
    u32x4* sa = reinterpret_cast<u32x4*>(smem);  // [k][NCI * 64] routed images, then [NCS * 64] the shared expert's
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const int nchI = I >> 3, nchS = (ns * I) >> 3;
    const int n = blockIdx.x * 4 + w, nn = min(n, N - 1);  // rows past the end are clamped (computed, never stored): every wave reaches the barrier
    // activation images by LDS-DMA (no staging registers; issued first, so they are the oldest operations in flight): instruction g
    // fills image chunks 64 g .. 64 g + 63, the four waves take every fourth one
    const int ninstr = k * NCI + NCS;
    for (int g = w; g < ninstr; g += 4) {
        const bf16_t* src;
        if (g < k * NCI) {
            const int j = g / NCI, cc = (g % NCI) * 64 + l;
            src = cc < nchI ? act + (long long)j * I + cc * 8 : reinterpret_cast<const bf16_t*>(decode_zero_page) + 8 * (l & 15);
        } else {
            const int cc = (g - k * NCI) * 64 + l;
            src = cc < nchS ? act + (long long)k * I + cc * 8 : reinterpret_cast<const bf16_t*>(decode_zero_page) + 8 * (l & 15);
        }
        glds16(src, sa + 64 * g);
    }
    // expert ids and scores of the k slots: ONE vector load each (lane j = slot j), broadcast as scalars -- an idx[j] read in front of every
    // expert's rows made each expert wait for everything issued before it (six dependent round trips in the first build's ISA)
    int my_e = 0, my_sc = 0;
    if (l < k) {
        my_e = idx[l];
        my_sc = int(uint32_t(scores[l]) << 16);
    }
    int e[DOWN_KMAX];
    float sc[DOWN_KMAX];
#pragma unroll
    for (int j = 0; j < DOWN_KMAX; ++j) {
        e[j] = read_lane(my_e, j);
        sc[j] = __builtin_bit_cast(float, read_lane(my_sc, j));
    }
    // every weight chunk of the wave's row: k NCI + NCS 16-byte loads per lane in flight before the first use
    u32x4 wr[DOWN_KMAX][NCI], ws[NCS];
    {
        const bf16_t* row = S2 + (long long)nn * ns * I;
#pragma unroll
        for (int i = 0; i < NCS; ++i) ws[i] = ldw16(row + min(l + 64 * i, nchS - 1) * 8);
    }
#pragma unroll
    for (int j = 0; j < DOWN_KMAX; ++j)
        if (j < k) {
            const bf16_t* row = W2 + ((long long)e[j] * N + nn) * I;
#pragma unroll
            for (int i = 0; i < NCI; ++i) wr[j][i] = ldw16(row + min(l + 64 * i, nchI - 1) * 8);
        }
    wait_vm<0>();  // this wave's image pieces have landed (the weight rows with them: they are needed next anyway)
    sync();
    float accs = 0.f;
#pragma unroll
    for (int j = 0; j < DOWN_KMAX; ++j)
        if (j < k) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NCI; ++i) {
                const u32x4 xv = sa[j * NCI * 64 + l + 64 * i];
#pragma unroll
                for (int q = 0; q < 4; ++q) s = dot2bf(wr[j][i][q], xv[q], s);
            }
            s = wave_sum_bcast(s);
            accs += rbf(rbf(s) * sc[j]);  // bf16(eo_j * score_j), summed in fp32 in slot order (combine_kernel)
        }
    float sh = 0.f;
#pragma unroll
    for (int i = 0; i < NCS; ++i) {
        const u32x4 xv = sa[k * NCI * 64 + l + 64 * i];
#pragma unroll
        for (int q = 0; q < 4; ++q) sh = dot2bf(ws[i][q], xv[q], sh);
    }
    sh = wave_sum_bcast(sh);
    if (l == 0 && n < N) out[n] = f2bf(bf2f(h[n]) + rbf(rbf(accs) + rbf(sh)));
}

// stand-alone form of the routing the up-projection runs in front of its rows (aria_decode_route: parity tests against aria_moe_route)
__global__ __launch_bounds__(64) void router_topk_kernel(const bf16_t* logits, int E, int k, bf16_t* scores, int32_t* idx) {
    const int l = threadIdx.x & 63;
    float my_score;
    int my_idx;
    route_one_token(logits, E, k, l, 0, my_score, my_idx);
    if (l < k) {
        scores[l] = f2bf(my_score);
        idx[l] = my_idx;
    }
}

// token_unpermutation + shared add + residual in one pass (unpermute_kernel of moe.hip followed by add_kernel of norm.hip):
//   m = bf16( bf16(sum_j bf16(eo_j * score_j)) + shared ),  out = bf16(h + m)
__global__ __launch_bounds__(256) void combine_kernel(const bf16_t* eo, const bf16_t* scores, int k, const bf16_t* shared, const bf16_t* h,
                                                      bf16_t* out, int D) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (D >> 3)) return;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int j = 0; j < k; ++j) {
        const float sc = bf2f(scores[j]);
        const u32x4 v = ld16(eo + (long long)j * D + c * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[2 * q] += rbf(bflo(v[q]) * sc);
            acc[2 * q + 1] += rbf(bfhi(v[q]) * sc);
        }
    }
    const u32x4 a = ld16(shared + c * 8), hv = ld16(h + c * 8);
    u32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float m0 = rbf(rbf(acc[2 * q]) + bflo(a[q])), m1 = rbf(rbf(acc[2 * q + 1]) + bfhi(a[q]));
        o[q] = pack2bf(bflo(hv[q]) + m0, bfhi(hv[q]) + m1);
    }
    st16(out + c * 8, o);
}

// RoPE (interleaved pairs, bf16 freqs_cis cache, one rounding: rope_interleaved_kernel of norm.hip) on q in place and on k while it
// moves into the cache at the device-side position; v is copied.  Also publishes kv_len = pos + 1 for the attention kernel.
__global__ __launch_bounds__(256) void rope_cache_kernel(bf16_t* qkv, const bf16_t* fc, const int32_t* pos, bf16_t* k_cache,
                                                         bf16_t* v_cache, int D, int hd, int32_t* kv_len) {
    const int ps = pos[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) kv_len[0] = ps + 1;
    const int nch = D >> 3, cph = hd >> 3;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < 3 * nch; c += gridDim.x * blockDim.x) {
        const int part = c / nch, cc = c % nch;
        const u32x4 a = ld16(qkv + (long long)part * D + cc * 8);
        if (part == 2) {
            st16(v_cache + (long long)ps * D + cc * 8, a);
            continue;
        }
        const u32x4 f = ld16(fc + (long long)ps * hd + (cc % cph) * 8);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x0 = bflo(a[q]), x1 = bfhi(a[q]), cs = bflo(f[q]), sn = bfhi(f[q]);
            o[q] = pack2bf(x0 * cs - x1 * sn, x1 * cs + x0 * sn);
        }
        if (part == 0)
            st16(qkv + cc * 8, o);
        else
            st16(k_cache + (long long)ps * D + cc * 8, o);
    }
}

// Attention of ONE new token over the static cache, fused with RoPE and the cache write (gptfast/model.py:413-447, KVCache.update
// :67-93): 4 waves per workgroup.  HD/8 lanes share a key (8 features each, one 16-byte load per lane and key), so a wave works on
// 64/(HD/8) keys at a time, 4 deep (all K and V loads of an iteration are issued before the first use); every lane group keeps an online
// softmax state (m, l, o[8]) which is merged across groups and waves at the end (flash-decoding).  Softmax in the log2 domain, P rounded
// to bf16 before it multiplies V (what the flash kernel feeds its MFMA), fp32 accumulation.  The generic flash kernel spends ~19 us per
// layer on this (20 workgroups built for 256 queries); decode_attn_kernel ~5.
//
// DecodeAttn<HD>::run is the body shared by the two kernels below: keys [kbeg, kend) of one head, `writes_new` (block-uniform) = this
// workgroup owns the new position and puts the rotated key / value into the cache before reading it back.  On return the lanes with
// w == 0 && grp == 0 hold the UNNORMALISED state (m, lsum, o[8] for features sub*8 ..) of the whole range.
template <int HD, int NWV = 4>
struct DecodeAttn {
    // NWV waves per workgroup, U keys per lane group and pass: NWV x KPW x U keys per pass over the range, all K / V rows of a pass in
    // flight before the first use.  Round 3: the one-workgroup-per-head kernel went from 4 waves x U 4 (64 keys per pass at hd 128) to
    // 16 waves x U 8 (512): at the 300-500 keys of a chat prompt its loop was 6 DEPENDENT HBM round trips -- 13 us per layer in the
    // kernel trace, the third largest kernel of a token, for 3.4 MB of cache -- and is one now.  (Requesting the next pass ahead of the
    // current one from two register sets did not survive the compiler: it merged the two bodies and drained the loads to do so.)
    static constexpr int LPK = HD / 8, KPW = 64 / LPK, U = 8, PER_ITER = NWV * KPW * U;

    static __device__ __forceinline__ u32x4 rope(const u32x4& a, const u32x4& f) {
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x0 = bflo(a[q]), x1 = bfhi(a[q]), cs = bflo(f[q]), sn = bfhi(f[q]);
            o[q] = pack2bf(x0 * cs - x1 * sn, x1 * cs + x0 * sn);
        }
        return o;
    }

    static __device__ __forceinline__ void merge(float& m, float& lsum, float (&o)[8], float m2, float l2, const float (&o2)[8]) {
        const float mm = fmaxf(m, m2);
        const float a = mm == -INFINITY ? 0.f : exp2_fast(m - mm), b = mm == -INFINITY ? 0.f : exp2_fast(m2 - mm);
        lsum = lsum * a + l2 * b;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = o[e] * a + o2[e] * b;
        m = mm;
    }

    // One wave's share of the range: wave `w` of NWV.  On return every lane group of the wave holds the wave's unnormalised state.
    static __device__ __forceinline__ void wave_state(const bf16_t* qkv, const bf16_t* fc, int ps, bf16_t* k_cache, bf16_t* v_cache, int D,
                                                      float scale, int head, int kbeg, int kend, bool writes_new, int w, float& m,
                                                      float& lsum, float (&o)[8]) {
        const int l = threadIdx.x & 63, sub = l % LPK, grp = l / LPK;
        const long long col = (long long)head * HD + sub * 8;
        const u32x4 f = ld16(fc + (long long)ps * HD + sub * 8);
        const u32x4 qr = rope(ld16(qkv + col), f);
        // the new token's rotated key / value slice, in EVERY lane group: the pass that covers position ps takes it from these registers, so
        // nobody waits for the cache write below to become visible (it was: store, barrier, read back -- ~2 us in front of the first K / V load)
        const u32x4 knew = rope(ld16(qkv + D + col), f), vnew = ld16(qkv + 2 * D + col);
        if (writes_new && w == 0 && grp == 0) {
            st16(k_cache + (long long)ps * D + col, knew);
            st16(v_cache + (long long)ps * D + col, vnew);
        }
        const float scale2 = scale * 1.4426950408889634f;
        m = -INFINITY, lsum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
        for (int j0 = kbeg; j0 < kend; j0 += PER_ITER) {  // block-uniform trip count (zero for an empty range)
            u32x4 kx[U], vx[U];
            int key[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                key[u] = j0 + (w * U + u) * KPW + grp;
                const long long row = (long long)min(key[u], kend - 1) * D + col;
                kx[u] = ld16(k_cache + row);
                vx[u] = ld16(v_cache + row);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (key[u] == ps) {  // (the row being written by this very workgroup)
                    kx[u] = knew;
                    vx[u] = vnew;
                }
                float sc = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) sc = dot2bf(kx[u][q], qr[q], sc);
                sc = group_sum<LPK>(sc);  // the HD / 8 lanes of a key (DPP: no LDS round trips in front of every key's softmax step)
                const float s2 = key[u] < kend ? sc * scale2 : -INFINITY;
                const float m_new = fmaxf(m, s2);
                if (m_new == -INFINITY) continue;  // nothing seen yet by this lane group (uniform within the group)
                const float alpha = exp2_fast(m - m_new), p = exp2_fast(s2 - m_new), pb = rbf(p);
                lsum = lsum * alpha + p;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[2 * q] = o[2 * q] * alpha + pb * bflo(vx[u][q]);
                    o[2 * q + 1] = o[2 * q + 1] * alpha + pb * bfhi(vx[u][q]);
                }
                m = m_new;
            }
        }
        // merge the lane groups of a wave (lanes with equal `sub`), then the waves through LDS
#pragma unroll
        for (int d = LPK; d < 64; d <<= 1) {
            float o2[8];
            const float m2 = shfl_xor(m, d), l2 = shfl_xor(lsum, d);
#pragma unroll
            for (int e = 0; e < 8; ++e) o2[e] = shfl_xor(o[e], d);
            merge(m, lsum, o, m2, l2, o2);
        }
    }
    // lane group 0 of a wave leaves the wave's state in red[w]
    static __device__ __forceinline__ void publish(float (*red)[LPK][10], int w, float m, float lsum, const float (&o)[8]) {
        const int l = threadIdx.x & 63, sub = l % LPK, grp = l / LPK;
        if (grp == 0) {
            red[w][sub][0] = m;
            red[w][sub][1] = lsum;
#pragma unroll
            for (int e = 0; e < 8; ++e) red[w][sub][2 + e] = o[e];
        }
    }
    // The NWV wave states as a tree, run by ONE wave: lane group g folds waves g, g + NG, ..., then the groups merge by butterfly.
    // OWN0: the wave IS wave 0 and lane group 0 starts from its registers (which hold exactly what it published in red[0]); otherwise
    // group 0 reads red[0] back -- the same bits, so whichever wave folds, the result is the same.
    template <bool OWN0>
    static __device__ __forceinline__ void fold(const float (*red)[LPK][10], float& m, float& lsum, float (&o)[8]) {
        const int l = threadIdx.x & 63, sub = l % LPK, grp = l / LPK;
        constexpr int NG = 64 / LPK;
        if (grp > 0) {
            m = -INFINITY, lsum = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.f;
        } else if (!OWN0) {
            m = red[0][sub][0], lsum = red[0][sub][1];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = red[0][sub][2 + e];
        }
        for (int ww = grp ? grp : NG; ww < NWV; ww += NG) {
            float o2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o2[e] = red[ww][sub][2 + e];
            const float m2 = red[ww][sub][0], l2 = red[ww][sub][1];
            merge(m, lsum, o, m2, l2, o2);
        }
#pragma unroll
        for (int d = LPK; d < 64; d <<= 1) {
            float o2[8];
            const float m2 = shfl_xor(m, d), l2 = shfl_xor(lsum, d);
#pragma unroll
            for (int e = 0; e < 8; ++e) o2[e] = shfl_xor(o[e], d);
            merge(m, lsum, o, m2, l2, o2);
        }
    }
    static __device__ __forceinline__ void run(const bf16_t* qkv, const bf16_t* fc, int ps, bf16_t* k_cache, bf16_t* v_cache, int D,
                                               float scale, int head, int kbeg, int kend, bool writes_new, float (&red)[NWV][LPK][10],
                                               float& m, float& lsum, float (&o)[8]) {
        const int w = threadIdx.x >> 6;
        wave_state(qkv, fc, ps, k_cache, v_cache, D, scale, head, kbeg, kend, writes_new, w, m, lsum, o);
        publish(red, w, m, lsum, o);
        sync();
        if (w == 0) fold<true>(red, m, lsum, o);
    }
};

// grid = heads: one workgroup (16 waves) walks the whole context of its head and normalises
constexpr int DECODE_ATTN_WAVES = 16;
template <int HD>
__global__ __launch_bounds__(DECODE_ATTN_WAVES * 64) void decode_attn_kernel(const bf16_t* qkv, const bf16_t* fc, const int32_t* pos,
                                                                            bf16_t* k_cache, bf16_t* v_cache, bf16_t* out, int D, float scale) {
    using A = DecodeAttn<HD, DECODE_ATTN_WAVES>;
    ARIA_SMEM_STATIC float red[DECODE_ATTN_WAVES][A::LPK][10];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, sub = l % A::LPK, grp = l / A::LPK, head = blockIdx.x;
    const int ps = pos[0];
    float m, lsum, o[8];
    A::run(qkv, fc, ps, k_cache, v_cache, D, scale, head, 0, ps + 1, true, red, m, lsum, o);
    if (w == 0 && grp == 0) {
        const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
        u32x4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = pack2bf(o[2 * q] * inv, o[2 * q + 1] * inv);
        st16(out + (long long)head * HD + sub * 8, r);
    }
}

// Long contexts: one workgroup per head streams 2 * nkeys * HD * 2 bytes alone (8.4 MB at 16 K keys, 20 workgroups on 256 CUs).  The
// split form cuts the keys of a head into NS contiguous ranges (grid = heads x NS, ranges derived from the DEVICE-side position, a
// multiple of the per-iteration key count so every range but the last is full), each workgroup leaving its unnormalised online-softmax
// state (m, l, o[HD], fp32) in part[head][split][2 + HD]; decode_attn_merge_kernel folds the NS states.  The workgroup whose range
// holds the new position writes the rotated key / value into the cache first (nobody else reads that row).  Same per-key arithmetic as
// decode_attn_kernel; only the order in which partial states are merged differs.
constexpr int DECODE_MAX_SPLITS = 32;

template <int HD>
__global__ __launch_bounds__(256) void decode_attn_split_kernel(const bf16_t* qkv, const bf16_t* fc, const int32_t* pos, bf16_t* k_cache,
                                                                bf16_t* v_cache, float* part, int D, float scale) {
    using A = DecodeAttn<HD>;
    ARIA_SMEM_STATIC float red[4][A::LPK][10];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, sub = l % A::LPK, grp = l / A::LPK, head = blockIdx.x, split = blockIdx.y;
    const int NS = gridDim.y;
    const int ps = pos[0], nkeys = ps + 1;
    const int chunk = ((nkeys + NS - 1) / NS + A::PER_ITER - 1) / A::PER_ITER * A::PER_ITER;
    const int kbeg = split * chunk, kend = min(nkeys, kbeg + chunk);
    float m, lsum, o[8];
    A::run(qkv, fc, ps, k_cache, v_cache, D, scale, head, kbeg, kend, ps >= kbeg && ps < kend, red, m, lsum, o);
    if (w == 0 && grp == 0) {
        float* dst = part + ((long long)head * NS + split) * (HD + 2);
        if (sub == 0) {
            dst[0] = m;
            dst[1] = lsum;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[2 + sub * 8 + e] = o[e];
    }
}

// grid = heads, HD threads: thread f folds feature f of the NS partial states of its head (m, l are read by every thread: broadcast loads)
__global__ void decode_attn_merge_kernel(const float* part, bf16_t* out, int NS, int HD) {
    const int head = blockIdx.x, f = threadIdx.x;
    const float* src = part + (long long)head * NS * (HD + 2);
    float mm = -INFINITY;
    for (int s = 0; s < NS; ++s) mm = fmaxf(mm, src[s * (HD + 2)]);
    float lsum = 0.f, acc = 0.f;
    for (int s = 0; s < NS; ++s) {
        const float ms = src[s * (HD + 2)];
        const float wgt = ms == -INFINITY ? 0.f : exp2_fast(ms - mm);
        lsum += src[s * (HD + 2) + 1] * wgt;
        acc += src[s * (HD + 2) + 2 + f] * wgt;
    }
    const float r = lsum > 0.f ? acc * (1.f / lsum) : 0.f;
    out[(long long)head * HD + f] = f2bf(r);
}

struct Scratch {
    bf16_t *xa, *xb, *qkv, *ao, *rl, *scores, *act, *eo;
    int32_t *idx, *kv_len;
    float* part;  // split-KV attention states: H * DECODE_MAX_SPLITS * (hd + 2) floats
    // rl / scores / idx are kept PER LAYER (strides below, in elements): after a token the scratch holds every layer's router logits and
    // choice, which is what the full-depth parity case reads back (aria_decode_trace_layout; tests/fullwidth_cases.py::case_decode_full_depth)
    size_t rl_stride, sc_stride, idx_stride;
    size_t bytes;
};

// ---- sampling (gptfast/generate.py:35-58: logits_to_probs + multinomial_sample_one_no_sync) -------------------------------------------
// One token's sampling as ONE launch of one workgroup (1024 threads) instead of ~20 tiny tensor kernels (temperature, top-k, where,
// softmax, exponential, divide, arg-max, casts: 0.15 ms of a 2.16 ms token in the round-3 trace):
//   keep the logits >= the k-th largest (ties kept, as `logits < v[k-1] -> -inf` does), p_i ~ exp((l_i - max) / T), return argmax_i p_i / q_i
// with q_i ~ Exp(1) drawn by the caller (torch's generator: one `exponential_` launch, same seeds -> same stream as the tensor path).
// The k-th largest value is found EXACTLY by a two-pass radix select on the 16 bits of a bf16 (order-preserving key: flip all bits of a
// negative value, set the sign bit of a non-negative one): 256-bin histograms in LDS over the high byte, then over the low byte inside the
// boundary bin.  The softmax's normaliser does not move the arg-max and is skipped.  The logits (200 KB) and q (400 KB) stay in the L2.
constexpr int SAMPLE_THREADS = 1024;
__device__ __forceinline__ uint32_t bf16_order_key(bf16_t b) { return (b & 0x8000u) ? (~uint32_t(b) & 0xffffu) : (uint32_t(b) | 0x8000u); }

__global__ __launch_bounds__(SAMPLE_THREADS) void sample_topk_kernel(const bf16_t* logits, const float* q, int V, int k, float inv_temp,
                                                                     int32_t* out) {
    constexpr int NWV = SAMPLE_THREADS / 64;
    ARIA_SMEM_STATIC int hist[NWV][256];  // one histogram per wave (logits cluster in a handful of exponent bins: 16 waves adding into
                                          // ONE set of counters serialised on them), summed into hist[0] afterwards
    ARIA_SMEM_STATIC int sel[4];          // [0] boundary high byte, [1] rank still wanted inside it, [2] threshold key
    ARIA_SMEM_STATIC float red_v[NWV];
    ARIA_SMEM_STATIC int red_i[NWV];
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    const bool all = k >= V;
    const int nch = V >> 3;  // 16-byte chunks (8 logits); the tail (V % 8) goes element-wise
    float mx = -INFINITY;
    auto clear = [&]() {
        for (int i = t; i < NWV * 256; i += SAMPLE_THREADS) (&hist[0][0])[i] = 0;
        sync();
    };
    auto fold = [&]() {  // hist[0][b] = sum over the waves
        sync();
        if (t < 256) {
            int c = 0;
            for (int ww = 0; ww < NWV; ++ww) c += hist[ww][t];
            hist[0][t] = c;
        }
        sync();
    };
    // pass 1: histogram of the high byte (+ the maximum)
    clear();
    for (int c = t; c < nch; c += SAMPLE_THREADS) {
        const u32x4 v = ld16(logits + (long long)c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bf16_t b = bf16_t(e & 1 ? v[e >> 1] >> 16 : v[e >> 1] & 0xffffu);
            mx = fmaxf(mx, bf2f(b));
            if (!all) atomic_add(&hist[w][bf16_order_key(b) >> 8], 1);
        }
    }
    for (int i = nch * 8 + t; i < V; i += SAMPLE_THREADS) {
        const bf16_t b = logits[i];
        mx = fmaxf(mx, bf2f(b));
        if (!all) atomic_add(&hist[w][bf16_order_key(b) >> 8], 1);
    }
    mx = wave_max(mx);
    if (l == 0) red_v[w] = mx;
    fold();
    if (t == 0) {
        float m = red_v[0];
        for (int i = 1; i < NWV; ++i) m = fmaxf(m, red_v[i]);
        red_v[0] = m;
        int above = 0, b1 = 0;
        if (!all) {
            for (b1 = 255; b1 > 0; --b1) {
                if (above + hist[0][b1] >= k) break;
                above += hist[0][b1];
            }
        }
        sel[0] = b1;
        sel[1] = k - above;
    }
    sync();
    mx = red_v[0];
    const int b1 = sel[0], want = sel[1];
    sync();
    // pass 2: histogram of the low byte inside the boundary bin -> the exact 16-bit threshold
    if (!all) {
        clear();
        for (int c = t; c < nch; c += SAMPLE_THREADS) {
            const u32x4 v = ld16(logits + (long long)c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t key = bf16_order_key(bf16_t(e & 1 ? v[e >> 1] >> 16 : v[e >> 1] & 0xffffu));
                if (int(key >> 8) == b1) atomic_add(&hist[w][key & 255u], 1);
            }
        }
        for (int i = nch * 8 + t; i < V; i += SAMPLE_THREADS) {
            const uint32_t key = bf16_order_key(logits[i]);
            if (int(key >> 8) == b1) atomic_add(&hist[w][key & 255u], 1);
        }
        fold();
        if (t == 0) {
            int cum = 0, b2 = 255;
            for (; b2 > 0; --b2) {
                cum += hist[0][b2];
                if (cum >= want) break;
            }
            sel[2] = (b1 << 8) | b2;
        }
        sync();
    }
    const uint32_t thr = all ? 0u : uint32_t(sel[2]);
    // pass 3: arg-max of p_i / q_i over the kept logits (the lowest index wins ties)
    float best = -1.f;
    int besti = 0x7fffffff;
    auto consider = [&](bf16_t b, int i) {
        if (bf16_order_key(b) >= thr) {
            const float s = expf((bf2f(b) - mx) * inv_temp) / q[i];
            if (s > best || (s == best && i < besti)) {
                best = s;
                besti = i;
            }
        }
    };
    for (int c = t; c < nch; c += SAMPLE_THREADS) {
        const u32x4 v = ld16(logits + (long long)c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) consider(bf16_t(e & 1 ? v[e >> 1] >> 16 : v[e >> 1] & 0xffffu), c * 8 + e);
    }
    for (int i = nch * 8 + t; i < V; i += SAMPLE_THREADS) consider(logits[i], i);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ov = shfl_xor(best, d);
        const int oi = shfl_xor(besti, d);
        if (ov > best || (ov == best && oi < besti)) {
            best = ov;
            besti = oi;
        }
    }
    if (l == 0) {
        red_v[w] = best;
        red_i[w] = besti;
    }
    sync();
    if (t == 0) {
        for (int i = 1; i < NWV; ++i)
            if (red_v[i] > best || (red_v[i] == best && red_i[i] < besti)) {
                best = red_v[i];
                besti = red_i[i];
            }
        out[0] = besti;
    }
}

// splits == 1: the one-workgroup-per-head kernel; otherwise the split form + merge (part: H * splits * (hd + 2) floats)
int launch_decode_attn(void* stream, const bf16_t* qkv, const bf16_t* freqs, const int32_t* pos, bf16_t* kc, bf16_t* vc, bf16_t* out,
                       float* part, int64_t H, int64_t hd, int64_t D, int splits) {
    const float sc = 1.0f / sqrtf(float(hd));
    if (splits <= 1) {
        if (hd == 128)
            ARIA_LAUNCH((decode_attn_kernel<128>), dim3(unsigned(H)), dim3(DECODE_ATTN_WAVES * 64), 0, stream, qkv, freqs, pos, kc, vc, out, int(D), sc);
        else
            ARIA_LAUNCH((decode_attn_kernel<64>), dim3(unsigned(H)), dim3(DECODE_ATTN_WAVES * 64), 0, stream, qkv, freqs, pos, kc, vc, out, int(D), sc);
        return aria_check_launch();
    }
    if (hd == 128)
        ARIA_LAUNCH((decode_attn_split_kernel<128>), dim3(unsigned(H), unsigned(splits)), dim3(256), 0, stream, qkv, freqs, pos, kc, vc, part,
                    int(D), sc);
    else
        ARIA_LAUNCH((decode_attn_split_kernel<64>), dim3(unsigned(H), unsigned(splits)), dim3(256), 0, stream, qkv, freqs, pos, kc, vc, part,
                    int(D), sc);
    int rc = aria_check_launch();
    if (rc != ARIA_OK) return rc;
    ARIA_LAUNCH(decode_attn_merge_kernel, dim3(unsigned(H)), dim3(unsigned(hd)), 0, stream, (const float*)part, out, splits, int(hd));
    return aria_check_launch();
}

// ARIA_DECODE_SPLIT_KV: unset / "1" = one split per 1024 cache slots (2..32) for caches beyond 2048 slots; "0" = off (one workgroup
// per head; contexts > 16 K take the generic flash kernel); N > 1 = exactly N splits.  Measured on MI355X (profiles/r02_decode_split_kv.json):
// whole decode step 6.04 -> 3.19 ms at S_max 4096, 15.2 -> 3.70 ms at 16 K, 39.3 -> 4.37 ms at 32 K; the attention launch alone
// 384 -> 42 us at a 16 K fill.  Parity: tests/kernel_cases.py::case_decode_attention on hardware, 7 shapes.
int decode_splits_for(int64_t Smax) {
    const char* e = std::getenv("ARIA_DECODE_SPLIT_KV");  // read per call (a few ns against ~7 launches): tests flip it in-process
    const int mode = e ? atoi(e) : 1;
    if (mode <= 0 || Smax <= 2048) return 1;
    const int64_t n = mode == 1 ? (Smax + 1023) / 1024 : mode;
    return int(n < 2 ? 2 : n > DECODE_MAX_SPLITS ? DECODE_MAX_SPLITS : n);
}

// ARIA_DECODE_FUSE: unset / "1" = the 6-launch schedule (router + shared up | routed up | down + combine); "0" = the 7-launch one
// (tests compare the two bit for bit; tools/decode_bench.py times both)
bool decode_fuse_enabled() {
    const char* e = std::getenv("ARIA_DECODE_FUSE");
    return !e || atoi(e) != 0;
}

// NC = 16-byte chunks per lane = ceil(K / 512), a template parameter so that every load of a wave is issued up front
#define ARIA_NC_SWITCH(nc, CALL)   \
    switch (nc) {                  \
        case 1: CALL(1); break;    \
        case 2: CALL(2); break;    \
        case 3: CALL(3); break;    \
        case 4: CALL(4); break;    \
        case 5: CALL(5); break;    \
        case 6: CALL(6); break;    \
        case 7: CALL(7); break;    \
        default: CALL(8); break;   \
    }
inline int chunks_per_lane(long long K) { return int((K / 8 + 63) / 64); }

int launch_gemv(int N, void* stream, const bf16_t* W, long long ldw, const bf16_t* x, const bf16_t* norm_w, float eps, int K,
                const bf16_t* residual, bf16_t* y) {
    // rows per wave: 4 when that still leaves well over a thousand waves, else 2 (small N must still cover 256 CUs)
    if (N >= 4096) {
#define CALL(NC) ARIA_LAUNCH((gemv_kernel<4, NC>), dim3((N + 15) / 16), dim3(256), 0, stream, W, ldw, x, norm_w, eps, K, N, residual, y)
        ARIA_NC_SWITCH(chunks_per_lane(K), CALL)
#undef CALL
    } else {
#define CALL(NC) ARIA_LAUNCH((gemv_kernel<2, NC>), dim3((N + 7) / 8), dim3(256), 0, stream, W, ldw, x, norm_w, eps, K, N, residual, y)
        ARIA_NC_SWITCH(chunks_per_lane(K), CALL)
#undef CALL
    }
    return aria_check_launch();
}

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

Scratch carve(char* base, int64_t L, int64_t D, int64_t H, int64_t hd, int64_t E, int64_t k, int64_t I, int64_t Is) {
    Scratch s{};
    size_t off = 0;
    auto take = [&](size_t n) {
        char* p = base ? base + off : nullptr;
        off += align256(n);
        return p;
    };
    s.xa = reinterpret_cast<bf16_t*>(take(D * 2));
    s.xb = reinterpret_cast<bf16_t*>(take(D * 2));
    s.qkv = reinterpret_cast<bf16_t*>(take(3 * D * 2));
    s.ao = reinterpret_cast<bf16_t*>(take(D * 2));
    s.rl_stride = size_t(E + 63) & ~size_t(63), s.sc_stride = 64, s.idx_stride = 64;  // (k <= 8; 128- / 256-byte aligned rows)
    s.rl = reinterpret_cast<bf16_t*>(take(L * s.rl_stride * 2));
    s.scores = reinterpret_cast<bf16_t*>(take(L * s.sc_stride * 2));
    s.act = reinterpret_cast<bf16_t*>(take((k * I + Is) * 2));  // routed rows, then the shared expert's activation vector
    s.eo = reinterpret_cast<bf16_t*>(take((k + 1) * D * 2));     // routed outputs, then the shared expert's output
    s.idx = reinterpret_cast<int32_t*>(take(L * s.idx_stride * 4));
    s.kv_len = reinterpret_cast<int32_t*>(take(4));
    s.part = reinterpret_cast<float*>(take(size_t(H) * DECODE_MAX_SPLITS * size_t(hd + 2) * 4));
    s.bytes = off;
    return s;
}

}  // namespace

extern "C" {

int aria_decode_trace_layout(const int64_t* dims, int64_t* out) {
    if (!dims || !out) return ARIA_ERR_INVALID;
    char* const base = reinterpret_cast<char*>(uintptr_t(1) << 20);  // (never dereferenced: carve only does address arithmetic)
    const Scratch s = carve(base, dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], dims[6], dims[7]);
    out[0] = reinterpret_cast<char*>(s.rl) - base, out[1] = int64_t(s.rl_stride) * 2;      // router logits: bf16 [L][E], row stride in bytes
    out[2] = reinterpret_cast<char*>(s.idx) - base, out[3] = int64_t(s.idx_stride) * 4;    // expert ids: int32 [L][k]
    out[4] = reinterpret_cast<char*>(s.scores) - base, out[5] = int64_t(s.sc_stride) * 2;  // scores: bf16 [L][k]
    return ARIA_OK;
}

int64_t aria_decode_scratch_bytes(const int64_t* dims) {
    if (!dims) return 0;
    return int64_t(carve(nullptr, dims[0], dims[1], dims[2], dims[3], dims[4], dims[5], dims[6], dims[7]).bytes);
}

int aria_decode_token(const void* const* ptrs, const int64_t* dims, float eps, void* stream) {
    if (!ptrs || !dims) return ARIA_ERR_INVALID;
    const int64_t L = dims[0], D = dims[1], H = dims[2], hd = dims[3], E = dims[4], k = dims[5], I = dims[6], Is = dims[7], V = dims[8],
                  Smax = dims[9];
    if (L <= 0 || D <= 0 || H * hd != D || (D & 7) || (I & 7) || (Is & 7) || D > 4096 || I > 4096 || Is > 4096 || k > 8 || E > 256)  // 8 chunks/lane
        return ARIA_ERR_UNSUPPORTED;
    if (hd != 64 && hd != 128) return ARIA_ERR_UNSUPPORTED;
    if (Is % I) return ARIA_ERR_UNSUPPORTED;  // the shared expert is handled as Is / I further experts of width I
    const int ns = int(Is / I);
    for (int i = 0; i < ARIA_DECODE_HEADER_PTRS + ARIA_DECODE_LAYER_PTRS * L; ++i)
        if (!ptrs[i]) return ARIA_ERR_INVALID;
    const bf16_t* freqs = static_cast<const bf16_t*>(ptrs[0]);
    const bf16_t* final_norm = static_cast<const bf16_t*>(ptrs[1]);
    const bf16_t* out_w = static_cast<const bf16_t*>(ptrs[2]);
    const Scratch s = carve(static_cast<char*>(const_cast<void*>(ptrs[3])), L, D, H, hd, E, k, I, Is);
    const int32_t* pos = static_cast<const int32_t*>(ptrs[4]);
    const bf16_t* x = static_cast<const bf16_t*>(ptrs[5]);
    bf16_t* logits = static_cast<bf16_t*>(const_cast<void*>(ptrs[6]));
    int rc;
#define ARIA_TRY(call)              \
    do {                            \
        rc = (call);                \
        if (rc != ARIA_OK) return rc; \
    } while (0)
    const bool fuse = decode_fuse_enabled();
    for (int64_t li = 0; li < L; ++li) {
        const void* const* lp = ptrs + ARIA_DECODE_HEADER_PTRS + ARIA_DECODE_LAYER_PTRS * li;
        const bf16_t *attn_norm = static_cast<const bf16_t*>(lp[0]), *wqkv = static_cast<const bf16_t*>(lp[1]),
                     *wo = static_cast<const bf16_t*>(lp[2]), *ffn_norm = static_cast<const bf16_t*>(lp[3]),
                     *gate = static_cast<const bf16_t*>(lp[4]), *w1 = static_cast<const bf16_t*>(lp[5]),
                     *w3 = static_cast<const bf16_t*>(lp[6]), *w2 = static_cast<const bf16_t*>(lp[7]),
                     *sw1 = static_cast<const bf16_t*>(lp[8]), *sw3 = static_cast<const bf16_t*>(lp[9]),
                     *sw2 = static_cast<const bf16_t*>(lp[10]);
        bf16_t *kc = static_cast<bf16_t*>(const_cast<void*>(lp[11])), *vc = static_cast<bf16_t*>(const_cast<void*>(lp[12]));
        bf16_t* h = s.xa;  // hidden state after the attention block
        bf16_t *const rl = s.rl + li * s.rl_stride, *const scores = s.scores + li * s.sc_stride;  // this layer's routing record
        int32_t* const idx = s.idx + li * s.idx_stride;
        // attention block: h = x + wo( attn( rope(wqkv(norm(x))) ) )
        ARIA_TRY(launch_gemv(int(3 * D), stream, wqkv, (long long)D, x, attn_norm, eps, int(D), nullptr, s.qkv));
        const int splits = decode_splits_for(Smax);
        if (Smax <= 16384 || splits > 1) {  // one workgroup per head walks the whole context, or (opt-in) heads x splits workgroups
            ARIA_TRY(launch_decode_attn(stream, (const bf16_t*)s.qkv, freqs, pos, kc, vc, s.ao, s.part, H, hd, D, splits));
        } else {
            ARIA_LAUNCH(rope_cache_kernel, dim3(unsigned((3 * D / 8 + 255) / 256)), dim3(256), 0, stream, s.qkv, freqs, pos, kc, vc, int(D),
                        int(hd), s.kv_len);
            ARIA_TRY(aria_check_launch());
            ARIA_TRY(aria_attn_fwd(s.qkv, kc, vc, s.ao, nullptr, s.kv_len, nullptr, 1, 1, Smax, H, hd, D, D, D, D, 1.0f / sqrtf(float(hd)),
                                   0, stream));
        }
        ARIA_TRY(launch_gemv(int(D), stream, wo, (long long)D, s.ao, nullptr, 0.f, int(D), x, h));
        // MoE block on hn = norm(h): out = h + ( sum_j score_j * expert_j(hn) + shared(hn) )
        const int ncD = chunks_per_lane(D), ncI = chunks_per_lane(I), ncS = chunks_per_lane(Is);
        const bool fused_down = fuse && k <= DOWN_KMAX && ((ncI == 4 && ncS == 7) || (ncI == 1 && ncS == 1) || (ncI <= 2 && ncS <= 4));
        const int ns_up = fuse ? 0 : ns;  // shared-expert slices handled by expert_up_kernel (fused schedule: by router_shared_up_kernel)
        if (fuse) {  // three launches: router logits + shared up | routed up (top-k inside) | all down-projections + combine
            const int nrb = int((E + 7) / 8);
#define CALL(NC)                                                                                                                          \
    ARIA_LAUNCH((router_shared_up_kernel<2, NC>), dim3(unsigned(nrb + (Is + 7) / 8)), dim3(256), 0, stream, gate, int(E), rl, sw1, sw3, \
                int(Is), (const bf16_t*)h, ffn_norm, eps, int(D), s.act + k * I, nrb)
            ARIA_NC_SWITCH(ncD, CALL)
#undef CALL
        } else {  // four launches: router logits | routed + shared up (top-k inside) | down-projections | combine
            ARIA_TRY(launch_gemv(int(E), stream, gate, (long long)D, h, ffn_norm, eps, int(D), nullptr, rl));
        }
        // (top-k + softmax of the router run inside expert_up_kernel)
        if (I * (k + ns) >= 8192) {  // enough rows for 4 per wave (8 row reads of 5 KiB in flight per wave) and still > 2000 waves
#define CALL(NC)                                                                                                                       \
    ARIA_LAUNCH((expert_up_kernel<4, NC>), dim3(unsigned((I + 15) / 16), unsigned(k + ns_up)), dim3(256), 0, stream, w1, w3, sw1, sw3, \
                (const bf16_t*)rl, int(E), scores, idx, int(k), (const bf16_t*)h, ffn_norm, eps, int(D), int(I), s.act)
            ARIA_NC_SWITCH(ncD, CALL)
#undef CALL
        } else {
#define CALL(NC)                                                                                                                      \
    ARIA_LAUNCH((expert_up_kernel<2, NC>), dim3(unsigned((I + 7) / 8), unsigned(k + ns_up)), dim3(256), 0, stream, w1, w3, sw1, sw3, \
                (const bf16_t*)rl, int(E), scores, idx, int(k), (const bf16_t*)h, ffn_norm, eps, int(D), int(I), s.act)
            ARIA_NC_SWITCH(ncD, CALL)
#undef CALL
        }
        if (fused_down) {
#define DOWNC(NCI, NCS)                                                                                                                 \
    ARIA_LAUNCH((expert_down_combine_kernel<NCI, NCS>), dim3(unsigned((D + 3) / 4)), dim3(256), size_t(k * NCI + NCS) * 1024, stream, w2, \
                sw2, (const int32_t*)idx, (const bf16_t*)scores, int(k), ns, (const bf16_t*)s.act, int(I), int(D), (const bf16_t*)h, s.xb)
            if (ncI == 4 && ncS == 7) {  // Aria: I = 1664, shared 3328
                DOWNC(4, 7);
            } else if (ncI == 1 && ncS == 1) {
                DOWNC(1, 1);
            } else {
                DOWNC(2, 4);
            }
#undef DOWNC
        } else {
#define DOWN(NCI, NCS)                                                                                                               \
    ARIA_LAUNCH((expert_down_kernel<2, NCI, NCS>), dim3(unsigned((D + 7) / 8), unsigned(k + 1)), dim3(256), 0, stream, w2, sw2, \
                (const int32_t*)idx, int(k), ns, (const bf16_t*)s.act, int(I), int(D), s.eo)
            if (ncI == 4 && ncS == 7) {  // Aria: I = 1664, shared 3328
                DOWN(4, 7);
            } else if (ncI == 1 && ncS == 1) {
                DOWN(1, 1);
            } else if (ncI <= 2 && ncS <= 4) {
                DOWN(2, 4);
            } else {
                DOWN(8, 8);  // any other width: correct (chunks past the end read as zeros), more load instructions than needed
            }
#undef DOWN
            ARIA_LAUNCH(combine_kernel, dim3(unsigned((D / 8 + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)s.eo, (const bf16_t*)scores,
                        int(k), (const bf16_t*)(s.eo + k * D), (const bf16_t*)h, s.xb, int(D));
        }
        ARIA_TRY(aria_check_launch());
        x = s.xb;  // the next layer reads x = xb and writes its h into xa again (h is dead once this add has run)
    }
    ARIA_TRY(launch_gemv(int(V), stream, out_w, (long long)D, x, final_norm, eps, int(D), nullptr, logits));
#undef ARIA_TRY
    return ARIA_OK;
}

int aria_sample_topk(const void* logits, const float* q, int64_t V, int64_t top_k, float temperature, int32_t* out, void* stream) {
    if (!logits || !q || !out || V <= 0) return ARIA_ERR_INVALID;
    if (V >= (1ll << 31)) return ARIA_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(logits) & 15) return ARIA_ERR_ALIGN;  // 16-byte loads
    const float t = temperature > 1e-5f ? temperature : 1e-5f;  // (gptfast/generate.py:47: max(temperature, 1e-5))
    const int k = (top_k <= 0 || top_k >= V) ? int(V) : int(top_k);
    ARIA_LAUNCH(sample_topk_kernel, dim3(1), dim3(SAMPLE_THREADS), 0, stream, static_cast<const bf16_t*>(logits), q, int(V), k, 1.0f / t, out);
    return aria_check_launch();
}

int aria_decode_route(const void* logits, int64_t E, int64_t k, void* scores, int32_t* idx, void* stream) {
    if (!logits || !scores || !idx || E <= 0 || k <= 0) return ARIA_ERR_INVALID;
    if (E > 256 || k > 8 || k > E) return ARIA_ERR_UNSUPPORTED;
    ARIA_LAUNCH(router_topk_kernel, dim3(1), dim3(64), 0, stream, static_cast<const bf16_t*>(logits), int(E), int(k), static_cast<bf16_t*>(scores),
                idx);
    return aria_check_launch();
}

int64_t aria_decode_attn_workspace_bytes(int64_t H, int64_t hd, int64_t splits) {
    if (H <= 0 || hd <= 0 || splits <= 1) return 0;
    return H * splits * (hd + 2) * 4;
}

int aria_decode_attn(const void* qkv, const void* freqs_cis, const int32_t* pos, void* k_cache, void* v_cache, void* out, int64_t H, int64_t hd,
                     int64_t splits, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!qkv || !freqs_cis || !pos || !k_cache || !v_cache || !out || H <= 0) return ARIA_ERR_INVALID;
    if (hd != 64 && hd != 128) return ARIA_ERR_UNSUPPORTED;
    if (splits < 1 || splits > DECODE_MAX_SPLITS) return ARIA_ERR_INVALID;
    if (splits > 1 && (!workspace || workspace_bytes < aria_decode_attn_workspace_bytes(H, hd, splits))) return ARIA_ERR_INVALID;
    return launch_decode_attn(stream, static_cast<const bf16_t*>(qkv), static_cast<const bf16_t*>(freqs_cis), pos,
                              static_cast<bf16_t*>(k_cache), static_cast<bf16_t*>(v_cache), static_cast<bf16_t*>(out),
                              static_cast<float*>(workspace), H, hd, H * hd, int(splits));
}

// The enqueue sequence of aria_decode_token reads the position (and through it the KV-cache slot and kv_len) from DEVICE memory, so it
// is the same for every token: capture it once into a HIP graph and replay it with one launch per token (the ~400 tiny launches of a
// token cost more host time than their kernels run).  Captured on a private stream; replayed on the caller's stream.
struct AriaDecodeGraph {
#ifndef ARIA_EMU
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
#else
    const void* const* ptrs = nullptr;
    const int64_t* dims = nullptr;
    float eps = 0.f;
#endif
};

void* aria_decode_graph_create(const void* const* ptrs, const int64_t* dims, float eps) {
    AriaDecodeGraph* g = new AriaDecodeGraph();
#ifndef ARIA_EMU
    hipStream_t cap = nullptr;
    if (hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) != hipSuccess) {
        delete g;
        return nullptr;
    }
    bool ok = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
        const int rc = aria_decode_token(ptrs, dims, eps, cap);
        const hipError_t e = hipStreamEndCapture(cap, &g->graph);
        ok = rc == ARIA_OK && e == hipSuccess && g->graph;
    }
    if (ok) ok = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0) == hipSuccess;
    (void)hipStreamDestroy(cap);
    if (!ok) {
        if (g->graph) (void)hipGraphDestroy(g->graph);
        (void)hipGetLastError();
        delete g;
        return nullptr;
    }
#else
    g->ptrs = ptrs;  // the emulator has no graphs: replay = enqueue again (the caller keeps the tables alive)
    g->dims = dims;
    g->eps = eps;
#endif
    return g;
}

int aria_decode_graph_launch(void* handle, void* stream) {
    if (!handle) return ARIA_ERR_INVALID;
    AriaDecodeGraph* g = static_cast<AriaDecodeGraph*>(handle);
#ifndef ARIA_EMU
    return hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream)) == hipSuccess ? ARIA_OK : ARIA_ERR_LAUNCH;
#else
    return aria_decode_token(g->ptrs, g->dims, g->eps, stream);
#endif
}

void aria_decode_graph_destroy(void* handle) {
    if (!handle) return;
    AriaDecodeGraph* g = static_cast<AriaDecodeGraph*>(handle);
#ifndef ARIA_EMU
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
#endif
    delete g;
}

}  // extern "C"
