// MoE routing / token dispatch kernels (HBM-bound integer + byte work: coalesced 16-byte rows, wave
// ballots for the stable sort, no host round trips).  Reference: aria/model/moe_lm.py:243-365, 505-507.
#include "aria_device.h"
#include "aria_hip.h"
#include <cstdlib>
#include <algorithm>

namespace {
using namespace ad;

constexpr int kMaxPerLane = 4;  // E <= 256

// ------------------------------------------------------------------------------------------- K1: router GEMM + routing, one launch
// TopKRouter.forward (moe_lm.py:190-201, 243-293): logits = x W^T ([T, D] x [E, D]^T, E = 32 NB), then per token top-k with ties to the
// lowest expert id, softmax over the selected logits in fp32, scores cast to bf16, and the tokens-per-expert histogram -- the logits GEMM
// (a 75 %-empty 256-wide tile of the general kernel), a memset and route_kernel as ONE launch.  One wave = 32 tokens:
//   * logits on the matrix pipe, 32 tokens x 32 experts per wave; accumulation order = the general kernels' (k ascending in blocks of 16 on
//     v_mfma_f32_32x32x16_bf16): the bf16 logits are bit-identical to aria_gemm_bf16's, so is everything derived from them;
//   * the rounded logits go to the wave's LDS tile [32 tokens][E] and from there to HBM (the backward wants them) in 16-byte pieces;
//   * routing per token by route_one_token (aria_device.h: the decode engine's; lane = expert) on the LDS row; histogram per wave in LDS,
//     one global atomic per (wave, expert with a count).
// One wave per 32-expert block (NB waves per workgroup, the block's 32 tokens' routing dealt to them), operands STAGED THROUGH THE LDS.  An MFMA
// fragment is 16 bytes of 32 different rows: read straight from global memory (round 5's first form) that is 32 cache lines per wave-instruction
// through the vector L1 -- 102 us per launch with one wave per token block, 74 us with two and three chunks of prefetch, against ~20 us of
// traffic.  Here a wave-instruction is an
// LDS-DMA piece of 8 whole 128-byte row chunks (global_load_lds_dwordx4: lane -> row 8 j + (l >> 3), 16-byte piece (l & 7) ^ (row & 7) -- the XOR
// spreads a fragment read's 8 rows over the banks), four stages deep with counted waits, and the fragments come back by ds_read_b128.  Per chunk of
// 64 reduction indices: the block's x tile (32 rows, shared by its waves: each wave brings half of it) + one 32-row weight tile per wave.
// 54.9 us at 16 384 x 2560 -> 64 (same box: the direct form 73.9, gemm + route 84.2).  Every logit is ONE accumulator run over the whole reduction
// in gemm's order: bit-identical to gemm + route.
template <int NB>
__global__ __launch_bounds__(64 * NB) void router_fused_kernel(const bf16_t* x, const bf16_t* w, bf16_t* logits, bf16_t* scores, int32_t* indices,
                                                                   int32_t* counts, int T, int D, int k, long long ldx) {
    constexpr int E = 32 * NB, CH = 4, ST = 4, XT = 4096, WT = 4096, STAGE = XT + NB * WT;
    constexpr int XP = 4 / NB;              // x pieces per wave and chunk (the tile's four 8-row pieces dealt to the waves)
    constexpr int PER = XP + 4;             // DMA pieces per wave and chunk
    ARIA_SMEM_STATIC char stage[ST * STAGE];
    ARIA_SMEM_STATIC bf16_t tile[32 * E];
    ARIA_SMEM_STATIC int hist[64];
    const int l = threadIdx.x & 63, lr = l & 31, kh = l >> 5, b = first_lane(int(threadIdx.x) >> 6);
    const int t0 = blockIdx.x * 32;
    if (b == 0) hist[l] = 0;
    const int nch = D / 64;
    // this lane's part of a piece: row (l >> 3) of the piece's 8, source piece (l & 7) ^ (row & 7) (row & 7 == l >> 3)
    const int prow = l >> 3, psrc = ((l & 7) ^ prow) * 8;
    const bf16_t* xsrc[XP];
#pragma unroll
    for (int j = 0; j < XP; ++j) xsrc[j] = x + (long long)min(t0 + 8 * (b * XP + j) + prow, T - 1) * ldx + psrc;   // (rows past T: clamped, never stored)
    const bf16_t* wsrc = w + (long long)(32 * b + prow) * D + psrc;
    auto issue = [&](int c) __attribute__((always_inline)) {   // chunk c -> stage c % ST (past the end: the last chunk again, into a free stage)
        const int cc = min(c, nch - 1);
        char* st = stage + (c % ST) * STAGE;
#pragma unroll
        for (int j = 0; j < XP; ++j) glds16_raw(xsrc[j] + cc * 64, st + (b * XP + j) * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_raw(wsrc + (long long)(8 * j) * D + cc * 64, st + XT + b * WT + j * 1024);
    };
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
    for (int c = 0; c < ST - 1; ++c) issue(c);
    const int fo = lr * 128;   // fragment row offset; piece q of row lr sits in slot q ^ (lr & 7)
    for (int c = 0; c < nch; ++c) {
        wait_vm<(ST - 2) * PER>();   // this wave's pieces of chunk c have landed (the two newer chunks stay in flight)
        sync();                      // ... and every other wave's; the readers of chunk c - 1 are done with its stage
        issue(c + ST - 1);           // -> stage (c - 1) % ST
        const char* st = stage + (c % ST) * STAGE;
        s16x8 xa[CH], wb[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int slot = ((2 * i + kh) ^ (lr & 7)) * 16;
            xa[i] = *reinterpret_cast<const s16x8*>(st + fo + slot);
            wb[i] = *reinterpret_cast<const s16x8*>(st + XT + b * WT + fo + slot);
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) acc = mfma32(xa[i], wb[i], acc);
    }
    wait_vm<0>();   // (the duplicate pieces of the last steps: nothing may still be writing the LDS when the block ends)
    // C layout of the 32 x 32 tile: register j of lane l = (token (j & 3) + 8 (j >> 2) + 4 (l >> 5), expert l & 31)
#pragma unroll
    for (int j = 0; j < 16; ++j) tile[((j & 3) + 8 * (j >> 2) + 4 * kh) * E + 32 * b + lr] = f2bf(acc[j]);
    sync();
#pragma unroll
    for (int it = 0; it < (32 * E) / (64 * NB * 8); ++it) {
        const int ci = it * 64 * NB + int(threadIdx.x), tr = ci / (E / 8);
        if (t0 + tr < T) st16(logits + (long long)t0 * E + ci * 8, ld16(tile + ci * 8));
    }
    const int nt = min(32, T - t0);
    for (int t = b; t < nt; t += NB) {
        float sc;
        int id;
        route_one_token(tile + t * E, E, k, l, 0, sc, id);
        if (l < k) {
            scores[(long long)(t0 + t) * k + l] = f2bf(sc);
            indices[(long long)(t0 + t) * k + l] = id;
            atomic_add(&hist[id], 1);
        }
    }
    sync();
    if (b == 0 && l < E && hist[l]) atomic_add(&counts[l], hist[l]);
}

// ------------------------------------------------------------------------------------------- route
// one wave per token; lane i owns experts i, i+64, ...
template <bool F32>
__global__ __launch_bounds__(256) void route_kernel(const void* logits_, void* scores_, int32_t* indices, int32_t* counts,
                                                    int T, int E, int k) {
    // tokens-per-expert histogram: T*k atomics on E global counters serialise in L2 (measured 266 us for 16384 x 6 on 64 counters);
    // count in LDS per block, then one global atomic per (block, expert)
    ARIA_SMEM_STATIC int hist[64 * kMaxPerLane];
    for (int i = threadIdx.x; i < E; i += blockDim.x) hist[i] = 0;
    sync();
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int npl = (E + 63) >> 6;
    for (int t = wave; t < T; t += nwaves) {
        float v[kMaxPerLane];
#pragma unroll
        for (int i = 0; i < kMaxPerLane; ++i) {
            const int e = l + 64 * i;
            v[i] = -INFINITY;
            if (i < npl && e < E)
                v[i] = F32 ? static_cast<const float*>(logits_)[(long long)t * E + e]
                           : bf2f(static_cast<const bf16_t*>(logits_)[(long long)t * E + e]);
        }
        float top[8];
        int topi = -1;  // lane j keeps the j-th selected index
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            top[j] = -INFINITY;
            if (j < k) {  // wave-uniform
                float bv = v[0];
                int bi = l;
#pragma unroll
                for (int i = 1; i < kMaxPerLane; ++i)
                    if (v[i] > bv) {  // strict: lower expert id wins ties
                        bv = v[i];
                        bi = l + 64 * i;
                    }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const float ov = shfl_xor(bv, d);
                    const int oi = shfl_xor(bi, d);
                    if (ov > bv || (ov == bv && oi < bi)) {
                        bv = ov;
                        bi = oi;
                    }
                }
                top[j] = bv;
                if (l == j) topi = bi;
                if ((bi & 63) == l) {
#pragma unroll
                    for (int i = 0; i < kMaxPerLane; ++i)
                        if (i == (bi >> 6)) v[i] = -INFINITY;
                }
            }
        }
        // softmax over the k selected logits in fp32, cast to the logits dtype (moe_lm.py:262)
        const float m = top[0];
        float den = 0.f, mine = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < k) den += expf(top[j] - m);
            if (j == l) mine = top[j];
        }
        if (l < k) {
            const float s = expf(mine - m) / den;
            if (F32)
                static_cast<float*>(scores_)[(long long)t * k + l] = s;
            else
                static_cast<bf16_t*>(scores_)[(long long)t * k + l] = f2bf(s);
            indices[(long long)t * k + l] = topi;
            atomic_add(&hist[topi], 1);
        }
    }
    sync();
    for (int i = threadIdx.x; i < E; i += blockDim.x)
        if (hist[i]) atomic_add(&counts[i], hist[i]);
}

// ------------------------------------------------------------------------------------------- stable sort (E <= 64)
constexpr int SORT_CHUNK = 2048;

__device__ __forceinline__ unsigned long long match_mask(int key, const unsigned long long (&bal)[6], unsigned long long valid) {
    unsigned long long m = valid;
#pragma unroll
    for (int b = 0; b < 6; ++b) m &= ((key >> b) & 1) ? bal[b] : ~bal[b];
    return m;
}

// pass 1: per-chunk histogram (lane e = expert e)
__global__ __launch_bounds__(64) void sort_count_kernel(const int32_t* idx, int32_t* chunk_counts, int M) {
    const int l = threadIdx.x, c = blockIdx.x;
    int cnt = 0;
    const int begin = c * SORT_CHUNK, end = min(M, begin + SORT_CHUNK);
    for (int i0 = begin; i0 < end; i0 += 64) {
        const int i = i0 + l;
        const bool valid = i < end;
        const int e = valid ? idx[i] : 0;
        unsigned long long bal[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) bal[b] = ballot((e >> b) & 1);
        const unsigned long long vm = ballot(valid);
        cnt += __builtin_popcountll(match_mask(l, bal, vm));
    }
    chunk_counts[c * 64 + l] = cnt;
}

// pass 2: offsets = exclusive scan of counts; chunk_base[c][e] = offsets[e] + sum_{c' < c} chunk_counts[c'][e]
__global__ __launch_bounds__(64) void sort_scan_kernel(const int32_t* counts, const int32_t* chunk_counts, int32_t* chunk_base,
                                                       int32_t* offsets, int E, int nchunks) {
    const int l = threadIdx.x;
    const int cnt = l < E ? counts[l] : 0;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = shfl(incl, (l - d) & 63);
        if (l >= d) incl += v;
    }
    int running = incl - cnt;
    if (l < E) offsets[l] = running;
    if (l == E - 1) offsets[E] = incl;
    for (int c = 0; c < nchunks; ++c) {
        chunk_base[c * 64 + l] = running;
        running += chunk_counts[c * 64 + l];
    }
}

// pass 3: stable positions
__global__ __launch_bounds__(64) void sort_scatter_kernel(const int32_t* idx, const int32_t* chunk_base, int32_t* sorted_src,
                                                          int32_t* inv, int M) {
    const int l = threadIdx.x, c = blockIdx.x;
    int counter = chunk_base[c * 64 + l];  // lane e: next free position of expert e
    const int begin = c * SORT_CHUNK, end = min(M, begin + SORT_CHUNK);
    const unsigned long long below = (1ull << l) - 1ull;
    for (int i0 = begin; i0 < end; i0 += 64) {
        const int i = i0 + l;
        const bool valid = i < end;
        const int e = valid ? idx[i] : 0;
        unsigned long long bal[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) bal[b] = ballot((e >> b) & 1);
        const unsigned long long vm = ballot(valid);
        const unsigned long long same = match_mask(e, bal, vm);
        const int base = shfl(counter, e & 63);
        if (valid) {
            const int pos = base + __builtin_popcountll(same & below);
            sorted_src[pos] = i;
            inv[i] = pos;
        }
        counter += __builtin_popcountll(match_mask(l, bal, vm));
    }
}

// ------------------------------------------------------------------------------------------- permute (row gather)
// one wave per destination row, 16-byte chunks, grid-stride
__global__ __launch_bounds__(256) void permute_kernel(const bf16_t* x, const int32_t* sorted_src, bf16_t* out, int M, int D,
                                                      int k, long long ldx) {
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nch = D >> 3;
    for (int p = wave; p < M; p += nwaves) {
        const int t = sorted_src[p] / k;
        const bf16_t* src = x + (long long)t * ldx;
        bf16_t* dst = out + (long long)p * D;
        for (int c = l; c < nch; c += 64) st16(dst + c * 8, ld16(src + c * 8));
    }
}

// Row width known at compile time (D = 512 NC: Aria's 2560 is NC = 5): every 16-byte load of a row is issued before the first store, and
// the NEXT row's index is fetched while the current row moves -- the generic kernel above has one load per lane in flight (index load,
// then chunk after chunk), ~32 KB per CU against the ~64 KB that 8 TB/s x ~2 us of latency need (r04; it ran at 3.5-3.8 TB/s).
template <int NC>
__global__ __launch_bounds__(256) void permute_rows_kernel(const bf16_t* x, const int32_t* sorted_src, bf16_t* out, int M, int k, long long ldx) {
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    constexpr int D = NC * 512;
    int p = wave;
    int src = p < M ? sorted_src[p] : 0;
    while (p < M) {
        const int pn = p + nwaves;
        const int src_n = pn < M ? sorted_src[pn] : 0;
        const bf16_t* row = x + (long long)(src / k) * ldx;
        u32x4 v[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) v[i] = ld16(row + (l + 64 * i) * 8);
        bf16_t* dst = out + (long long)p * D;
#pragma unroll
        for (int i = 0; i < NC; ++i) st16(dst + (l + 64 * i) * 8, v[i]);
        p = pn;
        src = src_n;
    }
}

// ------------------------------------------------------------------------------------------- unpermute (+ shared add)
// (compile-time row width, see permute_rows_kernel: the k rows of a token x NC chunks + the shared expert's chunks are all in flight together;
// same arithmetic in the same order as unpermute_kernel -- bit-identical)
template <int NC, int K>
__global__ __launch_bounds__(256) void unpermute_rows_kernel(const bf16_t* eo, const int32_t* inv, const bf16_t* scores, const bf16_t* add,
                                                             const bf16_t* res, bf16_t* out, int T) {
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    constexpr int D = NC * 512;
    for (int t = wave; t < T; t += nwaves) {
        int rows[K];
        float sc[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            rows[j] = inv[(long long)t * K + j];
            sc[j] = scores ? bf2f(scores[(long long)t * K + j]) : 1.f;
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = l + 64 * i;
            u32x4 v[K], a = zero16();
#pragma unroll
            for (int j = 0; j < K; ++j) v[j] = ld16(eo + (long long)rows[j] * D + c * 8);
            if (add) a = ld16(add + (long long)t * D + c * 8);
            float acc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (scores) {  // product materialised in bf16 by the reference (moe_lm.py:362)
                        acc[2 * q] += rbf(bflo(v[j][q]) * sc[j]);
                        acc[2 * q + 1] += rbf(bfhi(v[j][q]) * sc[j]);
                    } else {
                        acc[2 * q] += bflo(v[j][q]);
                        acc[2 * q + 1] += bfhi(v[j][q]);
                    }
                }
            u32x4 o;
            if (add) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = pack2bf(rbf(acc[2 * q]) + bflo(a[q]), rbf(acc[2 * q + 1]) + bfhi(a[q]));
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = pack2bf(acc[2 * q], acc[2 * q + 1]);
            }
            if (res) {   // r05b: the decoder layer's residual add `h + moe(h)` (moe_lm.py:617-627 through LlamaDecoderLayer) as a second rounding step
                const u32x4 rv = ld16(res + (long long)t * D + c * 8);   // here instead of a launch of its own: bf16(h + bf16(moe)), as add_kernel gives
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = pack2bf(bflo(rv[q]) + bflo(o[q]), bfhi(rv[q]) + bfhi(o[q]));
            }
            st16(out + (long long)t * D + c * 8, o);
        }
    }
}

// backward of unpermute with a compile-time row width: dout's chunks are fetched ONCE per token (the generic kernel re-reads them for each of
// the k rows), a row's NC chunks are in flight together, and the dot product's wave sum runs on the DPP ladder instead of six ds_bpermute
// round trips per row.  d_eo is bit-identical to the generic kernel's; dscores differs from it in the order of the 64-lane sum only (the
// per-lane partial sums are formed in the same order), within the rounding of the bf16 it is stored in.
template <int NC, int K>
__global__ __launch_bounds__(256) void unpermute_bwd_rows_kernel(const bf16_t* dout, const bf16_t* eo, const int32_t* inv, const bf16_t* scores,
                                                                 bf16_t* d_eo, bf16_t* dscores, int T) {
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    constexpr int D = NC * 512;
    for (int t = wave; t < T; t += nwaves) {
        u32x4 g[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) g[i] = ld16(dout + (long long)t * D + (l + 64 * i) * 8);
        int rows[K];
        float sc[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            rows[j] = inv[(long long)t * K + j];
            sc[j] = bf2f(scores[(long long)t * K + j]);
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            u32x4 v[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) v[i] = ld16(eo + (long long)rows[j] * D + (l + 64 * i) * 8);
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dot += bflo(g[i][q]) * bflo(v[i][q]) + bfhi(g[i][q]) * bfhi(v[i][q]);
                    o[q] = pack2bf(bflo(g[i][q]) * sc[j], bfhi(g[i][q]) * sc[j]);
                }
                st16(d_eo + (long long)rows[j] * D + (l + 64 * i) * 8, o);
            }
            dot = wave_sum_bcast(dot);
            if (l == 0) dscores[(long long)t * K + j] = f2bf(dot);
        }
    }
}

template <int K_MAX>
__global__ __launch_bounds__(256) void unpermute_kernel(const bf16_t* eo, const int32_t* inv, const bf16_t* scores,
                                                        const bf16_t* add, bf16_t* out, int T, int D, int k) {
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nch = D >> 3;
    for (int t = wave; t < T; t += nwaves) {
        int rows[K_MAX];
        float sc[K_MAX];
#pragma unroll
        for (int j = 0; j < K_MAX; ++j) {
            rows[j] = j < k ? inv[(long long)t * k + j] : 0;
            sc[j] = (j < k && scores) ? bf2f(scores[(long long)t * k + j]) : 1.f;
        }
        for (int c = l; c < nch; c += 64) {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
            for (int j = 0; j < K_MAX; ++j) {
                if (j < k) {
                    const u32x4 v = ld16(eo + (long long)rows[j] * D + c * 8);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (scores) {  // product materialised in bf16 by the reference (moe_lm.py:362)
                            acc[2 * q] += rbf(bflo(v[q]) * sc[j]);
                            acc[2 * q + 1] += rbf(bfhi(v[q]) * sc[j]);
                        } else {
                            acc[2 * q] += bflo(v[q]);
                            acc[2 * q + 1] += bfhi(v[q]);
                        }
                    }
                }
            }
            u32x4 o;
            if (add) {
                const u32x4 a = ld16(add + (long long)t * D + c * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = pack2bf(rbf(acc[2 * q]) + bflo(a[q]), rbf(acc[2 * q + 1]) + bfhi(a[q]));
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = pack2bf(acc[2 * q], acc[2 * q + 1]);
            }
            st16(out + (long long)t * D + c * 8, o);
        }
    }
}

// backward of unpermute: d_eo[inv[t,j]] = bf16(dout[t] * s_j); dscores[t,j] = <eo[inv[t,j]], dout[t]>
__global__ __launch_bounds__(256) void unpermute_bwd_kernel(const bf16_t* dout, const bf16_t* eo, const int32_t* inv,
                                                            const bf16_t* scores, bf16_t* d_eo, bf16_t* dscores, int T, int D,
                                                            int k) {
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nch = D >> 3;
    for (int t = wave; t < T; t += nwaves) {
        for (int j = 0; j < k; ++j) {
            const int row = inv[(long long)t * k + j];
            const float s = bf2f(scores[(long long)t * k + j]);
            float dot = 0.f;
            for (int c = l; c < nch; c += 64) {
                const u32x4 g = ld16(dout + (long long)t * D + c * 8);
                const u32x4 v = ld16(eo + (long long)row * D + c * 8);
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    dot += bflo(g[q]) * bflo(v[q]) + bfhi(g[q]) * bfhi(v[q]);
                    o[q] = pack2bf(bflo(g[q]) * s, bfhi(g[q]) * s);
                }
                st16(d_eo + (long long)row * D + c * 8, o);
            }
            dot = wave_sum(dot);
            if (l == 0) dscores[(long long)t * k + j] = f2bf(dot);
        }
    }
}

// ------------------------------------------------------------------------------------------- route backward (+ aux losses)
__global__ __launch_bounds__(256) void route_bwd_kernel(const bf16_t* logits, const int32_t* indices, const bf16_t* scores,
                                                        const bf16_t* dscores, const int32_t* counts, bf16_t* dlogits, int T,
                                                        int E, int k, float z_coeff, float aux_coeff, float aux_scale) {
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const bool aux = (z_coeff != 0.f || aux_coeff != 0.f);
    for (int t = wave; t < T; t += nwaves) {
        float x[kMaxPerLane], g[kMaxPerLane], cnt[kMaxPerLane];
#pragma unroll
        for (int i = 0; i < kMaxPerLane; ++i) {
            const int e = l + 64 * i;
            x[i] = e < E ? bf2f(logits[(long long)t * E + e]) : -INFINITY;
            cnt[i] = (aux && e < E) ? float(counts[e]) : 0.f;
            g[i] = 0.f;
        }
        // d softmax over the selected k
        float sdot = 0.f;
        for (int j = 0; j < k; ++j) sdot += bf2f(scores[(long long)t * k + j]) * bf2f(dscores[(long long)t * k + j]);
        for (int j = 0; j < k; ++j) {
            const int e = indices[(long long)t * k + j];
            const float s = bf2f(scores[(long long)t * k + j]);
            const float v = s * (bf2f(dscores[(long long)t * k + j]) - sdot);
#pragma unroll
            for (int i = 0; i < kMaxPerLane; ++i)
                if (e == l + 64 * i) g[i] += v;
        }
        if (aux) {
            float m = x[0];
#pragma unroll
            for (int i = 1; i < kMaxPerLane; ++i) m = fmaxf(m, x[i]);
            m = wave_max(m);
            float se = 0.f;
#pragma unroll
            for (int i = 0; i < kMaxPerLane; ++i) se += (l + 64 * i < E) ? expf(x[i] - m) : 0.f;
            se = wave_sum(se);
            const float lse = m + logf(se);
            float pc = 0.f, prob[kMaxPerLane];
#pragma unroll
            for (int i = 0; i < kMaxPerLane; ++i) {
                prob[i] = (l + 64 * i < E) ? expf(x[i] - lse) : 0.f;
                pc += prob[i] * cnt[i];
            }
            pc = wave_sum(pc);
            const float zc = aux_scale * z_coeff * 2.f * lse / float(T);
            const float ac = aux_scale * aux_coeff * float(E) / (float(T) * float(k)) / float(T);
#pragma unroll
            for (int i = 0; i < kMaxPerLane; ++i) g[i] += zc * prob[i] + ac * prob[i] * (cnt[i] - pc);
        }
#pragma unroll
        for (int i = 0; i < kMaxPerLane; ++i) {
            const int e = l + 64 * i;
            if (e < E) dlogits[(long long)t * E + e] = f2bf(g[i]);
        }
    }
}

// ------------------------------------------------------------------------------------------- swiglu
__device__ __forceinline__ float silu_f(float a) { return silu_fast(a); }

__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* h, const bf16_t* h2, bf16_t* act, long long nchunks,
                                                         int I, long long lda, long long ldb) {
    const int cpr = I >> 3;  // chunks per row
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (long long)gridDim.x * blockDim.x) {
        const long long row = c / cpr;
        const int col = int(c % cpr) * 8;
        const u32x4 a = ld16(h + row * lda + col);
        const u32x4 b = ld16(h2 + row * ldb + col);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            o[q] = pack2bf(rbf(silu_f(bflo(a[q]))) * bflo(b[q]), rbf(silu_f(bfhi(a[q]))) * bfhi(b[q]));
        st16(act + row * I + col, o);
    }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* h, const bf16_t* h2, const bf16_t* dact, bf16_t* dh,
                                                         bf16_t* dh2, long long nchunks, int I, long long lda, long long ldb) {
    const int cpr = I >> 3;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (long long)gridDim.x * blockDim.x) {
        const long long row = c / cpr;
        const int col = int(c % cpr) * 8;
        const u32x4 a = ld16(h + row * lda + col);
        const u32x4 b = ld16(h2 + row * ldb + col);
        const u32x4 g = ld16(dact + row * I + col);
        u32x4 oa, ob;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x2 da, db;
            swiglu_bwd_pair(f32x2{bflo(a[q]), bfhi(a[q])}, f32x2{bflo(b[q]), bfhi(b[q])}, f32x2{bflo(g[q]), bfhi(g[q])}, da, db);
            oa[q] = pack2bf(da.x, da.y);
            ob[q] = pack2bf(db.x, db.y);
        }
        st16(dh + row * lda + col, oa);
        st16(dh2 + row * ldb + col, ob);
    }
}

// ------------------------------------------------------------------------------------------- embedding backward
// dW[ids[t], :] += dy[t, :]  (bf16 pairs updated with a 32-bit CAS loop: order-independent up to bf16 rounding)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const bf16_t* dy, const int32_t* ids, bf16_t* dw, int T, int D) {
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nw = D >> 1;
    for (int t = wave; t < T; t += nwaves) {
        const int row = ids[t];
        if (row < 0) continue;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(dy + (long long)t * D);
        uint32_t* dst = reinterpret_cast<uint32_t*>(dw + (long long)row * D);
        for (int c = l; c < nw; c += 64) {
            const uint32_t g = src[c];
#ifdef ARIA_EMU
            dst[c] = pack2bf(bflo(dst[c]) + bflo(g), bfhi(dst[c]) + bfhi(g));
#else
            uint32_t old = dst[c], assumed;
            do {
                assumed = old;
                old = atomicCAS(dst + c, assumed, pack2bf(bflo(assumed) + bflo(g), bfhi(assumed) + bfhi(g)));
            } while (old != assumed);
#endif
        }
    }
}

// ARIA_MOE_GENERIC_DISPATCH=1: the generic-width permute / unpermute kernels everywhere (A/B measurements, bit-identity tests)
bool generic_dispatch_kernels() {
    const char* e = std::getenv("ARIA_MOE_GENERIC_DISPATCH");
    return e && e[0] == '1';
}

int grid_for_waves(long long n_items, int waves_per_block = 4) {
    long long g = (n_items + waves_per_block - 1) / waves_per_block;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return int(g);
}

}  // namespace

extern "C" {

int aria_moe_route(const void* logits, int logits_f32, void* scores, int32_t* indices, int32_t* counts, int64_t T, int64_t E,
                   int64_t k, void* stream) {
    if (!logits || !scores || !indices || !counts || T < 0 || E <= 0 || k <= 0) return ARIA_ERR_INVALID;
    if (E > 64 * kMaxPerLane || k > 8 || k > E) return ARIA_ERR_UNSUPPORTED;
#ifdef ARIA_EMU
    std::memset(counts, 0, sizeof(int32_t) * E);
#else
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * E, static_cast<hipStream_t>(stream)) != hipSuccess) return ARIA_ERR_LAUNCH;
#endif
    if (T == 0) return ARIA_OK;
    dim3 grid(std::min(grid_for_waves(T), 512)), block(256);  // few enough blocks that the per-block histogram flush stays cheap
    if (logits_f32)
        ARIA_LAUNCH((route_kernel<true>), grid, block, 0, stream, logits, scores, indices, counts, int(T), int(E), int(k));
    else
        ARIA_LAUNCH((route_kernel<false>), grid, block, 0, stream, logits, scores, indices, counts, int(T), int(E), int(k));
    return aria_check_launch();
}

int aria_moe_router_fused(const void* x, const void* w, void* logits, void* scores, int32_t* indices, int32_t* counts, int64_t T, int64_t D,
                          int64_t E, int64_t k, int64_t ldx, void* stream) {
    if (!x || !w || !logits || !scores || !indices || !counts || T < 0 || D <= 0 || E <= 0 || k <= 0) return ARIA_ERR_INVALID;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w) & 15) || (reinterpret_cast<uintptr_t>(logits) & 15) || (ldx & 7))
        return ARIA_ERR_ALIGN;
    if ((E != 32 && E != 64) || (D % 256) || k > 8 || k > E || T >= (1ll << 26)) return ARIA_ERR_UNSUPPORTED;   // (else: aria_gemm_bf16 + aria_moe_route)
#ifdef ARIA_EMU
    for (int64_t e = 0; e < E; ++e) counts[e] = 0;
#else
    if (hipMemsetAsync(counts, 0, sizeof(int32_t) * E, static_cast<hipStream_t>(stream)) != hipSuccess) return ARIA_ERR_LAUNCH;
#endif
    if (T == 0) return ARIA_OK;
    const dim3 grid(unsigned((T + 31) / 32)), block(unsigned(2 * E));   // (one wave per 32-expert block)
    if (E == 64)
        ARIA_LAUNCH((router_fused_kernel<2>), grid, block, 0, stream, static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w),
                    static_cast<bf16_t*>(logits), static_cast<bf16_t*>(scores), indices, counts, int(T), int(D), int(k), (long long)ldx);
    else
        ARIA_LAUNCH((router_fused_kernel<1>), grid, block, 0, stream, static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w),
                    static_cast<bf16_t*>(logits), static_cast<bf16_t*>(scores), indices, counts, int(T), int(D), int(k), (long long)ldx);
    return aria_check_launch();
}

int aria_moe_sort(const int32_t* indices, const int32_t* counts, int32_t* offsets, int32_t* sorted_src, int32_t* inv,
                  int32_t* workspace, int64_t T, int64_t E, int64_t k, void* stream) {
    if (!indices || !counts || !offsets || !sorted_src || !inv || !workspace || T < 0 || E <= 0 || k <= 0) return ARIA_ERR_INVALID;
    if (E > 64) return ARIA_ERR_UNSUPPORTED;
    const int M = int(T * k);
    const int nchunks = (M + SORT_CHUNK - 1) / SORT_CHUNK;
    int32_t* chunk_counts = workspace;
    int32_t* chunk_base = workspace + (size_t)nchunks * 64;
    if (nchunks > 0) {
        ARIA_LAUNCH(sort_count_kernel, dim3(nchunks), dim3(64), 0, stream, indices, chunk_counts, M);
    }
    ARIA_LAUNCH(sort_scan_kernel, dim3(1), dim3(64), 0, stream, counts, (const int32_t*)chunk_counts, chunk_base, offsets, int(E),
                nchunks);
    if (nchunks > 0) {
        ARIA_LAUNCH(sort_scatter_kernel, dim3(nchunks), dim3(64), 0, stream, indices, (const int32_t*)chunk_base, sorted_src,
                    inv, M);
    }
    return aria_check_launch();
}

int aria_moe_permute(const void* x, const int32_t* sorted_src, void* permuted, int64_t M, int64_t D, int64_t k, int64_t ldx,
                     void* stream) {
    if (!x || !sorted_src || !permuted || M < 0 || D <= 0 || k <= 0) return ARIA_ERR_INVALID;
    if ((D & 7) || (ldx & 7)) return ARIA_ERR_ALIGN;
    if (M == 0) return ARIA_OK;
    if (D == 2560 && !generic_dispatch_kernels())
        ARIA_LAUNCH((permute_rows_kernel<5>), dim3(grid_for_waves(M)), dim3(256), 0, stream, static_cast<const bf16_t*>(x), sorted_src,
                    static_cast<bf16_t*>(permuted), int(M), int(k), (long long)ldx);
    else if (D == 512 && !generic_dispatch_kernels())
        ARIA_LAUNCH((permute_rows_kernel<1>), dim3(grid_for_waves(M)), dim3(256), 0, stream, static_cast<const bf16_t*>(x), sorted_src,
                    static_cast<bf16_t*>(permuted), int(M), int(k), (long long)ldx);
    else
        ARIA_LAUNCH(permute_kernel, dim3(grid_for_waves(M)), dim3(256), 0, stream, static_cast<const bf16_t*>(x), sorted_src,
                    static_cast<bf16_t*>(permuted), int(M), int(D), int(k), (long long)ldx);
    return aria_check_launch();
}

int aria_moe_unpermute(const void* expert_out, const int32_t* inv, const void* scores, const void* add, void* out, int64_t T,
                       int64_t D, int64_t k, void* stream) {
    return aria_moe_unpermute_res(expert_out, inv, scores, add, nullptr, out, T, D, k, stream);
}

int aria_moe_unpermute_res(const void* expert_out, const int32_t* inv, const void* scores, const void* add, const void* residual, void* out,
                           int64_t T, int64_t D, int64_t k, void* stream) {
    if (!expert_out || !inv || !out || T < 0 || D <= 0 || k <= 0) return ARIA_ERR_INVALID;
    if (D & 7) return ARIA_ERR_ALIGN;
    if (k > 8) return ARIA_ERR_UNSUPPORTED;
    const bool rows_form = ((D == 2560 && k == 6) || (D == 512 && k == 2)) && !generic_dispatch_kernels();
    if (residual && !rows_form) return ARIA_ERR_UNSUPPORTED;   // (the generic kernel has no residual step: aria_moe_unpermute + aria_add_bf16)
    if (T == 0) return ARIA_OK;
    const bf16_t *eo = static_cast<const bf16_t*>(expert_out), *sc = static_cast<const bf16_t*>(scores), *ad = static_cast<const bf16_t*>(add);
    const bf16_t* rs = static_cast<const bf16_t*>(residual);
    if (D == 2560 && k == 6 && !generic_dispatch_kernels())   // Aria's width and top-k: compile-time row width, everything in flight together
        ARIA_LAUNCH((unpermute_rows_kernel<5, 6>), dim3(grid_for_waves(T)), dim3(256), 0, stream, eo, inv, sc, ad, rs, static_cast<bf16_t*>(out), int(T));
    else if (D == 512 && k == 2 && !generic_dispatch_kernels())   // (the same template at a width the CPU suite runs)
        ARIA_LAUNCH((unpermute_rows_kernel<1, 2>), dim3(grid_for_waves(T)), dim3(256), 0, stream, eo, inv, sc, ad, rs, static_cast<bf16_t*>(out), int(T));
    else
        ARIA_LAUNCH((unpermute_kernel<8>), dim3(grid_for_waves(T)), dim3(256), 0, stream, eo, inv, sc, ad, static_cast<bf16_t*>(out), int(T), int(D),
                    int(k));
    return aria_check_launch();
}

int aria_moe_unpermute_bwd(const void* dout, const void* expert_out, const int32_t* inv, const void* scores, void* d_expert_out,
                           void* dscores, int64_t T, int64_t D, int64_t k, void* stream) {
    if (!dout || !expert_out || !inv || !scores || !d_expert_out || !dscores || T < 0 || D <= 0 || k <= 0) return ARIA_ERR_INVALID;
    if (D & 7) return ARIA_ERR_ALIGN;
    if (T == 0) return ARIA_OK;
    const bf16_t *g = static_cast<const bf16_t*>(dout), *eo = static_cast<const bf16_t*>(expert_out), *sc = static_cast<const bf16_t*>(scores);
    if (D == 2560 && k == 6 && !generic_dispatch_kernels())
        ARIA_LAUNCH((unpermute_bwd_rows_kernel<5, 6>), dim3(grid_for_waves(T)), dim3(256), 0, stream, g, eo, inv, sc, static_cast<bf16_t*>(d_expert_out),
                    static_cast<bf16_t*>(dscores), int(T));
    else if (D == 512 && k == 2 && !generic_dispatch_kernels())
        ARIA_LAUNCH((unpermute_bwd_rows_kernel<1, 2>), dim3(grid_for_waves(T)), dim3(256), 0, stream, g, eo, inv, sc, static_cast<bf16_t*>(d_expert_out),
                    static_cast<bf16_t*>(dscores), int(T));
    else
        ARIA_LAUNCH(unpermute_bwd_kernel, dim3(grid_for_waves(T)), dim3(256), 0, stream, g, eo, inv, sc, static_cast<bf16_t*>(d_expert_out),
                    static_cast<bf16_t*>(dscores), int(T), int(D), int(k));
    return aria_check_launch();
}

int aria_moe_route_bwd(const void* logits, const int32_t* indices, const void* scores, const void* dscores, const int32_t* counts,
                       void* dlogits, int64_t T, int64_t E, int64_t k, float z_coeff, float aux_coeff, float aux_scale,
                       void* stream) {
    if (!logits || !indices || !scores || !dscores || !counts || !dlogits || T < 0 || E <= 0 || k <= 0) return ARIA_ERR_INVALID;
    if (E > 64 * kMaxPerLane) return ARIA_ERR_UNSUPPORTED;
    if (T == 0) return ARIA_OK;
    ARIA_LAUNCH(route_bwd_kernel, dim3(grid_for_waves(T)), dim3(256), 0, stream, static_cast<const bf16_t*>(logits), indices,
                static_cast<const bf16_t*>(scores), static_cast<const bf16_t*>(dscores), counts, static_cast<bf16_t*>(dlogits),
                int(T), int(E), int(k), z_coeff, aux_coeff, aux_scale);
    return aria_check_launch();
}

int aria_swiglu_fwd(const void* h, const void* h2, void* act, int64_t M, int64_t I, void* stream) {
    if (!h || !act || M < 0 || I <= 0) return ARIA_ERR_INVALID;
    if (I & 7) return ARIA_ERR_ALIGN;
    if (M == 0) return ARIA_OK;
    const bf16_t* a = static_cast<const bf16_t*>(h);
    const bf16_t* b = h2 ? static_cast<const bf16_t*>(h2) : a + I;
    const long long ld = h2 ? I : 2 * I;
    const long long nchunks = M * (I >> 3);
    long long g = (nchunks + 255) / 256;
    if (g > 4096) g = 4096;
    ARIA_LAUNCH(swiglu_fwd_kernel, dim3(int(g)), dim3(256), 0, stream, a, b, static_cast<bf16_t*>(act), nchunks, int(I), ld, ld);
    return aria_check_launch();
}

int aria_swiglu_bwd(const void* h, const void* h2, const void* dact, void* dh, void* dh2, int64_t M, int64_t I, void* stream) {
    if (!h || !dact || !dh || M < 0 || I <= 0) return ARIA_ERR_INVALID;
    if (I & 7) return ARIA_ERR_ALIGN;
    if ((h2 == nullptr) != (dh2 == nullptr)) return ARIA_ERR_INVALID;
    if (M == 0) return ARIA_OK;
    const bf16_t* a = static_cast<const bf16_t*>(h);
    const bf16_t* b = h2 ? static_cast<const bf16_t*>(h2) : a + I;
    bf16_t* da = static_cast<bf16_t*>(dh);
    bf16_t* db = dh2 ? static_cast<bf16_t*>(dh2) : da + I;
    const long long ld = h2 ? I : 2 * I;
    const long long nchunks = M * (I >> 3);
    long long g = (nchunks + 255) / 256;
    if (g > 4096) g = 4096;
    ARIA_LAUNCH(swiglu_bwd_kernel, dim3(int(g)), dim3(256), 0, stream, a, b, static_cast<const bf16_t*>(dact), da, db, nchunks,
                int(I), ld, ld);
    return aria_check_launch();
}

int aria_embedding_bwd(const void* dy, const int32_t* ids, void* dw, int64_t T, int64_t D, void* stream) {
    if (!dy || !ids || !dw || T < 0 || D <= 0) return ARIA_ERR_INVALID;
    if (D & 1) return ARIA_ERR_ALIGN;
    if (T == 0) return ARIA_OK;
    ARIA_LAUNCH(embedding_bwd_kernel, dim3(grid_for_waves(T)), dim3(256), 0, stream, static_cast<const bf16_t*>(dy), ids,
                static_cast<bf16_t*>(dw), int(T), int(D));
    return aria_check_launch();
}

}  // extern "C"
