// RMSNorm (+ fused residual add), RoPE, small elementwise kernels: all HBM-bound, 16-byte accesses,
// one wave per row.  Reference: transformers/models/llama/modeling_llama.py:62-67, 130-160, 295-325
// (inherited by aria/model/moe_lm.py:580-602); gptfast/model.py:461-472.
#include "aria_device.h"
#include "aria_hip.h"
#include <cmath>

namespace {
using namespace ad;

constexpr int MAX_CPL = 5;  // chunks (of 8 elements) per lane: D <= 2560

__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* x, const bf16_t* res, const bf16_t* w, bf16_t* h_out,
                                                          bf16_t* y, float* rstd, int T, int D, float eps) {
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nch = D >> 3;
    for (int t = wave; t < T; t += nwaves) {
        float v[MAX_CPL][8];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_CPL; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                u32x4 a = ld16(x + (long long)t * D + c * 8);
                if (res) {
                    const u32x4 b = ld16(res + (long long)t * D + c * 8);
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] = pack2bf(bflo(a[q]) + bflo(b[q]), bfhi(a[q]) + bfhi(b[q]));
                    st16(h_out + (long long)t * D + c * 8, a);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[i][2 * q] = bflo(a[q]);
                    v[i][2 * q + 1] = bfhi(a[q]);
                    ss += v[i][2 * q] * v[i][2 * q] + v[i][2 * q + 1] * v[i][2 * q + 1];
                }
            }
        }
        ss = wave_sum_bcast(ss);
        const float r = rsqrtf(ss / float(D) + eps);
        if (l == 0 && rstd) rstd[t] = r;
#pragma unroll
        for (int i = 0; i < MAX_CPL; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                const u32x4 wv = ld16(w + c * 8);
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o[q] = pack2bf(bflo(wv[q]) * rbf(v[i][2 * q] * r), bfhi(wv[q]) * rbf(v[i][2 * q + 1] * r));
                st16(y + (long long)t * D + c * 8, o);
            }
        }
    }
}

// dx = rstd * (g - hn * mean(g * hn)) (+ dres),  g = dy * w, hn = h * rstd;  dw_partial[block][d] = sum_rows dy * bf16(hn)
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* dy, const bf16_t* h, const bf16_t* w, const float* rstd,
                                                          const bf16_t* dres, bf16_t* dx, float* dw_partial, int T, int D) {
    ARIA_DYN_SMEM(smem);
    float* red = reinterpret_cast<float*>(smem);  // [4][D]
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nch = D >> 3;
    float dwacc[MAX_CPL][8];
#pragma unroll
    for (int i = 0; i < MAX_CPL; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) dwacc[i][q] = 0.f;
    for (int t = wave; t < T; t += nwaves) {
        const float r = rstd[t];
        float g[MAX_CPL][8], hn[MAX_CPL][8];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_CPL; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                const u32x4 a = ld16(dy + (long long)t * D + c * 8);
                const u32x4 b = ld16(h + (long long)t * D + c * 8);
                const u32x4 ww = ld16(w + c * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int z = 0; z < 2; ++z) {
                        const float dyv = z ? bfhi(a[q]) : bflo(a[q]);
                        const float hv = (z ? bfhi(b[q]) : bflo(b[q])) * r;
                        const float wvv = z ? bfhi(ww[q]) : bflo(ww[q]);
                        g[i][2 * q + z] = dyv * wvv;
                        hn[i][2 * q + z] = hv;
                        dot += dyv * wvv * hv;
                        dwacc[i][2 * q + z] += dyv * rbf(hv);
                    }
                }
            }
        }
        dot = wave_sum_bcast(dot) / float(D);
#pragma unroll
        for (int i = 0; i < MAX_CPL; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                u32x4 o;
                u32x4 dr = zero16();
                if (dres) dr = ld16(dres + (long long)t * D + c * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float a0 = r * (g[i][2 * q] - hn[i][2 * q] * dot), a1 = r * (g[i][2 * q + 1] - hn[i][2 * q + 1] * dot);
                    if (dres) {
                        a0 += bflo(dr[q]);
                        a1 += bfhi(dr[q]);
                    }
                    o[q] = pack2bf(a0, a1);
                }
                st16(dx + (long long)t * D + c * 8, o);
            }
        }
    }
    // block reduce of dw over the 4 waves
#pragma unroll
    for (int i = 0; i < MAX_CPL; ++i) {
        const int c = l + 64 * i;
        if (c < nch)
#pragma unroll
            for (int q = 0; q < 8; ++q) red[wv * D + c * 8 + q] = dwacc[i][q];
    }
    sync();
    for (int d = threadIdx.x; d < D; d += blockDim.x)
        dw_partial[(long long)blockIdx.x * D + d] = red[d] + red[D + d] + red[2 * D + d] + red[3 * D + d];
}

// 32 columns x 8 row-groups per block: 128-byte row segments, 8x the parallelism of one-thread-per-column
__global__ __launch_bounds__(256) void colsum_kernel(const float* partial, bf16_t* out, int nrows, int D, int accumulate) {
    ARIA_SMEM_STATIC float red[8][32];
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int d = blockIdx.x * 32 + c;
    float s = 0.f;
    if (d < D)
        for (int r = rg; r < nrows; r += 8) s += partial[(long long)r * D + d];
    red[rg][c] = s;
    sync();
    if (rg == 0 && d < D) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += red[k][c];
        if (accumulate) tot += bf2f(out[d]);
        out[d] = f2bf(tot);
    }
}

// in-place half-split RoPE on n_heads consecutive heads of each row
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* x, const bf16_t* cs, const bf16_t* sn, long long nitems, int S,
                                                   int n_heads, int hd, long long ld, int inverse) {
    const int half = hd >> 1, cph = half >> 3;  // 16-byte chunks per half head
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < nitems; it += (long long)gridDim.x * blockDim.x) {
        const int c = int(it % cph);
        const long long r = it / cph;
        const int head = int(r % n_heads);
        const long long t = r / n_heads;
        const int pos = int(t % S);
        bf16_t* p = x + t * ld + (long long)head * hd + c * 8;
        const u32x4 a = ld16(p), b = ld16(p + half);
        const u32x4 c1 = ld16(cs + (long long)pos * hd + c * 8), c2 = ld16(cs + (long long)pos * hd + half + c * 8);
        const u32x4 s1 = ld16(sn + (long long)pos * hd + c * 8), s2 = ld16(sn + (long long)pos * hd + half + c * 8);
        u32x4 oa, ob;
        const float sg = inverse ? -1.f : 1.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float ra[2], rb[2];
#pragma unroll
            for (int z = 0; z < 2; ++z) {
                const float av = z ? bfhi(a[q]) : bflo(a[q]), bv = z ? bfhi(b[q]) : bflo(b[q]);
                const float ca = z ? bfhi(c1[q]) : bflo(c1[q]), cb = z ? bfhi(c2[q]) : bflo(c2[q]);
                const float sa = sg * (z ? bfhi(s1[q]) : bflo(s1[q])), sb = sg * (z ? bfhi(s2[q]) : bflo(s2[q]));
                // q*cos + rotate_half(q)*sin ; rotate_half = cat(-x2, x1)
                ra[z] = rbf(av * ca) + rbf(-bv * sa);
                rb[z] = rbf(bv * cb) + rbf(av * sb);
            }
            oa[q] = pack2bf(ra[0], ra[1]);
            ob[q] = pack2bf(rb[0], rb[1]);
        }
        st16(p, oa);
        st16(p + half, ob);
    }
}

__global__ __launch_bounds__(256) void add_kernel(const bf16_t* a, const bf16_t* b, bf16_t* out, long long nchunks) {
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (long long)gridDim.x * blockDim.x) {
        const u32x4 x = ld16(a + c * 8), y = ld16(b + c * 8);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pack2bf(bflo(x[q]) + bflo(y[q]), bfhi(x[q]) + bfhi(y[q]));
        st16(out + c * 8, o);
    }
}

// x *= *scale (a device-side fp32 scalar), in place: autograd hands a fused loss node its upstream gradient as a 0-dim device tensor (ones for a
// plain loss.backward(), 1 / accumulation_steps under gradient accumulation), and the node's ready-made gradients ([V, D] for lm_head, [T, D]
// for the hidden state) have to be scaled by it without a host read.  torch's mixed-dtype tensor x 0-dim-tensor multiply runs on its
// strided element-wise path: 778 + 111 us per config #3 step.  Here: the identity (scale == 1.0f, the common case) touches no memory at all,
// anything else streams at the copy rate; arithmetic = torch's (bf16 -> fp32, multiply, round to bf16).
__global__ __launch_bounds__(256) void scale_kernel(bf16_t* x, const float* scale, long long nchunks) {
    const float s = *scale;
    if (s == 1.0f) return;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (long long)gridDim.x * blockDim.x) {
        const u32x4 v = ld16(x + c * 8);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pack2bf(bflo(v[q]) * s, bfhi(v[q]) * s);
        st16(x + c * 8, o);
    }
}

// LoRA's input dropout (nn.Dropout(lora_dropout) in front of lora_A: aria/lora/layers.py:83-85, 131; recipes/config_lora.yaml:46): inverted
// dropout on 8-element chunks -- out = keep ? bf16(x / (1 - p)) : 0 and ONE mask byte per chunk (bit e = element e kept), so the backward
// re-applies the same mask from 1/16 of the bytes instead of keeping a second activation-sized tensor.  Counter-based randomness: the 8
// draws of chunk c are the 16-bit fields of two splitmix64 outputs of (seed, c) -- reproducible from (seed, index), no generator state.
__device__ __forceinline__ unsigned long long drop_mix(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void dropout_fwd_kernel(const bf16_t* x, bf16_t* out, unsigned char* mask, long long nchunks, unsigned thresh,
                                                          float scale, unsigned long long seed) {
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (long long)gridDim.x * blockDim.x) {
        const unsigned long long r0 = drop_mix(seed ^ (unsigned long long)(2 * c)), r1 = drop_mix(seed ^ (unsigned long long)(2 * c + 1));
        const u32x4 v = ld16(x + c * 8);
        u32x4 o;
        unsigned m = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long r = q < 2 ? r0 : r1;
            const bool k0 = unsigned((r >> (32 * (q & 1))) & 0xffffu) >= thresh, k1 = unsigned((r >> (32 * (q & 1) + 16)) & 0xffffu) >= thresh;
            m |= (unsigned(k0) << (2 * q)) | (unsigned(k1) << (2 * q + 1));
            o[q] = pack2bf(k0 ? bflo(v[q]) * scale : 0.f, k1 ? bfhi(v[q]) * scale : 0.f);
        }
        st16(out + c * 8, o);
        mask[c] = (unsigned char)m;
    }
}

// dx (+)= keep ? term / (1 - p) : 0 -- the backward of the dropout above applied to the adapter's input gradient
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const bf16_t* term, const unsigned char* mask, bf16_t* dx, long long nchunks, float scale,
                                                          int accumulate) {
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (long long)gridDim.x * blockDim.x) {
        const u32x4 t = ld16(term + c * 8);
        const unsigned m = mask[c];
        u32x4 o = {0u, 0u, 0u, 0u};
        if (accumulate) o = ld16(dx + c * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = ((m >> (2 * q)) & 1u) ? bflo(t[q]) * scale : 0.f, b = ((m >> (2 * q + 1)) & 1u) ? bfhi(t[q]) * scale : 0.f;
            o[q] = accumulate ? pack2bf(bflo(o[q]) + rbf(a), bfhi(o[q]) + rbf(b)) : pack2bf(a, b);
        }
        st16(dx + c * 8, o);
    }
}

// gptfast RoPE (gptfast/model.py:519-531): interleaved pairs (x[2i], x[2i+1]), bf16 freqs_cis cache [S, hd/2, 2] = (cos, sin),
// arithmetic in fp32 with ONE rounding; position of row t = pos[t] (device int32, e.g. the decode cursor) or t % S.
__global__ __launch_bounds__(256) void rope_interleaved_kernel(bf16_t* x, const bf16_t* fc, const int32_t* pos, long long nitems,
                                                               int S, int n_heads, int hd, long long ld) {
    const int cph = hd >> 3;
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < nitems; it += (long long)gridDim.x * blockDim.x) {
        const int c = int(it % cph);
        const long long r = it / cph;
        const int head = int(r % n_heads);
        const long long t = r / n_heads;
        const int ps = pos ? pos[t] : int(t % S);
        bf16_t* p = x + t * ld + (long long)head * hd + c * 8;
        const u32x4 a = ld16(p), f = ld16(fc + (long long)ps * hd + c * 8);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x0 = bflo(a[q]), x1 = bfhi(a[q]), cs = bflo(f[q]), sn = bfhi(f[q]);
            o[q] = pack2bf(x0 * cs - x1 * sn, x1 * cs + x0 * sn);
        }
        st16(p, o);
    }
}

// AdamW with fp32 master weights (decoupled weight decay, bias correction), one pass over HBM:
//   g = grad (bf16) * grad_scale; m, v updated; master -= lr * (m_hat / (sqrt(v_hat) + eps) + wd * master); param = bf16(master)
__global__ __launch_bounds__(256) void adamw_kernel(bf16_t* param, const bf16_t* grad, float* master, float* m, float* v,
                                                    long long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                    float bc2, float grad_scale) {
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += (long long)gridDim.x * blockDim.x * 2) {
        const uint32_t gw = *reinterpret_cast<const uint32_t*>(grad + i);
        float out[2];
#pragma unroll
        for (int z = 0; z < 2; ++z) {
            const float g = (z ? bfhi(gw) : bflo(gw)) * grad_scale;
            const float mm = b1 * m[i + z] + (1.f - b1) * g;
            const float vv = b2 * v[i + z] + (1.f - b2) * g * g;
            m[i + z] = mm;
            v[i + z] = vv;
            float w = master[i + z];
            w -= lr * ((mm / bc1) / (sqrtf(vv / bc2) + eps) + wd * w);
            master[i + z] = w;
            out[z] = w;
        }
        *reinterpret_cast<uint32_t*>(param + i) = pack2bf(out[0], out[1]);
    }
}

// sum of squares of a bf16 vector in fp32: the gradient norm of HF Trainer's max_grad_norm clipping (recipes/accelerate_configs/zero2.yaml:5
// gradient_clipping: auto).  Deterministic two-stage form: block b writes its partial to partial[b]; sumsq_final adds them in index order.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const bf16_t* x0, long long n0, float* partial) {
    ARIA_SMEM_STATIC float red[4];
    float acc = 0.f;
    // a shard of a flattened gradient starts on an element boundary only: up to 7 head elements in front of the first 16-byte chunk
    long long head = (long long)((16 - (reinterpret_cast<uintptr_t>(x0) & 15)) & 15) >> 1;
    if (head > n0) head = n0;
    const bf16_t* x = x0 + head;
    const long long n = n0 - head, n8 = n >> 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const u32x4 v = ld16(x + i * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += bflo(v[q]) * bflo(v[q]) + bfhi(v[q]) * bfhi(v[q]);
    }
    if (blockIdx.x == 0) {  // head and tail elements
        if (threadIdx.x < int(head)) {
            const float t = bf2f(x0[threadIdx.x]);
            acc += t * t;
        } else if (threadIdx.x >= 8 && threadIdx.x - 8 < int(n & 7)) {
            const float t = bf2f(x[(n8 << 3) + threadIdx.x - 8]);
            acc += t * t;
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    sync();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void sumsq_final_kernel(const float* partial, int nblocks, float* out, int accumulate) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) acc += partial[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) *out = (accumulate ? *out : 0.f) + acc;
}

int grid1d(long long n, int per_block, int cap = 4096) {
    long long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return int(g);
}

}  // namespace

extern "C" {

int aria_rmsnorm_fwd(const void* x, const void* res, const void* w, void* h_out, void* y, float* rstd, int64_t T, int64_t D,
                     float eps, void* stream) {
    if (!x || !w || !y || T < 0 || D <= 0 || (res && !h_out)) return ARIA_ERR_INVALID;
    if (D & 7) return ARIA_ERR_ALIGN;
    if (D > 64 * 8 * MAX_CPL) return ARIA_ERR_UNSUPPORTED;
    if (T == 0) return ARIA_OK;
    ARIA_LAUNCH(rmsnorm_fwd_kernel, dim3(grid1d(T, 4, 2048)), dim3(256), 0, stream, static_cast<const bf16_t*>(x),
                static_cast<const bf16_t*>(res), static_cast<const bf16_t*>(w), static_cast<bf16_t*>(h_out),
                static_cast<bf16_t*>(y), rstd, int(T), int(D), eps);
    return aria_check_launch();
}

int aria_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dres, void* dx,
                     float* dw_partial, int64_t nblocks, int64_t T, int64_t D, void* stream) {
    if (!dy || !h || !w || !rstd || !dx || !dw_partial || nblocks <= 0 || T < 0 || D <= 0) return ARIA_ERR_INVALID;
    if (D & 7) return ARIA_ERR_ALIGN;
    if (D > 64 * 8 * MAX_CPL) return ARIA_ERR_UNSUPPORTED;
    ARIA_LAUNCH(rmsnorm_bwd_kernel, dim3(int(nblocks)), dim3(256), size_t(4 * D * sizeof(float)), stream,
                static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(h), static_cast<const bf16_t*>(w), rstd,
                static_cast<const bf16_t*>(dres), static_cast<bf16_t*>(dx), dw_partial, int(T), int(D));
    return aria_check_launch();
}

int aria_colsum_f32(const float* partial, void* out, int64_t nrows, int64_t D, int accumulate, void* stream) {
    if (!partial || !out || nrows < 0 || D <= 0) return ARIA_ERR_INVALID;
    ARIA_LAUNCH(colsum_kernel, dim3(int((D + 31) / 32)), dim3(256), 0, stream, partial, static_cast<bf16_t*>(out), int(nrows),
                int(D), accumulate);
    return aria_check_launch();
}

int aria_rope_inplace(void* x, const void* cos, const void* sin, int64_t T, int64_t S, int64_t n_heads, int64_t hd, int64_t ld,
                      int inverse, void* stream) {
    if (!x || !cos || !sin || T < 0 || S <= 0 || n_heads <= 0 || hd <= 0) return ARIA_ERR_INVALID;
    if ((hd & 15) || (ld & 7)) return ARIA_ERR_ALIGN;
    if (T == 0) return ARIA_OK;
    const long long nitems = T * n_heads * (hd / 16);
    ARIA_LAUNCH(rope_kernel, dim3(grid1d(nitems, 256)), dim3(256), 0, stream, static_cast<bf16_t*>(x),
                static_cast<const bf16_t*>(cos), static_cast<const bf16_t*>(sin), nitems, int(S), int(n_heads), int(hd),
                (long long)ld, inverse);
    return aria_check_launch();
}

int aria_sumsq_bf16(const void* x, int64_t n, float* out, int accumulate, float* workspace, void* stream) {
    if (!out || !workspace || n < 0 || (n > 0 && !x)) return ARIA_ERR_INVALID;
    if (reinterpret_cast<uintptr_t>(x) & 1) return ARIA_ERR_ALIGN;
    const int nb = n == 0 ? 0 : grid1d((n + 7) / 8, 256, ARIA_SUMSQ_WORKSPACE_FLOATS);
    if (nb) ARIA_LAUNCH(sumsq_partial_kernel, dim3(nb), dim3(256), 0, stream, static_cast<const bf16_t*>(x), (long long)n, workspace);
    ARIA_LAUNCH(sumsq_final_kernel, dim3(1), dim3(64), 0, stream, (const float*)workspace, nb, out, accumulate);
    return aria_check_launch();
}

int aria_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
    if (!a || !b || !out || n < 0) return ARIA_ERR_INVALID;
    if (n & 7) return ARIA_ERR_ALIGN;
    if (n == 0) return ARIA_OK;
    ARIA_LAUNCH(add_kernel, dim3(grid1d(n / 8, 256)), dim3(256), 0, stream, static_cast<const bf16_t*>(a),
                static_cast<const bf16_t*>(b), static_cast<bf16_t*>(out), (long long)(n / 8));
    return aria_check_launch();
}

int aria_scale_bf16(void* x, const float* scale, int64_t n, void* stream) {
    if (!x || !scale || n < 0) return ARIA_ERR_INVALID;
    if ((n & 7) || (reinterpret_cast<uintptr_t>(x) & 15)) return ARIA_ERR_ALIGN;
    if (n == 0) return ARIA_OK;
    ARIA_LAUNCH(scale_kernel, dim3(grid1d(n / 8, 256)), dim3(256), 0, stream, static_cast<bf16_t*>(x), scale, (long long)(n / 8));
    return aria_check_launch();
}

int aria_dropout_fwd_bf16(const void* x, void* out, void* mask, int64_t n, float p, uint64_t seed, void* stream) {
    if (!x || !out || !mask || n < 0 || !(p >= 0.f && p < 1.f)) return ARIA_ERR_INVALID;
    if ((n & 7) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return ARIA_ERR_ALIGN;
    if (n == 0) return ARIA_OK;
    const unsigned thresh = unsigned(p * 65536.f + 0.5f);
    ARIA_LAUNCH(dropout_fwd_kernel, dim3(grid1d(n / 8, 256)), dim3(256), 0, stream, static_cast<const bf16_t*>(x), static_cast<bf16_t*>(out),
                static_cast<unsigned char*>(mask), (long long)(n / 8), thresh, 1.f / (1.f - p), (unsigned long long)seed);
    return aria_check_launch();
}

int aria_dropout_bwd_bf16(const void* term, const void* mask, void* dx, int64_t n, float p, int accumulate, void* stream) {
    if (!term || !mask || !dx || n < 0 || !(p >= 0.f && p < 1.f)) return ARIA_ERR_INVALID;
    if ((n & 7) || (reinterpret_cast<uintptr_t>(term) & 15) || (reinterpret_cast<uintptr_t>(dx) & 15)) return ARIA_ERR_ALIGN;
    if (n == 0) return ARIA_OK;
    ARIA_LAUNCH(dropout_bwd_kernel, dim3(grid1d(n / 8, 256)), dim3(256), 0, stream, static_cast<const bf16_t*>(term),
                static_cast<const unsigned char*>(mask), static_cast<bf16_t*>(dx), (long long)(n / 8), 1.f / (1.f - p), accumulate);
    return aria_check_launch();
}

int aria_rope_interleaved_inplace(void* x, const void* freqs_cis, const int32_t* pos, int64_t T, int64_t S, int64_t n_heads,
                                  int64_t hd, int64_t ld, void* stream) {
    if (!x || !freqs_cis || T < 0 || S <= 0 || n_heads <= 0 || hd <= 0) return ARIA_ERR_INVALID;
    if ((hd & 7) || (ld & 7)) return ARIA_ERR_ALIGN;
    if (T == 0) return ARIA_OK;
    const long long nitems = T * n_heads * (hd / 8);
    ARIA_LAUNCH(rope_interleaved_kernel, dim3(grid1d(nitems, 256)), dim3(256), 0, stream, static_cast<bf16_t*>(x),
                static_cast<const bf16_t*>(freqs_cis), pos, nitems, int(S), int(n_heads), int(hd), (long long)ld);
    return aria_check_launch();
}

int aria_adamw_step(void* param, const void* grad, float* master, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
    if (!param || !grad || !master || !m || !v || n < 0 || step <= 0) return ARIA_ERR_INVALID;
    if ((n & 1) || (reinterpret_cast<uintptr_t>(param) & 3) || (reinterpret_cast<uintptr_t>(grad) & 3)) return ARIA_ERR_ALIGN;
    if (n == 0) return ARIA_OK;
    const float bc1 = 1.f - powf(beta1, float(step)), bc2 = 1.f - powf(beta2, float(step));
    ARIA_LAUNCH(adamw_kernel, dim3(grid1d(n / 2, 256)), dim3(256), 0, stream, static_cast<bf16_t*>(param),
                static_cast<const bf16_t*>(grad), master, m, v, (long long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
    return aria_check_launch();
}

}  // extern "C"
