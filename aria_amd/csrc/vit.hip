// ViT / projector support kernels: LayerNorm, tanh-GELU, patch mask, bucketised position ids, im2col for the
// 14x14 patch-embed GEMM, position-embedding gather-add, bf16 column sums (bias gradients).
// Reference: Idefics2VisionEmbeddings / EncoderLayer (transformers/models/idefics2/modeling_idefics2.py:130-173, 330-363)
// inherited by aria/model/vision_encoder.py:58-152; aria/model/projector.py:26-189.
#include "aria_device.h"
#include "aria_hip.h"
#include <cstdlib>

namespace {
using namespace ad;

constexpr int MAX_CPL = 5;  // D <= 2560

// ------------------------------------------------------------------------------------------- LayerNorm
// y = bf16((x - mean) * rstd * w + b), statistics in fp32 (torch's bf16 LayerNorm); one wave per row
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y,
                                                            float* mean, float* rstd, int T, int D, float eps) {
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nch = D >> 3;
    for (int t = wave; t < T; t += nwaves) {
        float v[MAX_CPL][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_CPL; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                const u32x4 a = ld16(x + (long long)t * D + c * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[i][2 * q] = bflo(a[q]);
                    v[i][2 * q + 1] = bfhi(a[q]);
                    s += v[i][2 * q] + v[i][2 * q + 1];
                }
            }
        }
        const float mu = wave_sum_bcast(s) / float(D);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_CPL; ++i) {
            const int c = l + 64 * i;
            if (c < nch)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float d = v[i][q] - mu;
                    ss += d * d;
                }
        }
        const float r = rsqrtf(wave_sum_bcast(ss) / float(D) + eps);
        if (l == 0) {
            if (mean) mean[t] = mu;
            if (rstd) rstd[t] = r;
        }
#pragma unroll
        for (int i = 0; i < MAX_CPL; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                const u32x4 wv = ld16(w + c * 8), bv = ld16(b + c * 8);
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o[q] = pack2bf((v[i][2 * q] - mu) * r * bflo(wv[q]) + bflo(bv[q]),
                                   (v[i][2 * q + 1] - mu) * r * bfhi(wv[q]) + bfhi(bv[q]));
                st16(y + (long long)t * D + c * 8, o);
            }
        }
    }
}

// The same with a compile-time row width (NCPL 16-byte chunks per lane) and TWO rows in flight per wave (r05b: the ViT's 54 LayerNorms per step
// ran at 3.9 TB/s -- a wave fetched a row, reduced it twice and stored it before it asked for the next one; here the second row's loads
// are requested before the first row's reductions start, and only the chunks the width has are held in registers).  Same arithmetic per row,
// same order: bit-identical to layernorm_fwd_kernel.
template <int NCPL>
__global__ __launch_bounds__(256) void layernorm_fwd2_kernel(const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y, float* mean,
                                                             float* rstd, int T, int D, float eps) {
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nch = D >> 3;
    for (int t0 = wave; t0 < T; t0 += 2 * nwaves) {
        u32x4 raw[2][NCPL];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int t = min(t0 + r * nwaves, T - 1);   // (a second row past the end: the last row again, computed, not stored)
#pragma unroll
            for (int i = 0; i < NCPL; ++i) {
                const int c = l + 64 * i;
                raw[r][i] = c < nch ? ld16(x + (long long)t * D + c * 8) : zero16();
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int t = t0 + r * nwaves;
            float v[NCPL][8];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NCPL; ++i) {
                const int c = l + 64 * i;
                if (c < nch) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[i][2 * q] = bflo(raw[r][i][q]);
                        v[i][2 * q + 1] = bfhi(raw[r][i][q]);
                        s += v[i][2 * q] + v[i][2 * q + 1];
                    }
                }
            }
            const float mu = wave_sum_bcast(s) / float(D);
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < NCPL; ++i) {
                const int c = l + 64 * i;
                if (c < nch)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float d = v[i][q] - mu;
                        ss += d * d;
                    }
            }
            const float rs = rsqrtf(wave_sum_bcast(ss) / float(D) + eps);
            if (t >= T) continue;   // wave-uniform
            if (l == 0) {
                if (mean) mean[t] = mu;
                if (rstd) rstd[t] = rs;
            }
#pragma unroll
            for (int i = 0; i < NCPL; ++i) {
                const int c = l + 64 * i;
                if (c < nch) {
                    const u32x4 wv = ld16(w + c * 8), bv = ld16(b + c * 8);
                    u32x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        o[q] = pack2bf((v[i][2 * q] - mu) * rs * bflo(wv[q]) + bflo(bv[q]),
                                       (v[i][2 * q + 1] - mu) * rs * bfhi(wv[q]) + bfhi(bv[q]));
                    st16(y + (long long)t * D + c * 8, o);
                }
            }
        }
    }
}

// dx = rstd * (g - mean(g) - xh * mean(g * xh)), g = dy * w, xh = (x - mean) * rstd; partials of dw = sum dy*xh, db = sum dy
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* dy, const bf16_t* x, const bf16_t* w, const float* mean,
                                                            const float* rstd, bf16_t* dx, float* dw_partial, float* db_partial,
                                                            int T, int D) {
    ARIA_DYN_SMEM(smem);
    float* red = reinterpret_cast<float*>(smem);  // [4][2][D]
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nch = D >> 3;
    float dwacc[MAX_CPL][8], dbacc[MAX_CPL][8];
#pragma unroll
    for (int i = 0; i < MAX_CPL; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) dwacc[i][q] = dbacc[i][q] = 0.f;
    for (int t = wave; t < T; t += nwaves) {
        const float mu = mean[t], r = rstd[t];
        float g[MAX_CPL][8], xh[MAX_CPL][8];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < MAX_CPL; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                const u32x4 a = ld16(dy + (long long)t * D + c * 8);
                const u32x4 b = ld16(x + (long long)t * D + c * 8);
                const u32x4 ww = ld16(w + c * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int z = 0; z < 2; ++z) {
                        const float dyv = z ? bfhi(a[q]) : bflo(a[q]);
                        const float xv = ((z ? bfhi(b[q]) : bflo(b[q])) - mu) * r;
                        const float gv = dyv * (z ? bfhi(ww[q]) : bflo(ww[q]));
                        g[i][2 * q + z] = gv;
                        xh[i][2 * q + z] = xv;
                        sg += gv;
                        sgx += gv * xv;
                        dwacc[i][2 * q + z] += dyv * xv;
                        dbacc[i][2 * q + z] += dyv;
                    }
            }
        }
        sg = wave_sum_bcast(sg) / float(D);
        sgx = wave_sum_bcast(sgx) / float(D);
#pragma unroll
        for (int i = 0; i < MAX_CPL; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                u32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o[q] = pack2bf(r * (g[i][2 * q] - sg - xh[i][2 * q] * sgx), r * (g[i][2 * q + 1] - sg - xh[i][2 * q + 1] * sgx));
                st16(dx + (long long)t * D + c * 8, o);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAX_CPL; ++i) {
        const int c = l + 64 * i;
        if (c < nch)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                red[(wv * 2 + 0) * D + c * 8 + q] = dwacc[i][q];
                red[(wv * 2 + 1) * D + c * 8 + q] = dbacc[i][q];
            }
    }
    sync();
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a += red[(k * 2 + 0) * D + d];
            b += red[(k * 2 + 1) * D + d];
        }
        dw_partial[(long long)blockIdx.x * D + d] = a;
        db_partial[(long long)blockIdx.x * D + d] = b;
    }
}

// ------------------------------------------------------------------------------------------- GELU (tanh form)
__device__ __forceinline__ float gelu_tanh_f(float x) { return gelu_tanh(x); }
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
    const float k = 0.7978845608028654f;
    const float u = k * (x + 0.044715f * x * x * x);
    const float th = tanhf(u);
    return 0.5f * (1.f + th) + 0.5f * x * (1.f - th * th) * k * (1.f + 3.f * 0.044715f * x * x);
}

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* x, bf16_t* y, long long nchunks) {
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (long long)gridDim.x * blockDim.x) {
        const u32x4 a = ld16(x + c * 8);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 y = gelu_tanh2(f32x2{bflo(a[q]), bfhi(a[q])});
            o[q] = pack2bf(y.x, y.y);
        }
        st16(y + c * 8, o);
    }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* x, const bf16_t* dy, bf16_t* dx, long long nchunks) {
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (long long)gridDim.x * blockDim.x) {
        const u32x4 a = ld16(x + c * 8), g = ld16(dy + c * 8);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            o[q] = pack2bf(bflo(g[q]) * gelu_tanh_grad_f(bflo(a[q])), bfhi(g[q]) * gelu_tanh_grad_f(bfhi(a[q])));
        st16(dx + c * 8, o);
    }
}

// ------------------------------------------------------------------------------------------- patch mask / position ids
// patch_mask[n, i, j] = any(pixel_mask[n, i*p:(i+1)*p, j*p:(j+1)*p])      (vision_encoder.py:132-145)
__global__ __launch_bounds__(256) void patch_mask_kernel(const uint8_t* pm, uint8_t* out, int N, int R, int Hp, int p) {
    const long long total = (long long)N * Hp * Hp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = int(i / (Hp * Hp)), rem = int(i % (Hp * Hp)), pi = rem / Hp, pj = rem % Hp;
        int any = 0;
        for (int y = 0; y < p; ++y)
            for (int xx = 0; xx < p; ++xx) any |= pm[((long long)n * R + pi * p + y) * R + pj * p + xx];
        out[i] = any ? 1 : 0;
    }
}

// Idefics2VisionEmbeddings position ids in fp32, exactly as the reference computes them on CPU
// (modeling_idefics2.py:141-170): frac = fl32(i) * fl32(1/n_valid), clamp to fl32(1-1e-6), id = #(boundaries <= frac).
// One block per image; boundaries = host-built torch.arange(1/n, 1, 1/n) (fp32).
__global__ __launch_bounds__(256) void pos_ids_kernel(const uint8_t* patch_mask, const float* bound, int32_t* ids, int Hp, int Wp,
                                                      int n_side) {
    ARIA_SMEM_STATIC int s_nh, s_nw;
    const int n = blockIdx.x;
    const uint8_t* m = patch_mask + (long long)n * Hp * Wp;
    if (threadIdx.x == 0) {
        int nh = 0, nw = 0;
        for (int i = 0; i < Hp; ++i) nh += m[i * Wp];
        for (int j = 0; j < Wp; ++j) nw += m[j];
        s_nh = nh;
        s_nw = nw;
    }
    sync();
    const float step_h = 1.0f / float(s_nh), step_w = 1.0f / float(s_nw);
    const float cap = 0.999999f;  // fl32(1.0 - 1e-6)
    for (int idx = threadIdx.x; idx < Hp * Wp; idx += blockDim.x) {
        const int i = idx / Wp, j = idx % Wp;
        const float fh = fminf(float(i) * step_h, cap), fw = fminf(float(j) * step_w, cap);
        int bh = 0, bw = 0;
        for (int b = 0; b < n_side - 1; ++b) {
            bh += bound[b] <= fh;
            bw += bound[b] <= fw;
        }
        ids[(long long)n * Hp * Wp + idx] = m[idx] ? bh * n_side + bw : 0;
    }
}

// ------------------------------------------------------------------------------------------- im2col (patch-embed GEMM A operand)
// patches[(n*Hp + pi)*Wp + pj, c*p*p + y*p + x] = pixel[n, c, pi*p + y, pj*p + x]; columns >= 3*p*p are zero (K padded to KP)
template <bool F32IN>
__global__ __launch_bounds__(256) void im2col_kernel(const void* px_, bf16_t* out, int N, int Cc, int R, int Hp, int p, int KP) {
    const long long total = (long long)N * Hp * Hp * KP;
    const int K = Cc * p * p;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int col = int(i % KP);
        const long long row = i / KP;
        bf16_t v = 0;
        if (col < K) {
            const int c = col / (p * p), rem = col % (p * p), y = rem / p, x = rem % p;
            const int pj = int(row % Hp), pi = int((row / Hp) % Hp), n = int(row / ((long long)Hp * Hp));
            const long long src = (((long long)n * Cc + c) * R + pi * p + y) * R + pj * p + x;
            v = F32IN ? f2bf(static_cast<const float*>(px_)[src]) : static_cast<const bf16_t*>(px_)[src];
        }
        out[i] = v;
    }
}

// x[t, :] = bf16(x[t, :] + table[ids[t], :])
__global__ __launch_bounds__(256) void gather_add_kernel(bf16_t* x, const bf16_t* table, const int32_t* ids, int T, int D) {
    const int l = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nch = D >> 3;
    for (int t = wave; t < T; t += nwaves) {
        const bf16_t* src = table + (long long)ids[t] * D;
        for (int c = l; c < nch; c += 64) {
            const u32x4 a = ld16(x + (long long)t * D + c * 8), b = ld16(src + c * 8);
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = pack2bf(bflo(a[q]) + bflo(b[q]), bfhi(a[q]) + bfhi(b[q]));
            st16(x + (long long)t * D + c * 8, o);
        }
    }
}

// out[d] = bf16(sum_t x[t, d]) (bias gradient): one thread per column pair, rows split over gridDim.y with fp32 atomics avoided
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* x, float* partial, int T, int D, long long ld) {
    const int d2 = blockIdx.x * blockDim.x + threadIdx.x;  // column pair
    if (d2 * 2 >= D) return;
    const int rows_per = (T + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(T, r0 + rows_per);
    float a = 0.f, b = 0.f;
    for (int t = r0; t < r1; ++t) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(x + (long long)t * ld + d2 * 2);
        a += bflo(v);
        b += bfhi(v);
    }
    partial[(long long)blockIdx.y * D + d2 * 2] = a;
    partial[(long long)blockIdx.y * D + d2 * 2 + 1] = b;
}

int grid1d(long long n, int per_block, int cap = 4096) {
    long long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return int(g);
}

}  // namespace

extern "C" {

int aria_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, int64_t T, int64_t D,
                       float eps, void* stream) {
    if (!x || !w || !b || !y || T < 0 || D <= 0) return ARIA_ERR_INVALID;
    if (D & 7) return ARIA_ERR_ALIGN;
    if (D > 64 * 8 * MAX_CPL) return ARIA_ERR_UNSUPPORTED;
    if (T == 0) return ARIA_OK;
    const char* v1 = std::getenv("ARIA_LAYERNORM_V1");   // "1": the one-row-at-a-time kernel (A/B, bit-identity test)
    if (D <= 64 * 8 * 3 && !(v1 && v1[0] == '1'))          // the ViT's 1152 (and the CPU suite's toy widths): two rows in flight per wave
        ARIA_LAUNCH((layernorm_fwd2_kernel<3>), dim3(grid1d((T + 1) / 2, 4, 2048)), dim3(256), 0, stream, static_cast<const bf16_t*>(x),
                    static_cast<const bf16_t*>(w), static_cast<const bf16_t*>(b), static_cast<bf16_t*>(y), mean, rstd, int(T), int(D), eps);
    else
        ARIA_LAUNCH(layernorm_fwd_kernel, dim3(grid1d(T, 4, 2048)), dim3(256), 0, stream, static_cast<const bf16_t*>(x),
                    static_cast<const bf16_t*>(w), static_cast<const bf16_t*>(b), static_cast<bf16_t*>(y), mean, rstd, int(T), int(D), eps);
    return aria_check_launch();
}

int aria_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, void* dx,
                       float* dw_partial, float* db_partial, int64_t nblocks, int64_t T, int64_t D, void* stream) {
    if (!dy || !x || !w || !mean || !rstd || !dx || !dw_partial || !db_partial || nblocks <= 0 || T < 0 || D <= 0) return ARIA_ERR_INVALID;
    if (D & 7) return ARIA_ERR_ALIGN;
    if (D > 64 * 8 * MAX_CPL) return ARIA_ERR_UNSUPPORTED;
    ARIA_LAUNCH(layernorm_bwd_kernel, dim3(int(nblocks)), dim3(256), size_t(8 * D * sizeof(float)), stream,
                static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w), mean, rstd,
                static_cast<bf16_t*>(dx), dw_partial, db_partial, int(T), int(D));
    return aria_check_launch();
}

int aria_gelu_tanh_fwd(const void* x, void* y, int64_t n, void* stream) {
    if (!x || !y || n < 0) return ARIA_ERR_INVALID;
    if (n & 7) return ARIA_ERR_ALIGN;
    if (n == 0) return ARIA_OK;
    ARIA_LAUNCH(gelu_fwd_kernel, dim3(grid1d(n / 8, 256)), dim3(256), 0, stream, static_cast<const bf16_t*>(x), static_cast<bf16_t*>(y),
                (long long)(n / 8));
    return aria_check_launch();
}

int aria_gelu_tanh_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream) {
    if (!x || !dy || !dx || n < 0) return ARIA_ERR_INVALID;
    if (n & 7) return ARIA_ERR_ALIGN;
    if (n == 0) return ARIA_OK;
    ARIA_LAUNCH(gelu_bwd_kernel, dim3(grid1d(n / 8, 256)), dim3(256), 0, stream, static_cast<const bf16_t*>(x),
                static_cast<const bf16_t*>(dy), static_cast<bf16_t*>(dx), (long long)(n / 8));
    return aria_check_launch();
}

int aria_vit_patch_mask(const uint8_t* pixel_mask, uint8_t* patch_mask, int64_t N, int64_t R, int64_t patch, void* stream) {
    if (!pixel_mask || !patch_mask || N < 0 || R <= 0 || patch <= 0 || R % patch) return ARIA_ERR_INVALID;
    if (N == 0) return ARIA_OK;
    const int Hp = int(R / patch);
    ARIA_LAUNCH(patch_mask_kernel, dim3(grid1d(N * Hp * Hp, 256)), dim3(256), 0, stream, pixel_mask, patch_mask, int(N), int(R), Hp,
                int(patch));
    return aria_check_launch();
}

int aria_vit_pos_ids(const uint8_t* patch_mask, const float* boundaries, int32_t* ids, int64_t N, int64_t Hp, int64_t Wp,
                     int64_t n_side, void* stream) {
    if (!patch_mask || !boundaries || !ids || N < 0 || Hp <= 0 || Wp <= 0 || n_side <= 1) return ARIA_ERR_INVALID;
    if (N == 0) return ARIA_OK;
    ARIA_LAUNCH(pos_ids_kernel, dim3(unsigned(N)), dim3(256), 0, stream, patch_mask, boundaries, ids, int(Hp), int(Wp), int(n_side));
    return aria_check_launch();
}

int aria_vit_im2col(const void* pixels, int pixels_f32, void* patches, int64_t N, int64_t C, int64_t R, int64_t patch, int64_t KP,
                    void* stream) {
    if (!pixels || !patches || N < 0 || C <= 0 || R <= 0 || patch <= 0 || R % patch || KP < C * patch * patch) return ARIA_ERR_INVALID;
    if (KP & 7) return ARIA_ERR_ALIGN;
    if (N == 0) return ARIA_OK;
    const int Hp = int(R / patch);
    const long long total = N * Hp * Hp * KP;
    if (pixels_f32)
        ARIA_LAUNCH((im2col_kernel<true>), dim3(grid1d(total, 256, 16384)), dim3(256), 0, stream, pixels, static_cast<bf16_t*>(patches),
                    int(N), int(C), int(R), Hp, int(patch), int(KP));
    else
        ARIA_LAUNCH((im2col_kernel<false>), dim3(grid1d(total, 256, 16384)), dim3(256), 0, stream, pixels, static_cast<bf16_t*>(patches),
                    int(N), int(C), int(R), Hp, int(patch), int(KP));
    return aria_check_launch();
}

int aria_gather_add_rows(void* x, const void* table, const int32_t* ids, int64_t T, int64_t D, void* stream) {
    if (!x || !table || !ids || T < 0 || D <= 0) return ARIA_ERR_INVALID;
    if (D & 7) return ARIA_ERR_ALIGN;
    if (T == 0) return ARIA_OK;
    ARIA_LAUNCH(gather_add_kernel, dim3(grid1d(T, 4, 2048)), dim3(256), 0, stream, static_cast<bf16_t*>(x),
                static_cast<const bf16_t*>(table), ids, int(T), int(D));
    return aria_check_launch();
}

int aria_colsum_bf16(const void* x, float* partial, int64_t nparts, int64_t T, int64_t D, int64_t ld, void* stream) {
    if (!x || !partial || nparts <= 0 || T < 0 || D <= 0) return ARIA_ERR_INVALID;
    if ((D & 1) || (ld & 1)) return ARIA_ERR_ALIGN;
    ARIA_LAUNCH(colsum_bf16_kernel, dim3(unsigned((D / 2 + 255) / 256), unsigned(nparts)), dim3(256), 0, stream,
                static_cast<const bf16_t*>(x), partial, int(T), int(D), (long long)ld);
    return aria_check_launch();
}

}  // extern "C"
