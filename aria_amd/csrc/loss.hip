// Masked mean cross-entropy over the vocabulary (aria/model/modeling_aria.py:301-323), fused forward + backward:
// one workgroup per token row, online logsumexp in fp32 over bf16 logits, then dlogits = (softmax - onehot) * scale
// written in place of the logits if the caller aliases them.  HBM-bound: 2 reads + 1 write of the row.
#include "aria_device.h"
#include "aria_hip.h"

namespace {
using namespace ad;

__global__ __launch_bounds__(256) void ce_kernel(const bf16_t* logits, const int32_t* labels, float* loss_sum, int32_t* count,
                                                 bf16_t* dlogits, float grad_scale, const int32_t* count_in, int V, long long ld) {
    ARIA_SMEM_STATIC float red_m[4];
    ARIA_SMEM_STATIC float red_s[4];
    const int t = blockIdx.x, tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int label = labels[t];
    const bf16_t* row = logits + (long long)t * ld;
    const int nch = V >> 3;
    if (label < 0) {  // ignored position (block-uniform)
        if (dlogits) {
            bf16_t* drow = dlogits + (long long)t * ld;
            for (int c = tid; c < nch; c += 256) st16(drow + c * 8, zero16());
        }
        return;
    }
    float m = -INFINITY, s = 0.f;
    for (int c = tid; c < nch; c += 256) {
        const u32x4 v = ld16(row + c * 8);
        float x[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            x[2 * q] = bflo(v[q]);
            x[2 * q + 1] = bfhi(v[q]);
        }
        float cm = x[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) cm = fmaxf(cm, x[i]);
        const float nm = fmaxf(m, cm);
        float cs = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) cs += __expf(x[i] - nm);
        s = s * __expf(m - nm) + cs;
        m = nm;
    }
    const float wm = wave_max(m);
    s = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - wm));  // lanes (or whole waves) without data carry m = -inf
    if (l == 0) {
        red_m[w] = wm;
        red_s[w] = s;
    }
    sync();
    const float M = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    const float S = red_s[0] * __expf(red_m[0] - M) + red_s[1] * __expf(red_m[1] - M) + red_s[2] * __expf(red_m[2] - M) +
                    red_s[3] * __expf(red_m[3] - M);
    const float lse = M + logf(S);
    if (tid == 0) {
        atomic_add(loss_sum, lse - bf2f(row[label]));
        atomic_add(count, 1);
    }
    if (dlogits) {
        float scale = grad_scale;
        if (count_in) scale /= float(max(1, *count_in));
        bf16_t* drow = dlogits + (long long)t * ld;
        for (int c = tid; c < nch; c += 256) {
            const u32x4 v = ld16(row + c * 8);
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float p0 = __expf(bflo(v[q]) - lse), p1 = __expf(bfhi(v[q]) - lse);
                if (c * 8 + 2 * q == label) p0 -= 1.f;
                if (c * 8 + 2 * q + 1 == label) p1 -= 1.f;
                o[q] = pack2bf(p0 * scale, p1 * scale);
            }
            st16(drow + c * 8, o);
        }
    }
}

}  // namespace

extern "C" int aria_cross_entropy(const void* logits, const int32_t* labels, float* loss_sum, int32_t* count, void* dlogits,
                                  float grad_scale, const int32_t* count_in, int64_t T, int64_t V, int64_t ld, void* stream) {
    if (!logits || !labels || !loss_sum || !count || T < 0 || V <= 0) return ARIA_ERR_INVALID;
    if ((V & 7) || (ld & 7)) return ARIA_ERR_ALIGN;
    if (T == 0) return ARIA_OK;
    ARIA_LAUNCH(ce_kernel, dim3(int(T)), dim3(256), 0, stream, static_cast<const bf16_t*>(logits), labels, loss_sum, count,
                static_cast<bf16_t*>(dlogits), grad_scale, count_in, int(V), (long long)ld);
    return aria_check_launch();
}
