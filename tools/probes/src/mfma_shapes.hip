// Round 4 probes, registers only (no memory), all waves of a workgroup kept in step by one barrier per iteration:
//  (1) issue time of the bf16 MFMA shapes on gfx950: 32x32x16 (8 elements per lane), 32x32x8 "_1k" (4 per lane), 16x16x32, 16x16x16 "_1k" --
//      does the half-K 32x32x8 step occupy the matrix pipe for half the time of a 32x32x16 step (ViT attention, head dim 72 = 4.5 steps of 16)?
//  (2) do v_exp_f32 (transcendental, quarter rate) and full-rate FMAs of the same or the other wave(s) of a SIMD run UNDER the MFMAs, or do the
//      times add?  One iteration = the ViT attention kernel's per-tile mix: NM MFMAs + NE exponentials + NF fused multiply-adds.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/src/mfma_shapes.hip -o build/abl/mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NM, int NE, int NF, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(unsigned long long* out, int iters, float seed) {
    s16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = short(0x3C00 + ((threadIdx.x * 37 + e * 101) & 0x3FF) + (((threadIdx.x + e) & 1) << 15));
        b[e] = short(0x3C00 + ((threadIdx.x * 53 + e * 29) & 0x3FF) + (((threadIdx.x >> 1) + e) & 1) * 0x8000);
    }
    const s16x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
    constexpr int NACC = 4;
    f32x16 acc32[NACC];
    f32x4 acc16[NACC];
    for (int i = 0; i < NACC; ++i) {
        for (int r = 0; r < 16; ++r) acc32[i][r] = seed + 0.001f * float(i + r);
        for (int r = 0; r < 4; ++r) acc16[i][r] = seed + 0.001f * float(i + r);
    }
    float ex[8], fm[8];
    for (int i = 0; i < 8; ++i) ex[i] = seed - 0.01f * float(i + (threadIdx.x & 7)), fm[i] = seed + 0.5f + 0.01f * float(i);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (SHAPE == 3216) acc32[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[m % NACC], 0, 0, 0);
            if (SHAPE == 3208) acc32[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc32[m % NACC], 0, 0, 0);
            if (SHAPE == 1632) acc16[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc16[m % NACC], 0, 0, 0);
            if (SHAPE == 1616) acc16[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc16[m % NACC], 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) ex[e % 8] = __builtin_amdgcn_exp2f(ex[e % 8] * 0.25f - 1.0f);   // stays in (-2, 0): no denormals
#pragma unroll
        for (int f = 0; f < NF; ++f) fm[f % 8] = __builtin_fmaf(fm[f % 8], 0.999f, 0.001f);
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc32[i][0] + acc16[i][0];
    for (int i = 0; i < 8; ++i) s += ex[i] + fm[i];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 3] = t1 - t0;
        out[blockIdx.x * 3 + 1] = r1 - r0;
    }
    if (s == 12345.f) out[blockIdx.x * 3 + 2] = 1;
}

template <int SHAPE, int NM, int NE, int NF, int WAVES>
void run(const char* what, unsigned long long* out) {
    const int iters = 3000, grid = 256;
    hipLaunchKernelGGL((k<SHAPE, NM, NE, NF, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, 100, 0.f);
    hipLaunchKernelGGL((k<SHAPE, NM, NE, NF, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, iters, 0.f);
    hipDeviceSynchronize();
    unsigned long long h[256 * 3];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < grid; ++i) cyc += double(h[3 * i]), rt += double(h[3 * i + 1]);
    const double per_iter = cyc / grid / iters;   // cycles per iteration per workgroup (= per SIMD: WAVES / 4 waves each run NM + NE + NF)
    printf("{\"what\": \"%s\", \"waves_per_simd\": %d, \"mfma\": %d, \"exp\": %d, \"fma\": %d, \"cycles_per_iter\": %.1f, \"cycles_per_mfma_per_simd\": %.2f, "
           "\"clock_GHz\": %.2f}\n", what, WAVES / 4, NM, NE, NF, per_iter, NM ? per_iter / (NM * (WAVES / 4)) : 0.0, cyc / (rt * 10.0));
}

int main() {
    unsigned long long* out;
    hipMalloc(&out, 256 * 3 * 8);
    hipMemset(out, 0, 256 * 3 * 8);
    // (1) shapes: 16 MFMAs per iteration, 2 waves per SIMD in step
    run<3216, 16, 0, 0, 8>("32x32x16 bf16", out);
    run<3208, 16, 0, 0, 8>("32x32x8 bf16_1k", out);
    run<1632, 16, 0, 0, 8>("16x16x32 bf16", out);
    run<1616, 16, 0, 0, 8>("16x16x16 bf16_1k", out);
    // (2) the ViT attention tile mix (per wave and 64-key tile: 22 MFMAs, 32 exponentials, ~70 full-rate VALU), 3 waves per SIMD as shipped
    run<3216, 22, 0, 0, 12>("mfma only", out);
    run<3216, 0, 32, 0, 12>("exp only", out);
    run<3216, 0, 0, 70, 12>("fma only", out);
    run<3216, 0, 32, 70, 12>("exp + fma", out);
    run<3216, 22, 32, 0, 12>("mfma + exp", out);
    run<3216, 22, 0, 70, 12>("mfma + fma", out);
    run<3216, 22, 32, 70, 12>("mfma + exp + fma (the tile)", out);
    run<3216, 22, 16, 118, 12>("half of the exponentials as ~6 fma each", out);
    run<3216, 20, 32, 70, 12>("20 mfma (4.5-step QK if 32x32x8 is half) + exp + fma", out);
    // the same at 2 waves per SIMD
    run<3216, 22, 32, 70, 8>("2 waves/SIMD: mfma + exp + fma", out);
    run<3216, 22, 0, 0, 8>("2 waves/SIMD: mfma only", out);
    return 0;
}
