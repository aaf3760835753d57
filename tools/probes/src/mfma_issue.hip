// How fast does ONE wave per SIMD issue back-to-back independent MFMAs, against two waves per SIMD -- and does it depend on the accumulators
// living in AGPRs (16 x 16 registers = the 128 x 128 wave tile of a 4-wave GEMM) or in a few VGPR tuples, or on the MFMA shape?
// Registers only, no memory.  Prints shader-clock cycles per MFMA per SIMD (32 = the pipe's rate for 32x32x16 bf16, 16 for 16x16x32) and the clock.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/src/mfma_issue.hip -o build/abl/mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int SHAPE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(unsigned long long* out, int iters, float seed) {
    s16x8 a, b;
    for (int e = 0; e < 8; ++e) {  // bf16 bit patterns of ordinary magnitudes (0x3C00..0x3FFF = 0.0078 .. 2), both signs: not zeros / denormals
        a[e] = short(0x3C00 + ((threadIdx.x * 37 + e * 101) & 0x3FF) + (((threadIdx.x + e) & 1) << 15));
        b[e] = short(0x3C00 + ((threadIdx.x * 53 + e * 29) & 0x3FF) + (((threadIdx.x >> 1) + e) & 1) * 0x8000);
    }
    f32x16 acc32[SHAPE == 32 ? NACC : 1];
    f32x4 acc16[SHAPE == 16 ? NACC : 1];
    for (int i = 0; i < (SHAPE == 32 ? NACC : 1); ++i)
        for (int r = 0; r < 16; ++r) acc32[i][r] = seed + 0.001f * float(i + r);
    for (int i = 0; i < (SHAPE == 16 ? NACC : 1); ++i)
        for (int r = 0; r < 4; ++r) acc16[i][r] = seed + 0.001f * float(i + r);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (SHAPE == 32)
                acc32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[i], 0, 0, 0);
            else
                acc16[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc16[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < (SHAPE == 32 ? NACC : 1); ++i) s += acc32[i][0];
    for (int i = 0; i < (SHAPE == 16 ? NACC : 1); ++i) s += acc16[i][0];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 3] = t1 - t0;
        out[blockIdx.x * 3 + 1] = r1 - r0;
    }
    if (s == 12345.f) out[blockIdx.x * 3 + 2] = 1;
}

template <int NACC, int SHAPE, int WAVES>
void run(const char* what, unsigned long long* out) {
    const int iters = 4000, grid = 256;
    hipLaunchKernelGGL((k<NACC, SHAPE, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, 100, 0.f);
    hipLaunchKernelGGL((k<NACC, SHAPE, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, iters, 0.f);
    hipDeviceSynchronize();
    unsigned long long h[256 * 3];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < grid; ++i) {
        cyc += double(h[3 * i]);
        rt += double(h[3 * i + 1]);
    }
    const double per_simd = double(iters) * NACC * (WAVES / 4);  // MFMAs one SIMD executed
    const double flops = (SHAPE == 32 ? 32768.0 : 16384.0);
    printf("%-52s waves/SIMD %d: %6.1f cycles per MFMA per SIMD, clock %.2f GHz, %7.0f TF/s chip-wide\n", what, WAVES / 4, cyc / grid / per_simd,
           cyc / (rt * 10.0), per_simd * flops * 4 * 256 / (rt / grid * 10e-9) / 1e12);
}

// The phase pattern of a GEMM K loop: bursts of `BURST` MFMAs over NACC accumulators in rotation (dependency distance NACC), each burst followed by
// a barrier (all waves of the workgroup in step) and optionally preceded by NREAD ds_read_b128 that the burst's first MFMA waits for.
template <int NACC, int BURST, int NREAD, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void kp(unsigned long long* out, int iters, float seed) {
    __shared__ s16x8 lds[WAVES * 64 * 2];
    s16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = short(0x3C00 + ((threadIdx.x * 37 + e * 101) & 0x3FF) + (((threadIdx.x + e) & 1) << 15));
        b[e] = short(0x3C00 + ((threadIdx.x * 53 + e * 29) & 0x3FF) + (((threadIdx.x >> 1) + e) & 1) * 0x8000);
    }
    lds[threadIdx.x] = a;
    lds[threadIdx.x + WAVES * 64] = b;
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = seed + 0.001f * float(i + r);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        s16x8 fr[NREAD > 0 ? NREAD : 1];
#pragma unroll
        for (int q = 0; q < NREAD; ++q) fr[q] = lds[(threadIdx.x + 64 * q) % (WAVES * 64 * 2)];
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < BURST; ++m) {
            const s16x8 aa = NREAD > 0 ? fr[m % NREAD] : a;
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa, b, acc[m % NACC], 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 3] = t1 - t0;
        out[blockIdx.x * 3 + 1] = r1 - r0;
    }
    if (s == 12345.f) out[blockIdx.x * 3 + 2] = 1;
}

template <int NACC, int BURST, int NREAD, int WAVES>
void runp(const char* what, unsigned long long* out) {
    const int iters = 4000, grid = 256;
    hipLaunchKernelGGL((kp<NACC, BURST, NREAD, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, 100, 0.f);
    hipLaunchKernelGGL((kp<NACC, BURST, NREAD, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, iters, 0.f);
    hipDeviceSynchronize();
    unsigned long long h[256 * 3];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < grid; ++i) {
        cyc += double(h[3 * i]);
        rt += double(h[3 * i + 1]);
    }
    const double per_simd = double(iters) * BURST * (WAVES / 4);
    printf("%-64s %6.1f cycles per MFMA per SIMD, clock %.2f GHz, %6.0f TF/s\n", what, cyc / grid / per_simd, cyc / (rt * 10.0),
           per_simd * 32768.0 * 4 * 256 / (rt / grid * 10e-9) / 1e12);
}

int main() {
    unsigned long long* out;
    hipMalloc(&out, 256 * 3 * 8);
    hipMemset(out, 0, 256 * 3 * 8);
    run<16, 32, 4>("32x32x16, 16 accumulators (256 regs -> AGPRs)", out);
    run<4, 32, 4>("32x32x16, 4 accumulators (VGPRs)", out);
    run<8, 32, 8>("32x32x16, 8 accumulators per wave", out);
    run<4, 32, 8>("32x32x16, 4 accumulators per wave", out);
    run<32, 16, 4>("16x16x32, 32 accumulators (128 regs)", out);
    run<8, 16, 4>("16x16x32, 8 accumulators", out);
    run<16, 16, 8>("16x16x32, 16 accumulators per wave", out);
    printf("-- phase patterns, 8 waves in step (2 per SIMD), barrier before and after every burst\n");
    runp<2, 8, 0, 8>("8 MFMAs over 2 accumulators (v3 phase), no reads", out);
    runp<4, 16, 0, 8>("16 MFMAs over 4 accumulators (two-phase K-tile), no reads", out);
    runp<8, 32, 0, 8>("32 MFMAs over 8 accumulators (one phase per K-tile), no reads", out);
    runp<2, 8, 8, 8>("8 MFMAs over 2 accumulators + 8 ds_read_b128 in front", out);
    runp<4, 16, 12, 8>("16 MFMAs over 4 accumulators + 12 ds_read_b128 in front", out);
    runp<8, 32, 24, 8>("32 MFMAs over 8 accumulators + 24 ds_read_b128 in front", out);
    printf("-- the same with 4 waves (1 per SIMD)\n");
    runp<4, 16, 12, 4>("16 MFMAs over 4 accumulators + 12 ds_read_b128 in front", out);
    return 0;
}
