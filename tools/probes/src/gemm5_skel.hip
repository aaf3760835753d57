// Skeleton of a 4-wave / 128x128-per-wave GEMM K loop (candidate "v5"): does one wave per SIMD with 256 accumulator registers, fragment
// reads and LDS-DMA pieces issued BETWEEN its own MFMAs, and ONE barrier per K-tile beat v3's 8 waves / two groups / 8 barriers?
// rc,rc only, M, N multiples of 256, K of 64; same LDS images and source-side swizzle as v3.  Prints TF/s, cycles per K-tile, and checks
// sampled outputs against the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iaria_amd/csrc -Iinclude tools/probes/src/gemm5_skel.hip -o build/abl/gemm5_skel
#include "aria_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
using namespace ad;

#ifndef SKEL_ABL
#define SKEL_ABL 0
#endif
constexpr int BK = 64;
constexpr int LDS_OPERAND = 65536, LDS_HALF = 32768, LDS_BUF = 16384;

__device__ __forceinline__ s16x8 lds_read16(uint32_t a, int off) {
    s16x8 f;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(a), "n"(0));
    return f;
}
template <int OFF>
__device__ __forceinline__ s16x8 rd(uint32_t a) {
    s16x8 f;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(a), "n"(OFF));
    return f;
}

struct Ctx {
    const char* gA;  // operand bases at the current K-tile to stage (advance 128 bytes per tile)
    const char* gB;
    uint32_t offA[2][4], offB[2][4];  // per-lane byte offsets of this wave's 4 pieces of each half
    char* lds;                        // smem + 4096 * w: this wave's 4 pieces inside a half image
    uint32_t fa, fb;                  // fragment read addresses (kk = 0), A half wm / B half wn, buffer 0
};

// piece j (0..15) of the K-tile whose bases are c.gA / c.gB, into buffer BUF: j>>2 selects (A0, A1, B0, B1), j&3 the piece
template <int BUF, int J>
__device__ __forceinline__ void dma_piece(const Ctx& c) {
    constexpr int OP = J >> 3, H = (J >> 2) & 1, S = J & 3;
    const char* g = OP == 0 ? c.gA + c.offA[H][S] : c.gB + c.offB[H][S];
    glds16(g, c.lds + OP * LDS_OPERAND + H * LDS_HALF + BUF * LDS_BUF + S * 1024);
}

template <int BUF, int KK, bool DMA, int DBUF>
__device__ __forceinline__ void step(f32x16 (&acc)[4][4], s16x8 (&fa)[2][4], s16x8 (&fb)[2][4], Ctx& c) {
    constexpr int cur = KK & 1, nxt = cur ^ 1;
    // next fragments: kk + 1 of this tile, or kk = 0 of the next tile (other buffer)
    constexpr int NB = KK < 3 ? BUF : BUF ^ 1, NK = KK < 3 ? KK + 1 : 0;
    const uint32_t a = (c.fa ^ uint32_t(NK << 5)), b = (c.fb ^ uint32_t(NK << 5));
    // the first MFMA goes out before anything else (the matrix pipe must not wait for this step's own issue work); the next fragments'
    // reads follow in pairs behind the first four MFMAs, the DMA pieces one per MFMA
#define RR(Q)                                                  \
    if (Q < 4 && !(SKEL_ABL & 2)) {                                               \
        fa[nxt][Q & 3] = rd<NB * LDS_BUF + (Q & 3) * 4096>(a); \
        fb[nxt][Q & 3] = rd<NB * LDS_BUF + (Q & 3) * 4096>(b); \
    }
#define MM(i, j, P)                                              \
    if (!(SKEL_ABL & 1)) acc[i][j] = mfma32(fa[cur][i], fb[cur][j], acc[i][j]);       \
    sched_fence();                                               \
    RR(P)                                                        \
    if (DMA && !(SKEL_ABL & 4)) dma_piece<DBUF, P>(c);           \
    sched_fence();
    MM(0, 0, 0) MM(1, 0, 1) MM(2, 0, 2) MM(3, 0, 3) MM(0, 1, 4) MM(1, 1, 5) MM(2, 1, 6) MM(3, 1, 7)
    MM(0, 2, 8) MM(1, 2, 9) MM(2, 2, 10) MM(3, 2, 11) MM(0, 3, 12) MM(1, 3, 13) MM(2, 3, 14) MM(3, 3, 15)
#undef RR
#undef MM
    sched_fence();
}

template <int BUF, bool STAGE>
__device__ __forceinline__ void k_tile5(f32x16 (&acc)[4][4], s16x8 (&fa)[2][4], s16x8 (&fb)[2][4], Ctx& c) {
    wait_lds();
    sched_fence();
    step<BUF, 0, false, 0>(acc, fa, fb, c);
    wait_lds();
    sched_fence();
    step<BUF, 1, false, 0>(acc, fa, fb, c);
    wait_lds();
    sched_fence();
    step<BUF, 2, false, 0>(acc, fa, fb, c);
    wait_lds();     // fragments (t, 3): this wave has read the last of buffer BUF
    if (!(SKEL_ABL & 32)) wait_vm<0>();   // this wave's pieces of tile t + 1 have landed
    raw_barrier();  // ... everybody's: buffer BUF ^ 1 is complete, buffer BUF is free
    step<BUF, 3, STAGE, BUF>(acc, fa, fb, c);  // reads (t + 1, 0); DMA of tile t + 2 into buffer BUF
    c.gA += 2 * BK;
    c.gB += 2 * BK;
}

__global__ __launch_bounds__(256) void gemm5_kernel(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, int noepi) {
    ARIA_DYN_SMEM(smem);
    const int t = threadIdx.x, l = t & 63, w = first_lane(t >> 6), wm = w >> 1, wn = w & 1;
    const int ntn = N / 256, ntm = M / 256, nwg = ntn * ntm;
    // XCD-contiguous chunks, groups of 4 row tiles walked column-major (as v3 order 4)
    int tile;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int GM = 4, per = GM * ntn, g = tile / per, in = tile % per, gm = min(GM, ntm - g * GM);
    const int tn = in / gm, tm = g * GM + in % gm;
    const int m0 = tm * 256, n0 = tn * 256;
    Ctx c;
    c.gA = reinterpret_cast<const char*>(A);
    c.gB = reinterpret_cast<const char*>(B);
    const uint32_t ld2 = uint32_t(2 * K);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = h * 128 + 32 * w + 8 * s + (l >> 3);
            const uint32_t chunk = uint32_t(((l & 7) ^ (l >> 4) ^ (4 * (s & 1))) * 16);
            c.offA[h][s] = uint32_t(m0 + row) * ld2 + chunk;
            c.offB[h][s] = uint32_t(n0 + row) * ld2 + chunk;
        }
    c.lds = smem + 4096 * w;
    const uint32_t lds0 = uint32_t(reinterpret_cast<uintptr_t>(smem));
    const uint32_t lane_part = uint32_t((l & 31) * 128 + ((((l >> 5) ^ ((l >> 1) & 7)) & 7) << 4));
    c.fa = lds0 + wm * LDS_HALF + lane_part;
    c.fb = lds0 + LDS_OPERAND + wn * LDS_HALF + lane_part;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    s16x8 fa[2][4], fb[2][4];
    const int nk = K / BK;
    // prologue: tiles 0 and 1 in flight, tile 0 landed, fragments (0, 0)
#define ALL16(BUF) dma_piece<BUF, 0>(c); dma_piece<BUF, 1>(c); dma_piece<BUF, 2>(c); dma_piece<BUF, 3>(c); dma_piece<BUF, 4>(c); dma_piece<BUF, 5>(c); \
    dma_piece<BUF, 6>(c); dma_piece<BUF, 7>(c); dma_piece<BUF, 8>(c); dma_piece<BUF, 9>(c); dma_piece<BUF, 10>(c); dma_piece<BUF, 11>(c);       \
    dma_piece<BUF, 12>(c); dma_piece<BUF, 13>(c); dma_piece<BUF, 14>(c); dma_piece<BUF, 15>(c);
    ALL16(0)
    c.gA += 2 * BK;
    c.gB += 2 * BK;
    if (nk > 1) {
        ALL16(1)
        wait_vm<16>();
    } else {
        wait_vm<0>();
    }
    c.gA += 2 * BK;  // (now at tile 2: what the first K-tile's kk = 3 stages)
    c.gB += 2 * BK;
    raw_barrier();
    {
        const uint32_t a = c.fa, b = c.fb;
        fa[0][0] = rd<0>(a); fb[0][0] = rd<0>(b); fa[0][1] = rd<4096>(a); fb[0][1] = rd<4096>(b);
        fa[0][2] = rd<8192>(a); fb[0][2] = rd<8192>(b); fa[0][3] = rd<12288>(a); fb[0][3] = rd<12288>(b);
    }
    // (skeleton: nk even and >= 4)
    for (int kt = 0; kt + 2 < nk; kt += 2) {
        k_tile5<0, true>(acc, fa, fb, c);
        k_tile5<1, true>(acc, fa, fb, c);
    }
    k_tile5<0, false>(acc, fa, fb, c);
    k_tile5<1, false>(acc, fa, fb, c);
    wait_lds();
    if (noepi) return;
    const int cc = l & 31, h = l >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col = n0 + wn * 128 + j * 32 + cc;
                C[(long long)row * N + col] = f2bf(acc[i][j][r]);
            }
}

static float b2f(uint16_t v) { uint32_t u = uint32_t(v) << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return uint16_t((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 8192;
    std::vector<uint16_t> hA(size_t(M) * K), hB(size_t(N) * K);
    srand(1);
    for (auto& v : hA) v = f2b(float(rand() % 2001 - 1000) / 1000.f);
    for (auto& v : hB) v = f2b(float(rand() % 2001 - 1000) / 50000.f);
    uint16_t *dA, *dB, *dC;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, size_t(M) * N * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm5_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int grid = (M / 256) * (N / 256);
    for (int noepi = 0; noepi < 2; ++noepi) {
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm5_kernel, dim3(grid), dim3(256), 131072, 0, dA, dB, dC, M, N, K, noepi);
        hipDeviceSynchronize();
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(gemm5_kernel, dim3(grid), dim3(256), 131072, 0, dA, dB, dC, M, N, K, noepi);
        hipEventRecord(b);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        ms /= 10;
        const double rounds = double(grid) / 256.0;
        printf("%s: %d x %d x %d  %.3f ms  %.1f TF/s  %.0f cycles (2.28 GHz) per K-tile per CU\n", noepi ? "no epilogue" : "with (narrow) epilogue", M, N, K, ms,
               2.0 * M * N * K / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.28e9 / (rounds * (K / 64)));
    }
    hipLaunchKernelGGL(gemm5_kernel, dim3(grid), dim3(256), 131072, 0, dA, dB, dC, M, N, K, 0);
    std::vector<uint16_t> hC(size_t(M) * N);
    hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int s = 0; s < 400; ++s) {
        const int r = rand() % M, cidx = rand() % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += double(b2f(hA[size_t(r) * K + k])) * double(b2f(hB[size_t(cidx) * K + k]));
        const double got = b2f(hC[size_t(r) * N + cidx]);
        worst = fmax(worst, fabs(got - ref) / (fabs(ref) + 0.05));
    }
    printf("worst relative error over 400 sampled outputs: %.4f %s\n", worst, worst < 0.02 ? "OK" : "MISMATCH");
    return worst < 0.02 ? 0 : 1;
}
