// r06 probe: the "lower-energy" K loop -- 4 waves x 128 x 128 per wave (one wave per SIMD, 256 accumulator registers,
// fragment reads and LDS-DMA pieces issued in the gaps between the wave's OWN MFMAs, ONE barrier per K-tile) on the 256 x 256 x 64 block
// tile of gemm3.hip -- as a stand-alone dense kernel, timed IN ONE PROCESS beside the product's gemm3_kernel<false, false, 3> /
// <false, true, 3> (libaria_hip.so through its C ABI) on the same operands.  Per K-tile a wave issues 64 MFMAs, 32 ds_read_b128 and 16 DMA
// pieces (v3: 32 MFMAs, 24 reads, 8 pieces per wave and twice the waves): 2/3 of the LDS fragment bytes per MFMA, no second wave on the SIMD.
// Round 2's HIP-level skeleton of this shape (gemm5_skel.hip) was level with v3 on wall time; what it did not report is what this one does:
// SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE under rocprofv3 (tools/probes/gemm6_probe.sh) -> effective clock and wave-cycles per
// MFMA for BOTH kernels, so "power-limited" and "issue-limited" can be told apart.
//
//   SCHED 0: tile t + 2's 16 DMA pieces in the kk = 3 step of tile t, one per MFMA gap (the skeleton's placement)
//   SCHED 1: 8 pieces in the kk = 3 step (gaps 8..15, behind the 8 fragment reads of gaps 0..3), 8 in the kk = 0 step of tile t + 1
//   SCHED 2: 4 pieces per step over kk = 3, 0, 1, 2 (gaps 12..15): the thinnest stream, the shortest flight for the last pieces
// B_OC: B given as [K][N] (expert-weight form): fragments by ds_read_b64_tr_b16 (two per fragment), source-side swizzle as in gemm3.hip.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iaria_amd/csrc -Iinclude tools/probes/src/gemm6_4wave.hip -o build/abl/gemm6_4wave -ldl
//   build/abl/gemm6_4wave M N K [b_oc] [iters]
#include "aria_device.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>
using namespace ad;

constexpr int BK = 64;
constexpr int LDS_OPERAND = 65536, LDS_HALF = 32768, LDS_BUF = 16384;
constexpr int ROWP = 528;  // LDS row pitch of the parked output tile

template <int OFF>
__device__ __forceinline__ s16x8 rd128(uint32_t a) {
    s16x8 f;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(a), "n"(OFF));
    return f;
}
template <int OFF>
__device__ __forceinline__ s16x8 rdtr(uint32_t a) {  // oc fragment: 8 reduction indices of one row = two transposing 64-bit reads, 4 k-rows apart
    s16x4 a0, a1;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=&v"(a0), "=&v"(a1) : "v"(a), "n"(OFF), "n"(OFF + 1024));
    s16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = a0[e], f[4 + e] = a1[e];
    return f;
}

template <bool B_OC>
struct Ctx {
    const char* gA;  // operand bases at the K-tile being staged
    const char* gB;
    uint32_t offA[2][4], offB[2][4];  // per-lane byte offsets of this wave's 4 pieces of each half
    long long stepB;                  // bytes per K-tile along k in B
    char* lds;                        // smem + 4096 * w: this wave's 4 pieces inside a half image
    uint32_t fa, fb[B_OC ? 4 : 1];    // fragment read addresses (kk = 0, buffer 0)
};

// piece J (0..15) of the K-tile whose bases are c.gA / c.gB, into buffer BUF: J >> 2 selects (A0, A1, B0, B1), J & 3 the piece
template <bool B_OC, int BUF, int J>
__device__ __forceinline__ void dma_piece(const Ctx<B_OC>& c) {
    constexpr int OP = J >> 3, H = (J >> 2) & 1, S = J & 3;
    const char* g = OP == 0 ? c.gA + c.offA[H][S] : c.gB + c.offB[H][S];
    glds16_raw(g, c.lds + OP * LDS_OPERAND + H * LDS_HALF + BUF * LDS_BUF + S * 1024);
}

// fragments (tile in buffer NB, k sub-step NK): A rows 32 q + .., B rows (columns of C) 32 q + ..
template <bool B_OC, int NB, int NK, int Q>
__device__ __forceinline__ void frag_pair(s16x8 (&fa)[4], s16x8 (&fb)[4], const Ctx<B_OC>& c) {
    fa[Q] = rd128<NB * LDS_BUF + Q * 4096>(c.fa ^ uint32_t(NK << 5));
    if (!B_OC)
        fb[Q] = rd128<NB * LDS_BUF + Q * 4096>(c.fb[0] ^ uint32_t(NK << 5));
    else
        fb[Q] = rdtr<NB * LDS_BUF + NK * 4096>(c.fb[Q]);
}

// one k sub-step: 16 MFMAs on fragments [KK & 1]; the next sub-step's 8 fragments are read behind MFMAs 0..3 (two per gap); DMA pieces of
// the tile this step stages go into the gaps selected by SCHED (P0..P0+NP-1 behind MFMAs G0..)
template <bool B_OC, int BUF, int KK, int NP, int P0, int G0, int DBUF>
__device__ __forceinline__ void step(f32x16 (&acc)[4][4], s16x8 (&fa)[2][4], s16x8 (&fb)[2][4], Ctx<B_OC>& c) {
    constexpr int cur = KK & 1, nxt = cur ^ 1;
    constexpr int NB = KK < 3 ? BUF : BUF ^ 1, NK = KK < 3 ? KK + 1 : 0;
#define DP(G)                                                                        \
    if constexpr (NP > 0 && G >= G0 && G < G0 + NP) dma_piece<B_OC, DBUF, (P0 + G - G0) & 15>(c);
#define MM(i, j, G)                                         \
    acc[i][j] = mfma32(fa[cur][i], fb[cur][j], acc[i][j]);  \
    sched_fence();                                          \
    if constexpr (G < 4) frag_pair<B_OC, NB, NK, G>(fa[nxt], fb[nxt], c); \
    DP(G)                                                   \
    sched_fence();
    MM(0, 0, 0) MM(1, 0, 1) MM(2, 0, 2) MM(3, 0, 3) MM(0, 1, 4) MM(1, 1, 5) MM(2, 1, 6) MM(3, 1, 7)
    MM(0, 2, 8) MM(1, 2, 9) MM(2, 2, 10) MM(3, 2, 11) MM(0, 3, 12) MM(1, 3, 13) MM(2, 3, 14) MM(3, 3, 15)
#undef MM
#undef DP
}

// K-tile t in buffer BUF.  STAGE: tile t + 2 exists (its pieces are issued from the kk = 3 step on, into buffer BUF, which every wave has
// finished reading at the barrier in front of that step); CONT: this tile is still issuing pieces of tile t + 1 (SCHED 1 / 2)
template <bool B_OC, int SCHED, int BUF, bool STAGE, bool CONT>
__device__ __forceinline__ void k_tile(f32x16 (&acc)[4][4], s16x8 (&fa)[2][4], s16x8 (&fb)[2][4], Ctx<B_OC>& c) {
    // pieces of tile t + 1 still to issue (the staging pointers are at tile t + 1 until the barrier below)
    constexpr int N0 = !CONT ? 0 : SCHED == 1 ? 8 : SCHED == 2 ? 4 : 0, N12 = (CONT && SCHED == 2) ? 4 : 0;
    wait_lds();
    sched_fence();
    step<B_OC, BUF, 0, N0, SCHED == 1 ? 8 : 4, SCHED == 1 ? 8 : 12, BUF ^ 1>(acc, fa, fb, c);
    wait_lds();
    sched_fence();
    step<B_OC, BUF, 1, N12, 8, 12, BUF ^ 1>(acc, fa, fb, c);
    wait_lds();
    sched_fence();
    step<B_OC, BUF, 2, N12, 12, 12, BUF ^ 1>(acc, fa, fb, c);
    wait_lds();     // fragments (t, 3): this wave has read the last of buffer BUF
    wait_vm<0>();   // this wave's pieces of tile t + 1 have landed
    raw_barrier();  // ... everybody's: buffer BUF ^ 1 is complete, buffer BUF is free
    c.gA += 2 * BK;
    c.gB += c.stepB;
    constexpr int N3 = !STAGE ? 0 : SCHED == 0 ? 16 : SCHED == 1 ? 8 : 4;
    step<B_OC, BUF, 3, N3, 0, SCHED == 0 ? 0 : SCHED == 1 ? 8 : 12, BUF>(acc, fa, fb, c);  // reads (t + 1, 0); first pieces of tile t + 2 into buffer BUF
}

template <bool B_OC, int SCHED>
__global__ __launch_bounds__(256) void gemm6_kernel(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, int noepi) {
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
    Ctx<B_OC> c;
    c.gA = reinterpret_cast<const char*>(A);
    c.gB = reinterpret_cast<const char*>(B);
    c.stepB = B_OC ? 2ll * BK * N : 2 * BK;
    const uint32_t ld2 = uint32_t(2 * K);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // rc: piece s of wave w holds rows 32 w + 8 s + (l >> 3) of the half; 16-byte chunk (l & 7) ^ ((row >> 1) & 7)
            const int row = h * 128 + 32 * w + 8 * s + (l >> 3);
            const uint32_t chunk = uint32_t(((l & 7) ^ (l >> 4) ^ (4 * (s & 1))) * 16);
            c.offA[h][s] = uint32_t(m0 + row) * ld2 + chunk;
            if (!B_OC) {
                c.offB[h][s] = uint32_t(n0 + row) * ld2 + chunk;
            } else {
                // oc: a half image = 64 k-rows x 256 bytes (128 columns); piece s of wave w holds k-rows 16 w + 4 s + (l >> 4); the lane's 8
                // columns are 64-byte chunk (((l & 15) >> 2) ^ (k & 3)), 16-byte piece (l & 3) inside it (gemm3.hip LaneSrc<true>)
                const int k = 16 * w + 4 * s + (l >> 4);
                const int col = ((((l & 15) >> 2) ^ (k & 3)) * 32) + (l & 3) * 8;
                c.offB[h][s] = uint32_t(k) * uint32_t(2 * N) + 2u * uint32_t(n0 + h * 128 + col);
            }
        }
    c.lds = smem + 4096 * w;
    const uint32_t lds0 = uint32_t(reinterpret_cast<uintptr_t>(smem));
    const uint32_t lane_part = uint32_t((l & 31) * 128 + ((((l >> 5) ^ ((l >> 1) & 7)) & 7) << 4));
    c.fa = lds0 + wm * LDS_HALF + lane_part;
    if (!B_OC) {
        c.fb[0] = lds0 + LDS_OPERAND + wn * LDS_HALF + lane_part;
    } else {
        const int k = 8 * (l >> 5) + ((l & 15) >> 2), within = 32 * ((l >> 4) & 1) + 8 * (l & 3);
#pragma unroll
        for (int q = 0; q < (B_OC ? 4 : 1); ++q) c.fb[q] = lds0 + LDS_OPERAND + wn * LDS_HALF + uint32_t(k * 256 + ((q ^ (k & 3)) << 6) + within);
    }

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
#define ALL16(BUF)                                                                                                                       \
    dma_piece<B_OC, BUF, 0>(c); dma_piece<B_OC, BUF, 1>(c); dma_piece<B_OC, BUF, 2>(c); dma_piece<B_OC, BUF, 3>(c); dma_piece<B_OC, BUF, 4>(c);     \
    dma_piece<B_OC, BUF, 5>(c); dma_piece<B_OC, BUF, 6>(c); dma_piece<B_OC, BUF, 7>(c); dma_piece<B_OC, BUF, 8>(c); dma_piece<B_OC, BUF, 9>(c);     \
    dma_piece<B_OC, BUF, 10>(c); dma_piece<B_OC, BUF, 11>(c); dma_piece<B_OC, BUF, 12>(c); dma_piece<B_OC, BUF, 13>(c); dma_piece<B_OC, BUF, 14>(c); \
    dma_piece<B_OC, BUF, 15>(c);
    ALL16(0)
    c.gA += 2 * BK;
    c.gB += c.stepB;
    ALL16(1)   // (probe: nk >= 4 and even)
    wait_vm<16>();
    raw_barrier();
    frag_pair<B_OC, 0, 0, 0>(fa[0], fb[0], c);
    frag_pair<B_OC, 0, 0, 1>(fa[0], fb[0], c);
    frag_pair<B_OC, 0, 0, 2>(fa[0], fb[0], c);
    frag_pair<B_OC, 0, 0, 3>(fa[0], fb[0], c);
    // the first K-tile has nothing of tile 1 left to issue (the prologue staged it whole)
    k_tile<B_OC, SCHED, 0, true, false>(acc, fa, fb, c);
    for (int kt = 1; kt + 3 < nk; kt += 2) {
        k_tile<B_OC, SCHED, 1, true, true>(acc, fa, fb, c);
        k_tile<B_OC, SCHED, 0, true, true>(acc, fa, fb, c);
    }
    k_tile<B_OC, SCHED, 1, true, true>(acc, fa, fb, c);    // tile nk - 3: stages tile nk - 1
    k_tile<B_OC, SCHED, 0, false, true>(acc, fa, fb, c);   // tile nk - 2
    k_tile<B_OC, SCHED, 1, false, false>(acc, fa, fb, c);  // tile nk - 1 (its kk = 3 step reads fragments of a tile that does not exist: harmless, LDS only)
    wait_lds();
    if (noepi) return;
    // ---- epilogue: the whole 256 x 256 tile parked in LDS as bf16 rows, written out as complete 512-byte rows (gemm3.hip store_tile3_rows)
    sync();
    const int cc = l & 31, h = l >> 5, odd = l & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {
                const float v0 = acc[i][j][2 * rp], v1 = acc[i][j][2 * rp + 1];
                const int r = 2 * rp;
                const int row = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h + odd;
                const float got = xor1(odd ? v0 : v1);
                const float lo = odd ? got : v0, hi = odd ? v1 : got;
                *reinterpret_cast<uint32_t*>(smem + row * ROWP + (wn * 128 + j * 32 + (cc & ~1)) * 2) = pack2bf(lo, hi);
            }
    sync();
    const int rr = l >> 5, c8 = (l & 31) * 8;
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) {
        const int row = w * 64 + s2 * 2 + rr;
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * ROWP + c8 * 2);
        *reinterpret_cast<u32x4*>(C + (long long)(m0 + row) * N + n0 + c8) = v;
    }
}

static float b2f(uint16_t v) { uint32_t u = uint32_t(v) << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); return uint16_t((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

typedef int (*gemm_fn)(const void*, const void*, void*, const void*, int64_t, int64_t, int64_t, int, int, int64_t, int64_t, int64_t, int, int, void*);

template <class F>
static double time_ms(F&& launch, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 3328, K = argc > 3 ? atoi(argv[3]) : 2560;
    const int b_oc = argc > 4 ? atoi(argv[4]) : 0, iters = argc > 5 ? atoi(argv[5]) : 20;
    if (M % 256 || N % 256 || K % 128 || K < 256) return fprintf(stderr, "M, N multiples of 256, K of 128 (>= 256)\n"), 2;
    std::vector<uint16_t> hA(size_t(M) * K), hB(size_t(N) * K);  // B kept on the host as [N][K]; uploaded as [K][N] for b_oc
    srand(1);
    for (auto& v : hA) v = f2b(float(rand() % 2001 - 1000) / 1000.f);
    for (auto& v : hB) v = f2b(float(rand() % 2001 - 1000) / 50000.f);
    std::vector<uint16_t> hBt;
    if (b_oc) {
        hBt.resize(hB.size());
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) hBt[size_t(k) * N + n] = hB[size_t(n) * K + k];
    }
    uint16_t *dA, *dB, *dC, *dC3;
    hipMalloc(&dA, hA.size() * 2), hipMalloc(&dB, hB.size() * 2), hipMalloc(&dC, size_t(M) * N * 2), hipMalloc(&dC3, size_t(M) * N * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, b_oc ? hBt.data() : hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    const int shmem = 256 * ROWP, grid = (M / 256) * (N / 256);
    const double flops = 2.0 * M * N * K;
    std::string libpath = "aria_amd/libaria_hip.so";
    if (const char* r = getenv("GRAFT_REPO_ROOT")) libpath = std::string(r) + "/" + libpath;   // (rocprofv3 runs from /tmp)
    void* lib = dlopen(libpath.c_str(), RTLD_NOW);
    if (!lib) fprintf(stderr, "no %s: gemm3 not timed\n", libpath.c_str());
    gemm_fn aria_gemm = lib ? reinterpret_cast<gemm_fn>(dlsym(lib, "aria_gemm_bf16")) : nullptr;
    printf("{\"shape\": [%d, %d, %d], \"b_oc\": %d, \"iters\": %d, \"tiles\": %d", M, N, K, b_oc, iters, grid);
#define RUN(BOC, SCH)                                                                                                                        \
    {                                                                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm6_kernel<BOC, SCH>), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);      \
        for (int noepi = 0; noepi < 2; ++noepi) {                                                                                            \
            const double ms = time_ms([&] { hipLaunchKernelGGL((gemm6_kernel<BOC, SCH>), dim3(grid), dim3(256), shmem, 0, dA, dB, dC, M, N, K, noepi); }, iters); \
            printf(", \"gemm6_sched%d%s\": {\"ms\": %.4f, \"tflops\": %.1f}", SCH, noepi ? "_no_epilogue" : "", ms, flops / (ms * 1e-3) / 1e12);   \
        }                                                                                                                                    \
        hipLaunchKernelGGL((gemm6_kernel<BOC, SCH>), dim3(grid), dim3(256), shmem, 0, dA, dB, dC, M, N, K, 0);                               \
        hipDeviceSynchronize();                                                                                                              \
    }
    double worst = 0;
    auto check = [&](const char* what) {
        std::vector<uint16_t> hC(size_t(M) * N);
        hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
        double wst = 0;
        srand(7);
        for (int s = 0; s < 600; ++s) {
            const int r = rand() % M, ci = rand() % N;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += double(b2f(hA[size_t(r) * K + k])) * double(b2f(hB[size_t(ci) * K + k]));
            wst = fmax(wst, fabs(b2f(hC[size_t(r) * N + ci]) - ref) / (fabs(ref) + 0.05));
        }
        printf(", \"%s_worst_rel_err\": %.4f", what, wst);
        worst = fmax(worst, wst);
        hipMemset(dC, 0, hC.size() * 2);
    };
    const int reps = argc > 6 ? atoi(argv[6]) : 2;
    const double sustain_s = argc > 7 ? atof(argv[7]) : 0.0;
    if (sustain_s > 0) {
        // SUSTAINED mode: each kernel runs back to back for sustain_s seconds, blocks alternating gemm3 / gemm6 (sched 1) / gemm3 / gemm6 ... -- the chip
        // only reaches its power-limited clock after ~1 s of load (a 5 ms burst runs at ~1.8 GHz, the training step's GEMMs at 1.45-1.56 GHz), so
        // "lower energy per flop" can only show here.  Per block: TF/s of every 0.5 s window, in order.
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm6_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm6_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm6_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm6_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
        auto block = [&](const char* name, auto&& launch) {
            printf(", \"%s\": [", name);
            double total = 0;
            bool first = true;
            while (total < sustain_s) {
                double win = 0;
                int n = 0;
                while (win < 0.5) {
                    const double ms = time_ms(launch, 40);   // (3 untimed + 40 timed launches per call)
                    win += ms * 40 * 1e-3, n += 40;
                }
                printf("%s%.0f", first ? "" : ", ", flops * n / win / 1e12);
                first = false;
                total += win;
            }
            printf("]");
        };
        for (int rep = 0; rep < reps; ++rep) {
            char nm[64];
            if (aria_gemm) {
                snprintf(nm, sizeof nm, "sustained_gemm3_block%d", rep);
                block(nm, [&] { aria_gemm(dA, dB, dC3, nullptr, M, N, K, 0, b_oc, K, b_oc ? N : K, N, 0, 0, nullptr); });
            }
            snprintf(nm, sizeof nm, "sustained_gemm6_sched1_block%d", rep);
            if (!b_oc) block(nm, [&] { hipLaunchKernelGGL((gemm6_kernel<false, 1>), dim3(grid), dim3(256), shmem, 0, dA, dB, dC, M, N, K, 0); });
            else block(nm, [&] { hipLaunchKernelGGL((gemm6_kernel<true, 1>), dim3(grid), dim3(256), shmem, 0, dA, dB, dC, M, N, K, 0); });
            snprintf(nm, sizeof nm, "sustained_gemm6_sched0_block%d", rep);
            if (!b_oc) block(nm, [&] { hipLaunchKernelGGL((gemm6_kernel<false, 0>), dim3(grid), dim3(256), shmem, 0, dA, dB, dC, M, N, K, 0); });
            else block(nm, [&] { hipLaunchKernelGGL((gemm6_kernel<true, 0>), dim3(grid), dim3(256), shmem, 0, dA, dB, dC, M, N, K, 0); });
        }
        printf("}\n");
        return 0;
    }
    for (int rep = 0; rep < reps; ++rep) {  // interleaved rounds: box drift / clock ramp shows as a difference between them
        if (aria_gemm) {
            const double ms = time_ms([&] { aria_gemm(dA, dB, dC3, nullptr, M, N, K, 0, b_oc, K, b_oc ? N : K, N, 0, 0, nullptr); }, iters);
            printf(", \"gemm3_round%d\": {\"ms\": %.4f, \"tflops\": %.1f}", rep, ms, flops / (ms * 1e-3) / 1e12);
        }
        if (!b_oc) {
            RUN(false, 0) if (rep == 0) check("sched0");
            RUN(false, 1) if (rep == 0) check("sched1");
            RUN(false, 2) if (rep == 0) check("sched2");
        } else {
            RUN(true, 0) if (rep == 0) check("sched0");
            RUN(true, 1) if (rep == 0) check("sched1");
            RUN(true, 2) if (rep == 0) check("sched2");
        }
    }
    printf(", \"ok\": %s}\n", worst < 0.02 ? "true" : "false");
    return worst < 0.02 ? 0 : 1;
}
