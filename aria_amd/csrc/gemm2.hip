// GEMM v2 for gfx950: block tile 256x256x64, 8 waves (2 x 4), wave tile 128x64 = 4x2 v_mfma_f32_32x32x16_bf16 tiles
// (128 fp32 accumulators per lane), K/V... A/B tiles DOUBLE-BUFFERED in LDS (147 KB) with ONE barrier per K-step:
//
//     sync -> write tile k+1 (registers -> other LDS buffer) -> issue HBM loads of tile k+2 -> 32 MFMAs/wave on tile k
//
// so the HBM latency has a whole K-step to hide under and the LDS writes of one wave overlap the MFMAs of its SIMD partner
// (2 waves per SIMD).  Compared with the 128x128 v1 tile this quarters... halves LDS write bytes per flop and cuts
// LDS fragment reads per flop by a third (6 fragment reads feed 8 MFMAs).
//
// Operand forms (same as v1): rc = reduction-contiguous ([rows][k], LDS pitch 72 -> conflict-free ds_read_b128), oc =
// output-contiguous ([k][rows], exactly as it streams from HBM).  oc fragments are read with ds_read_b64_tr_b16: the
// hardware transposes 4x16 blocks on the way out of LDS, so a row-major [k][n] expert weight tile yields k-contiguous MFMA
// fragments with two reads, no VALU repacking and natural column order; the LDS pitch (288 = 16 dwords mod 64) makes the
// four k-rows of a read land in different bank quarters (conflict-free).
//
// Tile order is XCD-aware: workgroup b runs on XCD b % 8, so tile ids are remapped to give every XCD a contiguous run of
// tiles (same A row-panel, neighbouring B panels) and its private L2 sees the reuse.
#include "aria_hip.h"
#include "gemm_params.h"
#include <cstdlib>

namespace {
using namespace ad;

constexpr int BM = 256, BN = 256, BK = 64;
template <int NW> struct WCfg { static constexpr int NTH = NW * 64, SP = 2048 / (NW * 64), NJ = NW == 8 ? 2 : 4, WNW = NJ * 32; };
constexpr int PR = BK + 8;    // rc pitch (elements)
constexpr int PO = 256 + 32;  // oc pitch (elements): 144 dwords = 16 (mod 64)
constexpr int TILE_ELEMS = BM * PR;  // 18432 elements = 36864 B (== 64 * PO)
static_assert(BM * PR == 64 * PO, "rc and oc images have the same size");

// FULL: the whole 256 x 64 tile is in range (block-uniform) -> unpredicated loads, no exec-mask juggling in the K loop
template <bool OC, bool FULL, int NW>
__device__ __forceinline__ void load_tile(u32x4 (&r)[WCfg<NW>::SP], const bf16_t* base, long long ld, int row0, int row_end, int k0,
                                          int k_end, int t) {
    constexpr int NTH = WCfg<NW>::NTH, SP = WCfg<NW>::SP;
    if (FULL) {
#pragma unroll
        for (int p = 0; p < SP; ++p) {
            const int c = t + NTH * p;
            if (!OC)
                r[p] = ld16(base + (long long)(row0 + (c >> 3)) * ld + k0 + (c & 7) * 8);
            else
                r[p] = ld16(base + (long long)(k0 + (c >> 5)) * ld + row0 + (c & 31) * 8);
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < SP; ++p) {
        const int c = t + NTH * p;
        if (!OC) {
            const int row = row0 + (c >> 3), k = k0 + (c & 7) * 8;
            r[p] = (row < row_end && k < k_end) ? ld16(base + (long long)row * ld + k) : zero16();
        } else {
            const int k = k0 + (c >> 5), row = row0 + (c & 31) * 8;
            r[p] = (k < k_end && row < row_end) ? ld16(base + (long long)k * ld + row) : zero16();
        }
    }
}

template <bool OC, int NW>
__device__ __forceinline__ void store_tile(const u32x4 (&r)[WCfg<NW>::SP], bf16_t* s, int t) {
    constexpr int NTH = WCfg<NW>::NTH, SP = WCfg<NW>::SP;
#pragma unroll
    for (int p = 0; p < SP; ++p) {
        const int c = t + NTH * p;
        if (!OC)
            st16(s + (c >> 3) * PR + (c & 7) * 8, r[p]);
        else
            st16(s + (c >> 5) * PO + (c & 31) * 8, r[p]);
    }
}

// fragment of 32 consecutive rows starting at `row0` for k-substep kk: lane l <-> row row0 + (l & 31), k = 16 kk + 8 (l >> 5) + e
template <bool OC>
__device__ __forceinline__ s16x8 frag(const bf16_t* s, int row0, int kk, int l) {
    if (!OC) {
        return *reinterpret_cast<const s16x8*>(s + (row0 + (l & 31)) * PR + kk * 16 + (l >> 5) * 8);
    } else {
        const bf16_t* p = s + (kk * 16 + 8 * (l >> 5) + ((l & 15) >> 2)) * PO + row0 + 16 * ((l >> 4) & 1) + 4 * (l & 3);
        const s16x4 a0 = ds_read_tr16(p);
        const s16x4 a1 = ds_read_tr16(p + 4 * PO);
        s16x8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[e] = a0[e];
            f[4 + e] = a1[e];
        }
        return f;
    }
}

// EDGE (block-uniform): the tile hangs over the end of its row group / of N; 32x32 MFMA tiles that lie wholly outside are
// skipped with their fragment reads (wave-uniform tests) -- a grouped GEMM's last row tile per expert is mostly empty.
template <bool A_OC, bool B_OC, int NW, bool EDGE>
__device__ __forceinline__ void compute_tile(f32x16 (&acc)[4][WCfg<NW>::NJ], const bf16_t* sA, const bf16_t* sB, int l, int wm, int wn,
                                             int rows_left, int cols_left) {
    constexpr int NJ = WCfg<NW>::NJ, WNW = WCfg<NW>::WNW;
    if (EDGE) {
        if (rows_left <= 0 || cols_left <= 0) return;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            s16x8 af[4], bf[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i * 32 < rows_left) af[i] = frag<A_OC>(sA, wm * 128 + i * 32, kk, l);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (j * 32 < cols_left) bf[j] = frag<B_OC>(sB, wn * WNW + j * 32, kk, l);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (i * 32 < rows_left && j * 32 < cols_left) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
        }
        return;
    }
    // fragments are double-buffered in registers: the LDS reads of k-substep kk+1 are in flight under the MFMAs of kk
    s16x8 af[2][4], bf[2][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[0][i] = frag<A_OC>(sA, wm * 128 + i * 32, 0, l);
#pragma unroll
    for (int j = 0; j < NJ; ++j) bf[0][j] = frag<B_OC>(sB, wn * WNW + j * 32, 0, l);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
        if (kk + 1 < BK / 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[(kk + 1) & 1][i] = frag<A_OC>(sA, wm * 128 + i * 32, kk + 1, l);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[(kk + 1) & 1][j] = frag<B_OC>(sB, wn * WNW + j * 32, kk + 1, l);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = mfma32(af[kk & 1][i], bf[kk & 1][j], acc[i][j]);
    }
}

// FULL: the tile's rows and columns are all in range.  ktail: the LAST K-step is partial (grouped-K wgrad: an expert's token
// count is arbitrary) -- only that step pays for predicated loads, every other one streams unpredicated.
template <bool A_OC, bool B_OC, bool FULL, int NW>
__device__ __forceinline__ void load_step(u32x4 (&ra)[WCfg<NW>::SP], u32x4 (&rb)[WCfg<NW>::SP], const bf16_t* A, const bf16_t* B,
                                          long long lda, long long ldb, int m0, int m_end, int n0, int N, int k0, int k_end, int t,
                                          bool partial) {
    if (FULL && !partial) {
        load_tile<A_OC, true, NW>(ra, A, lda, m0, m_end, k0, k_end, t);
        load_tile<B_OC, true, NW>(rb, B, ldb, n0, N, k0, k_end, t);
    } else {
        load_tile<A_OC, false, NW>(ra, A, lda, m0, m_end, k0, k_end, t);
        load_tile<B_OC, false, NW>(rb, B, ldb, n0, N, k0, k_end, t);
    }
}

template <bool A_OC, bool B_OC, bool STAGE_FIRST, bool FULL, int NW>
__device__ __forceinline__ void k_loop(f32x16 (&acc)[4][WCfg<NW>::NJ], bf16_t* sbase, const bf16_t* A, const bf16_t* B, long long lda,
                                       long long ldb, int m0, int m_end, int n0, int N, int k_begin, int k_end, int nk, int t, int l,
                                       int wm, int wn, bool ktail) {
    const int rows_left = m_end - m0 - wm * 128, cols_left = N - n0 - wn * WCfg<NW>::WNW;
    u32x4 ra[WCfg<NW>::SP], rb[WCfg<NW>::SP];
    if (nk > 0) {
        load_step<A_OC, B_OC, FULL, NW>(ra, rb, A, B, lda, ldb, m0, m_end, n0, N, k_begin, k_end, t, ktail && nk == 1);
        store_tile<A_OC, NW>(ra, sbase, t);
        store_tile<B_OC, NW>(rb, sbase + TILE_ELEMS, t);
        if (nk > 1) load_step<A_OC, B_OC, FULL, NW>(ra, rb, A, B, lda, ldb, m0, m_end, n0, N, k_begin + BK, k_end, t, ktail && nk == 2);
    }
    for (int kt = 0; kt < nk; ++kt) {
        sync();  // tile kt is complete in buffer kt&1; nobody still reads the other buffer
        const bf16_t* sA = sbase + (kt & 1) * 2 * TILE_ELEMS;
        bf16_t* nA = sbase + ((kt + 1) & 1) * 2 * TILE_ELEMS;
        if (!STAGE_FIRST) compute_tile<A_OC, B_OC, NW, !FULL>(acc, sA, sA + TILE_ELEMS, l, wm, wn, rows_left, cols_left);
        if (kt + 1 < nk) {
            store_tile<A_OC, NW>(ra, nA, t);
            store_tile<B_OC, NW>(rb, nA + TILE_ELEMS, t);
        }
        if (kt + 2 < nk)
            load_step<A_OC, B_OC, FULL, NW>(ra, rb, A, B, lda, ldb, m0, m_end, n0, N, k_begin + (kt + 2) * BK, k_end, t,
                                            ktail && kt + 3 == nk);
        if (STAGE_FIRST) compute_tile<A_OC, B_OC, NW, !FULL>(acc, sA, sA + TILE_ELEMS, l, wm, wn, rows_left, cols_left);
    }
}

template <bool A_OC, bool B_OC, int NW>
__global__ __launch_bounds__(NW * 64) void gemm2_kernel(GemmParams p) {
    constexpr int NJ = WCfg<NW>::NJ, WNW = WCfg<NW>::WNW;
    ARIA_DYN_SMEM(smem);
    bf16_t* sbase = reinterpret_cast<bf16_t*>(smem);  // [2 buffers][A tile | B tile]
    const int t = threadIdx.x, l = t & 63, w = t >> 6, wm = NW == 8 ? w >> 2 : w >> 1, wn = NW == 8 ? w & 3 : w & 1;

    // XCD-aware bijective remap of the workgroup id
    int tn = 0, tmi = 0;
    if (p.mode != 1 && !aria_tile_coords(p, blockIdx.x, gridDim.x, tmi, tn)) return;  // mode 1: aria_grouped_tile below
    const bf16_t* A = p.A;
    const int csz = p.c_f32 ? 4 : 2;
    long long b_off = 0, c_off = 0;
    int m0 = 0, m_end = 0, k_begin = 0, k_end = p.K;
    int n0 = tn * BN;
    if (p.mode == 0) {
        m0 = tmi * BM;
        m_end = p.M;
        if (m0 >= m_end) return;
    } else if (p.mode == 1) {
        int expert = 0;
        if (!aria_grouped_tile(p, blockIdx.x, l, expert, m0, m_end, tn)) return;
        n0 = tn * BN;
        b_off = (long long)expert * p.strideB;
    } else {
        const int e = blockIdx.y;
        m0 = tmi * BM;
        m_end = p.M;
        if (m0 >= m_end) return;
        k_begin = p.offsets[e];
        k_end = p.offsets[e + 1];
        c_off = (long long)e * p.strideC;
    }
    const bf16_t* B = p.B + b_off;
    char* C = static_cast<char*>(p.C) + c_off * csz;

    f32x16 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (k_end - k_begin + BK - 1) / BK;
    const bool full = (m0 + BM <= m_end) && (n0 + BN <= p.N);
    const bool ktail = (k_end - k_begin) % BK != 0;
    // The two waves that share a SIMD (w and w+4) run the K-step in opposite orders: waves 0-3 stage the next tile
    // (LDS writes + HBM loads) and then compute, waves 4-7 compute and then stage -- one of the two is always feeding
    // the matrix pipe.  Legal because staging only touches the OTHER LDS buffer, which nobody reads between two barriers.
    const bool stage_first = NW == 4 || first_lane(w) < 4;
    if (full) {
        if (stage_first)
            k_loop<A_OC, B_OC, true, true, NW>(acc, sbase, A, B, p.lda, p.ldb, m0, m_end, n0, p.N, k_begin, k_end, nk, t, l, wm, wn, ktail);
        else
            k_loop<A_OC, B_OC, false, true, NW>(acc, sbase, A, B, p.lda, p.ldb, m0, m_end, n0, p.N, k_begin, k_end, nk, t, l, wm, wn, ktail);
    } else {
        if (stage_first)
            k_loop<A_OC, B_OC, true, false, NW>(acc, sbase, A, B, p.lda, p.ldb, m0, m_end, n0, p.N, k_begin, k_end, nk, t, l, wm, wn, ktail);
        else
            k_loop<A_OC, B_OC, false, false, NW>(acc, sbase, A, B, p.lda, p.ldb, m0, m_end, n0, p.N, k_begin, k_end, nk, t, l, wm, wn, ktail);
    }

    // ---- epilogue: bias, pair exchange so every lane owns two adjacent columns of one row, (accumulate), round, store
    const int c = l & 31, h = l >> 5, odd = l & 1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * WNW + j * 32 + c;
        const float bv = (p.bias && n < p.N) ? bf2f(p.bias[n]) : 0.f;
        const int npair = n & ~1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int rp = 0; rp < 8; ++rp) {  // register pair (2rp, 2rp+1) = two consecutive rows
                const float v0 = aria_epilogue_act(p, acc[i][j][2 * rp] + bv), v1 = aria_epilogue_act(p, acc[i][j][2 * rp + 1] + bv);
                const int r = 2 * rp;
                const int mrow = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (p.c_f32) {
                    if (n < p.N) {
                        float* d0 = reinterpret_cast<float*>(C) + (long long)mrow * p.ldc + n;
                        if (mrow < m_end) *d0 = p.accumulate ? *d0 + v0 : v0;
                        if (mrow + 1 < m_end) d0[p.ldc] = p.accumulate ? d0[p.ldc] + v1 : v1;
                    }
                } else {
                    const float got = shfl_xor(odd ? v0 : v1, 1);   // wave-uniform control flow: every lane exchanges
                    const int m = mrow + odd;
                    float lo = odd ? got : v0, hi = odd ? v1 : got;  // columns npair, npair+1 of row m
                    if (m < m_end && npair < p.N) {
                        uint32_t* dst = reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(C) + (long long)m * p.ldc + npair);
                        if (p.accumulate) {
                            const uint32_t old = *dst;
                            lo += bflo(old);
                            hi += bfhi(old);
                        }
                        *dst = pack2bf(lo, hi);
                    }
                }
            }
        }
    }
}

}  // namespace

int aria_launch_gemm2(const GemmParams& p, int a_oc, int b_oc, int ntm, int grid_y, void* stream) {
    const size_t shmem = size_t(2) * 2 * TILE_ELEMS * sizeof(bf16_t);
    const int ntn = (p.N + BN - 1) / BN;
    GemmParams q = p;
    q.ntn = ntn;
    q.ntm = ntm;
    const char* ord = std::getenv("ARIA_GEMM_ORDER");
    q.order = ord ? std::atoi(ord) : 2;  // measured best on MI355X (profiles/r01_gemm_tuning.md)
    if (ntn * ntm <= 0 || grid_y <= 0) return ARIA_OK;
    dim3 grid(unsigned(aria_tile_grid(q)), unsigned(grid_y)), block(512);
    if (a_oc && !b_oc) return ARIA_ERR_INVALID;
    // (a 4-wave variant with 128x128 wave tiles was measured slower -- profiles/r01_gemm_tuning.md -- and is not instantiated)
    if (!a_oc && !b_oc)
        ARIA_LAUNCH((gemm2_kernel<false, false, 8>), grid, block, shmem, stream, q);
    else if (!a_oc && b_oc)
        ARIA_LAUNCH((gemm2_kernel<false, true, 8>), grid, block, shmem, stream, q);
    else
        ARIA_LAUNCH((gemm2_kernel<true, true, 8>), grid, block, shmem, stream, q);
    return aria_check_launch();
}
