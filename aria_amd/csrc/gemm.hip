// bf16 MFMA GEMM family for gfx950 (dense, grouped-by-rows, grouped-by-reduction).
//
// Replaces, behind the reference's seams:
//   * grouped_gemm.ops.gmm / sequential_gemm  -- aria/model/moe_lm.py:398-443,467-484 (fwd) and its
//     autograd backward (dgrad + per-expert wgrad)
//   * every nn.Linear GEMM on the path (q/k/v/o, shared expert, router gating, lm_head, ViT linears)
//     and their dgrad / wgrad.
//
// One kernel template, C[M,N] (+)= op(A) * op(B) [+ bias], fp32 accumulation on the matrix cores
// (v_mfma_f32_32x32x16_bf16), block tile 128x128x64, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles.
// An operand is either
//   "rc" reduction-contiguous: element (row, k) at base[row * ld + k]   (activations [M,K]; Linear W [N,K])
//   "oc" output-contiguous:    element (row, k) at base[k * ld + row]   (expert W[e] [K,N]; x and dy in wgrad)
// rc tiles sit in LDS as [row][k] (pitch 72: conflict-free ds_read_b128 fragments); oc tiles sit as
// [k][row] exactly as they stream from HBM (coalesced 16-byte loads), and a lane builds the fragments of
// TWO adjacent rows from eight ds_read_b32 (rows 2c, 2c+1 of the wave's slice live in the low / high half
// of one dword): MFMA tile t of the wave then owns rows {2c + t}.  Any permutation of the reduction index
// is legal as long as A and B use the same one, and any permutation of output rows / columns is undone in
// the epilogue, so no transposes are ever materialised in HBM or LDS.
//
// Modes
//   0 dense      : one problem.
//   1 grouped-M  : rows of A and C are grouped by expert (device-side offsets[E+1], no host sync -- the
//                  reference's GroupedGEMM.forward does tokens_per_expert.cpu(), moe_lm.py:478); block ->
//                  (expert, row tile) found in-kernel from the offsets; B += expert * strideB.
//   2 grouped-K  : per-expert wgrad dW[e] = A_e^T * dY_e, reduction over the expert's rows (variable
//                  length, masked); blockIdx.y = expert.
#include "aria_device.h"
#include "aria_hip.h"
#include "gemm_params.h"
#include <cstdlib>
#include <cstring>

namespace {
using namespace ad;

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int PR = BK + 8;   // rc LDS pitch (elements): 144 B -> 16 rows hit 16 distinct 16-B bank slots
constexpr int PO = 128;      // oc LDS pitch (elements)
constexpr int TILE_ELEMS = BM * PR;  // 9216 elements = 18432 B  (>= 64 * 128 for the oc image)

template <bool OC>
__device__ __forceinline__ void load_tile(u32x4 (&r)[4], const bf16_t* base, long long ld, int row0, int row_end, int k0,
                                          int k_end, int t) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (!OC) {
            const int row = row0 + (t >> 3) + 32 * p, k = k0 + (t & 7) * 8;
            r[p] = (row < row_end && k < k_end) ? ld16(base + (long long)row * ld + k) : zero16();
        } else {
            const int k = k0 + (t >> 4) + 16 * p, row = row0 + (t & 15) * 8;
            r[p] = (k < k_end && row < row_end) ? ld16(base + (long long)k * ld + row) : zero16();
        }
    }
}

template <bool OC>
__device__ __forceinline__ void store_tile(const u32x4 (&r)[4], bf16_t* s, int t) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (!OC)
            st16(s + ((t >> 3) + 32 * p) * PR + (t & 7) * 8, r[p]);
        else
            st16(s + ((t >> 4) + 16 * p) * PO + (t & 15) * 8, r[p]);
    }
}

// fragments of the wave's two 32-row MFMA tiles for k-substep kk (16 reduction indices)
template <bool OC>
__device__ __forceinline__ void load_frags(s16x8 (&f)[2], const bf16_t* s, int wbase, int kk, int l) {
    if (!OC) {
#pragma unroll
        for (int tIdx = 0; tIdx < 2; ++tIdx)
            f[tIdx] = *reinterpret_cast<const s16x8*>(s + (wbase + tIdx * 32 + (l & 31)) * PR + kk * 16 + (l >> 5) * 8);
    } else {
        uint32_t d[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            d[i] = *reinterpret_cast<const uint32_t*>(s + (kk * 16 + (l >> 5) * 8 + i) * PO + wbase + 2 * (l & 31));
        u32x4 lo, hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo[j] = (d[2 * j] & 0xffffu) | (d[2 * j + 1] << 16);
            hi[j] = (d[2 * j] >> 16) | (d[2 * j + 1] & 0xffff0000u);
        }
        f[0] = __builtin_bit_cast(s16x8, lo);
        f[1] = __builtin_bit_cast(s16x8, hi);
    }
}

template <bool A_OC, bool B_OC>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmParams p) {
    ARIA_DYN_SMEM(smem);
    bf16_t* sA = reinterpret_cast<bf16_t*>(smem);
    bf16_t* sB = sA + TILE_ELEMS;
    const int t = threadIdx.x, l = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;

    const int tile = blockIdx.x;
    const int tn = tile % p.ntn;
    int tmi = tile / p.ntn;
    const bf16_t* A = p.A;
    const int csz = p.c_f32 ? 4 : 2;
    // NOTE: expert offsets are carried as integers and applied once below; a pointer that is only modified on
    // one mode's path was left undefined on another path by hipcc (ROCm 7.2) -- found in the ISA, see DESIGN.md.
    long long b_off = 0, c_off = 0;
    int m0 = 0, m_end = 0, k_begin = 0, k_end = p.K;
    const int n0 = tn * BN;
    if (p.mode == 0) {
        m0 = tmi * BM;
        m_end = p.M;
    } else if (p.mode == 1) {
        // block -> (expert, local row tile): walk the (<=64-entry per step) offset table
        int e_found = -1, start = 0, end = 0, base = 0;
        for (int e0 = 0; e0 < p.E && e_found < 0; e0 += 64) {
            const int e = e0 + l;
            int o0 = 0, o1 = 0;
            if (e < p.E) {
                o0 = p.offsets[e];
                o1 = p.offsets[e + 1];
            }
            const int nt = (o1 - o0 + BM - 1) / BM;
            int incl = nt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = shfl(incl, (l - d) & 63);
                if (l >= d) incl += v;
            }
            const int excl = base + incl - nt;
            const bool mine = e < p.E && tmi >= excl && tmi < excl + nt;
            const unsigned long long mask = ballot(mine);
            if (mask) {
                const int src = __builtin_ctzll(mask);
                e_found = e0 + src;
                start = shfl(o0, src);
                end = shfl(o1, src);
                tmi -= shfl(excl, src);
            }
            base += shfl(incl, 63);
        }
        if (e_found < 0) return;  // uniform across the block
        m0 = start + tmi * BM;
        m_end = end;
        b_off = (long long)e_found * p.strideB;
    } else {
        const int e = blockIdx.y;
        m0 = tmi * BM;
        m_end = p.M;
        k_begin = p.offsets[e];
        k_end = p.offsets[e + 1];
        c_off = (long long)e * p.strideC;
    }
    const bf16_t* B = p.B + b_off;
    char* C = static_cast<char*>(p.C) + c_off * csz;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    u32x4 ra[4], rb[4];
    const int nk = (k_end - k_begin + BK - 1) / BK;
    if (nk > 0) {
        load_tile<A_OC>(ra, A, p.lda, m0, m_end, k_begin, k_end, t);
        load_tile<B_OC>(rb, B, p.ldb, n0, p.N, k_begin, k_end, t);
    }
    for (int kt = 0; kt < nk; ++kt) {
        store_tile<A_OC>(ra, sA, t);
        store_tile<B_OC>(rb, sB, t);
        sync();
        if (kt + 1 < nk) {  // next tile's HBM loads fly under this tile's MFMAs
            const int k0 = k_begin + (kt + 1) * BK;
            load_tile<A_OC>(ra, A, p.lda, m0, m_end, k0, k_end, t);
            load_tile<B_OC>(rb, B, p.ldb, n0, p.N, k0, k_end, t);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            s16x8 af[2], bf[2];
            load_frags<A_OC>(af, sA, wm * 64, kk, l);
            load_frags<B_OC>(bf, sB, wn * 64, kk, l);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
        }
        sync();
    }

    // ---- epilogue: undo the row / column permutations, add bias, (accumulate), round, store
    const int c = l & 31, h = l >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rt = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = m0 + wm * 64 + (A_OC ? 2 * rt + i : i * 32 + rt);
            if (m >= m_end) continue;
            if (B_OC) {
                const int n = n0 + wn * 64 + 2 * c;  // columns n, n+1 (N % 2 == 0)
                if (n >= p.N) continue;
                float v0 = acc[i][0][r], v1 = acc[i][1][r];
                if (p.bias) {
                    v0 += bf2f(p.bias[n]);
                    v1 += bf2f(p.bias[n + 1]);
                }
                v0 = aria_epilogue_act(p, v0);
                v1 = aria_epilogue_act(p, v1);
                if (p.c_f32) {
                    float* dst = reinterpret_cast<float*>(C) + (long long)m * p.ldc + n;
                    if (p.accumulate) {
                        v0 += dst[0];
                        v1 += dst[1];
                    }
                    dst[0] = v0;
                    dst[1] = v1;
                } else {
                    uint32_t* dst = reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(C) + (long long)m * p.ldc + n);
                    if (p.accumulate) {
                        const uint32_t old = *dst;
                        v0 += bflo(old);
                        v1 += bfhi(old);
                    }
                    *dst = pack2bf(v0, v1);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + wn * 64 + j * 32 + c;
                    if (n >= p.N) continue;
                    float v = acc[i][j][r];
                    if (p.bias) v += bf2f(p.bias[n]);
                    v = aria_epilogue_act(p, v);
                    if (p.c_f32) {
                        float* dst = reinterpret_cast<float*>(C) + (long long)m * p.ldc + n;
                        if (p.accumulate) v += *dst;
                        *dst = v;
                    } else {
                        bf16_t* dst = reinterpret_cast<bf16_t*>(C) + (long long)m * p.ldc + n;
                        if (p.accumulate) v += bf2f(*dst);
                        *dst = f2bf(v);
                    }
                }
            }
        }
    }
}

int launch_gemm(const GemmParams& p, int a_oc, int b_oc, int grid_x, int grid_y, void* stream) {
    const size_t shmem = 2 * TILE_ELEMS * sizeof(bf16_t);
    dim3 grid(grid_x, grid_y), block(NTHREADS);
    if (grid_x <= 0 || grid_y <= 0) return ARIA_OK;
    if (!a_oc && !b_oc)
        ARIA_LAUNCH((gemm_kernel<false, false>), grid, block, shmem, stream, p);
    else if (!a_oc && b_oc)
        ARIA_LAUNCH((gemm_kernel<false, true>), grid, block, shmem, stream, p);
    else if (a_oc && b_oc)
        ARIA_LAUNCH((gemm_kernel<true, true>), grid, block, shmem, stream, p);
    else
        return ARIA_ERR_INVALID;  // (oc, rc) never occurs on this path
    return aria_check_launch();
}

thread_local int g_last_variant = 0;
constexpr int GROUPED_V3_DEFAULT = 3;  // r02 in-bench A/B with the wide epilogue: v3 for both weight forms 702 ms/step, v2 for the [K,N] form 710
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// v1 (128x128 tiles, 3 blocks/CU) wins when the 256x256 grid would not fill the chip; v2 otherwise.
// ARIA_GEMM_FORCE=1|2 pins a variant (tests exercise both on small problems).
bool use_v2(long long tiles256) {
    const char* force = std::getenv("ARIA_GEMM_FORCE");
    if (force && force[0] == '1') return false;
    if (force && force[0] == '2') return true;
    return tiles256 >= 192;
}

// v3 (LDS-DMA staged, phase-scheduled 256x256 tile, gemm3.hip): the operand byte offsets must fit 32 bits.
// ARIA_GEMM_FORCE=3 pins it where eligible.
bool use_v3(long long tiles256, long long K, long long bytesA, long long bytesB, long long extentA, long long extentB) {
    if (K < 64 || bytesA >= (1ll << 32) || bytesB >= (1ll << 32) || extentA < 8 || extentB < 8) return false;
    const char* force = std::getenv("ARIA_GEMM_FORCE");
    if (force) return force[0] == '3';
    return tiles256 >= 192;
}

}  // namespace

extern "C" {

int aria_last_gemm_variant(void) { return g_last_variant; }

int64_t aria_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int a_oc, int b_oc) {
    if (M <= 0 || N <= 0 || (a_oc && !b_oc)) return 0;
    return aria_gemm3_workspace_bytes(M, N, K);
}

int aria_gemm_bf16(const void* A, const void* B, void* C, const void* bias, int64_t M, int64_t N, int64_t K, int a_oc,
                   int b_oc, int64_t lda, int64_t ldb, int64_t ldc, int c_f32, int accumulate, void* stream) {
    return aria_gemm_bf16_ws(A, B, C, bias, M, N, K, a_oc, b_oc, lda, ldb, ldc, c_f32, accumulate, nullptr, 0, stream);
}

int aria_gemm_bf16_ws(const void* A, const void* B, void* C, const void* bias, int64_t M, int64_t N, int64_t K, int a_oc,
                      int b_oc, int64_t lda, int64_t ldb, int64_t ldc, int c_f32, int accumulate, void* workspace,
                      int64_t workspace_bytes, void* stream) {
    return aria_gemm_act_bf16(A, B, C, bias, M, N, K, a_oc, b_oc, lda, ldb, ldc, c_f32, accumulate, ARIA_ACT_NONE, workspace,
                              workspace_bytes, stream);
}

int aria_gemm_act_bf16(const void* A, const void* B, void* C, const void* bias, int64_t M, int64_t N, int64_t K, int a_oc,
                       int b_oc, int64_t lda, int64_t ldb, int64_t ldc, int c_f32, int accumulate, int act, void* workspace,
                       int64_t workspace_bytes, void* stream) {
    if (act != ARIA_ACT_NONE && act != ARIA_ACT_GELU_TANH) return ARIA_ERR_UNSUPPORTED;
    if (!A || !B || !C || M < 0 || N < 0 || K < 0) return ARIA_ERR_INVALID;
    if (M == 0 || N == 0) return ARIA_OK;
    if (!aligned16(A) || !aligned16(B) || (lda & 7) || (ldb & 7) || (N & 1) || (ldc & 1)) return ARIA_ERR_ALIGN;
    if ((!a_oc || !b_oc) && (K & 7)) return ARIA_ERR_ALIGN;  // rc operands: 16-byte chunks along K
    if (a_oc && (M & 7)) return ARIA_ERR_ALIGN;
    if (b_oc && (N & 7)) return ARIA_ERR_ALIGN;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A);
    p.B = static_cast<const bf16_t*>(B);
    p.C = C;
    p.bias = static_cast<const bf16_t*>(bias);
    p.lda = lda;
    p.ldb = ldb;
    p.ldc = ldc;
    p.M = int(M);
    p.N = int(N);
    p.K = int(K);
    p.mode = 0;
    p.c_f32 = c_f32;
    p.accumulate = accumulate;
    p.act = act;
    p.ntn = int((N + BN - 1) / BN);
    const int ntm = int((M + BM - 1) / BM);
    const long long t256 = ((M + 255) / 256) * ((N + 255) / 256);
    const long long bytesA = 2 * (a_oc ? K * lda : M * lda), bytesB = 2 * (b_oc ? K * ldb : N * ldb);
    // with a workspace the partly filled last round of tiles is split along K (gemm3.hip), so small outputs fill the chip too
    const long long ws_need = workspace ? aria_gemm3_workspace_bytes(M, N, K) : 0;
    const bool splits = ws_need > 0 && workspace_bytes >= ws_need;
    if ((!a_oc || b_oc) && use_v3(splits ? (t256 < 192 ? 192 : t256) : t256, K, bytesA, bytesB, M, N))
        return g_last_variant = 3, aria_launch_gemm3(p, a_oc, b_oc, int((M + 255) / 256), stream, workspace, workspace_bytes);
    if (use_v2(t256)) return g_last_variant = 2, aria_launch_gemm2(p, a_oc, b_oc, int((M + 255) / 256), 1, stream);
    return g_last_variant = 1, launch_gemm(p, a_oc, b_oc, p.ntn * ntm, 1, stream);
}

int aria_grouped_gemm_bf16(const void* A, const void* B, void* C, const int32_t* offsets, int64_t E, int64_t M_total,
                           int64_t N, int64_t K, int b_oc, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldc,
                           void* stream) {
    if (!A || !B || !C || !offsets || E <= 0 || M_total < 0) return ARIA_ERR_INVALID;
    if (M_total == 0 || N == 0) return ARIA_OK;
    if (!aligned16(A) || !aligned16(B) || (lda & 7) || (ldb & 7) || (K & 7) || (N & 7) || (strideB & 7) || (ldc & 1))
        return ARIA_ERR_ALIGN;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A);
    p.B = static_cast<const bf16_t*>(B);
    p.C = C;
    p.lda = lda;
    p.ldb = ldb;
    p.ldc = ldc;
    p.M = int(M_total);
    p.N = int(N);
    p.K = int(K);
    p.mode = 1;
    p.offsets = offsets;
    p.E = int(E);
    p.strideB = strideB;
    p.ntn = int((N + BN - 1) / BN);
    // every expert adds at most one partial row tile
    const int max_tm = int(M_total / BM + E);
    // ARIA_GEMM_GROUPED_V3 (bit 0: [N,K] weights = dgrad, bit 1: [K,N] weights = forward) picks v3 per operand form; see the
    // in-bench A/B in profiles/r01_gemm_tuning.md for the default
    const char* gsel = std::getenv("ARIA_GEMM_GROUPED_V3");
    const int gmask = gsel ? std::atoi(gsel) : GROUPED_V3_DEFAULT;
    const char* gforce = std::getenv("ARIA_GEMM_FORCE");
    const bool gv3 = (gforce && gforce[0] == '3') || ((gmask >> (b_oc ? 1 : 0)) & 1);
    if (gv3 && use_v3((M_total / 256 + 1) * ((N + 255) / 256), K, 2 * M_total * lda, 2 * (b_oc ? K * ldb : N * ldb), M_total, N))
        return g_last_variant = 3, aria_launch_gemm3(p, 0, b_oc, int(M_total / 256 + E), stream);
    if (use_v2((M_total / 256 + 1) * ((N + 255) / 256))) return g_last_variant = 2, aria_launch_gemm2(p, 0, b_oc, int(M_total / 256 + E), 1, stream);
    return g_last_variant = 1, launch_gemm(p, 0, b_oc, p.ntn * max_tm, 1, stream);
}

// shared validation of the fused fc1 + SwiGLU entries: N2 = 2 I with I % 128 == 0 (a tile = 128 gate + 128 up columns), 16-byte aligned rows
static int glu_check(const void* A, const void* B, const void* H, const void* ACT, int64_t M, int64_t N2, int64_t K, int64_t lda, int64_t ldb,
                     int64_t ldh, int64_t ldact) {
    if (!A || !B || !ACT || M < 0 || N2 <= 0 || K <= 0) return ARIA_ERR_INVALID;
    if (!aligned16(A) || !aligned16(B) || !aligned16(ACT) || (H && !aligned16(H)) || (lda & 7) || (ldb & 7) || (K & 7) || (ldact & 7) ||
        (H && (ldh & 7)))
        return ARIA_ERR_ALIGN;
    if ((N2 & 1) || ((N2 / 2) % 128) || K < 64 || 2 * lda >= (1ll << 24) || 2 * ldb >= (1ll << 24)) return ARIA_ERR_UNSUPPORTED;
    return ARIA_OK;
}

int aria_grouped_gemm_swiglu_bf16(const void* A, const void* B, void* H, void* ACT, const int32_t* offsets, int64_t E, int64_t M_total,
                                  int64_t N2, int64_t K, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldh, int64_t ldact,
                                  void* stream) {
    if (!offsets || E <= 0) return ARIA_ERR_INVALID;
    const int rc = glu_check(A, B, H, ACT, M_total, N2, K, lda, ldb, ldh, ldact);
    if (rc != ARIA_OK) return rc;
    if (strideB & 7) return ARIA_ERR_ALIGN;
    if (2 * M_total * lda >= (1ll << 32) || 2 * K * ldb >= (1ll << 32)) return ARIA_ERR_UNSUPPORTED;
    if (M_total == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A);
    p.B = static_cast<const bf16_t*>(B);
    p.C = H;
    p.C2 = ACT;
    p.lda = lda, p.ldb = ldb, p.ldc = ldh, p.ldc2 = ldact;
    p.M = int(M_total), p.N = int(N2), p.K = int(K);
    p.mode = 1;
    p.offsets = offsets;
    p.E = int(E);
    p.strideB = strideB;
    p.glu = 1;
    return g_last_variant = 3, aria_launch_gemm3(p, 0, 1, int(M_total / 256 + E), stream);
}

int aria_gemm_swiglu_bf16(const void* A, const void* B, void* H, void* ACT, int64_t M, int64_t N2, int64_t K, int64_t lda, int64_t ldb,
                          int64_t ldh, int64_t ldact, void* stream) {
    const int rc = glu_check(A, B, H, ACT, M, N2, K, lda, ldb, ldh, ldact);
    if (rc != ARIA_OK) return rc;
    if (2 * M * lda >= (1ll << 32) || 2 * N2 * ldb >= (1ll << 32)) return ARIA_ERR_UNSUPPORTED;
    if (M == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A);
    p.B = static_cast<const bf16_t*>(B);   // [N2, K] row-major (nn.Linear layout: gate rows, then up rows)
    p.C = H;
    p.C2 = ACT;
    p.lda = lda, p.ldb = ldb, p.ldc = ldh, p.ldc2 = ldact;
    p.M = int(M), p.N = int(N2), p.K = int(K);
    p.mode = 0;
    p.glu = 1;
    return g_last_variant = 3, aria_launch_gemm3(p, 0, 0, int((M + 255) / 256), stream);
}

// gate / up weights as two [.., I, K] tensors of one allocation (gptfast's w1 / w3): rows of the up tensor as an offset from the gate tensor
static int glu_split_rows(const void* Bg, const void* Bu, int64_t I, int64_t ldb, int64_t extra_rows, int* up_rows) {
    const long long diff = static_cast<const char*>(Bu) - static_cast<const char*>(Bg);
    if (diff <= 0 || diff % (2 * ldb)) return ARIA_ERR_UNSUPPORTED;
    const long long rows = diff / (2 * ldb);
    // per-lane DMA offsets are 32-bit and the row index goes through a 24-bit multiply
    if (rows < I || rows + extra_rows + I >= (1ll << 24) || (rows + extra_rows + I) * 2 * ldb >= (1ll << 32)) return ARIA_ERR_UNSUPPORTED;
    *up_rows = int(rows);
    return ARIA_OK;
}

int aria_grouped_gemm_swiglu_split_bf16(const void* A, const void* Bg, const void* Bu, void* H, void* ACT, const int32_t* offsets, int64_t E,
                                        int64_t M_total, int64_t I, int64_t K, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldh,
                                        int64_t ldact, void* stream) {
    if (!offsets || E <= 0 || !Bu) return ARIA_ERR_INVALID;
    int rc = glu_check(A, Bg, H, ACT, M_total, 2 * I, K, lda, ldb, ldh, ldact);
    if (rc != ARIA_OK) return rc;
    if ((strideB & 7) || !aligned16(Bu)) return ARIA_ERR_ALIGN;
    if ((K % 64) || 2 * M_total * lda >= (1ll << 32)) return ARIA_ERR_UNSUPPORTED;
    int up_rows = 0;
    rc = glu_split_rows(Bg, Bu, I, ldb, 0, &up_rows);  // (the expert's base enters the 64-bit operand pointer, not the per-lane offset)
    if (rc != ARIA_OK) return rc;
    if (M_total == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A);
    p.B = static_cast<const bf16_t*>(Bg);
    p.C = H;
    p.C2 = ACT;
    p.lda = lda, p.ldb = ldb, p.ldc = ldh, p.ldc2 = ldact;
    p.M = int(M_total), p.N = int(2 * I), p.K = int(K);
    p.mode = 1;
    p.offsets = offsets;
    p.E = int(E);
    p.strideB = strideB;
    p.glu = 1;
    p.glu_up_rows = up_rows;
    return g_last_variant = 3, aria_launch_gemm3(p, 0, 0, int(M_total / 256 + E), stream);
}

int aria_gemm_swiglu_split_bf16(const void* A, const void* Bg, const void* Bu, void* H, void* ACT, int64_t M, int64_t I, int64_t K, int64_t lda,
                                int64_t ldb, int64_t ldh, int64_t ldact, void* stream) {
    if (!Bu) return ARIA_ERR_INVALID;
    int rc = glu_check(A, Bg, H, ACT, M, 2 * I, K, lda, ldb, ldh, ldact);
    if (rc != ARIA_OK) return rc;
    if (!aligned16(Bu)) return ARIA_ERR_ALIGN;
    if ((K % 64) || 2 * M * lda >= (1ll << 32)) return ARIA_ERR_UNSUPPORTED;
    int up_rows = 0;
    rc = glu_split_rows(Bg, Bu, I, ldb, 0, &up_rows);
    if (rc != ARIA_OK) return rc;
    if (M == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A);
    p.B = static_cast<const bf16_t*>(Bg);
    p.C = H;
    p.C2 = ACT;
    p.lda = lda, p.ldb = ldb, p.ldc = ldh, p.ldc2 = ldact;
    p.M = int(M), p.N = int(2 * I), p.K = int(K);
    p.mode = 0;
    p.glu = 1;
    p.glu_up_rows = up_rows;
    return g_last_variant = 3, aria_launch_gemm3(p, 0, 0, int((M + 255) / 256), stream);
}

// Expert parallelism: grouped launches over the SEGMENTS of an all-to-all's output (ordered (source rank, local expert)): segment g uses
// the weight of local expert g % n_local.  v3 kernels only (ARIA_ERR_UNSUPPORTED otherwise: the caller re-orders and takes the plain entry).
int aria_grouped_gemm_seg_bf16(const void* A, const void* B, void* C, const int32_t* offsets, int64_t n_seg, int64_t n_local, int64_t M_total,
                               int64_t N, int64_t K, int b_oc, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldc, void* stream) {
    if (!A || !B || !C || !offsets || n_seg <= 0 || n_local <= 0 || (n_seg % n_local) || M_total < 0 || N <= 0 || K <= 0) return ARIA_ERR_INVALID;
    if (!aligned16(A) || !aligned16(B) || (lda & 7) || (ldb & 7) || (strideB & 7) || (K & 7) || (b_oc && (N & 7)) || (ldc & 1) ||
        (reinterpret_cast<uintptr_t>(C) & 3))
        return ARIA_ERR_ALIGN;
    if ((K % 64) || K < 64 || 2 * M_total * lda >= (1ll << 32) || 2 * (b_oc ? K : N) * ldb >= (1ll << 32) || 2 * lda >= (1ll << 24) || 2 * ldb >= (1ll << 24) ||
        N < 8)
        return ARIA_ERR_UNSUPPORTED;
    if (M_total == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A);
    p.B = static_cast<const bf16_t*>(B);
    p.C = C;
    p.lda = lda, p.ldb = ldb, p.ldc = ldc;
    p.M = int(M_total), p.N = int(N), p.K = int(K);
    p.mode = 1;
    p.offsets = offsets;
    p.E = int(n_seg);
    p.strideB = strideB;
    p.expert_mod = int(n_local);
    return g_last_variant = 3, aria_launch_gemm3(p, 0, b_oc, int(M_total / 256 + n_seg), stream);
}

int aria_grouped_gemm_swiglu_seg_bf16(const void* A, const void* B, void* H, void* ACT, const int32_t* offsets, int64_t n_seg, int64_t n_local,
                                      int64_t M_total, int64_t N2, int64_t K, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldh, int64_t ldact,
                                      void* stream) {
    if (!offsets || n_seg <= 0 || n_local <= 0 || (n_seg % n_local)) return ARIA_ERR_INVALID;
    const int rc = glu_check(A, B, H, ACT, M_total, N2, K, lda, ldb, ldh, ldact);
    if (rc != ARIA_OK) return rc;
    if (strideB & 7) return ARIA_ERR_ALIGN;
    if ((K % 64) || 2 * M_total * lda >= (1ll << 32) || 2 * K * ldb >= (1ll << 32)) return ARIA_ERR_UNSUPPORTED;
    if (M_total == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A);
    p.B = static_cast<const bf16_t*>(B);
    p.C = H;
    p.C2 = ACT;
    p.lda = lda, p.ldb = ldb, p.ldc = ldh, p.ldc2 = ldact;
    p.M = int(M_total), p.N = int(N2), p.K = int(K);
    p.mode = 1;
    p.offsets = offsets;
    p.E = int(n_seg);
    p.strideB = strideB;
    p.glu = 1;
    p.expert_mod = int(n_local);
    return g_last_variant = 3, aria_launch_gemm3(p, 0, 1, int(M_total / 256 + n_seg), stream);
}

// K2: the two fused SwiGLU launches over grouped rows with the dispatcher's gather folded into the A loader.  X is the un-permuted token
// matrix [T, K]; rows[r] (int32, device, r < M_total) is the token row permuted row r would hold.
int aria_grouped_gemm_swiglu_gather_bf16(const void* X, const int32_t* rows, int64_t T, const void* B, void* H, void* ACT, const int32_t* offsets,
                                         int64_t E, int64_t M_total, int64_t N2, int64_t K, int64_t ldx, int64_t ldb, int64_t strideB, int64_t ldh,
                                         int64_t ldact, void* stream) {
    if (!offsets || !rows || E <= 0 || T <= 0) return ARIA_ERR_INVALID;
    const int rc = glu_check(X, B, H, ACT, M_total, N2, K, ldx, ldb, ldh, ldact);
    if (rc != ARIA_OK) return rc;
    if (strideB & 7) return ARIA_ERR_ALIGN;
    if ((K % 64) || 2 * T * ldx >= (1ll << 32) || T >= (1ll << 24) || 2 * K * ldb >= (1ll << 32)) return ARIA_ERR_UNSUPPORTED;
    if (M_total == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(X);
    p.B = static_cast<const bf16_t*>(B);
    p.C = H;
    p.C2 = ACT;
    p.lda = ldx, p.ldb = ldb, p.ldc = ldh, p.ldc2 = ldact;
    p.M = int(M_total), p.N = int(N2), p.K = int(K);
    p.mode = 1;
    p.offsets = offsets;
    p.E = int(E);
    p.strideB = strideB;
    p.glu = 1;
    p.gather_rows = rows;
    return g_last_variant = 3, aria_launch_gemm3(p, 0, 1, int(M_total / 256 + E), stream);
}

int aria_grouped_gemm_swiglu_split_gather_bf16(const void* X, const int32_t* rows, int64_t T, const void* Bg, const void* Bu, void* H, void* ACT,
                                               const int32_t* offsets, int64_t E, int64_t M_total, int64_t I, int64_t K, int64_t ldx, int64_t ldb,
                                               int64_t strideB, int64_t ldh, int64_t ldact, void* stream) {
    if (!offsets || !rows || E <= 0 || !Bu || T <= 0) return ARIA_ERR_INVALID;
    int rc = glu_check(X, Bg, H, ACT, M_total, 2 * I, K, ldx, ldb, ldh, ldact);
    if (rc != ARIA_OK) return rc;
    if ((strideB & 7) || !aligned16(Bu)) return ARIA_ERR_ALIGN;
    if ((K % 64) || 2 * T * ldx >= (1ll << 32) || T >= (1ll << 24)) return ARIA_ERR_UNSUPPORTED;
    int up_rows = 0;
    rc = glu_split_rows(Bg, Bu, I, ldb, 0, &up_rows);
    if (rc != ARIA_OK) return rc;
    if (M_total == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(X);
    p.B = static_cast<const bf16_t*>(Bg);
    p.C = H;
    p.C2 = ACT;
    p.lda = ldx, p.ldb = ldb, p.ldc = ldh, p.ldc2 = ldact;
    p.M = int(M_total), p.N = int(2 * I), p.K = int(K);
    p.mode = 1;
    p.offsets = offsets;
    p.E = int(E);
    p.strideB = strideB;
    p.glu = 1;
    p.glu_up_rows = up_rows;
    p.gather_rows = rows;
    return g_last_variant = 3, aria_launch_gemm3(p, 0, 0, int(M_total / 256 + E), stream);
}

// K7: wqkv projection + interleaved RoPE + KV-cache write in one launch (gptfast/model.py:413-435, 67-93).
int aria_gemm_qkv_rope_cache_bf16(const void* X, const void* Wqkv, void* Q, void* Kc, void* Vc, const void* freqs_cis, const int32_t* pos,
                                  int64_t M, int64_t D, int64_t K, int64_t hd, int64_t S, int64_t S_cache, int64_t ldx, int64_t ldw, int64_t ldq,
                                  int64_t ld_cache, void* stream) {
    if (!X || !Wqkv || !Q || !Kc || !Vc || !freqs_cis || M < 0 || D <= 0 || K <= 0 || hd <= 0 || S <= 0 || S_cache <= 0) return ARIA_ERR_INVALID;
    if (S > S_cache) return ARIA_ERR_INVALID;  // a sequence longer than the cache (rows t % S would land in the next sequence's slots)
    if (!aligned16(X) || !aligned16(Wqkv) || !aligned16(Q) || !aligned16(Kc) || !aligned16(Vc) || !aligned16(freqs_cis) || (ldx & 7) || (ldw & 7) ||
        (ldq & 7) || (ld_cache & 7))
        return ARIA_ERR_ALIGN;
    if ((D % 256) || (K % 64) || K < 64 || (hd & 7) || (D % hd) || 2 * M * ldx >= (1ll << 32) || 2 * 3 * D * ldw >= (1ll << 32) || 2 * ldx >= (1ll << 24) ||
        2 * ldw >= (1ll << 24))
        return ARIA_ERR_UNSUPPORTED;
    if (M == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(X);
    p.B = static_cast<const bf16_t*>(Wqkv);   // [3 D, K] row-major (q rows, k rows, v rows: gptfast's wqkv)
    p.C = Q;
    p.lda = ldx, p.ldb = ldw, p.ldc = ldq;
    p.M = int(M), p.N = int(3 * D), p.K = int(K);
    p.mode = 0;
    p.rope_fc = static_cast<const bf16_t*>(freqs_cis);
    p.rope_pos = pos;
    p.rope_hd = int(hd), p.rope_D = int(D), p.rope_S = int(S), p.cache_S = int(S_cache);
    p.kc = Kc, p.vc = Vc;
    p.ld_cache = ld_cache;
    return g_last_variant = 3, aria_launch_gemm3(p, 0, 0, int((M + 255) / 256), stream);
}

// The HF layer's q | k | v projection + half-split RoPE in one launch (LlamaAttention.forward, modeling_llama.py:243-281, 130-160)
int aria_gemm_qkv_rope_hf_bf16(const void* X, const void* Wqkv, void* QKV, const void* cos, const void* sin, int64_t M, int64_t D, int64_t K,
                               int64_t hd, int64_t S, int64_t ldx, int64_t ldw, int64_t ldc, void* stream) {
    if (!X || !Wqkv || !QKV || !cos || !sin || M < 0 || D <= 0 || K <= 0 || hd <= 0 || S <= 0) return ARIA_ERR_INVALID;
    if (!aligned16(X) || !aligned16(Wqkv) || !aligned16(QKV) || !aligned16(cos) || !aligned16(sin) || (ldx & 7) || (ldw & 7) || (ldc & 7)) return ARIA_ERR_ALIGN;
    if ((D % 256) || (256 % hd) || (hd & 15) || (K % 64) || K < 64 || 2 * M * ldx >= (1ll << 32) || 2 * 3 * D * ldw >= (1ll << 32) || 2 * ldx >= (1ll << 24) ||
        2 * ldw >= (1ll << 24) || !use_v3(((M + 255) / 256) * (3 * D / 256), K, 2 * M * ldx, 2 * 3 * D * ldw, M, 3 * D))
        return ARIA_ERR_UNSUPPORTED;
    if (M == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(X);
    p.B = static_cast<const bf16_t*>(Wqkv);
    p.C = QKV;
    p.lda = ldx, p.ldb = ldw, p.ldc = ldc;
    p.M = int(M), p.N = int(3 * D), p.K = int(K);
    p.mode = 0;
    p.rope_fc = static_cast<const bf16_t*>(cos);
    p.rope_sn = static_cast<const bf16_t*>(sin);
    p.rope_hd = int(hd), p.rope_D = int(D), p.rope_S = int(S);
    return g_last_variant = 3, aria_launch_gemm3(p, 0, 0, int((M + 255) / 256), stream);
}

// shared validation of the fused input-gradient + SwiGLU-backward entries (gemm3_kernel<.., .., 5>)
static int dglu_check(const void* A, const void* B, const void* H, const void* DH, int64_t M, int64_t I, int64_t K, int64_t lda, int64_t ldb,
                      int64_t ldh, int64_t lddh) {
    if (!A || !B || !H || !DH || M < 0 || I <= 0 || K <= 0) return ARIA_ERR_INVALID;
    if (!aligned16(A) || !aligned16(B) || !aligned16(H) || !aligned16(DH) || (lda & 7) || (ldb & 7) || (ldh & 7) || (lddh & 7)) return ARIA_ERR_ALIGN;
    if ((I % 128) || (K % 64) || K < 64 || 2 * lda >= (1ll << 24) || 2 * ldb >= (1ll << 24)) return ARIA_ERR_UNSUPPORTED;
    return ARIA_OK;
}

int aria_grouped_gemm_dswiglu_bf16(const void* dY, const void* B, const void* H, void* DH, const int32_t* offsets, int64_t E, int64_t M_total,
                                   int64_t I, int64_t K, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldh, int64_t lddh, void* stream) {
    if (!offsets || E <= 0) return ARIA_ERR_INVALID;
    const int rc = dglu_check(dY, B, H, DH, M_total, I, K, lda, ldb, ldh, lddh);
    if (rc != ARIA_OK) return rc;
    if (strideB & 7) return ARIA_ERR_ALIGN;
    if (2 * M_total * lda >= (1ll << 32) || 2 * I * ldb >= (1ll << 32)) return ARIA_ERR_UNSUPPORTED;  // per-lane DMA offsets are 32-bit
    if (M_total == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(dY);
    p.B = static_cast<const bf16_t*>(B);  // [E][I, K]: the "B operand" rows are the I output columns ([N, K] form)
    p.C = DH;
    p.H = static_cast<const bf16_t*>(H);
    p.lda = lda, p.ldb = ldb, p.ldc = lddh, p.ldh = ldh;
    p.M = int(M_total), p.N = int(I), p.K = int(K);
    p.mode = 1;
    p.offsets = offsets;
    p.E = int(E);
    p.strideB = strideB;
    p.dglu = 1;
    return g_last_variant = 3, aria_launch_gemm3(p, 0, 0, int(M_total / 256 + E), stream);
}

int aria_gemm_dswiglu_bf16(const void* dY, const void* B, const void* H, void* DH, int64_t M, int64_t I, int64_t K, int b_oc, int64_t lda,
                           int64_t ldb, int64_t ldh, int64_t lddh, void* stream) {
    const int rc = dglu_check(dY, B, H, DH, M, I, K, lda, ldb, ldh, lddh);
    if (rc != ARIA_OK) return rc;
    if (2 * M * lda >= (1ll << 32) || 2 * (b_oc ? K : I) * ldb >= (1ll << 32)) return ARIA_ERR_UNSUPPORTED;
    if (M == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(dY);
    p.B = static_cast<const bf16_t*>(B);
    p.C = DH;
    p.H = static_cast<const bf16_t*>(H);
    p.lda = lda, p.ldb = ldb, p.ldc = lddh, p.ldh = ldh;
    p.M = int(M), p.N = int(I), p.K = int(K);
    p.mode = 0;
    p.dglu = 1;
    return g_last_variant = 3, aria_launch_gemm3(p, 0, b_oc ? 1 : 0, int((M + 255) / 256), stream);
}

// The weight gradient of experts.fc1 WITHOUT the permuted copy of the tokens (moe_lm.py:326-334: permuted = x.index_select(0, sorted // topk)):
// dW[e] = sum over the expert's permuted rows r of x[rows[r]]^T dY[r].  X [T, K] un-permuted tokens, rows int32 [M_total + 64] (token row per
// permuted row; the 64 entries of padding are read, never used).  v3 only.
int aria_grouped_gemm_wgrad_gather_bf16(const void* X, const int32_t* rows, const void* dY, void* dW, const int32_t* offsets, int64_t E, int64_t T,
                                        int64_t K, int64_t N, int64_t ldx, int64_t ldy, int c_f32, int accumulate, void* stream) {
    // T == 0 (no tokens: every expert's reduction is empty) is a valid call like its sibling aria_grouped_gemm_wgrad_bf16's: dW is
    // zero-filled (or left alone with `accumulate`) -- ADVICE r5
    if (!dW || !offsets || E <= 0 || T < 0 || K < 0 || N < 0) return ARIA_ERR_INVALID;
    if (K == 0 || N == 0) return ARIA_OK;
    if (T == 0) {
        const size_t nbytes = size_t(E) * size_t(K) * size_t(N) * (c_f32 ? 4 : 2);
#ifdef ARIA_EMU
        if (!accumulate) std::memset(dW, 0, nbytes);
#else
        if (!accumulate && hipMemsetAsync(dW, 0, nbytes, static_cast<hipStream_t>(stream)) != hipSuccess) return ARIA_ERR_LAUNCH;
#endif
        return ARIA_OK;
    }
    if (!X || !rows || !dY) return ARIA_ERR_INVALID;
    if (!aligned16(X) || !aligned16(dY) || (ldx & 7) || (ldy & 7) || (K & 7) || (N & 7) || (reinterpret_cast<uintptr_t>(rows) & 3)) return ARIA_ERR_ALIGN;
    if (T >= (1ll << 24) || 2 * ldx >= (1ll << 24) || 2 * T * ldx >= (1ll << 32) || !use_v3(((K + 255) / 256) * ((N + 255) / 256) * E, 64, 0, 0, K, N))
        return ARIA_ERR_UNSUPPORTED;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(X);
    p.B = static_cast<const bf16_t*>(dY);
    p.C = dW;
    p.lda = ldx, p.ldb = ldy, p.ldc = N;
    p.M = int(K), p.N = int(N), p.K = 0;
    p.mode = 2;
    p.offsets = offsets;
    p.E = int(E);
    p.strideC = K * N;
    p.c_f32 = c_f32;
    p.accumulate = accumulate;
    p.gather_rows = rows;
    p.ntn = int((N + BN - 1) / BN);
    return g_last_variant = 3, aria_launch_gemm3(p, 1, 1, int((K + 255) / 256), stream);
}

// ---- LoRA as a K-extension of the base GEMM (GemmParams::ext_k; aria/lora/layers.py:129-139, peft's Linear adapter): v3 only -- shapes the
// 256 x 256 kernels do not take return ARIA_ERR_UNSUPPORTED and the caller runs the adapter as launches of its own
static int lora_ext(GemmParams& p, const void* EA, const void* EB, int64_t ext_k, int64_t ld_ea, int64_t ld_eb, int64_t stride_eb) {
    if (!EA || !EB || ext_k <= 0) return ARIA_ERR_INVALID;
    if (ext_k > 64 || (ext_k & 7) || (p.K % 64) || p.K < 64) return ARIA_ERR_UNSUPPORTED;
    if (!aligned16(EA) || !aligned16(EB) || (ld_ea & 7) || (ld_eb & 7) || (stride_eb & 7)) return ARIA_ERR_ALIGN;
    p.extA = static_cast<const bf16_t*>(EA);
    p.extB = static_cast<const bf16_t*>(EB);
    p.ext_k = int(ext_k);
    p.ld_extA = ld_ea, p.ld_extB = ld_eb, p.stride_extB = stride_eb;
    return ARIA_OK;
}

int aria_gemm_lora_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int b_oc, int64_t lda, int64_t ldb, int64_t ldc,
                        const void* EA, const void* EB, int64_t ext_k, int64_t ld_ea, int64_t ld_eb, void* stream) {
    if (!A || !B || !C || M < 0 || N <= 0 || K <= 0) return ARIA_ERR_INVALID;
    if (!aligned16(A) || !aligned16(B) || (lda & 7) || (ldb & 7) || (K & 7) || (N & 1) || (ldc & 1) || (b_oc && (N & 7))) return ARIA_ERR_ALIGN;
    if (M == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A), p.B = static_cast<const bf16_t*>(B), p.C = C;
    p.lda = lda, p.ldb = ldb, p.ldc = ldc;
    p.M = int(M), p.N = int(N), p.K = int(K);
    p.mode = 0;
    p.ntn = int((N + BN - 1) / BN);
    const int rc = lora_ext(p, EA, EB, ext_k, ld_ea, ld_eb, 0);
    if (rc != ARIA_OK) return rc;
    const long long t256 = ((M + 255) / 256) * ((N + 255) / 256);
    if (!use_v3(t256, K, 2 * M * lda, 2 * (b_oc ? K * ldb : N * ldb), M, N) || 2 * lda >= (1ll << 24) || 2 * ldb >= (1ll << 24))
        return ARIA_ERR_UNSUPPORTED;
    return g_last_variant = 3, aria_launch_gemm3(p, 0, b_oc ? 1 : 0, int((M + 255) / 256), stream);
}

int aria_gemm_swiglu_lora_bf16(const void* A, const void* B, void* H, void* ACT, int64_t M, int64_t N2, int64_t K, int64_t lda, int64_t ldb,
                               int64_t ldh, int64_t ldact, const void* EA, const void* EB, int64_t ext_k, int64_t ld_ea, int64_t ld_eb,
                               void* stream) {
    int rc = glu_check(A, B, H, ACT, M, N2, K, lda, ldb, ldh, ldact);
    if (rc != ARIA_OK) return rc;
    if (2 * M * lda >= (1ll << 32) || 2 * N2 * ldb >= (1ll << 32)) return ARIA_ERR_UNSUPPORTED;
    if (M == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A), p.B = static_cast<const bf16_t*>(B);
    p.C = H, p.C2 = ACT;
    p.lda = lda, p.ldb = ldb, p.ldc = ldh, p.ldc2 = ldact;
    p.M = int(M), p.N = int(N2), p.K = int(K);
    p.mode = 0;
    p.glu = 1;
    rc = lora_ext(p, EA, EB, ext_k, ld_ea, ld_eb, 0);
    if (rc != ARIA_OK) return rc;
    return g_last_variant = 3, aria_launch_gemm3(p, 0, 0, int((M + 255) / 256), stream);
}

int aria_grouped_gemm_lora_bf16(const void* A, const void* B, void* C, const int32_t* offsets, int64_t E, int64_t M_total, int64_t N, int64_t K,
                                int b_oc, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldc, const void* EA, const void* EB, int64_t ext_k,
                                int64_t ld_ea, int64_t ld_eb, int64_t stride_eb, void* stream) {
    if (!A || !B || !C || !offsets || E <= 0 || M_total < 0 || N <= 0) return ARIA_ERR_INVALID;
    if (!aligned16(A) || !aligned16(B) || (lda & 7) || (ldb & 7) || (K & 7) || (N & 7) || (strideB & 7) || (ldc & 1)) return ARIA_ERR_ALIGN;
    if (M_total == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A), p.B = static_cast<const bf16_t*>(B), p.C = C;
    p.lda = lda, p.ldb = ldb, p.ldc = ldc;
    p.M = int(M_total), p.N = int(N), p.K = int(K);
    p.mode = 1;
    p.offsets = offsets;
    p.E = int(E);
    p.strideB = strideB;
    p.ntn = int((N + BN - 1) / BN);
    const int rc = lora_ext(p, EA, EB, ext_k, ld_ea, ld_eb, stride_eb);
    if (rc != ARIA_OK) return rc;
    if (!use_v3((M_total / 256 + 1) * ((N + 255) / 256), K, 2 * M_total * lda, 2 * (b_oc ? K * ldb : N * ldb), M_total, N) ||
        2 * lda >= (1ll << 24) || 2 * ldb >= (1ll << 24))
        return ARIA_ERR_UNSUPPORTED;
    return g_last_variant = 3, aria_launch_gemm3(p, 0, b_oc ? 1 : 0, int(M_total / 256 + E), stream);
}

int aria_grouped_gemm_swiglu_lora_bf16(const void* A, const void* B, void* H, void* ACT, const int32_t* offsets, int64_t E, int64_t M_total,
                                       int64_t N2, int64_t K, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldh, int64_t ldact,
                                       const void* EA, const void* EB, int64_t ext_k, int64_t ld_ea, int64_t ld_eb, int64_t stride_eb,
                                       void* stream) {
    if (!offsets || E <= 0) return ARIA_ERR_INVALID;
    int rc = glu_check(A, B, H, ACT, M_total, N2, K, lda, ldb, ldh, ldact);
    if (rc != ARIA_OK) return rc;
    if (strideB & 7) return ARIA_ERR_ALIGN;
    if (2 * M_total * lda >= (1ll << 32) || 2 * K * ldb >= (1ll << 32)) return ARIA_ERR_UNSUPPORTED;
    if (M_total == 0) return ARIA_OK;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A), p.B = static_cast<const bf16_t*>(B);
    p.C = H, p.C2 = ACT;
    p.lda = lda, p.ldb = ldb, p.ldc = ldh, p.ldc2 = ldact;
    p.M = int(M_total), p.N = int(N2), p.K = int(K);
    p.mode = 1;
    p.offsets = offsets;
    p.E = int(E);
    p.strideB = strideB;
    p.glu = 1;
    rc = lora_ext(p, EA, EB, ext_k, ld_ea, ld_eb, stride_eb);
    if (rc != ARIA_OK) return rc;
    return g_last_variant = 3, aria_launch_gemm3(p, 0, 1, int(M_total / 256 + E), stream);
}

int aria_grouped_gemm_wgrad_bf16(const void* A, const void* dY, void* dW, const int32_t* offsets, int64_t E, int64_t K,
                                 int64_t N, int64_t lda, int64_t ldy, int c_f32, int accumulate, void* stream) {
    if (!A || !dY || !dW || !offsets || E <= 0) return ARIA_ERR_INVALID;
    if (K == 0 || N == 0) return ARIA_OK;
    if (!aligned16(A) || !aligned16(dY) || (lda & 7) || (ldy & 7) || (K & 7) || (N & 7)) return ARIA_ERR_ALIGN;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A);   // "A operand" = A^T: element (feature i, token k) at A[k * lda + i]
    p.B = static_cast<const bf16_t*>(dY);  // "B operand": element (token k, n) at dY[k * ldy + n]
    p.C = dW;
    p.lda = lda;
    p.ldb = ldy;
    p.ldc = N;
    p.M = int(K);
    p.N = int(N);
    p.K = 0;
    p.mode = 2;
    p.offsets = offsets;
    p.E = int(E);
    p.strideC = K * N;
    p.c_f32 = c_f32;
    p.accumulate = accumulate;
    p.ntn = int((N + BN - 1) / BN);
    const int ntm = int((K + BM - 1) / BM);
    // (the reduction length is per expert and data dependent: 64 stands in for "long enough" in the eligibility test)
    // both operands are token-major (output-contiguous): their per-lane DMA offsets stay inside one 64-token tile, no 4 GiB limit
    // Measured on MI355X (98304 routed rows, 64 experts): v3 749 TF/s vs v2 800-830 on these short (~24 K-tile) reductions with two
    // transposed operands -- v3's longer prologue and its 24-read first phase do not pay here, so v2 stays the default.
    const char* force3 = std::getenv("ARIA_GEMM_FORCE");
    // r02: with the wide epilogue v3 is level with v2 here too (in-bench 710.3 vs 714.6 ms/step); ARIA_GEMM_WGRAD_V3=0 goes back to v2
    const char* wg3 = std::getenv("ARIA_GEMM_WGRAD_V3");
    if (((force3 && force3[0] == '3') || !(wg3 && wg3[0] == '0')) && !(force3 && (force3[0] == '1' || force3[0] == '2')) && use_v3(((K + 255) / 256) * ((N + 255) / 256) * E, 64, 0, 0, K, N))
        return g_last_variant = 3, aria_launch_gemm3(p, 1, 1, int((K + 255) / 256), stream);
    if (use_v2(((K + 255) / 256) * ((N + 255) / 256) * E)) return g_last_variant = 2, aria_launch_gemm2(p, 1, 1, int((K + 255) / 256), int(E), stream);
    return g_last_variant = 1, launch_gemm(p, 1, 1, p.ntn * ntm, int(E), stream);
}

}  // extern "C"
