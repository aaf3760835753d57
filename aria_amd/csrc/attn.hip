// Flash attention for gfx950: forward, and backward as two kernels (dK/dV per key tile, dQ per query tile).
//
// Replaces the attention class the reference selects through LLAMA_ATTENTION_CLASSES / config._attn_implementation
// (aria/model/moe_lm.py:594; eager semantics transformers/models/llama/modeling_llama.py:192-215, flash-attn in the
// reference's GPU recipe) and the ViT's bidirectional masked attention (idefics2 eager attention, key padding from
// aria/model/vision_encoder.py:147-152).  No S x S tensor is ever materialised (gptfast's default mask is 4.3 GB at 64K).
//
// Layout ideas (all MFMA 32x32x16 bf16, fp32 accumulate):
//  * Forward / dQ compute the TRANSPOSED score tile S^T = K Q^T, so in the accumulator layout a lane owns ONE query
//    column: the online-softmax row statistics are lane-local (one exchange with lane^32), and the probabilities a
//    lane holds are exactly the 8-per-k-step values the next MFMA wants as its B operand (P^T): no LDS round trip,
//    no cross-lane shuffles.  The output is accumulated transposed as well (O^T = V^T P^T, dQ^T = K^T dS^T), so the
//    per-query rescale stays lane-local.
//  * dK/dV compute S = Q K^T untransposed, so a lane owns one KEY column and P^T / dS^T are lane-local A operands.
//  * Operands that are needed "k-strided" (V^T, K^T, dO, Q as B/A operands along the token axis) are read from the
//    row-major LDS tile through ds_read_b64_tr_b16 (hardware 4x16 transpose).  Any permutation of the reduction index is
//    legal as long as both operands agree, so the k order follows the accumulator layout.
//  * tensors are [B, S, H, hd] views of token-major activations (row stride ld*): no head transposes in HBM.
#include "aria_device.h"
#include "aria_hip.h"
#include <cstdlib>

namespace {
using namespace ad;

constexpr int NT = 256;  // 4 waves

template <int HD>
struct Cfg {
    static constexpr int KS = (HD + 15) / 16;   // k-substeps over the head dim (hd 72: 5 -- the reduction is padded to 80 with zero columns)
    static constexpr int DT = (HD + 31) / 32;   // 32-wide feature tiles (hd 72: 3 -- feature rows 72..95 of the transposed results are discarded)
    static constexpr int HDP = DT * 32 > KS * 16 ? DT * 32 : KS * 16;  // columns a tile row must hold for the fragment reads
    static constexpr bool PADDED = HDP != HD;    // hd 72 (ViT / projector): LDS pad columns HD..HDP-1 are zeroed once, loads and stores are guarded
    static constexpr int PITCH = HDP + 8;  // elements; 16-byte fragments of 16 consecutive rows hit distinct bank slots (pitches 72 / 104 / 136 elements = 36 / 52 / 68 dwords: 16 rows -> 16 different multiples of 4 dwords mod 64)
    static constexpr int CPR = HD / 8;     // 16-byte chunks per row
    static constexpr int NCH64 = (64 * CPR + NT - 1) / NT;  // chunks per thread for a 64-row tile
};

// ---- 64-row tile: global -> registers -> LDS ------------------------------------------------------------
template <int HD>
__device__ __forceinline__ void tile_load(u32x4 (&r)[Cfg<HD>::NCH64], const bf16_t* base, long long ld, int row0, int row_end,
                                          int t) {
#pragma unroll
    for (int p = 0; p < Cfg<HD>::NCH64; ++p) {
        const int c = t + NT * p;
        const int row = row0 + c / Cfg<HD>::CPR, col = (c % Cfg<HD>::CPR) * 8;
        r[p] = (row < row_end && (!Cfg<HD>::PADDED || c < 64 * Cfg<HD>::CPR)) ? ld16(base + (long long)row * ld + col) : zero16();
    }
}
template <int HD>
__device__ __forceinline__ void tile_store(const u32x4 (&r)[Cfg<HD>::NCH64], bf16_t* s, int t) {
#pragma unroll
    for (int p = 0; p < Cfg<HD>::NCH64; ++p) {
        const int c = t + NT * p;
        if (!Cfg<HD>::PADDED || c < 64 * Cfg<HD>::CPR) st16(s + (c / Cfg<HD>::CPR) * Cfg<HD>::PITCH + (c % Cfg<HD>::CPR) * 8, r[p]);
    }
}
// hd 72: zero the pad columns HD..HDP-1 of `ntiles` 64-row tiles once (tile_store never touches them): the reduction's pad must be zero in
// BOTH operands, and the discarded feature rows must not turn into NaN traps for anybody reading the dump
template <int HD>
__device__ __forceinline__ void tile_zero_pads(bf16_t* s, int ntiles, int t) {
    if (Cfg<HD>::PADDED) {
        constexpr int PC = (Cfg<HD>::HDP - HD) / 8;  // 16-byte pad chunks per row
        for (int i = t; i < ntiles * 64 * PC; i += NT) st16(s + (i / PC) * Cfg<HD>::PITCH + HD + (i % PC) * 8, zero16());
    }
}

// rc fragment: row `row` of the tile, reduction indices 16*kk + 8*(l>>5) + 0..7
template <int HD>
__device__ __forceinline__ s16x8 frag_rc(const bf16_t* s, int row, int kk, int l) {
    return *reinterpret_cast<const s16x8*>(s + row * Cfg<HD>::PITCH + kk * 16 + (l >> 5) * 8);
}

// accumulator regs 8u..8u+7 of a 32x32 tile -> bf16 fragment (lane-local)
__device__ __forceinline__ s16x8 pack_frag(const f32x16& p, int u) {
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = pack2bf(p[8 * u + 2 * j], p[8 * u + 2 * j + 1]);
    return __builtin_bit_cast(s16x8, v);
}

__device__ __forceinline__ int acc_row(int r, int l) { return (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); }

// (a0 * b + c, a1 * b + c) as ONE v_pk_fma_f32 (two fp32 lanes per instruction at the plain-VALU rate; each lane is a fused multiply-add,
// the same rounding as two v_fma_f32): the softmax's score scaling is 32 of a key tile's ~100 vector instructions in the forward kernels,
// which are vector-ALU-bound as much as matrix-bound (profiles/r04_attn_fwd_ablate.json)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(float a0, float a1, float b, float c) {
    const f32x2 a = {a0, a1}, bb = {b, b}, cc = {c, c};
    return __builtin_elementwise_fma(a, bb, cc);
}

__device__ __forceinline__ f32x16 zero_acc() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// 1-D grid -> (block along the sequence, head, batch) for the 512-thread kernels.  Workgroup n runs on XCD n % 8 (own L2).
//  * causal: blocks differ 16x in length, so the sequence block is the SLOWEST index and runs longest-first (`reverse` says whether the
//    last or the first block is the longest); all blocks of one (b, head) share an XCD when H*B % 8 == 0.
//  * otherwise: every block of one (b, head) streams the same K/V (or Q/dO), so the sequence block is the FASTEST index inside an
//    XCD's run and the (b, head) pairs are dealt round-robin to the XCDs: the ~32 blocks an XCD runs at a time share 1-2 K/V sets.
// The launch uses attn_grid() workgroups; ids that fall past the last (b, head) pair return false.
__device__ __forceinline__ bool attn_block_coords(int nblk, int H, int B, int causal, bool reverse, int& blk, int& head, int& b) {
    const int n = blockIdx.x, nbh = H * B;
    int bh;
    if (causal == 2) {
        // r05, causal == 2 (the host asks for it: the FORWARD from 32 K tokens up): like the non-causal order -- the blocks of one (b, head) run
        // TOGETHER on one XCD, longest first, so the ~32 workgroups an XCD has resident share a few K / V panels.  Measured on one box
        // (profiles/r05_attn_causal_order_ab.json), this order against the one below: forward 64K 22.76 -> 21.45 ms (-5.7 %), but 16K 1.48 -> 1.60
        // and 8 x 2048 0.312 -> 0.356 ms, backward 0.866 -> 0.95 / 4.86 -> 5.12 / 76.6 -> 76.2 ms: below ~32 K tokens the chip-wide
        // longest-first order's load balance is worth more than the panels' L2 residency, and the backward kernels do not notice the L2 at all
        // (the config #3 step: 585.9 vs 594.0 ms with this order everywhere).  Equal work per XCD needs the number of units to be a multiple of
        // 8: a (b, head) is cut into P = 8 / gcd(H B, 8) interleaved parts (blocks p, p + P, ...), units are dealt round-robin to the XCDs.
        const int g = (nbh & 7) == 0 ? 8 : (nbh & 3) == 0 ? 4 : (nbh & 1) == 0 ? 2 : 1, P = 8 / g;
        const int nbu = (nblk + P - 1) / P;
        const int xcd = n & 7, j = n >> 3;
        const int u = (j / nbu) * 8 + xcd, t = j % nbu;
        bh = u / P;
        const int pos = t * P + (u % P);   // 0 = the longest block
        if (pos >= nblk) return false;
        blk = reverse ? nblk - 1 - pos : pos;
    } else if (causal) {
        const int z = n / nbh;
        if (z >= nblk) return false;
        bh = n % nbh;
        blk = reverse ? nblk - 1 - z : z;
    } else {
        const int xcd = n & 7, j = n >> 3;
        blk = j % nblk;
        bh = (j / nblk) * 8 + xcd;
    }
    if (bh >= nbh) return false;
    head = bh % H;
    b = bh / H;
    return true;
}
inline unsigned attn_grid(long long nblk, long long H, long long B) {
    // covers both orders: (b, head) pairs rounded up to 8 times the blocks, and P parts of ceil(nblk / P) blocks per (b, head) for the causal one
    const long long nbh = H * B, g = (nbh & 7) == 0 ? 8 : (nbh & 3) == 0 ? 4 : (nbh & 1) == 0 ? 2 : 1, P = 8 / g;
    const long long causal_ids = ((nbh * P + 7) / 8) * 8 * ((nblk + P - 1) / P);
    return unsigned(std::max(((nbh + 7) & ~7ll) * nblk, causal_ids));
}

// =========================================================================================== forward v2
// NW waves / 32 NW queries per block (8 for hd 64/128; 12 for hd 72, whose ~165 VGPRs allow three waves per SIMD -- the extra
// wave hides LDS / MFMA-result latency that two waves cannot), K/V double-buffered in LDS (ONE barrier per 64-key tile), V fragments through
// ds_read_b64_tr_b16 (row-major [key][d] tile read as key-contiguous fragments: half the LDS cycles of the dword-pair
// read and no VALU repacking), softmax in the log2 domain on v_exp_f32 with the scale folded in, lazy (wave-uniform)
// rescale, mask arithmetic only on edge / diagonal / padded tiles, native head dims 64 / 72 / 128 (72 = ViT and
// projector: reduction padded to 80, output tiles 32+32+8).
// Timing-only ablations of attn_fwd2_kernel (tools/probes/attn_fwd_ablate.py builds the variants; results are garbage by design):
// 1 no QK MFMAs, 2 no exponentials, 4 no PV MFMAs (nor V reads), 8 V fragments not read (MFMAs stay), 16 K fragments not read, 32 next tile
// neither loaded nor staged, 64 no barrier, 128 no running-maximum update.  0 in the library: every branch below folds away.
#ifndef ARIA_ATTN_ABL
#define ARIA_ATTN_ABL 0
#endif
template <typename T>
__device__ __forceinline__ void abl_keep(const T& v) {
#ifndef ARIA_EMU
    asm volatile("" : : "v"(v));
#endif
}

template <int HD, int NW = 8>
struct Cfg2 {
    static constexpr int NT2 = NW * 64;   // threads per block
    static constexpr int QB = NW * 32;    // queries per block (32 per wave)
    static constexpr int KS = (HD + 15) / 16;
    static constexpr int HDK = KS * 16;
    static constexpr int DT = (HD + 31) / 32;
    static constexpr int CPR = HD / 8;
    static constexpr int KP = HDK + 8;                 // K tile pitch (elements): conflict-free ds_read_b128 fragments
    static constexpr int VP = HD == 128 ? 160 : 96;    // V tile pitch: 16 (mod 64) dwords apart rows -> conflict-free tr reads
    static constexpr int NCH = (64 * CPR + NT2 - 1) / NT2;
    // hd 72: the output tiles cover 96 feature rows, 24 of them padding.  V's pad column HD is set to 1.0, so feature row HD of O^T
    // accumulates sum_k P[k][q] on the matrix pipe -- the softmax denominator for free (33 v_add_f32 per tile less on a VALU-bound
    // kernel), rescaled together with O.  It sums the bf16-rounded P that also multiplies V (self-consistent normalisation).
    static constexpr bool ROWSUM_IN_MFMA = (DT * 32 > HD);
    static constexpr size_t SMEM = size_t(2) * 64 * (KP + VP) * 2 + 2 * 64 + 16;
};

template <int HD, int NW>
using Chunks2 = u32x4[Cfg2<HD, NW>::NCH];

template <int HD, int NW>
__device__ __forceinline__ void tile2_load(Chunks2<HD, NW>& r, const bf16_t* base, long long ld, int row0, int row_end,
                                           int t) {
    using C = Cfg2<HD, NW>;
#pragma unroll
    for (int p = 0; p < C::NCH; ++p) {
        const int c = t + C::NT2 * p;
        const int row = row0 + c / C::CPR, col = (c % C::CPR) * 8;
        r[p] = (c < 64 * C::CPR && row < row_end) ? ld16(base + (long long)row * ld + col) : zero16();
    }
}
template <int HD, int NW, int PITCH>
__device__ __forceinline__ void tile2_store(const Chunks2<HD, NW>& r, bf16_t* s, int t) {
    using C = Cfg2<HD, NW>;
#pragma unroll
    for (int p = 0; p < C::NCH; ++p) {
        const int c = t + C::NT2 * p;
        if (c < 64 * C::CPR) st16(s + (c / C::CPR) * PITCH + (c % C::CPR) * 8, r[p]);
    }
}

template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd2_kernel(const bf16_t* Q, const bf16_t* K, const bf16_t* V, bf16_t* O, float* LSE,
                                                        const int32_t* kv_len, const uint8_t* key_mask, int Sq, int S, int H,
                                                        long long ldq, long long ldk, long long ldv, long long ldo, float scale,
                                                        int causal, int nbatch) {
    using C = Cfg2<HD, NW>;
    ARIA_DYN_SMEM(smem);
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem);          // [2][64][KP]
    bf16_t* sV = sK + 2 * 64 * C::KP;                      // [2][64][VP]
    uint8_t* sM = reinterpret_cast<uint8_t*>(sV + 2 * 64 * C::VP);  // [2][64]
    int* sFlag = reinterpret_cast<int*>(sM + 128);         // [2]
    const int t = threadIdx.x, l = t & 63, w = t >> 6, h2 = l >> 5;
    int qblk, head, b;
    if (!attn_block_coords((Sq + C::QB - 1) / C::QB, H, nbatch, causal, true, qblk, head, b)) return;
    const int q0 = qblk * C::QB;
    const long long tok0 = (long long)b * S, tokq0 = (long long)b * Sq;
    const bf16_t* Qb = Q + tokq0 * ldq + head * HD;
    const bf16_t* Kb = K + tok0 * ldk + head * HD;
    const bf16_t* Vb = V + tok0 * ldv + head * HD;
    const uint8_t* kmb = key_mask ? key_mask + tok0 : nullptr;
    const int q_wmin = q0 + 32 * w, q_abs = q_wmin + (l & 31);
    const int klen = kv_len ? min(S, kv_len[b]) : S;
    const float scale2 = scale * 1.4426950408889634f;

    s16x8 qf[C::KS];
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) {
        u32x4 v = zero16();
        const int col = kk * 16 + h2 * 8;
        if (q_abs < Sq && col < HD) v = ld16(Qb + (long long)q_abs * ldq + col);
        qf[kk] = __builtin_bit_cast(s16x8, v);
    }
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) settle(qf[kk]);  // (see settle(): keeps the wait for Q out of the key-tile loop)
    f32x16 o[C::DT];
#pragma unroll
    for (int i = 0; i < C::DT; ++i) o[i] = zero_acc();
    float m = -INFINITY, lsum = 0.f;

    int kv_end = klen;
    if (causal) kv_end = min(kv_end, q0 + C::QB);
    const int ntiles = (kv_end + 63) / 64;

    // pad columns: K's reduction pad must be zero (Q's is), V's pad only feeds discarded output rows but must not be NaN-free
    // garbage either way -> zero both once (tile stores never touch them)
    if (C::HDK != HD || C::VP > HD) {
        for (int i = t; i < 2 * 64; i += C::NT2) {
            if (C::HDK != HD) st16(sK + i * C::KP + HD, zero16());
            for (int c = HD; c < C::VP; c += 8) st16(sV + i * C::VP + c, zero16());
            if (C::ROWSUM_IN_MFMA) sV[i * C::VP + HD] = 0x3F80;  // bf16 1.0: the "ones" column
        }
    }
    u32x4 rk[C::NCH], rv[C::NCH];
    if (ntiles > 0) {
        tile2_load<HD, NW>(rk, Kb, ldk, 0, S, t);
        tile2_load<HD, NW>(rv, Vb, ldv, 0, S, t);
        tile2_store<HD, NW, C::KP>(rk, sK, t);
        tile2_store<HD, NW, C::VP>(rv, sV, t);
        if (kmb && t < 64) {
            const uint8_t mv = t < S ? kmb[t] : 0;
            sM[t] = mv;
            const unsigned long long all = ballot(mv != 0);
            if (t == 0) sFlag[0] = int(all == ~0ull) | (int(all == 0ull) << 1);  // bit 0: every key of the tile valid, bit 1: none
        }
    }
    for (int it = 0; it < ntiles; ++it) {
        if (!(ARIA_ATTN_ABL & 64)) sync();  // tile `it` complete in buffer it&1; every wave is done reading the other buffer
        const int cur = it & 1, kv0 = it * 64;
        const bool more = it + 1 < ntiles && !(ARIA_ATTN_ABL & 32);
        if (more) {
            tile2_load<HD, NW>(rk, Kb, ldk, kv0 + 64, S, t);
            tile2_load<HD, NW>(rv, Vb, ldv, kv0 + 64, S, t);
        }
        const bf16_t* cK = sK + cur * 64 * C::KP;
        const bf16_t* cV = sV + cur * 64 * C::VP;
        // (a tile without a single valid key -- the padded bottom rows of an image are one contiguous patch range -- contributes exactly
        // nothing: every probability is exp2(-inf) = 0 and the running maximum does not move)
        if (!(causal && kv0 > q_wmin + 31) && !(kmb && (sFlag[cur] & 2))) {  // wave-uniform: this wave has at least one visible key in the tile
            f32x16 st[2];
            st[0] = zero_acc();
            st[1] = zero_acc();
            if (ARIA_ATTN_ABL & 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[i][r] = float((it * 7 + r * 3 + i + l) & 31) * 0.03125f;
            } else {
                // r04: the K fragments of k-step kk + 1 are requested before the two MFMAs of k-step kk are issued -- left to itself the
                // scheduler sank every read to just in front of its MFMA (read, lgkmcnt(0), multiply: ten exposed LDS round trips per tile)
                auto kf = [&](int i, int kk) __attribute__((always_inline)) -> s16x8 {
                    return (ARIA_ATTN_ABL & 16) ? qf[(kk + i) % C::KS]
                                                : *reinterpret_cast<const s16x8*>(cK + (i * 32 + (l & 31)) * C::KP + kk * 16 + h2 * 8);
                };
                s16x8 fr[2][2];
                fr[0][0] = kf(0, 0);
                fr[0][1] = kf(1, 0);
#pragma unroll
                for (int kk = 0; kk < C::KS; ++kk) {
                    if (kk + 1 < C::KS) {
                        fr[(kk + 1) & 1][0] = kf(0, kk + 1);
                        fr[(kk + 1) & 1][1] = kf(1, kk + 1);
                    }
                    sched_fence();
#pragma unroll
                    for (int i = 0; i < 2; ++i) st[i] = mfma32(fr[kk & 1][i], qf[kk], st[i]);
                    sched_fence();
                }
            }
            const bool need_mask = (kv0 + 64 > klen) || (causal && kv0 + 63 > q_wmin) || (kmb && !(sFlag[cur] & 1));
            if (need_mask) {
                const uint8_t* cM = sM + cur * 64;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kvl = i * 32 + acc_row(r, l);
                        const int kv = kv0 + kvl;
                        bool dead = kv >= klen || (causal && kv > q_abs);
                        if (kmb) dead = dead || !cM[kvl];
                        if (dead) st[i][r] = -INFINITY;
                    }
            }
            float mx = st[0][0];
            if (!(ARIA_ATTN_ABL & 128)) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[i][r]);
                mx = fmaxf(mx, shfl_xor(mx, 32));
            } else {
                mx = 8.f;
            }
            const float m_new = fmaxf(m, mx * scale2);
            const float m_safe = m_new == -INFINITY ? 0.f : m_new;
            if (ballot(m_new > m) != 0ull) {  // wave-uniform lazy rescale: exact (alpha == 1 whenever it is skipped)
                const float alpha = exp2_fast(m - m_safe);
                if (!C::ROWSUM_IN_MFMA) lsum *= alpha;
#pragma unroll
                for (int i = 0; i < C::DT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
                m = m_new;
            }
            float ps = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 x = fma2(st[i][r], st[i][r + 1], scale2, -m_safe);
#pragma unroll
                    for (int z = 0; z < 2; ++z) {
                        const float p = (ARIA_ATTN_ABL & 2) ? x[z] : exp2_fast(x[z]);
                        st[i][r + z] = p;
                        if (!C::ROWSUM_IN_MFMA) ps += p;
                    }
                }
            if (!C::ROWSUM_IN_MFMA) lsum += ps;
            // O^T += V^T P^T
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const s16x8 pf = pack_frag(st[i], u);
                    if (ARIA_ATTN_ABL & 4) {
                        abl_keep(pf);
                        continue;
                    }
                    const bf16_t* vrow = cV + (i * 32 + 16 * u + 4 * h2 + ((l & 15) >> 2)) * C::VP + 16 * ((l >> 4) & 1) + 4 * (l & 3);
#pragma unroll
                    for (int dt = 0; dt < C::DT; ++dt) {
                        s16x8 vf;
                        if (ARIA_ATTN_ABL & 8) {
                            vf = qf[dt];
                        } else {
                            const s16x4 a0 = ds_read_tr16(vrow + 32 * dt);
                            const s16x4 a1 = ds_read_tr16(vrow + 8 * C::VP + 32 * dt);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                vf[e] = a0[e];
                                vf[4 + e] = a1[e];
                            }
                        }
                        o[dt] = mfma32(vf, pf, o[dt]);
                    }
                }
        }
        if (more) {  // stage tile it+1 into the other buffer (nobody reads it before the next barrier)
            const int nb = cur ^ 1, kvn = kv0 + 64;
            tile2_store<HD, NW, C::KP>(rk, sK + nb * 64 * C::KP, t);
            tile2_store<HD, NW, C::VP>(rv, sV + nb * 64 * C::VP, t);
            if (kmb && t < 64) {
                const uint8_t mv = (kvn + t < S) ? kmb[kvn + t] : 0;
                sM[nb * 64 + t] = mv;
                const unsigned long long all = ballot(mv != 0);
                if (t == 0) sFlag[nb] = int(all == ~0ull) | (int(all == 0ull) << 1);
            }
        }
    }
    // ROWSUM_IN_MFMA: feature row HD = 32 (DT-1) + 8 sits in accumulator register 4 of the lanes with h2 == 0
    const float ltot = C::ROWSUM_IN_MFMA ? shfl(o[C::DT - 1][4], l & 31) : lsum + shfl_xor(lsum, 32);
    const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
    if (q_abs < Sq) {
        if (h2 == 0 && LSE)
            LSE[((long long)b * H + head) * Sq + q_abs] = (ltot > 0.f) ? (m + log2f(ltot)) * 0.6931471805599453f : -INFINITY;
        bf16_t* orow = O + (tokq0 + q_abs) * ldo + head * HD;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d0 = 32 * dt + 8 * rg + 4 * h2;
                if (d0 < HD) {
                    u32x2 v;
                    v[0] = pack2bf(o[dt][4 * rg] * inv, o[dt][4 * rg + 1] * inv);
                    v[1] = pack2bf(o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv);
                    *reinterpret_cast<u32x2*>(orow + d0) = v;
                }
            }
    }
}

// =========================================================================================== delta = rowsum(O * dO)
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* O, const bf16_t* dO, float* delta, int S, int H, int HD,
                                                         long long ldo, long long lddo, long long nrows) {
    // one 16-lane group per (token, head) row
    const int g = threadIdx.x >> 4, gl = threadIdx.x & 15;
    for (long long base = (long long)blockIdx.x * 16; base < nrows; base += (long long)gridDim.x * 16) {  // block-uniform trip count
        const long long row = base + g;
        const bool valid = row < nrows;
        const long long tok = valid ? row / H : 0;
        const int head = valid ? int(row % H) : 0;
        float acc = 0.f;
        if (valid)
            for (int c = gl; c < HD / 8; c += 16) {
                const u32x4 a = ld16(O + tok * ldo + head * HD + c * 8);
                const u32x4 g2 = ld16(dO + tok * lddo + head * HD + c * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += bflo(a[q]) * bflo(g2[q]) + bfhi(a[q]) * bfhi(g2[q]);
            }
#pragma unroll
        for (int msk = 8; msk >= 1; msk >>= 1) acc += shfl_xor(acc, msk);
        if (valid && gl == 0) {
            const long long bidx = tok / S, s = tok % S;
            delta[(bidx * H + head) * S + s] = acc;
        }
    }
}

// =========================================================================================== backward v2
// Two kernels (dK/dV per key tile, dQ per query tile): K/V (dQ kernel) or Q/dO (dK/dV kernel) tiles double-buffered in LDS with
// one barrier per tile, token-strided operands through ds_read_b64_tr_b16 (natural feature order, no VALU repacking), mask
// arithmetic only on edge / diagonal / padded tiles, probabilities as exp2(s * scale*log2e - lse*log2e).

// fragment of 32 consecutive feature columns d0..d0+31 whose 8 k-slots are token rows rb + 4h + (e&3) + 8(e>>2)
template <int HD>
__device__ __forceinline__ s16x8 frag_tr(const bf16_t* s, int rb, int d0, int l) {
    const bf16_t* p = s + (rb + 4 * (l >> 5) + ((l & 15) >> 2)) * Cfg<HD>::PITCH + d0 + 16 * ((l >> 4) & 1) + 4 * (l & 3);
    const s16x4 a0 = ds_read_tr16(p);
    const s16x4 a1 = ds_read_tr16(p + 8 * Cfg<HD>::PITCH);
    s16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[e] = a0[e];
        f[4 + e] = a1[e];
    }
    return f;
}

template <int HD>
__global__ __launch_bounds__(NT) void attn_bwd2_dkdv_kernel(const bf16_t* Q, const bf16_t* K, const bf16_t* V, const bf16_t* dO,
                                                            const float* LSE, const float* DELTA, bf16_t* dK, bf16_t* dV,
                                                            const int32_t* kv_len, const uint8_t* key_mask, int Sq, int S, int H,
                                                            long long ldq, long long ldk, long long ldv, long long lddo,
                                                            long long lddk, long long lddv, float scale, int causal) {
    using C = Cfg<HD>;
    constexpr int DT = C::DT;
    ARIA_DYN_SMEM(smem);
    bf16_t* sQ = reinterpret_cast<bf16_t*>(smem);       // [2][64][PITCH]
    bf16_t* sdO = sQ + 2 * 64 * C::PITCH;               // [2][64][PITCH]
    float* sLse = reinterpret_cast<float*>(sdO + 2 * 64 * C::PITCH);  // [2][64]  (lse * log2e)
    float* sDel = sLse + 128;                           // [2][64]
    const int t = threadIdx.x, l = t & 63, w = t >> 6, h2 = l >> 5;
    const int b = blockIdx.z, head = blockIdx.y, kv0 = blockIdx.x * 128;
    const long long tok0 = (long long)b * S, tokq0 = (long long)b * Sq;
    const bf16_t* Qb = Q + tokq0 * ldq + head * HD;
    const bf16_t* Kb = K + tok0 * ldk + head * HD;
    const bf16_t* Vb = V + tok0 * ldv + head * HD;
    const bf16_t* dOb = dO + tokq0 * lddo + head * HD;
    const float* lseb = LSE + ((long long)b * H + head) * Sq;
    const float* delb = DELTA + ((long long)b * H + head) * Sq;
    const int kv_wmin = kv0 + 32 * w, kv_abs = kv_wmin + (l & 31);
    const int klen = kv_len ? min(S, kv_len[b]) : S;
    const bool key_ok = kv_abs < klen && (!key_mask || key_mask[tok0 + kv_abs] != 0);
    const bool all_keys_ok = ballot(key_ok) == ~0ull;
    const float scale2 = scale * 1.4426950408889634f;

    s16x8 kf[C::KS], vf[C::KS];
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) {
        u32x4 a = zero16(), c = zero16();
        if (kv_abs < S && (!C::PADDED || kk * 16 + h2 * 8 < HD)) {
            a = ld16(Kb + (long long)kv_abs * ldk + kk * 16 + h2 * 8);
            c = ld16(Vb + (long long)kv_abs * ldv + kk * 16 + h2 * 8);
        }
        kf[kk] = __builtin_bit_cast(s16x8, a);
        vf[kk] = __builtin_bit_cast(s16x8, c);
    }
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) {
        settle(kf[kk]);
        settle(vf[kk]);
    }
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) {
        dk[i] = zero_acc();
        dv[i] = zero_acc();
    }
    const int q_begin = causal ? (kv0 / 64) * 64 : 0;
    const int ntiles = kv0 < klen ? (Sq - q_begin + 63) / 64 : 0;
    u32x4 rq[C::NCH64], rdo[C::NCH64];
    tile_zero_pads<HD>(sQ, 4, t);  // (sQ and sdO are contiguous: 4 tiles)
    if (ntiles > 0) {
        tile_load<HD>(rq, Qb, ldq, q_begin, Sq, t);
        tile_load<HD>(rdo, dOb, lddo, q_begin, Sq, t);
        tile_store<HD>(rq, sQ, t);
        tile_store<HD>(rdo, sdO, t);
        if (t < 64) {
            const int q = q_begin + t;
            sLse[t] = q < Sq ? lseb[q] * 1.4426950408889634f : 0.f;
            sDel[t] = q < Sq ? delb[q] : 0.f;
        }
    }
    for (int it = 0; it < ntiles; ++it) {
        sync();
        const int cur = it & 1, qt0 = q_begin + it * 64;
        const bool more = it + 1 < ntiles;
        if (more) {
            tile_load<HD>(rq, Qb, ldq, qt0 + 64, Sq, t);
            tile_load<HD>(rdo, dOb, lddo, qt0 + 64, Sq, t);
        }
        const bf16_t* cQ = sQ + cur * 64 * C::PITCH;
        const bf16_t* cdO = sdO + cur * 64 * C::PITCH;
        const float* cL = sLse + cur * 64;
        const float* cD = sDel + cur * 64;
        if (!(causal && kv_wmin > qt0 + 63)) {  // wave-uniform: some query of the tile can see some key of this wave
            f32x16 sc[2], dp[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                sc[i] = zero_acc();
                dp[i] = zero_acc();
            }
#pragma unroll
            for (int kk = 0; kk < C::KS; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    sc[i] = mfma32(frag_rc<HD>(cQ, i * 32 + (l & 31), kk, l), kf[kk], sc[i]);
                    dp[i] = mfma32(frag_rc<HD>(cdO, i * 32 + (l & 31), kk, l), vf[kk], dp[i]);
                }
            const bool need_mask = !all_keys_ok || (qt0 + 64 > Sq) || (causal && kv_wmin + 31 > qt0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = i * 32 + acc_row(r, l);
                    float p = exp2_fast(sc[i][r] * scale2 - cL[ql]);
                    if (need_mask) {
                        const int q = qt0 + ql;
                        if (!(q < Sq && key_ok && !(causal && kv_abs > q))) p = 0.f;
                    }
                    sc[i][r] = p;
                    dp[i][r] = p * (dp[i][r] - cD[ql]) * scale;
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const s16x8 pf = pack_frag(sc[i], u);
                    const s16x8 dsf = pack_frag(dp[i], u);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        dv[dt] = mfma32(pf, frag_tr<HD>(cdO, i * 32 + 16 * u, 32 * dt, l), dv[dt]);
                        dk[dt] = mfma32(dsf, frag_tr<HD>(cQ, i * 32 + 16 * u, 32 * dt, l), dk[dt]);
                    }
                }
        }
        if (more) {
            const int nb = cur ^ 1;
            tile_store<HD>(rq, sQ + nb * 64 * C::PITCH, t);
            tile_store<HD>(rdo, sdO + nb * 64 * C::PITCH, t);
            if (t < 64) {
                const int q = qt0 + 64 + t;
                sLse[nb * 64 + t] = q < Sq ? lseb[q] * 1.4426950408889634f : 0.f;
                sDel[nb * 64 + t] = q < Sq ? delb[q] : 0.f;
            }
        }
    }
    // accumulators: rows = keys kv_wmin + acc_row, cols = features 32 dt + (l & 31)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kv = kv_wmin + acc_row(r, l);
            if (kv >= S) continue;
            const int d = 32 * dt + (l & 31);
            if (C::PADDED && d >= HD) continue;
            dK[(tok0 + kv) * lddk + head * HD + d] = f2bf(dk[dt][r]);
            dV[(tok0 + kv) * lddv + head * HD + d] = f2bf(dv[dt][r]);
        }
}

template <int HD>
__global__ __launch_bounds__(NT) void attn_bwd2_dq_kernel(const bf16_t* Q, const bf16_t* K, const bf16_t* V, const bf16_t* dO,
                                                          const float* LSE, const float* DELTA, bf16_t* dQ, const int32_t* kv_len,
                                                          const uint8_t* key_mask, int Sq, int S, int H, long long ldq,
                                                          long long ldk, long long ldv, long long lddo, long long lddq,
                                                          float scale, int causal) {
    using C = Cfg<HD>;
    constexpr int DT = C::DT;
    ARIA_DYN_SMEM(smem);
    bf16_t* sK = reinterpret_cast<bf16_t*>(smem);   // [2][64][PITCH]
    bf16_t* sV = sK + 2 * 64 * C::PITCH;            // [2][64][PITCH]
    uint8_t* sM = reinterpret_cast<uint8_t*>(sV + 2 * 64 * C::PITCH);  // [2][64]
    int* sFlag = reinterpret_cast<int*>(sM + 128);
    const int t = threadIdx.x, l = t & 63, w = t >> 6, h2 = l >> 5;
    const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 128;
    const long long tok0 = (long long)b * S, tokq0 = (long long)b * Sq;
    const bf16_t* Qb = Q + tokq0 * ldq + head * HD;
    const bf16_t* Kb = K + tok0 * ldk + head * HD;
    const bf16_t* Vb = V + tok0 * ldv + head * HD;
    const bf16_t* dOb = dO + tokq0 * lddo + head * HD;
    const uint8_t* kmb = key_mask ? key_mask + tok0 : nullptr;
    const int q_wmin = q0 + 32 * w, q_abs = q_wmin + (l & 31);
    const int klen = kv_len ? min(S, kv_len[b]) : S;
    const float scale2 = scale * 1.4426950408889634f;
    s16x8 qf[C::KS], dof[C::KS];
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) {
        u32x4 a = zero16(), c = zero16();
        if (q_abs < Sq && (!C::PADDED || kk * 16 + h2 * 8 < HD)) {
            a = ld16(Qb + (long long)q_abs * ldq + kk * 16 + h2 * 8);
            c = ld16(dOb + (long long)q_abs * lddo + kk * 16 + h2 * 8);
        }
        qf[kk] = __builtin_bit_cast(s16x8, a);
        dof[kk] = __builtin_bit_cast(s16x8, c);
    }
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) {
        settle(qf[kk]);
        settle(dof[kk]);
    }
    float lse2 = 0.f, del = 0.f;
    if (q_abs < Sq) {
        lse2 = LSE[((long long)b * H + head) * Sq + q_abs] * 1.4426950408889634f;
        del = DELTA[((long long)b * H + head) * Sq + q_abs];
    }
    settle(lse2);
    settle(del);
    const bool all_q_ok = q_wmin + 31 < Sq;
    f32x16 dq[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) dq[i] = zero_acc();
    int kv_end = klen;
    if (causal) kv_end = min(kv_end, q0 + 128);
    const int ntiles = (kv_end + 63) / 64;
    u32x4 rk[C::NCH64], rv[C::NCH64];
    tile_zero_pads<HD>(sK, 4, t);  // (sK and sV are contiguous: 4 tiles)
    if (ntiles > 0) {
        tile_load<HD>(rk, Kb, ldk, 0, S, t);
        tile_load<HD>(rv, Vb, ldv, 0, S, t);
        tile_store<HD>(rk, sK, t);
        tile_store<HD>(rv, sV, t);
        if (kmb && t < 64) {
            const uint8_t mv = t < S ? kmb[t] : 0;
            sM[t] = mv;
            const unsigned long long all = ballot(mv != 0);
            if (t == 0) sFlag[0] = int(all == ~0ull) | (int(all == 0ull) << 1);  // bit 0: every key of the tile valid, bit 1: none
        }
    }
    for (int it = 0; it < ntiles; ++it) {
        sync();
        const int cur = it & 1, kv0 = it * 64;
        const bool more = it + 1 < ntiles;
        if (more) {
            tile_load<HD>(rk, Kb, ldk, kv0 + 64, S, t);
            tile_load<HD>(rv, Vb, ldv, kv0 + 64, S, t);
        }
        const bf16_t* cK = sK + cur * 64 * C::PITCH;
        const bf16_t* cV = sV + cur * 64 * C::PITCH;
        if (!(causal && kv0 > q_wmin + 31) && !(kmb && (sFlag[cur] & 2))) {
            f32x16 st[2], dpt[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                st[i] = zero_acc();
                dpt[i] = zero_acc();
            }
#pragma unroll
            for (int kk = 0; kk < C::KS; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    st[i] = mfma32(frag_rc<HD>(cK, i * 32 + (l & 31), kk, l), qf[kk], st[i]);
                    dpt[i] = mfma32(frag_rc<HD>(cV, i * 32 + (l & 31), kk, l), dof[kk], dpt[i]);
                }
            const bool need_mask = !all_q_ok || (kv0 + 64 > klen) || (causal && kv0 + 63 > q_wmin) || (kmb && !(sFlag[cur] & 1));
            const uint8_t* cM = sM + cur * 64;
            // (r04: the mask test outside the element loops -- as one loop the compiler kept a scalar branch per element, ~100 per tile)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[i][r] = exp2_fast(st[i][r] * scale2 - lse2);
            if (need_mask) {
                int ll = l;
                hold(ll);  // (keeps the 32 row indices from being hoisted out of the tile loop into 32 VGPRs)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kvl = i * 32 + acc_row(r, ll);
                        const int kv = kv0 + kvl;
                        bool ok = q_abs < Sq && kv < klen && !(causal && kv > q_abs);
                        if (kmb) ok = ok && cM[kvl];
                        if (!ok) st[i][r] = 0.f;
                    }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) dpt[i][r] = st[i][r] * (dpt[i][r] - del) * scale;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const s16x8 dsf = pack_frag(dpt[i], u);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) dq[dt] = mfma32(frag_tr<HD>(cK, i * 32 + 16 * u, 32 * dt, l), dsf, dq[dt]);
                }
        }
        if (more) {
            const int nb = cur ^ 1, kvn = kv0 + 64;
            tile_store<HD>(rk, sK + nb * 64 * C::PITCH, t);
            tile_store<HD>(rv, sV + nb * 64 * C::PITCH, t);
            if (kmb && t < 64) {
                const uint8_t mv = (kvn + t < S) ? kmb[kvn + t] : 0;
                sM[nb * 64 + t] = mv;
                const unsigned long long all = ballot(mv != 0);
                if (t == 0) sFlag[nb] = int(all == ~0ull) | (int(all == 0ull) << 1);
            }
        }
    }
    if (q_abs < Sq) {
        bf16_t* row = dQ + (tokq0 + q_abs) * lddq + head * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d0 = 32 * dt + 8 * rg + 4 * h2;
                if (C::PADDED && d0 >= HD) continue;
                u32x2 v;
                v[0] = pack2bf(dq[dt][4 * rg], dq[dt][4 * rg + 1]);
                v[1] = pack2bf(dq[dt][4 * rg + 2], dq[dt][4 * rg + 3]);
                *reinterpret_cast<u32x2*>(row + d0) = v;
            }
    }
}

// =========================================================================================== backward v3 (hd = 128)
// bwd2 needs 336-472 VGPRs per wave (dK and dV accumulators + K and V fragments + two score tiles), i.e. ONE wave per SIMD, and a
// lone wave cannot hide its own LDS latency, softmax VALU work and MFMA dependency chains: measured 8 % MFMA utilisation.
// v3 splits the dK/dV work of a 32-key group between TWO waves that share a SIMD (w and w + 4):
//
//     role A (waves 0-3):  S = Q K^T  ->  P = exp2(S*c - lse)  -> publishes P (fp32, lane-linear) in LDS ->  dV += P^T dO
//     role B (waves 4-7):  dP = dO V^T            .. barrier ..  reads P -> dS = P (dP - delta) scale   ->  dK += dS^T Q
//
// Both roles run the SAME instruction stream on different operands (first GEMM: own fragment x rc tile; second GEMM: packed
// score tile x transposed tile), 16 + 16 MFMAs each per 64-query tile, no redundant GEMM, < 256 VGPRs -> two waves per SIMD,
// and A's softmax VALU work overlaps B's MFMAs.  Q / dO tiles are stored UNPADDED ([64][128] bf16, 256-byte rows) with the
// 16-byte chunk index XOR-swizzled by f(row) = ((row & 3) << 2) | ((row >> 2) & 3): conflict-free both for the ds_read_b128
// row fragments (16 lanes = 16 rows with distinct row & 15 -> 16 distinct chunks) and for the ds_read_b64_tr_b16 transposed
// fragments (4 rows x 64 bytes -> four different 64-byte bank quarters); bwd2's pitch-136 tiles were 4-way conflicted there.
template <int HD>
struct Cfg3 {
    static constexpr int ROW = HD * 2;                 // bytes per row
    static constexpr int TILE = 64 * ROW;              // bytes per 64-row tile
    static constexpr int CPR = HD / 8;                 // 16-byte chunks per row
    static constexpr int NCH = 64 * CPR / 512;         // chunks per thread per tile (512 threads)
    static constexpr int KS = HD / 16, DT = HD / 32;
    static constexpr int SMEM_DKDV = 4 * TILE + 2 * 2 * 64 * 4 + 4 * 8 * 1024;  // Q,dO double-buffered + lse,delta + P exchange
};
__device__ __forceinline__ int swz3(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int HD>
__device__ __forceinline__ void tile_load3(u32x4 (&r)[Cfg3<HD>::NCH], const bf16_t* base, long long ld, int row0, int row_end, int t) {
#pragma unroll
    for (int p = 0; p < Cfg3<HD>::NCH; ++p) {
        const int c = t + 512 * p;
        const int row = row0 + c / Cfg3<HD>::CPR, col = (c % Cfg3<HD>::CPR) * 8;
        r[p] = row < row_end ? ld16(base + (long long)row * ld + col) : zero16();
    }
}
template <int HD>
__device__ __forceinline__ void tile_store3(const u32x4 (&r)[Cfg3<HD>::NCH], char* s, int t) {
#pragma unroll
    for (int p = 0; p < Cfg3<HD>::NCH; ++p) {
        const int c = t + 512 * p;
        const int row = c / Cfg3<HD>::CPR, ch = c % Cfg3<HD>::CPR;
        st16(s + row * Cfg3<HD>::ROW + ((ch ^ swz3(row)) << 4), r[p]);
    }
}
// the same 64-row tile image written by LDS-DMA (no staging registers, no ds_write pass): the wave's piece pc = w + 8 s is the 1 KiB at
// pc * 1024, i.e. rows 4 pc + (l >> 4), image chunk l & 15 -- which holds SOURCE chunk (l & 15) ^ swz3(row), and swz3(4 pc + (l >> 4)) =
// ((l >> 4) << 2) | (w & 3) does not depend on s.  Rows past row_last are clamped (finite data: whoever reads them masks the result).
template <int HD>
__device__ __forceinline__ void tile_dma3(const bf16_t* base, int ld, int row0, int row_last, char* s, int w, int l) {
    static_assert(Cfg3<HD>::CPR == 16, "one wave piece = 4 rows of 16 chunks");
    const int ch = (l & 15) ^ (((l >> 4) << 2) | (w & 3));
#pragma unroll
    for (int si = 0; si < 2; ++si) {
        const int pc = w + 8 * si;
        const int row = min(row0 + 4 * pc + (l >> 4), row_last);
        glds16_raw(base + (long long)row * ld + ch * 8, s + pc * 1024);
    }
}
// row fragment: row `row`, reduction indices 16 kk + 8 (l >> 5) + 0..7
template <int HD>
__device__ __forceinline__ s16x8 frag_rc3(const char* s, int row, int kk, int l) {
    return *reinterpret_cast<const s16x8*>(s + row * Cfg3<HD>::ROW + (((2 * kk + (l >> 5)) ^ swz3(row)) << 4));
}
// transposed fragment: feature columns d0 .. d0+31 (lane l & 31), k-slots = token rows rb + 4h + (e & 3) + 8 (e >> 2); rb % 16 == 0
template <int HD>
__device__ __forceinline__ s16x8 frag_tr3(const char* s, int rb, int d0, int l) {
    const int row = rb + 4 * (l >> 5) + ((l & 15) >> 2);
    const int col = d0 + 16 * ((l >> 4) & 1) + 4 * (l & 3);
    const int ch = col >> 3, within = (col & 7) * 2;
    const s16x4 a0 = ds_read_tr16(reinterpret_cast<const bf16_t*>(s + row * Cfg3<HD>::ROW + ((ch ^ swz3(row)) << 4) + within));
    const s16x4 a1 = ds_read_tr16(reinterpret_cast<const bf16_t*>(s + (row + 8) * Cfg3<HD>::ROW + ((ch ^ swz3(row + 8)) << 4) + within));
    s16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[e] = a0[e];
        f[4 + e] = a1[e];
    }
    return f;
}

// Timing-only ablations of attn_bwd3_dkdv_kernel (tools/probes/build_attn_abl.sh with -DARIA_DKDV_ABL; results are garbage by design):
// 1 no first GEMM (S / dP), 2 no exponentials, 4 no second GEMM (dV / dK), 8 no P exchange through LDS, 16 no mid-tile barrier,
// 32 next tile not staged, 64 no tile barrier, 128 role B's dS arithmetic off.
#ifndef ARIA_DKDV_ABL
#define ARIA_DKDV_ABL 0
#endif
// grid (H, B, ceil(S/128)), 512 threads: wave pair (g, g + 4) owns keys kv0 + 32 g + (l & 31); loop over 64-query tiles
template <int HD>
__global__ __launch_bounds__(512) void attn_bwd3_dkdv_kernel(const bf16_t* Q, const bf16_t* K, const bf16_t* V, const bf16_t* dO,
                                                             const float* LSE, const float* DELTA, bf16_t* dK, bf16_t* dV,
                                                             const int32_t* kv_len, const uint8_t* key_mask, int Sq, int S, int H,
                                                             long long ldq, long long ldk, long long ldv, long long lddo,
                                                             long long lddk, long long lddv, float scale, int causal, int nbatch,
                                                             const bf16_t* rope_cs, const bf16_t* rope_sn, int rope_S) {
    using C = Cfg3<HD>;
    ARIA_DYN_SMEM(smem);
    char* sQ = smem;                                   // [2] tiles
    char* sdO = smem + 2 * C::TILE;                    // [2] tiles
    float* sLse = reinterpret_cast<float*>(smem + 4 * C::TILE);  // [2][64] (lse * log2e)
    float* sDel = sLse + 128;                          // [2][64]
    char* sP = reinterpret_cast<char*>(sDel + 128);    // [4 pairs][8][64 lanes] x 16 bytes
    const int t = threadIdx.x, l = t & 63, w = first_lane(t >> 6), role = w >> 2, g = w & 3, h2 = l >> 5;
    int kblk, head, b;
    if (!attn_block_coords((S + 127) / 128, H, nbatch, causal, false, kblk, head, b)) return;  // key block 0 is the longest
    const int kv0 = kblk * 128;
    const long long tok0 = (long long)b * S, tokq0 = (long long)b * Sq;
    const bf16_t* Qb = Q + tokq0 * ldq + head * HD;
    const bf16_t* dOb = dO + tokq0 * lddo + head * HD;
    const float* lseb = LSE + ((long long)b * H + head) * Sq;
    const float* delb = DELTA + ((long long)b * H + head) * Sq;
    const int kv_wmin = kv0 + 32 * g, kv_abs = kv_wmin + (l & 31);
    const int klen = kv_len ? min(S, kv_len[b]) : S;
    const bool key_ok = kv_abs < klen && (!key_mask || key_mask[tok0 + kv_abs] != 0);
    const bool all_keys_ok = ballot(key_ok) == ~0ull;
    const float scale2 = scale * 1.4426950408889634f;

    // own fragment: K rows (role A) or V rows (role B) of the wave's 32 keys
    const bf16_t* own = (role ? V + tok0 * ldv : K + tok0 * ldk) + head * HD;
    const long long ldown = role ? ldv : ldk;
    s16x8 of[C::KS];
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) {
        u32x4 a = zero16();
        if (kv_abs < S) a = ld16(own + (long long)kv_abs * ldown + kk * 16 + h2 * 8);
        of[kk] = __builtin_bit_cast(s16x8, a);
    }
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) settle(of[kk]);
    f32x16 acc[C::DT];  // dV (role A) / dK (role B): rows = keys, cols = features
#pragma unroll
    for (int i = 0; i < C::DT; ++i) acc[i] = zero_acc();
    char* myP = sP + g * 8192 + l * 16;

    const int q_begin = causal ? (kv0 / 64) * 64 : 0;
    const int ntiles = kv0 < klen ? (Sq - q_begin + 63) / 64 : 0;
    u32x4 rq[C::NCH], rdo[C::NCH];
    if (ntiles > 0) {
        tile_load3<HD>(rq, Qb, ldq, q_begin, Sq, t);
        tile_load3<HD>(rdo, dOb, lddo, q_begin, Sq, t);
        tile_store3<HD>(rq, sQ, t);
        tile_store3<HD>(rdo, sdO, t);
        if (t < 64) {
            const int q = q_begin + t;
            sLse[t] = q < Sq ? lseb[q] * 1.4426950408889634f : 0.f;
            sDel[t] = q < Sq ? delb[q] : 0.f;
        }
    }
    for (int it = 0; it < ntiles; ++it) {
        wait_vm<0>();  // this wave's LDS-DMA pieces of tile `it` have landed
        if (!(ARIA_DKDV_ABL & 64)) sync();  // tile `it` is complete in buffer it & 1; the other buffer and the P exchange are free
        const int cur = it & 1, qt0 = q_begin + it * 64;
        const bool more = it + 1 < ntiles && !(ARIA_DKDV_ABL & 32);
        // next tile straight into the other buffer by LDS-DMA (its last readers finished before the barrier): no staging registers, no
        // ds_write pass; issued as the kernel's own instruction so that the compiler does not drain it in front of the next LDS read
        float lse_n = 0.f, del_n = 0.f;
        if (more) {
            tile_dma3<HD>(Qb, int(ldq), qt0 + 64, Sq - 1, sQ + (cur ^ 1) * C::TILE, w, l);
            tile_dma3<HD>(dOb, int(lddo), qt0 + 64, Sq - 1, sdO + (cur ^ 1) * C::TILE, w, l);
            if (t < 64) {  // the tile's statistics: fetched now, parked at the end of the iteration (a load issued there holds wave 0, and
                const int qn = min(qt0 + 64 + t, Sq - 1);  // with it the barrier, for a memory round trip)
                lse_n = lseb[qn];
                del_n = delb[qn];
            }
        }
        const char* cQ = sQ + cur * C::TILE;
        const char* cdO = sdO + cur * C::TILE;
        const char* first = role ? cdO : cQ;   // rc operand of the first GEMM
        const char* second = role ? cQ : cdO;  // transposed operand of the second GEMM
        const bool active = !(causal && kv_wmin > qt0 + 63);  // wave-uniform: some query of the tile can see some key of this pair
        f32x16 sc[2];
        if (active) {
            sc[0] = zero_acc();
            sc[1] = zero_acc();
            if (ARIA_DKDV_ABL & 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[i][r] = float((it * 5 + r * 3 + i + l) & 31) * 0.03125f;
            } else {
                // r04: the row fragments of k-step kk + 1 are requested BEFORE the two MFMAs of k-step kk are issued (the scheduler, short of
                // registers, had sunk every read to just in front of its MFMA: read, wait out the LDS latency, multiply -- sixteen times)
                s16x8 fr[2][2];
                fr[0][0] = frag_rc3<HD>(first, (l & 31), 0, l);
                fr[0][1] = frag_rc3<HD>(first, 32 + (l & 31), 0, l);
#pragma unroll
                for (int kk = 0; kk < C::KS; ++kk) {
                    if (kk + 1 < C::KS) {
                        fr[(kk + 1) & 1][0] = frag_rc3<HD>(first, (l & 31), kk + 1, l);
                        fr[(kk + 1) & 1][1] = frag_rc3<HD>(first, 32 + (l & 31), kk + 1, l);
                    }
                    sched_fence();
#pragma unroll
                    for (int i = 0; i < 2; ++i) sc[i] = mfma32(fr[kk & 1][i], of[kk], sc[i]);
                    sched_fence();
                }
            }
            if (role == 0) {
                const float* cL = sLse + cur * 64;
                const bool need_mask = !all_keys_ok || (qt0 + 64 > Sq) || (causal && kv_wmin + 31 > qt0);
                // r04: the tile's 32 log-sum-exp values first, all eight reads in flight together (the loop used to fetch one f32x4, wait,
                // use it, fetch the next: eight exposed LDS round trips per tile), and the mask test OUTSIDE the element loop (the compiler
                // had turned the wave-uniform `if (need_mask)` into one scalar branch per element: 32 per tile)
                f32x4 ls[2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) ls[i][rg] = *reinterpret_cast<const f32x4*>(cL + i * 32 + 8 * rg + 4 * h2);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sc[i][r] = (ARIA_DKDV_ABL & 2) ? sc[i][r] * scale2 - ls[i][r >> 2][r & 3] : exp2_fast(sc[i][r] * scale2 - ls[i][r >> 2][r & 3]);
                if (need_mask) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int q = qt0 + i * 32 + 8 * (r >> 2) + 4 * h2 + (r & 3);
                            if (!(q < Sq && key_ok && !(causal && kv_abs > q))) sc[i][r] = 0.f;
                        }
                }
                if (!(ARIA_DKDV_ABL & 8)) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg)
                            *reinterpret_cast<f32x4*>(myP + (i * 4 + rg) * 1024) = f32x4{sc[i][4 * rg], sc[i][4 * rg + 1], sc[i][4 * rg + 2], sc[i][4 * rg + 3]};
                }
            }
        }
        if (!(ARIA_DKDV_ABL & 16)) sync();  // P published
        if (active) {
            if (role == 1 && !(ARIA_DKDV_ABL & 128)) {
                const float* cD = sDel + cur * 64;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const f32x4 pv = (ARIA_DKDV_ABL & 8) ? f32x4{0.5f, 0.25f, 0.125f, 1.f} : *reinterpret_cast<const f32x4*>(myP + (i * 4 + rg) * 1024);
                        const f32x4 dl = *reinterpret_cast<const f32x4*>(cD + i * 32 + 8 * rg + 4 * h2);
#pragma unroll
                        for (int e = 0; e < 4; ++e) sc[i][4 * rg + e] = pv[e] * (sc[i][4 * rg + e] - dl[e]) * scale;
                    }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const s16x8 pf = pack_frag(sc[i], u);
                    if (ARIA_DKDV_ABL & 4) {
                        abl_keep(pf);
                        continue;
                    }
#pragma unroll
                    // (r06: operands swapped -- the accumulators are dV^T / dK^T, see the epilogue)
                    for (int dt = 0; dt < C::DT; ++dt) acc[dt] = mfma32(frag_tr3<HD>(second, i * 32 + 16 * u, 32 * dt, l), pf, acc[dt]);
                }
        }
        if (more && t < 64) {
            hold(lse_n);
            hold(del_n);
            const bool ok = qt0 + 64 + t < Sq;
            sLse[(cur ^ 1) * 64 + t] = ok ? lse_n * 1.4426950408889634f : 0.f;
            sDel[(cur ^ 1) * 64 + t] = ok ? del_n : 0.f;
        }
    }
    // r06: the accumulators are held TRANSPOSED (the product above is (Q | dO)^T x (dS | P), features x keys): lane (l & 31, h2) owns ONE key
    // and, per feature tile dt and q = 0..3, the four consecutive features 32 dt + 8 q + 4 h2 + 0..3 (registers 4 q .. 4 q + 3) -- 16 eight-byte
    // stores per lane instead of 64 two-byte ones, and the inverse RoPE's tables come in as 8-byte vectors (32 loads instead of 128 two-byte
    // ones).  At the training shape (S = 2048: 2 .. 32 query tiles per workgroup) this epilogue was a fifth of the kernel.  Same sums, same
    // roundings: bit-identical.
    bf16_t* out = role ? dK : dV;
    const long long ldout = role ? lddk : lddv;
    const int kv = kv_wmin + (l & 31);
    if (kv >= S) return;
    bf16_t* orow = out + (tok0 + kv) * ldout + head * HD + 4 * h2;
    if (role == 1 && rope_cs) {
        // r05: the inverse half-split RoPE of dK (the chain rule through apply_rotary_pos_emb, modeling_llama.py:130-160; was an in-place pass
        // over [T, 2 D] after this kernel: rope_kernel, 55 us per layer).  Feature f pairs with f + HD / 2 = tile dt + DT / 2, same lane, same
        // register; the key's position is its index in the sequence.  Rounding for rounding what the two launches did: the gradient rounded to
        // bf16 (the store), then bf16(bf16(a cos) + bf16(b sin)) / bf16(bf16(b cos) - bf16(a sin)).
        static_assert(C::DT % 2 == 0, "half-split pairs are whole feature tiles");
        const bf16_t* cs = rope_cs + (long long)(kv % rope_S) * HD + 4 * h2;
        const bf16_t* sn = rope_sn + (long long)(kv % rope_S) * HD + 4 * h2;
#pragma unroll
        for (int dt = 0; dt < C::DT / 2; ++dt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = 32 * dt + 8 * q;
                const u32x2 c1 = *reinterpret_cast<const u32x2*>(cs + f), c2 = *reinterpret_cast<const u32x2*>(cs + f + HD / 2);
                const u32x2 s1 = *reinterpret_cast<const u32x2*>(sn + f), s2 = *reinterpret_cast<const u32x2*>(sn + f + HD / 2);
                float lo[4], hi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = rbf(acc[dt][4 * q + j]), b = rbf(acc[dt + C::DT / 2][4 * q + j]);
                    const float vc1 = (j & 1) ? bfhi(c1[j >> 1]) : bflo(c1[j >> 1]), vc2 = (j & 1) ? bfhi(c2[j >> 1]) : bflo(c2[j >> 1]);
                    const float vs1 = (j & 1) ? bfhi(s1[j >> 1]) : bflo(s1[j >> 1]), vs2 = (j & 1) ? bfhi(s2[j >> 1]) : bflo(s2[j >> 1]);
                    lo[j] = rbf(a * vc1) + rbf(b * vs1);
                    hi[j] = rbf(b * vc2) + rbf(-a * vs2);
                }
                *reinterpret_cast<u32x2*>(orow + f) = u32x2{pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3])};
                *reinterpret_cast<u32x2*>(orow + f + HD / 2) = u32x2{pack2bf(hi[0], hi[1]), pack2bf(hi[2], hi[3])};
            }
        return;
    }
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<u32x2*>(orow + 32 * dt + 8 * q) =
                u32x2{pack2bf(acc[dt][4 * q], acc[dt][4 * q + 1]), pack2bf(acc[dt][4 * q + 2], acc[dt][4 * q + 3])};
}


// =========================================================================================== dQ v5 (hd = 128): no role split
// Round 2's dQ kernel halved the feature columns between two waves of a SIMD that exchanged P and dP through LDS (128 KiB of fp32 traffic
// and a second barrier per key tile; 24 MFMAs per wave and tile against 40 KiB of LDS reads + 16 KiB of exchange): 48.2 ms per layer at
// S = 65 536 against 29.4 ms for this one, bit-identical results (profiles/r03_attn_bwd_ab.json).  With the key / value tiles brought in
// by LDS-DMA there are no staging registers left in the loop, and ONE wave can hold the whole job for its 32 queries in < 256 VGPRs:
// Q and dO fragments (32 + 32), S^T and dP^T score tiles (32 + 32), dQ^T accumulators for all 128 features (64).  So: 8 waves x 32 queries
// = 256 queries per workgroup, per 64-key tile 16 + 16 + 16 MFMAs per wave against 48 KiB of fragment reads (1 KiB per MFMA, the ratio
// of the forward kernel), no exchange, ONE barrier per tile.
template <int HD>
__global__ __launch_bounds__(512) void attn_bwd5_dq_kernel(const bf16_t* Q, const bf16_t* K, const bf16_t* V, const bf16_t* dO,
                                                           const float* LSE, const float* DELTA, bf16_t* dQ, const int32_t* kv_len,
                                                           const uint8_t* key_mask, int Sq, int S, int H, int ldq, int ldk, int ldv,
                                                           int lddo, int lddq, float scale, int causal, int nbatch,
                                                           const bf16_t* rope_cs, const bf16_t* rope_sn, int rope_S) {
    using C = Cfg3<HD>;
    ARIA_DYN_SMEM(smem);
    char* sK = smem;                                   // [2] tiles
    char* sV = smem + 2 * C::TILE;                     // [2] tiles
    uint8_t* sM = reinterpret_cast<uint8_t*>(smem + 4 * C::TILE);  // [2][64]
    int* sFlag = reinterpret_cast<int*>(sM + 128);
    const int t = threadIdx.x, l = t & 63, w = first_lane(t >> 6), h2 = l >> 5;
    int qblk, head, b;
    if (!attn_block_coords((Sq + 255) / 256, H, nbatch, causal, true, qblk, head, b)) return;
    const int q0 = qblk * 256;
    const long long tok0 = (long long)b * S, tokq0 = (long long)b * Sq;
    const bf16_t* Kb = K + tok0 * ldk + head * HD;
    const bf16_t* Vb = V + tok0 * ldv + head * HD;
    const uint8_t* kmb = key_mask ? key_mask + tok0 : nullptr;
    const int q_wmin = q0 + 32 * w, q_abs = q_wmin + (l & 31);
    const int klen = kv_len ? min(S, kv_len[b]) : S;
    const float scale2 = scale * 1.4426950408889634f;
    s16x8 qf[C::KS], dof[C::KS];
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) {
        u32x4 a = zero16(), c = zero16();
        if (q_abs < Sq) {
            a = ld16(Q + (tokq0 + q_abs) * ldq + head * HD + kk * 16 + h2 * 8);
            c = ld16(dO + (tokq0 + q_abs) * lddo + head * HD + kk * 16 + h2 * 8);
        }
        qf[kk] = __builtin_bit_cast(s16x8, a);
        dof[kk] = __builtin_bit_cast(s16x8, c);
    }
    float lse2 = 0.f, del = 0.f;
    if (q_abs < Sq) {
        lse2 = LSE[((long long)b * H + head) * Sq + q_abs] * 1.4426950408889634f;
        del = DELTA[((long long)b * H + head) * Sq + q_abs];
    }
    const bool all_q_ok = q_wmin + 31 < Sq;
    f32x16 dq[C::DT];  // dQ^T: rows = features, cols = this wave's queries
#pragma unroll
    for (int i = 0; i < C::DT; ++i) dq[i] = zero_acc();
    int kv_end = klen;
    if (causal) kv_end = min(kv_end, q0 + 256);
    const int ntiles = (kv_end + 63) / 64;
    if (ntiles > 0) {  // first tile through registers (rows past the sequence as zeros)
        u32x4 rk[C::NCH], rv[C::NCH];
        tile_load3<HD>(rk, Kb, ldk, 0, S, t);
        tile_load3<HD>(rv, Vb, ldv, 0, S, t);
        tile_store3<HD>(rk, sK, t);
        tile_store3<HD>(rv, sV, t);
        if (kmb && t < 64) {
            const uint8_t mv = t < S ? kmb[t] : 0;
            sM[t] = mv;
            const unsigned long long all = ballot(mv != 0);
            if (t == 0) sFlag[0] = int(all == ~0ull) | (int(all == 0ull) << 1);  // bit 0: every key of the tile valid, bit 1: none
        }
    }
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) {
        settle(qf[kk]);
        settle(dof[kk]);
    }
    settle(lse2);
    settle(del);
    for (int it = 0; it < ntiles; ++it) {
        wait_vm<0>();  // this wave's LDS-DMA pieces of tile `it` have landed
        sync();        // tile `it` complete in buffer it & 1; every wave is done reading the other buffer
        const int cur = it & 1, kv0 = it * 64;
        const bool more = it + 1 < ntiles;
        if (more) {  // rows past the sequence are clamped: those keys are masked below
            tile_dma3<HD>(Kb, ldk, kv0 + 64, S - 1, sK + (cur ^ 1) * C::TILE, w, l);
            tile_dma3<HD>(Vb, ldv, kv0 + 64, S - 1, sV + (cur ^ 1) * C::TILE, w, l);
        }
        const char* cK = sK + cur * C::TILE;
        const char* cV = sV + cur * C::TILE;
        if (!(causal && kv0 > q_wmin + 31) && !(kmb && (sFlag[cur] & 2))) {  // wave-uniform: this wave sees at least one key of the tile (and the tile has one)
            f32x16 st[2], dpt[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                st[i] = zero_acc();
                dpt[i] = zero_acc();
            }
            // (an explicit one-step fragment prefetch as in the dK/dV kernel spills here: 252 VGPRs already)
#pragma unroll
            for (int kk = 0; kk < C::KS; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    st[i] = mfma32(frag_rc3<HD>(cK, i * 32 + (l & 31), kk, l), qf[kk], st[i]);
                    dpt[i] = mfma32(frag_rc3<HD>(cV, i * 32 + (l & 31), kk, l), dof[kk], dpt[i]);
                }
            const bool need_mask = !all_q_ok || (kv0 + 64 > klen) || (causal && kv0 + 63 > q_wmin) || (kmb && !(sFlag[cur] & 1));
            const uint8_t* cM = sM + cur * 64;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = exp2_fast(st[i][r] * scale2 - lse2);
                    if (need_mask) {
                        const int kvl = i * 32 + acc_row(r, l);
                        const int kv = kv0 + kvl;
                        bool ok = q_abs < Sq && kv < klen && !(causal && kv > q_abs);
                        if (kmb) ok = ok && cM[kvl];
                        if (!ok) p = 0.f;
                    }
                    dpt[i][r] = p * (dpt[i][r] - del) * scale;
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const s16x8 dsf = pack_frag(dpt[i], u);
#pragma unroll
                    for (int dt = 0; dt < C::DT; ++dt) dq[dt] = mfma32(frag_tr3<HD>(cK, i * 32 + 16 * u, 32 * dt, l), dsf, dq[dt]);
                }
        }
        if (more && kmb && t < 64) {
            const int nb = cur ^ 1, kvn = kv0 + 64;
            const uint8_t mv = (kvn + t < S) ? kmb[kvn + t] : 0;
            sM[nb * 64 + t] = mv;
            const unsigned long long all = ballot(mv != 0);
            if (t == 0) sFlag[nb] = int(all == ~0ull) | (int(all == 0ull) << 1);
        }
    }
    if (q_abs < Sq && rope_cs) {
        // r05: the inverse half-split RoPE of dQ in the register epilogue (see attn_bwd3_dkdv_kernel): a lane owns ONE query, features
        // d0 .. d0 + 3 of tile dt pair with the same registers of tile dt + DT / 2
        static_assert(C::DT % 2 == 0, "half-split pairs are whole feature tiles");
        bf16_t* row = dQ + (tokq0 + q_abs) * lddq + head * HD;
        const bf16_t* cs = rope_cs + (long long)(q_abs % rope_S) * HD;
        const bf16_t* sn = rope_sn + (long long)(q_abs % rope_S) * HD;
#pragma unroll
        for (int dt = 0; dt < C::DT / 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d0 = 32 * dt + 8 * rg + 4 * h2;
                const u32x2 c1 = *reinterpret_cast<const u32x2*>(cs + d0), c2 = *reinterpret_cast<const u32x2*>(cs + d0 + HD / 2);
                const u32x2 s1 = *reinterpret_cast<const u32x2*>(sn + d0), s2 = *reinterpret_cast<const u32x2*>(sn + d0 + HD / 2);
                u32x2 va, vb;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float oa[2], ob[2];
#pragma unroll
                    for (int z = 0; z < 2; ++z) {
                        const float a = rbf(dq[dt][4 * rg + 2 * q + z]), b = rbf(dq[dt + C::DT / 2][4 * rg + 2 * q + z]);
                        const float ca = z ? bfhi(c1[q]) : bflo(c1[q]), cb = z ? bfhi(c2[q]) : bflo(c2[q]);
                        const float sa = z ? bfhi(s1[q]) : bflo(s1[q]), sb = z ? bfhi(s2[q]) : bflo(s2[q]);
                        oa[z] = rbf(a * ca) + rbf(b * sa);
                        ob[z] = rbf(b * cb) + rbf(-a * sb);
                    }
                    va[q] = pack2bf(oa[0], oa[1]);
                    vb[q] = pack2bf(ob[0], ob[1]);
                }
                *reinterpret_cast<u32x2*>(row + d0) = va;
                *reinterpret_cast<u32x2*>(row + d0 + HD / 2) = vb;
            }
    } else if (q_abs < Sq) {
        bf16_t* row = dQ + (tokq0 + q_abs) * lddq + head * HD;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d0 = 32 * dt + 8 * rg + 4 * h2;
                u32x2 v;
                v[0] = pack2bf(dq[dt][4 * rg], dq[dt][4 * rg + 1]);
                v[1] = pack2bf(dq[dt][4 * rg + 2], dq[dt][4 * rg + 3]);
                *reinterpret_cast<u32x2*>(row + d0) = v;
            }
    }
}


// =========================================================================================== forward v3 (hd 128 / 72): software-pipelined
// Round 4.  An ablation of attn_fwd2 (profiles/r04_attn_fwd_ablate.json) showed what a key tile's time is made of: the vector ALU alone
// (scaling, v_exp_f32 at quarter rate, packing) needs 1.22 ms of the ViT launch's 2.26, the matrix pipe with its fragment reads alone 1.6 --
// their SUM, not their maximum, is what the kernel took (hd 128 at 64K: 9.2 + 15.6 against 23.2), and staging the next tile through
// registers + ds_write_b128 (13 LDS cycles per wave-instruction) another 15-20 %.  A wave's own chain QK -> softmax -> PV leaves nothing
// for the other pipe to do, and the tile barrier puts both waves of a SIMD into the same phase.  Here:
//   * the score tile of key tile t + 1 is computed INSIDE the block that runs the softmax and P V of tile t (S^T double-buffered in
//     registers, loop unrolled by two so no copies): 16 + 16 (hd 72: 10 + 12) MFMAs and the tile's ~100 vector instructions are
//     independent work in ONE basic block for the scheduler to interleave -- every wave feeds both pipes all the time;
//   * K runs one tile ahead of V in LDS; both arrive by LDS-DMA (no staging registers, no ds_write pass), one barrier per tile;
//   * hd 128: the backward kernels' tile image (256-byte rows, chunk swizzle swz3: conflict-free for the row fragments and the
//     transposing reads).  hd 72: K rows of 144 bytes as they are (36 dwords: the 16 rows of a ds_read_b128 lane group hit 16 different
//     bank quads; the reduction's 8 pad columns read the next row's first chunk against Q's zero pad); V rows of 160 bytes = 9 data chunks
//     + one constant chunk [1, 0 x 7] fetched from a device constant (the "ones" column that puts the softmax denominator on the matrix
//     pipe, as in v2), key j stored at row rho(j) = 8 (j >> 3) + 2 (j & 3) + ((j >> 2) & 1) so that the four keys of a transposing read lie
//     16 banks apart.
// Same arithmetic in the same order as v2: results are bit-identical (tests compare the two).

template <int HD>
struct FwdFmt;
template <>
struct FwdFmt<128> {
    static constexpr int KS = 8, DT = 4, KTILE = 16384, VTILE = 16384;
    static constexpr bool ROWSUM_IN_MFMA = false;
    int dummy;
    __device__ __forceinline__ void init(int, int, int) {}  // (tile_dma3: 8 waves x 2 pieces)
    __device__ __forceinline__ void dma_k(const bf16_t* base, long long ld, int row0, int row_last, char* s, int w, int l) const {
        tile_dma3<128>(base, int(ld), row0, row_last, s, w, l);
    }
    __device__ __forceinline__ void dma_v(const bf16_t* base, long long ld, int row0, int row_last, char* s, int w, int l) const {
        tile_dma3<128>(base, int(ld), row0, row_last, s, w, l);
    }
    static __device__ __forceinline__ s16x8 kfrag(const char* s, int row, int kk, int l) { return frag_rc3<128>(s, row, kk, l); }
    static __device__ __forceinline__ s16x8 vfrag(const char* s, int rb, int d0, int l) { return frag_tr3<128>(s, rb, d0, l); }
};
template <int HD>
struct Fwd3Smem {
    using F = FwdFmt<HD>;
    static constexpr int K0 = 0, V0 = 2 * F::KTILE, M0 = V0 + 2 * F::VTILE, FLAG0 = M0 + 3 * 64, BYTES = FLAG0 + 16;
};

// S^T tile of one key tile: rows = keys (two sub-tiles of 32), cols = this wave's 32 queries
template <int HD>
__device__ __forceinline__ void fwd3_qk(f32x16 (&st)[2], const char* cK, const s16x8 (&qf)[FwdFmt<HD>::KS], int l) {
    st[0] = zero_acc();
    st[1] = zero_acc();
#pragma unroll
    for (int kk = 0; kk < FwdFmt<HD>::KS; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) st[i] = mfma32(FwdFmt<HD>::kfrag(cK, i * 32 + (l & 31), kk, l), qf[kk], st[i]);
}

// key-mask / length / causal mask of one score tile (edge, diagonal and padded tiles only)
__device__ __forceinline__ void fwd3_mask(f32x16 (&st)[2], const uint8_t* cM, bool use_km, int kv0, int klen, int causal, int q_abs, int l) {
    hold(l);  // (the 32 row indices below are loop invariants: hoisted out of the tile loop they would sit in 32+ VGPRs across the hot block)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kvl = i * 32 + acc_row(r, l);
            const int kv = kv0 + kvl;
            bool dead = kv >= klen || (causal && kv > q_abs);
            if (use_km) dead = dead || !cM[kvl];
            if (dead) st[i][r] = -INFINITY;
        }
}

// One key tile of the pipelined loop: softmax + P V of the tile whose (masked) scores are in `st`, and the scores of the following tile
// into `sn` (ALWAYS: behind a wave's last tile the image it reads is a stale or never-written buffer and the result is dropped -- one code
// path keeps the accumulators in place; every extra variant of this block cost a second register copy of O^T at the merge points).  After the running-maximum update the tile is cut into FOUR sub-steps, one per 16-key group (i, u) of the P V product:
//     a quarter of the next tile's score MFMAs | the group's 8 exponentials + packing | the group's P V MFMAs (one per 32 features)
// -- independent matrix and vector work side by side in every sub-step; scheduling fences between the sub-steps keep the compiler from
// hoisting all fragment reads of the tile to the top (256 VGPRs and spills without them).
template <int HD, bool PIPE>
__device__ __forceinline__ void fwd3_step(f32x16 (&st)[2], f32x16 (&sn)[2], f32x16 (&o)[FwdFmt<HD>::DT], float& m, float& lsum,
                                          const s16x8 (&qf)[FwdFmt<HD>::KS], const char* cV, const char* nK, float scale2, int l) {
    using F = FwdFmt<HD>;
    constexpr int KS = F::KS;
    float mx = st[0][0];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[i][r]);
    mx = fmaxf(mx, shfl_xor(mx, 32));
    const float m_new = fmaxf(m, mx * scale2);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    if (ballot(m_new > m) != 0ull) {  // wave-uniform lazy rescale: exact (alpha == 1 whenever it is skipped)
        const float alpha = exp2_fast(m - m_safe);
        if (!F::ROWSUM_IN_MFMA) lsum *= alpha;
#pragma unroll
        for (int i = 0; i < F::DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        m = m_new;
    }
    if (PIPE) {
        sn[0] = zero_acc();
        sn[1] = zero_acc();
    }
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = j >> 1, u = j & 1;
        if (PIPE) {
            sched_fence();
            // k-steps [KS j / 4, KS (j + 1) / 4) of the next tile's scores (hd 128: 2 of 8; hd 72: 1, 1, 1, 2 of 5)
#pragma unroll
            for (int kk = KS * j / 4; kk < KS * (j + 1) / 4; ++kk)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) sn[ii] = mfma32(F::kfrag(nK, ii * 32 + (l & 31), kk, l), qf[kk], sn[ii]);
        }
#pragma unroll
        for (int r = 8 * u; r < 8 * u + 8; r += 2) {
            const f32x2 x = fma2(st[i][r], st[i][r + 1], scale2, -m_safe);
#pragma unroll
            for (int z = 0; z < 2; ++z) {
                const float p = exp2_fast(x[z]);
                st[i][r + z] = p;
                if (!F::ROWSUM_IN_MFMA) ps += p;
            }
        }
        const s16x8 pf = pack_frag(st[i], u);
#pragma unroll
        for (int dt = 0; dt < F::DT; ++dt) o[dt] = mfma32(F::vfrag(cV, i * 32 + 16 * u, 32 * dt, l), pf, o[dt]);
    }
    if (PIPE) sched_fence();
    if (!F::ROWSUM_IN_MFMA) lsum += ps;
}

// PIPE: the next tile's scores inside the current tile's block (S^T double-buffered: ~190 / 242 VGPRs, two waves per SIMD) -- the hd 128
// form and the default there.  !PIPE: scores at the top of the tile's own iteration as in v2, K and V tiles in step (hd 72 with NW = 12:
// 145 VGPRs, THREE waves per SIMD).  Measured for the ViT's shape (profiles/r04_attn_fwd3_ab.json): the third wave is worth more than the
// in-wave overlap, and with it LDS-DMA staging into the conflict-free images (SQ_LDS_BANK_CONFLICT 0, was 8 % of the LDS cycles) is level
// with v2's register staging -- so hd 72 stays on v2 by default and both v3 forms are kept selectable (ARIA_ATTN_FWD) as measurements.
template <int HD, int NW, bool PIPE>
__global__ __launch_bounds__(NW * 64) void attn_fwd3_kernel(const bf16_t* Q, const bf16_t* K, const bf16_t* V, bf16_t* O, float* LSE,
                                                        const int32_t* kv_len, const uint8_t* key_mask, int Sq, int S, int H,
                                                        long long ldq, long long ldk, long long ldv, long long ldo, float scale,
                                                        int causal, int nbatch) {
    using F = FwdFmt<HD>;
    using L = Fwd3Smem<HD>;
    constexpr int KS = F::KS, DT = F::DT;
    ARIA_DYN_SMEM(smem);
    char* sK = smem + L::K0;
    char* sV = smem + L::V0;
    uint8_t* sM = reinterpret_cast<uint8_t*>(smem + L::M0);   // [3][64]: the key-mask bytes of tiles t, t + 1, t + 2
    int* sFlag = reinterpret_cast<int*>(smem + L::FLAG0);     // [3]: bit 0 every key of the tile valid, bit 1 none
    const int t = threadIdx.x, l = t & 63, w = first_lane(t >> 6), h2 = l >> 5;
    int qblk, head, b;
    constexpr int QB = NW * 32, KA = PIPE ? 1 : 0;  // queries per workgroup; how many tiles K runs ahead of V in LDS
    static_assert(HD != 128 || NW == 8, "tile_dma3 deals a tile's 16 pieces to 8 waves");
    if (!attn_block_coords((Sq + QB - 1) / QB, H, nbatch, causal, true, qblk, head, b)) return;
    const int q0 = qblk * QB;
    const long long tok0 = (long long)b * S, tokq0 = (long long)b * Sq;
    const bf16_t* Qb = Q + tokq0 * ldq + head * HD;
    const bf16_t* Kb = K + tok0 * ldk + head * HD;
    const bf16_t* Vb = V + tok0 * ldv + head * HD;
    const uint8_t* kmb = key_mask ? key_mask + tok0 : nullptr;
    const int q_wmin = q0 + 32 * w, q_abs = q_wmin + (l & 31);
    const int klen = kv_len ? min(S, kv_len[b]) : S;
    const float scale2 = scale * 1.4426950408889634f;
    F fmt;
    fmt.init(w, l, NW);

    s16x8 qf[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        u32x4 v = zero16();
        const int col = kk * 16 + h2 * 8;
        if (q_abs < Sq && col < HD) v = ld16(Qb + (long long)q_abs * ldq + col);
        qf[kk] = __builtin_bit_cast(s16x8, v);
    }
    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] = zero_acc();
    float m = -INFINITY, lsum = 0.f;
    int kv_end = klen;
    if (causal) kv_end = min(kv_end, q0 + QB);
    const int ntiles = (kv_end + 63) / 64;

    // key-mask bytes of one tile -> ring slot tile % 3 (threads 0..63: wave 0)
    auto mask_fetch = [&](int tile) __attribute__((always_inline)) -> uint8_t { return (tile * 64 + t < S) ? kmb[tile * 64 + t] : uint8_t(0); };
    auto mask_park = [&](int tile, uint8_t mv) __attribute__((always_inline)) {
        sM[(tile % 3) * 64 + t] = mv;
        const unsigned long long all = ballot(mv != 0);
        if (t == 0) sFlag[tile % 3] = int(all == ~0ull) | (int(all == 0ull) << 1);
    };
    if (ntiles > 0) {
        if (HD == 72) {  // what the pad-column reads behind the last rows touch must hold finite values (0 x NaN = NaN)
            if (t < 2) st16(sK + t * F::KTILE + F::KTILE - 16, zero16());
            if (t >= 64 && t < 68) st16(sV + ((t - 64) >> 1) * F::VTILE + F::VTILE - 32 + 16 * (t & 1), zero16());
        }
        fmt.dma_k(Kb, ldk, 0, S - 1, sK, w, l);
        fmt.dma_v(Vb, ldv, 0, S - 1, sV, w, l);
        if (PIPE && ntiles > 1) fmt.dma_k(Kb, ldk, 64, S - 1, sK + F::KTILE, w, l);
        if (kmb && t < 64) {
            mask_park(0, mask_fetch(0));
            if (ntiles > 1) mask_park(1, mask_fetch(1));
        }
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) settle(qf[kk]);
    // tiles this wave works on: the causal diagonal ends its run early (the workgroup's later tiles hold no key it may see); a tile
    // WITHOUT a valid key (padded image rows) is worked on like any other -- every probability is exp2(-inf) = 0 and the running maximum
    // does not move, so it contributes exactly nothing (v2 skipped it; same bits)
    const int nmine = causal ? min(ntiles, (q_wmin + 31) / 64 + 1) : ntiles;
    auto tile_masked = [&](int tile) __attribute__((always_inline)) -> bool {
        return (tile * 64 + 64 > klen) || (causal && tile * 64 + 63 > q_wmin) || (kmb && !(sFlag[tile % 3] & 1));
    };
    // the part of an iteration every wave owes the workgroup: the tile barrier, its DMA pieces of K(it + 2) / V(it + 1), the mask bytes
    uint8_t mv = 0;
    auto duties = [&](int it) __attribute__((always_inline)) {
        wait_vm<0>();  // this wave's pieces of K(it + KA) and V(it) have landed
        sync();        // ... everybody's; every wave is done with K(it + KA - 1) and V(it - 1)
        if (it + KA + 1 < ntiles) fmt.dma_k(Kb, ldk, (it + KA + 1) * 64, S - 1, sK + ((it + KA + 1) & 1) * F::KTILE, w, l);
        if (it + 1 < ntiles) fmt.dma_v(Vb, ldv, (it + 1) * 64, S - 1, sV + ((it + 1) & 1) * F::VTILE, w, l);
        if (kmb && t < 64 && it + 2 < ntiles) mv = mask_fetch(it + 2);
    };
    auto park = [&](int it) __attribute__((always_inline)) {
        if (kmb && t < 64 && it + 2 < ntiles) mask_park(it + 2, mv);
    };
    f32x16 sa[2], sb[2];
    if (PIPE && ntiles > 0) {
        wait_vm<0>();
        sync();
        if (nmine > 0) fwd3_qk<HD>(sa, sK, qf, l);
    }
    // one iteration: tile `it` (PIPE: its scores are in `cur`) -> softmax + P V; PIPE: tile it + 1 -> scores into `nxt`
    auto iterate = [&](int it, f32x16 (&cur)[2], f32x16 (&nxt)[2]) __attribute__((always_inline)) {
        duties(it);
        if (!PIPE) fwd3_qk<HD>(cur, sK + (it & 1) * F::KTILE, qf, l);
        if (tile_masked(it)) fwd3_mask(cur, sM + (it % 3) * 64, kmb != nullptr, it * 64, klen, causal, q_abs, l);
        fwd3_step<HD, PIPE>(cur, nxt, o, m, lsum, qf, sV + (it & 1) * F::VTILE, sK + ((it + 1) & 1) * F::KTILE, scale2, l);
        park(it);
    };
    int it = 0;
    if (PIPE) {
        for (; it + 1 < nmine; it += 2) {
            iterate(it, sa, sb);
            iterate(it + 1, sb, sa);
        }
        if (it < nmine) {
            iterate(it, sa, sb);
            ++it;
        }
    } else {
        for (; it < nmine; ++it) iterate(it, sa, sb);
    }
    for (; it < ntiles; ++it) {  // the workgroup's remaining tiles (other waves' diagonals): barriers and staging only
        duties(it);
        park(it);
    }
    // ROWSUM_IN_MFMA: feature row HD = 32 (DT-1) + 8 sits in accumulator register 4 of the lanes with h2 == 0
    const float ltot = F::ROWSUM_IN_MFMA ? shfl(o[DT - 1][4], l & 31) : lsum + shfl_xor(lsum, 32);
    const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
    if (q_abs < Sq) {
        if (h2 == 0 && LSE)
            LSE[((long long)b * H + head) * Sq + q_abs] = (ltot > 0.f) ? (m + log2f(ltot)) * 0.6931471805599453f : -INFINITY;
        bf16_t* orow = O + (tokq0 + q_abs) * ldo + head * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d0 = 32 * dt + 8 * rg + 4 * h2;
                if (d0 < HD) {
                    u32x2 v;
                    v[0] = pack2bf(o[dt][4 * rg] * inv, o[dt][4 * rg + 1] * inv);
                    v[1] = pack2bf(o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv);
                    *reinterpret_cast<u32x2*>(orow + d0) = v;
                }
            }
    }
}


thread_local int g_last_fwd_variant = 0;  // 2: attn_fwd2 (staged through registers), 3: attn_fwd3 (software-pipelined, LDS-DMA)
thread_local int g_last_bwd_variant = 0;  // 2: the padded-tile pair (hd 64 / 72), 5: role-split dK/dV + dQ v5 (hd 128) -- tests assert which ran
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int aria_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* kv_len, const uint8_t* key_mask,
                  int64_t B, int64_t Sq, int64_t Skv, int64_t H, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                  float scale, int causal, void* stream) {
    if (!q || !k || !v || !o || B < 0 || Sq < 0 || Skv < 0 || H <= 0) return ARIA_ERR_INVALID;
    if (hd != 64 && hd != 72 && hd != 128) return ARIA_ERR_UNSUPPORTED;
    if (causal && Sq != Skv) return ARIA_ERR_UNSUPPORTED;
    if (!al16(q) || !al16(k) || !al16(v) || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3) || (reinterpret_cast<uintptr_t>(o) & 7))
        return ARIA_ERR_ALIGN;
    if (B == 0 || Sq == 0) return ARIA_OK;
    const bf16_t *Q = static_cast<const bf16_t*>(q), *K = static_cast<const bf16_t*>(k), *V = static_cast<const bf16_t*>(v);
    const char* nw72 = std::getenv("ARIA_ATTN_HD72_WAVES");  // "8": the two-waves-per-SIMD variant (A/B measurements)
    const char* fwdv = std::getenv("ARIA_ATTN_FWD");          // "2": the round-1..3 kernels (A/B measurements, bit-identity tests)
    const char* grp = std::getenv("ARIA_ATTN_CAUSAL_GROUPED");   // "1" / "0": force the XCD-grouped causal block order on / off (tests, A/B)
    const int causal_arg = causal ? ((grp ? grp[0] == '1' : Sq >= 32768) ? 2 : 1) : 0;   // (2: attn_block_coords' grouped order; default: the forward from 32 K tokens up)
    // default: hd 128 -> v3 pipelined (+4..8 % over v2 from 2K to 64K tokens); hd 72 -> v2 with 12 waves (measured, same box: v2 2.23-2.46 ms,
    // v3 with 12 waves + LDS-DMA 2.38-2.45, v3 pipelined with 8 waves 2.78-2.81 per ViT launch: profiles/r04_attn_fwd3_ab.json).
    // ARIA_ATTN_FWD = "2": v2 everywhere.
    // (r05: the two hd-72 forms of v3 -- measured slower than v2's 12 waves -- left the library: tools/probes/src/attn_fwd3_hd72.patch)
    const bool v3 = !(fwdv && fwdv[0] == '2') && hd == 128 && Skv > 0;
    g_last_fwd_variant = v3 ? 3 : 2;
    if (v3)
        ARIA_LAUNCH((attn_fwd3_kernel<128, 8, true>), dim3(attn_grid((Sq + 255) / 256, H, B)), dim3(512), size_t(Fwd3Smem<128>::BYTES), stream, Q, K, V,
                    static_cast<bf16_t*>(o), lse, kv_len, key_mask, int(Sq), int(Skv), int(H), (long long)ldq, (long long)ldk,
                    (long long)ldv, (long long)ldo, scale, causal_arg, int(B));
    else if (hd == 128)
        ARIA_LAUNCH((attn_fwd2_kernel<128, 8>), dim3(attn_grid((Sq + 255) / 256, H, B)), dim3(512), size_t(Cfg2<128>::SMEM), stream, Q, K, V,
                    static_cast<bf16_t*>(o), lse, kv_len, key_mask, int(Sq), int(Skv), int(H), (long long)ldq, (long long)ldk,
                    (long long)ldv, (long long)ldo, scale, causal, int(B));
    else if (hd == 72 && nw72 && nw72[0] == '8')
        ARIA_LAUNCH((attn_fwd2_kernel<72, 8>), dim3(attn_grid((Sq + 255) / 256, H, B)), dim3(512), size_t(Cfg2<72>::SMEM), stream, Q, K, V,
                    static_cast<bf16_t*>(o), lse, kv_len, key_mask, int(Sq), int(Skv), int(H), (long long)ldq, (long long)ldk,
                    (long long)ldv, (long long)ldo, scale, causal, int(B));
    else if (hd == 72)
        ARIA_LAUNCH((attn_fwd2_kernel<72, 12>), dim3(attn_grid((Sq + 383) / 384, H, B)), dim3(768), size_t(Cfg2<72>::SMEM), stream, Q, K, V,
                    static_cast<bf16_t*>(o), lse, kv_len, key_mask, int(Sq), int(Skv), int(H), (long long)ldq, (long long)ldk,
                    (long long)ldv, (long long)ldo, scale, causal, int(B));
    else
        ARIA_LAUNCH((attn_fwd2_kernel<64, 8>), dim3(attn_grid((Sq + 255) / 256, H, B)), dim3(512), size_t(Cfg2<64>::SMEM), stream, Q, K, V,
                    static_cast<bf16_t*>(o), lse, kv_len, key_mask, int(Sq), int(Skv), int(H), (long long)ldq, (long long)ldk,
                    (long long)ldv, (long long)ldo, scale, causal, int(B));
    return aria_check_launch();
}

int aria_last_attn_bwd_variant(void) { return g_last_bwd_variant; }
int aria_last_attn_fwd_variant(void) { return g_last_fwd_variant; }

int aria_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, float* delta,
                  void* dq, void* dk, void* dv, const int32_t* kv_len, const uint8_t* key_mask, int64_t B, int64_t Sq, int64_t Skv,
                  int64_t H, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv,
                  float scale, int causal, void* stream) {
    return aria_attn_bwd_rope(q, k, v, o, d_o, lse, delta, dq, dk, dv, kv_len, key_mask, B, Sq, Skv, H, hd, ldq, ldk, ldv, ldo, lddq, lddk, lddv,
                              scale, causal, nullptr, nullptr, 0, stream);
}

int aria_attn_bwd_rope(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, float* delta,
                       void* dq, void* dk, void* dv, const int32_t* kv_len, const uint8_t* key_mask, int64_t B, int64_t Sq, int64_t Skv,
                       int64_t H, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv,
                       float scale, int causal, const void* rope_cos, const void* rope_sin, int64_t rope_S, void* stream) {
    if (!q || !k || !v || !o || !d_o || !lse || !delta || !dq || !dk || !dv || B < 0 || Sq < 0 || Skv < 0 || H <= 0)
        return ARIA_ERR_INVALID;
    if ((rope_cos != nullptr) != (rope_sin != nullptr) || (rope_cos && rope_S <= 0)) return ARIA_ERR_INVALID;
    if (rope_cos && hd != 128) return ARIA_ERR_UNSUPPORTED;   // (the decoder's heads; the hd 64 / 72 kernels have no such epilogue)
    if (rope_cos && ((reinterpret_cast<uintptr_t>(rope_cos) | reinterpret_cast<uintptr_t>(rope_sin)) & 7)) return ARIA_ERR_ALIGN;
    const bf16_t *rcs = static_cast<const bf16_t*>(rope_cos), *rsn = static_cast<const bf16_t*>(rope_sin);
    if (hd != 64 && hd != 72 && hd != 128) return ARIA_ERR_UNSUPPORTED;
    if (causal && Sq != Skv) return ARIA_ERR_UNSUPPORTED;
    if (!al16(q) || !al16(k) || !al16(v) || !al16(o) || !al16(d_o) || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 7) ||
        (lddq & 3) || (lddk & 1) || (lddv & 1) || (reinterpret_cast<uintptr_t>(dq) & 7))
        return ARIA_ERR_ALIGN;
    // (r06: the hd 128 dK / dV kernel stores 8 bytes per lane: rows 8-byte aligned like dq's)
    if (hd == 128 && ((lddk & 3) || (lddv & 3) || ((reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dv)) & 7))) return ARIA_ERR_ALIGN;
    if (B == 0 || Sq == 0 || Skv == 0) return ARIA_OK;
    const bf16_t *Q = static_cast<const bf16_t*>(q), *K = static_cast<const bf16_t*>(k), *V = static_cast<const bf16_t*>(v);
    const bf16_t *Op = static_cast<const bf16_t*>(o), *dO = static_cast<const bf16_t*>(d_o);
    // dO is laid out like O (same leading dimension)
    const long long nrows = B * Sq * H;
    long long g = (nrows + 15) / 16;
    if (g > 8192) g = 8192;
    ARIA_LAUNCH(attn_delta_kernel, dim3(unsigned(g)), dim3(256), 0, stream, Op, dO, delta, int(Sq), int(H), int(hd), (long long)ldo,
                (long long)ldo, nrows);
    dim3 gridk(unsigned((Skv + 127) / 128), unsigned(H), unsigned(B)), gridq(unsigned((Sq + 127) / 128), unsigned(H), unsigned(B));
    dim3 block(NT);
    if (hd == 128) {  // the decoder's heads: role-split dK/dV kernel + dQ v5
        ARIA_LAUNCH((attn_bwd3_dkdv_kernel<128>), dim3(attn_grid((Skv + 127) / 128, H, B)), dim3(512), size_t(Cfg3<128>::SMEM_DKDV), stream, Q, K, V, dO, lse,
                    (const float*)delta, static_cast<bf16_t*>(dk), static_cast<bf16_t*>(dv), kv_len, key_mask, int(Sq), int(Skv),
                    int(H), (long long)ldq, (long long)ldk, (long long)ldv, (long long)ldo, (long long)lddk, (long long)lddv, scale,
                    causal, int(B), rcs, rsn, int(rope_S));
        ARIA_LAUNCH((attn_bwd5_dq_kernel<128>), dim3(attn_grid((Sq + 255) / 256, H, B)), dim3(512), size_t(4 * Cfg3<128>::TILE + 128 + 16), stream, Q, K, V,
                    dO, lse, (const float*)delta, static_cast<bf16_t*>(dq), kv_len, key_mask, int(Sq), int(Skv), int(H), int(ldq), int(ldk),
                    int(ldv), int(ldo), int(lddq), scale, causal, int(B), rcs, rsn, int(rope_S));
        g_last_bwd_variant = 5;
    } else if (hd == 72) {  // ViT / projector heads (an unfrozen ViT, the trainable projector's cross-attention): the v2 pair with padded tiles
        using C = Cfg<72>;
        ARIA_LAUNCH((attn_bwd2_dkdv_kernel<72>), gridk, block, size_t(4 * 64 * C::PITCH * 2 + 256 * 4), stream, Q, K, V, dO, lse,
                    (const float*)delta, static_cast<bf16_t*>(dk), static_cast<bf16_t*>(dv), kv_len, key_mask, int(Sq), int(Skv),
                    int(H), (long long)ldq, (long long)ldk, (long long)ldv, (long long)ldo, (long long)lddk, (long long)lddv, scale,
                    causal);
        ARIA_LAUNCH((attn_bwd2_dq_kernel<72>), gridq, block, size_t(4 * 64 * C::PITCH * 2 + 128 + 16), stream, Q, K, V, dO, lse,
                    (const float*)delta, static_cast<bf16_t*>(dq), kv_len, key_mask, int(Sq), int(Skv), int(H), (long long)ldq,
                    (long long)ldk, (long long)ldv, (long long)ldo, (long long)lddq, scale, causal);
        g_last_bwd_variant = 2;
    } else {
        using C = Cfg<64>;
        ARIA_LAUNCH((attn_bwd2_dkdv_kernel<64>), gridk, block, size_t(4 * 64 * C::PITCH * 2 + 256 * 4), stream, Q, K, V, dO, lse,
                    (const float*)delta, static_cast<bf16_t*>(dk), static_cast<bf16_t*>(dv), kv_len, key_mask, int(Sq), int(Skv),
                    int(H), (long long)ldq, (long long)ldk, (long long)ldv, (long long)ldo, (long long)lddk, (long long)lddv, scale,
                    causal);
        ARIA_LAUNCH((attn_bwd2_dq_kernel<64>), gridq, block, size_t(4 * 64 * C::PITCH * 2 + 128 + 16), stream, Q, K, V, dO, lse,
                    (const float*)delta, static_cast<bf16_t*>(dq), kv_len, key_mask, int(Sq), int(Skv), int(H), (long long)ldq,
                    (long long)ldk, (long long)ldv, (long long)ldo, (long long)lddq, scale, causal);
        g_last_bwd_variant = 2;
    }
    return aria_check_launch();
}

}  // extern "C"
