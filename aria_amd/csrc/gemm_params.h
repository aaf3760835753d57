// Parameter block shared by the GEMM kernels (gemm.hip: 128x128 tile v1, gemm2.hip: 256x256 tile v2).
#pragma once
#include "aria_device.h"

struct GemmParams {
    const ad::bf16_t* A;
    const ad::bf16_t* B;
    void* C;
    const ad::bf16_t* bias;
    long long lda, ldb, ldc;
    int M, N, K;
    int mode;  // 0 dense, 1 grouped-M (rows grouped by expert), 2 grouped-K (per-expert wgrad)
    const int* offsets;
    int E;
    long long strideB, strideC;
    int c_f32, accumulate;
    int act;  // epilogue activation applied to the bf16-ROUNDED (acc + bias), before any accumulate: 0 none, 1 gelu_tanh
    int ntn, ntm;  // column / row tile counts of the launch (256-wide tiles for v2/v3)
    // v3 remainder split-K (mode 0): tile-list positions >= split_first are each computed by `split` workgroups over disjoint K
    // ranges into fp32 slabs (ws), which a second kernel sums in a fixed order.  split <= 1: off.
    int split, split_first;
    float* ws;
    // v3, fused SwiGLU epilogue (GroupedMLP.forward's fc1 + glu, moe_lm.py:505-507,522-523; the shared expert's gate/up + act): the B
    // operand holds [gate | up] columns (N = 2 I); a workgroup's two 128-column B halves are gate columns n0.. and up columns I + n0..,
    // so gate and up of one output element meet in one lane.  C2 [M, I] receives silu(gate) * up with the reference's rounding points;
    // C ([M, 2 I], may be null) the un-activated product.
    int glu;
    void* C2;
    long long ldc2;
    int reserved0;   // (was the round-2 start-up stagger experiment; the field keeps the kernarg offsets of what follows)
    int wide_store;  // v3: C rows are 16-byte aligned at every 8th column (pointer, ldc, strideC): full-width column tiles take the wide epilogue
    int order;     // tile-order variant (tuning knob): 0 = XCD-contiguous row-major, 1 = plain, >= 2 = groups of `order` row tiles
    // (appended: the fields above keep their kernarg offsets, the default kernels' ISA does not move)
    // v3, fused SwiGLU-BACKWARD epilogue (dglu; gemm3_kernel<.., .., 5>): the GEMM is the down-projection's input gradient d_act = dY W2^T
    // (N = I columns); H [M, 2 I] holds the forward's [gate | up]; C [M, 2 I] receives [d_gate | d_up] = swiglu_bwd(H, d_act) -- d_act
    // itself (rounded to bf16 exactly where the two-step chain materialises it) never visits HBM.  Needs I % 128 == 0.
    int dglu;
    const ad::bf16_t* H;
    long long ldh;
    // fused SwiGLU epilogue with gate and up weights in TWO tensors of one allocation (gptfast: cond_ffn.w1 / w3 [E, I, D], shared_ffn.w1 /
    // w3 [Is, D]): the up rows of an expert start `glu_up_rows` rows of the B operand behind its gate rows (0: the [gate | up] layout, N / 2)
    int glu_up_rows;
    // K2 (fused SwiGLU launches over grouped rows only): A is the UN-permuted token matrix [T, K] and gather_rows[r] the token row that
    // permuted row r would hold (TokenDispatcher.token_permutation's index_select, moe_lm.py:326-334, folded into the A loader)
    const int* gather_rows;
    // K7 (gemm3_kernel<false, false, 7>, dense): the fused wqkv projection of gptfast's Attention.forward (gptfast/model.py:413-435) with
    // the interleaved RoPE and the KV-cache write as its epilogue.  N = 3 D; a 256-column tile lies wholly in the q, k or v block
    // (D % 256 == 0).  q columns: rotated, written to C [M, ldc]; k columns: rotated, written to kc; v columns: written to vc -- both cache
    // tensors [B, cache_S, D] with row stride ld_cache, row of token t = (t / rope_S) * cache_S + pos(t), pos(t) = rope_pos[t] (or t % rope_S).
    // The rotation reads the bf16-rounded product, computes in fp32 and rounds once -- the bits of gemm + rope_interleaved + copy.
    const ad::bf16_t* rope_fc;   // freqs_cis [positions, hd / 2, 2] bf16 (cos, sin)
    const int* rope_pos;
    int rope_hd, rope_D, rope_S, cache_S;
    void* kc;
    void* vc;
    long long ld_cache;
    // grouped rows whose groups are SEGMENTS of an expert-parallel exchange (rows arrive ordered (source rank, local expert)): group g uses
    // the weight of expert g % expert_mod (0: group g uses weight g).  The all-to-all's output is consumed in arrival order, no re-order pass.
    int expert_mod;
    // v3, K-EXTENSION (LoRA inside the base GEMM; GroupedGemmLoraLayer.forward aria/lora/layers.py:129-139, peft's Linear adapter):
    //   C = A B + extA extB, the second product carried by ONE extra K-tile behind the reduction: its DMA granules take their 16 bytes from
    //   extA [M, ext_k] (k-contiguous rows, like A) and from extB (the B operand's own form: [N, ext_k] rows for k-contiguous weights,
    //   [ext_k, N] for [K, N] weights; per expert at stride_extB) wherever the reduction index is < ext_k and from the zero page beyond --
    //   the mechanism of the ragged last K-tile.  ext_k % 8 == 0, <= 64; K % 64 == 0; every epilogue (SwiGLU, SwiGLU backward, wide
    //   stores) sees base + adapter in its accumulators, rounded ONCE.  0 = off: the kernels' steady-state code does not read these fields.
    // the HF layer's q | k | v projection with the half-split RoPE in its epilogue (gemm3_kernel<false, false, 10>): rope_fc = cos, rope_sn = sin,
    // both [rope_S, rope_hd] bf16; columns [0, 2 rope_D) rotate, position of row m = m % rope_S
    const ad::bf16_t* rope_sn;
    const ad::bf16_t* extA;
    const ad::bf16_t* extB;
    long long ld_extA, ld_extB, stride_extB;
    int ext_k;
};
// (the device helpers below are templates on the block's type so that a kernel may also hand them the block where it lies in the
// kernarg segment -- a reference into the constant address space: scalar loads at the point of use instead of registers held live)

// position in the (mode 0 / 2) tile list -> (row tile, column tile)
template <class P>
__device__ __forceinline__ bool aria_tile_from_pos(const P& p, int tile, int& tmi, int& tn) {
    tn = tile % p.ntn;
    tmi = tile / p.ntn;
    if ((p.order & 255) >= 2) {
        const int GM = p.order & 255, per = GM * p.ntn;
        const int g = tile / per, in = tile % per;
        const int gm = min(GM, p.ntm - g * GM);
        tn = in / gm;
        tmi = g * GM + in % gm;
    }
    return tmi < p.ntm;
}

// Workgroup id -> (row tile, column tile) for the 256x256 kernels (v2, v3).  Workgroup b runs on XCD b % 8 and each XCD has its own
// L2, so ids are remapped to give every XCD runs of neighbouring tiles: order >= 2 walks groups of GM row tiles column-major, so the
// ~32 tiles an XCD works on at a time form a GM x (32/GM) patch that shares A row panels and B column panels.
// Each XCD gets one contiguous chunk of the tile list (modes 0 and 2; mode 1 uses aria_grouped_tile below).
// Returns false when the workgroup has no tile.
template <class P>
__device__ __forceinline__ bool aria_tile_coords(const P& p, int bid, int nwg, int& tmi, int& tn) {
    int tile = bid;
    if ((p.order & 255) != 1) {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    return aria_tile_from_pos(p, tile, tmi, tn);
}
// Grouped rows (mode 1): expert-major tile list.  The tiles of expert e are its nt_e = ceil(n_e / 256) row tiles x ntn column
// tiles, walked COLUMN-major (all row tiles of a column first), experts in order; the list (T real tiles -- known only on the
// device, the offsets never visit the host) is cut into 8 equal contiguous chunks, one per XCD.  So the ~32 tiles an XCD works on at
// a time belong to one expert and share its weight column panels (each fetched from HBM once instead of once per pair of row
// tiles) and its row panels, and every XCD gets the same number of REAL tiles however uneven the routing is.
// Every lane of the calling wave must take part (wave collectives).  Returns false when the workgroup has no tile.
// p.order bit 9 (ragged-last): an expert's last row tile usually holds only a few rows and finishes early, which lets its CU drift out
// of step with the CUs that share its panels -- measured, fc1 forward: L2 hit rate 47 % with routed counts against 75 % with every expert at
// a multiple of 256 rows (tools/gpu_pmc_grouped.sh).  With the bit set every XCD first runs its share of the FULL row tiles (same list
// order) and then its share of the ragged ones (one per expert and column), which are alike among themselves.
// p.order bit 12 (r06, per-expert interleave): instead of one contiguous eighth of the whole list, XCD x takes an eighth of EVERY expert's tiles
// (share y = (x + e) & 7 of expert e: positions [c_e y / 8, c_e (y + 1) / 8) of its column-major list -- the rotation by e spreads the
// rounding remainders), in expert order.  Every XCD then sees the same mix of experts at the same time, as the per-expert grids of the weight
// gradients do by construction (their XCDs finish within 0.3 % of each other; the contiguous eighths of the grouped-row launches 5-8 % apart,
// profiles/r06_grouped6_timeline_*.json) -- at the price of an XCD's patch of one expert being 8 tiles instead of ~32.
template <class P>
__device__ __forceinline__ bool aria_grouped_tile_interleaved(const P& p, int bid, int l, int& expert, int& m0, int& m_end, int& tn) {
    const bool split = (p.order >> 9) & 1, rfirst = (p.order >> 11) & 1;
    const int xcd = bid & 7, idx = bid >> 3;
    // pass 0: this XCD's tile counts in the two lists; pass 1: locate
    int nF = 0, nR = 0;
    for (int e0 = 0; e0 < p.E; e0 += 64) {
        const int e = e0 + l;
        int sf = 0, sr = 0;
        if (e < p.E) {
            const int n = p.offsets[e + 1] - p.offsets[e], y = (xcd + e) & 7;
            const int cf = (split ? n / 256 : (n + 255) / 256) * p.ntn, cr = (split && (n & 255)) ? p.ntn : 0;
            sf = cf * (y + 1) / 8 - cf * y / 8;
            sr = cr * (y + 1) / 8 - cr * y / 8;
        }
        nF += ad::wave_bcast(ad::wave_incl_scan(sf), 63);
        if (split) nR += ad::wave_bcast(ad::wave_incl_scan(sr), 63);
    }
    if (idx >= nF + nR) return false;
    const bool ragged = rfirst ? idx < nR : idx >= nF;
    const int v = ragged ? idx - (rfirst ? 0 : nF) : idx - (rfirst ? nR : 0);   // position in this XCD's share of the list
    int base = 0;
    for (int e0 = 0; e0 < p.E; e0 += 64) {
        const int e = e0 + l;
        int o0 = 0, o1 = 0;
        if (e < p.E) {
            o0 = p.offsets[e];
            o1 = p.offsets[e + 1];
        }
        const int n = o1 - o0, y = (xcd + e) & 7;
        const int nt = split ? n / 256 : (n + 255) / 256;
        const int c = ragged ? ((n & 255) ? p.ntn : 0) : nt * p.ntn;
        const int first = c * y / 8, share = (e < p.E) ? c * (y + 1) / 8 - first : 0;
        const int incl = ad::wave_incl_scan(share);
        const int excl = base + incl - share;
        const unsigned long long mask = ad::ballot(share > 0 && v >= excl && v < excl + share);
        if (mask) {
            const int src = __builtin_ctzll(mask);
            const int local = ad::wave_bcast(first, src) + v - ad::wave_bcast(excl, src), nts = ad::wave_bcast(nt, src);
            expert = e0 + src;
            int row;
            if (ragged) {
                tn = local;
                row = nts;
            } else {
                tn = local / nts;
                row = local % nts;
            }
            m0 = ad::wave_bcast(o0, src) + row * 256;
            m_end = ad::wave_bcast(o1, src);
            return true;
        }
        base += ad::wave_bcast(incl, 63);
    }
    return false;
}

template <class P>
__device__ __forceinline__ bool aria_grouped_tile(const P& p, int bid, int l, int& expert, int& m0, int& m_end, int& tn) {
    if ((p.order >> 12) & 1) return aria_grouped_tile_interleaved(p, bid, l, expert, m0, m_end, tn);
    const bool split = (p.order >> 9) & 1;
    int TF = 0, TR = 0;  // tiles in the list of full (or, without the bit, all) row tiles / of ragged row tiles
    for (int e0 = 0; e0 < p.E; e0 += 64) {
        const int e = e0 + l;
        int cf = 0, cr = 0;
        if (e < p.E) {
            const int n = p.offsets[e + 1] - p.offsets[e];
            cf = (split ? n / 256 : (n + 255) / 256) * p.ntn;
            cr = (split && (n & 255)) ? p.ntn : 0;
        }
        TF += ad::wave_bcast(ad::wave_incl_scan(cf), 63);
        if (split) TR += ad::wave_bcast(ad::wave_incl_scan(cr), 63);
    }
    // (order bits 16-18, diagnostic: the eighth of the list a hardware XCD takes is rotated by that many places -- do the late XCDs of a launch
    // follow the hardware id or the share?  profiles/r06_xcd_rotation.json)
    const int xcd = ((bid & 7) + ((p.order >> 16) & 7)) & 7, idx = bid >> 3;
    const int loF = int((long long)TF * xcd / 8), nF = int((long long)TF * (xcd + 1) / 8) - loF;
    const int loR = int((long long)TR * xcd / 8), nR = int((long long)TR * (xcd + 1) / 8) - loR;
    if (idx >= nF + nR) return false;
    // bit 11 (with bit 9): the ragged tiles FIRST -- the short tiles open every XCD's run, where all 32 CUs start
    // together anyway, and the run ends on full tiles, whose partial last round costs what it always costs
    const bool rfirst = (p.order >> 11) & 1;
    const bool ragged = rfirst ? idx < nR : idx >= nF;
    const int v = ragged ? loR + idx - (rfirst ? 0 : nF) : loF + idx - (rfirst ? nR : 0);
    int base = 0;
    for (int e0 = 0; e0 < p.E; e0 += 64) {
        const int e = e0 + l;
        int o0 = 0, o1 = 0;
        if (e < p.E) {
            o0 = p.offsets[e];
            o1 = p.offsets[e + 1];
        }
        const int n = o1 - o0;
        const int nt = split ? n / 256 : (n + 255) / 256;  // row tiles of this expert in the full list
        const int c = ragged ? ((n & 255) ? p.ntn : 0) : nt * p.ntn;
        const int incl = ad::wave_incl_scan(c);  // (DPP, no LDS round trips: this runs in front of every grouped tile)
        const int excl = base + incl - c;
        const unsigned long long mask = ad::ballot(c > 0 && v >= excl && v < excl + c);
        if (mask) {
            const int src = __builtin_ctzll(mask);
            const int local = v - ad::wave_bcast(excl, src), nts = ad::wave_bcast(nt, src);
            expert = e0 + src;
            int row;
            if (ragged) {  // the expert's last (partial) row tile, columns in order
                tn = local;
                row = nts;
            } else {  // column-major over the expert's nts row tiles
                tn = local / nts;
                row = local % nts;
            }
            m0 = ad::wave_bcast(o0, src) + row * 256;
            m_end = ad::wave_bcast(o1, src);
            return true;
        }
        base += ad::wave_bcast(incl, 63);
    }
    return false;
}
// number of workgroups aria_tile_coords / aria_grouped_tile need
inline int aria_tile_grid(const GemmParams& p) {
    // aria_grouped_tile: 8 XCD chunks of ceil(T / 8) <= bound / 8 + 1 tiles; the ragged-last order (order bit 9) cuts TWO lists (full
    // row tiles, ragged last row tiles) 8 ways each, so one XCD can need ceil(TF / 8) + ceil(TR / 8) slots: 16 spare workgroups
    // (bit 12, per-expert interleave: an XCD's share is sum_e of a rounded eighth -- at most one tile per expert and list above the mean)
    if (p.mode == 1) return p.ntn * p.ntm + 16 + (((p.order >> 12) & 1) ? 16 * p.E : 0);
    if (p.split > 1) return p.split_first + (p.ntn * p.ntm - p.split_first) * p.split;
    return p.ntn * p.ntm;
}

// the epilogue value of every GEMM kernel: bias added by the caller; activation exactly as a separate elementwise kernel would see it
template <class P>
__device__ __forceinline__ float aria_epilogue_act(const P& p, float v) {
    return p.act == 1 ? ad::gelu_tanh(ad::rbf(v)) : v;
}
// the same with the activation as a compile-time constant: inside an unrolled epilogue the run-time form costs one branch PER VALUE
template <int ACT>
__device__ __forceinline__ float aria_epilogue_act_c(float v) {
    return ACT == 1 ? ad::gelu_tanh(ad::rbf(v)) : v;
}

// the same on a pair (r06: packed arithmetic in the GELU epilogue of the ViT's fc1)
template <int ACT>
__device__ __forceinline__ ad::f32x2 aria_epilogue_act2(ad::f32x2 v) {
    return ACT == 1 ? ad::gelu_tanh2(ad::rbf2(v)) : v;
}

// (internal to the library: hidden, not part of the C ABI)
// v2 launcher (gemm2.hip); returns ARIA_* status
__attribute__((visibility("hidden"))) int aria_launch_gemm2(const GemmParams& p, int a_oc, int b_oc, int ntm_or_max_tm, int grid_y, void* stream);
// v3 launcher (gemm3.hip): modes 0/1, K % 64 == 0
__attribute__((visibility("hidden"))) int aria_launch_gemm3(const GemmParams& p, int a_oc, int b_oc, int ntm_or_max_tm, void* stream, void* workspace = nullptr,
                      long long workspace_bytes = 0);
// fp32 workspace bytes the v3 remainder split-K wants for a dense problem (0: it would not split)
__attribute__((visibility("hidden"))) long long aria_gemm3_workspace_bytes(long long M, long long N, long long K);
