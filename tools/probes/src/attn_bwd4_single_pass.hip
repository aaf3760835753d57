// NOT PART OF THE LIBRARY -- kept as the source of a measurement (profiles/r03_attention_notes.md, r03_attn_bwd_ab.json, r03_l2_atomics.json).
// The single-pass attention backward (5 GEMM units: dK, dV and the key block's share of dQ from ONE S / dP computation, dQ accumulated with
// global_atomic_add_f32 into an fp32 image).  Correct on MI355X (dK / dV bit-identical to the two-kernel form, dQ within 1e-3) and SLOWER than
// the two kernels at every sequence length: 139 vs 75 ms per layer at S = 65 536, 1.23 vs 0.88 ms at 8 x 2048 -- the chip retires ~330 G
// fp32 atomic adds per second whatever their scope or locality.  It was wired in as aria_attn_bwd_ws(..., workspace) at commit "bwd4
// simplified"; this file is the kernel as it sat in aria_amd/csrc/attn.hip (it uses that file's Cfg3 / tile_dma3 / frag_* helpers).
// =========================================================================================== backward v4 (hd = 128): ONE pass
// The two-kernel form computes S and dP twice (7 GEMM units of S x S x hd per head for the algorithm's 5).  Here the dK/dV workgroup of a
// 128-key block also produces the block's contribution to dQ and ADDS it to an fp32 image of dQ:
//
//     role A:  S = Q K^T -> P -> publishes P (fp32) ....................... dV += P^T dO
//     role B:  dP = dO V^T ............ dS = P (dP - delta) scale ......... dK += dS^T Q     publishes dS (bf16, [key][query])
//     all 8 waves, one barrier later:   dQ[64 q][128 f] += dS[64 q][128 keys] K[128 keys][128 f]  -- wave w owns the 32 x 32 tile
//     (queries 32 (w & 1).., features 32 (w >> 1)..), its eight K^T fragments live in registers for the whole kernel (the key block is
//     fixed), the dS fragments come out of LDS through the transposing read; 8 MFMAs + 16 global_atomic_add_f32 per wave and tile.
//
// The dQ GEMM of tile t runs at the top of iteration t + 1 (between the two barriers that are there anyway), so the loop keeps two
// barriers per 64-query tile.  16 + 16 + 8 MFMAs per wave and tile: 5 GEMM units.  The fp32 adds make dQ's summation order depend on the
// schedule (not bit-reproducible run to run): the two-kernel form stays as the deterministic mode.
template <int HD>
struct Cfg4 {
    using C3 = Cfg3<HD>;
    static constexpr int DSP = 64 * 2 + 16;  // bytes per key row of the dS image: 64 queries bf16 + 16 B (rows 36 dwords apart: the 8-byte
                                             // writes of 16 keys and the transposing reads of 4 rows x 32 B land in distinct banks)
    static constexpr int SMEM = 4 * C3::TILE + 2 * 2 * 64 * 4 + 4 * 8 * 1024 + 128 * DSP + 2 * C3::TILE;  // ... + the block's K rows
};

// one (key block, head, batch) item
template <int HD>
__device__ __forceinline__ void attn_bwd4_item(char* smem, const bf16_t* Q, const bf16_t* K, const bf16_t* V, const bf16_t* dO,
                                               const float* LSE, const float* DELTA, bf16_t* dK, bf16_t* dV, float* dQacc,
                                               const int32_t* kv_len, const uint8_t* key_mask, int Sq, int S, int H, int ldq,
                                               int ldk, int ldv, int lddo, int lddk, int lddv,
                                               int lddqa, float scale, int causal, int kblk, int head, int b) {
    using C = Cfg3<HD>;
    using C4 = Cfg4<HD>;
    static_assert(HD == 128, "the dQ tile split (2 x 4 tiles of 32 x 32 over 8 waves) is written for hd 128");
    char* sQ = smem;                                   // [2] tiles
    char* sdO = smem + 2 * C::TILE;                    // [2] tiles
    float* sLse = reinterpret_cast<float*>(smem + 4 * C::TILE);  // [2][64] (lse * log2e)
    float* sDel = sLse + 128;                          // [2][64]
    char* sP = reinterpret_cast<char*>(sDel + 128);    // [4 pairs][8][64 lanes] x 16 bytes
    char* sDS = sP + 4 * 8192;                         // [128 keys][DSP]: dS of the current tile, bf16, key-major
    char* sK = sDS + 128 * C4::DSP;                    // [2] tiles: the block's 128 K rows (B operand of the dQ GEMM, read key-strided)
    const int t = threadIdx.x, l = t & 63, w = first_lane(t >> 6), role = w >> 2, g = w & 3, h2 = l >> 5;
    const int kv0 = kblk * 128;
    const long long tok0 = (long long)b * S, tokq0 = (long long)b * Sq;
    const bf16_t* Qb = Q + tokq0 * ldq + head * HD;
    const bf16_t* dOb = dO + tokq0 * lddo + head * HD;
    const bf16_t* Kb = K + tok0 * ldk + head * HD;
    const float* lseb = LSE + ((long long)b * H + head) * Sq;
    const float* delb = DELTA + ((long long)b * H + head) * Sq;
    const int kv_wmin = kv0 + 32 * g, kv_abs = kv_wmin + (l & 31);
    const int klen = kv_len ? min(S, kv_len[b]) : S;
    const bool key_ok = kv_abs < klen && (!key_mask || key_mask[tok0 + kv_abs] != 0);
    const bool all_keys_ok = ballot(key_ok) == ~0ull;
    const float scale2 = scale * 1.4426950408889634f;

    const int q_begin = causal ? (kv0 / 64) * 64 : 0;
    const int ntiles = kv0 < klen ? (Sq - q_begin + 63) / 64 : 0;

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
    // B operand of the dQ GEMM: the block's 128 K rows stay in LDS for the whole item; the transposing read hands them out key-strided
    // (eight register-resident fragments per wave would be 32 more VGPRs than two waves per SIMD leave)
    const int qt_w = w & 1, ft_w = w >> 1;
    u32x4 rq[C::NCH], rdo[C::NCH];
    if (ntiles > 0) {
        tile_load3<HD>(rq, Kb, ldk, kv0, S, t);
        tile_load3<HD>(rdo, Kb, ldk, kv0 + 64, S, t);
        tile_store3<HD>(rq, sK, t);
        tile_store3<HD>(rdo, sK + C::TILE, t);   // (first read: behind the loop's first barrier)
    }
#pragma unroll
    for (int kk = 0; kk < C::KS; ++kk) settle(of[kk]);
    f32x16 acc[C::DT];  // dV (role A) / dK (role B): rows = keys, cols = features
#pragma unroll
    for (int i = 0; i < C::DT; ++i) acc[i] = zero_acc();
    char* myP = sP + g * 8192 + l * 16;
    // dS image: this lane's key row (written by role B), and the lane part of the transposing read of the dQ GEMM's A operand
    char* ds_row = sDS + (32 * g + (l & 31)) * C4::DSP + 8 * h2;
    const char* ds_frag = sDS + (4 * h2 + ((l & 15) >> 2)) * C4::DSP + (32 * qt_w + 16 * ((l >> 4) & 1) + 4 * (l & 3)) * 2;
    float* dq_lane = dQacc + (tokq0 + 32 * qt_w + 4 * h2) * lddqa + head * HD + 32 * ft_w + (l & 31);

    // dQ[tile at q0] += dS K for the pairs that were active on it (their rows of the dS image are valid)
    auto dq_tile = [&](int q0) {
        f32x16 dq = zero_acc();
        const bool all_pairs = !(causal && kv0 + 96 > q0 + 63);  // wave-uniform; false only on the block's two diagonal tiles
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (!all_pairs && kv0 + 32 * (kk >> 1) > q0 + 63) continue;  // pair kk / 2 sat this tile out
            const char* p = ds_frag + 16 * kk * C4::DSP;
            const s16x4 a0 = ds_read_tr16(reinterpret_cast<const bf16_t*>(p));
            const s16x4 a1 = ds_read_tr16(reinterpret_cast<const bf16_t*>(p + 8 * C4::DSP));
            s16x8 f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[e] = a0[e];
                f[4 + e] = a1[e];
            }
            dq = mfma32(f, frag_tr3<HD>(sK + (kk >> 2) * C::TILE, (kk & 3) * 16, 32 * ft_w, l), dq);
        }
        float* base = dq_lane + (long long)q0 * lddqa;
        if (q0 + 64 <= Sq) {  // (wave-uniform) every row of the tile exists: sixteen adds, no branches
#pragma unroll
            for (int r = 0; r < 16; ++r) atomic_add_f32_noret<true>(base + (long long)((r & 3) + 8 * (r >> 2)) * lddqa, dq[r]);
        } else {  // the sequence's last tile
            const int qrow = q0 + 32 * qt_w + 4 * h2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (qrow + dr < Sq) atomic_add_f32_noret<true>(base + (long long)dr * lddqa, dq[r]);
            }
        }
    };

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
        wait_vm<0>();  // this wave's LDS-DMA pieces of tile `it` have landed (and its adds of the previous tile have left)
        sync();  // tile `it` is complete in buffer it & 1; the other buffer and the P exchange are free; the dS image of tile it - 1 is complete
        const int cur = it & 1, qt0 = q_begin + it * 64;
        const bool more = it + 1 < ntiles;
        if (more) {  // next tile straight into the other buffer (its last readers finished before the barrier)
            tile_dma3<HD>(Qb, ldq, qt0 + 64, Sq - 1, sQ + (cur ^ 1) * C::TILE, w, l);
            tile_dma3<HD>(dOb, lddo, qt0 + 64, Sq - 1, sdO + (cur ^ 1) * C::TILE, w, l);
        }
        // next tile's statistics: fetched now (clamped index, no arithmetic on the value: the compiler waits at the first USE), parked in LDS
        // at the end of the iteration -- a load issued there would hold wave 0, and with it the barrier, for a memory round trip
        float lse_n = 0.f, del_n = 0.f;
        if (more && t < 64) {
            const int qn = min(qt0 + 64 + t, Sq - 1);
            lse_n = lseb[qn];
            del_n = delb[qn];
        }
        if (it > 0) dq_tile(qt0 - 64);
        const char* cQ = sQ + cur * C::TILE;
        const char* cdO = sdO + cur * C::TILE;
        const char* first = role ? cdO : cQ;   // rc operand of the first GEMM
        const char* second = role ? cQ : cdO;  // transposed operand of the second GEMM
        const bool active = !(causal && kv_wmin > qt0 + 63);  // wave-uniform: some query of the tile can see some key of this pair
        f32x16 sc[2];
        if (active) {
            sc[0] = zero_acc();
            sc[1] = zero_acc();
#pragma unroll
            for (int kk = 0; kk < C::KS; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) sc[i] = mfma32(frag_rc3<HD>(first, i * 32 + (l & 31), kk, l), of[kk], sc[i]);
            if (role == 0) {
                const float* cL = sLse + cur * 64;
                const bool need_mask = !all_keys_ok || (qt0 + 64 > Sq) || (causal && kv_wmin + 31 > qt0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const f32x4 ls = *reinterpret_cast<const f32x4*>(cL + i * 32 + 8 * rg + 4 * h2);
                        f32x4 pv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float p = exp2_fast(sc[i][4 * rg + e] * scale2 - ls[e]);
                            if (need_mask) {
                                const int q = qt0 + i * 32 + 8 * rg + 4 * h2 + e;
                                if (!(q < Sq && key_ok && !(causal && kv_abs > q))) p = 0.f;
                            }
                            sc[i][4 * rg + e] = p;
                            pv[e] = p;
                        }
                        *reinterpret_cast<f32x4*>(myP + (i * 4 + rg) * 1024) = pv;
                    }
            }
        }
        sync();  // P published; every wave is done with the dS image of tile it - 1
        if (active) {
            if (role == 1) {
                const float* cD = sDel + cur * 64;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const f32x4 pv = *reinterpret_cast<const f32x4*>(myP + (i * 4 + rg) * 1024);
                        const f32x4 dl = *reinterpret_cast<const f32x4*>(cD + i * 32 + 8 * rg + 4 * h2);
#pragma unroll
                        for (int e = 0; e < 4; ++e) sc[i][4 * rg + e] = pv[e] * (sc[i][4 * rg + e] - dl[e]) * scale;
                        // the same bf16 values that enter dK: queries i * 32 + 8 rg + 4 h2 + 0..3 of this lane's key
                        u32x2 dsv;
                        dsv[0] = pack2bf(sc[i][4 * rg], sc[i][4 * rg + 1]);
                        dsv[1] = pack2bf(sc[i][4 * rg + 2], sc[i][4 * rg + 3]);
                        *reinterpret_cast<u32x2*>(ds_row + (i * 32 + 8 * rg) * 2) = dsv;
                    }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const s16x8 pf = pack_frag(sc[i], u);
#pragma unroll
                    for (int dt = 0; dt < C::DT; ++dt) acc[dt] = mfma32(pf, frag_tr3<HD>(second, i * 32 + 16 * u, 32 * dt, l), acc[dt]);
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
    if (ntiles > 0) {
        sync();  // the dS image of the last tile is complete
        dq_tile(q_begin + (ntiles - 1) * 64);
    }
    bf16_t* out = role ? dK : dV;
    const long long ldout = role ? lddk : lddv;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kv = kv_wmin + acc_row(r, l);
            if (kv >= S) continue;
            out[(tok0 + kv) * ldout + head * HD + 32 * dt + (l & 31)] = f2bf(acc[dt][r]);
        }
}

// One workgroup per (key block, head, batch), longest key blocks first (attn_block_coords).  The adds carry device scope: measured on
// MI355X (profiles/r03_l2_atomics.json) global_atomic_add_f32 retires at the same ~330 G adds/s with or without scope bits, XCD-local or
// not -- there is no cheaper XCD-local form to schedule for.
template <int HD>
__global__ __launch_bounds__(512) void attn_bwd4_kernel(const bf16_t* Q, const bf16_t* K, const bf16_t* V, const bf16_t* dO,
                                                        const float* LSE, const float* DELTA, bf16_t* dK, bf16_t* dV, float* dQacc,
                                                        const int32_t* kv_len, const uint8_t* key_mask, int Sq, int S, int H,
                                                        int ldq, int ldk, int ldv, int lddo, int lddk, int lddv, int lddqa, float scale,
                                                        int causal, int nbatch) {
    ARIA_DYN_SMEM(smem);
    int kblk, head, b;
    if (!attn_block_coords((S + 127) / 128, H, nbatch, causal, false, kblk, head, b)) return;  // key block 0 is the longest
    attn_bwd4_item<HD>(smem, Q, K, V, dO, LSE, DELTA, dK, dV, dQacc, kv_len, key_mask, Sq, S, H, ldq, ldk, ldv, lddo, lddk, lddv, lddqa, scale,
                       causal, kblk, head, b);
}

// fp32 image -> bf16 rows (and the zero fill in front of the accumulation)
__global__ __launch_bounds__(256) void attn_dq_zero_kernel(float* acc, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        *reinterpret_cast<f32x4*>(acc + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
}
__global__ __launch_bounds__(256) void attn_dq_round_kernel(const float* acc, bf16_t* dq, long long rows, int cols, long long lda, long long lddq) {
    const int cpr = cols / 8;  // 8-element chunks per row
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < rows * cpr; i += (long long)gridDim.x * 256) {
        const long long row = i / cpr;
        const int c = int(i % cpr) * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(acc + row * lda + c), b4 = *reinterpret_cast<const f32x4*>(acc + row * lda + c + 4);
        u32x4 v;
        v[0] = pack2bf(a[0], a[1]);
        v[1] = pack2bf(a[2], a[3]);
        v[2] = pack2bf(b4[0], b4[1]);
        v[3] = pack2bf(b4[2], b4[3]);
        *reinterpret_cast<u32x4*>(dq + row * lddq + c) = v;
    }
}

