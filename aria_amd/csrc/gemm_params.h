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
    int ntn, ntm;  // column / row tile counts of the launch (256-wide tiles for v2/v3)
    // v3 remainder split-K (mode 0): tile-list positions >= split_first are each computed by `split` workgroups over disjoint K
    // ranges into fp32 slabs (ws), which a second kernel sums in a fixed order.  split <= 1: off.
    int split, split_first;
    float* ws;
    int order;     // tile-order variant (tuning knob): 0 = XCD-contiguous row-major, 1 = plain, >= 2 = groups of `order` row tiles
};

// position in the (mode 0 / 2) tile list -> (row tile, column tile)
__device__ __forceinline__ bool aria_tile_from_pos(const GemmParams& p, int tile, int& tmi, int& tn) {
    tn = tile % p.ntn;
    tmi = tile / p.ntn;
    if (p.order >= 2) {
        const int GM = p.order, per = GM * p.ntn;
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
//   mode 0 / 2: each XCD gets one contiguous chunk of the tile list.
//   mode 1 (grouped rows): the launch covers M/256 + E row tiles but the real count is data dependent and the empty ones all sit at
//     the high end, so GROUPS are dealt round-robin to the XCDs (group g -> XCD g % 8) and every XCD gets the same share of real work.
// Returns false when the workgroup has no tile.
__device__ __forceinline__ bool aria_tile_coords(const GemmParams& p, int bid, int nwg, int& tmi, int& tn) {
    if (p.mode == 1 && p.order >= 2) {
        const int GM = p.order, per = GM * p.ntn;
        const int xcd = bid & 7, idx = bid >> 3;
        const int g = (idx / per) * 8 + xcd, in = idx % per;
        if (g * GM >= p.ntm) return false;
        const int gm = min(GM, p.ntm - g * GM);
        if (in >= gm * p.ntn) return false;
        tn = in / gm;
        tmi = g * GM + in % gm;
        return true;
    }
    int tile = bid;
    if (p.order != 1) {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    return aria_tile_from_pos(p, tile, tmi, tn);
}
// number of workgroups aria_tile_coords needs
inline int aria_tile_grid(const GemmParams& p) {
    if (p.mode == 1 && p.order >= 2) {
        const int GM = p.order, groups = (p.ntm + GM - 1) / GM;
        return 8 * ((groups + 7) / 8) * GM * p.ntn;
    }
    if (p.split > 1) return p.split_first + (p.ntn * p.ntm - p.split_first) * p.split;
    return p.ntn * p.ntm;
}

// v2 launcher (gemm2.hip); returns ARIA_* status
int aria_launch_gemm2(const GemmParams& p, int a_oc, int b_oc, int ntm_or_max_tm, int grid_y, void* stream);
// v3 launcher (gemm3.hip): modes 0/1, K % 64 == 0
int aria_launch_gemm3(const GemmParams& p, int a_oc, int b_oc, int ntm_or_max_tm, void* stream, void* workspace = nullptr,
                      long long workspace_bytes = 0);
// fp32 workspace bytes the v3 remainder split-K wants for a dense problem (0: it would not split)
long long aria_gemm3_workspace_bytes(long long M, long long N, long long K);
