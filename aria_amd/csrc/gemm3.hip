// GEMM v3 for gfx950: 256x256x64 block tile, 8 waves, operands staged by LDS-DMA (global_load_lds_dwordx4) with a COUNTED
// vmcnt, and a K-step cut into four barrier-delimited phases so that exactly one of the two waves that share a SIMD is in
// its MFMA section while the other one issues LDS reads and the next DMA pieces.
//
// Why (measured on MI355X, profiles/r01_gemm_tuning.md): in v2 (register staging, one __syncthreads per K-step) the MFMA +
// fragment-read section alone costs 3350 cycles per K-step against 2048 of pure MFMA issue, and the staging section (global
// loads -> VGPRs -> ds_write_b128) adds 1900 more that do not overlap.  Here there are no staging VGPRs and no ds_write pass,
// two half-tiles stay in flight across every barrier, and the barriers alternate the two wave groups on the matrix pipe.
//
// Geometry.  Wave (wm, wn), wm = w >> 2, wn = w & 3, owns rows {a*128 + wm*64 + [0,64)} x cols {b*128 + wn*32 + [0,32)},
// a, b in {0,1}: four 64x32 quadrants, each = 2 x v_mfma_f32_32x32x16_bf16 tiles x 4 k-substeps.  A "half" of an operand
// tile is 128 rows x 64 k (16 KiB); every wave needs BOTH halves of A and of B, one quadrant per phase:
//
//   phase 1: quadrant (A0,B0)  reads B0 (4 fragments) + A0 (8)      DMA: A1 of tile t+1 -> other buffer (last read: phase 3 of t-1)
//   phase 2: quadrant (A0,B1)  reads B1 (4)                          DMA: B1 of tile t+1 -> other buffer (last read: phase 2 of t-1)
//   phase 3: quadrant (A1,B1)  reads A1 (8)                          DMA: A0 of tile t+2 -> this buffer  (last read: phase 1)
//   phase 4: quadrant (A1,B0)  reads nothing (B0 kept)               DMA: B0 of tile t+2 -> this buffer  (last read: phase 1)
//
// so each half-tile slot is refilled at least TWO phases after its last read (its readers' lgkmcnt(0) is two barriers back for either
// wave group) and every piece has at least FOUR phases (~2000 cycles, about 1 us) to land -- the loads stream from HBM when the
// operands are big (1.09 GB of expert weights), and a 2-phase budget measurably stalled there.  Two counted waits per K-tile: phase 4
// waits until A0/B0 of the next tile are in (vmcnt(8): the four newer half-tiles stay in flight), phase 1 until B1/A1 of the current
// one are (vmcnt(6)); a barrier follows each before the first read.  Up to 64 KiB per CU are in flight.
// Every phase is
//
//   ds_reads | 2 DMA pieces | s_barrier | s_waitcnt lgkmcnt(0) | s_setprio 1 | 8 MFMA | s_setprio 0 | s_barrier
//
// and waves 4-7 run one barrier behind waves 0-3, so group 0's MFMA section coincides with group 1's read/DMA section and
// the LDS latency of one group's reads hides under the other group's read issue.
//
// LDS images (128 KiB: operand x half x buffer x 16 KiB).  LDS-DMA writes lane-linearly (wave-uniform base + 16 * lane), so
// padding is impossible; bank conflicts are avoided by permuting the SOURCE granules instead:
//   rc operand ([rows][k], k contiguous): row r = 128 B; 16-byte chunk c is stored at chunk c ^ ((r >> 1) & 7) -> the 16 lanes
//     of every ds_read_b128 service group hit 16 different 16-byte bank groups.
//   oc operand ([k][rows], rows contiguous, e.g. expert weights [K][N]): k-row = 256 B; 64-byte chunk c is stored at
//     c ^ (k & 3) -> the four k-rows of a ds_read_b64_tr_b16 (hardware transpose) land in different bank quarters.
// Both permutations keep every 64/128-byte segment of a cache line inside one DMA instruction (fully coalesced).
//
// Edges.  Rows or columns beyond the edge of the tile's row group / of N are CLAMPED to the last valid one (the products land in
// accumulators that are never stored).  A reduction that is not a whole number of 64-deep tiles (grouped-K weight gradients: an
// expert's token count is arbitrary; K = 4304 in the ViT MLP) is handled in the LAST K-tile only: every DMA granule whose
// reduction index lies past the end takes a 256-byte zero page as its source address instead -- the source is per lane, so the
// LDS image simply receives zeros there and nothing else in the pipeline changes.
#include "aria_hip.h"
#include "gemm_params.h"
#include <cstdlib>
#include <type_traits>

namespace {
using namespace ad;

#ifndef ARIA_ABL
#define ARIA_ABL 0
#endif
#if ARIA_ABL & 512
// (timing experiment) per-workgroup wall-clock marks, 10 ns units: 0 entry, 1 first operands landed, 2 K loop done, 3 tile parked,
// 4 stores issued, 5 stores acknowledged -- read back with aria_abl_ts()
// (r06: 16384 workgroups by FLATTENED id, so that the per-expert grids of the weight gradients -- blockIdx.y = expert -- are sampled too)
__device__ unsigned long long aria_ts[16384 * 8];
__device__ unsigned aria_ts_hw[16384];   // where the workgroup ran: HW_ID[15:0] (wave, SIMD, pipe, CU, SH, SE) | XCC_ID << 16
__device__ __forceinline__ void ts_mark(int i) {
    const unsigned id = blockIdx.x + blockIdx.y * gridDim.x;
    if (threadIdx.x == 0 && id < 16384) {
        aria_ts[id * 8 + i] = __builtin_amdgcn_s_memrealtime();
        if (i == 0) aria_ts_hw[id] = (__builtin_amdgcn_s_getreg(4 | (15 << 11)) & 0xffffu) | ((__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xfu) << 16);
        // slots 6 / 7: the SHADER-clock counter at marks 1 / 2 (K loop start / end) -> average core clock inside the K loop
        if (i == 1 || i == 2) aria_ts[id * 8 + 5 + i] = __builtin_readcyclecounter();
    }
}
#else
__device__ __forceinline__ void ts_mark(int) {}
#endif
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int LDS_OPERAND = 65536, LDS_HALF = 32768, LDS_BUF = 16384;  // byte strides: operand (A,B) / half / buffer

// The 16 bytes lane l of wave w fetches for piece s (0 / 1) of a half-tile, as a byte offset from the operand base pointer at k = 0.
// Only two per-lane values are kept (they do not depend on the tile); the offset itself is put together from them and the tile's
// wave-uniform origin at the point of use (5 VALU operations per piece against ~250 cycles of matrix work): 2-4 VGPRs instead of 8
// precomputed offsets, and a change of tile (persistent form) touches scalar registers only.
//   rc ([rows][k]):  piece q = 2 w + s holds rows r = 8 q + (l >> 3); the 16-byte chunk is (l & 7) ^ ((r >> 1) & 7) = c0 ^ 4 s
//   oc ([k][rows]):  piece q holds k-rows k = 4 q + (l >> 4); the lane's 8 columns are chunk * 32 + (l & 3) * 8, chunk = ((l & 15) >> 2) ^ (k & 3)
template <bool OC>
struct LaneSrc {
    int a, b;  // rc: row of piece 0 inside the half (piece 1: + 8), chunk byte offset of piece 0 (piece 1: ^ 64); oc: k-row of piece 0
               // (piece 1: + 4), first column inside the half
    __device__ __forceinline__ void init(int w, int l) {
        if (!OC) {
            a = 16 * w + (l >> 3);
            b = ((l & 7) ^ (l >> 4)) * 16;
        } else {
            a = 8 * w + (l >> 4);
            b = (((l & 15) >> 2) ^ ((l >> 4) & 3)) * 32 + (l & 3) * 8;
        }
    }
    // first: first row / column of the half in the operand; limit: rows / columns of the operand (clamped: see "Edges" above);
    // ld2 = leading dimension in BYTES (< 2^24, checked by the launcher: one full-rate 24-bit multiply)
    __device__ __forceinline__ uint32_t offset(int s, int first, int limit, uint32_t ld2) const {
        if (!OC) {
            const int row = min(first + a + 8 * s, limit - 1);
            return mul24(uint32_t(row), ld2) + uint32_t(b ^ (64 * s));
        } else {
            const int col = min(first + b, limit - 8);
            return mul24(uint32_t(a + 4 * s), ld2) + 2u * uint32_t(col);
        }
    }
};

// lane-dependent part of the fragment addresses inside a half-tile image (bytes)
template <bool OC>
struct FragAddr {
    uint32_t v[4];
    __device__ __forceinline__ void init(int tile_base, int l) {  // tile_base: first row of the wave's rows inside the half
        if (!OC) {
            // chunk (2 kk + (l >> 5)) ^ ((l >> 1) & 7) = (2 kk) ^ x with x = (l >> 5) ^ ((l >> 1) & 7) (2 kk is even), and the row part is a
            // multiple of 128: the address for kk is v[0] ^ (kk << 5) -- one register instead of four
            v[0] = uint32_t((tile_base + (l & 31)) * 128 + ((((l >> 5) ^ ((l >> 1) & 7)) & 7) << 4));
            v[1] = v[2] = v[3] = 0;
        } else {
            const int k = 8 * (l >> 5) + ((l & 15) >> 2), within = 32 * ((l >> 4) & 1) + 8 * (l & 3);
#pragma unroll
            for (int i = 0; i < 2; ++i) v[i] = uint32_t(k * 256 + ((((tile_base >> 5) + i) ^ (k & 3)) << 6) + within);
            v[2] = v[3] = 0;
        }
    }
    // fragment of rows tile_base + 32 i + (l & 31), k = 16 kk + 8 (l >> 5) + 0..7
    __device__ __forceinline__ s16x8 read(const char* half, int i, int kk) const {
        if (!OC) {
            return *reinterpret_cast<const s16x8*>(half + (v[0] ^ uint32_t(kk << 5)) + i * 4096);
        } else {
            const bf16_t* p = reinterpret_cast<const bf16_t*>(half + v[i] + kk * 4096);
            const s16x4 a0 = ds_read_tr16(p);
            const s16x4 a1 = ds_read_tr16(p + 512);  // four k-rows further
            s16x8 f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[e] = a0[e];
                f[4 + e] = a1[e];
            }
            return f;
        }
    }
    // The same fragment, read by INLINE ASSEMBLY from the half image that starts BASE bytes into the workgroup's LDS (v3 K loop on
    // hardware); the CALLER waits (s_waitcnt lgkmcnt) before the first use.  hipcc's own wait insertion costs the pipeline dearly here:
    // it puts `s_waitcnt vmcnt(0)` in front of every __builtin_amdgcn_ds_read_tr16_b64 that follows an LDS-DMA in flight (it cannot
    // tell the read from the DMA's pending LDS write) -- two complete drains of the prefetch queue per K-tile in the kernels with
    // [k][n] operands -- and an lgkmcnt(0) in the middle of a phase's ds_read_b128 run (profiles/r02_gemm_tile_timeline.md).
    template <int BASE>
    __device__ __forceinline__ s16x8 read_raw(const char* smem, int i, int kk) const {
#ifdef ARIA_EMU
        return read(smem + BASE, i, kk);
#else
        constexpr int HI = BASE >= 65536 ? 65536 : 0, OFF = BASE - HI;  // (the instruction's offset field has 16 bits)
        const uint32_t lds0 = uint32_t(reinterpret_cast<uintptr_t>(smem)) + HI;  // low half of a flat LDS address = the LDS offset
        if (!OC) {
            const uint32_t a = lds0 + (v[0] ^ uint32_t(kk << 5));
            s16x8 f;
            if (i == 0)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(a), "n"(OFF));
            else
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(a), "n"(OFF + 4096));
            return f;
        } else {
            const uint32_t a = lds0 + v[i];
            s16x4 a0, a1;
            if (kk == 0)
                asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=&v"(a0), "=&v"(a1) : "v"(a), "n"(OFF), "n"(OFF + 1024));
            else if (kk == 1)
                asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=&v"(a0), "=&v"(a1) : "v"(a), "n"(OFF + 4096), "n"(OFF + 5120));
            else if (kk == 2)
                asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=&v"(a0), "=&v"(a1) : "v"(a), "n"(OFF + 8192), "n"(OFF + 9216));
            else
                asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=&v"(a0), "=&v"(a1) : "v"(a), "n"(OFF + 12288), "n"(OFF + 13312));
            s16x8 f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[e] = a0[e];
                f[4 + e] = a1[e];
            }
            return f;
        }
#endif
    }
};

// 256 zero bytes: the source of every DMA granule that lies beyond the end of a ragged reduction (last K-tile only)
__device__ const uint32_t aria_zero_page[64] = {};

struct Stage {  // everything a wave needs to issue its two DMA pieces of any half-tile
    // per launch (wave-uniform)
    int w;                     // wave id
    int limA, limB;            // rows of the A operand (M) / of the B operand (N): sources are clamped to the last one
    int bhalf;                 // distance between the first rows of the two B halves of a tile: 128, or N / 2 with the fused SwiGLU epilogue
    uint32_t ldA2, ldB2;       // leading dimensions in bytes
    long long kstepA, kstepB;  // bytes per K-tile along k
    char* lds;                 // smem + 2048 * w
    // per tile (wave-uniform)
    int nk, tail_k;  // K-tiles of this tile, valid reduction indices in the last one (64 = it is full)
    int g0;          // persistent form: index, in the workgroup's running K-tile count, of this tile's K-tile 0 (else 0)
    int m0, n0;      // first row of the tile in A / in B
    const char* gA;  // operand bases at the tile's first reduction index
    const char* gB;
    // per lane, tile-independent
    int la_a, la_b, lb_a, lb_b;
    // GATHER (K2: fc1's A rows are token rows reached through the dispatcher's index, moe_lm.py:326-334 without the permuted copy): byte
    // offsets of this lane's four A rows of the tile (half x piece) from the operand base at k = 0, chunk term included -- looked up ONCE per
    // tile (the rows of a tile do not change along k); dead fields in every other instantiation
    uint32_t ga[2][2];
    // GATHER with an output-contiguous A (the per-expert weight gradient dW1[e] = x[tok(r)]^T d_h1[r], moe_lm.py:326-334 read backwards: the
    // reduction runs over the expert's permuted rows r, whose A rows are TOKEN rows reached through the dispatcher's index -- the [6T, D]
    // permuted copy is never built).  A piece of an A half-tile = 4 reduction rows x 256 bytes of features: the lane's two rows of the tile
    // being staged, as byte offsets from the operand base (gk[CUR]: the tile phase 1 stages, gk[NXT]: the one phase 3 stages).  The indices
    // of K-tiles 0 and 1 arrive by scalar loads in the prologue, those of every later K-tile as one LDS-DMA piece (r06: idx_dma below).
    uint32_t gk[2][2];
    const int* grows;   // gather_rows at the expert's first reduction row (+ this workgroup's first K-tile)
    const char* gA0;    // the operand base (token row 0)
    char* idx_lds;      // r06: this wave's two 256-byte index slots (LDS_IDX + 512 w): the indices of K-tile j wait in slot j & 1
    // K-extension (GemmParams::ext_k): the LAST K-tile's sources.  ext = 0: off
    int ext;
    const char* eA;
    const char* eB;
    uint32_t ldeA2, ldeB2;
};

typedef int i32x8 __attribute__((ext_vector_type(8)));
// eight consecutive int32 at a WAVE-UNIFORM address as a scalar load: issue now, wait (lgkmcnt) at the point of use
__device__ __forceinline__ void sload8_issue(i32x8& r, const int* p) {
#ifdef ARIA_EMU
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = p[j];
#else
    asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(r) : "s"(p));
#endif
}
__device__ __forceinline__ void sload8_wait(i32x8& r) {
#ifndef ARIA_EMU
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r));   // (the registers pass THROUGH the wait: no use can be scheduled in front of it)
#endif
}
// this lane's two reduction rows of a tile (piece s holds rows 8 w + (l >> 4) + 4 s) -> byte offsets of their token rows
__device__ __forceinline__ void gather_k_rows(Stage& st, int which, const i32x8& r, int l) {
    const int q = l >> 4;
    const int a = q == 0 ? r[0] : q == 1 ? r[1] : q == 2 ? r[2] : r[3];
    const int b = q == 0 ? r[4] : q == 1 ? r[5] : q == 2 ? r[6] : r[7];
    st.gk[which][0] = mul24(uint32_t(a), st.ldA2);
    st.gk[which][1] = mul24(uint32_t(b), st.ldA2);
}

// r06: the gathered weight gradient's indices travel by LDS-DMA, not by scalar loads.  A scalar load counts on lgkmcnt, the counter of the
// fragment reads, and returns out of order -- the `s_waitcnt lgkmcnt(0)` in front of the NEXT phase's MFMAs therefore waited for it in full: ~0.5 us
// of index latency exposed in every K-tile (r06 timeline: 1.78 us per K-tile at 1.84 GHz against 1.64 at 1.68 GHz for the un-gathered oc,oc
// loop).  Now every wave copies the 64 indices of K-tile t + 3 into one of its two 256-byte slots as ONE more DMA piece in front of phase 1's
// two (vmcnt, in issue order: whatever wait later confirms those two has confirmed the indices), and reads its two entries of tile t + 2's
// slot between phases 1 and 2 -- an ordinary LDS read that the next fragment wait covers.  Slots: bytes [131072, 135168) of the allocation, which
// only the parked output tile uses (after the K loop).
constexpr int LDS_IDX = 2 * LDS_OPERAND;
typedef int i32x2 __attribute__((ext_vector_type(2)));
template <int SLOT>
__device__ __forceinline__ void idx_dma(const Stage& st, int tile) {   // (tile clamped: the index array is padded by one K-tile, not more)
    glds4_raw(st.grows + min(tile, st.nk - 1) * BK + lane_id(), st.idx_lds + SLOT * 256);
}
template <int SLOT>
__device__ __forceinline__ void idx_read_issue(const Stage& st, i32x2& v, int l) {   // entries 8 w + (l >> 4) and + 4: this lane's two reduction rows
#ifdef ARIA_EMU
    const int* sl = reinterpret_cast<const int*>(st.idx_lds + SLOT * 256) + 8 * st.w + (l >> 4);
    v[0] = sl[0], v[1] = sl[4];
#else
    const uint32_t a = uint32_t(reinterpret_cast<uintptr_t>(st.idx_lds)) + uint32_t(SLOT * 256 + 32 * st.w + 4 * (l >> 4));
    asm volatile("ds_read2_b32 %0, %1 offset1:4" : "=v"(v) : "v"(a));
#endif
}
__device__ __forceinline__ void idx_consume(Stage& st, int which, i32x2& v) {   // behind a wait_lds(): the registers pass through, no use moves in front
#ifndef ARIA_EMU
    asm volatile("" : "+v"(v));
#endif
    st.gk[which][0] = mul24(uint32_t(v[0]), st.ldA2);
    st.gk[which][1] = mul24(uint32_t(v[1]), st.ldA2);
}

template <bool A_OC, bool B_OC, int OPERAND, int HALF, int BUF, bool TAIL = true, int XM = 0, int GKW = -1>
__device__ __forceinline__ void stage_half(const Stage& st, int gtile) {
    // XM (compile time, so that the default kernels carry none of it): bit 0 = gathered A rows, bit 1 = K-extension tile behind the reduction
    constexpr bool GATHER = XM & 1, EXT = (XM & 2) != 0;
    constexpr bool OC = OPERAND == 0 ? A_OC : B_OC;
    const int tile = gtile - st.g0;
    const char* g = (OPERAND == 0 ? st.gA + tile * st.kstepA : st.gB + tile * st.kstepB);
    char* d = st.lds + OPERAND * LDS_OPERAND + HALF * LDS_HALF + BUF * LDS_BUF;
    LaneSrc<OC> ls;
    ls.a = OPERAND == 0 ? st.la_a : st.lb_a;
    ls.b = OPERAND == 0 ? st.la_b : st.lb_b;
    const int first = OPERAND == 0 ? st.m0 + HALF * 128 : st.n0 + HALF * st.bhalf, limit = OPERAND == 0 ? st.limA : st.limB;
    const uint32_t ld2 = OPERAND == 0 ? st.ldA2 : st.ldB2;
    const char* s0 = (GATHER && OPERAND == 0) ? g + st.ga[HALF][0] : g + ls.offset(0, first, limit, ld2);
    const char* s1 = (GATHER && OPERAND == 0) ? g + st.ga[HALF][1] : g + ls.offset(1, first, limit, ld2);
    if (GATHER && A_OC && OPERAND == 0) {   // gathered reduction rows: which row-offset pair (GKW; default: phase 1 stages half 1 = CUR, phase 3 half 0 = NXT)
        constexpr int WH = GKW >= 0 ? GKW : (HALF == 1 ? 0 : 1);
        const uint32_t col2 = 2u * uint32_t(min(first + ls.b, limit - 8));
        s0 = st.gA0 + st.gk[WH][0] + col2;
        s1 = st.gA0 + st.gk[WH][1] + col2;
    }
    // (the K-extension tile ALWAYS goes through here, also when it is a full 64 deep: its sources are the adapter's operands -- ADVICE r5)
    if (TAIL && (st.tail_k < BK || (EXT && st.ext > 0)) && tile == st.nk - 1) {  // wave-uniform: the ragged end of the reduction -> granules past it read zeros
        if (EXT && !(GATHER && OPERAND == 0)) {  // K-extension tile: the adapter's operands, same lane <-> (row, reduction index) map
            const char* e = OPERAND == 0 ? st.eA : st.eB;
            const uint32_t le2 = OPERAND == 0 ? st.ldeA2 : st.ldeB2;
            s0 = e + ls.offset(0, first, limit, le2);
            s1 = e + ls.offset(1, first, limit, le2);
        }
        // first reduction index (inside the tile) of this lane's 16 bytes, pieces 0 and 1
        const int k0 = OC ? ls.a : ls.b >> 1, k1 = OC ? ls.a + 4 : (ls.b ^ 64) >> 1;
        const int l = lane_id();
        if (k0 >= st.tail_k) s0 = reinterpret_cast<const char*>(aria_zero_page) + 16 * (l & 15);
        if (k1 >= st.tail_k) s1 = reinterpret_cast<const char*>(aria_zero_page) + 16 * (l & 15);
    }
    glds16(s0, d);
    glds16(s1, d + 1024);
}

template <bool A_OC, bool B_OC, bool SPLIT_GLU = false, class P>
__device__ __forceinline__ void stage_init(Stage& st, const P& p, int w, int l, char* smem) {
    st.w = w;
    st.limA = p.M;
    st.limB = p.N;
    st.bhalf = p.glu ? p.N / 2 : 128;
    if (SPLIT_GLU) {  // (gemm3_kernel<.., .., 6>) gate / up rows in two tensors of one allocation: the up rows lie glu_up_rows B-operand rows
        st.bhalf = p.glu_up_rows;  // behind the gate rows; no column edge in glu mode (I % 128 == 0), so the row clamp is not needed
        st.limB = 0x7fffffff;
    }
    st.ldA2 = uint32_t(2 * p.lda);
    st.ldB2 = uint32_t(2 * p.ldb);
    st.kstepA = A_OC ? 2 * BK * p.lda : 2 * BK;
    st.kstepB = B_OC ? 2 * BK * p.ldb : 2 * BK;
    st.lds = smem + 2048 * w;
    LaneSrc<A_OC> la;
    LaneSrc<B_OC> lb;
    la.init(w, l);
    lb.init(w, l);
    st.la_a = la.a, st.la_b = la.b, st.lb_a = lb.a, st.lb_b = lb.b;
}

// One phase: quadrant (QA, QB) of the K-tile in buffer BUF.  SO/SH/SB: operand, half, buffer of the DMA issued here.
// EDGE (block-uniform): the tile hangs over the edge of its row group / of N; 32-row and 32-column MFMA tiles that are wholly
// outside (wave-uniform tests on rows_left / cols_left, counted from the wave's first row / column) are skipped together with
// their fragment reads -- a grouped GEMM's last row tile per expert usually holds only a few rows.
// STEADY (interior tiles only): the K-tile this phase stages is neither beyond the reduction nor its ragged last one, and more pieces
// are in flight behind every wait -- the phase is straight-line code (the general form spends two scalar branches, a handful of selects
// and a chain of compares per phase on conditions that only change in a tile's last three K-tiles).
template <bool A_OC, bool B_OC, int QA, int QB, bool LOAD_A, bool LOAD_B, int BUF, int SO, int SH, int SB, int WAIT, bool EDGE,
          bool STEADY = false, int XM = 0>
__device__ __forceinline__ void phase(f32x16 (&acc)[2][2][2], s16x8 (&fa)[2][4], s16x8 (&fb)[2][4], const FragAddr<A_OC>& aa,
                                      const FragAddr<B_OC>& ab, const char* smem, const Stage& st, int stage_tile, bool do_stage,
                                      bool more_in_flight, int rows_left, int cols_left) {
    static_assert(!(STEADY && EDGE), "the steady form is for interior tiles");
    constexpr bool GK = (XM & 1) && A_OC;   // gathered reduction rows: phase 1 also copies tile t + 3's indices (stage_tile = t + 1) into slot SB
    const bool col_ok = !EDGE || QB * 128 < cols_left;
    const bool row_ok[2] = {!EDGE || QA * 128 < rows_left, !EDGE || QA * 128 + 32 < rows_left};
    if (!(ARIA_ABL & 2) && LOAD_B && (!EDGE || (col_ok && rows_left > 0))) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fb[QB][kk] = ab.template read_raw<LDS_OPERAND + QB * LDS_HALF + BUF * LDS_BUF>(smem, 0, kk);
    }
    sched_fence();
    if (!(ARIA_ABL & 2) && LOAD_A) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (row_ok[i]) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fa[i][kk] = aa.template read_raw<QA * LDS_HALF + BUF * LDS_BUF>(smem, i, kk);
            }
    }
    sched_fence();
    // DMA placement (compile time): with the LDS reads, before the barrier -- or LATE, inside the MFMA section behind the first two
    // MFMAs, where the pieces' issue cost hides under the matrix pipe (this phase's two pieces are then not yet issued at the wait).
    // LATE measures +4 % with two k-contiguous operands and +1..2 % with transposing reads (round 1 had -5 % there: that was the
    // compiler's vmcnt(0) in front of every transposing read, see FragAddr::read_raw)
    constexpr bool LATE = true;
    static_assert(LATE || !GK, "the index piece is counted for the LATE placement only");
    if (!LATE) {
        if (STEADY)
            stage_half<A_OC, B_OC, SO, SH, SB, false, XM>(st, stage_tile);
        else if (do_stage)
            stage_half<A_OC, B_OC, SO, SH, SB, true, XM>(st, stage_tile);
    }
    if (!(ARIA_ABL & 32) && WAIT == 1) {  // phase 1: B1 and A1 of THIS tile must have landed; newer = A0, B0 (and, early placement, A1) of the next tile
        if (STEADY || more_in_flight)
            wait_vm<LATE ? 4 : 6>();
        else
            wait_vm<0>();
    }
    if (!(ARIA_ABL & 32) && WAIT == 4) {  // phase 4: A0 and B0 of the NEXT tile must have landed; newer = its A1, B1 and A0 (, B0) of the tile after
        if (STEADY || more_in_flight)
            wait_vm<(LATE ? 6 : 8) + (GK ? 1 : 0)>();   // (GK: this tile's phase 1 issued an index piece in front of its two)
        else
            wait_vm<0>();
    }
    if (!(ARIA_ABL & 8)) raw_barrier();
    wait_lds();  // the fragment reads above (invisible to the compiler's own counting)
    sched_fence();
    wave_prio<1>();
    if (!EDGE) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (!(ARIA_ABL & 1)) acc[QA][i][QB] = mfma32(fb[QB][kk], fa[i][kk], acc[QA][i][QB]);
            if (LATE && kk == 0) {
                sched_fence();
                if (ARIA_ABL & 4) {
                    // (timing experiment: no DMA)
                } else if (STEADY) {
                    if (GK && WAIT == 1) idx_dma<SB>(st, stage_tile + 2);
                    stage_half<A_OC, B_OC, SO, SH, SB, false, XM>(st, stage_tile);
                } else if (do_stage) {
                    if (GK && WAIT == 1) idx_dma<SB>(st, stage_tile + 2);
                    stage_half<A_OC, B_OC, SO, SH, SB, true, XM>(st, stage_tile);
                }
                sched_fence();
            }
        }
    } else {
        if (LATE && do_stage) {
            if (GK && WAIT == 1) idx_dma<SB>(st, stage_tile + 2);
            stage_half<A_OC, B_OC, SO, SH, SB, true, XM>(st, stage_tile);
        }
        if (col_ok) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (row_ok[i]) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) acc[QA][i][QB] = mfma32(fb[QB][kk], fa[i][kk], acc[QA][i][QB]);
                }
        }
    }
    wave_prio<0>();
    if (!(ARIA_ABL & 8)) raw_barrier();
}

template <bool A_OC, bool B_OC, int BUF, bool EDGE, bool STEADY = false, int XM = 0>
__device__ __forceinline__ void k_tile(f32x16 (&acc)[2][2][2], s16x8 (&fa)[2][4], s16x8 (&fb)[2][4], const FragAddr<A_OC>& aa,
                                       const FragAddr<B_OC>& ab, const char* smem, Stage& st, int t, int nk, int rl, int cl) {
    const bool n1 = STEADY || t + 1 < nk, n2 = STEADY || t + 2 < nk;   // (steady K-tiles stage t + 1 and t + 2 by construction: no tests left)
    constexpr bool GK = (XM & 1) && A_OC;   // gathered reduction rows: tile t + 2's indices wait in slot BUF (copied there by tile t - 1's phase 1)
    i32x2 gi;
    phase<A_OC, B_OC, 0, 0, true, true, BUF, 0, 1, BUF ^ 1, 1, EDGE, STEADY, XM>(acc, fa, fb, aa, ab, smem, st, t + 1, n1, n1, rl, cl);
    if (GK && n2) idx_read_issue<BUF>(st, gi, lane_id());
    phase<A_OC, B_OC, 0, 1, false, true, BUF, 1, 1, BUF ^ 1, 0, EDGE, STEADY, XM>(acc, fa, fb, aa, ab, smem, st, t + 1, n1, false, rl, cl);
    if (GK && n2) idx_consume(st, 1, gi);   // (phase 2's fragment wait covered the read)
    phase<A_OC, B_OC, 1, 1, true, false, BUF, 0, 0, BUF, 0, EDGE, STEADY, XM>(acc, fa, fb, aa, ab, smem, st, t + 2, n2, false, rl, cl);
    phase<A_OC, B_OC, 1, 0, false, false, BUF, 1, 0, BUF, 4, EDGE, STEADY, XM>(acc, fa, fb, aa, ab, smem, st, t + 2, n2, n2, rl, cl);
    if (GK) st.gk[0][0] = st.gk[1][0], st.gk[0][1] = st.gk[1][1];
}

template <bool A_OC, bool B_OC, bool EDGE, int XM = 0>
__device__ __forceinline__ void k_loop3(f32x16 (&acc)[2][2][2], s16x8 (&fa)[2][4], s16x8 (&fb)[2][4], const FragAddr<A_OC>& aa,
                                        const FragAddr<B_OC>& ab, const char* smem, Stage& st, int nk, int rl, int cl) {
    int kt = 0;
    if (!EDGE) {
        // steady part: K-tile t stages tiles t + 1 and t + 2, both of which must exist and be full -- t + 2 <= last full tile
        const int last_full = (st.tail_k < BK || ((XM & 2) && st.ext > 0)) ? nk - 2 : nk - 1;  // (an extension tile is never a steady one)
        for (; kt + 1 <= last_full - 2; kt += 2) {
            k_tile<A_OC, B_OC, 0, false, true, XM>(acc, fa, fb, aa, ab, smem, st, kt, nk, rl, cl);
            k_tile<A_OC, B_OC, 1, false, true, XM>(acc, fa, fb, aa, ab, smem, st, kt + 1, nk, rl, cl);
        }
    }
    for (; kt + 1 < nk; kt += 2) {
        k_tile<A_OC, B_OC, 0, EDGE, false, XM>(acc, fa, fb, aa, ab, smem, st, kt, nk, rl, cl);
        k_tile<A_OC, B_OC, 1, EDGE, false, XM>(acc, fa, fb, aa, ab, smem, st, kt + 1, nk, rl, cl);
    }
    if (kt < nk) k_tile<A_OC, B_OC, 0, EDGE, false, XM>(acc, fa, fb, aa, ab, smem, st, kt, nk, rl, cl);
}

// ---- epilogues.  r06: the accumulators are held TRANSPOSED -- the K loop issues mfma(B fragment, A fragment), so register 4 q + j of lane
// (c = l & 31, h = l >> 5) of accumulator tile (a, i, b) is the output element
//     row  a * 128 + wm * 64 + i * 32 + c,      column  b * 128 + wn * 32 + 8 q + 4 h + j      (q, j = 0..3)
// i.e. a lane owns FOUR CONSECUTIVE COLUMNS of one row per q: two v_cvt_pk_bf16_f32 and one 8-byte LDS store park them.  With the natural
// operand order a lane owned 16 rows of ONE column and every pair of values cost an exchange with the neighbour lane (one DPP move + three
// selects) in front of its conversion: 320 vector instructions and 64 four-byte LDS stores per wave and tile against 64 + 32 now -- the
// packing was ~2 us of every tile's ~4.5 us epilogue (r06 tile timeline: "K loop end -> parked" 2.4-3.2 us at 1.6 GHz, vector-ALU bound).
// The products and their summation order per output element are the same (the emulator and the bit-identity cases against the 128 x 128
// and the register-staged kernels hold on hardware).
template <int ACT, class P>
__device__ __forceinline__ void store_tile3(const P& p, const f32x16 (&acc)[2][2][2], char* C, int m0, int m_end, int n0, int l,
                                            int wm, int wn) {
    const int c = l & 31, h = l >> 5;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + b * 128 + wn * 32 + 8 * q + 4 * h;
            float bv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = (p.bias && n + j < p.N) ? bf2f(p.bias[n + j]) : 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = m0 + a * 128 + wm * 64 + i * 32 + c;
                    if (m >= m_end) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (n + j >= p.N) continue;
                        const float v = aria_epilogue_act_c<ACT>(acc[a][i][b][4 * q + j] + bv[j]);
                        if (p.c_f32) {
                            float* d = reinterpret_cast<float*>(C) + (long long)m * p.ldc + n + j;
                            *d = p.accumulate ? *d + v : v;
                        } else {
                            bf16_t* d = reinterpret_cast<bf16_t*>(C) + (long long)m * p.ldc + n + j;
                            *d = f2bf(p.accumulate ? v + bf2f(*d) : v);
                        }
                    }
                }
        }
}

// this lane's four bias values of column group (b, q), or zeros
template <class P>
__device__ __forceinline__ void bias4(const P& p, int n, float (&bv)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = 0.f;
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n + j < p.N) bv[j] = bf2f(p.bias[n + j]);
    }
}

// Row form of the wide epilogue: the WHOLE workgroup parks the 256 x 256 tile in LDS ([256 rows][512 + 16 bytes]) and, after one barrier, wave w
// writes rows 32 w .. 32 w + 31 -- every store instruction covers two complete 512-byte tile rows (four full cache lines each).  (The per-wave
// form of round 2 -- sixteen 64-byte row pieces per instruction -- left the library in r06; ARIA_GEMM_WIDE_STORE=1 selects this form too.)
constexpr int ROWP3 = 528;  // LDS row pitch of the parked tile (bytes)
// parks the tile (bias, activation, rounding); OLD: the slots already hold the previous C tile -- each lane adds its values to the old four IN
// FP32 and rounds once
template <int ACT, bool OLD, class P>
__device__ __forceinline__ void park_tile3(const P& p, const f32x16 (&acc)[2][2][2], int n0, int l, int wm, int wn, char* smem) {
    const int c = l & 31, h = l >> 5;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = b * 128 + wn * 32 + 8 * q + 4 * h;
            float bv[4];
            bias4(p, n0 + col, bv);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = a * 128 + wm * 64 + i * 32 + c;  // inside the tile
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const f32x2 pr = aria_epilogue_act2<ACT>(f32x2{acc[a][i][b][4 * q + j] + bv[j], acc[a][i][b][4 * q + j + 1] + bv[j + 1]});
                        v[j] = pr.x, v[j + 1] = pr.y;
                    }
                    u32x2* slot = reinterpret_cast<u32x2*>(smem + row * ROWP3 + col * 2);
                    if (OLD) {
                        const u32x2 ov = *slot;
                        v[0] += bflo(ov[0]), v[1] += bfhi(ov[0]), v[2] += bflo(ov[1]), v[3] += bfhi(ov[1]);
                    }
                    *slot = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                }
        }
}

template <int ACT, class P>
__device__ __forceinline__ void store_tile3_rows(const P& p, const f32x16 (&acc)[2][2][2], char* C, int m0, int m_end, int n0, int l, int w,
                                                 int wm, int wn, char* smem) {
    park_tile3<ACT, false>(p, acc, n0, l, wm, wn, smem);
    sync();
    ts_mark(3);
    const int rr = l >> 5, cc = (l & 31) * 8;  // two rows per instruction, 32 lanes x 16 bytes each
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
        const int row = w * 32 + s2 * 2 + rr;
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * ROWP3 + cc * 2);
        const int m = m0 + row;
        if (m < m_end) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(C) + (long long)m * p.ldc + n0 + cc) = v;
    }
    if (ARIA_ABL & 512) {
        ts_mark(4);
        wait_vm<0>();
        ts_mark(5);
    }
}

// The row form for what used to fall back to the 4-byte-per-lane path (r05b): `accumulate` (C += tile: the frozen ViT's residual adds live in
// its out_proj / fc2 epilogues, vision.py forward_frozen -- 58 launches per step whose narrow read-modify-write epilogue was a third of an
// out_proj tile's time) and column tiles that hang over N (N = 1152 / 3456 / 4304 are 4.5 / 13.5 / 16.8 tiles wide).  The OLD tile comes in
// first, by the same complete-row mapping the write-out uses (16 coalesced 16-byte loads per lane instead of 64 strided dword loads), and
// waits in the LDS where the new values will be parked; each lane then adds its accumulators to the old values IN FP32 and rounds once --
// the narrow path's arithmetic, bit for bit -- and the rows leave as in store_tile3_rows.  N % 8 == 0 (the caller checks).
template <int ACT, class P>
__device__ __forceinline__ void store_tile3_rows_gen(const P& p, const f32x16 (&acc)[2][2][2], char* C, int m0, int m_end, int n0, int l, int w,
                                                     int wm, int wn, char* smem) {
    const int rr = l >> 5, cc = (l & 31) * 8;  // two rows per instruction, 32 lanes x 16 bytes each
    const bool col_in = n0 + cc < p.N;         // this lane's 8 columns exist (N % 8 == 0)
    const bool old = p.accumulate != 0;
    if (old) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x4 o[8];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                const int row = w * 32 + (half * 8 + s2) * 2 + rr, m = m0 + row;
                o[s2] = (m < m_end && col_in) ? ld16(reinterpret_cast<const bf16_t*>(C) + (long long)m * p.ldc + n0 + cc) : zero16();
            }
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                const int row = w * 32 + (half * 8 + s2) * 2 + rr;
                *reinterpret_cast<u32x4*>(smem + row * ROWP3 + cc * 2) = o[s2];
            }
        }
        sync();
        park_tile3<ACT, true>(p, acc, n0, l, wm, wn, smem);
    } else {
        park_tile3<ACT, false>(p, acc, n0, l, wm, wn, smem);
    }
    sync();
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
        const int row = w * 32 + s2 * 2 + rr;
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * ROWP3 + cc * 2);
        const int m = m0 + row;
        if (m < m_end && col_in) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(C) + (long long)m * p.ldc + n0 + cc) = v;
    }
}

// Row form of the epilogue for the fused wqkv projection (K7): the parked 256 x 256 tile leaves as complete 512-byte rows like
// store_tile3_rows; a lane's 16 bytes are four interleaved (x[2i], x[2i+1]) pairs of ONE head, so the rotation is lane-local:
// one 16-byte load of (cos, sin) x 4 from the bf16 freqs_cis row of the token's position per row piece.
template <class P>
__device__ __forceinline__ void store_tile3_rows_rope(const P& p, const f32x16 (&acc)[2][2][2], char* C, int m0, int m_end, int n0, int l, int w,
                                                      int wm, int wn, char* smem) {
    park_tile3<0, false>(p, acc, n0, l, wm, wn, smem);   // (no bias in this launch: the entry point rejects it)
    sync();
    const int region = n0 / p.rope_D;                 // 0 q, 1 k, 2 v (tile-uniform)
    const int col0 = n0 - region * p.rope_D;          // first column of the tile inside its block
    const int rr = l >> 5, cc = (l & 31) * 8;
    const int fcol = (col0 + cc) % p.rope_hd;         // position of the lane's 8 columns inside their head
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
        const int row = w * 32 + s2 * 2 + rr;
        u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * ROWP3 + cc * 2);
        const int m = m0 + row;
        if (m < m_end) {
            const int seq = m / p.rope_S;
            const int ps = p.rope_pos ? p.rope_pos[m] : m - seq * p.rope_S;
            if (region < 2) {
                const u32x4 f = ld16(p.rope_fc + (long long)ps * p.rope_hd + fcol);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x0 = bflo(v[q]), x1 = bfhi(v[q]), cs = bflo(f[q]), sn = bfhi(f[q]);
                    v[q] = pack2bf(x0 * cs - x1 * sn, x1 * cs + x0 * sn);
                }
            }
            bf16_t* dst;
            if (region == 0)
                dst = reinterpret_cast<bf16_t*>(C) + (long long)m * p.ldc + col0 + cc;
            else
                dst = static_cast<bf16_t*>(region == 1 ? p.kc : p.vc) + ((long long)seq * p.cache_S + ps) * p.ld_cache + col0 + cc;
            // a caller-supplied position outside the cache never becomes a store (a stray pos corrupted memory silently)
            if (region == 0 || unsigned(ps) < unsigned(p.cache_S)) *reinterpret_cast<u32x4*>(dst) = v;
        }
    }
}

// The same for the HF layer (LlamaAttention.forward, modeling_llama.py:243-281 through moe_lm.py:594): q | k | v as ONE wide projection whose q
// and k column tiles leave ROTATED -- apply_rotary_pos_emb's half-split form q cos + rotate_half(q) sin (modeling_llama.py:130-160) with the
// cos / sin tables [S, hd] (emb = cat(freqs, freqs)) of the token's position t % S.  A 256-column tile is two whole heads (hd = 128): the
// partner x[i +- hd / 2] of a lane's 8 columns lies in the SAME parked row, 128 bytes away -- one more 16-byte LDS read instead of a second
// pass over [T, 2 D] (rope_kernel: 55 us per layer).  Arithmetic = rope_kernel's, rounding for rounding: bf16(bf16(x cos) + bf16(+-x' sin)).
template <class P>
__device__ __forceinline__ void store_tile3_rows_rope_hf(const P& p, const f32x16 (&acc)[2][2][2], char* C, int m0, int m_end, int n0, int l, int w,
                                                         int wm, int wn, char* smem) {
    park_tile3<0, false>(p, acc, n0, l, wm, wn, smem);
    sync();
    const bool rotate = n0 < 2 * p.rope_D;            // q and k column tiles (tile-uniform); v leaves as it is
    const int rr = l >> 5, cc = (l & 31) * 8;
    const int half = p.rope_hd >> 1;
    const int fcol = (n0 + cc) % p.rope_hd;           // position of the lane's 8 columns inside their head
    const bool second = fcol >= half;
    const int pc = second ? cc - half : cc + half;    // the partner's columns: the other half of the same head, same parked row
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
        const int row = w * 32 + s2 * 2 + rr;
        u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * ROWP3 + cc * 2);
        const int m = m0 + row;
        if (m < m_end) {
            if (rotate) {
                const u32x4 pv = *reinterpret_cast<const u32x4*>(smem + row * ROWP3 + pc * 2);
                const int ps = m % p.rope_S;
                const u32x4 fc = ld16(p.rope_fc + (long long)ps * p.rope_hd + fcol), fs = ld16(p.rope_sn + (long long)ps * p.rope_hd + fcol);
                const float sg = second ? 1.f : -1.f;   // rotate_half = cat(-x2, x1)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    v[q] = pack2bf(rbf(bflo(v[q]) * bflo(fc[q])) + rbf(sg * bflo(pv[q]) * bflo(fs[q])),
                                   rbf(bfhi(v[q]) * bfhi(fc[q])) + rbf(sg * bfhi(pv[q]) * bfhi(fs[q])));
            }
            *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(C) + (long long)m * p.ldc + n0 + cc) = v;
        }
    }
}

template <int ACT, class P>
__device__ __forceinline__ void store_any3(const P& p, const f32x16 (&acc)[2][2][2], char* C, int m0, int m_end, int n0, int l, int w, int wm,
                                           int wn, char* smem) {
    if (!p.c_f32 && !p.accumulate && n0 + BN <= p.N && p.wide_store)
        store_tile3_rows<ACT>(p, acc, C, m0, m_end, n0, l, w, wm, wn, smem);
    else if (!p.c_f32 && p.wide_store && !(p.N & 7) && !(ARIA_ABL & 4096))   // accumulate and / or a column tile that hangs over N
        store_tile3_rows_gen<ACT>(p, acc, C, m0, m_end, n0, l, w, wm, wn, smem);
    else
        store_tile3<ACT>(p, acc, C, m0, m_end, n0, l, wm, wn);
}

// Per-wave parking blocks of the two SwiGLU epilogues: [64 rows][64 bytes], a lane's 8 bytes (row, columns 8 q + 4 h .. + 3) at 16-byte pair
// q ^ ((row >> 1) & 3), half h -- the 16 lanes of an 8-byte LDS store (16 consecutive rows) then meet two to a bank instead of eight, and a
// 16-byte read of pair P of a row finds it at pair P ^ ((row >> 1) & 3) (the four lanes of a row still cover its 64 bytes: conflict-free)
__device__ __forceinline__ int wave_blk_off(int row, int pair) { return row * 64 + ((pair ^ ((row >> 1) & 3)) << 4); }

// Fused SwiGLU epilogue (p.glu): accumulator block b = 0 holds gate columns n0 + wn*32 + .., block b = 1 the up columns I + the same, so a
// lane owns gate and up of the same output elements.  Rounding points as the unfused chain materialises them (moe_lm.py:505-507 on bf16
// tensors: h = fc1(x) rounded to bf16, silu(h_gate) rounded, the product rounded): act = bf16(bf16(silu(bf16(gate))) * bf16(up)).
// Everything leaves through the wide path (three 4 KiB blocks per 64-row group in the wave's 16 KiB of idle LDS): act -> C2, and, if C
// is given, the two halves of h -> C.
__device__ __forceinline__ float silu3(float a) { return silu_fast(a); }

template <class P>
__device__ __forceinline__ void store_tile3_glu(const P& p, const f32x16 (&acc)[2][2][2], char* C, int m0, int m_end, int n0, int l, int w,
                                                int wm, int wn, char* smem) {
    const int c = l & 31, h = l >> 5;
    char* mine = smem + 16384 * w;  // [gate | up | act][64 rows][64 bytes]
    const int rr = l >> 2, cc = (l & 3) * 8;
    const int I = p.N / 2;
    bf16_t* H = reinterpret_cast<bf16_t*>(C);
    bf16_t* ACT = static_cast<bf16_t*>(p.C2);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        wave_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float g[4], u[4], y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    g[j] = rbf(acc[a][i][0][4 * q + j]);
                    u[j] = rbf(acc[a][i][1][4 * q + j]);
                    y[j] = rbf(silu3(g[j])) * u[j];
                }
                const int off = wave_blk_off(i * 32 + c, q) + 8 * h;
                *reinterpret_cast<u32x2*>(mine + off) = u32x2{pack2bf(g[0], g[1]), pack2bf(g[2], g[3])};
                *reinterpret_cast<u32x2*>(mine + 4096 + off) = u32x2{pack2bf(u[0], u[1]), pack2bf(u[2], u[3])};
                *reinterpret_cast<u32x2*>(mine + 8192 + off) = u32x2{pack2bf(y[0], y[1]), pack2bf(y[2], y[3])};
            }
        wave_barrier();
#pragma unroll
        for (int s16 = 0; s16 < 4; ++s16) {
            const int row = s16 * 16 + rr;
            const int m = m0 + a * 128 + wm * 64 + row;
            const int n = n0 + wn * 32 + cc;
            const int off = wave_blk_off(row, l & 3);
            const u32x4 vg = *reinterpret_cast<const u32x4*>(mine + off);
            const u32x4 vu = *reinterpret_cast<const u32x4*>(mine + 4096 + off);
            const u32x4 vy = *reinterpret_cast<const u32x4*>(mine + 8192 + off);
            if (m < m_end) {
                if (H) {
                    *reinterpret_cast<u32x4*>(H + (long long)m * p.ldc + n) = vg;
                    *reinterpret_cast<u32x4*>(H + (long long)m * p.ldc + I + n) = vu;
                }
                *reinterpret_cast<u32x4*>(ACT + (long long)m * p.ldc2 + n) = vy;
            }
        }
    }
    if (ARIA_ABL & 512) {  // (timing build: marks 3 / 4 = stores issued, 5 = stores acknowledged)
        ts_mark(3);
        ts_mark(4);
        wait_vm<0>();
        ts_mark(5);
    }
}

// Fused SwiGLU-backward epilogue (gemm3_kernel<.., .., 5>; GroupedMLP's glu backward moe_lm.py:505-507 behind experts.fc2's input
// gradient, and the shared expert's): the tile is d_act = dY W2^T for columns n0 .. n0 + 255 of I.  Every wave parks its 128 x 64 block
// as bf16 in its own 16 KiB of the idle operand images (the rounding point of the two-step chain's d_act tensor), then walks it in 16-byte
// row pieces: gate and up of the same 8 elements come from H (two 16-byte loads per piece, the 16 of a 128-row half in flight before the
// first use -- the accumulators are dead by then), swiglu_bwd_elem gives d_gate / d_up, two 16-byte stores.  Saves the d_act round trip
// (write + read of M x I bf16) and a launch per GEMM; bit-identical to aria_swiglu_bwd on the unfused product.
// Column blocks of 128 are all-or-nothing (I % 128 == 0, checked by the entry point); rows past m_end are predicated off.
// (r06 timeline + ISA count: this epilogue is VECTOR-ALU bound, not memory bound -- ~2400 vector instructions per wave, 256 of them quarter-rate
// exponentials / reciprocals, two waves per SIMD = ~15 us per tile at 1.6 GHz; the request schedule of the H loads does not show.)
template <class P>
__device__ __forceinline__ void store_tile3_dglu(const P& p, const f32x16 (&acc)[2][2][2], char* C, int m0, int m_end, int n0, int l, int w,
                                                 int wm, int wn, char* smem) {
    const int c = l & 31, h = l >> 5;
    char* mine = smem + 16384 * w;  // [a][b][64 rows][64 bytes]
    const int I = p.N;
    const int rr = l >> 2, cc = (l & 3) * 8;  // this lane's row inside a 16-row slab / first of its 8 columns
    const bf16_t* H = p.H;
    bf16_t* DH = reinterpret_cast<bf16_t*>(C);
    // r05: the forward's [gate | up] values do not depend on the accumulators -- 20 of the 32 16-byte loads per lane (half 0 + the first quarter's half
    // of half 1) are requested right behind the parking, when the 128 accumulator registers are dead, and every finished quarter (32 registers
    // freed) requests the next 8: one exposed HBM round trip for the whole tile, half 1's rows land under half 0's arithmetic and stores.
    // (Requesting a part BEFORE the parking, beside the live accumulators, spilled 24 registers; 24 loads behind it spilled 16; 20 is spill-free.)
    u32x4 vg[2][2][4], vu[2][2][4];
    auto load_q = [&](int a, int b, int s0, int s1) {  // pieces s0 .. s1-1 of quarter (a, b) = this wave's 64 rows x 32 columns of a 128 x 128 block
        if (n0 + b * 128 < I) {  // block-uniform
#pragma unroll
            for (int s16 = s0; s16 < s1; ++s16) {
                const int m = min(m0 + a * 128 + wm * 64 + s16 * 16 + rr, m_end - 1);  // clamped: loaded, never stored
                const bf16_t* src = H + (long long)m * p.ldh + n0 + b * 128 + wn * 32 + cc;
                vg[a][b][s16] = ld16(src);
                vu[a][b][s16] = ld16(src + I);
            }
        }
    };
    wave_barrier();
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<u32x2*>(mine + (a * 2 + b) * 4096 + wave_blk_off(i * 32 + c, q) + 8 * h) =
                        u32x2{pack2bf(acc[a][i][b][4 * q], acc[a][i][b][4 * q + 1]), pack2bf(acc[a][i][b][4 * q + 2], acc[a][i][b][4 * q + 3])};
    wave_barrier();
    ts_mark(3);
    sched_fence();  // (half 1's loads must not be hoisted above the parking: the accumulators are still live there)
    load_q(0, 0, 0, 4);
    load_q(0, 1, 0, 4);
    load_q(1, 0, 0, 2);
    sched_fence();
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (n0 + b * 128 < I) {
#pragma unroll
                for (int s16 = 0; s16 < 4; ++s16) {
                    const int row = s16 * 16 + rr;
                    const u32x4 d = *reinterpret_cast<const u32x4*>(mine + (a * 2 + b) * 4096 + wave_blk_off(row, l & 3));
                    u32x4 og, ou;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x2 da, db;
                        swiglu_bwd_pair(f32x2{bflo(vg[a][b][s16][q]), bfhi(vg[a][b][s16][q])}, f32x2{bflo(vu[a][b][s16][q]), bfhi(vu[a][b][s16][q])},
                                        f32x2{bflo(d[q]), bfhi(d[q])}, da, db);
                        og[q] = pack2bf(da.x, da.y);
                        ou[q] = pack2bf(db.x, db.y);
                    }
                    const int m = m0 + a * 128 + wm * 64 + row;
                    if (m < m_end) {
                        bf16_t* dst = DH + (long long)m * p.ldc + n0 + b * 128 + wn * 32 + cc;
                        *reinterpret_cast<u32x4*>(dst) = og;
                        *reinterpret_cast<u32x4*>(dst + I) = ou;
                    }
                }
            }
            sched_fence();
            // rolling requests: a finished quarter frees 32 registers -> the next 8 loads (20 in flight at the peak)
            if (a == 0 && b == 0) {
                load_q(1, 0, 2, 4);
                load_q(1, 1, 0, 2);
                sched_fence();
            }
            if (a == 0 && b == 1) {
                load_q(1, 1, 2, 4);
                sched_fence();
            }
        }
    }
    if (ARIA_ABL & 512) {
        ts_mark(4);
        wait_vm<0>();
        ts_mark(5);
    }
}

template <bool A_OC, bool B_OC, int VER>
__global__ __launch_bounds__(512) void gemm3_kernel(GemmParams p) {
    ARIA_DYN_SMEM(smem);
    const int t = threadIdx.x, l = t & 63, w = first_lane(t >> 6), wm = w >> 2, wn = w & 3;
    ts_mark(0);
    // (round 2's start-up stagger experiment -- first-round workgroups sleeping up to their slot inside the XCD -- measured no gain and left
    // the kernel in round 4: a tile's fixed cost is the serial chain inside a CU, not write-burst contention between lock-stepped CUs)

    // XCD-aware bijective remap of the workgroup id + grouped tile order (same scheme as v2)
    int tn = 0, tmi = 0, slab = -1, ks = 0;  // slab >= 0: this workgroup computes one K range of a split tile into ws
    if (p.mode == 1) {
        // aria_grouped_tile below
    } else if (p.split > 1 && int(blockIdx.x) >= p.split_first) {
        const int r = blockIdx.x - p.split_first;
        slab = r;
        ks = r % p.split;
        if (!aria_tile_from_pos(p, p.split_first + r / p.split, tmi, tn)) return;
    } else if (!aria_tile_coords(p, blockIdx.x, p.split > 1 ? p.split_first : int(gridDim.x), tmi, tn)) {
        return;
    }
    long long b_off = 0, c_off = 0, e_off = 0;
    int m0 = 0, m_end = 0, k_begin = 0, k_len = p.K;
    const int bn_step = p.glu ? 128 : BN;  // fused SwiGLU: a tile covers 128 gate + the matching 128 up columns
    int n0 = tn * bn_step;
    if (p.mode == 0) {
        m0 = tmi * BM;
        m_end = p.M;
        if (m0 >= m_end) return;
    } else if (p.mode == 2) {  // per-expert weight gradient: the reduction runs over the expert's token rows
        const int e = blockIdx.y;
        m0 = tmi * BM;
        m_end = p.M;
        if (m0 >= m_end) return;
        k_begin = p.offsets[e];
        k_len = p.offsets[e + 1] - k_begin;
        c_off = (long long)e * p.strideC;
    } else {
        int expert = 0;
        if (!aria_grouped_tile(p, blockIdx.x, l, expert, m0, m_end, tn)) return;
        n0 = tn * bn_step;
        const int ew = p.expert_mod > 0 ? expert % p.expert_mod : expert;
        b_off = (ARIA_ABL & 16384) ? 0 : (long long)ew * p.strideB;
        e_off = (long long)ew * p.stride_extB;
    }
    char* C = static_cast<char*>(p.C) + c_off * (p.c_f32 ? 4 : 2);
    int nk = (k_len + BK - 1) / BK, kt_first = 0;
    const int ext_k = VER == 12 ? p.ext_k : 0;  // (its own instantiations: the default kernels keep none of the extension's registers live)
    if (ext_k > 0) nk += 1;                                 // K % 64 == 0 there: the extension is one more K-tile with ext_k valid indices
    const int nk_all = nk;
    if (slab >= 0) {  // K-steps [nk * ks / split, nk * (ks + 1) / split)
        kt_first = int((long long)nk * ks / p.split);
        nk = int((long long)nk * (ks + 1) / p.split) - kt_first;
    }

    // order bit 10 (experiment, measured 8-10 % SLOWER, profiles/r02_gemm_tile_timeline.md section 8): the two wave groups run IN STEP instead of
    // one barrier apart -- both waves of a SIMD in their MFMA sections together, fragment reads exposed
    const bool stagger_groups = !((p.order >> 10) & 1);
    // VER 8 / 9: the fused SwiGLU launches ([gate | up] weights / the gptfast two-tensor form) with GATHERED A rows
    constexpr bool GATHER = VER == 8 || VER == 9 || VER == 11;   // (11: the weight gradient with gathered reduction rows, A_OC)
    constexpr bool EXT = VER == 12;                                 // (12: the default epilogues behind a K-extension tile -- LoRA inside the base launch)
    constexpr int XM = (GATHER ? 1 : 0) | (EXT ? 2 : 0);
    Stage st;
    stage_init<A_OC, B_OC, VER == 6 || VER == 9>(st, p, w, l, smem);
    if (GATHER && !A_OC) {  // this lane's four A rows of the tile -> token rows (one index load each, ONCE per tile; rows past the end clamped)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const int row = min(((ARIA_ABL & 8192) ? 0 : m0) + hh * 128 + st.la_a + 8 * ss, p.M - 1);
                st.ga[hh][ss] = mul24(uint32_t(p.gather_rows[row]), st.ldA2) + uint32_t(st.la_b ^ (64 * ss));
            }
    }
    st.gA = reinterpret_cast<const char*>(p.A) + (A_OC ? 2 * k_begin * p.lda : 2 * (long long)k_begin) + kt_first * st.kstepA;
    st.gB = reinterpret_cast<const char*>(p.B + b_off) + (B_OC ? 2 * k_begin * p.ldb : 2 * (long long)k_begin) + kt_first * st.kstepB;
    st.g0 = 0;
    st.nk = nk;
    st.tail_k = (kt_first + nk == nk_all && nk_all > 0) ? k_len - (nk_all - 1) * BK : BK;  // only the overall last K-tile is ragged
    st.ext = ext_k;
    if (ext_k > 0) {
        st.tail_k = ext_k;
        st.eA = reinterpret_cast<const char*>(p.extA);
        st.eB = reinterpret_cast<const char*>(p.extB + e_off);
        st.ldeA2 = uint32_t(2 * p.ld_extA);
        st.ldeB2 = uint32_t(2 * p.ld_extB);
    }
    st.m0 = (ARIA_ABL & (2048 | 8192)) ? 0 : m0;  // (timing experiment: every workgroup loads tile (0, 0)'s operands -- all L2 hits)
    st.n0 = (ARIA_ABL & (2048 | 16384)) ? 0 : n0;
    // (r06 traffic attribution, profiles/r06_pmc_fc1_operands.json: bit 8192 = every workgroup stages the A rows of tile row 0 -- the launch's
    // fabric-side fetches are then the B operand's alone; bit 16384 = every workgroup stages expert 0's column tile 0 -- the A operand's alone)
    FragAddr<A_OC> aa;
    FragAddr<B_OC> ab;
    aa.init(wm * 64, l);
    ab.init(wn * 32, l);

    f32x16 acc[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][i][b][r] = 0.f;
    s16x8 fa[2][4], fb[2][4];  // A fragments of the current A half; B fragments of BOTH halves (B0 is used by phases 1 and 4)

    // rows / columns of this wave's part of half 0 that are in range (half 1 lies 128 further)
    const int rows_left = m_end - m0 - wm * 64, cols_left = p.glu ? BN : p.N - n0 - wn * 32;  // (glu: I % 128 == 0, no column edge)
    // order bit 13 (r06, opt-in): row tiles that hang over their expert's rows run the straight-line steady K loop too -- the rows past the end are
    // clamped loads whose products land in accumulators that are never stored.  The general form skips those MFMAs but is the slower loop: in the
    // r06 timeline the K loop's p90 (= the ragged tiles, 15 % of a grouped-row launch) sits 7-11 % ABOVE the median.  Measured
    // (profiles/r06_edge_steady_ab.json): launch by launch fc1 dgrad +3.1 %, fc2 forward +0.6-1.9 %, the rest level -- and the config #3 step
    // 1.2 ms SLOWER with it as the default of the plain grouped-row launches: the chip is power-limited there, and 7 % more MFMAs on clamped rows
    // cost the step more clock than the straight-line loop saves in time.  Column-edge tiles always keep the general form.
    const bool full_cols = p.glu || n0 + BN <= p.N;
    const bool interior = full_cols && (m0 + BM <= m_end || ((p.order >> 13) & 1));
    if (ARIA_ABL & 128) {
        // (timing experiment: no prologue, no K loop -- launch + tile lookup + epilogue only)
    } else {
        // ---- prologue: tile 0 completely, A0 and B0 of tile 1 (phases 1 and 2 of tile 0 issue its A1 and B1) -- the steady-state
        // queue shape
        if (GATHER && A_OC) {   // reduction rows of tiles 0 and 1 by scalar loads, both requested before the one wait (rows past the expert's end are
            // read -- the index array is padded -- and zero-paged); from tile 2 on the indices come through the LDS slots (idx_dma)
            st.gA0 = reinterpret_cast<const char*>(p.A);
            st.grows = p.gather_rows + k_begin + kt_first * BK;
            st.idx_lds = smem + LDS_IDX + 512 * w;
            i32x8 gr, gr1;
            sload8_issue(gr, st.grows + 8 * w);
            if (nk > 1) sload8_issue(gr1, st.grows + BK + 8 * w);
            sload8_wait(gr);
            gather_k_rows(st, 0, gr, l);
            st.gk[1][0] = st.gk[0][0], st.gk[1][1] = st.gk[0][1];
            if (nk > 1) {
                sload8_wait(gr1);
                gather_k_rows(st, 1, gr1, l);   // A0 of tile 1 below takes NXT; phase 1 of K-tile 0 (A1 of tile 1) takes CUR, set behind tile 0's pieces
            }
        }
        if (nk > 0) {
            stage_half<A_OC, B_OC, 0, 0, 0, true, XM, 0>(st, 0);   // (GKW = 0: tile 0's rows are in gk[0] whatever the half)
            stage_half<A_OC, B_OC, 1, 0, 0, true, XM & 2>(st, 0);
            stage_half<A_OC, B_OC, 1, 1, 0, true, XM & 2>(st, 0);
            stage_half<A_OC, B_OC, 0, 1, 0, true, XM, 0>(st, 0);
        }
        if (GATHER && A_OC && nk > 1) {
            st.gk[0][0] = st.gk[1][0], st.gk[0][1] = st.gk[1][1];
            idx_dma<0>(st, 2);   // tile 2's indices: in front of tile 1's pieces, so the wait below confirms them
        }
        // r06: the barrier in front of K-tile 0 only needs A0 and B0 of tile 0 (phase 1 reads nothing else and carries its own counted wait for
        // B1 / A1): the pieces issued behind them stay in flight -- B1, A1 of tile 0 (4), the gathered weight gradient's index piece (1), A0, B0 of
        // tile 1 (4).  (Was: everything of tile 0.)
        if (nk > 1) {
            stage_half<A_OC, B_OC, 0, 0, 1, true, XM>(st, 1);
            stage_half<A_OC, B_OC, 1, 0, 1, true, XM & 2>(st, 1);
            wait_vm<8 + ((GATHER && A_OC) ? 1 : 0)>();
        } else {
            wait_vm<4>();
        }
        raw_barrier();
        ts_mark(1);
        if (wm == 1 && stagger_groups) raw_barrier();  // waves 4-7 run one barrier behind waves 0-3
        if (interior)
            k_loop3<A_OC, B_OC, false, XM>(acc, fa, fb, aa, ab, smem, st, nk, rows_left, cols_left);
        else
            k_loop3<A_OC, B_OC, true, XM>(acc, fa, fb, aa, ab, smem, st, nk, rows_left, cols_left);
    }
    if (wm == 0 && stagger_groups) raw_barrier();  // balance the barrier count of the two groups
    ts_mark(2);

    if (slab >= 0) {  // raw fp32 partial sums for gemm3_reduce_kernel, which is their only reader -- so they leave in the ACCUMULATORS' order,
        // r05b: 16 bytes per lane and 1 KiB contiguous per store instruction (32 dwordx4 stores per lane; the tile-shaped [256][256] form was
        // 128 dword stores per lane, ~16 us of epilogue on a split tile).  Float offset of register 4 q + e of accumulator tile (b, a, i) of wave
        // w, lane l:  ((((w * 8 + (b * 2 + a) * 2 + i) * 4 + q) * 64 + l) * 4 + e
        float* dst = p.ws + (long long)slab * (BM * BN) + (long long)w * 8 * 4 * 64 * 4 + l * 4;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4*>(dst + (((b * 2 + a) * 2 + i) * 4 + q) * 256) =
                            f32x4{acc[a][i][b][4 * q], acc[a][i][b][4 * q + 1], acc[a][i][b][4 * q + 2], acc[a][i][b][4 * q + 3]};
        return;
    }
    if ((ARIA_ABL & 64) && p.M > 0) return;  // (timing experiment: no C write-out)
    if (VER == 5) {  // compile-time: the SwiGLU-backward instantiations carry only this epilogue, the default ones none of it
        store_tile3_dglu(p, acc, C, m0, m_end, n0, l, w, wm, wn, smem);
        return;
    }
    if (VER == 7) {  // fused wqkv projection: RoPE + KV-cache write epilogue
        store_tile3_rows_rope(p, acc, C, m0, m_end, n0, l, w, wm, wn, smem);
        return;
    }
    if (VER == 10) {  // the HF layer's q | k | v projection with the half-split RoPE as its epilogue
        store_tile3_rows_rope_hf(p, acc, C, m0, m_end, n0, l, w, wm, wn, smem);
        return;
    }
    if (p.glu)
        store_tile3_glu(p, acc, C, m0, m_end, n0, l, w, wm, wn, smem);
    else if (p.act == 1)  // (one wave-uniform branch here instead of one per value inside the unrolled epilogues)
        store_any3<1>(p, acc, C, m0, m_end, n0, l, w, wm, wn, smem);
    else
        store_any3<0>(p, acc, C, m0, m_end, n0, l, w, wm, wn, smem);
}

// Sums the `split` slabs of every split tile in slab order (deterministic), then bias / accumulate / round exactly like the
// main epilogue.  grid (split tiles, 16): block (r, part) finishes the part-th sixteenth of split tile r IN THE SLABS' ORDER (the accumulators'
// order, see the slab store in gemm3_kernel): a thread's four consecutive floats are registers 4 q .. 4 q + 3 of one lane = one row, four
// consecutive columns (r06: transposed accumulators).
__global__ __launch_bounds__(256) void gemm3_reduce_kernel(GemmParams p) {
    int tn, tmi;
    if (!aria_tile_from_pos(p, p.split_first + blockIdx.x, tmi, tn)) return;
    const int t = threadIdx.x;
    const float* ws = p.ws + (long long)blockIdx.x * p.split * (BM * BN);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int j = (blockIdx.y * 4 + it) * 256 + t;   // 16-byte piece of the slab
        const int l = j & 63, q = (j >> 6) & 3, t8 = (j >> 8) & 7, w = j >> 11;
        const int b = t8 >> 2, a = (t8 >> 1) & 1, i = t8 & 1, wm = w >> 2, wn = w & 3;
        // (r06, transposed accumulators: registers 4 q .. 4 q + 3 of a lane = one ROW, four consecutive columns)
        const int row = a * 128 + wm * 64 + i * 32 + (l & 31), col0 = b * 128 + wn * 32 + 8 * q + 4 * (l >> 5);
        f32x4 sum = *reinterpret_cast<const f32x4*>(ws + 4 * j);
        for (int s = 1; s < p.split; ++s) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ws + (long long)s * (BM * BN) + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[e] += v[e];
        }
        const int m = tmi * BM + row;
        if (m >= p.M) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = tn * BN + col0 + e;
            if (n >= p.N) continue;
            const float bv = p.bias ? bf2f(p.bias[n]) : 0.f;
            float v = aria_epilogue_act(p, sum[e] + bv);
            if (p.c_f32) {
                float* d = static_cast<float*>(p.C) + (long long)m * p.ldc + n;
                *d = p.accumulate ? *d + v : v;
            } else {
                bf16_t* d = static_cast<bf16_t*>(p.C) + (long long)m * p.ldc + n;
                if (p.accumulate) v += bf2f(*d);
                *d = f2bf(v);
            }
        }
    }
}

// Remainder split-K plan for a dense problem of `tiles` 256x256 tiles and nk K-steps: the last, partly filled round of
// R = tiles mod 256 tiles (all of them when tiles < 256) leaves 256 - R CUs idle for a whole tile time; computing each of those R
// tiles with S workgroups over K/S instead costs ceil(R S / 256) / S tile times plus the slab round trip (~1 MiB per slab pair
// of write + read against ~K * 29 ns of tile time).  Returns S (1 = do not split).
int plan_split(long long tiles, long long nk, long long* remainder) {
    const long long R = tiles % 256;
    *remainder = R;
    if (R == 0 || nk < 16) return 1;
    const double slab_cost = 4.5 * double(R) / double(nk * 64);  // tile times per unit of S
    double best = 1.0;
    int best_s = 1;
    for (int S = 2; S <= 8 && nk / S >= 8; ++S) {
        const double t = double((R * S + 255) / 256) / S + slab_cost * S;
        if (t < best) {
            best = t;
            best_s = S;
        }
    }
    return best < 0.9 ? best_s : 1;
}

}  // namespace

long long aria_gemm3_workspace_bytes(long long M, long long N, long long K) {
    // at least one full tile each way -- or a skinny output with a long reduction (router weight gradient [64 x 2560], K = tokens; the LoRA
    // factors' gradients [r x in] / [out x r], r = 8..24, which took 320 us each on 20 workgroups of the 128 x 128 kernel: profiles/
    // r05_lora_kernel_stats.txt) -- but never the decode GEMVs (M = batch < 8)
    if (K < 64 || M < 8 || N < 8 || ((M < 256 || N < 256) && K < 2048)) return 0;
    long long R = 0;
    const int S = plan_split(((M + 255) / 256) * ((N + 255) / 256), (K + 63) / 64, &R);
    return S > 1 ? R * S * (long long)(BM * BN) * 4 : 0;
}

// v3 eligibility is decided by the caller (gemm.hip): K % 64 == 0, K >= 64, mode 0 or 1, operand bytes < 4 GiB
int aria_launch_gemm3(const GemmParams& p, int a_oc, int b_oc, int ntm, void* stream, void* workspace, long long workspace_bytes) {
    const unsigned grid_y = p.mode == 2 ? unsigned(p.E) : 1u;
    const size_t shmem = size_t(256) * ROWP3;  // the operand images (128 KiB) or, at the end, the parked output tile (row form of the epilogue)
    const int ntn = p.glu ? (p.N / 2) / 128 : (p.N + BN - 1) / BN;
    GemmParams q = p;
    q.ntn = ntn;
    q.ntm = ntm;
    q.split = 1;
    q.split_first = 0;
    q.ws = nullptr;
    long long R = 0;
    if (p.mode == 0 && workspace && !p.glu && !p.ext_k) {
        const int S = plan_split((long long)ntn * ntm, (p.K + BK - 1) / BK, &R);
        if (S > 1 && workspace_bytes >= R * S * (long long)(BM * BN) * 4) {
            q.split = S;
            q.split_first = int((long long)ntn * ntm - R);
            q.ws = static_cast<float*>(workspace);
        }
    }
    const char* ord = std::getenv("ARIA_GEMM_ORDER");
    // bit 8 = DMA pieces inside the MFMA section: measured +4..6 % with two k-contiguous operands, -5 % when an operand goes through
    // the transposing reads (profiles/r01_gemm_tuning.md)
    q.order = ord ? std::atoi(ord) : 4;  // groups of 4 row tiles (dense); bit 9 = ragged-last (grouped rows, gemm_params.h)
    // r04: the fused fc1 + SwiGLU launches over grouped rows run ragged-last by default -- measured on the same box
    // (profiles/r04_grouped_order_ab.json, r04_pmc_grouped_orders_and_clocks.json): fabric-side fetches 7.0 -> 3.9 GB per launch, L2 hit rate
    // 49 -> 69 %, TF/s level (975 / 1023 vs 1012 / 993); the plain grouped launches lose 1-2 % under it and keep the expert-major order
    if (!ord && p.glu && p.mode == 1) q.order |= 512;
    // (order bit 13 -- ragged row tiles on the steady K loop -- wins 2-3 % on the plain grouped-row launches alone and LOSES 1.2 ms per step inside the
    // power-limited training step, profiles/r06_edge_steady_ab.json: opt-in)
    // wide epilogue: every 8-column piece of a C row must be 16-byte aligned.  ARIA_GEMM_WIDE_STORE=0 switches it off (A/B measurements).
    const char* wsd = std::getenv("ARIA_GEMM_WIDE_STORE");
    q.wide_store = !(wsd && wsd[0] == '0') && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && (p.ldc & 7) == 0 && (p.strideC & 7) == 0;
    if (q.wide_store) q.wide_store = 2;  // row form (whole tile parked, complete 512-byte rows per store); the per-wave form (=1) left the library in r06
    if (ntn * ntm <= 0) return ARIA_OK;
    if (a_oc && !b_oc) return ARIA_ERR_INVALID;
    if (p.ext_k) {  // K-extension: whole K-tiles in front of it, 16-byte granules, neither split-K nor the forms whose loaders it does not cover
        if (!p.extA || !p.extB || p.ext_k < 0 || p.ext_k > BK || (p.ext_k & 7) || (p.K % BK) || a_oc || p.mode == 2 || p.gather_rows || p.rope_fc ||
            (p.glu && p.glu_up_rows > 0) || p.dglu || (p.ld_extA & 7) || (p.ld_extB & 7) || (p.stride_extB & 7) ||
            (reinterpret_cast<uintptr_t>(p.extA) & 15) || (reinterpret_cast<uintptr_t>(p.extB) & 15) ||
            2 * (long long)p.M * p.ld_extA >= (1ll << 32) || 2 * p.ld_extA >= (1ll << 24) || 2 * p.ld_extB >= (1ll << 24) ||
            2 * (b_oc ? (long long)p.ext_k : (long long)p.N) * p.ld_extB >= (1ll << 32))
            return ARIA_ERR_INVALID;
    }
    dim3 grid(unsigned(aria_tile_grid(q)), grid_y), block(512);
    if (p.rope_fc && p.rope_sn) {  // HF form: q | k | v projection + half-split RoPE (two whole heads per column tile)
        if (a_oc || b_oc || p.mode != 0 || p.glu || p.dglu || p.c_f32 || p.accumulate || p.bias || p.act || (p.N % BN) || (p.rope_D % BN) ||
            (BN % p.rope_hd) || (p.rope_hd & 15) || !q.wide_store)
            return ARIA_ERR_INVALID;
        ARIA_LAUNCH((gemm3_kernel<false, false, 10>), grid, block, shmem, stream, q);
        return aria_check_launch();
    }
    if (p.rope_fc) {  // fused wqkv projection (K7): dense, both operands k-contiguous, whole column tiles
        if (a_oc || b_oc || p.mode != 0 || p.glu || p.dglu || p.c_f32 || p.accumulate || p.bias || p.act || (p.N % BN) || (p.rope_D % BN)) return ARIA_ERR_INVALID;
        ARIA_LAUNCH((gemm3_kernel<false, false, 7>), grid, block, shmem, stream, q);
        return aria_check_launch();
    }
    if (p.gather_rows && p.mode == 2) {  // the per-expert weight gradient with gathered reduction rows
        if (!a_oc || !b_oc || p.glu || p.dglu || p.ext_k) return ARIA_ERR_INVALID;
        ARIA_LAUNCH((gemm3_kernel<true, true, 11>), grid, block, shmem, stream, q);
        return aria_check_launch();
    }
    if (p.gather_rows) {  // gathered A rows (K2): the two fused SwiGLU launches over grouped rows only
        if (a_oc || !p.glu || p.mode != 1 || (p.glu_up_rows > 0 && b_oc) || (p.glu_up_rows == 0 && !b_oc)) return ARIA_ERR_INVALID;
        if (p.glu_up_rows > 0)
            ARIA_LAUNCH((gemm3_kernel<false, false, 9>), grid, block, shmem, stream, q);
        else
            ARIA_LAUNCH((gemm3_kernel<false, true, 8>), grid, block, shmem, stream, q);
        return aria_check_launch();
    }
    if (p.glu && p.glu_up_rows > 0) {  // fused SwiGLU with gate / up weights in two tensors ([N, K] form): its own instantiation
        if (a_oc || b_oc || p.mode == 2) return ARIA_ERR_INVALID;
        ARIA_LAUNCH((gemm3_kernel<false, false, 6>), grid, block, shmem, stream, q);
        return aria_check_launch();
    }
    if (p.dglu) {  // SwiGLU-backward epilogue: its own instantiations (the default kernels' code is untouched)
        if (a_oc || p.glu || p.c_f32 || p.accumulate || p.bias || p.act || p.mode == 2) return ARIA_ERR_INVALID;
        if (!b_oc)
            ARIA_LAUNCH((gemm3_kernel<false, false, 5>), grid, block, shmem, stream, q);
        else
            ARIA_LAUNCH((gemm3_kernel<false, true, 5>), grid, block, shmem, stream, q);
        return aria_check_launch();
    }
    if (p.ext_k) {  // K-extension (LoRA inside the base launch): its own instantiations of the default epilogues
        if (!b_oc)
            ARIA_LAUNCH((gemm3_kernel<false, false, 12>), grid, block, shmem, stream, q);
        else
            ARIA_LAUNCH((gemm3_kernel<false, true, 12>), grid, block, shmem, stream, q);
        return aria_check_launch();
    }
    if (!a_oc && !b_oc)
        ARIA_LAUNCH((gemm3_kernel<false, false, 3>), grid, block, shmem, stream, q);
    else if (!a_oc && b_oc)
        ARIA_LAUNCH((gemm3_kernel<false, true, 3>), grid, block, shmem, stream, q);
    else
        ARIA_LAUNCH((gemm3_kernel<true, true, 3>), grid, block, shmem, stream, q);
    if (q.split > 1) ARIA_LAUNCH(gemm3_reduce_kernel, dim3(unsigned(R), 16), dim3(256), 0, stream, q);
    return aria_check_launch();
}

#if ARIA_ABL & 512
extern "C" int aria_abl_ts(unsigned long long* host, int n_words) {
    return int(hipMemcpyFromSymbol(host, HIP_SYMBOL(aria_ts), size_t(n_words) * 8));
}
extern "C" int aria_abl_hw(unsigned* host, int n_words) {
    return int(hipMemcpyFromSymbol(host, HIP_SYMBOL(aria_ts_hw), size_t(n_words) * 4));
}
#endif
