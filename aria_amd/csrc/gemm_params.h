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
    int ntn;
    int order;  // tile-order variant (tuning knob, see gemm2.hip)
};

// v2 launcher (gemm2.hip); returns ARIA_* status
int aria_launch_gemm2(const GemmParams& p, int a_oc, int b_oc, int ntm_or_max_tm, int grid_y, void* stream);
