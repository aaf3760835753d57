/*
 * aria_hip.h -- C ABI of libaria_hip.so: the MI355X (gfx950) hot path of rhymes-ai/Aria.
 *
 * Drop-in boundary.  Every entry point takes raw device pointers, integer
 * sizes and a hipStream_t (as void*); outputs are caller-allocated; nothing allocates, nothing
 * synchronises with the host, no global state.  Return value: ARIA_OK or an ARIA_ERR_* code (the
 * Python host raises on non-zero).  Routing metadata (tokens_per_expert / offsets) stays on the
 * device -- the reference's GroupedGEMM.forward forces a D2H sync per call (aria/model/moe_lm.py:478).
 *
 * Each declaration cites the reference interface it replaces (paths relative to /root/reference;
 * "transformers/..." = arithmetic the reference inherits from transformers==4.46.3).
 * All matrices are bfloat16 (raw uint16 bits) unless stated, row-major, leading dimension in ELEMENTS.
 */
#ifndef ARIA_HIP_H
#define ARIA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARIA_OK 0
#define ARIA_ERR_INVALID 1     /* null pointer / negative size */
#define ARIA_ERR_ALIGN 2       /* pointer or leading dimension not 16-byte aligned where required */
#define ARIA_ERR_UNSUPPORTED 3 /* shape outside what the kernels implement (e.g. > 256 experts) */
#define ARIA_ERR_LAUNCH 4      /* hipGetLastError() != hipSuccess after the launch */

/* Library / ABI version: bumped whenever an entry point is added or a signature changes (3 = round 5's additions: aria_moe_router_fused,
 * aria_attn_bwd_rope, aria_moe_unpermute_res, aria_scale_bf16, aria_gemm_qkv_rope_hf_bf16, the *_lora_* family).  The Python host
 * (aria_amd/hip.py) refuses a library whose version is not the one it was written against. */
#define ARIA_ABI_VERSION 3
int aria_abi_version(void);
/* Test/diagnostic aid: which GEMM kernel family the calling thread's last aria_*gemm* call dispatched to
 * (1 = 128x128 tile, 2 = 256x256 register-staged, 3 = 256x256 LDS-DMA phase-scheduled; 0 = none yet). */
int aria_last_gemm_variant(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM family (gemm.hip) -- v_mfma_f32_32x32x16_bf16, fp32 accumulate.
 * a_oc / b_oc = 0: operand is reduction-contiguous  (A[M,K] row-major;  B given as [N,K] row-major)
 *             = 1: operand is output-contiguous     (A given as [K,M];  B given as [K,N] row-major)
 * C[M,N] = A*B (+ bias[N]) (+ C if accumulate); C is bf16, or fp32 when c_f32.
 * Replaces every nn.Linear / F.linear on the path and its dgrad/wgrad:
 *   Linear fwd  y = x W^T : (a_oc=0, b_oc=0)   router gating aria/model/moe_lm.py:190-201; q/k/v/o
 *                                              transformers/models/llama/modeling_llama.py:243-281;
 *                                              SharedExpertMLP moe_lm.py:368-395; lm_head; ViT linears
 *   Linear dgrad dx = dy W : (0, 1);  Linear wgrad dW = dy^T x : (1, 1)
 * ------------------------------------------------------------------------------------------------ */
int aria_gemm_bf16(const void* A, const void* B, void* C, const void* bias /* bf16[N] or NULL */, int64_t M, int64_t N,
                   int64_t K, int a_oc, int b_oc, int64_t lda, int64_t ldb, int64_t ldc, int c_f32, int accumulate,
                   void* stream);
/* Same GEMM with a caller-owned scratch buffer.  When the 256x256 tile list ends in a partly filled round (e.g. a
 * [2560,2560] weight gradient = 100 tiles on 256 CUs) those tiles are computed by several workgroups over disjoint K ranges
 * into fp32 slabs in `workspace` and summed in a fixed order by a second kernel (deterministic).  aria_gemm_workspace_bytes
 * returns the size that enables it for a problem (0: no split would be used); a NULL / too small workspace just disables it. */
int64_t aria_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int a_oc, int b_oc);
int aria_gemm_bf16_ws(const void* A, const void* B, void* C, const void* bias, int64_t M, int64_t N, int64_t K, int a_oc,
                      int b_oc, int64_t lda, int64_t ldb, int64_t ldc, int c_f32, int accumulate, void* workspace,
                      int64_t workspace_bytes, void* stream);
/* Same, with an epilogue activation fused after the bias: the activation sees bf16(acc + bias), exactly what a separate
 * elementwise kernel would read, so fused and unfused results are bit-identical (ViT MLP: fc1 + gelu_pytorch_tanh,
 * modeling_idefics2.py MLP / ACT2FN).  The activation is applied before `accumulate` adds the old C. */
#define ARIA_ACT_NONE 0
#define ARIA_ACT_GELU_TANH 1
int aria_gemm_act_bf16(const void* A, const void* B, void* C, const void* bias, int64_t M, int64_t N, int64_t K, int a_oc,
                       int b_oc, int64_t lda, int64_t ldb, int64_t ldc, int c_f32, int accumulate, int act, void* workspace,
                       int64_t workspace_bytes, void* stream);

/* experts_gemm(input, weight, tokens_per_expert)  -- seam B1, aria/model/moe_lm.py:431-443 (grouped_gemm.ops.gmm
 * or sequential_gemm :398-428), called from GroupedGEMM.forward :467-484.
 *   C[s_e : s_e+n_e, :] = A[s_e : s_e+n_e, :] * B_e       offsets[e] = s_e (device int32[E+1], offsets[E] = M_total)
 *   b_oc = 1: B_e = B + e*strideB is [K,N] row-major (the reference's weight layout, forward)
 *   b_oc = 0: B_e is [N,K] row-major, i.e. C = A * W_e^T with W_e [N,K] (dgrad through the same weights) */
int aria_grouped_gemm_bf16(const void* A, const void* B, void* C, const int32_t* offsets, int64_t E, int64_t M_total,
                           int64_t N, int64_t K, int b_oc, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldc,
                           void* stream);

/* GroupedMLP.forward's first half as ONE launch: fc1 (experts_gemm, [K, 2I] weights) + glu (aria/model/moe_lm.py:505-507, 522-523):
 *   H[s_e:s_e+n_e, :] = A[s_e:s_e+n_e, :] * B_e  (written only if H != NULL: the backward of glu needs it)
 *   ACT[:, j] = silu(H[:, j]) * H[:, I + j],  I = N2 / 2,  with the bf16 rounding points of the unfused chain (bit-identical to
 *   aria_grouped_gemm_bf16 followed by aria_swiglu_fwd).  Needs I % 128 == 0 (else ARIA_ERR_UNSUPPORTED: call the two-step form). */
int aria_grouped_gemm_swiglu_bf16(const void* A, const void* B, void* H, void* ACT, const int32_t* offsets, int64_t E, int64_t M_total,
                                  int64_t N2, int64_t K, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldh, int64_t ldact,
                                  void* stream);
/* The dense counterpart (SharedExpertMLP = LlamaMLP, moe_lm.py:368-395: act(gate_proj(x)) * up_proj(x)) on a [2I, K] weight whose first I
 * rows are gate_proj.weight and last I rows up_proj.weight. */
int aria_gemm_swiglu_bf16(const void* A, const void* B, void* H, void* ACT, int64_t M, int64_t N2, int64_t K, int64_t lda, int64_t ldb,
                          int64_t ldh, int64_t ldact, void* stream);

/* The same two fusions for the gptfast wire format (gptfast/model.py:262-325: ConditionalFeedForward.w1 / w3 [E, I, K] and FeedForward.w1 / w3
 * [I, K] are SEPARATE tensors in [N, K] form): Bg = w1 (gate rows), Bu = w3 (up rows).  Both must lie in ONE allocation with Bu a whole
 * number of rows (ldb elements) behind Bg -- aria_amd.gptfast lays its parameters out that way -- because the kernel reaches the up rows as a
 * row offset from the gate rows (32-bit per-lane addressing); anything else returns ARIA_ERR_UNSUPPORTED (call the two-step form).
 * strideB = elements between consecutive experts of EITHER tensor.  H may be NULL.  Bit-identical to two aria_grouped_gemm_bf16 /
 * aria_gemm_bf16 calls followed by aria_swiglu_fwd(h1, h3). */
int aria_grouped_gemm_swiglu_split_bf16(const void* A, const void* Bg, const void* Bu, void* H, void* ACT, const int32_t* offsets, int64_t E,
                                        int64_t M_total, int64_t I, int64_t K, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldh,
                                        int64_t ldact, void* stream);

/* K2: TokenDispatcher.token_permutation's row gather (moe_lm.py:326-334; gptfast/model.py:243-254) folded into the A loader of
 * the fused fc1 + SwiGLU launches: X is the UN-permuted token matrix [T, K] and rows[r] (int32 on the device, r < M_total) the token row
 * that permuted row r would hold (= sorted_src[r] / topk); each lane looks its four rows up once per tile.  The [M_total, K] permuted copy is
 * neither written nor read.  Same results, bit for bit, as aria_moe_permute followed by the un-gathered launch.  For paths that need no
 * weight gradient of fc1 afterwards (inference prefill, the forward pass of a checkpointed step): the weight gradient's loader wants the
 * permuted rows as a tensor.  K % 64 == 0, T < 2^24, 2 T ldx < 2^32. */
int aria_grouped_gemm_swiglu_gather_bf16(const void* X, const int32_t* rows, int64_t T, const void* B, void* H, void* ACT, const int32_t* offsets,
                                         int64_t E, int64_t M_total, int64_t N2, int64_t K, int64_t ldx, int64_t ldb, int64_t strideB, int64_t ldh,
                                         int64_t ldact, void* stream);
int aria_grouped_gemm_swiglu_split_gather_bf16(const void* X, const int32_t* rows, int64_t T, const void* Bg, const void* Bu, void* H, void* ACT,
                                               const int32_t* offsets, int64_t E, int64_t M_total, int64_t I, int64_t K, int64_t ldx, int64_t ldb,
                                               int64_t strideB, int64_t ldh, int64_t ldact, void* stream);

/* SURVEY 8(f)3 -- LoRA fused into the base GEMM (GroupedGemmLoraLayer.forward aria/lora/layers.py:129-139: result = base(x) + lora_B(lora_A(
 * dropout(x))) * scaling; peft's Linear adapter, same line for nn.Linear targets, recipes/config_lora.yaml:44-59).  The adapter's second
 * projection rides in the base launch as a K-EXTENSION: C = A B + EA EB, with EA [M, ext_k] = scaling * lora_A(dropout(x)) (k-contiguous rows,
 * leading dimension ld_ea) and EB the adapter's B factor in the base weight's own form -- b_oc = 0 ([N, K] weights, nn.Linear): EB [N, ext_k];
 * b_oc = 1 ([K, N] weights, the experts): EB [ext_k, N], per expert at stride_eb.  One extra K-tile whose DMA granules come from EA / EB (zero
 * page beyond ext_k): no output-sized add pass, no y + delta round trip, and the fused epilogues (SwiGLU) see base + adapter in ONE fp32
 * accumulator, rounded once (the reference rounds base, adapter and sum separately: the fused result is the closer one to exact).  Several
 * adapters that share an input (q / k / v; gate / up) go in as ONE extension: EA = their columns side by side, EB block-diagonal.
 * ext_k % 8 == 0, <= 64; K % 64 == 0; shapes the 256 x 256 kernels do not take: ARIA_ERR_UNSUPPORTED (run base and adapter as separate
 * launches, the second one accumulating).  The dgrad uses the same entries with (EA, EB) = (d_u, lora_A) when there is no dropout. */
int aria_gemm_lora_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int b_oc, int64_t lda, int64_t ldb, int64_t ldc,
                        const void* EA, const void* EB, int64_t ext_k, int64_t ld_ea, int64_t ld_eb, void* stream);
int aria_gemm_swiglu_lora_bf16(const void* A, const void* B, void* H, void* ACT, int64_t M, int64_t N2, int64_t K, int64_t lda, int64_t ldb,
                               int64_t ldh, int64_t ldact, const void* EA, const void* EB, int64_t ext_k, int64_t ld_ea, int64_t ld_eb,
                               void* stream);
int aria_grouped_gemm_lora_bf16(const void* A, const void* B, void* C, const int32_t* offsets, int64_t E, int64_t M_total, int64_t N, int64_t K,
                                int b_oc, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldc, const void* EA, const void* EB, int64_t ext_k,
                                int64_t ld_ea, int64_t ld_eb, int64_t stride_eb, void* stream);
int aria_grouped_gemm_swiglu_lora_bf16(const void* A, const void* B, void* H, void* ACT, const int32_t* offsets, int64_t E, int64_t M_total,
                                       int64_t N2, int64_t K, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldh, int64_t ldact,
                                       const void* EA, const void* EB, int64_t ext_k, int64_t ld_ea, int64_t ld_eb, int64_t stride_eb,
                                       void* stream);

/* nn.Dropout(lora_dropout) in front of lora_A (aria/lora/layers.py:83-85, 131; recipes/config_lora.yaml:46) and its backward.  Inverted
 * dropout on n bf16 elements (n % 8 == 0): out = keep ? bf16(x / (1 - p)) : 0, mask[n / 8] = one byte per 8 elements (bit e = kept).  The
 * draws are a counter-based function of (seed, element index): p is resolved to 1 / 65536.  _bwd: dx (+)= keep ? term / (1 - p) : 0. */
int aria_dropout_fwd_bf16(const void* x, void* out, void* mask, int64_t n, float p, uint64_t seed, void* stream);
int aria_dropout_bwd_bf16(const void* term, const void* mask, void* dx, int64_t n, float p, int accumulate, void* stream);

/* K2 for the TRAINING step (round 5): the weight gradient of experts.fc1 through the dispatcher's index -- dW[e] = sum over the expert's permuted rows r
 * of X[rows[r]]^T dY[r] (autograd of experts_gemm(index_select(x, sorted // topk), fc1), moe_lm.py:326-334, 467-484) -- so that the [6T, D] permuted copy
 * of the tokens is never written: forward = aria_grouped_gemm_swiglu_gather_bf16, weight gradient = this.  The indices of a K-tile reach the loader as one
 * more LDS-DMA piece of the counted-vmcnt pipeline (r06; K-tiles 0 and 1 by scalar loads in the prologue).  rows: int32 [M_total + 64] (64 entries of padding are read,
 * never used); T < 2^24, 2 T ldx < 2^32.  Bit-identical to aria_moe_permute + aria_grouped_gemm_wgrad_bf16.  Shapes the 256 x 256 kernels do not take:
 * ARIA_ERR_UNSUPPORTED. */
int aria_grouped_gemm_wgrad_gather_bf16(const void* X, const int32_t* rows, const void* dY, void* dW, const int32_t* offsets, int64_t E, int64_t T,
                                        int64_t K, int64_t N, int64_t ldx, int64_t ldy, int c_f32, int accumulate, void* stream);

/* K7: gptfast's Attention.forward up to the attention call (gptfast/model.py:413-435) as ONE launch: the fused wqkv projection
 * X [M, K] x Wqkv^T ([3 D, K]: q rows, k rows, v rows), the interleaved-pair RoPE of q and k (apply_rotary_emb :519-531: fp32 arithmetic on
 * the bf16-rounded product with the bf16 freqs_cis table [positions, hd / 2, 2], one rounding) and KVCache.update (:67-93) as the GEMM's
 * epilogue: q -> Q [M, ldq]; rotated k and v -> the static caches Kc / Vc [B, S_cache, D] (row stride ld_cache) at row
 * (t / S) * S_cache + pos[t] (pos int32 per token row on the device, or NULL: t % S).  The [M, 3 D] product never visits HBM.
 * Bit-identical to aria_gemm_bf16 + aria_rope_interleaved_inplace + row copies.  D % 256 == 0, K % 64 == 0, hd % 8 == 0. */
int aria_gemm_qkv_rope_cache_bf16(const void* X, const void* Wqkv, void* Q, void* Kc, void* Vc, const void* freqs_cis, const int32_t* pos,
                                  int64_t M, int64_t D, int64_t K, int64_t hd, int64_t S, int64_t S_cache, int64_t ldx, int64_t ldw, int64_t ldq,
                                  int64_t ld_cache, void* stream);

/* The HF layer's counterpart (LlamaAttention.forward, transformers modeling_llama.py:243-281 reached through moe_lm.py:594): the q | k | v
 * projections as one wide GEMM X [M, K] x Wqkv^T ([3 D, K]) with apply_rotary_pos_emb's half-split rotation (:130-160) of the q and k columns as
 * its epilogue -- cos / sin [S, hd] bf16 (emb = cat(freqs, freqs), as LlamaRotaryEmbedding builds them), position of row m = m % S.  QKV [M, 3 D]
 * holds rotated q, rotated k, v.  Bit-identical to aria_gemm_bf16 followed by aria_rope_inplace.  D % 256 == 0, 256 % hd == 0, K % 64 == 0,
 * enough tiles for the 256 x 256 kernels: ARIA_ERR_UNSUPPORTED otherwise (run the two calls). */
int aria_gemm_qkv_rope_hf_bf16(const void* X, const void* Wqkv, void* QKV, const void* cos, const void* sin, int64_t M, int64_t D, int64_t K,
                               int64_t hd, int64_t S, int64_t ldx, int64_t ldw, int64_t ldc, void* stream);

/* Expert parallelism (BASELINE config #5; the reference's dispatcher is local, moe_lm.py:313-365): the grouped GEMM and the fused fc1 + glu
 * launch over the SEGMENTS of an all-to-all's output.  Rows arrive ordered (source rank s, local expert e); offsets int32 [n_seg + 1] bound the
 * n_seg = ranks x n_local segments in that order, and segment g multiplies with the weight of local expert g % n_local (B + (g % n_local) *
 * strideB) -- the exchange's output is consumed where it lands, no re-order pass in front of the GEMM or behind it.  Everything else as
 * aria_grouped_gemm_bf16 / aria_grouped_gemm_swiglu_bf16.  K % 64 == 0 (ARIA_ERR_UNSUPPORTED otherwise: re-order and use the plain entries).
 * The weight gradient needs no entry of its own: aria_grouped_gemm_wgrad_bf16 once per source rank on that rank's n_local + 1 offsets,
 * accumulating. */
int aria_grouped_gemm_seg_bf16(const void* A, const void* B, void* C, const int32_t* offsets, int64_t n_seg, int64_t n_local, int64_t M_total,
                               int64_t N, int64_t K, int b_oc, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldc, void* stream);
int aria_grouped_gemm_swiglu_seg_bf16(const void* A, const void* B, void* H, void* ACT, const int32_t* offsets, int64_t n_seg, int64_t n_local,
                                      int64_t M_total, int64_t N2, int64_t K, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldh, int64_t ldact,
                                      void* stream);
int aria_gemm_swiglu_split_bf16(const void* A, const void* Bg, const void* Bu, void* H, void* ACT, int64_t M, int64_t I, int64_t K, int64_t lda,
                                int64_t ldb, int64_t ldh, int64_t ldact, void* stream);

/* Backward of GroupedMLP's glu (moe_lm.py:505-507) fused behind experts.fc2's input gradient, ONE launch:
 *   d_act[s_e:s_e+n_e, :] = dY[s_e:s_e+n_e, :] * W_e^T   (W_e = B + e*strideB is [I, K] row-major: fc2.weight[e], the forward's [K_fwd = I, N_fwd = K])
 *   DH[:, j] = d_act[:, j] * H[:, I + j] * silu'(H[:, j]),   DH[:, I + j] = d_act[:, j] * bf16(silu(H[:, j]))
 * with d_act rounded to bf16 where the two-step chain materialises it: bit-identical to aria_grouped_gemm_bf16 (b_oc = 0) followed by
 * aria_swiglu_bwd, without the M x I round trip through HBM.  H, DH: [M_total, 2 I] ([gate | up]).  Needs I % 128 == 0, K % 64 == 0
 * (else ARIA_ERR_UNSUPPORTED: call the two-step form). */
int aria_grouped_gemm_dswiglu_bf16(const void* dY, const void* B, const void* H, void* DH, const int32_t* offsets, int64_t E, int64_t M_total,
                                   int64_t I, int64_t K, int64_t lda, int64_t ldb, int64_t strideB, int64_t ldh, int64_t lddh, void* stream);
/* The dense counterpart (SharedExpertMLP's down_proj input gradient + act backward): d_act = dY * W with W [K, I] row-major
 * (down_proj.weight, b_oc = 1 form) or, b_oc = 0, W [I, K].  Bit-identical to the two-step chain through the SAME kernel family; the dense
 * two-step entry (aria_gemm_bf16_ws) may pick split-K for a partly filled last round of tiles, which sums the reduction in a different
 * order -- against that path the results agree to the GEMM tolerance, not bit for bit. */
int aria_gemm_dswiglu_bf16(const void* dY, const void* B, const void* H, void* DH, int64_t M, int64_t I, int64_t K, int b_oc, int64_t lda,
                           int64_t ldb, int64_t ldh, int64_t lddh, void* stream);

/* autograd backward of experts_gemm w.r.t. weight:  dW[e] (K x N) (+)= A[s_e:s_e+n_e]^T * dY[s_e:s_e+n_e].
 * Experts with zero rows get zeros (or keep dW when accumulate). */
int aria_grouped_gemm_wgrad_bf16(const void* A, const void* dY, void* dW, const int32_t* offsets, int64_t E, int64_t K,
                                 int64_t N, int64_t lda, int64_t ldy, int c_f32, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MoE routing / dispatch (moe.hip)
 * ------------------------------------------------------------------------------------------------ */
/* TopKRouter.routing, aria/model/moe_lm.py:243-273 (eval part): top-k over E logits per token with the
 * deterministic tie rule "lowest expert id wins" (SURVEY.md section 8a-R), softmax over the k selected logits
 * in fp32 then cast to the logits dtype (:262), histogram of selected ids (:264-269).
 *   logits [T,E] bf16 (logits_f32=0) or fp32;  scores [T,k] same dtype as logits;  indices int32 [T,k]
 *   (descending logit order, like torch.topk);  counts int32[E] (zeroed here, then accumulated). */
/* K1: TopKRouter.forward (moe_lm.py:190-201 gating + :243-293 routing) as ONE launch: logits [T, E] = x [T, D] W^T (W [E, D], the
 * router's weight) on the matrix pipe with the routing above as its epilogue.  Writes logits (bf16: the backward's input), scores [T, k]
 * bf16, indices int32 [T, k], counts int32 [E] (zeroed here).  Bit-identical to aria_gemm_bf16 followed by aria_moe_route (same accumulation
 * order, same tie rule, same softmax).  E = 32 or 64, D % 256 == 0, k <= 8: anything else returns ARIA_ERR_UNSUPPORTED (run the two calls). */
int aria_moe_router_fused(const void* x, const void* w, void* logits, void* scores, int32_t* indices, int32_t* counts, int64_t T, int64_t D,
                          int64_t E, int64_t k, int64_t ldx, void* stream);
int aria_moe_route(const void* logits, int logits_f32, void* scores, int32_t* indices, int32_t* counts, int64_t T,
                   int64_t E, int64_t k, void* stream);

/* TokenDispatcher.token_permutation bookkeeping, aria/model/moe_lm.py:326-334: stable sort of the flattened
 * expert ids.  sorted_src[p] = flat index t*k+j of the p-th row in expert-major order (== the reference's
 * `sorted_indices`), inv[t*k+j] = p, offsets = exclusive scan of counts.  workspace: int32[(nchunks*2)*64 + 64]
 * with nchunks = ceil(T*k / 2048). */
int aria_moe_sort(const int32_t* indices, const int32_t* counts, int32_t* offsets, int32_t* sorted_src, int32_t* inv,
                  int32_t* workspace, int64_t T, int64_t E, int64_t k, void* stream);

/* permuted[p, :] = x[sorted_src[p] / k, :]   (index_select, moe_lm.py:330). D % 8 == 0. */
int aria_moe_permute(const void* x, const int32_t* sorted_src, void* permuted, int64_t M, int64_t D, int64_t k,
                     int64_t ldx, void* stream);

/* TokenDispatcher.token_unpermutation, moe_lm.py:336-365, fused with `output += shared` (:575-576):
 *   out[t] = bf16( sum_j fp32( bf16(expert_out[inv[t*k+j]] * scores[t,j]) ) ) ; if add: out = bf16(out + add[t]).
 * scores == NULL -> plain sum (used as the backward of the permute gather). */
int aria_moe_unpermute(const void* expert_out, const int32_t* inv, const void* scores, const void* add, void* out,
                       int64_t T, int64_t D, int64_t k, void* stream);
/* The same followed by the decoder layer's residual add (`hidden = residual + moe(hidden)`, LlamaDecoderLayer.forward via moe_lm.py:617-627):
 * out = bf16(residual + bf16(unpermute (+ add))) -- the rounding sequence of aria_moe_unpermute + aria_add_bf16, one launch less.
 * residual [T, D] bf16 or NULL (= aria_moe_unpermute).  Compile-time row widths only (D = 2560 / k = 6, D = 512 / k = 2): ARIA_ERR_UNSUPPORTED
 * otherwise. */
int aria_moe_unpermute_res(const void* expert_out, const int32_t* inv, const void* scores, const void* add, const void* residual,
                           void* out, int64_t T, int64_t D, int64_t k, void* stream);

/* backward of token_unpermutation: d_expert_out[inv[t,j]] = bf16(dout[t] * scores[t,j]);
 * dscores[t,j] = <expert_out[inv[t,j]], dout[t]> (fp32 reduce, stored bf16). */
int aria_moe_unpermute_bwd(const void* dout, const void* expert_out, const int32_t* inv, const void* scores,
                           void* d_expert_out, void* dscores, int64_t T, int64_t D, int64_t k, void* stream);

/* backward of TopKRouter.routing incl. the training-only auxiliary losses (moe_lm.py:128-166, 203-241,
 * MoEAuxLossAutoScaler :84-125): dlogits[T,E] (bf16) from dscores, z-loss and load-balancing loss.
 * aux_scale = MoEAuxLossAutoScaler.main_loss_backward_scale; coefficients 0 disable a term. */
int aria_moe_route_bwd(const void* logits, const int32_t* indices, const void* scores, const void* dscores,
                       const int32_t* counts, void* dlogits, int64_t T, int64_t E, int64_t k, float z_coeff,
                       float aux_coeff, float aux_scale, void* stream);

/* GroupedMLP.glu, moe_lm.py:505-507 (and LlamaMLP's silu(gate)*up):  act = bf16(bf16(silu(a)) * b).
 * two_inputs = 0: h is [M, 2I], a = h[:, :I], b = h[:, I:];  two_inputs = 1: a = h, b = h2, each [M, I]. */
int aria_swiglu_fwd(const void* h, const void* h2, void* act, int64_t M, int64_t I, void* stream);
/* dh (same layout as h/h2) from dact. */
int aria_swiglu_bwd(const void* h, const void* h2, const void* dact, void* dh, void* dh2, int64_t M, int64_t I,
                    void* stream);

/* backward of the embed_tokens lookup (aria/model/modeling_aria.py:250): dW[ids[t], :] += dy[t, :] (dW bf16, caller
 * zero-initialises; ids < 0 are skipped).  The forward lookup is aria_moe_permute with k = 1. */
int aria_embedding_bwd(const void* dy, const int32_t* ids, void* dw, int64_t T, int64_t D, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Norms / RoPE / elementwise (norm.hip)
 * ------------------------------------------------------------------------------------------------ */
/* LlamaRMSNorm.forward, transformers/models/llama/modeling_llama.py:62-67 (== gptfast/model.py:461-472), optionally
 * fused with the residual add that precedes it in LlamaDecoderLayer.forward (:295-325):
 *   if res: h = bf16(x + res), written to h_out;  y = bf16(w * bf16(h * rsqrt(mean(h^2) + eps)));  rstd fp32 [T] saved. */
int aria_rmsnorm_fwd(const void* x, const void* res, const void* w, void* h_out, void* y, float* rstd, int64_t T,
                     int64_t D, float eps, void* stream);
/* dx (+= dres if given), partial dw [nblocks, D] fp32 (reduced by aria_colsum_f32). */
int aria_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dres, void* dx,
                     float* dw_partial, int64_t nblocks, int64_t T, int64_t D, void* stream);
/* out[D] (bf16, (+)=) = sum over rows of partial[nrows, D] fp32 */
int aria_colsum_f32(const float* partial, void* out, int64_t nrows, int64_t D, int accumulate, void* stream);

/* apply_rotary_pos_emb, transformers/models/llama/modeling_llama.py:130-160 (half-split rotate), in place on the
 * q and k column blocks of a [T, ld] activation: x = bf16(bf16(x*cos) + bf16(rot(x)*sin)), cos/sin bf16 [S, hd]
 * tables built on the host exactly like LlamaRotaryEmbedding (:96-127).  pos = t % S.  inverse != 0 applies the
 * transposed rotation (backward). */
int aria_rope_inplace(void* x, const void* cos, const void* sin, int64_t T, int64_t S, int64_t n_heads, int64_t hd,
                      int64_t ld, int inverse, void* stream);

/* gptfast apply_rotary_emb (gptfast/model.py:519-531) with the bf16 freqs_cis cache of precompute_freqs_cis (:500-516,
 * [S, hd/2, 2]): interleaved pairs, fp32 arithmetic, one rounding; in place on n_heads consecutive heads of each row.
 * pos int32[T] (device; the decode cursor) or NULL (position = t % S). */
int aria_rope_interleaved_inplace(void* x, const void* freqs_cis, const int32_t* pos, int64_t T, int64_t S, int64_t n_heads,
                                  int64_t hd, int64_t ld, void* stream);

/* AdamW step with fp32 master weights on a contiguous (shard of a) parameter: the optimizer the reference's recipe runs through
 * DeepSpeed ZeRO-2 (recipes/config_full.yaml:25-29 lr 5e-6, weight_decay 0.1, adam_beta2 0.95; accelerate_configs/zero2.yaml).
 * param/grad bf16 [n], master/m/v fp32 [n]; grad is multiplied by grad_scale first (1/grad_accum, clipping). n % 2 == 0. */
int aria_adamw_step(void* param, const void* grad, float* master, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int64_t step, float grad_scale, void* stream);

/* *out (+)= sum_i x[i]^2 in fp32 over a contiguous bf16 vector (any element alignment: a shard of a flattened gradient): one term of the global gradient norm that HF
 * Trainer's max_grad_norm clipping needs (the reference runs it through DeepSpeed: recipes/accelerate_configs/zero2.yaml:5
 * gradient_clipping: auto -> TrainingArguments.max_grad_norm = 1.0).  Deterministic (per-block partials summed in index order).
 * workspace: ARIA_SUMSQ_WORKSPACE_FLOATS floats owned by the caller. */
#define ARIA_SUMSQ_WORKSPACE_FLOATS 1024
int aria_sumsq_bf16(const void* x, int64_t n, float* out, int accumulate, float* workspace, void* stream);

/* out = bf16(a + b), n elements (n % 8 == 0) */
int aria_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);
/* x[0..n) *= *scale in place; scale: ONE fp32 on the device (autograd's upstream gradient of a loss node: torch's `grad.mul_(dloss)` in the
 * fused loss nodes behind modeling_aria.py:301-323).  Reads the scalar on the device (no host sync); the identity (1.0f) touches no memory.
 * n % 8 == 0, x 16-byte aligned. */
int aria_scale_bf16(void* x, const float* scale, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Attention (attn.hip)
 * ------------------------------------------------------------------------------------------------ */
/* softmax(Q K^T * scale + mask) V, flash style (no Sq x Skv tensor), fp32 softmax
 * (eager_attention_forward transformers/models/llama/modeling_llama.py:192-215; gptfast/model.py:439-442;
 * nn.MultiheadAttention of the projector, aria/model/projector.py:73-102).
 * q,o: [B, Sq, H, hd] and k,v: [B, Skv, H, hd] views of token-major activations with row strides ld* (elements); head h at
 * column h*hd.  causal != 0: decoder mask (needs Sq == Skv).  kv_len int32[B] or NULL: keys >= kv_len[b] are masked (right
 * padding).  key_mask uint8 [B, Skv] or NULL: keys with 0 are masked (ViT patch padding, vision_encoder.py:132-152).
 * lse fp32 [B,H,Sq] saved for the backward.  hd in {64, 128} (the host zero-pads other head dims). */
int aria_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int32_t* kv_len, const uint8_t* key_mask,
                  int64_t B, int64_t Sq, int64_t Skv, int64_t H, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                  float scale, int causal, void* stream);
int aria_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                  float* delta /* fp32 [B,H,Sq] scratch */, void* dq, void* dk, void* dv, const int32_t* kv_len,
                  const uint8_t* key_mask, int64_t B, int64_t Sq, int64_t Skv, int64_t H, int64_t hd, int64_t ldq, int64_t ldk,
                  int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, float scale, int causal, void* stream);
/* The same with the INVERSE half-split RoPE of dq and dk in the two kernels' register epilogues (the chain rule through
 * apply_rotary_pos_emb, transformers/models/llama/modeling_llama.py:130-160, which LlamaAttention.forward :243-281 applies to q and k in
 * front of the attention: q, k here are the ROTATED tensors, dq / dk leave as gradients of the un-rotated projections).  rope_cos /
 * rope_sin: [rope_S, hd] bf16 tables (emb = cat(freqs, freqs)); the position of token t of a sequence is t % rope_S.  Bit-identical to
 * aria_attn_bwd followed by aria_rope_inplace(inverse) on dq | dk.  hd == 128 only (ARIA_ERR_UNSUPPORTED otherwise); both tables NULL =
 * aria_attn_bwd. */
int aria_attn_bwd_rope(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse, float* delta,
                       void* dq, void* dk, void* dv, const int32_t* kv_len, const uint8_t* key_mask, int64_t B, int64_t Sq, int64_t Skv,
                       int64_t H, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk,
                       int64_t lddv, float scale, int causal, const void* rope_cos, const void* rope_sin, int64_t rope_S, void* stream);

/* which backward the calling thread's last aria_attn_bwd ran: 2 = padded-tile pair (hd 64 / 72), 5 = role-split dK/dV + dQ v5 (hd 128) */
int aria_last_attn_bwd_variant(void);
/* which forward the calling thread's last aria_attn_fwd launched: 3 = attn_fwd3 (hd 128 / 72: software-pipelined score tiles, K / V by
 * LDS-DMA), 2 = attn_fwd2 (hd 64; or ARIA_ATTN_FWD=2: the round-3 kernels for A/B runs and the bit-identity tests) */
int aria_last_attn_fwd_variant(void);

/* ------------------------------------------------------------------------------------------------
 * ViT / projector support (vit.hip)
 * ------------------------------------------------------------------------------------------------ */
/* nn.LayerNorm (ViT layer_norm1/2, transformers/models/idefics2/modeling_idefics2.py:330-363; projector layer_norm / ln_kv /
 * ln_ffn, aria/model/projector.py:66-67,152): y = bf16((x - mean) * rstd * w + b), fp32 statistics saved for the backward. */
int aria_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, int64_t T, int64_t D,
                       float eps, void* stream);
/* dx and per-block partial dw/db [nblocks, D] fp32 (reduce with aria_colsum_f32). */
int aria_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, void* dx,
                       float* dw_partial, float* db_partial, int64_t nblocks, int64_t T, int64_t D, void* stream);
/* tanh-form GELU: ACT2FN["gelu_pytorch_tanh"] (ViT MLP) == ACT2FN["gelu_new"] (projector FFN, projector.py:40). n % 8 == 0. */
int aria_gelu_tanh_fwd(const void* x, void* y, int64_t n, void* stream);
int aria_gelu_tanh_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream);
/* AriaVisionModel._create_patch_attention_mask, aria/model/vision_encoder.py:132-145: pixel_mask uint8 [N,R,R] ->
 * patch_mask uint8 [N, R/patch, R/patch] (1 = patch contains at least one valid pixel). */
int aria_vit_patch_mask(const uint8_t* pixel_mask, uint8_t* patch_mask, int64_t N, int64_t R, int64_t patch, void* stream);
/* Idefics2VisionEmbeddings position ids (modeling_idefics2.py:141-170), fp32 exactly as the reference computes them on CPU;
 * boundaries = fp32 torch.arange(1/n_side, 1.0, 1/n_side) built on the host.  ids int32 [N, Hp*Wp], 0 on padded patches. */
int aria_vit_pos_ids(const uint8_t* patch_mask, const float* boundaries, int32_t* ids, int64_t N, int64_t Hp, int64_t Wp,
                     int64_t n_side, void* stream);
/* A operand of the patch-embed GEMM (Conv2d(3 -> hidden, k = s = patch), modeling_idefics2.py:117-133):
 * patches [N*Hp*Hp, KP] bf16 with KP >= C*patch*patch, KP % 8 == 0, zero padded.  pixels [N,C,R,R] bf16 or fp32. */
int aria_vit_im2col(const void* pixels, int pixels_f32, void* patches, int64_t N, int64_t C, int64_t R, int64_t patch, int64_t KP,
                    void* stream);
/* x[t,:] = bf16(x[t,:] + table[ids[t],:])  -- `embeddings + position_embedding(position_ids)` modeling_idefics2.py:172 */
int aria_gather_add_rows(void* x, const void* table, const int32_t* ids, int64_t T, int64_t D, void* stream);
/* partial[p, d] = sum over the p-th slice of rows of x[t, d] (bias gradients; reduce with aria_colsum_f32) */
int aria_colsum_bf16(const void* x, float* partial, int64_t nparts, int64_t T, int64_t D, int64_t ld, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Loss (loss.hip)
 * ------------------------------------------------------------------------------------------------ */
/* shifted, masked mean cross-entropy of aria/model/modeling_aria.py:301-323 on logits [T,V] bf16:
 * labels int32[T] are ALREADY shifted and masked by the host (-100 = ignore).  loss_sum fp32[1] and
 * count int32[1] are accumulated (caller zeroes); if dlogits != NULL it receives
 * (softmax - onehot) * grad_scale for counted rows, 0 for ignored rows (may alias logits). */
int aria_cross_entropy(const void* logits, const int32_t* labels, float* loss_sum, int32_t* count, void* dlogits,
                       float grad_scale, const int32_t* count_in /* device, or NULL: scale = grad_scale / max(1,*count_in) */,
                       int64_t T, int64_t V, int64_t ld, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Single-token decode engine (decode.hip) -- gptfast/model.py:178-234 (Transformer.forward, one new token, batch 1),
 * :318-366 (MOEFeedForward, T < 50 path), :67-93 (KVCache.update), :413-447 (Attention with the static cache).
 * ONE call enqueues every kernel of every layer for one token: GEMV projections with RMSNorm / residual / SwiGLU folded in,
 * RoPE + KV-cache write, flash attention over the cache, device-indexed routed experts.  The position is read from device
 * memory (ptrs[4]), so the call needs no host sync and its launch sequence is identical for every token.
 * Six launches per layer (qkv | attention | wo + residual | router logits + shared up-projection | routed up-projection with the
 * top-k inside | every down-projection + combine + residual); ARIA_DECODE_FUSE=0 selects the older seven-launch schedule
 * (bit-identical results: tests/model_cases.py::case_decode_engine_fused_schedule).
 *
 * ptrs (host array of device pointers): [0] freqs_cis bf16 [S_max, hd/2, 2]   [1] final norm weight [D]   [2] output weight [V, D]
 *   [3] scratch (aria_decode_scratch_bytes)   [4] pos int32 [1]   [5] input embedding bf16 [D]   [6] logits out bf16 [V]   [7] unused (non-NULL)
 *   then per layer (13 each): attention_norm [D], wqkv [3D, D] (q/k rows permuted for interleaved RoPE), wo [D, D], ffn_norm [D],
 *   gate [E, D], cond_ffn.w1 [E, I, D], w3 [E, I, D], w2 [E, D, I], shared_ffn.w1 [Is, D], w3 [Is, D], w2 [D, Is],
 *   k_cache [S_max, D], v_cache [S_max, D]      (= the gptfast model.pth tensors, unchanged)
 * dims (host int64): n_layers, D, n_heads, head_dim (64 | 128), E, top_k, I, I_shared, V, S_max.
 * ------------------------------------------------------------------------------------------------ */
#define ARIA_DECODE_HEADER_PTRS 8
#define ARIA_DECODE_LAYER_PTRS 13
int64_t aria_decode_scratch_bytes(const int64_t* dims);
int aria_decode_token(const void* const* ptrs, const int64_t* dims, float eps, void* stream);
/* Where, inside the scratch buffer, the last aria_decode_token call left every layer's routing record (TopKRouter of
 * gptfast/model.py:355-366 as the engine evaluated it): out[0], out[1] = byte offset and per-layer byte stride of the router logits (bf16 [L][E]),
 * out[2], out[3] = the chosen expert ids (int32 [L][top_k]), out[4], out[5] = their scores (bf16 [L][top_k]).  Read by the full-depth parity
 * case, which forces the oracle onto the engine's routing (tests/fullwidth_cases.py::case_decode_full_depth). */
int aria_decode_trace_layout(const int64_t* dims, int64_t* out);
/* sample() of gptfast/generate.py:35-58 for one token as ONE launch: keep the logits >= the top_k-th largest (ties kept; top_k <= 0 or >= V:
 * all), p_i ~ exp((l_i - max) / max(temperature, 1e-5)), out[0] = argmax_i p_i / q_i with q [V] fp32 the caller's Exp(1) draws
 * (multinomial_sample_one_no_sync's trick: the caller's generator stays the source of randomness).  logits bf16 [V]. */
int aria_sample_topk(const void* logits, const float* q, int64_t V, int64_t top_k, float temperature, int32_t* out, void* stream);
/* The routing step of the engine on its own (gptfast/model.py:359-363 for one token; what every workgroup of the routed up-projection
 * runs in front of its rows): top-k of E bf16 logits with ties to the lowest expert id, softmax over the selected logits in fp32,
 * scores as bf16 -- the same function as aria_moe_route on one row (bit-exact ids and scores). */
int aria_decode_route(const void* logits, int64_t E, int64_t k, void* scores, int32_t* idx, void* stream);
/* The attention step of the engine on its own (gptfast/model.py:413-447 with one new token): rotates q (in registers) and k with
 * freqs_cis[pos], writes k / v of the new token into the static cache at row pos (KVCache.update :67-93), then softmax(q K^T / sqrt(hd)) V
 * over cache rows 0..pos.  qkv bf16 [3 * H * hd] (q | k | v of the new token), caches bf16 [S_max, H * hd], pos int32 [1] on the device,
 * out bf16 [H * hd].  splits == 1: one workgroup per head; 2..32: heads x splits workgroups over contiguous key ranges + a merge
 * launch (flash-decoding), workspace of aria_decode_attn_workspace_bytes.  aria_decode_token uses the split form for S_max > 2048 when
 * ARIA_DECODE_SPLIT_KV is set (1 = one split per 1024 cache rows, N > 1 = N splits). */
int64_t aria_decode_attn_workspace_bytes(int64_t H, int64_t hd, int64_t splits);
int aria_decode_attn(const void* qkv, const void* freqs_cis, const int32_t* pos, void* k_cache, void* v_cache, void* out, int64_t H, int64_t hd,
                     int64_t splits, void* workspace, int64_t workspace_bytes, void* stream);
/* The same enqueue sequence captured once into a HIP graph (valid for every token because the position is device-side) and
 * replayed with one launch per token.  create returns NULL on failure (use aria_decode_token then); the tables must stay alive. */
void* aria_decode_graph_create(const void* const* ptrs, const int64_t* dims, float eps);
int aria_decode_graph_launch(void* graph, void* stream);
void aria_decode_graph_destroy(void* graph);

#ifdef __cplusplus
}
#endif
#endif /* ARIA_HIP_H */
