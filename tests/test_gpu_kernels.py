"""Hardware parity tests (-m gpu): the gfx950 build of libaria_hip.so, called through the C ABI, against the
CPU oracle on the same seeded inputs (cases in tests/kernel_cases.py) plus full-size property checks."""
import pytest
import torch

from tests import kernel_cases as C

bf16 = torch.bfloat16
pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def real_library():
    from aria_amd import hip

    assert not hip.is_emulated()
    lib = hip.get_lib()  # raises if libaria_hip.so is missing: no fallback
    assert lib.path.endswith("libaria_hip.so")
    assert torch.cuda.is_available()
    yield


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (1, 8, 8), (130, 264, 200), (1024, 3328, 2560)])
@pytest.mark.parametrize("a_oc,b_oc", [(False, False), (False, True), (True, True)])
def test_gemm_layouts(M, N, K, a_oc, b_oc):
    C.case_gemm_layouts(DEV, M, N, K, a_oc, b_oc)


def test_gemm_strided_views():
    C.case_gemm_strided_views(DEV)


@pytest.mark.parametrize("counts", [[3, 0, 130, 5, 0, 0, 70, 1], [0, 0, 0, 0], [256], [1, 1, 1],
                                    [37 * (i % 5) + (i * 7) % 11 for i in range(64)]])
def test_grouped_gemm(counts):
    C.case_grouped_gemm(DEV, counts)


def test_grouped_gemm_aria_width():
    C.case_grouped_gemm(DEV, [130, 0, 77, 300], K=2560, N=3328)


@pytest.mark.parametrize("T,E,k", [(33, 8, 3), (257, 64, 6), (5, 200, 4), (16384, 64, 6)])
@pytest.mark.parametrize("dtype", [bf16, torch.float32])
def test_route_bit_exact_with_ties(T, E, k, dtype):
    C.case_route(DEV, T, E, k, dtype, exact=False)


@pytest.mark.parametrize("T,E,k", [(33, 8, 3), (700, 64, 6), (1, 4, 2), (16384, 64, 6)])
def test_dispatch_matches_reference_order(T, E, k):
    C.case_dispatch(DEV, T, E, k, exact=False)


def test_moe_backward_pieces():
    C.case_moe_backward_pieces(DEV)


def test_swiglu():
    C.case_swiglu(DEV, exact=False)


@pytest.mark.parametrize("T,D", [(9, 64), (130, 2560), (3, 1152), (4099, 2560)])
def test_rmsnorm(T, D):
    C.case_rmsnorm(DEV, T, D, exact=False)


def test_rope():
    C.case_rope(DEV)


def test_add():
    C.case_add(DEV)


@pytest.mark.parametrize("T,V", [(7, 128), (19, 1000), (64, 100352)])
def test_cross_entropy(T, V):
    C.case_cross_entropy(DEV, T, V)


@pytest.mark.parametrize("B,S,H,hd,causal,use_len", [(2, 70, 2, 64, True, False), (1, 200, 1, 64, False, True),
                                                     (1, 130, 1, 128, True, False), (2, 64, 1, 64, False, False),
                                                     (2, 1024, 3, 128, True, False), (3, 1225, 2, 128, False, True),
                                                     (2, 70, 2, 72, False, True), (1, 200, 3, 72, False, False), (1, 130, 1, 72, True, False)])
def test_attention_fwd_bwd(B, S, H, hd, causal, use_len):
    C.case_attention(DEV, B, S, H, hd, causal, use_len)


@pytest.mark.parametrize("B,Sq,Skv,H,hd", [(2, 40, 150, 2, 64), (1, 130, 70, 1, 128), (3, 256, 4900, 2, 128), (2, 40, 150, 2, 72), (1, 256, 300, 16, 72)])
def test_attention_cross_masked(B, Sq, Skv, H, hd):
    C.case_attention_cross_masked(DEV, B, Sq, Skv, H, hd)


@pytest.mark.parametrize("hd,H,S", [(72, 16, 4900), (128, 2, 1000), (64, 2, 330)])
def test_attention_with_whole_key_tiles_masked(hd, H, S):
    C.case_attention_masked_tiles(DEV, hd, H, S=S)


@pytest.mark.parametrize("B,Sq,Skv,H,masked", [(2, 70, 70, 2, True), (1, 300, 90, 1, False), (2, 4900, 4900, 4, True), (3, 256, 4900, 16, True)])
def test_attention_hd72_forward(B, Sq, Skv, H, masked):
    C.case_attention_hd72_forward(DEV, B, Sq, Skv, H, masked)


@pytest.mark.parametrize("B,Sq,Skv,H,hd,causal,masked,use_len", [
    (1, 64, 64, 1, 128, True, False, False), (2, 200, 200, 2, 128, True, False, False), (1, 330, 330, 1, 128, False, True, False),
    (2, 150, 150, 1, 128, False, False, True), (1, 513, 513, 1, 128, True, False, True), (1, 16, 200, 2, 128, False, False, False),
    (1, 64, 64, 1, 72, False, False, False), (2, 300, 300, 2, 72, False, True, False), (1, 385, 385, 1, 72, False, False, True),
    (1, 129, 129, 1, 72, True, False, False), (2, 40, 330, 1, 72, False, True, False),
    (2, 2048, 2048, 20, 128, True, False, False), (1, 8192, 8192, 4, 128, True, False, True), (2, 4900, 4900, 16, 72, False, True, False),
    (2, 256, 4900, 16, 72, False, True, False),
])
def test_attention_forward_variants_give_the_same_bits(B, Sq, Skv, H, hd, causal, masked, use_len):
    C.case_attention_forward_variants(DEV, B, Sq, Skv, H, hd, causal, masked, use_len)


@pytest.mark.parametrize("force", ["1", "2", "3", None])
@pytest.mark.parametrize("M,N,K", [(40, 72, 64), (264, 136, 192), (2048, 4304, 1152), (2560, 2560, 4096)])
def test_gemm_fused_gelu(monkeypatch, force, M, N, K):
    if force:
        monkeypatch.setenv("ARIA_GEMM_FORCE", force)
    C.case_gemm_fused_gelu(DEV, M, N, K)


@pytest.fixture
def force_gemm_v2(monkeypatch):
    monkeypatch.setenv("ARIA_GEMM_FORCE", "2")


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (264, 136, 200), (8, 8, 8), (2048, 3328, 2560)])
@pytest.mark.parametrize("a_oc,b_oc", [(False, False), (False, True), (True, True)])
def test_gemm_v2_layouts(force_gemm_v2, M, N, K, a_oc, b_oc):
    C.case_gemm_layouts(DEV, M, N, K, a_oc, b_oc)


@pytest.mark.parametrize("counts", [[3, 0, 130, 5, 0, 0, 300, 1], [0, 0], [1, 1, 1], [37 * (i % 5) + (i * 7) % 11 for i in range(64)]])
def test_grouped_gemm_v2(force_gemm_v2, counts):
    C.case_grouped_gemm(DEV, counts)


def test_grouped_gemm_v2_aria_width(force_gemm_v2):
    C.case_grouped_gemm(DEV, [130, 0, 777, 300], K=2560, N=3328)


# v3: LDS-DMA staged, phase-scheduled 256x256 kernel (gemm3.hip); repeated launches double as a race screen
@pytest.fixture
def force_gemm_v3(monkeypatch):
    monkeypatch.setenv("ARIA_GEMM_FORCE", "3")


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (264, 136, 192), (40, 520, 128), (2048, 3328, 2560), (1000, 520, 1152),
                                   (264, 136, 200), (72, 264, 72), (1304, 1152, 4304)])
@pytest.mark.parametrize("a_oc,b_oc", [(False, False), (False, True), (True, True)])
def test_gemm_v3_layouts(force_gemm_v3, M, N, K, a_oc, b_oc):
    from aria_amd import hip

    for _ in range(3):
        C.case_gemm_layouts(DEV, M, N, K, a_oc, b_oc)
    assert hip.get_lib().cdll.aria_last_gemm_variant() == 3


@pytest.mark.parametrize("counts", [[3, 0, 130, 5, 0, 0, 300, 1], [1, 1, 1], [37 * (i % 5) + (i * 7) % 11 for i in range(64)]])
def test_grouped_gemm_v3(force_gemm_v3, counts):
    from aria_amd import hip

    for _ in range(3):
        C.case_grouped_gemm(DEV, counts, K=128, N=192)
    assert hip.get_lib().cdll.aria_last_gemm_variant() == 3  # the last call is the per-expert weight gradient (ragged reductions)


def test_grouped_gemm_v3_ragged_everything(force_gemm_v3):
    C.case_grouped_gemm(DEV, [3, 0, 130, 5, 0, 0, 300, 1])  # K = 72, N = 136: no dimension is a multiple of the tile


def test_grouped_gemm_v3_aria_width(force_gemm_v3):
    C.case_grouped_gemm(DEV, [130, 0, 777, 300], K=2560, N=3328)


@pytest.mark.parametrize("a_oc,b_oc", [(False, False), (False, True), (True, True)])
def test_gemm_v3_split_k(a_oc, b_oc):
    """default dispatch: a [2560,2560]-sized output with a long reduction takes the remainder split-K path (workspace + reduce)"""
    from aria_amd import hip, ops

    lib = hip.get_lib().cdll
    assert lib.aria_gemm_workspace_bytes(2560, 2560, 16384, int(a_oc), int(b_oc)) > 0
    C.case_gemm_layouts(DEV, 2560, 2560, 4096, a_oc, b_oc)
    assert lib.aria_last_gemm_variant() == 3
    # split and unsplit agree to fp32 summation-order noise
    import torch
    a = torch.randn(4096, 2560, device=DEV).to(torch.bfloat16)
    b = torch.randn(4096, 2560, device=DEV).to(torch.bfloat16)
    got = ops.gemm(a, b, a_oc=True, b_oc=True, out_dtype=torch.float32)
    ops.GEMM_SPLIT_K = False
    try:
        ref = ops.gemm(a, b, a_oc=True, b_oc=True, out_dtype=torch.float32)
    finally:
        ops.GEMM_SPLIT_K = True
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("counts,K,I,T", [([300, 0, 70, 5, 0, 0, 260, 1], 128, 128, 300), ([1536] * 8 + [1500, 1580], 2560, 1664, 4096)])
def test_gemm_swiglu_fused(counts, K, I, T):
    C.case_gemm_swiglu_fused(DEV, counts, K, I, T)


@pytest.mark.parametrize("T,E,k,K,N", [(4096, 64, 6, 2560, 3328), (16384, 64, 6, 2560, 3328), (1000, 8, 2, 264, 136)])
def test_grouped_gemm_wgrad_with_gathered_rows(T, E, k, K, N, monkeypatch):   # fc1's weight gradient through the dispatcher's index, bit for bit
    if K < 2560:
        monkeypatch.setenv("ARIA_GEMM_FORCE", "3")
    C.case_grouped_gemm_wgrad_gather(DEV, T, E, k, K, N)


@pytest.mark.parametrize("B,S,D,hd,K", [(2, 37, 256, 128, 128), (8, 2048, 2560, 128, 2560), (3, 1000, 2560, 128, 2560)])
def test_gemm_qkv_rope_hf_is_gemm_plus_rope(B, S, D, hd, K):   # q | k | v projection with the HF-form rotation as its epilogue, bit for bit
    assert C.case_gemm_qkv_rope_hf(DEV, B, S, D, hd, K) == (D >= 2560)


@pytest.mark.parametrize("T,D,E,k", [(70, 256, 64, 6), (16384, 2560, 64, 6), (4099, 2560, 64, 6), (33, 512, 32, 2)])
def test_router_fused_is_gemm_plus_route(T, D, E, k):   # K1: gating GEMM + top-k + softmax + histogram in one launch, bit for bit
    C.case_router_fused(DEV, T, D, E, k)


@pytest.mark.parametrize("counts,K,I,T,r", [([300, 0, 70, 5, 0, 0, 260, 1], 128, 128, 300, 8), ([1536] * 8 + [1500, 1580], 2560, 1664, 4096, 8),
                                              ([1536] * 8 + [1500, 1580], 2560, 1664, 4096, 24),
                                              ([1536] * 8 + [1500, 1580], 2560, 1664, 4096, 64)])   # r = 64 (lora_r 32 on gate + up): a full extension tile
def test_gemm_lora_k_extension(counts, K, I, T, r):  # LoRA's second projection inside the base launch, every base form
    C.case_gemm_lora_ext(DEV, counts, K, I, T, r, expect_fused=K >= 2560)


@pytest.mark.parametrize("counts,K,I,T", [([300, 0, 70, 5, 0, 0, 260, 1], 128, 128, 300), ([1536] * 8 + [1500, 1580], 2560, 1664, 4096)])
def test_gemm_swiglu_split(counts, K, I, T):  # gptfast wire format: w1 / w3 as two tensors of one allocation
    C.case_gemm_swiglu_split(DEV, counts, K, I, T)


@pytest.mark.parametrize("counts,K,I,T", [([300, 0, 70, 5, 0, 0, 260, 1], 128, 128, 300), ([130, 520], 192, 384, 72),
                                          ([1536] * 8 + [1500, 1580], 2560, 1664, 4096)])
def test_gemm_dswiglu_fused(counts, K, I, T):  # fc2 input gradient + glu backward in one launch == the two-step chain (Aria widths last)
    C.case_gemm_dswiglu_fused(DEV, counts, K, I, T)


@pytest.mark.parametrize("V,top_k,temperature", [(100352, 200, 0.8), (5000, 200, 0.8), (1000, 1, 1.0), (300, 500, 0.7), (4099, None, 1.3), (40, 7, 1e-6)])
def test_sample_topk_matches_the_tensor_path(V, top_k, temperature):
    C.case_sample_topk(DEV, V, top_k, temperature)


@pytest.mark.parametrize("E,k", [(64, 6), (8, 3), (200, 8), (256, 2)])
def test_decode_route_matches_the_batched_router(E, k):
    C.case_decode_route(DEV, E, k)


@pytest.mark.parametrize("H,hd,pos,splits", [(2, 128, 63, 2), (3, 128, 64, 2), (2, 128, 2999, 16), (20, 128, 16383, 32), (3, 64, 127, 2),
                                             (2, 64, 1000, 5)])
def test_decode_attention_split_kv(H, hd, pos, splits):
    """split-KV decode attention (the default beyond 2048 cache slots) against the fp32 reference and the single-workgroup form, on hardware"""
    C.case_decode_attention(DEV, H, hd, pos, splits)


@pytest.mark.parametrize("T,E,k,K,I", [(70, 8, 2, 64, 128), (300, 8, 3, 128, 384), (16384, 64, 6, 2560, 1664), (4099, 64, 6, 2560, 1664)])
def test_fused_swiglu_with_the_row_gather_in_the_loader(T, E, k, K, I):
    C.case_gemm_swiglu_gather(DEV, T, E, k, K, I)


@pytest.mark.parametrize("B,S,D,hd,K,S_cache,shuffled", [(1, 70, 256, 64, 64, 96, False), (2, 33, 256, 128, 128, 40, True), (1, 4099, 2560, 128, 2560, 4104, False), (2, 2048, 2560, 128, 2560, 2048, True)])
def test_qkv_projection_with_rope_and_cache_write_epilogue(B, S, D, hd, K, S_cache, shuffled):
    C.case_gemm_qkv_rope_cache(DEV, B, S, D, hd, K, S_cache, shuffled)


@pytest.mark.parametrize("T,E,k,D", [(70, 8, 2, 512), (16384, 64, 6, 2560), (4099, 64, 6, 2560)])
def test_dispatch_kernels_with_a_compile_time_row_width(T, E, k, D):
    C.case_dispatch_fixed_width(DEV, T, E, k, D)


@pytest.mark.parametrize("B,S,H,causal,use_len,s_rope", [(2, 70, 2, True, False, None), (8, 2048, 20, True, False, None), (1, 16384, 4, True, True, None), (2, 1000, 3, False, True, 1024)])
def test_attention_backward_with_the_inverse_rope_in_its_epilogue(B, S, H, causal, use_len, s_rope):
    C.case_attention_bwd_rope(DEV, B, S, H, causal, use_len, s_rope)


@pytest.mark.parametrize("n", [8, 4096, 100352 * 256])
def test_scale_by_a_device_scalar(n):
    C.case_scale_by_device_scalar(DEV, n)


@pytest.mark.parametrize("M,N,K,b_oc", [(78400, 1152, 1152, False), (4900, 1152, 4304, False), (513, 264, 192, False), (2048, 2560, 2560, True), (40, 72, 64, False)])
def test_gemm_accumulate_into_bf16_is_one_rounding(M, N, K, b_oc, monkeypatch):
    monkeypatch.setenv("ARIA_GEMM_FORCE", "3")     # the 256 x 256 kernels also at the toy sizes (their epilogues are what is under test)
    C.case_gemm_accumulate_exact(DEV, M, N, K, b_oc)


@pytest.mark.parametrize("M,N,K,a_oc,b_oc", [(512, 512, 2048, False, False), (300, 520, 1536, False, True), (7680, 2560, 16384, True, True), (2560, 3328, 16384, True, True), (16384, 2560, 7680, False, True)])
def test_gemm_split_k_slabs_in_accumulator_order(M, N, K, a_oc, b_oc):
    C.case_gemm_split_k_slabs(DEV, M, N, K, a_oc, b_oc)


@pytest.mark.parametrize("T,D,k", [(16384, 2560, 6), (70, 512, 2), (33, 128, 3)])
def test_unpermute_with_the_residual_add_as_its_last_step(T, D, k):
    C.case_unpermute_with_residual(DEV, T, D, k, E=8 if k < 6 else 64)


@pytest.mark.parametrize("T,D", [(78400, 1152), (37, 1152), (4901, 1152), (300, 64)])
def test_layernorm_with_two_rows_in_flight_gives_the_same_bits(T, D):
    C.case_layernorm_two_rows_in_flight(DEV, T, D)


def test_adamw_step_values_vs_fp32_adamw():   # the optimizer kernel's gfx950 build value-checked (decay groups, odd numel)
    C.case_adamw_values(DEV)


def test_grouped_tile_orders_cover_every_tile_once(monkeypatch):
    monkeypatch.setenv("ARIA_GEMM_FORCE", "3")
    C.case_grouped_tile_orders(DEV)
    C.case_grouped_tile_orders(DEV, T=6000, E=9, k=2, K=256, I=256, seed=78)
