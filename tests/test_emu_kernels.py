"""CPU-side parity of the HIP kernels' *source* through the SIMT emulator (tests/emu): the same .hip files
compiled for the host.  Checks index arithmetic, masking and barrier placement against the oracle at small
sizes; the hardware parity tests (tests/test_gpu_kernels.py, -m gpu) run the gfx950 build through the same cases."""
import pytest
import torch

from tests import kernel_cases as C

bf16 = torch.bfloat16
DEV = "cpu"


@pytest.fixture(scope="module", autouse=True)
def emu():
    from tests.emu import emu_lib

    emu_lib.install()
    yield
    emu_lib.uninstall()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (1, 8, 8), (130, 264, 200)])
@pytest.mark.parametrize("a_oc,b_oc", [(False, False), (False, True), (True, True)])
def test_gemm_layouts(M, N, K, a_oc, b_oc):
    C.case_gemm_layouts(DEV, M, N, K, a_oc, b_oc)


def test_gemm_strided_views():
    C.case_gemm_strided_views(DEV)


@pytest.mark.parametrize("counts", [[3, 0, 130, 5, 0, 0, 70, 1], [0, 0, 0, 0], [256], [1, 1, 1]])
def test_grouped_gemm(counts):
    C.case_grouped_gemm(DEV, counts)


@pytest.mark.parametrize("T,E,k", [(33, 8, 3), (257, 64, 6), (5, 200, 4)])
@pytest.mark.parametrize("dtype", [bf16, torch.float32])
def test_route_bit_exact_with_ties(T, E, k, dtype):
    C.case_route(DEV, T, E, k, dtype, exact=True)


@pytest.mark.parametrize("T,E,k", [(33, 8, 3), (700, 64, 6), (1, 4, 2)])
def test_dispatch_matches_reference_order(T, E, k):
    C.case_dispatch(DEV, T, E, k, exact=True)


def test_moe_backward_pieces():
    C.case_moe_backward_pieces(DEV)


def test_swiglu():
    C.case_swiglu(DEV, exact=True)


@pytest.mark.parametrize("T,D", [(9, 64), (130, 2560), (3, 1152)])
def test_rmsnorm(T, D):
    C.case_rmsnorm(DEV, T, D, exact=True)


def test_rope():
    C.case_rope(DEV)


def test_add():
    C.case_add(DEV)


@pytest.mark.parametrize("T,V", [(7, 128), (19, 1000)])
def test_cross_entropy(T, V):
    C.case_cross_entropy(DEV, T, V)


@pytest.mark.parametrize("B,S,H,hd,causal,use_len", [(2, 70, 2, 64, True, False), (1, 200, 1, 64, False, True),
                                                     (1, 130, 1, 128, True, False), (2, 64, 1, 64, False, False),
                                                     (2, 200, 2, 128, False, True), (1, 333, 1, 128, True, False),
                                                     (2, 70, 2, 72, False, True), (1, 200, 3, 72, False, False), (1, 130, 1, 72, True, False)])
def test_attention_fwd_bwd(B, S, H, hd, causal, use_len):
    C.case_attention(DEV, B, S, H, hd, causal, use_len)


@pytest.mark.parametrize("hd,H", [(72, 2), (128, 1), (64, 1)])
def test_attention_with_whole_key_tiles_masked(hd, H):
    C.case_attention_masked_tiles(DEV, hd, H)


@pytest.mark.parametrize("defer", ["0", "1"])
def test_attention_bwd_default_kernels_under_both_dma_models(defer, monkeypatch):
    """The hd-128 backward kernels stage their tiles by LDS-DMA: pieces landing at once (adversarial for a slot restaged too early) and
    only at the counted wait (adversarial for a read not covered by wait + barrier); causal with a ragged tail, padded keys, Sq != Skv."""
    monkeypatch.setenv("ARIA_EMU_GLDS_DEFER", defer)
    C.case_attention(DEV, 1, 333, 2, 128, True, False)
    C.case_attention(DEV, 2, 200, 1, 128, False, True)
    C.case_attention_cross_masked(DEV, 1, 300, 170, 1, 128)


@pytest.mark.parametrize("B,Sq,Skv,H,hd", [(2, 40, 150, 2, 64), (1, 130, 70, 1, 128), (2, 40, 150, 2, 72), (1, 256, 300, 16, 72)])
def test_attention_cross_masked(B, Sq, Skv, H, hd):
    C.case_attention_cross_masked(DEV, B, Sq, Skv, H, hd)


@pytest.mark.parametrize("B,Sq,Skv,H,masked", [(2, 70, 70, 2, True), (1, 300, 90, 1, False), (1, 800, 130, 1, True)])
def test_attention_hd72_forward(B, Sq, Skv, H, masked):
    C.case_attention_hd72_forward(DEV, B, Sq, Skv, H, masked)


@pytest.mark.parametrize("B,Sq,Skv,H,hd,causal,masked,use_len", [
    (1, 64, 64, 1, 128, True, False, False), (2, 200, 200, 2, 128, True, False, False), (1, 330, 330, 1, 128, False, True, False),
    (2, 150, 150, 1, 128, False, False, True), (1, 513, 513, 1, 128, True, False, True), (1, 16, 200, 2, 128, False, False, False),
    (1, 64, 64, 1, 72, False, False, False), (2, 300, 300, 2, 72, False, True, False), (1, 385, 385, 1, 72, False, False, True),
    (1, 129, 129, 1, 72, True, False, False), (2, 40, 330, 1, 72, False, True, False),
])
def test_attention_forward_variants_give_the_same_bits(B, Sq, Skv, H, hd, causal, masked, use_len):
    C.case_attention_forward_variants(DEV, B, Sq, Skv, H, hd, causal, masked, use_len)


@pytest.mark.parametrize("force", ["1", "2", "3", None])
@pytest.mark.parametrize("M,N,K", [(40, 72, 64), (264, 136, 192), (72, 520, 128)])
def test_gemm_fused_gelu(monkeypatch, force, M, N, K):
    if force:
        monkeypatch.setenv("ARIA_GEMM_FORCE", force)
    C.case_gemm_fused_gelu(DEV, M, N, K)


@pytest.fixture
def force_gemm_v2(monkeypatch):
    monkeypatch.setenv("ARIA_GEMM_FORCE", "2")


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (264, 136, 200), (8, 8, 8)])
@pytest.mark.parametrize("a_oc,b_oc", [(False, False), (False, True), (True, True)])
def test_gemm_v2_layouts(force_gemm_v2, M, N, K, a_oc, b_oc):
    C.case_gemm_layouts(DEV, M, N, K, a_oc, b_oc)


@pytest.mark.parametrize("counts", [[3, 0, 130, 5, 0, 0, 300, 1], [0, 0], [1, 1, 1]])
def test_grouped_gemm_v2(force_gemm_v2, counts):
    C.case_grouped_gemm(DEV, counts)


# v3 (LDS-DMA staged, phase-scheduled).  The emulator's DMA model runs both ways: pieces land at once (a slot restaged too
# early clobbers its last reader) and pieces land only when a counted vmcnt wait forces them (a read not covered by a wait
# sees poison) -- together they bracket what the hardware may do.
@pytest.fixture(params=["0", "1"], ids=["dma-early", "dma-late"])
def force_gemm_v3(monkeypatch, request):
    monkeypatch.setenv("ARIA_GEMM_FORCE", "3")
    monkeypatch.setenv("ARIA_EMU_GLDS_DEFER", request.param)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (264, 136, 192), (40, 520, 128), (256, 256, 320), (264, 136, 200), (72, 264, 72)])
@pytest.mark.parametrize("a_oc,b_oc", [(False, False), (False, True), (True, True)])
def test_gemm_v3_layouts(force_gemm_v3, M, N, K, a_oc, b_oc):
    from aria_amd import hip

    C.case_gemm_layouts(DEV, M, N, K, a_oc, b_oc)
    assert hip.get_lib().cdll.aria_last_gemm_variant() == 3


@pytest.mark.parametrize("a_oc,b_oc", [(False, False), (False, True), (True, True)])
def test_gemm_v3_split_k(a_oc, b_oc):
    """Remainder split-K: 4 tiles, 16 K-steps -> every tile is computed by 2 workgroups into fp32 slabs + the reduce kernel
    (bias, accumulate, fp32 and bf16 outputs all go through it)."""
    from aria_amd import hip

    lib = hip.get_lib().cdll
    assert lib.aria_gemm_workspace_bytes(300, 264, 1024, int(a_oc), int(b_oc)) == 4 * 2 * 256 * 256 * 4
    assert lib.aria_gemm_workspace_bytes(300, 264, 40, int(a_oc), int(b_oc)) == 0
    C.case_gemm_layouts(DEV, 300, 264, 1024, a_oc, b_oc)
    assert lib.aria_last_gemm_variant() == 3


@pytest.mark.parametrize("M,N,a_oc,b_oc", [(8, 264, True, True), (520, 24, True, True), (264, 24, False, True), (264, 8, False, False)])
def test_gemm_v3_split_k_skinny_outputs(M, N, a_oc, b_oc):
    """The LoRA factors' gradients and projections: outputs 8 .. 24 wide (or tall) with a reduction over the tokens -- split along K into fp32
    slabs like the router's weight gradient, instead of a handful of 128 x 128 workgroups walking the whole reduction."""
    from aria_amd import hip

    lib = hip.get_lib().cdll
    assert lib.aria_gemm_workspace_bytes(M, N, 2048, int(a_oc), int(b_oc)) > 0
    C.case_gemm_layouts(DEV, M, N, 2048, a_oc, b_oc)
    assert lib.aria_last_gemm_variant() == 3


@pytest.mark.parametrize("counts", [[3, 0, 130, 5, 0, 0, 300, 1], [1, 1, 1]])
def test_grouped_gemm_v3(force_gemm_v3, counts):
    from aria_amd import hip

    C.case_grouped_gemm(DEV, counts, K=128, N=192)
    assert hip.get_lib().cdll.aria_last_gemm_variant() == 3  # the last call is the per-expert weight gradient (ragged reductions)


def test_grouped_gemm_v3_ragged_everything(force_gemm_v3):
    C.case_grouped_gemm(DEV, [3, 0, 130, 5, 0, 0, 300, 1])  # K = 72, N = 136: no dimension is a multiple of the tile


@pytest.mark.parametrize("counts,K,I,T", [([300, 0, 70, 5, 0, 0, 260, 1], 128, 128, 300), ([130, 520], 192, 256, 72)])
def test_gemm_swiglu_fused(force_gemm_v3, counts, K, I, T):
    C.case_gemm_swiglu_fused(DEV, counts, K, I, T)


@pytest.mark.parametrize("counts,K,I,T", [([300, 0, 70, 5, 0, 0, 260, 1], 128, 128, 300), ([130, 520], 192, 256, 72)])
def test_gemm_swiglu_split(force_gemm_v3, counts, K, I, T):
    C.case_gemm_swiglu_split(DEV, counts, K, I, T)


@pytest.mark.parametrize("counts,K,I,T,r", [([300, 0, 70, 5, 0, 0, 260, 1], 128, 128, 300, 8), ([130, 520], 192, 128, 72, 24), ([0, 0, 257], 64, 256, 9, 16),
                                              ([40, 0, 300], 128, 128, 40, 64), ([70, 9], 128, 128, 40, 56)])   # r = 64: a FULL extension tile
def test_gemm_lora_k_extension(force_gemm_v3, counts, K, I, T, r):
    C.case_gemm_lora_ext(DEV, counts, K, I, T, r)


@pytest.mark.parametrize("counts,K,I,T", [([300, 0, 70, 5, 0, 0, 260, 1], 128, 128, 300), ([130, 520], 192, 384, 72), ([0, 0, 257], 64, 256, 1)])
def test_gemm_dswiglu_fused(force_gemm_v3, counts, K, I, T):
    C.case_gemm_dswiglu_fused(DEV, counts, K, I, T)


@pytest.mark.parametrize("B,S,H", [(1, 600, 1), (2, 300, 2), (1, 1100, 5)])
def test_causal_forward_grouped_block_order_gives_the_same_bits(B, S, H, monkeypatch):
    """The XCD-grouped causal block order (the forward's from 32 K tokens up; here forced) only re-assigns blocks to workgroup ids."""
    import torch

    from aria_amd import ops

    D = H * 128
    qkv = torch.randn(B * S, 3 * D, generator=torch.Generator().manual_seed(S)).to(torch.bfloat16)
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("ARIA_ATTN_CAUSAL_GROUPED", mode)
        o, lse = ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, S, H, 128, 128 ** -0.5, True)
        outs.append((o.clone(), lse.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


# (r06: the last two cases give an expert 5 to 12 K-tiles -- from K-tile 2 on the indices reach the loader through the LDS slots, through the
# straight-line steady pairs and the general tail)
@pytest.mark.parametrize("T,E,k,K,N", [(150, 8, 2, 264, 136), (40, 4, 3, 64, 72), (700, 8, 1, 128, 256), (1100, 3, 1, 256, 256), (900, 4, 2, 64, 136)])
def test_grouped_gemm_wgrad_with_gathered_rows(force_gemm_v3, T, E, k, K, N):
    C.case_grouped_gemm_wgrad_gather(DEV, T, E, k, K, N)


@pytest.mark.parametrize("B,S,D,hd,K", [(2, 37, 256, 128, 128), (1, 300, 256, 64, 64)])
def test_gemm_qkv_rope_hf_is_gemm_plus_rope(force_gemm_v3, B, S, D, hd, K):
    assert C.case_gemm_qkv_rope_hf(DEV, B, S, D, hd, K)


@pytest.mark.parametrize("T,D,E,k", [(70, 256, 64, 6), (33, 512, 32, 2), (1, 256, 64, 1), (64, 256, 64, 8), (45, 2560, 64, 6)])
def test_router_fused_is_gemm_plus_route(T, D, E, k):
    C.case_router_fused(DEV, T, D, E, k)


@pytest.mark.parametrize("V,top_k,temperature", [(5000, 200, 0.8), (1000, 1, 1.0), (300, 500, 0.7), (4099, None, 1.3), (40, 7, 1e-6)])
def test_sample_topk_matches_the_tensor_path(V, top_k, temperature):
    C.case_sample_topk(DEV, V, top_k, temperature)


def test_sample_topk_degenerate_rows():
    """Every logit equal (the whole vocabulary ties at the threshold: all kept, the draw decides), masked (-inf) entries (never chosen),
    and a vocabulary smaller than one 16-byte chunk."""
    from aria_amd import gptfast as G
    from aria_amd import ops

    g = torch.Generator().manual_seed(5)
    for V, fill in ((1000, "const"), (1000, "masked"), (5, "rand")):
        logits = {"const": torch.full((V,), 0.75), "masked": torch.randn(V, generator=g), "rand": torch.randn(V, generator=g)}[fill].to(torch.bfloat16)
        if fill == "masked":
            logits[::2] = -float("inf")
        for _ in range(4):
            q = torch.empty(V).exponential_(1, generator=g)
            want = int(torch.argmax(G.logits_to_probs(logits, 0.8, 50) / q))
            got = int(ops.sample_topk(logits.clone(), q, 0.8, 50))
            assert got == want, (fill, got, want)
            assert fill != "masked" or got % 2 == 1


@pytest.mark.parametrize("E,k", [(64, 6), (8, 3), (200, 8), (256, 2), (64, 1)])
def test_decode_route_matches_the_batched_router(E, k):
    C.case_decode_route(DEV, E, k)


@pytest.mark.parametrize("H,hd,pos,splits", [(2, 128, 0, 4), (2, 128, 63, 2), (3, 128, 64, 2), (2, 128, 777, 3), (2, 128, 2999, 16),
                                             (2, 128, 1500, 32), (3, 64, 127, 2), (2, 64, 128, 2), (2, 64, 1000, 5)])
def test_decode_attention_split_kv(H, hd, pos, splits):
    C.case_decode_attention(DEV, H, hd, pos, splits)


@pytest.mark.parametrize("T,E,k,K,I", [(70, 8, 2, 64, 128), (300, 8, 3, 128, 384)])
def test_fused_swiglu_with_the_row_gather_in_the_loader(T, E, k, K, I):
    C.case_gemm_swiglu_gather(DEV, T, E, k, K, I)


@pytest.mark.parametrize("B,S,D,hd,K,S_cache,shuffled", [(1, 70, 256, 64, 64, 96, False), (2, 33, 256, 128, 128, 40, True)])
def test_qkv_projection_with_rope_and_cache_write_epilogue(B, S, D, hd, K, S_cache, shuffled):
    C.case_gemm_qkv_rope_cache(DEV, B, S, D, hd, K, S_cache, shuffled)


@pytest.mark.parametrize("T,E,k,D", [(70, 8, 2, 512), (37, 64, 6, 2560)])
def test_dispatch_kernels_with_a_compile_time_row_width(T, E, k, D):
    C.case_dispatch_fixed_width(DEV, T, E, k, D)


@pytest.mark.parametrize("B,S,H,causal,use_len,s_rope", [(2, 70, 2, True, False, None), (1, 300, 1, False, True, 512), (1, 130, 2, True, True, None)])
def test_attention_backward_with_the_inverse_rope_in_its_epilogue(B, S, H, causal, use_len, s_rope):
    C.case_attention_bwd_rope(DEV, B, S, H, causal, use_len, s_rope)


@pytest.mark.parametrize("n", [8, 4096])
def test_scale_by_a_device_scalar(n):
    C.case_scale_by_device_scalar(DEV, n)


@pytest.mark.parametrize("M,N,K,b_oc", [(300, 1152 // 4, 128, False), (513, 264, 192, False), (256, 512, 64, True), (40, 72, 64, False)])
def test_gemm_accumulate_into_bf16_is_one_rounding(M, N, K, b_oc, monkeypatch):
    monkeypatch.setenv("ARIA_GEMM_FORCE", "3")     # the 256 x 256 kernels also at the toy sizes (their epilogues are what is under test)
    C.case_gemm_accumulate_exact(DEV, M, N, K, b_oc)


@pytest.mark.parametrize("M,N,K,a_oc,b_oc", [(512, 512, 2048, False, False), (300, 520, 1536, False, True), (256, 264, 1024, True, True)])
def test_gemm_split_k_slabs_in_accumulator_order(M, N, K, a_oc, b_oc):
    C.case_gemm_split_k_slabs(DEV, M, N, K, a_oc, b_oc)


@pytest.mark.parametrize("T,D,k", [(70, 512, 2), (33, 128, 3)])
def test_unpermute_with_the_residual_add_as_its_last_step(T, D, k):
    C.case_unpermute_with_residual(DEV, T, D, k, E=8 if k < 6 else 64)


@pytest.mark.parametrize("T,D", [(1, 48), (37, 1152), (300, 64), (9000, 72)])
def test_layernorm_with_two_rows_in_flight_gives_the_same_bits(T, D):
    C.case_layernorm_two_rows_in_flight(DEV, T, D)


def test_adamw_step_values_vs_fp32_adamw():
    C.case_adamw_values(DEV)


def test_grouped_tile_orders_cover_every_tile_once(monkeypatch):   # (the order word only re-assigns tiles to workgroup ids: one DMA model)
    monkeypatch.setenv("ARIA_GEMM_FORCE", "3")
    C.case_grouped_tile_orders(DEV, T=600)
