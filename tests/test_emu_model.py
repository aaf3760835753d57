"""Module-level parity through the SIMT emulator (CPU): aria_amd modules vs the oracle on the golden fixtures."""
import pytest
import torch

from tests import model_cases as M

DEV = "cpu"


@pytest.fixture(scope="module", autouse=True)
def emu():
    from tests.emu import emu_lib

    emu_lib.install()
    yield
    emu_lib.uninstall()


def test_moe_layer_golden(golden):
    M.case_moe_layer_golden(DEV, golden)


def test_moe_layer_train_golden(golden):
    M.case_moe_layer_train_golden(DEV, golden)


def test_lm_golden(golden):
    M.case_lm_golden(DEV, golden)


@pytest.mark.parametrize("level", ["moe", "layer"])
def test_lm_golden_with_recompute_is_bit_identical(golden, level, monkeypatch):
    """recipes/config_full.yaml:17 gradient_checkpointing through the fused decoder node, both selective levels ("moe": the expert-row
    tensors are rebuilt in the backward; "layer": the layer re-runs with the flash (o, lse) kept): the reference fixture still holds and
    every gradient equals the non-recomputing run BIT FOR BIT (the same kernels see the same inputs)."""
    from aria_amd.moe_lm import AriaMoELMForCausalLM, load_reference_state_dict

    monkeypatch.setenv("ARIA_RECOMPUTE_LEVEL", level)
    M.case_lm_golden(DEV, golden, recompute=True)
    g = golden("lm")
    grads = []
    for rec in (False, True):
        lm = AriaMoELMForCausalLM(M.make_cfg(g["cfg"], gradient_checkpointing=rec))
        load_reference_state_dict(lm, g["weights"])
        lm.train()
        lm(input_ids=g["input_ids"], labels=g["input_ids"], return_logits=False).loss.backward()
        grads.append({n: p.grad.clone() for n, p in lm.named_parameters()})
    assert set(grads[0]) == set(grads[1])
    for n in grads[0]:
        assert torch.equal(grads[0][n], grads[1][n]), n


def test_moe_backward_fused_glu_epilogue_is_bit_identical(monkeypatch):
    """functional.moe_bwd with the SwiGLU backward as the epilogue of the two down-projection input gradients (gemm3_kernel<.., .., 5>, routed
    and shared) against the two-step chains (ARIA_FUSE_DSWIGLU=0): every gradient of a 1-layer LM at fusable widths (D 128, I 128) equal
    bit for bit, and the fused entry points are what ran."""
    from aria_amd import ops
    from aria_amd.moe_lm import AriaMoELMConfig, AriaMoELMForCausalLM

    cfg = AriaMoELMConfig(hidden_size=128, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=2, vocab_size=96,
                          moe_intermediate_size=128, moe_num_experts=8, moe_topk=2, moe_num_shared_experts=2, moe_z_loss_coeff=1e-5,
                          moe_aux_loss_coeff=1e-3)
    ids = torch.randint(1, 96, (2, 70), generator=torch.Generator().manual_seed(3))
    calls = {"grouped": 0, "dense": 0}
    og, od = ops.grouped_gemm_dswiglu, ops.gemm_dswiglu
    monkeypatch.setattr(ops, "grouped_gemm_dswiglu", lambda *a, **k: (calls.__setitem__("grouped", calls["grouped"] + 1), og(*a, **k))[1])
    monkeypatch.setattr(ops, "gemm_dswiglu", lambda *a, **k: (calls.__setitem__("dense", calls["dense"] + 1), od(*a, **k))[1])
    grads = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("ARIA_FUSE_DSWIGLU", fuse)
        torch.manual_seed(5)
        lm = AriaMoELMForCausalLM(cfg)
        with torch.no_grad():
            for n, p in lm.named_parameters():
                p.copy_((torch.ones(p.shape) if "norm" in n else torch.randn(p.shape) * 0.05).to(torch.bfloat16))
        lm.train()
        lm(input_ids=ids, labels=ids, return_logits=False).loss.backward()
        grads[fuse] = {n: p.grad.clone() for n, p in lm.named_parameters() if p.grad is not None}
        assert (calls["grouped"], calls["dense"]) == ((0, 0) if fuse == "0" else (1, 1)), calls
    assert set(grads["0"]) == set(grads["1"]) and len(grads["0"]) >= 12
    for n in grads["0"]:
        assert torch.isfinite(grads["1"][n].float()).all() and torch.equal(grads["0"][n], grads["1"][n]), n


def test_gptfast_gate_up_pairs_share_an_allocation_and_fuse(monkeypatch):
    """gptfast surface (model.pth wire format: cond_ffn.w1 / w3 and shared_ffn.w1 / w3 are separate tensors): setup_caches re-homes each
    pair in one allocation, the prefill then runs gate + up + SwiGLU as ONE launch (gemm3_kernel<false, false, 6>) -- logits equal to the
    three-launch chain bit for bit, the state-dict surface and the decode engine (which records parameter addresses) unaffected, and a
    ``.to()``-style re-allocation of the parameters is repaired lazily."""
    from aria_amd import gptfast as G
    from aria_amd import ops

    args = G.ModelArgs(block_size=64, vocab_size=136, n_layer=2, n_head=2, dim=128, intermediate_size=128, n_local_heads=2, head_dim=64,
                       rope_base=10000.0, norm_eps=1e-5, num_experts=8, router_topk=3, num_shared_experts=2)
    torch.manual_seed(21)
    m = G.Transformer(args)
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_((torch.ones(p.shape) + 0.1 * torch.randn(p.shape)).to(torch.bfloat16) if "norm" in n else (torch.randn(p.shape) * 0.08).to(torch.bfloat16))
    keys_before = list(m.state_dict())
    ref_sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.eval()
    m.setup_caches(1, 32)
    ff = m.layers[1].feed_forward
    assert ops.glu_split_fusable(ff.cond_ffn.w1, ff.cond_ffn.w3) and ops.glu_split_fusable(ff.shared_ffn.w1.weight, ff.shared_ffn.w3.weight)
    assert list(m.state_dict()) == keys_before and all(torch.equal(v, ref_sd[k]) for k, v in m.state_dict().items())
    calls = []
    og, od, ogg = ops.grouped_gemm_swiglu_split, ops.gemm_swiglu_split, ops.grouped_gemm_swiglu_split_gather
    monkeypatch.setattr(ops, "grouped_gemm_swiglu_split", lambda *a, **k: (calls.append("g"), og(*a, **k))[1])
    monkeypatch.setattr(ops, "gemm_swiglu_split", lambda *a, **k: (calls.append("d"), od(*a, **k))[1])
    monkeypatch.setattr(ops, "grouped_gemm_swiglu_split_gather", lambda *a, **k: (calls.append("G"), ogg(*a, **k))[1])
    ids = torch.randint(1, 136, (1, 24), generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        fused = m(ids, torch.arange(24)).float().clone()
        assert calls == ["G", "d"] * 2                      # r04: the routed launch also carries the dispatcher's row gather (K2)
        calls.clear()
        monkeypatch.setenv("ARIA_FUSE_GATHER", "0")         # ... and equals permute + the un-gathered launch bit for bit
        ungathered = m(ids, torch.arange(24)).float().clone()
        monkeypatch.delenv("ARIA_FUSE_GATHER")
        assert calls == ["g", "d"] * 2 and torch.equal(fused, ungathered)
        step = m(torch.tensor([[7]]), torch.tensor([24], dtype=torch.int32)).float().clone()   # decode engine on the re-homed parameters
        assert m._engine is not None
        monkeypatch.setenv("ARIA_FUSE_SWIGLU", "0")
        m.setup_caches(1, 32)
        plain = m(ids, torch.arange(24)).float().clone()
        step_plain = m(torch.tensor([[7]]), torch.tensor([24], dtype=torch.int32)).float().clone()
        monkeypatch.delenv("ARIA_FUSE_SWIGLU")
        assert torch.equal(fused, plain) and torch.equal(step, step_plain) and float(fused.abs().max()) > 0
        # parameters re-allocated one by one (what module.to(device) does): repaired at the next forward
        for p in m.parameters():
            p.data = p.data.clone()
        assert not ops.glu_split_fusable(ff.cond_ffn.w1, ff.cond_ffn.w3)
        calls.clear()
        again = m(ids, torch.arange(24)).float()
        assert calls == ["G", "d"] * 2 and torch.equal(again, fused) and ops.glu_split_fusable(ff.cond_ffn.w1, ff.cond_ffn.w3)


def test_vit_projector_golden(golden):
    M.case_vit_projector_golden(DEV, golden)


def test_aria_full_golden(golden):
    M.case_aria_full_golden(DEV, golden)


def test_lm_head_over_labelled_rows_only(golden, monkeypatch):
    """ARIA_LMHEAD_SKIP_MASKED=1: the lm_head GEMMs and the CE run over the positions that carry a label only (row gather / scatter, row
    count padded to 8) -- the full model's loss and gradients against the reference fixture are unchanged (labels of the fixture mask the
    prompt and the padding)."""
    monkeypatch.setenv("ARIA_LMHEAD_SKIP_MASKED", "1")
    M.case_aria_full_golden(DEV, golden)


def test_gptfast_golden(golden):
    M.case_gptfast_golden(DEV, golden)

def test_lora_grouped_gemm():
    M.case_lora_grouped_gemm(DEV)


def test_lora_linear_lm():
    M.case_lora_linear_lm(DEV)


@pytest.mark.parametrize("head_dim,max_seq", [(128, 16400)])  # (64, 24), (128, 24) run on hardware; hd 64 is covered by the split-KV case below
def test_decode_engine(head_dim, max_seq):
    M.case_decode_engine(DEV, head_dim, max_seq)


def test_decode_engine_split_kv(monkeypatch):
    """opt-in flash-decoding form of the engine's attention (ARIA_DECODE_SPLIT_KV): 4 key ranges for S_max = 4000, all but the first empty
    at these positions; tests/test_emu_kernels.py::test_decode_attention_split_kv covers populated ranges."""
    monkeypatch.setenv("ARIA_DECODE_SPLIT_KV", "1")
    M.case_decode_engine(DEV, 64, 4000)


def test_hf_to_gptfast_bridge(golden):
    M.case_hf_to_gptfast_bridge(DEV, golden)


def test_decode_engine_reference_golden(golden):
    M.case_decode_engine_reference_golden(DEV, golden)


def test_seam_attention_interface_native_hd72_without_grad():
    """aria_amd.seams.attention_interface as the (frozen) ViT calls it: non-causal, head_dim 72, a key mask that is not a prefix, no grad ->
    the native hd-72 forward kernel (no padding to 128); against fp32 torch on the same bf16 operands."""
    import types

    from aria_amd import seams

    B, H, S, hd = 2, 2, 70, 72
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(B, H, S, hd, generator=g).to(torch.bfloat16) for _ in range(3))
    mask = torch.ones(B, S, dtype=torch.bool)
    mask[1, 5:9] = False
    mask[1, 60:] = False
    with torch.no_grad():
        out, w = seams.attention_interface(types.SimpleNamespace(is_causal=False), q, k, v, mask, scaling=hd ** -0.5)
    assert w is None and out.shape == (B, S, H, hd)
    sc = torch.einsum("bhqd,bhkd->bhqk", q.float(), k.float()) * hd ** -0.5
    sc = sc.masked_fill(~mask[:, None, None, :], float("-inf"))
    want = torch.einsum("bhqk,bhkd->bqhd", torch.softmax(sc, -1), v.float())
    M.rel_close(out, want, 2e-2, "hd 72 attention through the seam")


@pytest.mark.parametrize("force_v3", [False, True])
def test_lora_fused_sites_with_dropout(force_v3, monkeypatch):
    if force_v3:   # the K-extension launches themselves (otherwise toy shapes take the two-launch fallback inside ops.*_lora)
        monkeypatch.setenv("ARIA_GEMM_FORCE", "3")
    M.case_lora_fused_sites(DEV)


@pytest.mark.parametrize("force_v3", [False, True])
def test_lora_fused_node_matches_modular(force_v3, monkeypatch):
    if force_v3:
        monkeypatch.setenv("ARIA_GEMM_FORCE", "3")
    M.case_lora_fused_vs_modular(DEV, layers=1 if force_v3 else 2)   # (one layer under the emulated 256 x 256 kernels: the CPU suite's time)


def test_training_step_without_permuted_copy(monkeypatch):
    monkeypatch.setenv("ARIA_GEMM_FORCE", "3")   # (toy shapes: the 256 x 256 kernels -- the gathered weight gradient is one of them)
    M.case_training_step_without_permuted_copy(DEV, layers=1)


def test_adapted_decoder_layer_with_gradient_checkpointing():
    """recipes/config_lora.yaml runs with gradient_checkpointing: the module-by-module decoder layer recomputed in backward gives the same
    loss and the same LoRA gradients as the stored-activation run (dropout off: bit-identical)."""
    from aria_amd.lora import apply_lora_from_config
    from aria_amd.moe_lm import AriaMoELMForCausalLM

    d = dict(hidden_size=64, num_attention_heads=1, num_key_value_heads=1, num_hidden_layers=2, vocab_size=96, moe_intermediate_size=32,
             moe_num_experts=8, moe_topk=2, moe_num_shared_experts=2)
    results = []
    for ckpt in (False, True):
        torch.manual_seed(3)
        lm = AriaMoELMForCausalLM(M.make_cfg(d, gradient_checkpointing=ckpt))
        with torch.no_grad():
            for n, p in lm.named_parameters():
                p.copy_(torch.ones(p.shape) if "norm" in n else (torch.randn(p.shape) * 0.08).to(torch.bfloat16))
        apply_lora_from_config(lm, dict(lora_r=8, lora_alpha=16, lora_target_modules=["fc1", "q_proj", "down_proj"]))
        with torch.no_grad():
            for n, p in lm.named_parameters():
                if "lora_B" in n:
                    p.copy_((torch.randn(p.shape) * 0.05).to(torch.bfloat16))
        lm.train()
        ids = torch.randint(1, 96, (2, 17), generator=torch.Generator().manual_seed(1))
        out = lm(input_ids=ids, labels=ids)
        out.loss.backward()
        results.append((out.loss.detach().clone(), {n: p.grad.clone() for n, p in lm.named_parameters() if p.grad is not None}))
    (l0, g0), (l1, g1) = results
    assert torch.equal(l0, l1) and set(g0) == set(g1) and len(g0) >= 10
    assert all(torch.equal(g0[n], g1[n]) for n in g0)


@pytest.mark.skipif(__import__("os").environ.get("ARIA_SLOW_TESTS") != "1", reason="~3 min through the emulator (130 M multiply-adds per token "
                    "at Aria's widths); runs on hardware in tests/test_gpu_model.py, and here with ARIA_SLOW_TESTS=1")
def test_decode_engine_aria_width():
    M.case_decode_engine_aria_width(DEV, n_tokens=3)


def test_decode_engine_fused_schedule_is_bit_identical():
    M.case_decode_engine_fused_schedule(DEV)


@pytest.mark.skipif(__import__("os").environ.get("ARIA_SLOW_TESTS") != "1", reason="~5 min through the emulator; runs on hardware in "
                    "tests/test_gpu_model.py, and here with ARIA_SLOW_TESTS=1")
def test_decode_engine_fused_schedule_aria_width():
    M.case_decode_engine_fused_schedule(DEV, aria_width=True, n_tokens=2)


def test_frozen_lm_head_skips_its_weight_gradient(golden):
    """freeze_llm-style runs: with lm_head frozen the fused lm_head + CE node still returns the hidden-state gradient but no [V, D] GEMM."""
    from aria_amd import autograd as AG
    from aria_amd import ops

    g = golden("lm")
    V, D = g["weights"]["lm_head.weight"].shape
    torch.manual_seed(0)
    hn = (torch.randn(12, D) * 0.5).to(torch.bfloat16).requires_grad_(True)
    w = g["weights"]["lm_head.weight"].to(torch.bfloat16)
    labels = torch.randint(0, V, (12,), dtype=torch.int32)
    calls, inner = [], ops.gemm
    ops.gemm = lambda a, b, **k: (calls.append((bool(k.get("a_oc")), bool(k.get("b_oc")))), inner(a, b, **k))[1]
    try:
        AG.LMHeadLossFn.apply(hn, w.clone().requires_grad_(False), labels).backward()
        frozen_calls, d_frozen = list(calls), hn.grad.clone()
        calls.clear()
        hn.grad = None
        wt = w.clone().requires_grad_(True)
        AG.LMHeadLossFn.apply(hn, wt, labels).backward()
    finally:
        ops.gemm = inner
    assert (True, True) not in frozen_calls and (True, True) in calls           # the [V, D] weight-gradient GEMM only when it is wanted
    assert torch.equal(d_frozen, hn.grad) and wt.grad is not None


def test_frozen_parameters_skip_their_weight_gradient_gemms(golden):
    """freeze_llm_layers / partially frozen runs: the fused decoder-layer backward computes no weight gradient for a frozen parameter, and
    what it does compute is bit-identical to the all-trainable run."""
    from aria_amd import ops
    from aria_amd.moe_lm import AriaMoELMForCausalLM, load_reference_state_dict

    g = golden("lm")
    ids = g["input_ids"]

    def run(freeze):
        lm = AriaMoELMForCausalLM(M.make_cfg(g["cfg"]))
        load_reference_state_dict(lm, g["weights"])
        lm.train()
        for n, p in lm.named_parameters():
            if freeze(n):
                p.requires_grad_(False)
        counts = {"wgrad": 0, "gwgrad": 0}
        inner, ginner = ops.gemm, ops.grouped_gemm_wgrad

        def gemm(a, b, **k):
            counts["wgrad"] += bool(k.get("a_oc")) and bool(k.get("b_oc"))
            return inner(a, b, **k)

        def gw(*a, **k):
            counts["gwgrad"] += 1
            return ginner(*a, **k)

        ops.gemm, ops.grouped_gemm_wgrad = gemm, gw
        try:
            lm(input_ids=ids, labels=ids).loss.backward()
        finally:
            ops.gemm, ops.grouped_gemm_wgrad = inner, ginner
        return {n: (None if p.grad is None else p.grad.clone()) for n, p in lm.named_parameters()}, counts

    full, c_full = run(lambda n: False)
    frozen = lambda n: n.startswith("model.layers.0.") or n.endswith("layers.1.mlp.experts.fc1.weight") or "layers.1.self_attn.o_proj" in n  # noqa: E731
    part, c_part = run(frozen)
    assert c_full["gwgrad"] == 4 and c_part["gwgrad"] == 1                # fc1 + fc2 of two layers  ->  fc2 of layer 1 only
    assert c_part["wgrad"] <= c_full["wgrad"] - 6                         # layer 0: qkv, o, down, gate/up, router; layer 1: o
    for n in full:
        if frozen(n):
            assert part[n] is None, n
        else:
            assert torch.equal(part[n], full[n]), n


def test_projection_weights_are_packed_and_the_fused_operand_is_a_view():
    M.case_packed_projection_weights(DEV)
