"""Pin the oracle against the LIVE reference on fresh random inputs (only where /root/reference exists: the build
container).  On the GPU box the committed fixtures (tests/test_oracle_golden.py) play this role."""
import pytest
import torch

from oracle import aria_oracle as O
from oracle.ref_shims import load_reference, reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference checkout not present")

TEXT = dict(hidden_size=48, num_attention_heads=3, num_key_value_heads=3, num_hidden_layers=1, vocab_size=64,
            intermediate_size=48, moe_intermediate_size=16, moe_num_experts=6, moe_topk=2, moe_num_shared_experts=2,
            rms_norm_eps=1e-6, rope_theta=5_000_000.0, max_position_embeddings=128, pad_token_id=0)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_moe_layer_and_lm_match_live_reference(seed):
    ns = load_reference()
    torch.manual_seed(seed)
    cfg = ns.moe.AriaMoELMConfig(**TEXT, attn_implementation="eager")
    lm = ns.moe.AriaMoELMForCausalLM(cfg).eval()
    with torch.no_grad():
        for p in lm.parameters():
            p.normal_(0, 0.08)
    ids = torch.randint(1, 64, (2, 13))
    w = {k: v.detach() for k, v in lm.state_dict().items()}
    ocfg = O.LMConfig(**{k: v for k, v in TEXT.items() if k in O.LMConfig.__dataclass_fields__})
    with torch.no_grad():
        want = lm(input_ids=ids).logits
        got = O.lm_forward(w["model.embed_tokens.weight"][ids], w, ocfg)
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-4)
    layer = lm.model.layers[0].mlp
    x = torch.randn(2, 9, 48)
    with torch.no_grad():
        s, i, t = layer.router(x.view(-1, 48))
        o = layer(x)
    wl = {k[len("model.layers.0.mlp."):]: v for k, v in w.items() if k.startswith("model.layers.0.mlp.")}
    out, im = O.moe_layer(x, wl, "", ocfg, return_intermediates=True)
    assert torch.equal(im["indices"], i) and torch.equal(im["tokens_per_expert"], t)
    assert torch.allclose(out, o, atol=2e-5)


def test_lora_restatement_matches_the_reference_grouped_gemm_modules():
    """aria/lora/layers.py needs peft (absent here); its forward line (:129-139) is three calls of the reference's own GroupedGEMM
    module, which IS importable -- compose them and pin oracle.lora_grouped_gemm / lora_delta_weight against that."""
    ns = load_reference()
    torch.manual_seed(5)
    E, K, N, r, alpha = 5, 24, 40, 8, 16
    tpe = torch.tensor([3, 0, 11, 1, 6])
    base, a, b = ns.moe.GroupedGEMM(K, N, E), ns.moe.GroupedGEMM(K, r, E), ns.moe.GroupedGEMM(r, N, E)
    with torch.no_grad():
        for m in (base, a, b):
            m.weight.normal_(0, 0.3)
    x = torch.randn(int(tpe.sum()), K)
    scaling = alpha / r
    with torch.no_grad():
        want = base(x, tpe) + b(a(x, tpe), tpe) * scaling
        got = O.lora_grouped_gemm(x, base.weight, a.weight, b.weight, tpe, scaling)
        assert torch.allclose(got, want, atol=1e-5, rtol=1e-5)
        merged = ns.moe.GroupedGEMM(K, N, E)
        merged.weight.copy_(base.weight + O.lora_delta_weight(a.weight, b.weight, scaling))
        assert torch.allclose(merged(x, tpe), want, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("case", range(6))
def test_lm_matches_live_reference_over_random_shapes(case):
    """heads x head_dim, experts / top-k / shared experts, depth, vocabulary, eps, RoPE base, batch and length drawn per case (a 60-case sweep
    of the same generator was run once offline: no mismatch)."""
    import random

    ns = load_reference()
    rnd = random.Random(100 + case)
    H, hd = rnd.choice([1, 2, 3, 4]), rnd.choice([8, 16, 32])
    E = rnd.choice([2, 3, 6, 8, 16])
    k, inter, shared = rnd.randint(1, min(4, E)), rnd.choice([8, 16, 24]), rnd.choice([1, 2, 3])
    text = dict(hidden_size=H * hd, num_attention_heads=H, num_key_value_heads=H, num_hidden_layers=rnd.choice([1, 2, 3]),
                vocab_size=rnd.choice([32, 64, 101]), intermediate_size=inter * shared, moe_intermediate_size=inter, moe_num_experts=E,
                moe_topk=k, moe_num_shared_experts=shared, rms_norm_eps=rnd.choice([1e-5, 1e-6]), rope_theta=rnd.choice([1e4, 5e6]),
                max_position_embeddings=128, pad_token_id=0)
    torch.manual_seed(case)
    lm = ns.moe.AriaMoELMForCausalLM(ns.moe.AriaMoELMConfig(**text, attn_implementation="eager")).eval()
    with torch.no_grad():
        for p in lm.parameters():
            p.normal_(0, 0.1)
    ids = torch.randint(1, text["vocab_size"], (rnd.randint(1, 3), rnd.randint(1, 24)))
    w = {n: v.detach() for n, v in lm.state_dict().items()}
    ocfg = O.LMConfig(**{n: v for n, v in text.items() if n in O.LMConfig.__dataclass_fields__})
    with torch.no_grad():
        assert torch.allclose(O.lm_forward(w["model.embed_tokens.weight"][ids], w, ocfg), lm(input_ids=ids).logits, atol=5e-5, rtol=2e-4)


@pytest.mark.parametrize("case", range(4))
def test_vit_and_projector_match_live_reference_over_random_shapes(case):
    """tower width / heads / depth, image side (2..5 patches), query count and a random valid pixel rectangle per image (down to a single
    pixel) drawn per case; the key-padding mask must be equal, features of valid patches and the projector output within 5e-5
    (a 40-case sweep of the same generator was run once offline: no mismatch)."""
    import random

    ns = load_reference()
    rnd = random.Random(200 + case)
    H, hd, side = rnd.choice([1, 2, 4]), rnd.choice([8, 12, 16]), rnd.choice([2, 3, 4, 5])
    img = 14 * side
    vision = dict(hidden_size=H * hd, num_attention_heads=H, num_hidden_layers=rnd.choice([1, 2, 3]), intermediate_size=rnd.choice([16, 40, 96]),
                  patch_size=14, image_size=img, num_channels=3, layer_norm_eps=1e-6, hidden_act="gelu_pytorch_tanh")
    p2q = {side * side: rnd.choice([1, 2, 4])}
    acfg = ns.cfg.AriaConfig(vision_config={**vision, "model_type": "aria_vision_model"}, text_config={**TEXT, "model_type": "aria_moe_lm"},
                             projector_patch_to_query_dict=p2q, image_token_index=9, attn_implementation="eager", pad_token_id=0)
    torch.manual_seed(case)
    model = ns.mdl.AriaForConditionalGeneration(acfg).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.normal_(0, 0.1) if p.dim() > 1 else p.normal_(1.0 if ("norm" in n or "ln" in n) else 0.0, 0.05)
    N = rnd.randint(1, 3)
    pv = torch.randn(N, 3, img, img).clamp(-1, 1)
    pm = torch.zeros(N, img, img, dtype=torch.bool)
    for j in range(N):
        pm[j, :rnd.randint(1, img), :rnd.randint(1, img)] = True
    with torch.no_grad():
        vout, vatts = model.vision_tower(pv, pixel_mask=pm)
        want = model.multi_modal_projector(vout.last_hidden_state, attn_mask=vatts)
    w = {n: v.detach() for n, v in model.state_dict().items()}
    vc = O.VisionConfig(**{n: v for n, v in vision.items() if n in O.VisionConfig.__dataclass_fields__})
    feat, atts = O.vit_forward(pv, pm, w, "vision_tower.", vc)
    got = O.projector_forward(feat, atts, w, "multi_modal_projector.", O.AriaOracleConfig(vision=vc, patch_to_query=p2q, projector_heads=H))
    assert torch.equal(atts, vatts)
    assert torch.allclose(feat[~vatts], vout.last_hidden_state[~vatts], atol=5e-5, rtol=2e-4)
    assert torch.allclose(got, want, atol=5e-5, rtol=2e-4)


@pytest.mark.parametrize("case", range(4))
def test_moe_training_path_matches_live_reference_over_random_shapes(case):
    """MoELayer in train mode (z-loss and load-balancing gradients injected through MoEAuxLossAutoScaler with a random scale): output, dx and
    every parameter gradient of the oracle against the reference module (a 30-case sweep was run once offline: no mismatch)."""
    import random

    ns = load_reference()
    rnd = random.Random(300 + case)
    E = rnd.choice([2, 4, 6, 8])
    text = dict(TEXT, hidden_size=rnd.choice([16, 32, 48]), num_attention_heads=2, num_key_value_heads=2, moe_num_experts=E,
                moe_topk=rnd.randint(1, min(3, E)), moe_intermediate_size=rnd.choice([8, 16]), moe_num_shared_experts=rnd.choice([1, 2]),
                moe_z_loss_coeff=rnd.choice([0.0, 1e-3]), moe_aux_loss_coeff=rnd.choice([0.0, 1e-2]))
    torch.manual_seed(case)
    layer = ns.moe.MoELayer(ns.moe.AriaMoELMConfig(**text, attn_implementation="eager")).train()
    with torch.no_grad():
        for p in layer.parameters():
            p.normal_(0, 0.15)
    scale = rnd.choice([1.0, 0.5, 0.25])
    ns.moe.MoEAuxLossAutoScaler.set_loss_scale(torch.tensor(scale))
    O._AuxLossScaler.scale = scale
    try:
        x = torch.randn(2, rnd.randint(1, 12), text["hidden_size"])
        gy = torch.randn_like(x)
        xr = x.clone().requires_grad_(True)
        y = layer(xr)
        y.backward(gy)
        w = {n: v.detach().clone().requires_grad_(True) for n, v in layer.state_dict().items()}
        ocfg = O.LMConfig(**{n: v for n, v in text.items() if n in O.LMConfig.__dataclass_fields__})
        xo = x.clone().requires_grad_(True)
        yo = O.moe_layer(xo, w, "", ocfg, training=True)
        yo.backward(gy)
    finally:
        O._AuxLossScaler.scale = 1.0
        ns.moe.MoEAuxLossAutoScaler.set_loss_scale(torch.tensor(1.0))
    assert torch.allclose(yo, y, atol=2e-5, rtol=1e-4) and torch.allclose(xo.grad, xr.grad, atol=5e-5, rtol=2e-4)
    for n, p in layer.named_parameters():
        assert torch.allclose(w[n].grad, p.grad, atol=5e-5, rtol=2e-4), n
