"""Generate tests/golden/*.pt from the REAL reference (TEST INFRASTRUCTURE).

Run in the build container (``python oracle/make_golden.py``); needs
``/root/reference``.  Imports the reference's own ``aria/model`` code and
``gptfast/model.py`` through ``oracle/ref_shims.py`` (SURVEY.md F5 shims),
builds tiny seeded models with explicit N(0, 0.05) init of the parameters the
reference leaves as ``torch.empty``, runs them on CPU in fp32 and
records inputs, weights, intermediates, outputs and gradients.  The fixtures
travel to the GPU box; the reference does not.

Fixtures (all fp32 unless noted):
  moe_layer.pt      MoELayer eval-mode forward + every intermediate of the dispatcher
  moe_layer_train.pt MoELayer training-mode forward/backward (aux + z loss gradients)
  lm.pt             AriaMoELMForCausalLM: logits, loss, gradients (2 layers)
  vit.pt            AriaVisionModel with a padded pixel_mask + AriaProjector
  aria.pt           AriaForConditionalGeneration end to end: logits, loss, gradients
  gptfast.pt        gptfast Transformer logits for both MoE code paths (T<50 and T>=50)
"""
from __future__ import annotations

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_shims import load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

TEXT = dict(hidden_size=64, num_attention_heads=4, num_key_value_heads=4, num_hidden_layers=2, vocab_size=128,
            intermediate_size=64, moe_intermediate_size=32, moe_num_experts=8, moe_topk=3,
            moe_num_shared_experts=2, rms_norm_eps=1e-6, rope_theta=5_000_000.0, max_position_embeddings=512,
            moe_z_loss_coeff=1e-3, moe_aux_loss_coeff=1e-2, pad_token_id=0)
VISION = dict(hidden_size=48, num_attention_heads=4, num_hidden_layers=2, intermediate_size=96, patch_size=14,
              image_size=56, num_channels=3, layer_norm_eps=1e-6, hidden_act="gelu_pytorch_tanh")
P2Q = {16: 4, 4: 2}
IMG_TOKEN = 9


def seed_all_params(model, std=0.05, seed=0):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("layernorm.weight") or n.endswith("norm.weight") or "layer_norm" in n and n.endswith("weight") \
                    or "ln_" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(std * torch.randn(p.shape, generator=g))


def sd(model):
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


def main():
    ns = load_reference()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234)

    # ---------------- MoE layer (eval)
    tcfg = ns.moe.AriaMoELMConfig(**TEXT, attn_implementation="eager")
    layer = ns.moe.MoELayer(tcfg)
    seed_all_params(layer, seed=1)
    layer.eval()
    x = torch.randn(3, 11, 64, generator=g)
    with torch.no_grad():
        scores, idx, tpe = layer.router(x.view(-1, 64))
        permuted = layer.token_dispatcher.token_permutation(x, idx)
        fc1 = layer.experts.fc1(permuted, tpe)
        act = layer.experts.activation_func(fc1)
        fc2 = layer.experts.fc2(act, tpe)
        unperm = layer.token_dispatcher.token_unpermutation(fc2, scores)
        out = layer(x)
        logits = layer.router.gating(x.view(-1, 64))
    torch.save(dict(cfg=TEXT, weights=sd(layer), x=x, logits=logits, scores=scores, indices=idx,
                    tokens_per_expert=tpe, permuted=permuted,
                    sorted_indices=layer.token_dispatcher.reversed_input_permutation_mapping.clone(),
                    fc1_out=fc1, act=act, fc2_out=fc2, unpermuted=unperm, out=out),
               os.path.join(OUT, "moe_layer.pt"))

    # ---------------- MoE layer (training: aux losses inject gradients)
    layer.train()
    ns.moe.MoEAuxLossAutoScaler.set_loss_scale(torch.tensor(0.5))
    xg = x.clone().requires_grad_(True)
    y = layer(xg)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    torch.save(dict(cfg=TEXT, weights=sd(layer), x=x, gy=gy, out=y.detach(), aux_scale=0.5, dx=xg.grad.clone(),
                    grads={n: p.grad.clone() for n, p in layer.named_parameters()}),
               os.path.join(OUT, "moe_layer_train.pt"))
    ns.moe.MoEAuxLossAutoScaler.set_loss_scale(torch.tensor(1.0))
    layer.zero_grad()

    # ---------------- LM
    lm = ns.moe.AriaMoELMForCausalLM(tcfg)
    seed_all_params(lm, seed=2)
    lm.eval()
    ids = torch.randint(1, 128, (2, 17), generator=g)
    with torch.no_grad():
        lg = lm(input_ids=ids).logits
    lm.train()
    emb = lm.model.embed_tokens(ids).detach().requires_grad_(True)
    lg_t = lm(inputs_embeds=emb).logits
    loss = torch.nn.functional.cross_entropy(lg_t[:, :-1].reshape(-1, 128), ids[:, 1:].reshape(-1))
    loss.backward()
    torch.save(dict(cfg=TEXT, weights=sd(lm), input_ids=ids, logits=lg, train_logits=lg_t.detach(), loss=loss.detach(),
                    d_emb=emb.grad.clone(), grads={n: p.grad.clone() for n, p in lm.named_parameters() if p.grad is not None}),
               os.path.join(OUT, "lm.pt"))

    # ---------------- full model (ViT + projector + LM)
    acfg = ns.cfg.AriaConfig(vision_config={**VISION, "model_type": "aria_vision_model"},
                             text_config={**TEXT, "model_type": "aria_moe_lm"},
                             projector_patch_to_query_dict=P2Q, image_token_index=IMG_TOKEN,
                             attn_implementation="eager", pad_token_id=0)
    model = ns.mdl.AriaForConditionalGeneration(acfg)
    seed_all_params(model, seed=3)
    model.eval()
    pv = torch.randn(2, 3, 56, 56, generator=g).clamp(-1, 1)
    pm = torch.ones(2, 56, 56, dtype=torch.bool)
    pm[1, 42:, :] = False  # bottom 25% rows padded -> 3x4 valid patch grid
    pm[1, :, 28:] = False  # and right half -> 3x2 valid
    with torch.no_grad():
        vout, vatts = model.vision_tower(pv, pixel_mask=pm)
        feat = vout.last_hidden_state
        pj = model.multi_modal_projector(feat, attn_mask=vatts)
        pv_small = torch.randn(1, 3, 28, 28, generator=g).clamp(-1, 1)
        vout_s, _ = model.vision_tower(pv_small, pixel_mask=torch.ones(1, 28, 28, dtype=torch.bool))
        pj_s = model.multi_modal_projector(vout_s.last_hidden_state, attn_mask=None)
    torch.save(dict(vision_cfg=VISION, p2q=P2Q, weights=sd(model), pixel_values=pv, pixel_mask=pm,
                    last_hidden_state=feat, image_atts=vatts, projected=pj,
                    pixel_values_small=pv_small, last_hidden_state_small=vout_s.last_hidden_state,
                    projected_small=pj_s),
               os.path.join(OUT, "vit.pt"))

    S = 20
    ids = torch.randint(10, 128, (2, S), generator=g)
    ids[0, 2:6] = IMG_TOKEN
    ids[1, 5:9] = IMG_TOKEN
    am = torch.ones(2, S, dtype=torch.long)
    am[1, 17:] = 0
    labels = ids.clone()
    labels[:, :8] = -100
    with torch.no_grad():
        o = model(input_ids=ids, pixel_values=pv, pixel_mask=pm, attention_mask=am, labels=labels)
    eval_logits, eval_loss = o.logits.clone(), o.loss.clone()
    model.train()
    model.zero_grad()
    o = model(input_ids=ids, pixel_values=pv, pixel_mask=pm, attention_mask=am, labels=labels)
    o.loss.backward()
    keep = ("language_model.model.layers.0.mlp.router.weight", "language_model.model.layers.1.mlp.experts.fc1.weight",
            "language_model.model.layers.0.mlp.experts.fc2.weight", "language_model.model.layers.0.self_attn.q_proj.weight",
            "language_model.model.layers.1.mlp.shared_experts.down_proj.weight", "language_model.lm_head.weight",
            "language_model.model.layers.0.input_layernorm.weight", "multi_modal_projector.ffn.linear_out.weight",
            "multi_modal_projector.query", "vision_tower.vision_model.encoder.layers.0.self_attn.q_proj.weight",
            "vision_tower.vision_model.embeddings.patch_embedding.weight", "language_model.model.embed_tokens.weight")
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if n in keep}
    torch.save(dict(text_cfg=TEXT, vision_cfg=VISION, p2q=P2Q, image_token_index=IMG_TOKEN, weights=sd(model),
                    input_ids=ids, pixel_values=pv, pixel_mask=pm, attention_mask=am, labels=labels,
                    logits=eval_logits, loss=eval_loss, train_logits=o.logits.detach(), train_loss=o.loss.detach(),
                    grads=grads),
               os.path.join(OUT, "aria.pt"))

    # ---------------- gptfast second implementation
    from oracle.aria_oracle import LMConfig, hf_to_gptfast_llm
    gm = ns.gptfast
    lcfg = LMConfig(**{k: v for k, v in TEXT.items() if k in LMConfig.__dataclass_fields__})
    args = gm.ModelArgs(block_size=128, vocab_size=128, n_layer=2, n_head=4, dim=64, intermediate_size=32,
                        n_local_heads=4, head_dim=16, rope_base=5_000_000.0, norm_eps=1e-6, num_experts=8,
                        router_topk=3, num_shared_experts=2)
    tf = gm.Transformer(args)
    lm.eval()
    conv = hf_to_gptfast_llm({"language_model." + k: v for k, v in sd(lm).items()}, lcfg)
    missing = tf.load_state_dict(conv, strict=False)
    assert not [m for m in missing.missing_keys if "kv_cache" not in m and "freqs" not in m and "mask" not in m], missing
    tf.eval()
    res = {}
    for name, T in (("short", 17), ("long", 64)):
        idsg = torch.randint(1, 128, (1, T), generator=g)
        with torch.device("cpu"):
            tf.setup_caches(1, T, training=True)
        with torch.no_grad():
            lg_g = tf(idsg)
            lg_h = lm(input_ids=idsg).logits
        res[name] = dict(input_ids=idsg, gptfast_logits=lg_g, hf_logits=lg_h)
        print(f"gptfast vs aria/model ({name}, T={T}): max abs diff {(lg_g - lg_h).abs().max().item():.3e}")
    torch.save(dict(cfg=TEXT, weights=sd(lm), gptfast_weights=conv, **res), os.path.join(OUT, "gptfast.pt"))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
