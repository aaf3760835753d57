"""Module-level parity AT THE CONFIG WIDTHS OF BASELINE.json (VERDICT r1 "missing" #1): the golden fixtures are toy-sized (D 64, 8
experts), so at those sizes every GEMM dispatches to the 128x128 v1 kernel and the benchmarked 256x256 kernels (gemm2 / gemm3, the
expert-major XCD tile list) are only reached by kernel-level cases.  Here whole blocks run at D 2560 / 20 x 128 / 64 experts top-6 /
I 1664 (decoder), 1152 / 16 x 72 / 4304 on 4900 patches (ViT) and config #1 end to end, against ``oracle/aria_oracle.py`` in fp32 on
the same bf16-rounded weights and inputs (reference: aria/model/moe_lm.py:548-602, modeling_aria.py:194-335).

Every case takes its dimensions as arguments: the hardware suite (-m gpu, tests/test_gpu_fullwidth.py) passes the real ones, the CPU
suite runs the SAME code at reduced width through the SIMT emulator (tests/test_emu_fullwidth.py) so that the comparison logic
itself is exercised without a GPU.

Protocol.
  * Router: ids equal wherever the oracle's logit gaps around the k-th place exceed 2 bf16 ulps of the logit scale (the
    kernel sees bf16-rounded logits); the fraction of tokens with the same expert SET is reported and bounded.
  * Everything downstream of the router is compared on IDENTICAL routing (``O.forced_routing`` with the ids the device chose): a
    flipped token would otherwise show up as a ~10 % error of one expert's weight gradient although no arithmetic is wrong.
  * Metrics per tensor: relative L2 error, cosine, and max-abs error over max-abs of the reference -- all three bounded, so a wrong
    small-magnitude block cannot hide behind a large one (VERDICT r1 weak #1b).  The measured values are collected in ``REPORT`` and
    written to gpurun_out/fullwidth_parity.json by the hardware suite (copied to profiles/ as evidence).
"""
import json
import os

import torch

from oracle import aria_oracle as O

bf16 = torch.bfloat16
REPORT = {}


def metrics(got, want):
    got, want = got.detach().double().cpu().flatten(), want.detach().double().cpu().flatten()
    assert got.shape == want.shape, (got.shape, want.shape)
    diff = got - want
    wn = want.norm().clamp(min=1e-30)
    return {"rel_l2": float(diff.norm() / wn), "cos": float(torch.dot(got, want) / (got.norm().clamp(min=1e-30) * wn)),
            "max_rel": float(diff.abs().max() / want.abs().max().clamp(min=1e-30))}


def check(case, name, got, want, rel_l2, max_rel, cos=None):
    m = metrics(got, want)
    REPORT.setdefault(case, {})[name] = {k: round(v, 6) for k, v in m.items()}
    cos = 1.0 - 0.5 * rel_l2 * rel_l2 * 4 if cos is None else cos  # |a-b| <= e|b| implies cos >= ~1 - e^2/2; allow 4x
    assert m["rel_l2"] <= rel_l2 and m["max_rel"] <= max_rel and m["cos"] >= cos, (case, name, m, (rel_l2, max_rel, cos))


def dump_report(path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    prev = {}
    if os.path.exists(path):
        with open(path) as f:
            prev = json.load(f)
    prev.update(REPORT)
    with open(path, "w") as f:
        json.dump(prev, f, indent=1, sort_keys=True)


def randw(shape, gen, std=0.02):
    return (torch.randn(shape, generator=gen) * std).to(bf16)


def lm_weights(cfg: O.LMConfig, seed: int):
    """Reference state-dict keys (moe_lm.py: LlamaForCausalLM layout), N(0, 0.02) like the benchmark, norm weights 1 +- 0.1."""
    g = torch.Generator().manual_seed(seed)
    D, E, I, V = cfg.hidden_size, cfg.moe_num_experts, cfg.moe_intermediate_size, cfg.vocab_size
    I2 = I * cfg.moe_num_shared_experts
    Dq = cfg.num_attention_heads * cfg.head_dim
    w = {"model.embed_tokens.weight": randw((V, D), g), "lm_head.weight": randw((V, D), g),
         "model.norm.weight": (1 + 0.1 * torch.randn(D, generator=g)).to(bf16)}
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = (1 + 0.1 * torch.randn(D, generator=g)).to(bf16)
        w[p + "post_attention_layernorm.weight"] = (1 + 0.1 * torch.randn(D, generator=g)).to(bf16)
        for n in "qkv":
            w[p + f"self_attn.{n}_proj.weight"] = randw((Dq, D), g)
        w[p + "self_attn.o_proj.weight"] = randw((D, Dq), g)
        w[p + "mlp.router.weight"] = randw((E, D), g)
        w[p + "mlp.experts.fc1.weight"] = randw((E, D, 2 * I), g)
        w[p + "mlp.experts.fc2.weight"] = randw((E, I, D), g)
        w[p + "mlp.shared_experts.gate_proj.weight"] = randw((I2, D), g)
        w[p + "mlp.shared_experts.up_proj.weight"] = randw((I2, D), g)
        w[p + "mlp.shared_experts.down_proj.weight"] = randw((D, I2), g)
    return w


class _Recorder:
    """Records what the device router chose (ops.moe_route) and which GEMM kernel family every grouped GEMM dispatched to."""

    def __init__(self):
        from aria_amd import hip, ops

        self.ops, self.hip = ops, hip
        self.idx, self.logits, self.variants = [], [], []

    def __enter__(self):
        ops = self.ops
        self._route, self._gg, self._ggs, self._rf = ops.moe_route, ops.grouped_gemm, ops.grouped_gemm_swiglu, ops.moe_router_fused
        self._ggsg = ops.grouped_gemm_swiglu_gather

        def ggsg(*a, **kw):            # (K2: the fused fc1 + SwiGLU launch on gathered rows -- the training step's form since r05)
            r = self._ggsg(*a, **kw)
            self.variants.append(int(self.hip.get_lib().cdll.aria_last_gemm_variant()))
            self.fused += 1
            return r

        def router_fused(x, w, k):     # (K1: the gating GEMM and the routing as one launch -- the same record)
            r = self._rf(x, w, k)
            self.idx.append(r[2].detach().cpu().long())
            self.logits.append(r[0].detach().float().cpu())
            return r

        def route(logits, k):
            r = self._route(logits, k)
            self.idx.append(r[1].detach().cpu().long())
            self.logits.append(logits.detach().float().cpu())
            return r

        def gg(*a, **kw):
            r = self._gg(*a, **kw)
            self.variants.append(int(self.hip.get_lib().cdll.aria_last_gemm_variant()))
            return r

        def ggs(*a, **kw):
            r = self._ggs(*a, **kw)
            self.variants.append(int(self.hip.get_lib().cdll.aria_last_gemm_variant()))
            self.fused += 1
            return r

        self._rope, self._abwd = ops.rope_, ops.attention_bwd

        def rope_(x, cos, sin, S, n_heads, hd, inverse=False):   # (r05: the inverse rotation of dq | dk belongs to the attention backward)
            self.inverse_rope += bool(inverse)
            return self._rope(x, cos, sin, S, n_heads, hd, inverse=inverse)

        def abwd(*a, **kw):
            self.bwd_rope += kw.get("rope") is not None
            return self._abwd(*a, **kw)

        self.fused = self.inverse_rope = self.bwd_rope = 0
        ops.moe_route, ops.grouped_gemm, ops.grouped_gemm_swiglu, ops.moe_router_fused = route, gg, ggs, router_fused
        ops.grouped_gemm_swiglu_gather = ggsg
        ops.rope_, ops.attention_bwd = rope_, abwd
        return self

    def __exit__(self, *exc):
        self.ops.moe_route, self.ops.grouped_gemm, self.ops.grouped_gemm_swiglu, self.ops.moe_router_fused = self._route, self._gg, self._ggs, self._rf
        self.ops.grouped_gemm_swiglu_gather = self._ggsg
        self.ops.rope_, self.ops.attention_bwd = self._rope, self._abwd
        return False


class _OracleLogits:
    """Records the oracle's router logits per router call (to judge which tokens have a resolvable top-k)."""

    def __enter__(self):
        self.logits = []
        self._orig = O.router_routing

        def rr(logits, topk, num_experts):
            self.logits.append(logits.detach().clone())
            return self._orig(logits, topk, num_experts)

        O.router_routing = rr
        return self

    def __exit__(self, *exc):
        O.router_routing = self._orig
        return False


def router_parity(case, layer, dev_idx, dev_logits, logits, k, min_same=0.9, logit_tol=(2e-2, 6e-2), require_safe=True):
    """Device router vs the oracle's own top-k on its fp32 logits.  The device sees ITS logits (bf16 GEMM output on bf16 activations
    that carry the rounding of everything upstream); with delta_t = max_e |device logit - oracle logit| of token t, the two top-k
    ORDERS are provably the same whenever every oracle gap down to the k / k+1 boundary exceeds 2 delta_t -- there the ids must be
    bit-equal -- and the two expert SETS are provably the same whenever the ONE gap at the k / k+1 boundary exceeds 2 delta_t (every
    selected expert then out-scores every unselected one in the device's logits too): there the sets must be equal, token by token
    (r03: the measured "0.93-0.98 of the tokens share the set" is thereby a consequence, not the criterion; the floor stays as a
    plausibility check of the logit error).  Also bounds the logit error itself."""
    check(case, f"router logits layer{layer}", dev_logits, logits, *logit_tol)
    _, own = O.topk_lowest_index(logits, k)
    srt = torch.sort(logits, dim=1, descending=True).values
    gaps = (srt[:, :k] - srt[:, 1:k + 1]).min(dim=1).values   # every gap down to the k / k+1 boundary
    delta = (dev_logits - logits).abs().max(dim=1).values
    safe = gaps > 2 * delta + 1e-12
    same_set = (torch.sort(dev_idx, 1).values == torch.sort(own, 1).values).all(1)
    set_safe = (srt[:, k - 1] - srt[:, k]) > 2 * delta + 1e-12
    REPORT.setdefault(case, {})[f"router.layer{layer}"] = {"tokens": int(logits.shape[0]), "safe_frac": round(float(safe.float().mean()), 4),
                                                            "same_set_frac": round(float(same_set.float().mean()), 4),
                                                            "set_safe_frac": round(float(set_safe.float().mean()), 4)}
    assert bool(same_set[set_safe].all()), f"{case}: expert SETS differ on a token whose k / k+1 gap is resolvable (layer {layer})"
    if require_safe:   # (full-depth cases: deep layers' logit error leaves few or no tokens with EVERY gap resolvable; the set criterion above stays)
        assert bool(safe.any()), REPORT[case][f"router.layer{layer}"]   # (at E = 64 the max error over 64 logits vs the min of 6 gaps: ~25 % qualify)
    assert torch.equal(dev_idx[safe], own[safe]), f"{case}: router ids differ on a token with resolvable gaps (layer {layer})"
    assert float(same_set.float().mean()) >= min_same, (case, layer, float(same_set.float().mean()))


_PINNED = {}


def oracle_device_pin(odev, *, hidden=2560, heads=20, experts=64, topk=6, inter=1664, vocab=2048, S=4096, seed=41, tol=2e-4):
    """The fp32 oracle evaluated by torch's fp32 kernels on ``odev`` == the same oracle on the host: one full-width decoder layer inside a
    1-layer LM at T = 4096, training mode (aux losses), block-wise attention, routing forced to the HOST oracle's own top-k (fp32 GEMMs in
    another summation order may flip an exact tie); logits, loss and every gradient to ``tol`` (relative L2 and max-abs / max-abs).  The
    long cases below run their oracle on the device only after this has passed in the same process (once per process and device)."""
    key = (str(odev), hidden, heads, experts, topk, inter)
    if _PINNED.get(key):
        return
    ocfg = O.LMConfig(hidden_size=hidden, num_hidden_layers=1, num_attention_heads=heads, num_key_value_heads=heads, vocab_size=vocab,
                      moe_intermediate_size=inter, moe_num_experts=experts, moe_topk=topk)
    w = lm_weights(ocfg, seed)
    ids = torch.randint(0, vocab, (1, S), generator=torch.Generator().manual_seed(seed + 1))
    results = []
    host_idx = None
    for d in ("cpu", odev):
        wf = {k: v.float().to(d).requires_grad_(True) for k, v in w.items()}
        idd = ids.to(d)
        with _OracleIds() as oi, (O.forced_routing(host_idx) if host_idx is not None else _null()), O.streamed_attention(1024):
            lg = O.lm_forward(wf["model.embed_tokens.weight"][idd], wf, ocfg, training=True)
        if host_idx is None:
            host_idx = [t.cpu() for t in oi.idx]
        loss = torch.nn.functional.cross_entropy(lg[:, :-1].reshape(-1, vocab), idd[:, 1:].reshape(-1))
        loss.backward()
        results.append((lg.detach().cpu(), float(loss.detach()), {k: v.grad.detach().cpu() for k, v in wf.items()}))
        del wf, lg, loss
    (lg0, l0, g0), (lg1, l1, g1) = results
    case = f"oracle_device_pin_{odev}"
    check(case, "logits", lg1, lg0, tol, tol)
    assert abs(l0 - l1) <= 1e-5 * abs(l0), (l0, l1)
    for k in g0:
        check(case, "grad " + k, g1[k], g0[k], tol, 10 * tol)
    _PINNED[key] = True


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class _OracleIds:
    """Records the expert ids the oracle's router returns (its own top-k unless routing is forced)."""

    def __enter__(self):
        self.idx = []
        self._orig = O.router_routing

        def rr(logits, topk, num_experts):
            r = self._orig(logits, topk, num_experts)
            self.idx.append(r[1].detach().clone())
            return r

        O.router_routing = rr
        return self

    def __exit__(self, *exc):
        O.router_routing = self._orig
        return False


def case_lm(dev, case, *, hidden, heads, experts, topk, inter, vocab, layers, B, S, expect_big_gemm, seed=5,
            act_tol=(2e-2, 4e-2), grad_tol=(3e-2, 6e-2), recompute=False, eval_pass=True, stream_block=None, oracle_device=None):
    """L-layer AriaMoELMForCausalLM at the given width: eval logits, training loss and EVERY gradient (aux losses on).
    ``recompute``: the recipe's gradient checkpointing (True: the level the model picks; "moe" / "layer": forced);
    ``stream_block``: the oracle evaluates attention block-wise (O.streamed_attention) -- needed beyond S ~ 16 K; ``oracle_device``: the
    SAME fp32 oracle code evaluated by torch's fp32 kernels on that device (``oracle_device_pin`` first) -- the 64K case is 17 minutes of
    host fp32 otherwise."""
    import contextlib

    from aria_amd.moe_lm import AriaMoELMConfig, AriaMoELMForCausalLM, load_reference_state_dict

    def oracle_ctx():
        return O.streamed_attention(stream_block) if stream_block else contextlib.nullcontext()

    ocfg = O.LMConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, num_key_value_heads=heads, vocab_size=vocab,
                      moe_intermediate_size=inter, moe_num_experts=experts, moe_topk=topk)
    w = lm_weights(ocfg, seed)
    cfg = AriaMoELMConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, vocab_size=vocab,
                          moe_intermediate_size=inter, moe_num_experts=experts, moe_topk=topk, moe_num_shared_experts=2,
                          rms_norm_eps=ocfg.rms_norm_eps, rope_theta=ocfg.rope_theta, moe_z_loss_coeff=ocfg.moe_z_loss_coeff,
                          moe_aux_loss_coeff=ocfg.moe_aux_loss_coeff, gradient_checkpointing=bool(recompute),
                          recompute_level=recompute if isinstance(recompute, str) else "auto")
    od = oracle_device or "cpu"
    if oracle_device:
        oracle_device_pin(oracle_device, hidden=hidden, heads=heads, experts=experts, topk=topk, inter=inter)
    lm = AriaMoELMForCausalLM(cfg)
    load_reference_state_dict(lm, w)
    lm = lm.to(dev)
    ids = torch.randint(0, vocab, (B, S), generator=torch.Generator().manual_seed(seed + 1))
    wf = {k: v.float().to(od).requires_grad_(True) for k, v in w.items()}
    ids_o = ids.to(od)

    # ---- eval: logits
    if eval_pass:
        lm.eval()
        with _Recorder() as rec, torch.no_grad():
            got = lm(input_ids=ids.to(dev)).logits.float().cpu()
        assert len(rec.idx) == layers
        if expect_big_gemm:  # the 256x256 kernel families (v2 / v3) are what runs at this size -- the point of the case
            assert rec.variants and min(rec.variants) >= 2, rec.variants
        if inter % 128 == 0:  # the fused fc1 + SwiGLU launch is the one under test wherever the width allows it
            assert rec.fused == layers, (rec.fused, layers)
        REPORT.setdefault(case, {})["grouped_gemm_variants"] = sorted(set(rec.variants))
        with _OracleLogits() as ol, O.forced_routing(rec.idx), oracle_ctx(), torch.no_grad():
            want = O.lm_forward(wf["model.embed_tokens.weight"][ids_o], wf, ocfg)
        for i in range(layers):
            router_parity(case, i, rec.idx[i], rec.logits[i], ol.logits[i].cpu(), topk)
        check(case, "logits", got, want, *act_tol)

    # ---- training: loss and gradients with the router's aux losses
    lm.train()
    with _Recorder() as rec:
        out = lm(input_ids=ids.to(dev), labels=ids.to(dev), return_logits=False)
        out.loss.backward()
    if recompute and len(rec.idx) != layers:  # ARIA_RECOMPUTE_LEVEL=layer: every layer's forward ran twice (the second time inside
        assert len(rec.idx) == 2 * layers, len(rec.idx)   # backward, last layer first) -- the routing must repeat itself.  (The default
        for i in range(layers):                           # level keeps the routing and rebuilds only the expert-row tensors.)
            assert torch.equal(rec.idx[i], rec.idx[2 * layers - 1 - i]), f"layer {i}: the recomputed forward routed differently"
        rec.idx, rec.logits = rec.idx[:layers], rec.logits[:layers]
    assert len(rec.idx) == layers, len(rec.idx)
    if expect_big_gemm:
        assert rec.variants and min(rec.variants) >= 2, rec.variants
    if os.environ.get("ARIA_FUSE_QKV_ROPE", "1") != "0" and ocfg.head_dim == 128:   # no stand-alone inverse-RoPE pass in the backward
        assert rec.bwd_rope == layers and rec.inverse_rope == 0, (rec.bwd_rope, rec.inverse_rope)
    with _OracleLogits() as ol, O.forced_routing(rec.idx), oracle_ctx():
        lgo = O.lm_forward(wf["model.embed_tokens.weight"][ids_o], wf, ocfg, training=True)
    if not eval_pass:
        for i in range(layers):
            router_parity(case, i, rec.idx[i], rec.logits[i], ol.logits[i].detach().cpu(), topk)
    loss_o = torch.nn.functional.cross_entropy(lgo[:, :-1].reshape(-1, vocab), ids_o[:, 1:].reshape(-1))
    loss_o.backward()
    rel = abs(float(out.loss.detach()) - float(loss_o.detach())) / abs(float(loss_o.detach()))
    REPORT[case]["loss"] = {"got": float(out.loss.detach()), "want": float(loss_o.detach()), "rel": round(rel, 6)}
    REPORT[case]["oracle_device"] = str(od)
    REPORT[case]["recompute_level"] = getattr(lm.model, "last_recompute_level", None)
    assert rel <= 5e-3, REPORT[case]["loss"]
    n = 0
    for name, p in lm.named_parameters():
        assert p.grad is not None and wf[name].grad is not None, name
        check(case, "grad " + name, p.grad, wf[name].grad, *grad_tol)
        n += 1
    assert n == 3 + 12 * layers


# ------------------------------------------------------------------------------------------------------------ ViT + projector
def vit_weights(vc: O.VisionConfig, n_queries: int, out_dim: int, ff_dim: int, seed: int):
    g = torch.Generator().manual_seed(seed)
    D, I, P = vc.hidden_size, vc.intermediate_size, (vc.image_size // vc.patch_size) ** 2
    w = {}
    v = "vision_tower.vision_model."
    w[v + "embeddings.patch_embedding.weight"] = randw((D, vc.num_channels, vc.patch_size, vc.patch_size), g)
    w[v + "embeddings.patch_embedding.bias"] = randw((D,), g)
    w[v + "embeddings.position_embedding.weight"] = randw((P, D), g)
    for i in range(vc.num_hidden_layers):
        p = v + f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            w[p + f"self_attn.{n}.weight"] = randw((D, D), g)
            w[p + f"self_attn.{n}.bias"] = randw((D,), g)
        for n in ("layer_norm1", "layer_norm2"):
            w[p + n + ".weight"] = (1 + 0.1 * torch.randn(D, generator=g)).to(bf16)
            w[p + n + ".bias"] = randw((D,), g)
        w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"] = randw((I, D), g), randw((I,), g)
        w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"] = randw((D, I), g), randw((D,), g)
    m = "multi_modal_projector."
    w[m + "query"] = randw((n_queries, D), g, 1.0)
    for n in ("q_proj", "k_proj", "v_proj"):
        w[m + f"cross_attn.{n}.weight"] = randw((D, D), g)
    w[m + "cross_attn.multihead_attn.in_proj_weight"] = randw((3 * D, D), g)
    w[m + "cross_attn.multihead_attn.in_proj_bias"] = randw((3 * D,), g)
    w[m + "cross_attn.multihead_attn.out_proj.weight"], w[m + "cross_attn.multihead_attn.out_proj.bias"] = randw((D, D), g), randw((D,), g)
    w[m + "cross_attn.linear.weight"], w[m + "cross_attn.linear.bias"] = randw((D, D), g), randw((D,), g)
    for n in ("cross_attn.layer_norm", "cross_attn.ln_kv", "ln_ffn"):
        w[m + n + ".weight"] = (1 + 0.1 * torch.randn(D, generator=g)).to(bf16)
        w[m + n + ".bias"] = randw((D,), g)
    w[m + "ffn.linear_in.weight"] = randw((ff_dim, D), g)
    w[m + "ffn.linear_out.weight"] = randw((out_dim, ff_dim), g)
    return w


def _load(module, sd, prefix):
    own = module.state_dict()
    missing = [k for k in own if prefix + k not in sd]
    assert not missing, missing[:5]
    with torch.no_grad():
        for k, v in own.items():
            v.copy_(sd[prefix + k].to(v.dtype))


def case_vit_projector(dev, case, *, hidden, heads, inter, image, layers, queries, out_dim, n_images, valid_rows, seed=9,
                       tol=(2e-2, 5e-2)):
    """AriaVisionModel (frozen fast path AND module path) + AriaProjector on images whose bottom rows are padding (pixel_mask)."""
    from aria_amd.vision import AriaProjector, AriaVisionConfig, AriaVisionModel

    vc = O.VisionConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=inter, image_size=image)
    P = (image // vc.patch_size) ** 2
    p2q = {P: queries}
    w = vit_weights(vc, queries, out_dim, out_dim, seed)
    cfg = AriaVisionConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=inter, image_size=image)
    vit = AriaVisionModel(cfg)
    _load(vit, w, "vision_tower.")
    proj = AriaProjector(p2q, hidden, heads, hidden, out_dim, out_dim)
    _load(proj, w, "multi_modal_projector.")
    vit, proj = vit.to(dev).eval(), proj.to(dev).eval()
    g = torch.Generator().manual_seed(seed + 1)
    pv = torch.randn((n_images, 3, image, image), generator=g).clamp_(-1, 1).to(bf16)
    pm = torch.ones((n_images, image, image), dtype=torch.bool)
    pm[0, valid_rows:, :] = False                      # image 0: bottom rows padded (not a multiple of the patch size on purpose)
    pm[0, :, image - 3 * vc.patch_size:] = False       # ... and three patch columns on the right
    wf = {k: v.float() for k, v in w.items()}
    ocfg = O.AriaOracleConfig(vision=vc, patch_to_query=p2q, projector_heads=heads)
    with torch.no_grad():
        want_feat, want_atts = O.vit_forward(pv.float(), pm, wf, "vision_tower.", vc)
        want_proj = O.projector_forward(want_feat, want_atts, wf, "multi_modal_projector.", ocfg)
        feat, atts = vit(pv.to(dev), pm.to(dev))
        pj = proj(feat, attn_mask=atts)
    assert torch.equal(atts.cpu(), want_atts)
    valid = ~want_atts
    check(case, "vit features (valid patches, frozen fast path)", feat.cpu()[valid], want_feat[valid], *tol)
    check(case, "projector", pj, want_proj, *tol)
    feat_m, _ = vit(pv.to(dev).requires_grad_(True), pm.to(dev))     # module-by-module path (a trainable tower)
    check(case, "vit features (module path)", feat_m.detach().cpu()[valid], want_feat[valid], *tol)


# ------------------------------------------------------------------------------------------------------------ config #1 end to end
def case_aria_config1(dev, case, *, text, vision, queries, n_text, seed=13, act_tol=(2e-2, 5e-2), grad_tol=(4e-2, 8e-2)):
    """BASELINE.json config #1: one image + text through AriaForConditionalGeneration (ViT -> projector -> masked_scatter -> MoE LM ->
    shifted masked CE), logits + loss + gradients of projector and LM (ViT frozen), vs O.aria_forward (modeling_aria.py:194-335)."""
    from aria_amd.modeling_aria import AriaConfig, AriaForConditionalGeneration

    tc = O.LMConfig(**text)
    vc = O.VisionConfig(**vision)
    P = (vc.image_size // vc.patch_size) ** 2
    p2q = {P: queries}
    IMG = 9
    w = {("language_model." + k): v for k, v in lm_weights(tc, seed).items()}
    w.update(vit_weights(vc, queries, tc.hidden_size, tc.hidden_size, seed + 1))
    cfg = AriaConfig(vision_config=dict(vision), text_config=dict(text, moe_num_shared_experts=2), projector_patch_to_query_dict=p2q,
                     image_token_index=IMG)
    model = AriaForConditionalGeneration(cfg)
    _load(model.vision_tower, w, "vision_tower.")
    _load(model.multi_modal_projector, w, "multi_modal_projector.")
    _load(model.language_model, w, "language_model.")
    model = model.to(dev)
    model.freeze_vit()
    g = torch.Generator().manual_seed(seed + 2)
    S = 14 + queries + n_text
    ids = torch.randint(10, tc.vocab_size, (1, S), generator=g)
    ids[0, 7:7 + queries] = IMG
    am = torch.ones((1, S), dtype=torch.long)
    labels = ids.clone()
    labels[0, :7 + queries + 4] = -100
    pv = torch.randn((1, 3, vc.image_size, vc.image_size), generator=g).clamp_(-1, 1).to(bf16)
    pm = torch.ones((1, vc.image_size, vc.image_size), dtype=torch.bool)
    pm[0, int(0.75 * vc.image_size):, :] = False
    wf = {k: (v.float().requires_grad_(not k.startswith("vision_tower."))) for k, v in w.items()}
    ocfg = O.AriaOracleConfig(text=tc, vision=vc, patch_to_query=p2q, projector_heads=vc.num_attention_heads, image_token_index=IMG)
    model.eval()
    with _Recorder() as rec, torch.no_grad():
        out = model(input_ids=ids.to(dev), pixel_values=pv.to(dev), pixel_mask=pm.to(dev), attention_mask=am.to(dev), labels=labels.to(dev),
                    return_logits=True)
    with _OracleLogits() as ol, O.forced_routing(rec.idx), torch.no_grad():
        want_logits, want_loss = O.aria_forward(ids, pv.float(), pm, am, labels, wf, ocfg)
    for i in range(tc.num_hidden_layers):
        router_parity(case, i, rec.idx[i], rec.logits[i], ol.logits[i], tc.moe_topk)
    check(case, "logits", out.logits, want_logits, *act_tol)
    assert abs(float(out.loss) - float(want_loss)) <= 5e-3 * abs(float(want_loss)), (float(out.loss), float(want_loss))
    model.train()
    with _Recorder() as rec:
        out = model(input_ids=ids.to(dev), pixel_values=pv.to(dev), pixel_mask=pm.to(dev), attention_mask=am.to(dev), labels=labels.to(dev))
        out.loss.backward()
    with O.forced_routing(rec.idx):
        _, lo = O.aria_forward(ids, pv.float(), pm, am, labels, wf, ocfg, training=True)
    lo.backward()
    assert abs(float(out.loss.detach()) - float(lo.detach())) <= 5e-3 * abs(float(lo.detach()))
    n = 0
    for name, p in model.named_parameters():
        if name.startswith("vision_tower."):
            assert p.grad is None, name
            continue
        assert p.grad is not None and wf[name].grad is not None, name
        check(case, "grad " + name, p.grad, wf[name].grad, *grad_tol)
        n += 1
    assert n >= 3 + 12 * tc.num_hidden_layers + 10


# ------------------------------------------------------------------------------------------------------------ long causal attention
def case_long_attention(dev, case, *, S, H, hd, B=1, seed=21, tol=(6e-3, 2e-2), gtol=(8e-3, 3e-2), stream_block=None):
    """Causal flash attention forward + backward at config #4's / the north_star's sequence scale (gptfast/model.py:137-149, 413-447),
    fp32 oracle head by head (eager up to S = 16 K; block-wise -- O.attention_causal_streamed, pinned on the eager form -- beyond)."""
    from aria_amd import ops

    D = H * hd
    g = torch.Generator().manual_seed(seed)
    qkv = torch.randn((B * S, 3 * D), generator=g).to(bf16)
    do = torch.randn((B * S, D), generator=g).to(bf16)
    scale = hd ** -0.5
    qd = qkv.to(dev)
    o, lse = ops.attention_fwd(qd[:, :D], qd[:, D:2 * D], qd[:, 2 * D:], B, S, H, hd, scale, True, None)
    dq, dk, dv = ops.attention_bwd(qd[:, :D], qd[:, D:2 * D], qd[:, 2 * D:], o, do.to(dev), lse, B, S, H, hd, scale, True, None)
    got = [t.float().cpu() for t in (o, dq, dk, dv)]
    want_o = torch.empty(B * S, D)
    grads = [torch.empty(B * S, D) for _ in range(3)]
    for h in range(H):   # head by head: one S x S fp32 score matrix (or one block row of it) at a time
        sl = slice(h * hd, (h + 1) * hd)
        q, k, v = (qkv[:, i * D:(i + 1) * D][:, sl].float().view(B, S, 1, hd).transpose(1, 2).clone().requires_grad_(True) for i in range(3))
        oh = O.attention_causal_streamed(q, k, v, scale, stream_block) if stream_block else O.attention_eager(q, k, v, scale, True)
        oh.backward(do[:, sl].float().view(B, S, 1, hd).transpose(1, 2))
        want_o[:, sl] = oh.detach().transpose(1, 2).reshape(B * S, hd)
        for dst, t in zip(grads, (q, k, v)):
            dst[:, sl] = t.grad.transpose(1, 2).reshape(B * S, hd)
    check(case, "o", got[0], want_o, *tol)
    for name, gt, wt in zip(("dq", "dk", "dv"), got[1:], grads):
        check(case, name, gt, wt, *gtol)


def case_vit_attention_bwd(dev, case, *, S, H, hd=72, B=2, seed=23, tol=(6e-3, 2e-2), gtol=(8e-3, 3e-2)):
    """The ViT's bidirectional attention with an arbitrary key-padding mask at the 980-px patch count (S = 4900, hd 72): forward AND
    backward (an unfrozen tower; idefics2 eager attention, aria/model/vision_encoder.py:147-152) vs the fp32 eager oracle."""
    from aria_amd import ops

    D = H * hd
    g = torch.Generator().manual_seed(seed)
    qkv = torch.randn((B * S, 3 * D), generator=g).to(bf16)
    do = torch.randn((B * S, D), generator=g).to(bf16)
    km = (torch.rand(B, S, generator=g) > 0.2).to(torch.uint8)
    km[:, 0] = 1
    km[0, S - S // 4:] = 0        # image 0: its bottom quarter is padding (contiguous run), the rest scattered
    scale = hd ** -0.5
    qd = qkv.to(dev)
    o, lse = ops.attention_fwd(qd[:, :D], qd[:, D:2 * D], qd[:, 2 * D:], B, S, H, hd, scale, False, key_mask=km.to(dev))
    dq, dk, dv = ops.attention_bwd(qd[:, :D], qd[:, D:2 * D], qd[:, 2 * D:], o, do.to(dev), lse, B, S, H, hd, scale, False, key_mask=km.to(dev))
    q, k, v = (qkv[:, i * D:(i + 1) * D].float().view(B, S, H, hd).transpose(1, 2).clone().requires_grad_(True) for i in range(3))
    want = O.attention_eager(q, k, v, scale, False, key_padding=(km == 0))
    want.backward(do.float().view(B, S, H, hd).transpose(1, 2))
    check(case, "o", o, want.detach().transpose(1, 2).reshape(B * S, D), *tol)
    for name, gt, t in zip(("dq", "dk", "dv"), (dq, dk, dv), (q, k, v)):
        check(case, name, gt, t.grad.transpose(1, 2).reshape(B * S, D), *gtol)
    # masked keys receive no gradient at all
    dead = (km == 0).reshape(-1)
    assert float(dk.float().cpu()[dead].abs().max()) == 0.0 and float(dv.float().cpu()[dead].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------------------ config #4 prefill
def case_prefill_gptfast(dev, case, *, hidden, heads, experts, topk, inter, vocab, layers, S, seed=31, tol=(3e-2, 6e-2), stream_block=4096,
                         expect_big_gemm=True, oracle_device=None):
    """BASELINE config #4's code path at its sequence length: the gptfast surface (aria_amd.gptfast.Transformer, model.pth wire format
    converted from the HF layout exactly as gptfast/scripts/convert_hf_checkpoint.py:90-162 does) prefills S tokens into its static bf16
    KV cache and returns the LAST position's logits (gptfast/model.py:178-234, 413-447); oracle = O.lm_forward (HF layout: the two
    reference implementations agree to 5e-7 in fp32) on the same bf16-rounded weights with block-wise attention, routing
    forced to the device's ids.  Also checks rows of the KV cache written at both ends of the sequence."""
    from aria_amd import gptfast as G
    from aria_amd.checkpoint import hf_to_gptfast

    ocfg = O.LMConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, num_key_value_heads=heads, vocab_size=vocab,
                      moe_intermediate_size=inter, moe_num_experts=experts, moe_topk=topk)
    w = lm_weights(ocfg, seed)
    args = G.ModelArgs(block_size=S, vocab_size=vocab, n_layer=layers, n_head=heads, dim=hidden, intermediate_size=inter,
                       rope_base=ocfg.rope_theta, norm_eps=ocfg.rms_norm_eps, num_experts=experts, router_topk=topk, num_shared_experts=2)
    tf = G.Transformer(args)
    gw = {k[len("llm."):]: v for k, v in hf_to_gptfast({"language_model." + k: v for k, v in w.items()}, heads, ocfg.head_dim).items()}
    own = tf.state_dict()
    assert set(own) == set(gw), set(own) ^ set(gw)
    with torch.no_grad():
        for k, v in own.items():
            v.copy_(gw[k].to(v.dtype))
    tf = tf.to(dev).eval()
    tf.setup_caches(1, S)
    ids = torch.randint(0, vocab, (1, S), generator=torch.Generator().manual_seed(seed + 1))
    with _Recorder() as rec, torch.no_grad():
        got = tf(ids.to(dev), torch.arange(S, device=dev), last_only=True).float().cpu()
    assert len(rec.idx) == layers and rec.variants and (not expect_big_gemm or min(rec.variants) >= 2), (len(rec.idx), rec.variants)
    od = oracle_device or "cpu"
    if oracle_device:   # the same fp32 oracle code on the device's fp32 kernels, pinned on the host oracle first (see oracle_device_pin)
        oracle_device_pin(oracle_device, hidden=hidden, heads=heads, experts=experts, topk=topk, inter=inter)
    wf = {k: v.float().to(od) for k, v in w.items()}
    with _OracleLogits() as ol, O.forced_routing(rec.idx), O.streamed_attention(stream_block), torch.no_grad():
        h = O.lm_forward(wf["model.embed_tokens.weight"][ids.to(od)], wf, ocfg, return_hidden=True)
        want = torch.nn.functional.linear(h[:, -1:], wf["lm_head.weight"]).cpu()
    wf = {k: v.cpu() for k, v in wf.items()}
    REPORT.setdefault(case, {})["oracle_device"] = str(od)
    for i in range(layers):
        router_parity(case, i, rec.idx[i], rec.logits[i], ol.logits[i].cpu(), topk)
    check(case, "last-position logits", got, want, *tol)
    # the static KV cache of layer 0 (gptfast/model.py:67-93): V rows are v_proj(RMSNorm(embedding)) -- layout-independent -- at both ends
    rows = torch.tensor([0, 1, S // 2, S - 2, S - 1])
    x0 = O.rms_norm(wf["model.embed_tokens.weight"][ids[0, rows]], wf["model.layers.0.input_layernorm.weight"], ocfg.rms_norm_eps)
    check(case, "layer-0 V cache rows", tf.layers[0].attention.kv_cache.v[0, rows.to(dev)].float().cpu(),
          torch.nn.functional.linear(x0, wf["model.layers.0.self_attn.v_proj.weight"]), 1e-2, 3e-2)
    assert int(got.argmax()) == int(want.argmax()) or float(want.flatten().topk(2).values.diff().abs()) < 4e-2 * float(want.abs().max())


# ------------------------------------------------------------------------------------------------------------ operands beyond 2^31 bytes
def case_grouped_gemm_beyond_2g(dev, case, *, rows=430080, K=2560, I=1664, E=64, seed=51, expect_v3=True):
    """The three fused grouped launches of the 64K-token step on operands whose byte offsets pass 2^31 (393 216 expert rows x 2560 x 2 B =
    2.01 GB; here 430 080 rows so that A, H, ACT and DH all cross the boundary): ``gemm3_kernel<.., .., 3>`` (fc1 + SwiGLU, moe_lm.py:505-525),
    ``<.., .., 5>`` (fc2 input gradient + SwiGLU backward) and ``<.., .., 6>`` (gptfast w1 / w3 split form, gptfast/model.py:278-325).
    Two checks per launch: (1) the rows of chosen experts -- the first, the ones straddling 2^31 bytes of every operand, the last -- are
    BIT-IDENTICAL to the same kernel run on that expert alone at low addresses (same tiles, same k order: only the addressing differs);
    (2) sampled rows against the fp32 oracle (sequential_gemm + glu, and autograd through glu) on the host."""
    from aria_amd import hip, ops

    g = torch.Generator(device=dev).manual_seed(seed)
    gc = torch.Generator().manual_seed(seed)
    # ragged counts: a few empty experts, a few tiny ones, the rest uneven; sum = rows
    wts = torch.rand(E, generator=gc) + 0.25
    wts[[3, min(17, E - 1)]] = 0.0
    counts = torch.floor(wts / wts.sum() * (rows - 7)).long()
    counts[5] += 7 + (rows - 7 - int(counts.sum()))
    assert int(counts.sum()) == rows and int(counts.min()) == 0
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(counts, 0)
    offd = off.to(dev)

    def rn(*shape, scale=1.0):
        return (torch.randn(shape, generator=g, device=dev) * scale).to(bf16)

    a = rn(rows, K)
    w1 = rn(E, K, 2 * I, scale=0.02)                       # experts.fc1.weight [E, K, 2I]
    lib = hip.get_lib().cdll

    def experts_at(byte_marks, row_bytes):
        """experts holding the rows at the given byte offsets of an operand with ``row_bytes`` per row"""
        out = set()
        for m in byte_marks:
            r = min(rows - 1, m // row_bytes)
            out.add(int(torch.searchsorted(off[1:].long(), torch.tensor(r), right=True)))
        return out

    marks = [1 << 31, (1 << 31) - 4096, (1 << 31) + 4096]
    chosen = {int((counts > 0).nonzero()[0]), int((counts > 0).nonzero()[-1])}
    for rb in (2 * K, 2 * 2 * I, 2 * I):
        chosen |= experts_at(marks, rb)
    chosen = sorted(e for e in chosen if counts[e] > 0)
    sample = torch.cat([torch.arange(0, 64), torch.arange(rows - 64, rows)] +
                       [torch.arange(max(0, min(rows, (1 << 31) // rb) - 32), min(rows, (1 << 31) // rb + 32)) for rb in (2 * K, 4 * I, 2 * I)]).unique()
    sample_e = torch.searchsorted(off[1:].long(), sample, right=True)

    def one_expert_offsets(n):
        return torch.tensor([0, n], dtype=torch.int32, device=dev)

    # ---- <3>: fc1 + SwiGLU
    h, act = ops.grouped_gemm_swiglu(a, w1, offd, want_h=True)
    assert lib.aria_last_gemm_variant() == 3 or not expect_v3
    for e in chosen:
        s0, s1 = int(off[e]), int(off[e + 1])
        h1, act1 = ops.grouped_gemm_swiglu(a[s0:s1].clone(), w1[e:e + 1].clone(), one_expert_offsets(s1 - s0), want_h=True)
        assert torch.equal(h[s0:s1], h1) and torch.equal(act[s0:s1], act1), f"<3> expert {e} rows {s0}:{s1} differ from the low-address run"
    hs = torch.stack([a[r].float().cpu() @ w1[e].float().cpu() for r, e in zip(sample.tolist(), sample_e.tolist())])
    check(case, "<3> h sampled rows", h[sample.to(dev)], hs, 1e-2, 3e-2)
    check(case, "<3> act sampled rows", act[sample.to(dev)], O.glu(hs.to(bf16).float()), 1.5e-2, 4e-2)

    # ---- <5>: fc2 input gradient + SwiGLU backward (dy [rows, K], fc2.weight [E, I, K], h from above)
    dy = rn(rows, K)
    w2 = rn(E, I, K, scale=0.02)
    dh = ops.grouped_gemm_dswiglu(dy, w2, offd, h)
    assert lib.aria_last_gemm_variant() == 3 or not expect_v3
    for e in chosen:
        s0, s1 = int(off[e]), int(off[e + 1])
        dh1 = ops.grouped_gemm_dswiglu(dy[s0:s1].clone(), w2[e:e + 1].clone(), one_expert_offsets(s1 - s0), h[s0:s1].clone())
        assert torch.equal(dh[s0:s1], dh1), f"<5> expert {e} rows {s0}:{s1} differ from the low-address run"
    dact = torch.stack([dy[r].float().cpu() @ w2[e].float().cpu().t() for r, e in zip(sample.tolist(), sample_e.tolist())]).to(bf16).float()
    hf = h[sample.to(dev)].float().cpu().requires_grad_(True)
    (O.glu(hf) * dact).sum().backward()
    check(case, "<5> dh sampled rows", dh[sample.to(dev)], hf.grad, 2e-2, 5e-2)
    del dh, dy, w2

    # ---- <6>: gate / up weights as two [E, I, K] tensors of one allocation (gptfast wire format)
    pair = torch.empty((2, E, I, K), dtype=bf16, device=dev)
    pair[0].copy_(w1[:, :, :I].transpose(1, 2))
    pair[1].copy_(w1[:, :, I:].transpose(1, 2))
    assert ops.glu_split_fusable(pair[0], pair[1])
    h6, act6 = ops.grouped_gemm_swiglu_split(a, pair[0], pair[1], offd, want_h=True)
    assert lib.aria_last_gemm_variant() == 3 or not expect_v3
    for e in chosen:
        s0, s1 = int(off[e]), int(off[e + 1])
        sub = torch.empty((2, 1, I, K), dtype=bf16, device=dev)
        sub[0, 0].copy_(pair[0, e]), sub[1, 0].copy_(pair[1, e])
        h1, act1 = ops.grouped_gemm_swiglu_split(a[s0:s1].clone(), sub[0], sub[1], one_expert_offsets(s1 - s0), want_h=True)
        assert torch.equal(h6[s0:s1], h1) and torch.equal(act6[s0:s1], act1), f"<6> expert {e} rows {s0}:{s1} differ from the low-address run"
    check(case, "<6> h sampled rows", h6[sample.to(dev)], hs, 1e-2, 3e-2)
    check(case, "<6> act sampled rows", act6[sample.to(dev)], O.glu(hs.to(bf16).float()), 1.5e-2, 4e-2)
    REPORT[case]["rows"] = rows
    REPORT[case]["experts_compared_bitwise"] = chosen
    REPORT[case]["bytes_A_H_ACT"] = [rows * 2 * K, rows * 4 * I, rows * 2 * I]


# ------------------------------------------------------------------------------------------------------------ FULL DEPTH
# Every published number is measured on 28 decoder layers + 27 ViT layers (gptfast/model.py:39-54, 539-551); the cases above stop at two.  The
# three cases below run the model at its real depth against the same fp32 oracle, evaluated by torch's fp32 kernels on the device after
# ``oracle_device_pin`` (fp32 weights 99.6 GB + the bf16 model 50 GB fit one 288 GB GPU).  What they add to the shallow cases: error growth
# through 28 residual blocks (recorded layer by layer -- the growth curve is part of the report), 28 routers choosing on activations that
# carry the rounding of everything before them, and the decode engine over a 28-layer cache.
def init_on_device(module, seed, dev):
    """N(0, 0.02) for every matrix (router / expert weights are torch.empty in the reference), 1 +- 0.1 for norm weights, drawn
    on the device (25 B values: the host generator needs minutes and 50 GB)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            flat = p.view(-1)
            norm = n.endswith("norm.weight") or n.endswith("layernorm.weight")
            for o in range(0, flat.numel(), 1 << 28):
                c = flat[o:o + (1 << 28)]
                c.copy_((torch.randn(c.shape, generator=g, device=dev) * (0.1 if norm else 0.02) + (1.0 if norm else 0.0)).to(c.dtype))


class _LayerOutputs:
    """Per-layer hidden states of the oracle (wraps O.decoder_layer), kept on the oracle's device."""

    def __enter__(self):
        self.h = []
        self._orig = O.decoder_layer

        def dl(*a, **k):
            y = self._orig(*a, **k)
            self.h.append(y.detach())
            return y

        O.decoder_layer = dl
        return self

    def __exit__(self, *exc):
        O.decoder_layer = self._orig
        return False


def _depth_tol(base, layer, layers):
    """rel-L2 bound of the hidden state after ``layer`` + 1 of ``layers`` blocks.  Measured on MI355X (profiles/r05_fullwidth_parity.json,
    28 layers, S = 2048): 1.10e-2 after block 0, 1.77e-2 after 3, 2.51e-2 after 12, 2.97e-2 after 27 -- slower than the sqrt(depth) of
    independent per-block roundings (every RMSNorm renormalises the stream, and the residual stream's norm grows with depth while a
    block's rounding error does not): (depth) ** 0.3 fits the curve to 5 %; ``base`` = 1.5 x the one-block level."""
    return base * (1.0 + layer) ** 0.3


def case_lm_full_depth(dev, case, *, hidden=2560, heads=20, experts=64, topk=6, inter=1664, vocab=100352, layers=28, S=2048,
                       grad_layers=(0, 13, 27), seed=61, oracle_device=None, block_tol=1.7e-2, grad_tol=(7e-2, 1.2e-1), expect_big_gemm=True,
                       bf16_arm=True, grad_arm_factor=1.25, grad_arm_slack=5e-3):
    """``layers``-layer AriaMoELMForCausalLM at Aria's widths, B = 1: eval logits, per-layer hidden states, training loss, and the gradients of
    the chosen layers + embedding + final norm + lm_head, vs O.lm_forward in fp32 (moe_lm.py:548-661) on the same bf16-rounded weights.
    Routing forced to the device's ids per layer (router set criterion per layer); the third arm -- the SAME oracle code run in bf16 on
    the oracle's device, i.e. what the reference's own bf16 execution does -- gives the scale the device's deviation is read against.
    r06: the third arm runs the TRAINING pass too (same routing, same loss, autograd in bf16), and every compared
    gradient must be no further from the fp32 oracle than ``grad_arm_factor`` x the bf16 arm's own distance (+ ``grad_arm_slack``, rel-L2) --
    the per-tensor tolerance is DERIVED from the reference's arithmetic; ``grad_tol`` stays as the outer sanity bound (and the only one if
    the arm could not run).  The arm's router logits give ITS top-k same-set fraction per layer, recorded beside the device's."""
    import gc

    from aria_amd.moe_lm import AriaMoELMConfig, AriaMoELMForCausalLM

    od = oracle_device or "cpu"
    if oracle_device:
        oracle_device_pin(oracle_device, hidden=hidden, heads=heads, experts=experts, topk=topk, inter=inter)
    ocfg = O.LMConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, num_key_value_heads=heads, vocab_size=vocab,
                      moe_intermediate_size=inter, moe_num_experts=experts, moe_topk=topk)
    cfg = AriaMoELMConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, vocab_size=vocab,
                          moe_intermediate_size=inter, moe_num_experts=experts, moe_topk=topk, moe_num_shared_experts=2,
                          rms_norm_eps=ocfg.rms_norm_eps, rope_theta=ocfg.rope_theta, moe_z_loss_coeff=ocfg.moe_z_loss_coeff,
                          moe_aux_loss_coeff=ocfg.moe_aux_loss_coeff)
    with torch.device(dev):
        lm = AriaMoELMForCausalLM(cfg)
    init_on_device(lm, seed, dev)
    ids = torch.randint(0, vocab, (1, S), generator=torch.Generator().manual_seed(seed + 1))
    rep = REPORT.setdefault(case, {})
    rep.update(layers=layers, tokens=S, oracle_device=str(od))

    # ---- device: eval logits + every layer's output; training loss + gradients
    lm.eval()
    dev_h = []
    hooks = [l.register_forward_hook(lambda m, a, out: dev_h.append(out.detach().float().to(od))) for l in lm.model.layers]
    with _Recorder() as rec_e, torch.no_grad():
        got_logits = lm(input_ids=ids.to(dev)).logits.float().to(od)
    for h in hooks:
        h.remove()
    assert len(rec_e.idx) == layers and len(dev_h) == layers
    if expect_big_gemm:
        assert rec_e.variants and min(rec_e.variants) >= 2, rec_e.variants
    lm.train()
    with _Recorder() as rec_t:
        out = lm(input_ids=ids.to(dev), labels=ids.to(dev), return_logits=False)
        out.loss.backward()
    assert len(rec_t.idx) == layers
    dev_loss = float(out.loss.detach())
    want_grads = ["model.embed_tokens.weight", "model.norm.weight", "lm_head.weight"]
    want_grads += [n for n, _ in lm.named_parameters() if any(n.startswith(f"model.layers.{i}.") for i in grad_layers)]
    dev_grad = {}
    for n, p in lm.named_parameters():
        assert p.grad is not None, n
        if n in want_grads:
            dev_grad[n] = p.grad.detach().float().to(od)
        p.grad = None
    del out
    # ---- the oracle's weights: fp32 copies of the SAME bf16 values
    wf = {n: p.detach().float().to(od) for n, p in lm.named_parameters()}
    ids_o = ids.to(od)

    # ---- third arm: the oracle's code in bf16 (the reference's dtype), same routing
    ref_h, ref_grad, ref_router_logits = None, {}, None
    if bf16_arm:
        try:
            wb = {n: p.detach().to(od) for n, p in lm.named_parameters()}
            with _LayerOutputs() as lo, _OracleLogits() as ol_b, O.forced_routing(rec_e.idx), torch.no_grad():
                ref_logits = O.lm_forward(wb["model.embed_tokens.weight"][ids_o], wb, ocfg).float()
            ref_h = [h.float() for h in lo.h]
            ref_router_logits = [x.float().cpu() for x in ol_b.logits]
            # the arm's TRAINING pass: the same oracle code, bf16 leaves, the device's training routing, the loss on fp32 logits as the
            # reference computes it (modeling_llama's upcast) -> bf16 gradients of the compared tensors
            ref_grad = {}
            for n in want_grads:
                wb[n] = wb[n].clone().requires_grad_(True)
            with O.forced_routing(rec_t.idx):
                lgb = O.lm_forward(wb["model.embed_tokens.weight"][ids_o], wb, ocfg, training=True)
            loss_b = torch.nn.functional.cross_entropy(lgb[:, :-1].reshape(-1, vocab).float(), ids_o[:, 1:].reshape(-1))
            loss_b.backward()
            rep["bf16_reference_loss"] = float(loss_b.detach())
            for n in want_grads:
                ref_grad[n] = wb[n].grad.detach().float()
            del wb, lgb, loss_b
        except Exception as ex:  # noqa: BLE001 -- the arm is context, not the claim: a kernel torch lacks in bf16 must not fail the case
            rep["bf16_reference_arm_error"] = f"{type(ex).__name__}: {ex}"[:200]
            ref_h, ref_grad, ref_router_logits = None, {}, None
    del lm
    gc.collect()
    if str(dev).startswith("cuda"):
        torch.cuda.empty_cache()

    # ---- oracle, eval: logits + every layer's output on the device's routing
    with _LayerOutputs() as lo, _OracleLogits() as ol, O.forced_routing(rec_e.idx), torch.no_grad():
        want_logits = O.lm_forward(wf["model.embed_tokens.weight"][ids_o], wf, ocfg)
    curve = []
    for i in range(layers):
        m = metrics(dev_h[i], lo.h[i])
        row = {"layer": i, "device_rel_l2": round(m["rel_l2"], 6), "device_max_rel": round(m["max_rel"], 6)}
        if ref_h is not None:
            row["bf16_reference_rel_l2"] = round(metrics(ref_h[i], lo.h[i])["rel_l2"], 6)
        curve.append(row)
    rep["hidden_state_growth"] = curve
    if ref_h is not None:
        rep["bf16_reference_logits"] = {k: round(v, 6) for k, v in metrics(ref_logits, want_logits).items()}
    for i in range(layers):
        grow = _depth_tol(1.0, i, layers)
        router_parity(case, i, rec_e.idx[i], rec_e.logits[i], ol.logits[i].cpu(), topk, min_same=0.75, logit_tol=(2e-2 * grow, 6e-2 * grow),
                      require_safe=False)
        if ref_router_logits is not None:   # the same statistic for the reference's own bf16 arithmetic: ITS top-k on ITS logits vs the fp32 oracle's
            lo32 = ol.logits[i].cpu().float()
            _, own32 = O.topk_lowest_index(lo32, topk)
            _, own16 = O.topk_lowest_index(ref_router_logits[i], topk)
            same16 = (torch.sort(own16, 1).values == torch.sort(own32, 1).values).all(1).float().mean()
            rep[f"router.layer{i}"]["bf16_reference_same_set_frac"] = round(float(same16), 4)
            rep[f"router.layer{i}"]["bf16_reference_logits_rel_l2"] = round(metrics(ref_router_logits[i], lo32)["rel_l2"], 6)
    for i in range(layers):
        assert curve[i]["device_rel_l2"] <= _depth_tol(block_tol, i, layers), (case, "hidden state", curve[i], _depth_tol(block_tol, i, layers))
    check(case, "logits", got_logits, want_logits, _depth_tol(block_tol, layers - 1, layers), 2 * _depth_tol(block_tol, layers - 1, layers))
    if ref_h is not None:   # the device is no further from fp32 than 1.25 x what the reference's own bf16 arithmetic is (measured: 0.90 x)
        assert rep["logits"]["rel_l2"] <= 1.25 * rep["bf16_reference_logits"]["rel_l2"] + 2e-3, (rep["logits"], rep["bf16_reference_logits"])
    del dev_h, ref_h, lo

    # ---- oracle, training: loss + the chosen gradients (aux losses on), on the training pass's routing
    for n in want_grads:
        wf[n].requires_grad_(True)
    with _OracleLogits() as ol, O.forced_routing(rec_t.idx):
        lgo = O.lm_forward(wf["model.embed_tokens.weight"][ids_o], wf, ocfg, training=True)
    loss_o = torch.nn.functional.cross_entropy(lgo[:, :-1].reshape(-1, vocab), ids_o[:, 1:].reshape(-1))
    loss_o.backward()
    rel = abs(dev_loss - float(loss_o.detach())) / abs(float(loss_o.detach()))
    rep["loss"] = {"got": dev_loss, "want": float(loss_o.detach()), "rel": round(rel, 6)}
    assert rel <= 5e-3, rep["loss"]
    worst = None
    for n in want_grads:
        assert wf[n].grad is not None, n
        check(case, "grad " + n, dev_grad[n], wf[n].grad, *grad_tol)
        if n in ref_grad:   # derived bound: the device's gradient vs what the reference's own bf16 arithmetic gives for the same tensor
            mb = metrics(ref_grad[n], wf[n].grad)
            e = rep["grad " + n]
            e["bf16_reference_rel_l2"], e["bf16_reference_max_rel"] = round(mb["rel_l2"], 6), round(mb["max_rel"], 6)
            e["device_over_bf16_reference"] = round(e["rel_l2"] / max(mb["rel_l2"], 1e-12), 3)
            if worst is None or e["device_over_bf16_reference"] > worst[1]:
                worst = (n, e["device_over_bf16_reference"])
            assert e["rel_l2"] <= grad_arm_factor * mb["rel_l2"] + grad_arm_slack, (case, "grad vs the bf16 reference arm", n, e)
    if worst is not None:
        rep["grad_worst_device_over_bf16_reference"] = {"tensor": worst[0], "ratio": worst[1], "bound": f"{grad_arm_factor} x arm + {grad_arm_slack}"}
    rep["gradients_compared"] = len(want_grads)
    rep["gradients_bounded_by_bf16_arm"] = len(ref_grad)
    assert len(want_grads) == 3 + 12 * len(grad_layers)


def case_vit_full_depth(dev, case, *, hidden=1152, heads=16, inter=4304, image=980, layers=27, queries=256, out_dim=2560, n_images=2,
                        valid_rows=735, seed=63, oracle_device=None, tol=(2e-2, 5e-2), pin_tol=2e-4):
    """The 27-layer Idefics2 tower + the 256-query projector on 980-px images (image 0 padded in rows and columns): valid-patch features of
    the frozen fast path and of the module path, and the projector output, vs O.vit_forward / O.projector_forward in fp32
    (vision_encoder.py:94-152, projector.py:160-189).  With ``oracle_device`` the oracle runs on that device AFTER a one-layer, one-image
    instance of the same functions has matched the host evaluation to ``pin_tol``."""
    from aria_amd.vision import AriaProjector, AriaVisionConfig, AriaVisionModel

    od = oracle_device or "cpu"
    vc = O.VisionConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=inter, image_size=image)
    P = (image // vc.patch_size) ** 2
    p2q = {P: queries}
    w = vit_weights(vc, queries, out_dim, out_dim, seed)
    cfg = AriaVisionConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=inter, image_size=image)
    vit = AriaVisionModel(cfg)
    _load(vit, w, "vision_tower.")
    proj = AriaProjector(p2q, hidden, heads, hidden, out_dim, out_dim)
    _load(proj, w, "multi_modal_projector.")
    vit, proj = vit.to(dev).eval(), proj.to(dev).eval()
    g = torch.Generator().manual_seed(seed + 1)
    pv = torch.randn((n_images, 3, image, image), generator=g).clamp_(-1, 1).to(bf16)
    pm = torch.ones((n_images, image, image), dtype=torch.bool)
    pm[0, valid_rows:, :] = False
    pm[0, :, image - 3 * vc.patch_size:] = False
    ocfg = O.AriaOracleConfig(vision=vc, patch_to_query=p2q, projector_heads=heads)
    wf = {k: v.float() for k, v in w.items()}
    if oracle_device:   # pin: one layer, the padded image, host vs device evaluation of the same fp32 code
        vc1 = O.VisionConfig(hidden_size=hidden, num_hidden_layers=1, num_attention_heads=heads, intermediate_size=inter, image_size=image)
        with torch.no_grad():
            f0, a0 = O.vit_forward(pv[:1].float(), pm[:1], wf, "vision_tower.", vc1)
            p0 = O.projector_forward(f0, a0, wf, "multi_modal_projector.", ocfg)
            wd = {k: v.to(od) for k, v in wf.items()}
            f1, a1 = O.vit_forward(pv[:1].float().to(od), pm[:1].to(od), wd, "vision_tower.", vc1)
            p1 = O.projector_forward(f1, a1, wd, "multi_modal_projector.", ocfg)
        assert torch.equal(a0, a1.cpu())
        check(case + "_oracle_pin", "vit layer (device fp32 vs host fp32)", f1, f0, pin_tol, 10 * pin_tol)
        check(case + "_oracle_pin", "projector (device fp32 vs host fp32)", p1, p0, pin_tol, 10 * pin_tol)
        wf = wd
    with torch.no_grad():
        want_feat, want_atts = O.vit_forward(pv.float().to(od), pm.to(od), wf, "vision_tower.", vc)
        want_proj = O.projector_forward(want_feat, want_atts, wf, "multi_modal_projector.", ocfg)
        feat, atts = vit(pv.to(dev), pm.to(dev))
        pj = proj(feat, attn_mask=atts)
    want_feat, want_atts, want_proj = want_feat.cpu(), want_atts.cpu(), want_proj.cpu()
    assert torch.equal(atts.cpu(), want_atts)
    valid = ~want_atts
    REPORT.setdefault(case, {}).update(layers=layers, images=n_images, patches=P, oracle_device=str(od))
    check(case, "vit features (valid patches, frozen fast path)", feat.cpu()[valid], want_feat[valid], *tol)
    check(case, "projector", pj, want_proj, *tol)
    feat_m, _ = vit(pv.to(dev).requires_grad_(True), pm.to(dev))     # module-by-module path (a trainable tower)
    check(case, "vit features (module path)", feat_m.detach().cpu()[valid], want_feat[valid], *tol)


def case_decode_full_depth(dev, case, *, hidden=2560, heads=20, experts=64, topk=6, inter=1664, vocab=100352, layers=28, prompt=280,
                           new_tokens=16, seed=65, oracle_device=None, block_tol=1.7e-2, expect_engine=True):
    """BASELINE config #2's path at full depth: the gptfast surface prefills ``prompt`` positions into its static bf16 KV cache, then the
    decode engine (aria_decode_token) produces ``new_tokens`` tokens greedily, one call each (gptfast/generate.py:71-110, model.py:178-234,
    318-325, 413-447).  After every step the engine's logits are compared with the fp32 oracle run over the WHOLE sequence so far
    (O.lm_forward, HF layout; the two reference implementations agree to 5e-7 in fp32) on the same bf16-rounded weights, routing
    forced to what the device chose: the prefill's ids from ops.moe_route, each decoded token's from the engine's per-layer routing record
    (aria_decode_trace_layout).  The token stream is checked too: wherever the oracle's top-1 / top-2 margin exceeds twice the measured
    logit error, its arg-max must be the token the engine chose."""
    from aria_amd import gptfast as G
    from aria_amd.checkpoint import hf_to_gptfast

    od = oracle_device or "cpu"
    if oracle_device:
        oracle_device_pin(oracle_device, hidden=hidden, heads=heads, experts=experts, topk=topk, inter=inter)
    ocfg = O.LMConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, num_key_value_heads=heads, vocab_size=vocab,
                      moe_intermediate_size=inter, moe_num_experts=experts, moe_topk=topk)
    args = G.ModelArgs(block_size=prompt + new_tokens + 8, vocab_size=vocab, n_layer=layers, n_head=heads, dim=hidden, intermediate_size=inter,
                       rope_base=ocfg.rope_theta, norm_eps=ocfg.rms_norm_eps, num_experts=experts, router_topk=topk, num_shared_experts=2)
    with torch.device(dev):
        tf = G.Transformer(args)
    own = dict(tf.state_dict())
    # weights: drawn on the device in the HF layout layer by layer, kept as fp32 for the oracle, converted exactly as
    # gptfast/scripts/convert_hf_checkpoint.py:90-162 does, copied into the model.pth-layout module
    g = torch.Generator(device=dev).manual_seed(seed)
    D, E, I, I2 = hidden, experts, inter, 2 * inter
    Dq = heads * ocfg.head_dim

    def rn(*shape, std=0.02, mean=0.0):
        return (torch.randn(shape, generator=g, device=dev) * std + mean).to(bf16)

    wf, seen = {}, set()

    def put(hf: dict):
        for k, v in hf.items():
            wf[k[len("language_model."):]] = v.float().to(od)
        for k, v in hf_to_gptfast(hf, heads, ocfg.head_dim).items():
            name = k[len("llm."):]
            with torch.no_grad():
                own[name].copy_(v.to(own[name].dtype))
            seen.add(name)

    put({"language_model.model.embed_tokens.weight": rn(vocab, D), "language_model.lm_head.weight": rn(vocab, D),
         "language_model.model.norm.weight": rn(D, std=0.1, mean=1.0)})
    for i in range(layers):
        p = f"language_model.model.layers.{i}."
        put({p + "input_layernorm.weight": rn(D, std=0.1, mean=1.0), p + "post_attention_layernorm.weight": rn(D, std=0.1, mean=1.0),
             p + "self_attn.q_proj.weight": rn(Dq, D), p + "self_attn.k_proj.weight": rn(Dq, D), p + "self_attn.v_proj.weight": rn(Dq, D),
             p + "self_attn.o_proj.weight": rn(D, Dq), p + "mlp.router.weight": rn(E, D), p + "mlp.experts.fc1.weight": rn(E, D, 2 * I),
             p + "mlp.experts.fc2.weight": rn(E, I, D), p + "mlp.shared_experts.gate_proj.weight": rn(I2, D),
             p + "mlp.shared_experts.up_proj.weight": rn(I2, D), p + "mlp.shared_experts.down_proj.weight": rn(D, I2)})
    assert seen == set(own), set(own) ^ seen
    tf = tf.eval()
    tf.use_decode_engine = True
    tf.setup_caches(1, prompt + new_tokens + 8)
    ids = torch.randint(0, vocab, (1, prompt), generator=torch.Generator().manual_seed(seed + 1))
    rep = REPORT.setdefault(case, {})
    rep.update(layers=layers, prompt=prompt, new_tokens=new_tokens, oracle_device=str(od))

    # ---- device: prefill, then greedy engine steps; logits and routing records of every step
    with _Recorder() as rec, torch.no_grad():
        lg = tf(ids.to(dev), torch.arange(prompt, device=dev), last_only=True).float().reshape(-1)
    assert len(rec.idx) == layers
    step_logits, step_idx, step_rl, toks = [lg.to(od)], [], [], []
    with torch.no_grad():
        for t in range(new_tokens):
            tok = int(step_logits[-1].argmax())
            toks.append(tok)
            lg = tf(torch.tensor([[tok]], device=dev), torch.tensor([prompt + t], dtype=torch.int32, device=dev)).float().reshape(-1)
            if expect_engine:
                assert tf._engine is not None, "the single-token step did not take the decode engine"
            rl, idx, _ = tf._engine.routing_trace()
            step_logits.append(lg.clone().to(od))
            step_idx.append(idx.long().cpu())          # [L, k]
            step_rl.append(rl.float().cpu())           # [L, E]
    # ---- oracle: ONE pass over prompt + every generated token (causal: the logits at position prompt - 1 + t are those of the run over the
    # first prompt + t tokens -- "the oracle with a growing sequence" -- and predict token t)
    seq = torch.cat([ids[0], torch.tensor(toks)]).to(od)
    forced = [torch.cat([rec.idx[l]] + [step_idx[s][l][None] for s in range(new_tokens)]) for l in range(layers)]
    with _OracleLogits() as ol, O.forced_routing(forced), torch.no_grad():
        h = O.lm_forward(wf["model.embed_tokens.weight"][seq[None]], wf, ocfg, return_hidden=True)
        want_all = torch.nn.functional.linear(h[0, prompt - 1:], wf["lm_head.weight"])      # [new_tokens + 1, V]
    for l in range(layers):
        grow = _depth_tol(1.0, l, layers)
        router_parity(case, l, rec.idx[l], rec.logits[l], ol.logits[l][:prompt].cpu(), topk, min_same=0.75,
                      logit_tol=(2e-2 * grow, 6e-2 * grow), require_safe=False)
    dec_logits_o = [ol.logits[l][prompt:].cpu() for l in range(layers)]
    errs, margins_ok = [], 0
    for t in range(new_tokens + 1):
        want = want_all[t]
        m = metrics(step_logits[t], want)
        errs.append(m)
        rep[f"step{t:02d} logits"] = {k: round(v, 6) for k, v in m.items()}
        top2 = want.topk(2).values
        delta = float((step_logits[t] - want).abs().max())
        if float(top2[0] - top2[1]) > 2 * delta and t < new_tokens:
            margins_ok += 1
            assert int(want.argmax()) == toks[t], f"{case}: step {t}: the oracle's arg-max differs from the engine's token on a resolvable margin"
    # decode-step routers: the engine's record against the oracle's logits of the same positions, the 16 steps of a layer together
    for l in range(layers):
        dl = torch.stack([step_rl[s][l] for s in range(new_tokens)])
        di = torch.stack([step_idx[s][l] for s in range(new_tokens)])
        grow = _depth_tol(1.0, l, layers)
        router_parity(case + " (decode steps)", l, di, dl, dec_logits_o[l], topk, min_same=0.5,
                      logit_tol=(2e-2 * grow, 8e-2 * grow), require_safe=False)
    rep["tokens"] = toks
    rep["tokens_checked_on_resolvable_margin"] = margins_ok
    tol = _depth_tol(block_tol, layers - 1, layers)
    for t, m in enumerate(errs):
        assert m["rel_l2"] <= tol and m["max_rel"] <= 2 * tol and m["cos"] >= 1 - 2 * tol * tol, (case, t, m, tol)


# ------------------------------------------------------------------------------------------------------------ LoRA at width
def case_lm_lora(dev, case, *, hidden, heads, experts, topk, inter, vocab, layers, B, S, r=8, alpha=32, seed=71, oracle_device=None,
                 grad_tol=(4e-2, 1e-1), expect_fused=True):
    """recipes/config_lora.yaml's adapter set (fc1, fc2, q/k/v/o_proj, gate/up/down_proj, lm_head; r = 8, alpha = 32) on an L-layer LM at the
    given width, dropout off (the dropout path is pinned mask-for-mask at site level: model_cases.case_lora_fused_sites): training loss and
    the gradient of EVERY LoRA factor of the fused node (adapters inside the base launches, aria_amd.lora_functional) against the fp32
    oracle on merged weights W + scaling * delta(A, B) with autograd through A and B (aria/lora/layers.py:129-139, 196-224), routing forced
    to the device's ids.  Base weights must stay without gradient."""
    from aria_amd import autograd as AG
    from aria_amd.lora import apply_lora_from_config
    from aria_amd.moe_lm import AriaMoELMConfig, AriaMoELMForCausalLM, load_reference_state_dict

    od = oracle_device or "cpu"
    if oracle_device:
        oracle_device_pin(oracle_device, hidden=hidden, heads=heads, experts=experts, topk=topk, inter=inter)
    ocfg = O.LMConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, num_key_value_heads=heads, vocab_size=vocab,
                      moe_intermediate_size=inter, moe_num_experts=experts, moe_topk=topk)
    w = lm_weights(ocfg, seed)
    cfg = AriaMoELMConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, vocab_size=vocab,
                          moe_intermediate_size=inter, moe_num_experts=experts, moe_topk=topk, moe_num_shared_experts=2,
                          rms_norm_eps=ocfg.rms_norm_eps, rope_theta=ocfg.rope_theta, moe_z_loss_coeff=ocfg.moe_z_loss_coeff,
                          moe_aux_loss_coeff=ocfg.moe_aux_loss_coeff)
    lm = AriaMoELMForCausalLM(cfg)
    load_reference_state_dict(lm, w)
    skipped = apply_lora_from_config(lm, dict(use_peft=True, lora_r=r, lora_alpha=alpha, lora_dropout=0.0,
                                              lora_target_modules=["fc1", "fc2", "q_proj", "k_proj", "v_proj", "linear", "o_proj", "up_proj", "down_proj",
                                                                   "out_proj", "gate_proj", "lm_head"]))
    assert skipped == []
    g = torch.Generator().manual_seed(seed + 3)
    with torch.no_grad():
        for n, p in lm.named_parameters():
            if "lora_B" in n:   # (peft initialises B = 0: give the adapters something to do)
                p.copy_((torch.randn(p.shape, generator=g) * 0.02).to(bf16))
    lm = lm.to(dev).train()
    ids = torch.randint(0, vocab, (B, S), generator=torch.Generator().manual_seed(seed + 1))
    calls = []
    orig = AG.LoraDecoderLayerFn.apply
    AG.LoraDecoderLayerFn.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        with _Recorder() as rec:
            out = lm(input_ids=ids.to(dev), labels=ids.to(dev))
            out.loss.backward()
    finally:
        AG.LoraDecoderLayerFn.apply = orig
    assert len(calls) == layers and len(rec.idx) == layers, (len(calls), len(rec.idx))
    if expect_fused:
        assert rec.hip.get_lib().cdll.aria_last_gemm_variant() >= 1
    sd = {k: v.detach().float().to(od) for k, v in lm.state_dict().items()}
    leaves, wm = {}, {}
    scaling = alpha / r
    for key, val in sd.items():
        if ".lora_A." in key or ".lora_B." in key:
            leaves[key] = val.clone().requires_grad_(True)
    for key, val in sd.items():
        if key.endswith(".base_layer.weight"):
            stem = key[: -len(".base_layer.weight")]
            la, lb = leaves[stem + ".lora_A.weight"], leaves[stem + ".lora_B.weight"]
            wm[stem + ".weight"] = val + (O.lora_delta_weight(la, lb, scaling) if val.dim() == 3 else O.lora_linear_delta_weight(la, lb, scaling))
        elif ".lora_" not in key:
            wm[key] = val
    ids_o = ids.to(od)
    with _OracleLogits() as ol, O.forced_routing(rec.idx):
        lg = O.lm_forward(wm["model.embed_tokens.weight"][ids_o], wm, ocfg, training=True)
    for i in range(layers):
        router_parity(case, i, rec.idx[i], rec.logits[i], ol.logits[i].detach().cpu(), topk)
    loss_o = torch.nn.functional.cross_entropy(lg[:, :-1].reshape(-1, vocab), ids_o[:, 1:].reshape(-1))
    loss_o.backward()
    rel = abs(float(out.loss.detach()) - float(loss_o.detach())) / abs(float(loss_o.detach()))
    REPORT.setdefault(case, {})["loss"] = {"got": float(out.loss.detach()), "want": float(loss_o.detach()), "rel": round(rel, 6)}
    assert rel <= 5e-3, REPORT[case]["loss"]
    n = 0
    for name, p in lm.named_parameters():
        if ".lora_" not in name:
            assert p.grad is None, name
            continue
        check(case, "grad " + name, p.grad, leaves[name].grad, *grad_tol)
        n += 1
    assert n == 2 * (9 * layers + 1)
    REPORT[case]["adapters"] = n // 2
