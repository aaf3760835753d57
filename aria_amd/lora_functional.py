"""The decoder layer with LoRA adapters on its GEMMs (recipes/config_lora.yaml:44-59: fc1, fc2, q/k/v/o_proj, gate/up/down_proj) as ONE
hand-written forward / backward -- SURVEY 8(f)3 "fuse as y += s (x A) B".

Reference semantics, per adapted module (aria/lora/layers.py:129-139 for the grouped expert GEMMs; peft's ``Linear.forward`` for the
nn.Linear targets -- the same line):

    result = base(x) + lora_B(lora_A(dropout(x))) * scaling              base weights frozen, A / B trainable

What runs here, per module:
  * ``u = scaling * lora_A(dropout(x))`` -- a skinny GEMM (N = r) on the dropped input (``ops.dropout``: one pass, one mask byte per 8
    elements kept for the backward);
  * the base launch with the adapter's second projection as a K-EXTENSION (``ops.*_lora``: one extra K-tile fed from u and lora_B): no
    output-sized ``base + delta`` pass, and the fused epilogues keep working -- fc1 + SwiGLU, gate|up + SwiGLU, the q|k|v and gate|up wide
    launches (their adapters enter side by side with a block-diagonal B);
  * backward: the base input gradients exactly as the un-adapted layer computes them (weight gradients of the frozen base are skipped);
    per adapter d_u = scaling * dy lora_B^T (skinny), d lora_B = u^T dy, d lora_A = dropout(x)^T d_u, and the adapter's input gradient
    ``mask * (d_u lora_A) / (1 - p)`` added into dx (``ops.dropout_bwd_``; without dropout: an accumulating GEMM).
Everything else of the layer (RMSNorm with the residual folded in, RoPE, flash attention, router, dispatch, un-permute + shared add) is the
un-adapted layer's code path (aria_amd.functional).  The module-by-module form (aria_amd.lora.GroupedGemmLoraLayer / LinearLoraLayer called
as modules) stays as the reference arrangement the tests compare against.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import functional as Fn
from . import ops

bf16 = torch.bfloat16
SITES = ("wq", "wk", "wv", "wo", "fc1", "fc2", "gate", "up", "down")   # functional.py's parameter keys that may carry an adapter


@dataclass
class LoraSite:
    """One adapter: ``a`` = lora_A.weight, ``b`` = lora_B.weight in the adapter module's own layout -- nn.Linear targets: a [r, in], b [out, r];
    grouped expert GEMMs: a [E, in, r], b [E, r, out]."""
    a: torch.Tensor
    b: torch.Tensor
    scaling: float
    p: float = 0.0


def _scaled(u: torch.Tensor, s: float) -> torch.Tensor:
    return u if s == 1.0 else (u.float() * s).to(bf16)   # (u is [rows, r]: a few hundred KB)


class _Drop:
    """dropout(x) for one adapter: the dropped tensor and the mask (None, None = identity: eval mode or p = 0)."""

    def __init__(self, x: torch.Tensor, p: float, training: bool, seed: int):
        self.p = p if training else 0.0
        if self.p > 0.0:
            self.xd, self.mask = ops.dropout(x, self.p, seed)
        else:
            self.xd, self.mask = x, None


def _seed(gen_state: List[int]) -> int:
    gen_state[0] += 1
    return gen_state[0] * 0x9E3779B97F4A7C15 & ((1 << 63) - 1)


# ------------------------------------------------------------------------------------------------------------------ dense (nn.Linear) sites
def dense_lora_fwd(x: torch.Tensor, w: torch.Tensor, sites: List[Optional[LoraSite]], rows: List[int], training: bool, seeds: List[int],
                   glu: bool = False, want_h: bool = True):
    """y = x w^T + sum_j [rows of block j] scaling_j lora_B_j lora_A_j dropout_j(x)  for a weight w = row-wise concatenation of blocks with
    ``rows[j]`` rows each (q | k | v, gate | up, or one block), block j adapted by sites[j] (None: not adapted).  glu: the launch carries the
    SwiGLU epilogue -> (h or None, act).  -> (y | (h, act), ctx)"""
    live = [(j, s) for j, s in enumerate(sites) if s is not None]
    if not live:
        out = ops.gemm_swiglu(x, w, want_h=want_h) if glu else ops.gemm(x, w)
        return out, None
    T = x.shape[0]
    R = sum(s.a.shape[0] for _, s in live)
    U = torch.empty((T, R), dtype=bf16, device=x.device)
    Bx = torch.zeros((w.shape[0], R), dtype=bf16, device=x.device)   # block-diagonal: adapter j only touches its own output rows
    drops, c0 = [], 0
    starts = [sum(rows[:j]) for j in range(len(rows))]
    for j, s in live:
        r = s.a.shape[0]
        d = _Drop(x, s.p, training, _seed(seeds))
        U[:, c0:c0 + r] = _scaled(ops.gemm(d.xd, s.a), s.scaling)
        Bx[starts[j]:starts[j] + rows[j], c0:c0 + r] = s.b
        drops.append(d)
        c0 += r
    out = ops.gemm_swiglu_lora(x, w, U, Bx, want_h=want_h) if glu else ops.gemm_lora(x, w, U, Bx)
    return out, dict(live=live, drops=drops, U=U, Bx=Bx, starts=starts, rows=rows)


def dense_lora_bwd(dy: torch.Tensor, ctx, dx: torch.Tensor, grads: dict, keys: List[str]) -> None:
    """Adds the adapters' input gradients into dx (which already holds the base input gradient) and writes d lora_A / d lora_B into
    grads[key] = (dA, dB)."""
    if ctx is None:
        return
    dU = ops.gemm(dy, ctx["Bx"], b_oc=True)                       # [T, R] = dy Bx  (block-diagonal: column block j = dy_j lora_B_j)
    dBx = ops.gemm(dy, ctx["U"], a_oc=True, b_oc=True)            # [N, R] = dy^T U (scaling is inside U)
    c0 = 0
    for (j, s), d in zip(ctx["live"], ctx["drops"]):
        r = s.a.shape[0]
        du = _scaled(dU[:, c0:c0 + r].contiguous(), s.scaling)
        dA = ops.gemm(du, d.xd, a_oc=True, b_oc=True)             # [r, K] = d_u^T dropout(x)
        dB = dBx[ctx["starts"][j]:ctx["starts"][j] + ctx["rows"][j], c0:c0 + r].contiguous()
        if d.mask is None:
            ops.gemm(du, s.a, b_oc=True, out=dx, accumulate=True)
        else:
            ops.dropout_bwd_(dx, ops.gemm(du, s.a, b_oc=True), d.mask, d.p)
        grads[keys[j]] = (dA, dB)
        c0 += r


# ------------------------------------------------------------------------------------------------------------------ grouped (expert) sites
def grouped_lora_fwd(x: torch.Tensor, w: torch.Tensor, offsets: torch.Tensor, site: Optional[LoraSite], training: bool, seeds: List[int],
                     glu: bool = False, want_h: bool = True):
    if site is None:
        out = ops.grouped_gemm_swiglu(x, w, offsets, want_h=want_h) if glu else ops.grouped_gemm(x, w, offsets)
        return out, None
    d = _Drop(x, site.p, training, _seed(seeds))
    U = _scaled(ops.grouped_gemm(d.xd, site.a, offsets), site.scaling)                      # [M, r]
    out = ops.grouped_gemm_swiglu_lora(x, w, offsets, U, site.b, want_h=want_h) if glu else ops.grouped_gemm_lora(x, w, offsets, U, site.b)
    return out, dict(site=site, drop=d, U=U)


def grouped_lora_bwd(dy: torch.Tensor, ctx, offsets: torch.Tensor, dx: torch.Tensor, grads: dict, key: str) -> None:
    if ctx is None:
        return
    s, d = ctx["site"], ctx["drop"]
    E = s.a.shape[0]
    du = _scaled(ops.grouped_gemm(dy, s.b, offsets, w_is_kn=False), s.scaling)              # [M, r] = dy lora_B[e]^T
    dB = ops.grouped_gemm_wgrad(ctx["U"], dy, offsets, E)                                   # [E, r, N]
    dA = ops.grouped_gemm_wgrad(d.xd, du, offsets, E)                                       # [E, K, r]
    term = ops.grouped_gemm(du, s.a, offsets, w_is_kn=False)                                # [M, K] = d_u lora_A[e]^T
    if d.mask is None:
        dx.add_(term)
    else:
        ops.dropout_bwd_(dx, term, d.mask, d.p)
    grads[key] = (dA, dB)


# ------------------------------------------------------------------------------------------------------------------ the layer
def decoder_layer_lora_fwd(x, p: dict, L: Dict[str, LoraSite], cos, sin, B: int, S: int, acfg: Fn.AttnConfig, mcfg: Fn.MoEConfig, eps: float,
                           kv_len=None, training: bool = True, seed: int = 0):
    """functional.decoder_layer_fwd with adapters L (key -> LoraSite, any subset of SITES).  -> (out, ctx)"""
    H, hd = acfg.num_heads, acfg.head_dim
    if acfg.num_kv_heads != H or Fn._pad_hd(hd) != hd:
        raise NotImplementedError("adapted layer: MHA with a native head dim (Aria: 20 x 128)")
    seeds = [int(seed)]
    k = mcfg.topk
    T, Dq = x.shape[0], H * hd
    xn, _, rstd1 = ops.rmsnorm(x, p["ln1"], eps, want_rstd=True)
    # ---- attention block (modeling_llama.py:243-281): q | k | v as one wide launch, their adapters side by side
    wqkv = Fn.fused_weight(p["wq"], p["wk"], p["wv"])
    qkv, c_qkv = dense_lora_fwd(xn, wqkv, [L.get("wq"), L.get("wk"), L.get("wv")], [Dq, Dq, Dq], training, seeds)
    ops.rope_(qkv[:, :2 * Dq], cos, sin, S, 2 * H, hd)
    o, actx = Fn.sdpa_fwd(qkv[:, :Dq], qkv[:, Dq:2 * Dq], qkv[:, 2 * Dq:], B, S, H, hd, hd ** -0.5, acfg.causal, kv_len)
    o = o if o.is_contiguous() else o.contiguous()
    a, c_o = dense_lora_fwd(o, p["wo"], [L.get("wo")], [p["wo"].shape[0]], training, seeds)
    hn, h, rstd2 = ops.rmsnorm(a, p["ln2"], eps, residual=x, want_rstd=True)
    # ---- MoE block (moe_lm.py:548-577)
    if ops.router_fusable(hn.shape[1], p["router"].shape[0], k):
        logits, scores, idx, counts = ops.moe_router_fused(hn, p["router"], k)
    else:
        logits = ops.gemm(hn, p["router"])
        scores, idx, counts = ops.moe_route(logits, k)
    offsets, sorted_src, inv = ops.moe_sort(idx, counts)
    perm = ops.moe_permute(hn, sorted_src, k)
    fc1, fc2 = p["fc1"], p["fc2"]
    if ops.glu_fusable(fc1.shape[1], fc1.shape[2]):
        (h1, act), c_fc1 = grouped_lora_fwd(perm, fc1, offsets, L.get("fc1"), training, seeds, glu=True)
    else:
        h1, c_fc1 = grouped_lora_fwd(perm, fc1, offsets, L.get("fc1"), training, seeds)
        act = ops.swiglu(h1)
    eo, c_fc2 = grouped_lora_fwd(act, fc2, offsets, L.get("fc2"), training, seeds)
    I2 = p["gate"].shape[0]
    wgu = Fn.fused_weight(p["gate"], p["up"])
    if ops.glu_fusable(hn.shape[1], 2 * I2):
        (gu, sact), c_gu = dense_lora_fwd(hn, wgu, [L.get("gate"), L.get("up")], [I2, I2], training, seeds, glu=True)
    else:
        gu, c_gu = dense_lora_fwd(hn, wgu, [L.get("gate"), L.get("up")], [I2, I2], training, seeds)
        sact = ops.swiglu(gu)
    sh, c_down = dense_lora_fwd(sact, p["down"], [L.get("down")], [p["down"].shape[0]], training, seeds)
    out = ops.moe_unpermute(eo, inv, scores, k, add=sh, residual=h)   # h + MoE(hn): the residual add is the un-permute launch's last step
    ctx = dict(x=x, h=h, hn=hn, rstd1=rstd1, rstd2=rstd2, xn=xn, o=o, actx=actx, wqkv=wqkv, wgu=wgu, B=B, S=S, acfg=acfg, mcfg=mcfg, kv_len=kv_len,
               logits=logits, scores=scores, idx=idx, counts=counts, offsets=offsets, inv=inv, perm=perm, h1=h1, act=act, eo=eo, gu=gu, sact=sact,
               c_qkv=c_qkv, c_o=c_o, c_fc1=c_fc1, c_fc2=c_fc2, c_gu=c_gu, c_down=c_down)
    return out, ctx


def decoder_layer_lora_bwd(dout, c, p: dict, cos, sin, need_base=None) -> Tuple[torch.Tensor, dict, dict]:
    """-> (dx, base grads {key: tensor or None}, adapter grads {key: (d lora_A, d lora_B)}).  ``need_base``: set of base-parameter keys whose
    gradient is wanted (None / empty: the recipe's case -- every base weight frozen)."""
    need = set(need_base or ())
    acfg, mcfg = c["acfg"], c["mcfg"]
    H, hd, k, E = acfg.num_heads, acfg.head_dim, mcfg.topk, mcfg.num_experts
    B, S = c["B"], c["S"]
    T, Dq = c["x"].shape[0], H * hd
    g: dict = {}
    gb = {key: None for key in ("ln1", "ln2", "wq", "wk", "wv", "wo", "router", "fc1", "fc2", "gate", "up", "down")}
    hn, offsets, inv = c["hn"], c["offsets"], c["inv"]
    I2 = p["gate"].shape[0]
    # ---- MoE block: routed experts
    d_eo, dscores = ops.moe_unpermute_bwd(dout, c["eo"], inv, c["scores"], k)
    if c["c_fc2"] is None and ops.dglu_fusable(p["fc2"].shape[1], p["fc2"].shape[2]):
        d_h1 = ops.grouped_gemm_dswiglu(d_eo, p["fc2"], offsets, c["h1"])
    else:   # the adapter's input gradient belongs to d_act, in front of the SwiGLU backward
        d_act = ops.grouped_gemm(d_eo, p["fc2"], offsets, w_is_kn=False)
        grouped_lora_bwd(d_eo, c["c_fc2"], offsets, d_act, g, "fc2")
        d_h1 = ops.swiglu_bwd(c["h1"], d_act)
    if "fc2" in need:
        gb["fc2"] = ops.grouped_gemm_wgrad(c["act"], d_eo, offsets, E)
    d_perm = ops.grouped_gemm(d_h1, p["fc1"], offsets, w_is_kn=False)
    grouped_lora_bwd(d_h1, c["c_fc1"], offsets, d_perm, g, "fc1")
    if "fc1" in need:
        gb["fc1"] = ops.grouped_gemm_wgrad(c["perm"], d_h1, offsets, E)
    dhn = ops.moe_unpermute(d_perm, inv, None, k)
    # ---- shared expert
    if c["c_down"] is None and ops.dglu_fusable(p["down"].shape[1], p["down"].shape[0]):
        d_gu = ops.gemm_dswiglu(dout, p["down"], c["gu"], b_oc=True)
    else:
        d_sact = ops.gemm(dout, p["down"], b_oc=True)
        dense_lora_bwd(dout, c["c_down"], d_sact, g, ["down"])
        d_gu = ops.swiglu_bwd(c["gu"], d_sact)
    if "down" in need:
        gb["down"] = ops.gemm(dout, c["sact"], a_oc=True, b_oc=True)
    ops.gemm(d_gu, c["wgu"], b_oc=True, out=dhn, accumulate=True)
    dense_lora_bwd(d_gu, c["c_gu"], dhn, g, ["gate", "up"])
    if "gate" in need or "up" in need:
        g_gu = ops.gemm(d_gu, hn, a_oc=True, b_oc=True)
        gb["gate"], gb["up"] = g_gu[:I2], g_gu[I2:]
    # ---- router
    dlogits = ops.moe_route_bwd(c["logits"], c["idx"], c["scores"], dscores, c["counts"], mcfg.z_loss_coeff, mcfg.aux_loss_coeff, mcfg.aux_scale)
    ops.gemm(dlogits, p["router"], b_oc=True, out=dhn, accumulate=True)
    if "router" in need:
        gb["router"] = ops.gemm(dlogits, hn, a_oc=True, b_oc=True)
    dh, gb["ln2"] = ops.rmsnorm_bwd(dhn, c["h"], p["ln2"], c["rstd2"], dres=dout)
    # ---- attention block
    o = c["o"]
    d_o = ops.gemm(dh, p["wo"], b_oc=True)
    dense_lora_bwd(dh, c["c_o"], d_o, g, ["wo"])
    if "wo" in need:
        gb["wo"] = ops.gemm(dh, o, a_oc=True, b_oc=True)
    dqkv = torch.empty((T, 3 * Dq), dtype=bf16, device=dh.device)
    fused_rope = c["actx"]["hdp"] == hd and ops.attention_bwd_rope_fusable(hd, S, cos)   # (the inverse rotation inside the attention backward's epilogues)
    Fn.sdpa_bwd(d_o, c["actx"], B, S, H, hd, hd ** -0.5, acfg.causal, c["kv_len"], dq=dqkv[:, :Dq], dk=dqkv[:, Dq:2 * Dq], dv=dqkv[:, 2 * Dq:],
                rope=(cos, sin) if fused_rope else None)
    if not fused_rope:
        ops.rope_(dqkv[:, :2 * Dq], cos, sin, S, 2 * H, hd, inverse=True)
    dxn = ops.gemm(dqkv, c["wqkv"], b_oc=True)
    dense_lora_bwd(dqkv, c["c_qkv"], dxn, g, ["wq", "wk", "wv"])
    if need & {"wq", "wk", "wv"}:
        g_qkv = ops.gemm(dqkv, c["xn"], a_oc=True, b_oc=True)
        gb["wq"], gb["wk"], gb["wv"] = g_qkv[:Dq], g_qkv[Dq:2 * Dq], g_qkv[2 * Dq:]
    dx, gb["ln1"] = ops.rmsnorm_bwd(dxn, c["x"], p["ln1"], c["rstd1"], dres=dh)
    return dx, gb, g


# ------------------------------------------------------------------------------------------------------------------ lm_head + loss
def lm_head_lora_loss_fwd_bwd(hn, lm_w, labels_shifted, site: LoraSite, training: bool, seed: int, need_hn: bool = True):
    """functional.lm_head_loss_fwd_bwd with an adapter on lm_head (recipes/config_lora.yaml:59): logits = hn W^T + scaling * (dropout(hn) A^T) B^T
    over the LABELLED rows only (the module-by-module form computes and adapts all B x S rows of the [., V] logits: 3.3 GB per elementwise
    pass at the recipe's micro-batch), shifted masked CE with the gradient written over the logits, then d_hn, d lora_A, d lora_B.
    -> (loss, d_hn | None, dA, dB)"""
    rows = torch.nonzero(labels_shifted >= 0).flatten().to(torch.int32)          # (host sync: the GEMMs' M)
    n = int(rows.numel())
    if n == 0:
        zero = torch.zeros((), dtype=torch.float32, device=hn.device)
        return zero, (torch.zeros_like(hn) if need_hn else None), torch.zeros_like(site.a), torch.zeros_like(site.b)
    pad = (-n) % 8
    if pad:
        rows = torch.cat([rows, rows[-1:].expand(pad)])
    labels_v = labels_shifted[rows.long()].contiguous()
    if pad:
        labels_v[n:] = -100
    hv = ops.moe_permute(hn, rows, 1)
    d = _Drop(hv, site.p, training, _seed([int(seed)]))
    U = _scaled(ops.gemm(d.xd, site.a), site.scaling)
    logits = ops.gemm_lora(hv, lm_w, U, site.b)
    count_in = torch.full((1,), n, dtype=torch.int32, device=hn.device)
    loss_sum, _, _ = ops.cross_entropy(logits, labels_v, grad_scale=1.0, dlogits=logits, count_in=count_in)
    loss = (loss_sum / float(n)).reshape(())
    du = _scaled(ops.gemm(logits, site.b, b_oc=True), site.scaling)              # [rows, r] = dlogits lora_B
    dB = ops.gemm(logits, U, a_oc=True, b_oc=True)                               # [V, r]
    dA = ops.gemm(du, d.xd, a_oc=True, b_oc=True)                                # [r, D]
    d_hn = None
    if need_hn:
        d_hv = ops.gemm(logits, lm_w, b_oc=True)
        if d.mask is None:
            ops.gemm(du, site.a, b_oc=True, out=d_hv, accumulate=True)
        else:
            ops.dropout_bwd_(d_hv, ops.gemm(du, site.a, b_oc=True), d.mask, d.p)
        d_hn = torch.zeros_like(hn)
        d_hn.index_copy_(0, rows[:n].long(), d_hv[:n])
    return loss, d_hn, dA, dB
