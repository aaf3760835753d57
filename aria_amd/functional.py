"""Hand-written forward/backward of the Aria decoder blocks on top of the C-ABI kernels.

Each ``*_fwd`` returns ``(output, ctx)`` and each ``*_bwd`` consumes ``ctx``; ``aria_amd.autograd`` wraps them
in ``torch.autograd.Function``s so the modules in ``aria_amd.moe_lm`` are differentiable drop-ins.  Everything is
2-D token-major ([T, features], bf16) -- batch/sequence only matter to RoPE and attention.

Reference semantics (file:line relative to /root/reference):
  MoE block       aria/model/moe_lm.py:548-577 (router :243-293, dispatcher :313-365, experts :505-525, shared :368-395)
  attention block transformers/models/llama/modeling_llama.py:243-281 (inherited through moe_lm.py:594)
  decoder layer   transformers/models/llama/modeling_llama.py:295-325 (MoEDecoderLayer moe_lm.py:580-602)
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import ops

bf16 = torch.bfloat16


# ----------------------------------------------------------------------------------------------- linear
def linear_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    return ops.gemm(x, w, bias=bias, out=out)


def linear_bwd(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, need_dx: bool = True, dx_out: Optional[torch.Tensor] = None,
               accumulate_dx: bool = False, need_dw: bool = True):
    """dx = dy @ W  ([T,N] x [N,K]),  dW = dy^T @ x ([N,K])."""
    dx = None
    if need_dx:
        dx = ops.gemm(dy, w, b_oc=True, out=dx_out, accumulate=accumulate_dx)
    dw = ops.gemm(dy, x, a_oc=True, b_oc=True) if need_dw else None
    return dx, dw


# ----------------------------------------------------------------------------------------------- MoE block
@dataclass
class MoEConfig:
    topk: int
    num_experts: int
    z_loss_coeff: float = 0.0
    aux_loss_coeff: float = 0.0
    aux_scale: float = 1.0  # MoEAuxLossAutoScaler.main_loss_backward_scale (train.py:229)


def _adjacent_rows_view(ws) -> Optional[torch.Tensor]:
    """The [sum rows, D] tensor the weights already ARE when they lie back to back in one allocation (``pack_adjacent_``), else None."""
    w0 = ws[0]
    if w0.dim() != 2 or not w0.is_contiguous():
        return None
    st, end, rows = w0.untyped_storage().data_ptr(), w0.storage_offset() + w0.numel(), w0.shape[0]
    for w in ws[1:]:
        if (w.dim() != 2 or w.shape[1] != w0.shape[1] or w.dtype != w0.dtype or w.device != w0.device or not w.is_contiguous()
                or w.untyped_storage().data_ptr() != st or w.storage_offset() != end):
            return None
        end += w.numel()
        rows += w.shape[0]
    return w0.detach().as_strided((rows, w0.shape[1]), (w0.shape[1], 1), w0.storage_offset())


def pack_adjacent_(*params: torch.Tensor) -> bool:
    """Re-home nn.Linear weights that read the same input (q|k|v, shared gate|up) into ONE allocation, in order, so that ``fused_weight`` is a
    view instead of a copy per forward (r05: the copy was 63 ``CatArrayBatchedCopy`` launches = 1.7 ms of the config #3 step).  The
    parameters keep their identity, shape and values (``p.data`` becomes a slice of the new buffer): state dicts, optimizers and DDP see
    nothing; ``nn.Module._apply`` (``.to()``, ``.bfloat16()``) gives every parameter an allocation of its own again, which the decoder layer
    answers by packing again (moe_lm.py ``_PackedWeights``).  False (nothing done) for meta / mixed-device / mixed-dtype / non-2-D weights."""
    w0 = params[0]
    if any(w.dim() != 2 or w.shape[1] != w0.shape[1] or w.dtype != w0.dtype or w.device != w0.device or w.device.type == "meta" for w in params):
        return False
    if _adjacent_rows_view(params) is not None:
        return True
    with torch.no_grad():
        buf = torch.empty((sum(w.shape[0] for w in params), w0.shape[1]), dtype=w0.dtype, device=w0.device)
        r = 0
        for w in params:
            n = w.shape[0]
            buf[r:r + n].copy_(w)
            w.data = buf[r:r + n]
            r += n
    return True


def fused_weight(*ws: torch.Tensor) -> torch.Tensor:
    """Row-wise concatenation of nn.Linear weights that read the same input (q/k/v: [3D, D]; shared gate/up: [2I, D]).  One wide GEMM
    instead of three (two) narrow ones: a [16384, 2560] output is 640 tiles = 2.5 rounds on 256 CUs, the fused [16384, 7680] one 7.5 --
    and the input gradient becomes ONE GEMM with the long reduction instead of accumulate passes over dx.  A VIEW when the weights lie back
    to back in one allocation (``pack_adjacent_``: the decoder layer's parameters do); otherwise built per forward (a 39 / 34 MB copy, ~27 us)
    and kept in the layer context for the backward -- deliberately NOT cached across calls (a cache keyed on addresses / versions can go
    stale when tensors are freed and re-allocated; adjacency is checked on the tensors themselves at every call)."""
    v = _adjacent_rows_view(ws)
    return v if v is not None else torch.cat([w.detach() for w in ws], dim=0)


def moe_fwd(x, router_w, fc1, fc2, gate_w, up_w, down_w, cfg: MoEConfig, save=True, residual=None):
    """MoELayer.forward on x [T,D] -> (out [T,D], ctx).  ``save``: True = everything the backward needs; False = nothing; "lean" = everything
    EXCEPT the four expert-row tensors (perm, h1, act, eo: 24 of the layer's 36 KB per token) -- ``moe_rematerialize`` rebuilds those in
    the backward from what is kept (selective recompute: the routed experts only, 60 % of the layer's forward GEMM flops)."""
    k = cfg.topk
    lean = save == "lean"
    if ops.router_fusable(x.shape[1], router_w.shape[0], k):         # K1: gating GEMM + routing as ONE launch (bit-identical to the two below)
        logits, scores, idx, counts = ops.moe_router_fused(x, router_w, k)
    else:
        logits = ops.gemm(x, router_w)                               # TopKRouter.gating  moe_lm.py:190-201
        scores, idx, counts = ops.moe_route(logits, k)               # routing :261-269 (device-side histogram)
    offsets, sorted_src, inv = ops.moe_sort(idx, counts)             # token_permutation :326-334 (stable)
    fused = ops.glu_fusable(fc1.shape[1], fc1.shape[2])
    rows = None
    if fused and ops.gather_fusable(fc1.shape[1]) and ((lean or not save) or ops.wgrad_gather_fusable(fc1.shape[1])):
        # K2: the row gather rides in fc1's A loader, the [6T, D] permuted copy is neither written nor read -- inference, the forward pass of a
        # checkpointed step, and (r05) the plain training step too: fc1's weight gradient reaches the token rows through the same index
        # (ops.grouped_gemm_wgrad_gather), so nothing in the backward wants `perm` as a tensor
        perm = None
        rows = ops.permuted_token_rows(sorted_src, k)
        h1, act = ops.grouped_gemm_swiglu_gather(x, rows, fc1, offsets, want_h=bool(save) and not lean)
    else:
        perm = ops.moe_permute(x, sorted_src, k)
        if fused:                                                    # experts.fc1 :522 + glu :505-507 in ONE launch (no D2H sync)
            h1, act = ops.grouped_gemm_swiglu(perm, fc1, offsets, want_h=bool(save) and not lean)   # (h1 is only kept for the backward of glu)
        else:
            h1 = ops.grouped_gemm(perm, fc1, offsets)
            act = ops.swiglu(h1)
    eo = ops.grouped_gemm(act, fc2, offsets)                         # experts.fc2 :524
    T = x.shape[0]
    I2 = gate_w.shape[0]
    wgu = fused_weight(gate_w, up_w)                                 # SharedExpertMLP :368-395
    if ops.glu_fusable(x.shape[1], 2 * I2):
        gu, sact = ops.gemm_swiglu(x, wgu, want_h=bool(save))
    else:
        gu = torch.empty((T, 2 * I2), dtype=bf16, device=x.device)
        ops.gemm(x, wgu, out=gu)
        sact = ops.swiglu(gu)
    sh = ops.gemm(sact, down_w)
    # token_unpermutation :336-365 + `output += shared` :576 (+ the decoder layer's residual add when the caller hands its stream in: r05b)
    out = ops.moe_unpermute(eo, inv, scores, k, add=sh, residual=residual)
    ctx = None
    if save:
        ctx = dict(x=x, logits=logits, scores=scores, idx=idx, counts=counts, offsets=offsets, inv=inv, sorted_src=sorted_src, rows=rows,
                   perm=None if lean else perm, h1=None if lean else h1, act=None if lean else act, eo=None if lean else eo,
                   gu=gu, sact=sact, cfg=cfg, wgu=wgu)
    return out, ctx


def moe_rematerialize(ctx, fc1, fc2) -> None:
    """Rebuild the expert-row tensors a lean ``moe_fwd`` did not keep (same kernels on the same inputs: bit-identical to the kept ones)."""
    if ctx["act"] is not None:
        return
    k = ctx["cfg"].topk
    perm = None
    if ctx.get("rows") is not None and ops.wgrad_gather_fusable(fc1.shape[1]):   # (the weight gradient will gather too: no permuted copy at all)
        h1, act = ops.grouped_gemm_swiglu_gather(ctx["x"], ctx["rows"], fc1, ctx["offsets"], want_h=True)
    else:
        perm = ops.moe_permute(ctx["x"], ctx["sorted_src"], k)
        if ops.glu_fusable(fc1.shape[1], fc1.shape[2]):
            h1, act = ops.grouped_gemm_swiglu(perm, fc1, ctx["offsets"], want_h=True)
        else:
            h1 = ops.grouped_gemm(perm, fc1, ctx["offsets"])
            act = ops.swiglu(h1)
    ctx.update(perm=perm, h1=h1, act=act, eo=ops.grouped_gemm(act, fc2, ctx["offsets"]))


def _want(need, *keys) -> bool:
    """need = None: every weight gradient; else the set of parameter keys whose gradient is wanted (frozen ones are skipped)."""
    return need is None or any(k in need for k in keys)


def moe_bwd(dout, ctx, router_w, fc1, fc2, gate_w, up_w, down_w, need=None):
    """-> dx, dict(grads); weight gradients of frozen parameters (keys missing from ``need``) are not computed (None)."""
    cfg: MoEConfig = ctx["cfg"]
    k, E = cfg.topk, cfg.num_experts
    x, inv, offsets = ctx["x"], ctx["inv"], ctx["offsets"]
    I2 = gate_w.shape[0]
    # routed experts
    d_eo, dscores = ops.moe_unpermute_bwd(dout, ctx["eo"], inv, ctx["scores"], k)
    if ops.dglu_fusable(fc2.shape[1], fc2.shape[2]):                 # fc2's input gradient + the backward of glu in ONE launch
        d_h1 = ops.grouped_gemm_dswiglu(d_eo, fc2, offsets, ctx["h1"])
    else:
        d_h1 = ops.swiglu_bwd(ctx["h1"], ops.grouped_gemm(d_eo, fc2, offsets, w_is_kn=False))
    g_fc2 = ops.grouped_gemm_wgrad(ctx["act"], d_eo, offsets, E) if _want(need, "fc2") else None
    d_perm = ops.grouped_gemm(d_h1, fc1, offsets, w_is_kn=False)
    g_fc1 = None
    if _want(need, "fc1"):
        if ctx["perm"] is None:   # the forward gathered: so does the weight gradient (or, shapes the gathered launch does not take, permute now)
            g_fc1 = ops.grouped_gemm_wgrad_gather(x, ctx["rows"], d_h1, offsets, E)
            if g_fc1 is None:
                g_fc1 = ops.grouped_gemm_wgrad(ops.moe_permute(x, ctx["sorted_src"], k), d_h1, offsets, E)
        else:
            g_fc1 = ops.grouped_gemm_wgrad(ctx["perm"], d_h1, offsets, E)
    dx = ops.moe_unpermute(d_perm, inv, None, k)                     # backward of the row gather
    # shared expert
    if ops.dglu_fusable(down_w.shape[1], down_w.shape[0]):
        d_gu = ops.gemm_dswiglu(dout, down_w, ctx["gu"], b_oc=True)
    else:
        d_gu = ops.swiglu_bwd(ctx["gu"], ops.gemm(dout, down_w, b_oc=True))
    g_down = ops.gemm(dout, ctx["sact"], a_oc=True, b_oc=True) if _want(need, "down") else None
    ops.gemm(d_gu, ctx["wgu"], b_oc=True, out=dx, accumulate=True)
    # gate and up weight gradients as ONE wide GEMM ([2*I2, D] = d_gu^T x): 260 tiles of 256x256 instead of 2 x 130
    g_gate = g_up = None
    if _want(need, "gate", "up"):
        g_gu = ops.gemm(d_gu, x, a_oc=True, b_oc=True)
        g_gate, g_up = g_gu[:I2], g_gu[I2:]
    # router (top-k softmax + z-loss + load-balancing loss gradients)
    dlogits = ops.moe_route_bwd(ctx["logits"], ctx["idx"], ctx["scores"], dscores, ctx["counts"], cfg.z_loss_coeff,
                                cfg.aux_loss_coeff, cfg.aux_scale)
    ops.gemm(dlogits, router_w, b_oc=True, out=dx, accumulate=True)
    g_router = ops.gemm(dlogits, x, a_oc=True, b_oc=True) if _want(need, "router") else None
    return dx, dict(router=g_router, fc1=g_fc1, fc2=g_fc2, gate=g_gate, up=g_up, down=g_down)


# ----------------------------------------------------------------------------------------------- attention block
@dataclass
class AttnConfig:
    num_heads: int
    num_kv_heads: int
    head_dim: int
    causal: bool = True


_SUPPORTED_HD = (64, 72, 128)  # forward + backward kernels (72 = ViT / projector heads: native, tiles padded inside the kernels)
_FWD_ONLY_HD = (64, 72, 128)   # forward kernel (72 = ViT / projector heads, native: no padding)


def _pad_hd(hd: int, need_bwd: bool = True) -> int:
    for s in (_SUPPORTED_HD if need_bwd else _FWD_ONLY_HD):
        if hd <= s:
            return s
    raise ValueError(f"head_dim {hd} > 128 is not supported")


def _pad_heads(t: torch.Tensor, H: int, hd: int, hdp: int) -> torch.Tensor:
    """[T, H*hd] -> zero-padded [T, H*hdp] (only for head dims the kernels do not implement natively)."""
    T = t.shape[0]
    out = torch.zeros((T, H, hdp), dtype=t.dtype, device=t.device)
    out[:, :, :hd] = t.reshape(T, H, hd)
    return out.view(T, H * hdp)


def _unpad_heads(t: torch.Tensor, H: int, hd: int, hdp: int) -> torch.Tensor:
    return t.view(t.shape[0], H, hdp)[:, :, :hd].reshape(t.shape[0], H * hd)


def sdpa_fwd(q, k, v, B, S, H, hd, scale, causal, kv_len=None):
    """q,k,v [B*S, H*hd] views -> (o [B*S, H*hd], ctx).  Pads the head dim to 64/128 when needed."""
    hdp = _pad_hd(hd)
    if hdp != hd:
        qp, kp, vp = (_pad_heads(t, H, hd, hdp) for t in (q, k, v))
        o, lse = ops.attention_fwd(qp, kp, vp, B, S, H, hdp, scale, causal, kv_len)
        return _unpad_heads(o, H, hd, hdp), dict(q=qp, k=kp, v=vp, o=o, lse=lse, hdp=hdp)
    o, lse = ops.attention_fwd(q, k, v, B, S, H, hd, scale, causal, kv_len)
    return o, dict(q=q, k=k, v=v, o=o, lse=lse, hdp=hd)


def sdpa_bwd(do, ctx, B, S, H, hd, scale, causal, kv_len=None, dq=None, dk=None, dv=None, rope=None):
    hdp = ctx["hdp"]
    if hdp != hd:
        assert rope is None
        dop = _pad_heads(do, H, hd, hdp)
        gq, gk, gv = ops.attention_bwd(ctx["q"], ctx["k"], ctx["v"], ctx["o"], dop, ctx["lse"], B, S, H, hdp, scale, causal, kv_len)
        outs = [_unpad_heads(t, H, hd, hdp) for t in (gq, gk, gv)]
        for dst, src in zip((dq, dk, dv), outs):
            if dst is not None:
                dst.copy_(src)
        return tuple(d if d is not None else s for d, s in zip((dq, dk, dv), outs))
    return ops.attention_bwd(ctx["q"], ctx["k"], ctx["v"], ctx["o"], do, ctx["lse"], B, S, H, hd, scale, causal, kv_len,
                             dq=dq, dk=dk, dv=dv, rope=rope)


def attn_block_fwd(x, wq, wk, wv, wo, cos, sin, B: int, S: int, cfg: AttnConfig, kv_len=None, save: bool = True, attn_cache=None,
                   keep_attn: bool = False):
    """LlamaAttention.forward on normalised x [B*S, D]: q/k/v proj, half-split RoPE, causal flash attention, o proj.
    ``keep_attn`` (with save=False): the returned ctx holds only the flash kernel's (o, lse); ``attn_cache`` = such a pair from an
    earlier, identical forward: the flash kernel is skipped (selective recompute: at 64K tokens it is 55 % of a layer's forward)."""
    H, Hkv, hd = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
    if Hkv != H:
        raise NotImplementedError("GQA (num_key_value_heads != num_attention_heads): Aria is MHA (gptfast/model.py:56-58)")
    T = x.shape[0]
    Dq = H * hd
    qkv = torch.empty((T, 3 * Dq), dtype=bf16, device=x.device)
    wqkv = fused_weight(wq, wk, wv)
    ops.gemm_qkv_rope(x, wqkv, cos, sin, S, hd, out=qkv)           # q | k | v projection with the rotation as its epilogue (one launch)
    if attn_cache is not None and _pad_hd(hd) == hd:
        o, lse = attn_cache
        actx = dict(q=qkv[:, :Dq], k=qkv[:, Dq:2 * Dq], v=qkv[:, 2 * Dq:], o=o, lse=lse, hdp=hd)
    else:
        o, actx = sdpa_fwd(qkv[:, :Dq], qkv[:, Dq:2 * Dq], qkv[:, 2 * Dq:], B, S, H, hd, hd ** -0.5, cfg.causal, kv_len)
    out = ops.gemm(o if o.is_contiguous() else o.contiguous(), wo)
    if save:
        ctx = dict(x=x, o=o, actx=actx, B=B, S=S, cfg=cfg, kv_len=kv_len, wqkv=wqkv)
    else:
        ctx = dict(attn_cache=(actx["o"], actx["lse"])) if keep_attn and actx["hdp"] == hd else None
    return out, ctx


def attn_block_bwd(dout, ctx, wq, wk, wv, wo, cos, sin, need=None):
    cfg: AttnConfig = ctx["cfg"]
    H, hd = cfg.num_heads, cfg.head_dim
    B, S, x, o = ctx["B"], ctx["S"], ctx["x"], ctx["o"]
    T, Dq = x.shape[0], H * hd
    d_o = ops.gemm(dout, wo, b_oc=True)
    g_wo = ops.gemm(dout, o if o.is_contiguous() else o.contiguous(), a_oc=True, b_oc=True) if _want(need, "wo") else None
    dqkv = torch.empty((T, 3 * Dq), dtype=bf16, device=x.device)
    # r05: the inverse rotation of dq | dk rides in the attention backward's register epilogues (was a pass over [T, 2 D] behind them)
    fused_rope = ctx["actx"]["hdp"] == hd and ops.attention_bwd_rope_fusable(hd, S, cos)
    sdpa_bwd(d_o, ctx["actx"], B, S, H, hd, hd ** -0.5, cfg.causal, ctx["kv_len"], dq=dqkv[:, :Dq], dk=dqkv[:, Dq:2 * Dq],
             dv=dqkv[:, 2 * Dq:], rope=(cos, sin) if fused_rope else None)
    if not fused_rope:
        ops.rope_(dqkv[:, :2 * Dq], cos, sin, S, 2 * H, hd, inverse=True)
    dx = ops.gemm(dqkv, ctx["wqkv"], b_oc=True)
    # q, k, v weight gradients as ONE wide GEMM ([3*Dq, D] = dqkv^T x)
    g_wq = g_wk = g_wv = None
    if _want(need, "wq", "wk", "wv"):
        g_qkv = ops.gemm(dqkv, x, a_oc=True, b_oc=True)
        g_wq, g_wk, g_wv = g_qkv[:Dq], g_qkv[Dq:2 * Dq], g_qkv[2 * Dq:]
    return dx, dict(q=g_wq, k=g_wk, v=g_wv, o=g_wo)


# ----------------------------------------------------------------------------------------------- decoder layer
def decoder_layer_fwd(x, p: dict, cos, sin, B: int, S: int, acfg: AttnConfig, mcfg: MoEConfig, eps: float, kv_len=None,
                      save=True, attn_cache=None, keep_attn: bool = False):
    """h = x + Attn(RMSNorm(x)); out = h + MoE(RMSNorm(h)).  p: dict of the layer's parameter tensors.
    save=False, keep_attn=True -> ctx = {"attn_cache": (o, lse)} for a later recomputing call with attn_cache=...;
    save="lean" -> the full ctx minus the MoE block's expert-row tensors (``moe_rematerialize`` before ``decoder_layer_bwd``)."""
    full = bool(save)
    xn, _, rstd1 = ops.rmsnorm(x, p["ln1"], eps, want_rstd=full)
    a, actx = attn_block_fwd(xn, p["wq"], p["wk"], p["wv"], p["wo"], cos, sin, B, S, acfg, kv_len, full, attn_cache, keep_attn)
    hn, h, rstd2 = ops.rmsnorm(a, p["ln2"], eps, residual=x, want_rstd=full)   # fused residual add
    out, mctx = moe_fwd(hn, p["router"], p["fc1"], p["fc2"], p["gate"], p["up"], p["down"], mcfg, save, residual=h)   # h + MoE(hn), one launch less
    ctx = dict(x=x, h=h, rstd1=rstd1, rstd2=rstd2, actx=actx, mctx=mctx, eps=eps) if full else actx
    return out, ctx


def decoder_layer_bwd(dout, ctx, p: dict, cos, sin, need=None):
    dhn, gm = moe_bwd(dout, ctx["mctx"], p["router"], p["fc1"], p["fc2"], p["gate"], p["up"], p["down"], need)
    dh, g_ln2 = ops.rmsnorm_bwd(dhn, ctx["h"], p["ln2"], ctx["rstd2"], dres=dout)       # + gradient of the residual stream
    dxn, ga = attn_block_bwd(dh, ctx["actx"], p["wq"], p["wk"], p["wv"], p["wo"], cos, sin, need)
    dx, g_ln1 = ops.rmsnorm_bwd(dxn, ctx["x"], p["ln1"], ctx["rstd1"], dres=dh)
    grads = dict(ln1=g_ln1, ln2=g_ln2, wq=ga["q"], wk=ga["k"], wv=ga["v"], wo=ga["o"], router=gm["router"], fc1=gm["fc1"],
                 fc2=gm["fc2"], gate=gm["gate"], up=gm["up"], down=gm["down"])
    return dx, grads


# ----------------------------------------------------------------------------------------------- RoPE tables (host)
def rope_tables(S: int, head_dim: int, theta: float, device) -> tuple:
    """bf16 cos/sin [S, head_dim] built exactly like LlamaRotaryEmbedding (transformers/.../modeling_llama.py:96-127):
    fp32 angles on the host, emb = cat(freqs, freqs), cast to the activation dtype."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    freqs = torch.arange(S, dtype=torch.float32)[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(bf16).to(device).contiguous(), emb.sin().to(bf16).to(device).contiguous()


# ----------------------------------------------------------------------------------------------- LM head + loss
def lm_head_loss_fwd_bwd(hn, lm_w, labels_shifted, need_grads: bool = True, need_w: bool = True):
    """logits = hn @ lm_w^T; masked-mean CE (labels already shifted, -100 = ignore).  The CE kernel overwrites the logits
    with d(loss)/d(logits) (scaled by 1/n_valid on the device: no host sync), so the backward GEMMs can run immediately.
    -> (loss fp32 scalar tensor, d_hn | None, g_lm_w | None)

    Positions whose label is ignored contribute neither to the loss nor to any gradient, so only the rows with a label go through the
    [rows, V] GEMMs (an SFT batch masks its prompts: 75 % of the positions in the benchmark's batch) -- one host sync for the row count,
    a row gather before and a row scatter after.  Measured on MI355X: 714.6 -> 701.7 ms per config #3 step, identical loss
    (ARIA_LMHEAD_SKIP_MASKED=0 switches it off)."""
    if os.environ.get("ARIA_LMHEAD_SKIP_MASKED", "1") != "0":
        return _lm_head_loss_valid_rows(hn, lm_w, labels_shifted, need_grads, need_w)
    logits = ops.gemm(hn, lm_w)
    count_in = (labels_shifted >= 0).sum(dtype=torch.int32).reshape(1)
    loss_sum, count, _ = ops.cross_entropy(logits, labels_shifted, grad_scale=1.0, dlogits=logits if need_grads else None,
                                           count_in=count_in)
    loss = (loss_sum / count_in.clamp(min=1).to(torch.float32)).reshape(())
    if not need_grads:
        return loss, None, None
    d_hn = ops.gemm(logits, lm_w, b_oc=True)
    g_w = ops.gemm(logits, hn, a_oc=True, b_oc=True) if need_w else None   # (a frozen lm_head needs no [V, D] weight gradient)
    return loss, d_hn, g_w


def _lm_head_loss_valid_rows(hn, lm_w, labels_shifted, need_grads: bool, need_w: bool = True):
    rows = torch.nonzero(labels_shifted >= 0).flatten().to(torch.int32)          # (host sync: the GEMMs' M)
    n = int(rows.numel())
    if n == 0:
        zero = torch.zeros((), dtype=torch.float32, device=hn.device)
        return zero, (torch.zeros_like(hn) if need_grads else None), (torch.zeros_like(lm_w) if need_grads and need_w else None)
    pad = (-n) % 8                                                                 # the wgrad reads dlogits [rows, V] output-contiguous: rows % 8
    if pad:
        rows = torch.cat([rows, rows[-1:].expand(pad)])
    labels_v = labels_shifted[rows.long()].contiguous()
    if pad:
        labels_v[n:] = -100                                                        # padding rows: no loss, zero gradient rows
    hv = ops.moe_permute(hn, rows, 1)                                              # row gather (the embedding-lookup kernel)
    logits = ops.gemm(hv, lm_w)
    count_in = torch.full((1,), n, dtype=torch.int32, device=hn.device)
    loss_sum, _, _ = ops.cross_entropy(logits, labels_v, grad_scale=1.0, dlogits=logits if need_grads else None, count_in=count_in)
    loss = (loss_sum / float(n)).reshape(())
    if not need_grads:
        return loss, None, None
    d_hv = ops.gemm(logits, lm_w, b_oc=True)
    g_w = ops.gemm(logits, hv, a_oc=True, b_oc=True) if need_w else None
    d_hn = torch.zeros_like(hn)
    d_hn.index_copy_(0, rows[:n].long(), d_hv[:n])                                 # distinct rows: a plain scatter
    return loss, d_hn, g_w


def shift_labels(labels: torch.Tensor, attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """aria/model/modeling_aria.py:301-316 as a per-position label vector: position t is scored against labels[t+1]
    iff attention_mask[t+1] != 0; the last position of every sequence is ignored.  [B,S] -> int32 [B*S]."""
    B, S = labels.shape
    out = torch.full((B, S), -100, dtype=torch.int32, device=labels.device)
    nxt = labels[:, 1:].to(torch.int32)
    if attention_mask is not None:
        nxt = torch.where(attention_mask[:, 1:] != 0, nxt, torch.full_like(nxt, -100))
    out[:, :-1] = nxt
    return out.reshape(-1)
