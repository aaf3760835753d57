"""``torch.autograd.Function`` wrappers that make the hand-written forward/backward blocks of
``aria_amd.functional`` differentiable drop-ins (seams B1-B3 of SURVEY.md section 8b)."""
from __future__ import annotations

import os

from typing import Optional

import torch

from . import functional as Fn
from . import ops

bf16 = torch.bfloat16


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ b) on 2-D x."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return ops.gemm(x, w, bias=bias)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dx = ops.gemm(dy, w, b_oc=True) if ctx.needs_input_grad[0] else None
        dw = ops.gemm(dy, x, a_oc=True, b_oc=True) if ctx.needs_input_grad[1] else None
        db = ops.colsum(dy) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    shp = x.shape
    y = LinearFn.apply(_c(x.reshape(-1, shp[-1])), w, bias)
    return y.view(*shp[:-1], w.shape[0])


class RMSNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, eps):
        y, _, rstd = ops.rmsnorm(x, w, eps)
        ctx.save_for_backward(x, w, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, rstd = ctx.saved_tensors
        dx, dw = ops.rmsnorm_bwd(_c(dy), x, w, rstd)
        return dx, dw, None


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    shp = x.shape
    return RMSNormFn.apply(_c(x.reshape(-1, shp[-1])), w, eps).view(shp)


class ExpertsGemmFn(torch.autograd.Function):
    """experts_gemm(input, weight, tokens_per_expert) -- seam B1 (aria/model/moe_lm.py:431-443), differentiable in
    input and weight like grouped_gemm.ops.gmm."""

    @staticmethod
    def forward(ctx, inp, weight, offsets):
        ctx.save_for_backward(inp, weight, offsets)
        return ops.grouped_gemm(inp, weight, offsets)

    @staticmethod
    def backward(ctx, dy):
        inp, weight, offsets = ctx.saved_tensors
        dy = _c(dy)
        dx = ops.grouped_gemm(dy, weight, offsets, w_is_kn=False) if ctx.needs_input_grad[0] else None
        dw = ops.grouped_gemm_wgrad(inp, dy, offsets, weight.shape[0]) if ctx.needs_input_grad[1] else None
        return dx, dw, None


def offsets_from_tokens_per_expert(tpe: torch.Tensor, device) -> torch.Tensor:
    """int32 [E+1] device offsets from a tokens_per_expert tensor (CPU int64 in the reference, moe_lm.py:478)."""
    off = torch.zeros(tpe.numel() + 1, dtype=torch.int32, device=tpe.device)
    off[1:] = torch.cumsum(tpe.to(torch.int32), 0)
    return off.to(device, non_blocking=True)


def experts_gemm(inp: torch.Tensor, weight: torch.Tensor, tokens_per_expert: torch.Tensor) -> torch.Tensor:
    return ExpertsGemmFn.apply(_c(inp), weight, offsets_from_tokens_per_expert(tokens_per_expert, inp.device))


class SwiGLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return ops.swiglu(h)

    @staticmethod
    def backward(ctx, dact):
        (h,) = ctx.saved_tensors
        return ops.swiglu_bwd(h, _c(dact))


class ExpertsGluFn(torch.autograd.Function):
    """glu(experts_gemm(input, fc1, tokens_per_expert)) -- GroupedMLP.forward's first half (moe_lm.py:505-507, 522-523) as ONE launch
    (fc1 GEMM with the SwiGLU epilogue), differentiable in input and weight; bit-identical to SwiGLUFn(ExpertsGemmFn(...))."""

    @staticmethod
    def forward(ctx, inp, weight, offsets):
        h, act = ops.grouped_gemm_swiglu(inp, weight, offsets, want_h=True)
        ctx.save_for_backward(inp, weight, offsets, h)
        return act

    @staticmethod
    def backward(ctx, dact):
        inp, weight, offsets, h = ctx.saved_tensors
        dh = ops.swiglu_bwd(h, _c(dact))
        dx = ops.grouped_gemm(dh, weight, offsets, w_is_kn=False) if ctx.needs_input_grad[0] else None
        dw = ops.grouped_gemm_wgrad(inp, dh, offsets, weight.shape[0]) if ctx.needs_input_grad[1] else None
        return dx, dw, None


class ExpertsGemmSegFn(torch.autograd.Function):
    """``ExpertsGemmFn`` over the SEGMENTS of an expert-parallel exchange: rows ordered (source rank, local expert), offsets int32
    [ranks * E_local + 1]; segment g uses weight[g % E_local] (``ops.grouped_gemm_seg``) -- the all-to-all's output is consumed in arrival
    order, forward and backward; the weight gradient sums the source ranks' segments of an expert (fp32 when there are several)."""

    @staticmethod
    def forward(ctx, inp, weight, offsets):
        ctx.save_for_backward(inp, weight, offsets)
        return ops.grouped_gemm_seg(inp, weight, offsets)

    @staticmethod
    def backward(ctx, dy):
        inp, weight, offsets = ctx.saved_tensors
        dy = _c(dy)
        dx = ops.grouped_gemm_seg(dy, weight, offsets, w_is_kn=False) if ctx.needs_input_grad[0] else None
        dw = ops.grouped_gemm_wgrad_seg(inp, dy, offsets, weight.shape[0]) if ctx.needs_input_grad[1] else None
        return dx, dw, None


class ExpertsGluSegFn(torch.autograd.Function):
    """``ExpertsGluFn`` over segments (fc1 + SwiGLU in one launch on the exchange's output as it arrived)."""

    @staticmethod
    def forward(ctx, inp, weight, offsets):
        h, act = ops.grouped_gemm_swiglu_seg(inp, weight, offsets, want_h=True)
        ctx.save_for_backward(inp, weight, offsets, h)
        return act

    @staticmethod
    def backward(ctx, dact):
        inp, weight, offsets, h = ctx.saved_tensors
        dh = ops.swiglu_bwd(h, _c(dact))
        dx = ops.grouped_gemm_seg(dh, weight, offsets, w_is_kn=False) if ctx.needs_input_grad[0] else None
        dw = ops.grouped_gemm_wgrad_seg(inp, dh, offsets, weight.shape[0]) if ctx.needs_input_grad[1] else None
        return dx, dw, None


class SharedGluFn(torch.autograd.Function):
    """silu(gate_proj(x)) * up_proj(x) of SharedExpertMLP (moe_lm.py:368-395) as ONE GEMM over the row-wise concatenation of the two
    weights with the SwiGLU epilogue; the input gradient is one GEMM with the long reduction, the two weight gradients one wide GEMM."""

    @staticmethod
    def forward(ctx, x, gate_w, up_w):
        wgu = Fn.fused_weight(gate_w, up_w)
        gu, act = ops.gemm_swiglu(x, wgu, want_h=True)
        ctx.save_for_backward(x, wgu, gu)
        ctx.I2 = gate_w.shape[0]
        return act

    @staticmethod
    def backward(ctx, dact):
        x, wgu, gu = ctx.saved_tensors
        d_gu = ops.swiglu_bwd(gu, _c(dact))
        dx = ops.gemm(d_gu, wgu, b_oc=True) if ctx.needs_input_grad[0] else None
        g_gate = g_up = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            g_gu = ops.gemm(d_gu, x, a_oc=True, b_oc=True)
            g_gate, g_up = g_gu[:ctx.I2], g_gu[ctx.I2:]
        return dx, g_gate, g_up


class MoELayerFn(torch.autograd.Function):
    """MoELayer.forward (moe_lm.py:548-577) as one node: router -> permute -> experts -> unpermute + shared."""

    @staticmethod
    def forward(ctx, x, router_w, fc1, fc2, gate_w, up_w, down_w, cfg):
        out, c = Fn.moe_fwd(x, router_w, fc1, fc2, gate_w, up_w, down_w, cfg, save=True)
        ctx.c = c
        ctx.save_for_backward(router_w, fc1, fc2, gate_w, up_w, down_w)
        return out

    @staticmethod
    def backward(ctx, dout):
        router_w, fc1, fc2, gate_w, up_w, down_w = ctx.saved_tensors
        dx, g = Fn.moe_bwd(_c(dout), ctx.c, router_w, fc1, fc2, gate_w, up_w, down_w)
        ctx.c = None
        return dx, g["router"], g["fc1"], g["fc2"], g["gate"], g["up"], g["down"], None


class AttnBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wq, wk, wv, wo, cos, sin, B, S, cfg, kv_len):
        out, c = Fn.attn_block_fwd(x, wq, wk, wv, wo, cos, sin, B, S, cfg, kv_len, save=True)
        ctx.c = c
        ctx.save_for_backward(wq, wk, wv, wo, cos, sin)
        return out

    @staticmethod
    def backward(ctx, dout):
        wq, wk, wv, wo, cos, sin = ctx.saved_tensors
        dx, g = Fn.attn_block_bwd(_c(dout), ctx.c, wq, wk, wv, wo, cos, sin)
        ctx.c = None
        return dx, g["q"], g["k"], g["v"], g["o"], None, None, None, None, None, None


_LAYER_KEYS = ("ln1", "wq", "wk", "wv", "wo", "ln2", "router", "fc1", "fc2", "gate", "up", "down")

RECOMPUTE_LEVELS = ("moe", "layer")


def recompute_kept_bytes(level: str, tokens: int, hidden: int, shared_inter: int, n_layers: int) -> int:
    """What ``DecoderLayerFn`` keeps between forward and backward over the whole stack, bf16: "moe" = every token-sized tensor of a layer
    (x, normed x, q|k|v, o, h, normed h, the shared expert's gate|up and activation, router logits and routing metadata: ~(8 D + 3 Is) values
    per token = 61 KB at Aria's width, 4.1 GB per layer at 64K tokens); "layer" = the layer input and the flash kernel's (o, lse)."""
    per_token = (8 * hidden + 3 * shared_inter + 256) * 2 if level == "moe" else (2 * hidden + 64) * 2
    return int(tokens) * per_token * int(n_layers)


def choose_recompute_level(tokens: int, hidden: int, shared_inter: int, expert_rows_per_token: int, expert_inter: int, n_layers: int,
                           pending_grad_bytes: int = 0, device=None, requested: str = "auto") -> str:
    """The recompute level of the recipe's ``gradient_checkpointing`` (level "moe" keeps ~115 GB at 64K tokens x 28 layers -- fine
    next to weights and gradients on 288 GB, not next to everything a long-sequence fine-tune may hold).  ``ARIA_RECOMPUTE_LEVEL`` or the
    config's ``recompute_level`` ("moe" / "layer") force a level; "auto" takes "moe" when what it keeps, plus the gradients still to be
    allocated, plus one layer's complete backward state and the loss buffers, fits in 85 % of the memory that is free NOW on the device,
    and the reference's own form ("layer": layer inputs only, recipes/config_full.yaml:17) otherwise.  Without a GPU (emulator): "moe"."""
    level = os.environ.get("ARIA_RECOMPUTE_LEVEL") or requested or "auto"
    if level in RECOMPUTE_LEVELS:
        return level
    if level != "auto":
        raise ValueError(f"recompute level {level!r}: expected one of {RECOMPUTE_LEVELS + ('auto',)}")
    if device is None or torch.device(device).type != "cuda" or not torch.cuda.is_available():
        return "moe"
    free, _total = torch.cuda.mem_get_info(device)
    free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)     # the caching allocator's idle blocks are ours too
    keep = recompute_kept_bytes("moe", tokens, hidden, shared_inter, n_layers)
    # one layer's full backward state (the kept part + the four expert-row tensors + as much again for the gradients flowing through it)
    # and lm_head logits / dlogits rows for a quarter of the positions at a 100K vocabulary
    one_layer = 2 * (keep // max(n_layers, 1) + tokens * expert_rows_per_token * (2 * hidden + 3 * expert_inter) * 2)
    loss = tokens // 4 * 100352 * 2 * 2
    return "moe" if keep + pending_grad_bytes + one_layer + loss <= 0.85 * free else "layer"



class DecoderLayerFn(torch.autograd.Function):
    """One MoEDecoderLayer (moe_lm.py:580-602) as a single autograd node with a hand-written backward.
    ``recompute`` (= the reference recipe's gradient_checkpointing, recipes/config_full.yaml:17) is SELECTIVE.  Of the 36 KB per token a
    layer's backward needs, 24 KB are the four expert-row tensors of the MoE block (perm, fc1 output, its activation, fc2 output: 6 rows
    per token each); the default level ("moe") keeps everything else and rebuilds those four in the backward -- a row gather and the two
    routed-expert GEMMs, 60 % of the layer's forward GEMM flops, none of its attention.  At 64K tokens that is 4.1 GB per layer kept
    instead of 12 (115 GB for 28 layers next to 101 GB of weights and gradients).  Level "layer": keep only the layer input and the flash
    kernel's (o, lse) and run the whole layer again (round 2's form, the reference recipe's: 1.0 GB per layer at 64K).  The model picks the
    level once per forward from the memory that is free (``choose_recompute_level``); ARIA_RECOMPUTE_LEVEL / config.recompute_level force it."""

    @staticmethod
    def forward(ctx, x, cos, sin, B, S, acfg, mcfg, eps, kv_len, recompute, *params):
        p = dict(zip(_LAYER_KEYS, params))
        # recompute: False / None, True (= "moe", or what ARIA_RECOMPUTE_LEVEL says) or the level itself (the model picks it once per forward)
        level = (recompute if recompute in RECOMPUTE_LEVELS else os.environ.get("ARIA_RECOMPUTE_LEVEL", "moe")) if recompute else None
        recompute = bool(recompute)
        if level == "moe":  # keep the layer's token-sized tensors, rebuild the four expert-row tensors (perm, h1, act, eo) in the backward
            out, c = Fn.decoder_layer_fwd(x, p, cos, sin, B, S, acfg, mcfg, eps, kv_len, save="lean")
        else:               # "layer": keep only the flash kernel's (o, lse), run the whole layer again
            out, c = Fn.decoder_layer_fwd(x, p, cos, sin, B, S, acfg, mcfg, eps, kv_len, save=not recompute, keep_attn=bool(recompute))
        ctx.c = c
        ctx.meta = (B, S, acfg, mcfg, eps, kv_len, level)
        ctx.save_for_backward(x, cos, sin, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, cos, sin, *params = ctx.saved_tensors
        B, S, acfg, mcfg, eps, kv_len, level = ctx.meta
        p = dict(zip(_LAYER_KEYS, params))
        c = ctx.c
        if level == "moe":
            Fn.moe_rematerialize(c["mctx"], p["fc1"], p["fc2"])
        elif level is not None:
            _, c = Fn.decoder_layer_fwd(x, p, cos, sin, B, S, acfg, mcfg, eps, kv_len, save=True,
                                        attn_cache=None if c is None else c.get("attn_cache"))
        wanted = {k for k, w in zip(_LAYER_KEYS, ctx.needs_input_grad[10:]) if w}   # frozen parameters: no weight-gradient GEMM
        dx, g = Fn.decoder_layer_bwd(_c(dout), c, p, cos, sin, None if len(wanted) == len(_LAYER_KEYS) else wanted)
        ctx.c = None
        return (dx, None, None, None, None, None, None, None, None, None) + tuple(g[k] if k in wanted else None for k in _LAYER_KEYS)


class LoraDecoderLayerFn(torch.autograd.Function):
    """One MoEDecoderLayer whose GEMMs carry LoRA adapters (recipes/config_lora.yaml:44-59) as a single node: the adapters' second projections
    ride inside the base launches (K-extension), everything else is the un-adapted layer's kernel sequence (aria_amd.lora_functional).
    ``keys``: the adapted parameter keys (subset of lora_functional.SITES), ``hyper``: (scaling, dropout p) per key; tensors: the 12 layer
    parameters, then (lora_A.weight, lora_B.weight) per key.  The dropout masks are a function of ``seed``.
    ``recompute`` (the recipe's gradient_checkpointing, recipes/config_lora.yaml:17) is done INSIDE the node (torch's non-reentrant
    checkpoint only intercepts save_for_backward tensors, so wrapping this node kept every activation alive in ``ctx.c`` AND ran the forward
    twice): the forward keeps the layer input and the seed, nothing else; the backward runs decoder_layer_lora_fwd again -- same seed, same
    masks, same bits -- and hands its context to decoder_layer_lora_bwd."""

    forward_calls = 0   # (test hook: how many times the layer forward ran)

    @staticmethod
    def forward(ctx, x, cos, sin, B, S, acfg, mcfg, eps, kv_len, training, seed, keys, hyper, recompute, *tensors):
        from . import lora_functional as LF

        params, ab = tensors[:len(_LAYER_KEYS)], tensors[len(_LAYER_KEYS):]
        p = dict(zip(_LAYER_KEYS, params))
        L = {k: LF.LoraSite(ab[2 * i], ab[2 * i + 1], hyper[i][0], hyper[i][1]) for i, k in enumerate(keys)}
        out, c = LF.decoder_layer_lora_fwd(x, p, L, cos, sin, B, S, acfg, mcfg, eps, kv_len, training, seed)
        LoraDecoderLayerFn.forward_calls += 1
        ctx.keys, ctx.hyper = keys, hyper
        ctx.meta = (B, S, acfg, mcfg, eps, kv_len, training, seed, bool(recompute))
        ctx.c = None if recompute else c
        ctx.save_for_backward(x, cos, sin, *tensors)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import lora_functional as LF

        x, cos, sin, *tensors = ctx.saved_tensors
        params, ab_in = tensors[:len(_LAYER_KEYS)], tensors[len(_LAYER_KEYS):]
        p = dict(zip(_LAYER_KEYS, params))
        B, S, acfg, mcfg, eps, kv_len, training, seed, recompute = ctx.meta
        c = ctx.c
        if recompute:
            L = {k: LF.LoraSite(ab_in[2 * i], ab_in[2 * i + 1], ctx.hyper[i][0], ctx.hyper[i][1]) for i, k in enumerate(ctx.keys)}
            with torch.no_grad():
                _, c = LF.decoder_layer_lora_fwd(x, p, L, cos, sin, B, S, acfg, mcfg, eps, kv_len, training, seed)
            LoraDecoderLayerFn.forward_calls += 1
        n0 = 14
        need = {k for k, w in zip(_LAYER_KEYS, ctx.needs_input_grad[n0:n0 + len(_LAYER_KEYS)]) if w}
        dx, gb, g = LF.decoder_layer_lora_bwd(_c(dout), c, p, cos, sin, need)
        ctx.c = None
        ab = []
        for k in ctx.keys:
            ab += list(g.get(k, (None, None)))
        return (dx,) + (None,) * 13 + tuple(gb[k] if k in need else None for k in _LAYER_KEYS) + tuple(ab)


class LMHeadLossFn(torch.autograd.Function):
    """final-norm output -> lm_head -> shifted masked CE (modeling_aria.py:301-323); gradients are produced during
    the forward (the logits buffer is overwritten with dlogits), backward just scales them."""

    @staticmethod
    def forward(ctx, hn, lm_w, labels_shifted):
        need = hn.requires_grad or lm_w.requires_grad
        loss, d_hn, g_w = Fn.lm_head_loss_fwd_bwd(hn, lm_w, labels_shifted, need_grads=need, need_w=ctx.needs_input_grad[1])
        ctx.d_hn, ctx.g_w = d_hn, g_w
        return loss

    @staticmethod
    def backward(ctx, dloss):
        d_hn, g_w = ctx.d_hn, ctx.g_w
        ctx.d_hn = ctx.g_w = None
        # dloss is a device scalar (ones for a plain loss.backward()): scale in place, no host sync
        if d_hn is not None:
            ops.scale_(d_hn, dloss)
        if g_w is not None:
            ops.scale_(g_w, dloss)
        return d_hn, g_w, None


class LoraLMHeadLossFn(torch.autograd.Function):
    """LMHeadLossFn with a LoRA adapter on lm_head (lora_functional.lm_head_lora_loss_fwd_bwd): labelled rows only, the adapter's second
    projection inside the lm_head launch; gradients produced during the forward like the un-adapted node."""

    @staticmethod
    def forward(ctx, hn, lm_w, labels_shifted, lora_a, lora_b, scaling, p, training, seed):
        from . import lora_functional as LF

        loss, d_hn, dA, dB = LF.lm_head_lora_loss_fwd_bwd(hn, lm_w, labels_shifted, LF.LoraSite(lora_a, lora_b, scaling, p), training, seed,
                                                         need_hn=hn.requires_grad)
        ctx.g = (d_hn, dA, dB)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        d_hn, dA, dB = ctx.g
        ctx.g = None
        for t in (d_hn, dA, dB):
            if t is not None:
                ops.scale_(t, dloss)
        return d_hn, None, None, dA, dB, None, None, None, None


class EmbeddingFn(torch.autograd.Function):
    """embed_tokens lookup (modeling_aria.py:250) as a row gather; backward = row scatter-add."""

    @staticmethod
    def forward(ctx, ids32, weight):
        ctx.save_for_backward(ids32)
        ctx.shape = weight.shape
        return ops.moe_permute(weight, ids32, 1)

    @staticmethod
    def backward(ctx, dy):
        (ids32,) = ctx.saved_tensors
        dw = torch.zeros(ctx.shape, dtype=bf16, device=dy.device)
        ops.embedding_bwd(_c(dy), ids32, dw)
        return None, dw


class RopeFn(torch.autograd.Function):
    """Half-split RoPE (modeling_llama.py:130-160) on the first n_heads*hd columns of token-major x [B*S, >= n_heads*hd] as its own
    node (the fused attention block applies it in place); the backward of a rotation is the inverse rotation."""

    @staticmethod
    def forward(ctx, x, cos, sin, S, n_heads, hd):
        ctx.save_for_backward(cos, sin)
        ctx.dims = (S, n_heads, hd)
        return ops.rope_(x.clone(), cos, sin, S, n_heads, hd)

    @staticmethod
    def backward(ctx, dy):
        cos, sin = ctx.saved_tensors
        S, n_heads, hd = ctx.dims
        return ops.rope_(dy.contiguous().clone(), cos, sin, S, n_heads, hd, inverse=True), None, None, None, None, None


class SdpaFn(torch.autograd.Function):
    """softmax(q k^T * scale + mask) v over token-major [B*S, H*hd] operands (aria_attn_fwd / aria_attn_bwd): causal and / or
    kv_len int32 [B] (right padding) and / or key_mask uint8 [B, Skv]; head dims 64 / 128."""

    @staticmethod
    def forward(ctx, q, k, v, key_mask, kv_len, B, Sq, Skv, H, hd, scale, causal):
        o, lse = ops.attention_fwd(q, k, v, B, Sq, H, hd, scale, causal, kv_len=kv_len, key_mask=key_mask, Skv=Skv)
        ctx.save_for_backward(q, k, v, o, lse, key_mask, kv_len)
        ctx.dims = (B, Sq, Skv, H, hd, scale, causal)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, key_mask, kv_len = ctx.saved_tensors
        B, Sq, Skv, H, hd, scale, causal = ctx.dims
        dq, dk, dv = ops.attention_bwd(q, k, v, o, _c(do), lse, B, Sq, H, hd, scale, causal, kv_len=kv_len, key_mask=key_mask, Skv=Skv)
        return dq, dk, dv, None, None, None, None, None, None, None, None, None


def sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, Sq: int, Skv: int, H: int, hd: int, scale: float, causal: bool,
         key_mask: Optional[torch.Tensor] = None, kv_len: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable flash attention on token-major bf16 operands.  Head dims the kernels do not implement natively are zero-padded on
    the host: the backward has 64 / 128, the forward also 72 (the frozen ViT under no_grad runs unpadded)."""
    need_bwd = torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)
    hdp = Fn._pad_hd(hd, need_bwd)
    if hdp != hd:
        q, k, v = (Fn._pad_heads(t, H, hd, hdp) for t in (q, k, v))
    if need_bwd:
        o = SdpaFn.apply(_c(q), _c(k), _c(v), key_mask, kv_len, B, Sq, Skv, H, hdp, float(scale), bool(causal))
    else:
        o = ops.attention_fwd(_c(q), _c(k), _c(v), B, Sq, H, hdp, float(scale), bool(causal), kv_len=kv_len, key_mask=key_mask, Skv=Skv)[0]
    return Fn._unpad_heads(o, H, hd, hdp) if hdp != hd else o


class CrossEntropyFn(torch.autograd.Function):
    """Masked-mean CE over logits [T, V] (labels already shifted, -100 = ignore; modeling_aria.py:301-323) for heads whose logits come
    out of a module (an adapted lm_head): the CE kernel leaves d(loss)/d(logits) in a copy during the forward."""

    @staticmethod
    def forward(ctx, logits, labels_shifted):
        count_in = (labels_shifted >= 0).sum(dtype=torch.int32).reshape(1)
        dlogits = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        loss_sum, _, _ = ops.cross_entropy(logits, labels_shifted, grad_scale=1.0, dlogits=dlogits, count_in=count_in)
        ctx.dlogits = dlogits
        return (loss_sum / count_in.clamp(min=1).to(torch.float32)).reshape(())

    @staticmethod
    def backward(ctx, dloss):
        d, ctx.dlogits = ctx.dlogits, None
        return (ops.scale_(d, dloss) if d is not None else None), None
