"""CPU oracle for the Aria hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (CPU, any float dtype; fp32 by default) restatement of the
reference algorithm for the path BASELINE.json's ``north_star`` names.  Each
function cites the reference ``file:line`` it follows (paths relative to
``/root/reference``; ``transformers/...`` = the arithmetic the reference
inherits from ``transformers==4.46.3``, restated from the installed copy).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module, and only as the checker.  Nothing
under ``aria_amd/`` imports it; the product path raises if the HIP library is
missing instead of falling back to anything here.

Pinning: the reference ships no golden vectors for this path,
so the oracle is pinned against outputs of the reference itself:
``oracle/make_golden.py`` imports the reference's own modules from
``/root/reference`` (through ``oracle/ref_shims.py``) and writes
``tests/golden/*.pt``; ``tests/test_oracle_golden.py`` checks every function
below against those fixtures, and ``tests/test_oracle_vs_reference.py`` checks
them against the live reference on fresh random inputs whenever
``/root/reference`` is present.

The language-model functions follow the DEVICE of their inputs.  On the CPU they are the oracle.  The two long-sequence parity cases
(one decoder layer at 65 536 tokens, the 53 248-token prefill: 17 + 5 minutes of host fp32) evaluate this same fp32 code with torch's own
fp32 kernels on the GPU -- never the product library -- and only after the same test has pinned device-fp32 == host-fp32 on a
full-width layer at T = 4096 (``tests/fullwidth_cases.py::oracle_device_pin``), so the device does not vouch for itself.

Weights are passed as a flat ``dict[str, Tensor]`` using the reference's own
state-dict key names, so the same dict drives the
reference, this oracle and ``aria_amd``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- config
@dataclass
class LMConfig:
    """Subset of AriaMoELMConfig (aria/model/moe_lm.py:43-80) + LlamaConfig fields used."""

    hidden_size: int = 2560
    num_hidden_layers: int = 28
    num_attention_heads: int = 20
    num_key_value_heads: int = 20
    vocab_size: int = 100352
    moe_intermediate_size: int = 1664
    moe_num_experts: int = 64
    moe_topk: int = 6
    moe_num_shared_experts: int = 2
    moe_z_loss_coeff: float = 1e-5
    moe_aux_loss_coeff: float = 1e-3
    rms_norm_eps: float = 1e-6
    rope_theta: float = 5_000_000.0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


@dataclass
class VisionConfig:
    """AriaVisionConfig (aria/model/vision_encoder.py:31-40) = SiglipVisionConfig fields."""

    hidden_size: int = 1152
    num_hidden_layers: int = 27
    num_attention_heads: int = 16
    intermediate_size: int = 4304
    patch_size: int = 14
    image_size: int = 980
    num_channels: int = 3
    layer_norm_eps: float = 1e-6


@dataclass
class AriaOracleConfig:
    text: LMConfig = field(default_factory=LMConfig)
    vision: VisionConfig = field(default_factory=VisionConfig)
    patch_to_query: Dict[int, int] = field(default_factory=lambda: {1225: 128, 4900: 256})
    projector_heads: int = 16
    image_token_index: int = 9


# --------------------------------------------------------------------------- router
def router_gating(x: Tensor, weight: Tensor) -> Tensor:
    """logits = x @ W^T, output in x's dtype.  aria/model/moe_lm.py:190-201."""
    return F.linear(x, weight)


def topk_lowest_index(logits: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """torch.topk(logits, k, dim=1) (moe_lm.py:261) with a *defined* tie rule.

    torch.topk's order among equal values is unspecified.  The parity
    protocol canonicalises ties to the lowest expert id; a
    stable descending sort gives exactly that, and equals torch.topk whenever
    the k-th and (k+1)-th logits differ.
    """
    order = torch.sort(logits, dim=1, descending=True, stable=True).indices[:, :k]
    return torch.gather(logits, 1, order), order


_FORCED_ROUTING: Optional[list] = None  # see forced_routing()


class forced_routing:
    """Test protocol, NOT reference behaviour: inside ``with forced_routing([idx_layer0, idx_layer1, ...])`` the
    i-th router call takes its top-k expert ids from the list (the ids the device kernel chose) instead of from its own top-k; scores,
    tokens_per_expert and every gradient follow from the oracle's OWN logits at those ids.  Used at Aria width, where ~2 % of the
    tokens have a k-th / (k+1)-th logit gap below the bf16 rounding of the logits: the router is checked separately
    (ids equal wherever the gap is resolvable), and the arithmetic downstream of it is compared on identical routing so that one
    flipped token does not masquerade as a 10 % error of an expert's weight gradient."""

    def __init__(self, indices):
        self.indices = list(indices)

    def __enter__(self):
        global _FORCED_ROUTING
        self._prev, _FORCED_ROUTING = _FORCED_ROUTING, list(self.indices)
        return self

    def __exit__(self, *exc):
        global _FORCED_ROUTING
        left, _FORCED_ROUTING = _FORCED_ROUTING, self._prev
        if exc[0] is None and left:
            raise AssertionError(f"forced_routing: {len(left)} index tensors were never consumed (fewer router calls than expected)")
        return False


def router_routing(logits: Tensor, topk: int, num_experts: int) -> Tuple[Tensor, Tensor, Tensor]:
    """scores, top_indices, tokens_per_expert.  moe_lm.py:243-293 (eval branch)."""
    if _FORCED_ROUTING is not None:
        if not _FORCED_ROUTING:  # a router call the caller supplied no ids for: never fall back to the oracle's own top-k silently
            raise AssertionError("forced_routing: more router calls than index tensors supplied")
        top_indices = _FORCED_ROUTING.pop(0).to(device=logits.device, dtype=torch.int64).reshape(logits.shape[0], topk)
        top_logits = torch.gather(logits, 1, top_indices)
    else:
        top_logits, top_indices = topk_lowest_index(logits, topk)
    scores = torch.softmax(top_logits, dim=-1, dtype=torch.float32).type_as(logits)  # :262
    tokens_per_expert = torch.bincount(top_indices.flatten(), minlength=num_experts)  # histc :264-269
    return scores, top_indices, tokens_per_expert


def z_loss_func(logits: Tensor, coeff: float) -> Tensor:
    """moe_lm.py:128-140."""
    return torch.mean(torch.square(torch.logsumexp(logits, dim=-1))) * coeff


def switch_load_balancing_loss_func(probs: Tensor, tokens_per_expert: Tensor, topk: int, coeff: float) -> Tensor:
    """moe_lm.py:143-166."""
    num_tokens = probs.shape[0] * topk
    num_experts = probs.shape[1]
    return torch.sum(probs.mean(dim=0) * tokens_per_expert) * (num_experts / num_tokens * coeff)


class _AuxLossScaler(torch.autograd.Function):
    """MoEAuxLossAutoScaler (moe_lm.py:84-125): identity on `output`; in backward the
    aux loss receives gradient ``scale`` (set by train.py:229 to 1/grad_accum)."""

    scale = 1.0

    @staticmethod
    def forward(ctx, output, aux_loss):
        ctx.save_for_backward(aux_loss)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        (aux_loss,) = ctx.saved_tensors
        return grad_output, torch.ones_like(aux_loss) * _AuxLossScaler.scale


def router_forward(x: Tensor, weight: Tensor, cfg: LMConfig, training: bool = False):
    """TopKRouter.forward (moe_lm.py:275-293) incl. the training-only aux losses (:243-273)."""
    logits = router_gating(x, weight).view(-1, cfg.moe_num_experts)
    if training:
        logits = _AuxLossScaler.apply(logits, z_loss_func(logits, cfg.moe_z_loss_coeff))  # :203-215
    scores, idx, tpe = router_routing(logits, cfg.moe_topk, cfg.moe_num_experts)
    if training:
        probs = torch.softmax(logits, dim=-1, dtype=torch.float32)  # :234
        aux = switch_load_balancing_loss_func(probs, tpe, cfg.moe_topk, cfg.moe_aux_loss_coeff)
        scores = _AuxLossScaler.apply(scores, aux)  # :241
    return scores, idx, tpe, logits


# --------------------------------------------------------------------------- dispatcher
def token_permutation(hidden: Tensor, indices: Tensor, topk: int) -> Tuple[Tensor, Tensor]:
    """moe_lm.py:313-334: stable argsort of the flattened expert ids, gather rows."""
    hidden = hidden.reshape(-1, hidden.size(-1))
    sorted_indices = torch.argsort(indices.flatten(), stable=True)
    return hidden.index_select(0, sorted_indices // topk), sorted_indices


def token_unpermutation(expert_out: Tensor, scores: Tensor, sorted_indices: Tensor, topk: int, out_shape) -> Tensor:
    """moe_lm.py:336-365: scatter back, weight by scores (in scores' dtype), sum over k."""
    buf = torch.zeros((scores.numel(), expert_out.size(1)), dtype=expert_out.dtype, device=expert_out.device)
    buf.index_copy_(0, sorted_indices, expert_out)
    buf = buf.reshape(-1, topk, expert_out.size(1))
    buf = buf * scores.unsqueeze(-1)
    return buf.sum(dim=1).type_as(expert_out).view(out_shape)


def _sequential_gemm_loop(inp: Tensor, weight: Tensor, tokens_per_expert: Tensor) -> Tensor:
    out = torch.zeros(inp.shape[0], weight.shape[-1], dtype=inp.dtype, device=inp.device)
    start = 0
    counts = [int(n) for n in tokens_per_expert.tolist()]   # one host read, not one per expert (the counts may live on a device)
    for e in range(weight.shape[0]):
        n = counts[e]
        if n:
            out[start : start + n] = inp[start : start + n] @ weight[e]
        start += n
    return out


class _SequentialGemm(torch.autograd.Function):
    """The loop above with its derivative written out (d inp[rows of e] = d out[rows of e] @ weight[e]^T, d weight[e] =
    inp[rows of e]^T @ d out[rows of e]) -- what autograd computes for the loop, without the full-size zero tensor it allocates per
    ``weight[e]`` select (64 x 3.5 GB at Aria width: 50 s per layer).  tests/test_oracle_golden.py checks it against plain autograd."""

    @staticmethod
    def forward(ctx, inp, weight, tokens_per_expert):
        ctx.save_for_backward(inp, weight)
        ctx.tpe = [int(n) for n in tokens_per_expert.tolist()]
        return _sequential_gemm_loop(inp, weight, tokens_per_expert)

    @staticmethod
    def backward(ctx, gout):
        inp, weight = ctx.saved_tensors
        ginp = torch.zeros_like(inp) if ctx.needs_input_grad[0] else None
        gw = torch.zeros_like(weight) if ctx.needs_input_grad[1] else None
        start = 0
        for e, n in enumerate(ctx.tpe):
            if n:
                if ginp is not None:
                    ginp[start : start + n] = gout[start : start + n] @ weight[e].t()
                if gw is not None:
                    torch.mm(inp[start : start + n].t(), gout[start : start + n], out=gw[e])
            start += n
        return ginp, gw, None


def sequential_gemm(inp: Tensor, weight: Tensor, tokens_per_expert: Tensor) -> Tensor:
    """moe_lm.py:398-428 -- the semantics of seam B1 ``experts_gemm(input, weight, tokens_per_expert)``:
    out[s_e:s_e+n_e] = inp[s_e:s_e+n_e] @ weight[e]."""
    if torch.is_grad_enabled() and (inp.requires_grad or weight.requires_grad):
        return _SequentialGemm.apply(inp, weight, tokens_per_expert)
    return _sequential_gemm_loop(inp, weight, tokens_per_expert)


def glu(x: Tensor) -> Tensor:
    """moe_lm.py:505-507: silu(first half) * second half."""
    a, b = torch.chunk(x, 2, dim=-1)
    return F.silu(a) * b


def lora_grouped_gemm(x: Tensor, base: Tensor, lora_a: Tensor, lora_b: Tensor, tpe: Tensor, scaling: float) -> Tensor:
    """GroupedGemmLoraLayer.forward, aria/lora/layers.py:129-139 (no dropout, no DoRA):
    result = base_layer(x, tpe) + lora_B(lora_A(x, tpe), tpe) * scaling, with lora_A = GroupedGEMM(in, r) [E,in,r] and
    lora_B = GroupedGEMM(r, out) [E,r,out] (layers.py:88-93), scaling = lora_alpha / r (:94)."""
    return sequential_gemm(x, base, tpe) + sequential_gemm(sequential_gemm(x, lora_a, tpe), lora_b, tpe) * scaling


def lora_delta_weight(lora_a: Tensor, lora_b: Tensor, scaling: float) -> Tensor:
    """get_delta_weight, aria/lora/layers.py:196-224: matmul(A, B) * scaling per expert (what merge() adds to the base weight)."""
    return torch.matmul(lora_a, lora_b) * scaling


def lora_linear(x: Tensor, base_w: Tensor, lora_a: Tensor, lora_b: Tensor, scaling: float) -> Tensor:
    """peft's stock LoRA ``Linear.forward`` (peft 0.x ``tuners/lora/layer.py``: ``result = base(x) + lora_B(lora_A(dropout(x))) * scaling``;
    the adapter the reference's recipe puts on q/k/v/o_proj, the shared experts and lm_head -- recipes/config_lora.yaml:47-59 through
    ``get_peft_model``, aria/train.py:100-112).  peft is absent from this image: restated from the published algorithm, dropout = 0.
    base_w [out, in], lora_a [r, in], lora_b [out, r]."""
    return x @ base_w.t() + (x @ lora_a.t()) @ lora_b.t() * scaling


def lora_linear_delta_weight(lora_a: Tensor, lora_b: Tensor, scaling: float) -> Tensor:
    """peft ``Linear.get_delta_weight``: B @ A * scaling, [out, in]."""
    return lora_b @ lora_a * scaling


def grouped_mlp(permuted: Tensor, fc1: Tensor, fc2: Tensor, tpe: Tensor) -> Tensor:
    """GroupedMLP.forward moe_lm.py:511-525."""
    return sequential_gemm(glu(sequential_gemm(permuted, fc1, tpe)), fc2, tpe)


def shared_mlp(x: Tensor, gate: Tensor, up: Tensor, down: Tensor) -> Tensor:
    """SharedExpertMLP = LlamaMLP with I = moe_intermediate*num_shared (moe_lm.py:368-395;
    transformers/models/llama/modeling_llama.py LlamaMLP.forward): down(silu(gate x) * up x)."""
    return F.linear(F.silu(F.linear(x, gate)) * F.linear(x, up), down)


def moe_layer(x: Tensor, w: Dict[str, Tensor], prefix: str, cfg: LMConfig, training: bool = False,
              return_intermediates: bool = False):
    """MoELayer.forward moe_lm.py:548-577."""
    scores, idx, tpe, logits = router_forward(x.reshape(-1, x.size(-1)), w[prefix + "router.weight"], cfg, training)
    permuted, sorted_idx = token_permutation(x, idx, cfg.moe_topk)
    fc1_out = sequential_gemm(permuted, w[prefix + "experts.fc1.weight"], tpe)
    act = glu(fc1_out)
    fc2_out = sequential_gemm(act, w[prefix + "experts.fc2.weight"], tpe)
    out = token_unpermutation(fc2_out, scores, sorted_idx, cfg.moe_topk, x.shape)
    shared = shared_mlp(x, w[prefix + "shared_experts.gate_proj.weight"],
                        w[prefix + "shared_experts.up_proj.weight"],
                        w[prefix + "shared_experts.down_proj.weight"])
    out = out + shared
    if return_intermediates:
        return out, dict(logits=logits, scores=scores, indices=idx, tokens_per_expert=tpe,
                         permuted=permuted, sorted_indices=sorted_idx, fc1_out=fc1_out, act=act,
                         fc2_out=fc2_out, shared=shared)
    return out


# --------------------------------------------------------------------------- norm / rope / attention
def rms_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """LlamaRMSNorm.forward transformers/models/llama/modeling_llama.py:62-67 (== gptfast/model.py:461-472):
    fp32 statistics, cast back to input dtype, THEN multiply by weight."""
    dt = x.dtype
    xf = x.to(torch.float32)
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return weight * xf.to(dt)


def rope_cos_sin(position_ids: Tensor, head_dim: int, theta: float, dtype: torch.dtype) -> Tuple[Tensor, Tensor]:
    """LlamaRotaryEmbedding.forward (transformers/.../modeling_llama.py:96-127): fp32 angles,
    emb = cat(freqs, freqs), cos/sin cast to the activation dtype.  [B, S, head_dim]."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64, device=position_ids.device).float() / head_dim))
    freqs = position_ids[:, :, None].float() * inv_freq[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x: Tensor) -> Tensor:
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2 :]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope_half(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor) -> Tuple[Tensor, Tensor]:
    """apply_rotary_pos_emb (transformers/.../modeling_llama.py:130-160), q,k: [B,H,S,hd]."""
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


def attention_eager(q: Tensor, k: Tensor, v: Tensor, scale: float, causal: bool,
                    key_padding: Optional[Tensor] = None) -> Tensor:
    """eager_attention_forward (transformers/.../modeling_llama.py:192-215; idefics2 same):
    softmax in fp32 then cast to q dtype.  q,k,v [B,H,S,hd]; key_padding [B,Skv] True = masked."""
    att = (q @ k.transpose(2, 3)) * scale
    sq, sk = q.shape[2], k.shape[2]
    if causal:
        m = torch.full((sq, sk), torch.finfo(att.dtype).min, dtype=att.dtype, device=att.device).triu(1 + sk - sq)
        att = att + m
    if key_padding is not None:
        att = att + key_padding[:, None, None, :].to(att.dtype) * torch.finfo(att.dtype).min
    att = torch.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
    return att @ v


class _StreamedCausalAttention(torch.autograd.Function):
    """``attention_eager(q, k, v, scale, causal=True)`` evaluated one (batch, head, query block) at a time, with its derivative written out,
    so that config #4 / the north_star's 64K-token sequences fit on a host: the eager form keeps an [H, S, S] fp32 score tensor (344 GB at
    S = 65 536, H = 20) plus autograd's copies.  Same arithmetic per row -- scores, fp32 softmax over the row's visible keys, P V; a masked
    entry contributes exp(finfo.min - max) = exactly 0 in the eager form and is simply not visited here -- so the two agree to fp32
    summation order (pinned against attention_eager + autograd in tests/test_oracle_golden.py).
    Backward per block (the closed form of softmax's derivative): dV += P^T dO, dP = dO V^T, dS = P * (dP - rowsum(dO * O)) * scale,
    dQ = dS K, dK += dS^T Q."""

    @staticmethod
    def forward(ctx, q, k, v, scale, block):
        B, H, S, hd = q.shape
        o = torch.empty_like(q)
        tri = torch.ones(min(block, S), min(block, S), dtype=torch.bool, device=q.device).triu(1)  # key j > query i inside the diagonal block
        for b in range(B):
            for h in range(H):
                for q0 in range(0, S, block):
                    q1 = min(S, q0 + block)
                    att = (q[b, h, q0:q1] @ k[b, h, :q1].t()).mul_(scale)
                    att[:, q0:q1].masked_fill_(tri[: q1 - q0, : q1 - q0], float("-inf"))  # keys before q0 are visible to every row of the block
                    o[b, h, q0:q1] = torch.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype) @ v[b, h, :q1]
        ctx.save_for_backward(q, k, v, o)
        ctx.scale, ctx.block = scale, block
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o = ctx.saved_tensors
        scale, block = ctx.scale, ctx.block
        B, H, S, hd = q.shape
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        tri = torch.ones(min(block, S), min(block, S), dtype=torch.bool, device=q.device).triu(1)
        for b in range(B):
            for h in range(H):
                for q0 in range(0, S, block):
                    q1 = min(S, q0 + block)
                    qb, dob = q[b, h, q0:q1], do[b, h, q0:q1]
                    att = (qb @ k[b, h, :q1].t()).mul_(scale)
                    att[:, q0:q1].masked_fill_(tri[: q1 - q0, : q1 - q0], float("-inf"))
                    p = torch.softmax(att, dim=-1, dtype=torch.float32).to(q.dtype)
                    del att
                    dv[b, h, :q1] += p.t() @ dob
                    dp = dob @ v[b, h, :q1].t()
                    delta = (dob * o[b, h, q0:q1]).sum(dim=-1, keepdim=True)
                    ds = dp.sub_(delta).mul_(p).mul_(scale)
                    del p, dp
                    dq[b, h, q0:q1] = ds @ k[b, h, :q1]
                    dk[b, h, :q1] += ds.t() @ qb
        return dq, dk, dv, None, None


_STREAM_BLOCK: Optional[int] = None


class streamed_attention:
    """Context: ``llama_attention`` evaluates its causal, unpadded attention block-wise (``_StreamedCausalAttention``, query blocks of
    ``block`` rows) instead of through the [H, S, S] eager tensor.  Test infrastructure for the long-sequence parity cases."""

    def __init__(self, block: int = 4096):
        self.block = block

    def __enter__(self):
        global _STREAM_BLOCK
        self._prev, _STREAM_BLOCK = _STREAM_BLOCK, self.block
        return self

    def __exit__(self, *exc):
        global _STREAM_BLOCK
        _STREAM_BLOCK = self._prev
        return False


def attention_causal_streamed(q: Tensor, k: Tensor, v: Tensor, scale: float, block: int = 4096) -> Tensor:
    return _StreamedCausalAttention.apply(q, k, v, scale, block)


def llama_attention(x: Tensor, w: Dict[str, Tensor], prefix: str, cfg: LMConfig, position_ids: Tensor,
                    attention_mask: Optional[Tensor] = None) -> Tensor:
    """LlamaAttention.forward transformers/.../modeling_llama.py:243-281 (no cache). MHA/GQA."""
    B, S, D = x.shape
    H, Hkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    q = F.linear(x, w[prefix + "q_proj.weight"]).view(B, S, H, hd).transpose(1, 2)
    k = F.linear(x, w[prefix + "k_proj.weight"]).view(B, S, Hkv, hd).transpose(1, 2)
    v = F.linear(x, w[prefix + "v_proj.weight"]).view(B, S, Hkv, hd).transpose(1, 2)
    cos, sin = rope_cos_sin(position_ids, hd, cfg.rope_theta, x.dtype)
    q, k = apply_rope_half(q, k, cos, sin)
    if Hkv != H:
        k = k.repeat_interleave(H // Hkv, dim=1)
        v = v.repeat_interleave(H // Hkv, dim=1)
    pad = None if attention_mask is None else (attention_mask == 0)
    if _STREAM_BLOCK is not None and pad is None:
        o = attention_causal_streamed(q, k, v, hd ** -0.5, _STREAM_BLOCK)
    else:
        o = attention_eager(q, k, v, hd ** -0.5, causal=True, key_padding=pad)
    o = o.transpose(1, 2).reshape(B, S, H * hd)
    return F.linear(o, w[prefix + "o_proj.weight"])


def decoder_layer(x: Tensor, w: Dict[str, Tensor], prefix: str, cfg: LMConfig, position_ids: Tensor,
                  attention_mask: Optional[Tensor] = None, training: bool = False) -> Tensor:
    """MoEDecoderLayer (moe_lm.py:580-602) with LlamaDecoderLayer.forward
    (transformers/.../modeling_llama.py:295-325)."""
    h = x + llama_attention(rms_norm(x, w[prefix + "input_layernorm.weight"], cfg.rms_norm_eps), w,
                            prefix + "self_attn.", cfg, position_ids, attention_mask)
    return h + moe_layer(rms_norm(h, w[prefix + "post_attention_layernorm.weight"], cfg.rms_norm_eps), w,
                         prefix + "mlp.", cfg, training)


def lm_forward(inputs_embeds: Tensor, w: Dict[str, Tensor], cfg: LMConfig, prefix: str = "",
               attention_mask: Optional[Tensor] = None, training: bool = False,
               return_hidden: bool = False) -> Tensor:
    """AriaMoELMForCausalLM forward on embeddings: 28x layer, final RMSNorm, lm_head
    (moe_lm.py:605-661; LlamaModel/LlamaForCausalLM.forward)."""
    B, S, _ = inputs_embeds.shape
    position_ids = torch.arange(S, device=inputs_embeds.device)[None, :].expand(B, S)
    h = inputs_embeds
    for i in range(cfg.num_hidden_layers):
        h = decoder_layer(h, w, f"{prefix}model.layers.{i}.", cfg, position_ids, attention_mask, training)
    h = rms_norm(h, w[prefix + "model.norm.weight"], cfg.rms_norm_eps)
    if return_hidden:
        return h
    return F.linear(h, w[prefix + "lm_head.weight"])


def causal_lm_loss(logits: Tensor, labels: Tensor, attention_mask: Optional[Tensor]) -> Tensor:
    """aria/model/modeling_aria.py:301-323: shift, filter by attention_mask, mean CE (ignore -100)."""
    if attention_mask is not None:
        sm = attention_mask[:, -(logits.shape[1] - 1):]
        sl = logits[..., :-1, :][sm != 0].contiguous()
        tl = labels[..., 1:][sm != 0].contiguous()
    else:
        sl = logits[..., :-1, :].contiguous()
        tl = labels[..., 1:].contiguous()
    return F.cross_entropy(sl.view(-1, sl.size(-1)), tl.view(-1))


# --------------------------------------------------------------------------- ViT
def vit_patch_mask(pixel_mask: Tensor, patch: int) -> Tensor:
    """aria/model/vision_encoder.py:132-145: a patch is valid if any pixel in it is."""
    sub = pixel_mask.unfold(1, patch, patch).unfold(2, patch, patch)
    return (sub.sum(dim=(-1, -2)) > 0).bool()


def vit_position_ids(patch_mask: Tensor, n_side: int) -> Tensor:
    """Integer restatement of Idefics2VisionEmbeddings' bucketised position ids
    (transformers/models/idefics2/modeling_idefics2.py:141-170, fp32 coordinates):
    id = floor(i*n_side/n_h)*n_side + floor(j*n_side/n_w) on valid patches, 0 on padding.
    (low-precision coordinates corrupt this; the integer form is the fp32 answer.)"""
    B, Hp, Wp = patch_mask.shape
    nh = patch_mask[:, :, 0].sum(dim=1)
    nw = patch_mask[:, 0, :].sum(dim=1)
    ids = torch.zeros(B, Hp * Wp, dtype=torch.long)
    for b in range(B):
        fh = (torch.arange(Hp, dtype=torch.float32) * (1.0 / nh[b].float())).clamp(max=1.0 - 1e-6)
        fw = (torch.arange(Wp, dtype=torch.float32) * (1.0 / nw[b].float())).clamp(max=1.0 - 1e-6)
        bound = torch.arange(1 / n_side, 1.0, 1 / n_side)
        bh = torch.bucketize(fh, bound, right=True)
        bw = torch.bucketize(fw, bound, right=True)
        p = (bh[:, None] * n_side + bw[None, :]).reshape(-1)
        m = patch_mask[b].reshape(-1)
        ids[b][m] = p[m]
    return ids


def vit_position_ids_table(patch_mask: Tensor, n_side: int) -> Tensor:
    """Vectorised restatement of the same fp32 computation, in the form the HIP patch-embed kernel
    uses: frac = fl32(i) * fl32(1/n_valid) clamped to fl32(1-1e-6); id = #(boundaries <= frac), with
    the boundary table built on the host exactly like the reference builds it on CPU
    (torch.arange(1/n, 1.0, 1/n), modeling_idefics2.py:137-139).  NOTE: this is NOT always
    floor(i*n_side/n_valid): fp32 rounding moves ~1% of ids by one bucket, and the reference's
    answer (this one) is the contract."""
    B, Hp, Wp = patch_mask.shape
    bound = torch.arange(1 / n_side, 1.0, 1 / n_side)
    nh = patch_mask[:, :, 0].sum(dim=1)
    nw = patch_mask[:, 0, :].sum(dim=1)
    fh = (torch.arange(Hp, dtype=torch.float32)[None, :] * (1.0 / nh)[:, None]).clamp(max=1.0 - 1e-6)
    fw = (torch.arange(Wp, dtype=torch.float32)[None, :] * (1.0 / nw)[:, None]).clamp(max=1.0 - 1e-6)
    bh = (bound[None, None, :] <= fh[:, :, None]).sum(-1)
    bw = (bound[None, None, :] <= fw[:, :, None]).sum(-1)
    ids = bh[:, :, None] * n_side + bw[:, None, :]
    return (ids * patch_mask).reshape(B, -1)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def vit_embeddings(pixel_values: Tensor, patch_mask: Tensor, w: Dict[str, Tensor], prefix: str,
                   cfg: VisionConfig) -> Tensor:
    """Idefics2VisionEmbeddings.forward transformers/.../modeling_idefics2.py:130-173: Conv2d(k=s=patch)
    == GEMM over 3*p*p patch vectors, + position embedding."""
    x = F.conv2d(pixel_values, w[prefix + "patch_embedding.weight"], w[prefix + "patch_embedding.bias"],
                 stride=cfg.patch_size)
    x = x.flatten(2).transpose(1, 2)
    ids = vit_position_ids(patch_mask.cpu(), cfg.image_size // cfg.patch_size).to(x.device)   # (integer bookkeeping on the host; the
    return x + w[prefix + "position_embedding.weight"][ids]                                    # arithmetic follows the device of its inputs)


def vit_encoder_layer(x: Tensor, key_padding: Optional[Tensor], w: Dict[str, Tensor], prefix: str,
                      cfg: VisionConfig) -> Tensor:
    """Idefics2EncoderLayer transformers/.../modeling_idefics2.py:330-363 (+ attention :203-278,
    MLP gelu_pytorch_tanh): pre-LN, biases everywhere."""
    B, P, D = x.shape
    H = cfg.num_attention_heads
    hd = D // H
    h = layer_norm(x, w[prefix + "layer_norm1.weight"], w[prefix + "layer_norm1.bias"], cfg.layer_norm_eps)
    q = F.linear(h, w[prefix + "self_attn.q_proj.weight"], w[prefix + "self_attn.q_proj.bias"])
    k = F.linear(h, w[prefix + "self_attn.k_proj.weight"], w[prefix + "self_attn.k_proj.bias"])
    v = F.linear(h, w[prefix + "self_attn.v_proj.weight"], w[prefix + "self_attn.v_proj.bias"])
    q, k, v = (t.view(B, P, H, hd).transpose(1, 2) for t in (q, k, v))
    o = attention_eager(q, k, v, hd ** -0.5, causal=False, key_padding=key_padding)
    o = o.transpose(1, 2).reshape(B, P, D)
    x = x + F.linear(o, w[prefix + "self_attn.out_proj.weight"], w[prefix + "self_attn.out_proj.bias"])
    h = layer_norm(x, w[prefix + "layer_norm2.weight"], w[prefix + "layer_norm2.bias"], cfg.layer_norm_eps)
    h = F.linear(h, w[prefix + "mlp.fc1.weight"], w[prefix + "mlp.fc1.bias"])
    h = F.gelu(h, approximate="tanh")
    h = F.linear(h, w[prefix + "mlp.fc2.weight"], w[prefix + "mlp.fc2.bias"])
    return x + h


def vit_forward(pixel_values: Tensor, pixel_mask: Optional[Tensor], w: Dict[str, Tensor], prefix: str,
                cfg: VisionConfig) -> Tuple[Tensor, Optional[Tensor]]:
    """AriaVisionModel.forward aria/model/vision_encoder.py:94-130: returns (last_hidden_state,
    image_atts) with image_atts True = padded patch; post-layernorm is Identity (:58-67)."""
    B = pixel_values.shape[0]
    n = pixel_values.shape[2] // cfg.patch_size
    if pixel_mask is None:
        patch_mask = torch.ones(B, n, pixel_values.shape[3] // cfg.patch_size, dtype=torch.bool, device=pixel_values.device)
        key_padding = None
    else:
        patch_mask = vit_patch_mask(pixel_mask, cfg.patch_size)
        key_padding = ~patch_mask.flatten(1)
        if not key_padding.any():
            key_padding_attn = None  # Idefics2VisionTransformer drops an all-ones mask
        else:
            key_padding_attn = key_padding
    x = vit_embeddings(pixel_values, patch_mask, w, prefix + "vision_model.embeddings.", cfg)
    kp = None if pixel_mask is None else key_padding_attn
    for i in range(cfg.num_hidden_layers):
        x = vit_encoder_layer(x, kp, w, f"{prefix}vision_model.encoder.layers.{i}.", cfg)
    return x, (None if pixel_mask is None else key_padding)


# --------------------------------------------------------------------------- projector
def gelu_new(x: Tensor) -> Tensor:
    """transformers.activations NewGELUActivation (ACT2FN['gelu_new'], projector.py:40)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def projector_forward(x: Tensor, attn_mask: Optional[Tensor], w: Dict[str, Tensor], prefix: str,
                      cfg: AriaOracleConfig) -> Tensor:
    """AriaProjector.forward aria/model/projector.py:160-189 with CrossAttention (:73-102) and
    FFN (:42-45).  Note the double projection: q/k/v_proj THEN nn.MultiheadAttention's own
    in_proj/out_proj, then `linear`; no residual."""
    B, P, Dk = x.shape
    Q = cfg.patch_to_query[P]
    E = w[prefix + "query"].shape[1]
    H = cfg.projector_heads
    hd = E // H
    queries = w[prefix + "query"][:Q].unsqueeze(0).expand(B, Q, E)
    c = prefix + "cross_attn."
    qn = layer_norm(queries, w[c + "layer_norm.weight"], w[c + "layer_norm.bias"], 1e-5)
    q = F.linear(qn, w[c + "q_proj.weight"])
    xn = layer_norm(x, w[c + "ln_kv.weight"], w[c + "ln_kv.bias"], 1e-5)
    k = F.linear(xn, w[c + "k_proj.weight"])
    v = F.linear(xn, w[c + "v_proj.weight"])
    wi, bi = w[c + "multihead_attn.in_proj_weight"], w[c + "multihead_attn.in_proj_bias"]
    q = F.linear(q, wi[:E], bi[:E]).view(B, Q, H, hd).transpose(1, 2)
    k = F.linear(k, wi[E:2 * E], bi[E:2 * E]).view(B, P, H, hd).transpose(1, 2)
    v = F.linear(v, wi[2 * E:], bi[2 * E:]).view(B, P, H, hd).transpose(1, 2)
    att = (q * hd ** -0.5) @ k.transpose(2, 3)
    if attn_mask is not None:
        att = att.masked_fill(attn_mask[:, None, None, :], float("-inf"))
    att = torch.softmax(att, dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, Q, E)
    o = F.linear(o, w[c + "multihead_attn.out_proj.weight"], w[c + "multihead_attn.out_proj.bias"])
    o = F.linear(o, w[c + "linear.weight"], w[c + "linear.bias"])
    h = layer_norm(o, w[prefix + "ln_ffn.weight"], w[prefix + "ln_ffn.bias"], 1e-5)
    h = gelu_new(F.linear(h, w[prefix + "ffn.linear_in.weight"]))
    return F.linear(h, w[prefix + "ffn.linear_out.weight"])


# --------------------------------------------------------------------------- full model
def aria_forward(input_ids: Tensor, pixel_values: Optional[Tensor], pixel_mask: Optional[Tensor],
                 attention_mask: Optional[Tensor], labels: Optional[Tensor], w: Dict[str, Tensor],
                 cfg: AriaOracleConfig, training: bool = False):
    """AriaForConditionalGeneration.forward aria/model/modeling_aria.py:194-335."""
    emb = w["language_model.model.embed_tokens.weight"][input_ids]
    if pixel_values is not None:
        feat, atts = vit_forward(pixel_values, pixel_mask, w, "vision_tower.", cfg.vision)
        img = projector_forward(feat, atts, w, "multi_modal_projector.", cfg)
        is_img = input_ids == cfg.image_token_index
        if int(is_img.sum()) != img.shape[0] * img.shape[1]:
            raise ValueError("Image features and image tokens do not match")  # :267-271
        emb = emb.masked_scatter(is_img.unsqueeze(-1).expand_as(emb), img.to(emb.dtype))
    logits = lm_forward(emb, w, cfg.text, "language_model.", attention_mask, training)
    loss = None if labels is None else causal_lm_loss(logits, labels, attention_mask)
    return logits, loss


# --------------------------------------------------------------------------- gptfast wire format
def hf_to_gptfast_llm(w: Dict[str, Tensor], cfg: LMConfig, prefix: str = "language_model.") -> Dict[str, Tensor]:
    """gptfast/scripts/convert_hf_checkpoint.py:90-162: key map, q/k permute for interleaved-pair
    RoPE (:110-116), wqkv fusion (:145-153), fc1 -> w1/w3 split + transposes, fc2 -> w2 (:154-162)."""
    H, hd, D = cfg.num_attention_heads, cfg.head_dim, cfg.hidden_size

    def permute(t, n_head):
        return t.view(n_head, 2, hd // 2, D).transpose(1, 2).reshape(n_head * hd, D)

    out = {
        "tok_embeddings.weight": w[prefix + "model.embed_tokens.weight"],
        "norm.weight": w[prefix + "model.norm.weight"],
        "output.weight": w[prefix + "lm_head.weight"],
    }
    I = cfg.moe_intermediate_size
    for i in range(cfg.num_hidden_layers):
        s, d = f"{prefix}model.layers.{i}.", f"layers.{i}."
        q = permute(w[s + "self_attn.q_proj.weight"], H)
        k = permute(w[s + "self_attn.k_proj.weight"], cfg.num_key_value_heads)
        out[d + "attention.wqkv.weight"] = torch.cat([q, k, w[s + "self_attn.v_proj.weight"]])
        out[d + "attention.wo.weight"] = w[s + "self_attn.o_proj.weight"]
        fc1 = w[s + "mlp.experts.fc1.weight"]
        out[d + "feed_forward.cond_ffn.w1"] = fc1[:, :, :I].transpose(1, 2).contiguous()
        out[d + "feed_forward.cond_ffn.w3"] = fc1[:, :, I:].transpose(1, 2).contiguous()
        out[d + "feed_forward.cond_ffn.w2"] = w[s + "mlp.experts.fc2.weight"].transpose(1, 2).contiguous()
        out[d + "feed_forward.gate.weight"] = w[s + "mlp.router.weight"]
        out[d + "feed_forward.shared_ffn.w1.weight"] = w[s + "mlp.shared_experts.gate_proj.weight"]
        out[d + "feed_forward.shared_ffn.w3.weight"] = w[s + "mlp.shared_experts.up_proj.weight"]
        out[d + "feed_forward.shared_ffn.w2.weight"] = w[s + "mlp.shared_experts.down_proj.weight"]
        out[d + "attention_norm.weight"] = w[s + "input_layernorm.weight"]
        out[d + "ffn_norm.weight"] = w[s + "post_attention_layernorm.weight"]
    return out
