"""The reference-side seams of SURVEY section 8(b), as code a maintainer can call instead of hand-editing the reference:

* B1 ``install_experts_gemm(moe_lm)``: rebinds the module-level function pointer ``aria/model/moe_lm.py:431-443`` (``experts_gemm``, picked
  at import between ``grouped_gemm.ops.gmm`` and ``sequential_gemm``) to the HIP grouped GEMM.  ``GroupedGEMM.forward`` (:467-484) and
  ``GroupedGemmLoraLayer.forward`` (``aria/lora/layers.py:115-139``) call through that name, so the reference's own ``MoELayer`` /
  ``AriaMoELMForCausalLM`` then run their expert GEMMs (forward, dgrad, wgrad) in ``libaria_hip.so`` with nothing else changed.
* B2 ``register_attention(name)``: the attention registry.  transformers 4.46 (the reference's pin) looks a *class* up in
  ``LLAMA_ATTENTION_CLASSES[config._attn_implementation]`` (``moe_lm.py:594``) -> ``HFAriaAttention`` (``aria_amd.moe_lm.AriaAttention``
  behind that release's ``LlamaAttention.forward`` signature) is registered there; transformers >= 4.48 looks a *function* up in
  ``ALL_ATTENTION_FUNCTIONS`` (``modeling_llama.py:264-266``) -> :func:`attention_interface` (q,k,v already projected and rotated by
  ``LlamaAttention.forward``) is registered under ``name``.

Both need bf16 tensors on the device the library drives; there is no fallback -- a missing ``libaria_hip.so`` raises at the first call.
"""
from __future__ import annotations

from typing import Optional

import torch

from .autograd import _c, experts_gemm, sdpa

bf16 = torch.bfloat16


def install_experts_gemm(moe_lm_module) -> None:
    """``moe_lm.experts_gemm = <HIP grouped GEMM>`` (same call contract: input [M,K] grouped by expert, weight [E,K,N], tokens_per_expert [E]
    on any device; differentiable in input and weight)."""
    moe_lm_module.experts_gemm = experts_gemm


def attention_interface(module, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, attention_mask: Optional[torch.Tensor],
                        dropout: float = 0.0, scaling: Optional[float] = None, **kwargs):
    """``ALL_ATTENTION_FUNCTIONS`` entry (transformers >= 4.48; same contract as ``sdpa_attention_forward``): query [B,H,Sq,hd],
    key/value [B,Hkv,Skv,hd] -> (attn_output [B,Sq,H,hd], None).  ``attention_mask`` is what the mask interface registered next to it
    (``flash_attention_mask``) produces: ``None`` or the 2-D padding mask [B,Skv] (1 = attend).  Causal when the module says so and Sq == Skv (prefill / training);
    a single new query against a cache (Sq == 1) needs no causal mask."""
    if dropout:
        raise NotImplementedError("aria_hip attention: attention dropout is not implemented (Aria trains with attention_dropout = 0)")
    B, H, Sq, hd = query.shape
    Hkv, Skv = key.shape[1], key.shape[2]
    if Hkv != H:  # grouped-query models: expand like repeat_kv (Aria itself is MHA)
        if H % Hkv:
            raise ValueError(f"aria_hip attention: {H} query heads over {Hkv} kv heads")
        key = key.repeat_interleave(H // Hkv, dim=1)
        value = value.repeat_interleave(H // Hkv, dim=1)
    causal = bool(getattr(module, "is_causal", True)) and Sq > 1
    if causal and Sq != Skv:
        raise NotImplementedError("aria_hip attention: chunked prefill against a non-empty cache (1 < Sq < Skv) is not implemented")
    key_mask = None
    if attention_mask is not None:
        if attention_mask.dim() != 2 or attention_mask.shape != (B, Skv):
            raise NotImplementedError(f"aria_hip attention: expected the 2-D padding mask [B,Skv], got {tuple(attention_mask.shape)}")
        key_mask = (attention_mask != 0).to(torch.uint8).contiguous()
    if scaling is None:
        scaling = hd ** -0.5

    def tok(t, S):  # [B,H,S,hd] -> token-major [B*S, H*hd]
        return t.transpose(1, 2).reshape(B * S, H * hd).to(bf16)

    o = sdpa(tok(query, Sq), tok(key, Skv), tok(value, Skv), B, Sq, Skv, H, hd, scaling, causal, key_mask)
    return o.view(B, Sq, H, hd).to(query.dtype), None


def _kv_len_from_mask(attention_mask: Optional[torch.Tensor], B: int, S: int) -> Optional[torch.Tensor]:
    """Valid-key counts int32 [B] from what LlamaModel hands its attention (transformers 4.46 ``_update_causal_mask``): ``None``, the 2-D
    padding mask [B,S] (1 = attend) or the 4-D additive mask [B,1,S,S] (0 = attend) whose last query row sees every valid key.  The kernel's
    kv_len form needs right padding -- the reference's collate pads on the right (``aria/data.py:108-118``)."""
    if attention_mask is None:
        return None
    if attention_mask.dim() == 2:
        valid = attention_mask != 0
    elif attention_mask.dim() == 4:
        valid = attention_mask[:, 0, -1, :S] == 0
    else:
        raise NotImplementedError(f"aria_hip attention: attention_mask of rank {attention_mask.dim()}")
    n = valid.sum(dim=1)
    if not bool((valid == (torch.arange(S, device=valid.device)[None, :] < n[:, None])).all()):
        raise NotImplementedError("aria_hip attention: only right-padded batches are supported")
    return n.to(torch.int32)


def hf_attention_class():
    from . import autograd as AG
    from .moe_lm import AriaAttention

    class HFAriaAttention(AriaAttention):
        """``LLAMA_ATTENTION_CLASSES`` entry with the transformers-4.46 ``LlamaAttention.forward`` signature (``modeling_llama.py:243-281`` of
        that release; called from ``LlamaDecoderLayer.forward`` with keywords): q/k/v projection, half-split RoPE, causal flash attention and
        the output projection run as one autograd node.  Training / prefill only -- generation with a KV cache goes through the gptfast
        surface (seam B4)."""

        def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, output_attentions=False,
                    use_cache=False, cache_position=None, position_embeddings=None, **kwargs):
            if output_attentions:
                raise NotImplementedError("aria_hip attention never materialises the S x S weights")
            if past_key_value is not None and use_cache:
                raise NotImplementedError("aria_hip attention class: KV-cache decoding is served by aria_amd.gptfast (seam B4)")
            if position_embeddings is None:
                raise ValueError("aria_hip attention needs position_embeddings=(cos, sin) from the model-level rotary embedding")
            B, S, D = hidden_states.shape
            cos, sin = position_embeddings
            if position_ids is not None and not bool((position_ids == torch.arange(S, device=position_ids.device)[None, :]).all()):
                raise NotImplementedError("aria_hip attention: position_ids other than arange(S)")
            cos, sin = (t[0] if t.dim() == 3 else t for t in (cos, sin))
            cos, sin = cos.to(bf16).contiguous(), sin.to(bf16).contiguous()
            x = hidden_states.reshape(B * S, D)
            out = AG.AttnBlockFn.apply(_c(x), self.q_proj.weight, self.k_proj.weight, self.v_proj.weight, self.o_proj.weight, cos, sin,
                                       B, S, self.attn_config(), _kv_len_from_mask(attention_mask, B, S))
            return out.view(B, S, D), None, past_key_value

    return HFAriaAttention


def register_attention(name: str = "aria_hip") -> str:
    """Make ``attn_implementation=name`` select the HIP attention in whichever registry the installed transformers has; returns ``name``."""
    import transformers.models.llama.modeling_llama as ml

    try:
        from transformers import AttentionInterface
    except ImportError:  # transformers <= 4.47 (the reference's pin): a class per implementation name, moe_lm.py:594
        ml.LLAMA_ATTENTION_CLASSES[name] = hf_attention_class()
        return name
    AttentionInterface.register(name, attention_interface)
    # the mask the function is handed: without an entry here transformers DROPS the padding mask for unknown implementation names
    # (masking_utils._preprocess_mask_arguments), which only right-padded causal batches survive; the FA2 form is the 2-D mask or None
    from transformers.masking_utils import AttentionMaskInterface, flash_attention_mask

    AttentionMaskInterface.register(name, flash_attention_mask)
    classes = getattr(ml, "LLAMA_ATTENTION_CLASSES", None)
    if isinstance(classes, dict):  # a compatibility dict someone re-created for moe_lm.py:31: the stock class dispatches on the name
        classes[name] = ml.LlamaAttention
    return name
