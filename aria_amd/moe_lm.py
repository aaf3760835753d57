"""MI355X-native mirror of the reference's MoE decoder (``aria/model/moe_lm.py``): same class names, constructor
arguments, parameter names / shapes (so reference checkpoints load unchanged: ``model.layers.{i}.mlp.router.weight``,
``...experts.fc1.weight [E, D, 2I]``, ``...experts.fc2.weight [E, I, D]``, ``...shared_experts.{gate,up,down}_proj.weight``,
``self_attn.{q,k,v,o}_proj.weight``, ``input_layernorm / post_attention_layernorm.weight``) and the same seams
(``experts_gemm``, ``GroupedGEMM``), but every forward/backward runs the hand-written HIP kernels of ``libaria_hip.so``.

This is host-side glue only.  There is no PyTorch fallback: on a machine without the library / a GPU the ops raise.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import autograd as AG
from . import functional as Fn
from . import ops
from .autograd import experts_gemm  # noqa: F401  (seam B1, re-exported under the reference's name)

bf16 = torch.bfloat16


class AriaMoELMConfig:
    """Fields of the reference's AriaMoELMConfig (aria/model/moe_lm.py:43-80) + the LlamaConfig fields the path uses.
    Aria-25.3B values are the defaults (gptfast/model.py:39-54)."""

    model_type = "aria_moe_lm"

    def __init__(self, moe_intermediate_size: int = 1664, moe_num_experts: int = 64, moe_topk: int = 6,
                 moe_z_loss_coeff: float = 1e-5, moe_aux_loss_coeff: float = 1e-3, moe_num_shared_experts: int = 2,
                 hidden_size: int = 2560, num_hidden_layers: int = 28, num_attention_heads: int = 20,
                 num_key_value_heads: Optional[int] = None, vocab_size: int = 100352, rms_norm_eps: float = 1e-6,
                 rope_theta: float = 5_000_000.0, max_position_embeddings: int = 65536, pad_token_id: Optional[int] = None,
                 gradient_checkpointing: bool = False, recompute_level: str = "auto", **kwargs):
        self.moe_intermediate_size = moe_intermediate_size
        self.moe_num_experts = moe_num_experts
        self.moe_topk = moe_topk
        self.moe_z_loss_coeff = moe_z_loss_coeff
        self.moe_aux_loss_coeff = moe_aux_loss_coeff
        self.moe_num_shared_experts = moe_num_shared_experts
        self.hidden_size = hidden_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads or num_attention_heads
        self.vocab_size = vocab_size
        self.rms_norm_eps = rms_norm_eps
        self.rope_theta = rope_theta
        self.max_position_embeddings = max_position_embeddings
        self.pad_token_id = pad_token_id
        self.gradient_checkpointing = gradient_checkpointing
        self.recompute_level = recompute_level   # "auto" (by free memory) / "moe" / "layer": autograd.choose_recompute_level
        self.extra = kwargs

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


class MoEAuxLossAutoScaler:
    """Scale of the auxiliary-loss gradients (reference: autograd.Function of the same name, moe_lm.py:84-125;
    ``aria/train.py:229`` sets it to 1/gradient_accumulation_steps).  Here the aux/z-loss gradients are produced
    inside ``aria_moe_route_bwd``; this class only carries the scale."""

    main_loss_backward_scale: float = 1.0

    @staticmethod
    def set_loss_scale(scale) -> None:
        MoEAuxLossAutoScaler.main_loss_backward_scale = float(scale)


def _param(*shape) -> nn.Parameter:
    return nn.Parameter(torch.empty(*shape, dtype=bf16))


class Linear(nn.Module):
    """nn.Linear-shaped parameter holder (weight [out, in], optional bias) whose forward is the MFMA GEMM."""

    def __init__(self, in_features: int, out_features: int, bias: bool = False):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = _param(out_features, in_features)
        self.bias = _param(out_features) if bias else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return AG.linear(x, self.weight, self.bias)


class _PackedWeights(nn.Module):
    """Keeps the weights of the projections that read the same input (``_packed()``: q|k|v, shared gate|up) back to back in one allocation,
    so the fused blocks' [3D, D] / [2I, D] operand is a view of the parameters, not a per-forward copy (functional.py ``fused_weight`` /
    ``pack_adjacent_``).  Packed at construction and again after every ``_apply`` (``.to()``, ``.bfloat16()``, ``.cuda()`` re-allocate each
    parameter on its own); anything that breaks the adjacency later (``load_state_dict(assign=True)``, an adapter wrapping one projection)
    only brings the copy back -- ``fused_weight`` checks the tensors at every call.
    Copies: ``copy.deepcopy`` / pickling re-create every parameter in a storage of its own, so ``__setstate__`` packs the copy
    again (EMA / cloned models keep the view instead of silently going back to the per-forward ``torch.cat``).  A packed parameter is a
    SLICE of the shared buffer: ``torch.save`` of a whole ``state_dict()`` stores that buffer once, but saving ONE such tensor on its own
    serialises the whole buffer behind it -- ``.clone()`` a slice before saving a partial state dict."""

    def _packed(self):
        return ()

    def pack_weights_(self) -> None:
        mods = self._packed()
        if mods and all(type(m) is Linear and isinstance(m.weight, nn.Parameter) for m in mods):
            Fn.pack_adjacent_(*(m.weight for m in mods))

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.pack_weights_()
        return out

    def __setstate__(self, state):   # deepcopy / unpickling: the copies' parameters have lost the adjacency
        super().__setstate__(state)
        self.pack_weights_()


def _plain_linears(*mods) -> bool:
    """True when every module is the stock ``Linear`` (the fused blocks read ``.weight`` directly); False as soon as an adapter
    (aria_amd/lora.py) wraps one of them -- the block then runs module by module so the adapter's forward is what executes."""
    return all(type(m) is Linear for m in mods)


class RMSNorm(nn.Module):
    """LlamaRMSNorm (transformers/models/llama/modeling_llama.py:62-67)."""

    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=bf16))
        self.variance_epsilon = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return AG.rms_norm(x, self.weight, self.variance_epsilon)


class TopKRouter(nn.Module):
    """moe_lm.py:170-293.  forward(input [T, D]) -> (scores [T,k], top_indices int64 [T,k], tokens_per_expert int64 [E]).
    Ties are broken towards the lowest expert id (torch.topk leaves the order unspecified).  The auxiliary losses only
    exist as gradients (they are never added to the reported loss, moe_lm.py:203-241) and are applied by MoELayer."""

    def __init__(self, config: AriaMoELMConfig):
        super().__init__()
        self.config = config
        self.weight = _param(config.moe_num_experts, config.hidden_size)

    def gating(self, input: torch.Tensor) -> torch.Tensor:
        return AG.linear(input, self.weight)

    def routing(self, logits: torch.Tensor):
        scores, idx, counts = ops.moe_route(logits.detach().contiguous(), self.config.moe_topk)
        return scores, idx.long(), counts.long()

    def forward(self, input: torch.Tensor):
        logits = self.gating(input).view(-1, self.config.moe_num_experts)
        return self.routing(logits)


class TokenDispatcher:
    """moe_lm.py:297-365 (stateful, like the reference).  Inference-only convenience; MoELayer uses the fused path."""

    def __init__(self, config: AriaMoELMConfig):
        self.config = config
        self.hidden_states_shape = None
        self.reversed_input_permutation_mapping = None
        self._inv = None

    def token_permutation(self, hidden_states: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
        k, E = self.config.moe_topk, self.config.moe_num_experts
        self.hidden_states_shape = hidden_states.shape
        x = hidden_states.reshape(-1, hidden_states.size(-1)).contiguous()
        idx32 = indices.to(torch.int32).contiguous()
        counts = torch.bincount(idx32.flatten().long(), minlength=E).to(torch.int32)
        _, sorted_src, inv = ops.moe_sort(idx32, counts)
        self.reversed_input_permutation_mapping = sorted_src.long()
        self._inv = inv
        return ops.moe_permute(x, sorted_src, k)

    def token_unpermutation(self, permuted_tokens: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        out = ops.moe_unpermute(permuted_tokens.contiguous(), self._inv, scores.contiguous(), self.config.moe_topk)
        return out.view(self.hidden_states_shape)


class GroupedGEMM(nn.Module):
    """moe_lm.py:446-484: weight [groups, in_features, out_features]; forward(input, tokens_per_expert)."""

    def __init__(self, in_features: int, out_features: int, groups: int):
        super().__init__()
        self.in_features, self.out_features, self.groups = in_features, out_features, groups
        self.weight = _param(groups, in_features, out_features)

    def forward(self, input: torch.Tensor, tokens_per_expert: torch.Tensor) -> torch.Tensor:
        return experts_gemm(input, self.weight, tokens_per_expert)  # no .cpu() sync, no set_device


class GroupedMLP(nn.Module):
    """moe_lm.py:487-525."""

    def __init__(self, config: AriaMoELMConfig):
        super().__init__()
        self.config = config
        self.fc1 = GroupedGEMM(config.hidden_size, config.moe_intermediate_size * 2, config.moe_num_experts)
        self.fc2 = GroupedGEMM(config.moe_intermediate_size, config.hidden_size, config.moe_num_experts)

    def forward(self, permuted_tokens: torch.Tensor, tokens_per_expert: torch.Tensor) -> torch.Tensor:
        h = self.fc1(permuted_tokens, tokens_per_expert)
        return self.fc2(AG.SwiGLUFn.apply(h), tokens_per_expert)


class SharedExpertMLP(_PackedWeights):
    """moe_lm.py:368-395 (LlamaMLP with I = moe_intermediate_size * moe_num_shared_experts, no bias)."""

    def _packed(self):
        return (self.gate_proj, self.up_proj)

    def __init__(self, config: AriaMoELMConfig):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.intermediate_size = config.moe_intermediate_size * config.moe_num_shared_experts
        self.gate_proj = Linear(self.hidden_size, self.intermediate_size)
        self.up_proj = Linear(self.hidden_size, self.intermediate_size)
        self.down_proj = Linear(self.intermediate_size, self.hidden_size)
        self.pack_weights_()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        gu = torch.cat([self.gate_proj(x2), self.up_proj(x2)], dim=-1)
        return self.down_proj(AG.SwiGLUFn.apply(gu)).view(shp)


class MoELayer(nn.Module):
    """moe_lm.py:528-577.  forward(hidden_states [B,S,D] or [T,D]) -> same shape."""

    def __init__(self, config: AriaMoELMConfig):
        super().__init__()
        self.config = config
        self.router = TopKRouter(config)
        self.token_dispatcher = TokenDispatcher(config)
        self.experts = GroupedMLP(config)
        self.shared_experts = SharedExpertMLP(config)
        self.ep_group = None     # set by enable_expert_parallel: the process group the routed experts are sharded over
        self.ep_enabled = False

    def enable_expert_parallel(self, group=None) -> None:
        """Config #5: keep only this rank's E/W routed experts (rank g owns experts [g*E/W, (g+1)*E/W)); from now on forward() dispatches
        rows to their owners with an all-to-all (``aria_amd.expert_parallel.ep_moe_forward``).  The local shards are marked ``_ep_local``:
        their gradients are complete on the owner, so ``GradSync`` does not all-reduce them and ``ShardedAdamW`` keeps their state whole."""
        import torch.distributed as dist

        W, r = dist.get_world_size(group), dist.get_rank(group)
        E = self.config.moe_num_experts
        if E % W:
            raise ValueError(f"{E} experts do not shard over {W} ranks")
        per = E // W
        for fc in (self.experts.fc1, self.experts.fc2):
            if type(fc) is not GroupedGEMM:
                raise NotImplementedError("expert parallelism with an adapter on the expert GEMMs")
            local = nn.Parameter(fc.weight.detach()[r * per:(r + 1) * per].clone(), requires_grad=fc.weight.requires_grad)
            local._ep_local = True
            fc.weight, fc.groups = local, per
        self.ep_group, self.ep_enabled = group, True

    def moe_config(self) -> Fn.MoEConfig:
        c = self.config
        train = self.training
        return Fn.MoEConfig(topk=c.moe_topk, num_experts=c.moe_num_experts,
                            z_loss_coeff=c.moe_z_loss_coeff if train else 0.0,
                            aux_loss_coeff=c.moe_aux_loss_coeff if train else 0.0,
                            aux_scale=MoEAuxLossAutoScaler.main_loss_backward_scale)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        shp = hidden_states.shape
        x = hidden_states.reshape(-1, shp[-1])
        x = x if x.is_contiguous() else x.contiguous()
        se = self.shared_experts
        if self.ep_enabled:
            from .expert_parallel import ep_moe_forward

            return ep_moe_forward(x, self.router.weight, self.experts.fc1.weight, self.experts.fc2.weight, se.gate_proj.weight,
                                  se.up_proj.weight, se.down_proj.weight, self.moe_config(), self.ep_group).view(shp)
        if self.has_adapter():
            return self.forward_modular(x).view(shp)  # an adapter (LoRA) sits on an expert GEMM or a shared-expert projection
        out = AG.MoELayerFn.apply(x, self.router.weight, self.experts.fc1.weight, self.experts.fc2.weight,
                                  se.gate_proj.weight, se.up_proj.weight, se.down_proj.weight, self.moe_config())
        return out.view(shp)

    def has_adapter(self) -> bool:
        e, se = self.experts, self.shared_experts
        return (type(e.fc1) is not GroupedGEMM or type(e.fc2) is not GroupedGEMM
                or not _plain_linears(se.gate_proj, se.up_proj, se.down_proj))

    def forward_modular(self, x: torch.Tensor) -> torch.Tensor:
        """The same layer (moe_lm.py:548-577) built from its differentiable pieces, calling ``self.experts`` / ``self.shared_experts``
        as MODULES -- the seam a ``GroupedGemmLoraLayer`` wraps (aria/lora/layers.py)."""
        from .expert_parallel import PermuteFn, RouteFn, UnpermuteFn

        cfg = self.moe_config()
        logits = AG.linear(x, self.router.weight)
        scores, idx, counts = RouteFn.apply(logits, cfg)
        offsets, sorted_src, inv = ops.moe_sort(idx, counts)
        permuted = PermuteFn.apply(x, sorted_src, inv, cfg.topk)
        expert_out = self.experts(permuted, counts)
        return UnpermuteFn.apply(expert_out, inv, scores, self.shared_experts(x), cfg.topk)


class AriaAttention(_PackedWeights):
    """LlamaAttention's parameter surface (q/k/v/o_proj, no bias; seam B2) with RoPE + flash attention in HIP."""

    def _packed(self):
        return (self.q_proj, self.k_proj, self.v_proj)

    def __init__(self, config: AriaMoELMConfig, layer_idx: int = 0):
        super().__init__()
        self.config, self.layer_idx = config, layer_idx
        D, H, Hkv, hd = config.hidden_size, config.num_attention_heads, config.num_key_value_heads, config.head_dim
        self.q_proj = Linear(D, H * hd)
        self.k_proj = Linear(D, Hkv * hd)
        self.v_proj = Linear(D, Hkv * hd)
        self.o_proj = Linear(H * hd, D)
        self.pack_weights_()

    def attn_config(self) -> Fn.AttnConfig:
        c = self.config
        return Fn.AttnConfig(c.num_attention_heads, c.num_key_value_heads, c.head_dim, causal=True)

    def forward(self, hidden_states: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                kv_len: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, S, D = hidden_states.shape
        if not _plain_linears(self.q_proj, self.k_proj, self.v_proj, self.o_proj):
            return self.forward_modular(hidden_states, cos, sin, kv_len)
        x = hidden_states.reshape(B * S, D)
        out = AG.AttnBlockFn.apply(x if x.is_contiguous() else x.contiguous(), self.q_proj.weight, self.k_proj.weight,
                                   self.v_proj.weight, self.o_proj.weight, cos, sin, B, S, self.attn_config(), kv_len)
        return out.view(B, S, D)

    def forward_modular(self, hidden_states: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                        kv_len: Optional[torch.Tensor] = None) -> torch.Tensor:
        """LlamaAttention.forward (modeling_llama.py:243-281) from its differentiable pieces, calling the four projections as MODULES --
        what an adapter on q/k/v/o_proj (recipes/config_lora.yaml:47-59) needs; same kernels as the fused node."""
        c = self.config
        H, Hkv, hd = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        if Hkv != H:
            raise NotImplementedError("GQA (num_key_value_heads != num_attention_heads): Aria is MHA (gptfast/model.py:56-58)")
        B, S, D = hidden_states.shape
        x = hidden_states.reshape(B * S, D)
        q = AG.RopeFn.apply(self.q_proj(x), cos, sin, S, H, hd)
        k = AG.RopeFn.apply(self.k_proj(x), cos, sin, S, H, hd)
        o = AG.sdpa(q, k, self.v_proj(x), B, S, S, H, hd, hd ** -0.5, True, kv_len=kv_len)
        return self.o_proj(o).view(B, S, D)


class MoEDecoderLayer(nn.Module):
    """moe_lm.py:580-602 (+ LlamaDecoderLayer.forward).  forward(hidden_states [B,S,D], cos, sin, kv_len) -> [B,S,D]."""

    def __init__(self, config: AriaMoELMConfig, layer_idx: int):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.config = config
        self.self_attn = AriaAttention(config, layer_idx)
        self.mlp = MoELayer(config)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def _lora_sites(self):
        """{functional key: adapter module} when this layer's GEMMs carry LoRA adapters the fused node serves (aria_amd.lora layers, not merged,
        not disabled, not under expert parallelism; ARIA_LORA_FUSED=0 keeps the module-by-module arrangement) -- else None."""
        import os

        from .lora import GroupedGemmLoraLayer, LinearLoraLayer

        a, m = self.self_attn, self.mlp
        mods = dict(wq=a.q_proj, wk=a.k_proj, wv=a.v_proj, wo=a.o_proj, fc1=m.experts.fc1, fc2=m.experts.fc2, gate=m.shared_experts.gate_proj,
                    up=m.shared_experts.up_proj, down=m.shared_experts.down_proj)
        found = {}
        for key, mod in mods.items():
            if isinstance(mod, (GroupedGemmLoraLayer, LinearLoraLayer)):
                if mod.disable_adapters or mod.merged:   # the base weight alone is the module's function (merged: delta already inside it)
                    if mod.disable_adapters and mod.merged:
                        return None                      # (the module un-merges itself on its next call: let it)
                    continue
                if getattr(mod, "bias", None) is not None and isinstance(mod, LinearLoraLayer):
                    return None
                found[key] = mod
            elif type(mod) not in (Linear, GroupedGEMM):
                return None
        if not found or m.ep_enabled or os.environ.get("ARIA_LORA_FUSED", "1") == "0":
            return None
        if a.config.num_key_value_heads != a.config.num_attention_heads or a.config.head_dim not in (64, 128):
            return None
        return found

    def layer_params(self):
        a, m = self.self_attn, self.mlp
        return (self.input_layernorm.weight, a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, a.o_proj.weight,
                self.post_attention_layernorm.weight, m.router.weight, m.experts.fc1.weight, m.experts.fc2.weight,
                m.shared_experts.gate_proj.weight, m.shared_experts.up_proj.weight, m.shared_experts.down_proj.weight)

    def forward(self, hidden_states: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                kv_len: Optional[torch.Tensor] = None, recompute_level: Optional[str] = None) -> torch.Tensor:
        B, S, D = hidden_states.shape
        a = self.self_attn
        sites = self._lora_sites()
        if sites is not None:   # LoRA adapters on this layer's GEMMs: ONE node, the adapters inside the base launches (aria_amd.lora_functional)
            keys = tuple(sites)
            hyper = tuple((float(s.scaling), float(getattr(s.lora_dropout, "p", 0.0))) for s in sites.values())
            ab = [t for s in sites.values() for t in (s.lora_A.weight, s.lora_B.weight)]
            x = hidden_states.reshape(B * S, D)
            x = x if x.is_contiguous() else x.contiguous()
            seed = int(torch.empty((), dtype=torch.int64).random_()) if self.training else 0   # (host generator: torch.manual_seed governs it)
            mcfg, acfg, eps = self.mlp.moe_config(), self.self_attn.attn_config(), self.config.rms_norm_eps

            # the recipe's gradient_checkpointing (recipes/config_lora.yaml:17) is the node's own recompute flag: the forward keeps the layer
            # input and the seed, the backward re-runs the layer (same masks) -- no torch.utils.checkpoint wrapper
            recompute = bool(self.config.gradient_checkpointing and self.training and torch.is_grad_enabled())
            return AG.LoraDecoderLayerFn.apply(x, cos, sin, B, S, acfg, mcfg, eps, kv_len, self.training, seed, keys, hyper, recompute,
                                               *self.layer_params(), *ab).view(B, S, D)
        if self.mlp.ep_enabled or self.mlp.has_adapter() or not _plain_linears(a.q_proj, a.k_proj, a.v_proj, a.o_proj):
            # an adapter (aria_amd/lora.py) wraps a GEMM of this layer, or the experts are sharded over ranks (an all-to-all sits inside
            # the MoE block): LlamaDecoderLayer.forward module by module (modeling_llama.py:295-325)
            def block(x):
                h = x + self.self_attn(self.input_layernorm(x), cos, sin, kv_len)
                return h + self.mlp(self.post_attention_layernorm(h))

            if self.config.gradient_checkpointing and self.training and torch.is_grad_enabled() and not self.mlp.ep_enabled:
                # the recipe's gradient_checkpointing for the module-by-module form (the fused node has its own recompute flag); not with
                # sharded experts: a recomputed forward would issue its all-to-alls a second time, out of step with the other ranks' backward
                from torch.utils.checkpoint import checkpoint

                return checkpoint(block, hidden_states, use_reentrant=False)
            return block(hidden_states)
        x = hidden_states.reshape(B * S, D)
        x = x if x.is_contiguous() else x.contiguous()
        recompute = bool(self.config.gradient_checkpointing and self.training and torch.is_grad_enabled())
        if recompute and recompute_level in AG.RECOMPUTE_LEVELS:
            recompute = recompute_level
        out = AG.DecoderLayerFn.apply(x, cos, sin, B, S, self.self_attn.attn_config(), self.mlp.moe_config(),
                                      self.config.rms_norm_eps, kv_len, recompute, *self.layer_params())
        return out.view(B, S, D)


@dataclass
class CausalLMOutput:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    hidden_states: Optional[torch.Tensor] = None


class AriaMoELMModel(nn.Module):
    """moe_lm.py:605-636: embed_tokens, 28 x MoEDecoderLayer, final norm."""

    def __init__(self, config: AriaMoELMConfig):
        super().__init__()
        self.config = config
        self.padding_idx = config.pad_token_id
        self.vocab_size = config.vocab_size
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, dtype=bf16)
        self.layers = nn.ModuleList([MoEDecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self._rope = None

    def rope(self, S: int, device):
        if self._rope is None or self._rope[0].shape[0] < S or self._rope[0].device != torch.device(device):
            self._rope = Fn.rope_tables(max(S, 16), self.config.head_dim, self.config.rope_theta, device)
        return self._rope

    def embed(self, input_ids: torch.Tensor) -> torch.Tensor:
        B, S = input_ids.shape
        e = AG.EmbeddingFn.apply(input_ids.reshape(-1).to(torch.int32).contiguous(), self.embed_tokens.weight)
        return e.view(B, S, -1)

    @staticmethod
    def kv_len_from_mask(attention_mask: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """Right-padded batches only: the number of valid tokens per sequence (keys beyond it are masked; with a causal
        mask real tokens never see the padding anyway)."""
        if attention_mask is None:
            return None
        return attention_mask.ne(0).sum(dim=1).to(torch.int32).contiguous()

    def forward(self, input_ids: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                inputs_embeds: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = self.embed(input_ids) if inputs_embeds is None else inputs_embeds
        B, S, _ = h.shape
        cos, sin = self.rope(S, h.device)
        kv_len = self.kv_len_from_mask(attention_mask)
        level = self.recompute_level_for(B * S, h.device)
        for layer in self.layers:
            h = layer(h, cos, sin, kv_len, level)
        return self.norm(h)

    def recompute_level_for(self, tokens: int, device) -> Optional[str]:
        """The level of the recipe's gradient checkpointing for THIS forward (None: off), decided once for all layers from the memory that is
        free now and the gradients that are still to be allocated; recorded in ``last_recompute_level`` (bench.py / train.py report it)."""
        c = self.config
        if not (c.gradient_checkpointing and self.training and torch.is_grad_enabled()):
            self.last_recompute_level = None
            return None
        pending = sum(p.numel() * p.element_size() for p in self.parameters() if p.requires_grad and p.grad is None)
        self.last_recompute_level = AG.choose_recompute_level(
            tokens, c.hidden_size, c.moe_intermediate_size * c.moe_num_shared_experts, c.moe_topk, c.moe_intermediate_size, len(self.layers),
            pending_grad_bytes=pending, device=device, requested=getattr(c, "recompute_level", "auto"))
        return self.last_recompute_level


class AriaMoELMForCausalLM(nn.Module):
    """moe_lm.py:639-679.  forward(input_ids | inputs_embeds, attention_mask, labels, num_logits_to_keep) -> loss / logits.
    With labels the lm_head GEMM, the shifted masked cross-entropy and their backward are fused (no [T,V] fp32 tensor)."""

    _no_split_modules = ["MoEDecoderLayer"]

    def __init__(self, config: AriaMoELMConfig):
        super().__init__()
        self.config = config
        self.model = AriaMoELMModel(config)
        self.vocab_size = config.vocab_size
        self.lm_head = Linear(config.hidden_size, config.vocab_size)

    def _lm_head_lora_fusable(self) -> bool:
        import os

        from .lora import LinearLoraLayer

        h = self.lm_head
        return (isinstance(h, LinearLoraLayer) and not h.merged and not h.disable_adapters and h.bias is None and not h.base_layer.weight.requires_grad
                and os.environ.get("ARIA_LORA_FUSED", "1") != "0")

    def set_z_loss_coeff(self, v: float):
        self.config.moe_z_loss_coeff = v

    def set_aux_loss_coeff(self, v: float):
        self.config.moe_aux_loss_coeff = v

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, labels=None, num_logits_to_keep: int = 0,
                return_logits: Optional[bool] = None, labels_are_shifted: bool = False) -> CausalLMOutput:
        hn = self.model(input_ids=input_ids, attention_mask=attention_mask, inputs_embeds=inputs_embeds)
        B, S, D = hn.shape
        loss = logits = None
        if labels is not None:
            ls = labels.reshape(-1).to(torch.int32) if labels_are_shifted else Fn.shift_labels(labels, attention_mask)
            head = self.lm_head
            if _plain_linears(head):
                loss = AG.LMHeadLossFn.apply(hn.reshape(B * S, D), head.weight, ls.contiguous())
            elif self._lm_head_lora_fusable():   # recipes/config_lora.yaml:59: the adapter inside the lm_head launch, labelled rows only
                seed = int(torch.empty((), dtype=torch.int64).random_()) if self.training else 0
                loss = AG.LoraLMHeadLossFn.apply(hn.reshape(B * S, D), head.weight, ls.contiguous(), head.lora_A.weight, head.lora_B.weight,
                                                 float(head.scaling), float(getattr(head.lora_dropout, "p", 0.0)), self.training, seed)
            else:  # an adapted lm_head: its own forward produces the logits, then the CE kernel
                loss = AG.CrossEntropyFn.apply(self.lm_head(hn.reshape(B * S, D)), ls.contiguous())
        if return_logits or (labels is None and return_logits is None):
            hl = hn[:, -num_logits_to_keep:, :] if num_logits_to_keep else hn
            logits = self.lm_head(hl)
        return CausalLMOutput(loss=loss, logits=logits, hidden_states=hn)


def load_reference_state_dict(module: nn.Module, state_dict: dict, prefix: str = "", strict: bool = True):
    """Load a state dict keyed like the reference (fp32 or bf16 tensors) into the bf16 modules."""
    own = module.state_dict()
    missing = [k for k in own if prefix + k not in state_dict]
    if strict and missing:
        raise KeyError(f"missing keys: {missing[:5]}...")
    with torch.no_grad():
        for k, v in own.items():
            if prefix + k in state_dict:
                v.copy_(state_dict[prefix + k].to(v.dtype))
    return missing
