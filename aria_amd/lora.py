"""LoRA for the reference's cheapest fine-tune recipe (recipes/config_lora.yaml:44-59).

Grouped expert GEMMs: ``experts.fc1`` / ``experts.fc2`` get per-expert rank-r factors through the same
``experts_gemm`` seam (aria/lora/layers.py:30-224):

    y = base(x, tpe) + lora_B(lora_A(dropout(x), tpe), tpe) * (lora_alpha / r)        (layers.py:129-139)

``lora_A`` is a GroupedGEMM(in, r, groups) (weight [E, in, r]) and ``lora_B`` a GroupedGEMM(r, out, groups) (weight [E, r, out]); both are
plain calls of the MI355X grouped GEMM (N = r for A, K = r for B; r must be a multiple of 8 for the 16-byte operand granules).  peft is
not in this image, so the layer is a stand-alone module with peft's surface for this class: ``merge`` / ``unmerge`` /
``get_delta_weight`` / ``scaling`` / ``disable_adapters``; ``apply_lora_to_experts`` is the part of ``get_peft_model`` the recipe uses
(wrap the target modules, freeze everything else).

The recipe's other targets (q/k/v/o_proj, gate/up/down_proj of the shared experts, lm_head) are plain ``nn.Linear``-shaped modules; in the
reference they get peft's stock Linear adapter (peft/tuners/lora/layer.py, ``Linear.forward``):

    y = base(x) + lora_B(lora_A(dropout(x))) * (lora_alpha / r)          lora_A: Linear(in, r), lora_B: Linear(r, out), no bias

``LinearLoraLayer`` is that adapter on this package's ``Linear`` (three MFMA GEMMs, N = r for A, K = r for B); the decoder blocks notice a
wrapped projection and run module by module instead of through their fused autograd nodes (``moe_lm._plain_linears``).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable

import torch
from torch import nn

from .moe_lm import GroupedGEMM, Linear


class GroupedGemmLoraLayer(nn.Module):
    def __init__(self, base_layer: GroupedGEMM, r: int = 8, lora_alpha: int = 16, lora_dropout: float = 0.0,
                 init_lora_weights: bool = True):
        super().__init__()
        if r <= 0:
            raise ValueError(f"`r` should be a positive integer value but the value passed is {r}")
        if r % 8:
            raise ValueError("r must be a multiple of 8 (16-byte bf16 granules of the grouped GEMM operands)")
        self.base_layer = base_layer
        self.in_features, self.out_features, self.groups = base_layer.in_features, base_layer.out_features, base_layer.groups
        self.r, self.lora_alpha, self.scaling = r, lora_alpha, lora_alpha / r
        self.lora_dropout = nn.Dropout(p=lora_dropout) if lora_dropout > 0.0 else nn.Identity()
        self.lora_A = GroupedGEMM(self.in_features, r, self.groups)
        self.lora_B = GroupedGEMM(r, self.out_features, self.groups)
        self.lora_A.to(base_layer.weight.device)
        self.lora_B.to(base_layer.weight.device)
        self.merged = False
        self.disable_adapters = False
        if init_lora_weights:
            self.reset_lora_parameters()
        base_layer.weight.requires_grad_(False)

    @property
    def weight(self):  # peft's LoraLayer exposes the base weight the same way
        return self.base_layer.weight

    def reset_lora_parameters(self):
        """peft's default: A ~ kaiming_uniform(a = sqrt(5)), B = 0 -> the adapted layer starts as the base layer."""
        with torch.no_grad():
            a = torch.empty(self.lora_A.weight.shape, dtype=torch.float32)
            nn.init.kaiming_uniform_(a, a=math.sqrt(5))
            self.lora_A.weight.copy_(a.to(self.lora_A.weight.dtype))
            self.lora_B.weight.zero_()

    def forward(self, x: torch.Tensor, tokens_per_expert: torch.Tensor) -> torch.Tensor:
        if self.disable_adapters:
            if self.merged:
                self.unmerge()
            return self.base_layer(x, tokens_per_expert)
        result = self.base_layer(x, tokens_per_expert)
        if self.merged:
            return result
        dtype = result.dtype
        x = x.to(self.lora_A.weight.dtype)
        result = result + self.lora_B(self.lora_A(self.lora_dropout(x), tokens_per_expert), tokens_per_expert) * self.scaling
        return result.to(dtype)

    def get_delta_weight(self) -> torch.Tensor:
        """layers.py:196-224: matmul(A, B) * scaling, per expert, in the adapter dtype."""
        return torch.matmul(self.lora_A.weight, self.lora_B.weight) * self.scaling

    def merge(self) -> None:
        if not self.merged:
            self.base_layer.weight.data += self.get_delta_weight()
            self.merged = True

    def unmerge(self) -> None:
        if self.merged:
            self.base_layer.weight.data -= self.get_delta_weight()
            self.merged = False


class LinearLoraLayer(nn.Module):
    """peft's LoRA ``Linear`` on ``aria_amd.moe_lm.Linear`` (weight [out, in]): ``lora_A.weight`` [r, in] ~ kaiming_uniform(a = sqrt(5)),
    ``lora_B.weight`` [out, r] = 0, delta weight = B @ A * scaling."""

    def __init__(self, base_layer: Linear, r: int = 8, lora_alpha: int = 16, lora_dropout: float = 0.0, init_lora_weights: bool = True):
        super().__init__()
        if r <= 0:
            raise ValueError(f"`r` should be a positive integer value but the value passed is {r}")
        if r % 8:
            raise ValueError("r must be a multiple of 8 (16-byte bf16 granules of the GEMM operands)")
        self.base_layer = base_layer
        self.in_features, self.out_features = base_layer.in_features, base_layer.out_features
        self.r, self.lora_alpha, self.scaling = r, lora_alpha, lora_alpha / r
        self.lora_dropout = nn.Dropout(p=lora_dropout) if lora_dropout > 0.0 else nn.Identity()
        self.lora_A = Linear(self.in_features, r).to(base_layer.weight.device)
        self.lora_B = Linear(r, self.out_features).to(base_layer.weight.device)
        self.merged = False
        self.disable_adapters = False
        if init_lora_weights:
            self.reset_lora_parameters()
        for p in base_layer.parameters():
            p.requires_grad_(False)

    @property
    def weight(self):
        return self.base_layer.weight

    @property
    def bias(self):
        return self.base_layer.bias

    def reset_lora_parameters(self):
        with torch.no_grad():
            a = torch.empty(self.lora_A.weight.shape, dtype=torch.float32)
            nn.init.kaiming_uniform_(a, a=math.sqrt(5))
            self.lora_A.weight.copy_(a.to(self.lora_A.weight.dtype))
            self.lora_B.weight.zero_()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.disable_adapters:
            if self.merged:
                self.unmerge()
            return self.base_layer(x)
        result = self.base_layer(x)
        if self.merged:
            return result
        return result + self.lora_B(self.lora_A(self.lora_dropout(x))) * self.scaling

    def get_delta_weight(self) -> torch.Tensor:
        return torch.matmul(self.lora_B.weight, self.lora_A.weight) * self.scaling

    def merge(self) -> None:
        if not self.merged:
            self.base_layer.weight.data += self.get_delta_weight()
            self.merged = True

    def unmerge(self) -> None:
        if self.merged:
            self.base_layer.weight.data -= self.get_delta_weight()
            self.merged = False


def apply_lora_to_experts(model: nn.Module, r: int = 8, lora_alpha: int = 16, lora_dropout: float = 0.0,
                          target_suffixes: Iterable[str] = ("experts.fc1", "experts.fc2")) -> nn.Module:
    """Freeze every parameter, wrap each ``GroupedGEMM`` whose qualified name ends with one of ``target_suffixes`` (the recipe's
    ``lora_target_modules`` for the experts, aria/train.py:100-112) and leave only the LoRA factors trainable."""
    for p in model.parameters():
        p.requires_grad_(False)
    targets = [(name, mod) for name, mod in model.named_modules()
               if isinstance(mod, GroupedGEMM) and any(name.endswith(s) for s in target_suffixes)]
    for name, mod in targets:
        parent = model.get_submodule(name.rsplit(".", 1)[0]) if "." in name else model
        setattr(parent, name.rsplit(".", 1)[-1], GroupedGemmLoraLayer(mod, r, lora_alpha, lora_dropout))
    return model


def lora_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    return {k: v for k, v in model.state_dict().items() if ".lora_A." in k or ".lora_B." in k}


def get_lora_target_modules(model_named_modules, cfg) -> list:
    """aria/lora/utils.py:29-63: every module name that contains one of ``cfg["lora_target_modules"]`` and is not under a frozen
    part (freeze_vit / freeze_projector / freeze_llm / freeze_llm_layers)."""
    out = []
    for key in model_named_modules:
        if cfg.get("freeze_vit") and "vision_tower" in key:
            continue
        if cfg.get("freeze_projector") and "multi_modal_projector" in key:
            continue
        if cfg.get("freeze_llm") and "language_model" in key:
            continue
        if any(f"language_model.model.layers.{i}." in key for i in (cfg.get("freeze_llm_layers") or [])):
            continue
        if any(m in key for m in cfg.get("lora_target_modules") or []):
            out.append(key)
    return out


def apply_lora_from_config(model: nn.Module, cfg) -> list:
    """``use_peft: true`` of recipes/config_lora.yaml (aria/train.py:100-112): among the selected target modules the grouped expert GEMMs
    get a ``GroupedGemmLoraLayer`` and the ``Linear`` projections of the language model (q/k/v/o_proj, shared experts, lm_head) a
    ``LinearLoraLayer``; everything else is frozen.  Returned: the selected names that were NOT adapted (``nn.Linear`` modules of the
    projector / vision tower when the recipe does not freeze them: their blocks have no module-by-module path here)."""
    names = [n for n, _ in model.named_modules()]
    targets = get_lora_target_modules(names, cfg)
    r, alpha, drop = int(cfg.get("lora_r", 8)), int(cfg.get("lora_alpha", 32)), float(cfg.get("lora_dropout", 0.0))
    grouped = [n for n in targets if isinstance(model.get_submodule(n), GroupedGEMM)]
    linear = [n for n in targets if type(model.get_submodule(n)) is Linear and ("language_model." in n or n.startswith(("model.", "lm_head")))]
    for p in model.parameters():
        p.requires_grad_(False)
    for name in grouped + linear:
        parent = model.get_submodule(name.rsplit(".", 1)[0]) if "." in name else model
        base = model.get_submodule(name)
        layer = GroupedGemmLoraLayer(base, r, alpha, drop) if name in grouped else LinearLoraLayer(base, r, alpha, drop)
        setattr(parent, name.rsplit(".", 1)[-1], layer)
    adapted = grouped + linear
    return [n for n in targets if n not in adapted and not any(n.startswith(g + ".") for g in adapted)]


def merge_and_unload(model: nn.Module) -> nn.Module:
    """peft's ``merge_and_unload``: fold every adapter into its base weight (W += delta) and put the plain base module back, so the
    fused blocks, ``save_pretrained`` and ``to_gptfast()`` / the decode engine see ordinary reference-layout weights again."""
    wrapped = [(n, m) for n, m in model.named_modules() if isinstance(m, (GroupedGemmLoraLayer, LinearLoraLayer))]
    for name, layer in wrapped:
        layer.merge()
        parent = model.get_submodule(name.rsplit(".", 1)[0]) if "." in name else model
        setattr(parent, name.rsplit(".", 1)[-1], layer.base_layer)
    return model


def load_lora_adapter(model: nn.Module, path: str, **freeze) -> list:
    """Re-create the adapters a ``use_peft`` run of ``aria_amd.train`` saved (``adapter_config.json`` + ``adapter_model.safetensors``) on a
    freshly loaded base model: wraps the same target modules (``freeze`` = the run's freeze_vit / freeze_projector / freeze_llm /
    freeze_llm_layers selection, default: the recipe's freeze_vit + freeze_projector) and copies the factors in.  Returns the adapted names."""
    import json
    import os

    from safetensors.torch import load_file

    with open(os.path.join(path, "adapter_config.json")) as f:
        ac = json.load(f)
    cfg = dict(lora_r=ac["r"], lora_alpha=ac["lora_alpha"], lora_dropout=ac.get("lora_dropout", 0.0),
               lora_target_modules=ac["target_modules"], freeze_vit=True, freeze_projector=True)
    cfg.update(freeze)
    apply_lora_from_config(model, cfg)
    factors = load_file(os.path.join(path, "adapter_model.safetensors"))
    own = lora_state_dict(model)
    if set(own) != set(factors):
        raise KeyError(f"adapter / model mismatch: {sorted(set(own) ^ set(factors))[:4]}")
    with torch.no_grad():
        for k, v in own.items():
            v.copy_(factors[k].to(v.dtype))
    return sorted({k.rsplit(".lora_", 1)[0] for k in own})
