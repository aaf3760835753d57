"""``AriaForConditionalGeneration`` on the HIP hot path -- mirror of aria/model/modeling_aria.py:125-365 and
aria/model/configuration_aria.py:31-114 (same sub-module names: ``vision_tower``, ``multi_modal_projector``, ``language_model``;
same forward keyword arguments for the tensors the path uses; same freeze_* / set_moe_* helpers)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn as nn

from .moe_lm import AriaMoELMConfig, AriaMoELMForCausalLM
from .vision import AriaProjector, AriaVisionConfig, AriaVisionModel

bf16 = torch.bfloat16


class AriaConfig:
    model_type = "aria"

    def __init__(self, vision_config=None, text_config=None, projector_patch_to_query_dict: Optional[Dict[int, int]] = None,
                 ignore_index: int = -100, image_token_index: int = 32000, **kwargs):
        if isinstance(vision_config, dict):
            vision_config = AriaVisionConfig(**{k: v for k, v in vision_config.items() if k != "model_type"})
        if isinstance(text_config, dict):
            text_config = AriaMoELMConfig(**{k: v for k, v in text_config.items() if k != "model_type"})
        self.vision_config = vision_config or AriaVisionConfig()
        self.text_config = text_config or AriaMoELMConfig()
        p2q = projector_patch_to_query_dict or {1225: 128, 4900: 256}
        self.projector_patch_to_query_dict = {int(k): int(v) for k, v in p2q.items()}
        self.ignore_index = ignore_index
        self.image_token_index = image_token_index
        self.num_hidden_layers = self.text_config.num_hidden_layers
        self.extra = kwargs


def build_mm_projector(config: AriaConfig) -> AriaProjector:
    """modeling_aria.py:104-121."""
    return AriaProjector(patch_to_query_dict=config.projector_patch_to_query_dict, embed_dim=config.vision_config.hidden_size,
                         num_heads=config.vision_config.num_attention_heads, kv_dim=config.vision_config.hidden_size,
                         ff_dim=config.text_config.hidden_size, output_dim=config.text_config.hidden_size)


@dataclass
class AriaCausalLMOutputWithPast:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    image_hidden_states: Optional[torch.Tensor] = None


class AriaForConditionalGeneration(nn.Module):
    def __init__(self, config: AriaConfig):
        super().__init__()
        self.config = config
        self.vision_tower = AriaVisionModel(config.vision_config)
        self.multi_modal_projector = build_mm_projector(config)
        self.vocab_size = config.text_config.vocab_size
        self.language_model = AriaMoELMForCausalLM(config.text_config)

    # ---- modeling_aria.py:145-192
    def freeze_vit(self):
        for p in self.vision_tower.parameters():
            p.requires_grad = False

    def freeze_projector(self):
        for p in self.multi_modal_projector.parameters():
            p.requires_grad = False

    def freeze_llm(self):
        for p in self.language_model.parameters():
            p.requires_grad = False

    def get_input_embeddings(self):
        return self.language_model.get_input_embeddings()

    def set_moe_z_loss_coeff(self, value: float):
        self.language_model.set_z_loss_coeff(value)

    def set_moe_aux_loss_coeff(self, value: float):
        self.language_model.set_aux_loss_coeff(value)

    def image_features(self, pixel_values: torch.Tensor, pixel_mask: Optional[torch.Tensor]) -> torch.Tensor:
        feat, atts = self.vision_tower(pixel_values, pixel_mask)
        return self.multi_modal_projector(feat, attn_mask=atts)

    def forward(self, input_ids: torch.Tensor = None, pixel_values: Optional[torch.Tensor] = None,
                pixel_mask: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                inputs_embeds: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                num_logits_to_keep: int = 0, return_logits: Optional[bool] = None,
                validate_image_tokens: bool = True) -> AriaCausalLMOutputWithPast:
        if inputs_embeds is None:
            inputs_embeds = self.language_model.model.embed(input_ids)                       # :250
        image_features = None
        if pixel_values is not None:
            image_features = self.image_features(pixel_values, pixel_mask)                   # :254-262
            is_img = input_ids == self.config.image_token_index
            if validate_image_tokens:                                                        # :265-271 (one host sync)
                n_tok = int(is_img.sum().item())
                n_feat = image_features.shape[0] * image_features.shape[1]
                if n_tok != n_feat:
                    raise ValueError(f"Image features and image tokens do not match: tokens: {n_tok}, features {n_feat}")
            mask = is_img.unsqueeze(-1).expand_as(inputs_embeds)
            inputs_embeds = inputs_embeds.masked_scatter(mask, image_features.to(inputs_embeds.dtype))   # :272-283
        out = self.language_model(inputs_embeds=inputs_embeds, attention_mask=attention_mask, labels=labels,
                                  num_logits_to_keep=num_logits_to_keep, return_logits=return_logits)
        return AriaCausalLMOutputWithPast(loss=out.loss, logits=out.logits, image_hidden_states=image_features)
