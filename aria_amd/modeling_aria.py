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


class _ScatterRowsFn(torch.autograd.Function):
    """out = embeds with rows ``rows`` replaced by ``feats`` -- ``masked_scatter`` with a mask that is constant along the feature axis
    (modeling_aria.py:272-283: the image-token positions, expanded over D), done on the 4 096 ROWS instead of the 42 M elements: the element-wise
    form costs a prefix sum over the whole mask, a partition, and two element-wise passes in the backward (~2 ms of the config #3 step)."""

    @staticmethod
    def forward(ctx, embeds, feats, rows):
        ctx.save_for_backward(rows)
        out = embeds.clone()
        out.index_copy_(0, rows, feats)
        return out

    @staticmethod
    def backward(ctx, dy):
        (rows,) = ctx.saved_tensors
        d_feats = dy.index_select(0, rows) if ctx.needs_input_grad[1] else None
        d_emb = None
        if ctx.needs_input_grad[0]:
            d_emb = dy.clone()
            d_emb.index_fill_(0, rows, 0)      # the token embeddings under the image positions were overwritten: no gradient
        return d_emb, d_feats, None


def scatter_image_rows(inputs_embeds: torch.Tensor, is_img: torch.Tensor, image_features: torch.Tensor) -> torch.Tensor:
    """``inputs_embeds.masked_scatter(is_img[..., None].expand_as(inputs_embeds), image_features)`` (modeling_aria.py:272-283) row by row: the
    i-th image-token position (row-major over [B, S]) receives the i-th feature row.  The position list has a known length (the feature
    rows: the count check above is the reference's own), so it is built without a host sync (``nonzero_static``)."""
    D = inputs_embeds.shape[-1]
    feats = image_features.reshape(-1, D)
    flat = is_img.reshape(-1)
    rows = torch.nonzero_static(flat, size=feats.shape[0], fill_value=0).reshape(-1)
    return _ScatterRowsFn.apply(inputs_embeds.reshape(-1, D), feats, rows).view(inputs_embeds.shape)


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

    # ---- expert parallelism (BASELINE config #5)
    def enable_expert_parallel(self, group=None) -> None:
        """Shard the routed experts of every decoder layer over ``group`` (default: all ranks); everything else stays replicated."""
        for layer in self.language_model.model.layers:
            layer.mlp.enable_expert_parallel(group)

    def expert_parallel_group(self):
        mlp = self.language_model.model.layers[0].mlp
        return (mlp.ep_group, True) if mlp.ep_enabled else (None, False)

    def full_state_dict(self) -> dict:
        """``state_dict()`` in the reference layout on EVERY rank: the expert shards of an expert-parallel model are all-gathered back
        into ``[E, ...]`` tensors (a collective: call it on all ranks); identical to ``state_dict()`` otherwise."""
        sd = {k: v.detach() for k, v in self.state_dict().items()}
        group, ep = self.expert_parallel_group()
        if not ep:
            return sd
        import torch.distributed as dist

        W = dist.get_world_size(group)
        for k in [k for k in sd if k.endswith(("mlp.experts.fc1.weight", "mlp.experts.fc2.weight"))]:
            parts = [torch.empty_like(sd[k]) for _ in range(W)]
            dist.all_gather(parts, sd[k].contiguous(), group=group)
            sd[k] = torch.cat(parts, dim=0)
        return sd

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None) -> None:
        """HF's switch (the Trainer calls it for ``gradient_checkpointing: true``): per-layer recompute in the decoder."""
        self.config.text_config.gradient_checkpointing = True

    def gradient_checkpointing_disable(self) -> None:
        self.config.text_config.gradient_checkpointing = False

    def num_parameters(self, only_trainable: bool = False) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad or not only_trainable)

    def get_input_embeddings(self):
        return self.language_model.get_input_embeddings()

    def set_moe_z_loss_coeff(self, value: float):
        self.language_model.set_z_loss_coeff(value)

    def set_moe_aux_loss_coeff(self, value: float):
        self.language_model.set_aux_loss_coeff(value)

    # ---- HF checkpoint directory <-> module (config.json + sharded safetensors, the layout `from_pretrained("rhymes-ai/Aria")` reads)
    @classmethod
    def from_pretrained(cls, path: str, device="cpu", strict: bool = True) -> "AriaForConditionalGeneration":
        """``config.json`` (``vision_config`` / ``text_config`` dicts, ``projector_patch_to_query_dict``, ``image_token_index``:
        configuration_aria.py:31-111) + the weight shards of a local checkpoint directory.  Unknown config keys are kept in ``.extra``."""
        import json
        import os

        from .checkpoint import load_hf_dir_into

        with open(os.path.join(path, "config.json")) as f:
            raw = json.load(f)
        keys = ("vision_config", "text_config", "projector_patch_to_query_dict", "ignore_index", "image_token_index")
        config = AriaConfig(**{k: raw[k] for k in keys if k in raw}, **{k: v for k, v in raw.items() if k not in keys})
        prev = torch.get_default_device() if hasattr(torch, "get_default_device") else None
        torch.set_default_device(device)
        try:
            model = cls(config)
        finally:
            torch.set_default_device(prev if prev is not None else "cpu")
        load_hf_dir_into(model, path, strict=strict)  # shard by shard: the host never holds the whole checkpoint
        return model.eval()

    def save_pretrained(self, path: str, max_shard_bytes: int = 5 << 30, state_dict: Optional[dict] = None) -> None:
        import json
        import os

        from .checkpoint import save_checkpoint_dir

        os.makedirs(path, exist_ok=True)
        t, v = self.config.text_config, self.config.vision_config
        text = dict(model_type=t.model_type, moe_intermediate_size=t.moe_intermediate_size, moe_num_experts=t.moe_num_experts,
                    moe_topk=t.moe_topk, moe_z_loss_coeff=t.moe_z_loss_coeff, moe_aux_loss_coeff=t.moe_aux_loss_coeff,
                    moe_num_shared_experts=t.moe_num_shared_experts, hidden_size=t.hidden_size, num_hidden_layers=t.num_hidden_layers,
                    num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads, vocab_size=t.vocab_size,
                    rms_norm_eps=t.rms_norm_eps, rope_theta=t.rope_theta, max_position_embeddings=t.max_position_embeddings,
                    pad_token_id=t.pad_token_id)
        vision = dict(model_type=v.model_type, hidden_size=v.hidden_size, num_hidden_layers=v.num_hidden_layers,
                      num_attention_heads=v.num_attention_heads, intermediate_size=v.intermediate_size, patch_size=v.patch_size,
                      image_size=v.image_size, num_channels=v.num_channels, layer_norm_eps=v.layer_norm_eps)
        cfg = dict(model_type=self.config.model_type, architectures=["AriaForConditionalGeneration"], text_config=text, vision_config=vision,
                   projector_patch_to_query_dict={str(k): q for k, q in self.config.projector_patch_to_query_dict.items()},
                   ignore_index=self.config.ignore_index, image_token_index=self.config.image_token_index, torch_dtype="bfloat16")
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(cfg, f, indent=2)
        save_checkpoint_dir(self.state_dict() if state_dict is None else state_dict, path, max_shard_bytes=max_shard_bytes)

    # ---- inference bridge: HF surface (training layout) -> gptfast surface (inference layout)
    def to_gptfast(self):
        """An ``aria_amd.gptfast.Aria`` twin of this model: the weights go through the reference's own conversion
        (gptfast/scripts/convert_hf_checkpoint.py = ``aria_amd.checkpoint.hf_to_gptfast``: q/k rows permuted for interleaved RoPE,
        ``wqkv`` fusion, experts ``fc1 -> w1/w3 [E,I,K]``, ``fc2 -> w2``), so a model trained or loaded on the HF surface decodes
        through the fast path (tile kernels for the prefill, the native one-call-per-token engine for decode).  The twin owns a
        converted COPY of the language model (expert matrices change layout); vision tower and projector tensors are shared."""
        from . import gptfast as G
        from .checkpoint import hf_to_gptfast

        t, v = self.config.text_config, self.config.vision_config
        args = G.ModelArgs(block_size=t.max_position_embeddings, vocab_size=t.vocab_size, n_layer=t.num_hidden_layers,
                           n_head=t.num_attention_heads, dim=t.hidden_size, intermediate_size=t.moe_intermediate_size,
                           n_local_heads=t.num_key_value_heads, rope_base=t.rope_theta, norm_eps=t.rms_norm_eps,
                           num_experts=t.moe_num_experts, router_topk=t.moe_topk, num_shared_experts=t.moe_num_shared_experts,
                           image_token_index=self.config.image_token_index)
        dev = self.language_model.lm_head.weight.device
        prev = torch.get_default_device() if hasattr(torch, "get_default_device") else None
        torch.set_default_device(dev)
        try:
            twin = G.Aria(args, v, self.config.projector_patch_to_query_dict)
        finally:
            torch.set_default_device(prev if prev is not None else "cpu")
        sd = {k: p.detach() for k, p in self.state_dict().items()}
        if any(".lora_" in k or ".base_layer." in k for k in sd):
            raise RuntimeError("to_gptfast()/generate(): LoRA adapters are still attached -- fold them in first (aria_amd.lora.merge_and_unload)")
        if self.expert_parallel_group()[1]:
            raise RuntimeError("to_gptfast()/generate(): the experts are sharded over ranks -- generate from a checkpoint (full_state_dict / save_pretrained)")
        conv = hf_to_gptfast(sd, t.num_attention_heads, t.head_dim, t.num_key_value_heads)
        G.load_model_pth(twin, conv, strict=True)
        twin.vision_tower, twin.multi_modal_projector = self.vision_tower, self.multi_modal_projector  # share, do not copy
        return twin.eval()

    @property
    def device(self) -> torch.device:
        return self.language_model.lm_head.weight.device

    @property
    def dtype(self) -> torch.dtype:
        return self.language_model.lm_head.weight.dtype

    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor = None, pixel_values: Optional[torch.Tensor] = None, pixel_mask: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, max_new_tokens: int = 128, do_sample: bool = True, temperature: float = 0.8,
                 top_k: Optional[int] = 200, stop_strings=None, tokenizer=None, stop_token: Optional[int] = None, refresh: bool = False,
                 **unused) -> torch.Tensor:
        """The README quick start's call (README.md:45-88: ``model.generate(**inputs, max_new_tokens=..., stop_strings=["<|im_end|>"],
        tokenizer=processor.tokenizer, do_sample=True, temperature=0.9)``) for batch 1, served by the sampling loop of
        gptfast/generate.py:112-177 on the gptfast twin (built on first use; ``refresh=True`` rebuilds it after the weights changed).
        input_ids [1, T] -> [1, T + new] like HF (prompt included).  ``do_sample=False`` is greedy; a stop string that is a single token of
        ``tokenizer`` is compared on the device, longer ones through the reference's decode-and-compare hook; ``attention_mask`` of a single
        unpadded sequence carries no information and is ignored."""
        from . import gptfast as G

        if input_ids is None or input_ids.dim() != 2 or input_ids.shape[0] != 1:
            raise NotImplementedError("generate(): batch 1 (the gptfast decode path); input_ids [1, T]")
        if attention_mask is not None and not bool(attention_mask.ne(0).all()):
            raise NotImplementedError("generate(): padded prompts are not supported (batch 1 has nothing to pad)")
        if refresh or getattr(self, "_gptfast_twin", None) is None:
            object.__setattr__(self, "_gptfast_twin", self.to_gptfast())
            object.__setattr__(self, "_gptfast_decoder", None)
        if not do_sample:
            temperature, top_k = 1.0, 1
        callback = None
        if stop_strings:
            if tokenizer is None:
                raise ValueError("generate(stop_strings=...) needs tokenizer=... (like HF's)")
            multi = []
            for text in stop_strings:
                ids = tokenizer.encode(text, add_special_tokens=False) if hasattr(tokenizer, "convert_tokens_to_ids") else tokenizer.encode(text)
                if len(ids) == 1 and stop_token is None:
                    stop_token = int(ids[0])
                else:
                    multi.append(text)
            if multi:
                def callback(tokens):
                    decoded = tokenizer.decode(torch.cat(tokens).tolist())
                    return any(decoded.endswith(t) for t in multi)
        key = (float(temperature), top_k)
        if getattr(self, "_gptfast_sampling", key) != key:  # the cached decoder closes over its sampling parameters
            object.__setattr__(self, "_gptfast_decoder", None)
        object.__setattr__(self, "_gptfast_sampling", key)
        out, dec = G.generate(self._gptfast_twin, input_ids, max_new_tokens, pixel_values=pixel_values, pixel_mask=pixel_mask,
                              temperature=temperature, top_k=top_k, decoder=self._gptfast_decoder, stop_token=stop_token, callback=callback)
        object.__setattr__(self, "_gptfast_decoder", dec)
        return out.view(1, -1)

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
            is_img = input_ids == self.config.image_token_index
            pending = None
            if validate_image_tokens and is_img.is_cuda:
                # :265-271 is `.sum().item()`: a host sync in the middle of forward.  Same check, same ValueError, no stall of the GPU: the count
                # is enqueued FIRST and copied to pinned memory, the ViT + projector (hundreds of launches) are enqueued behind it, and the
                # host reads the count after that -- by then it has long arrived, and the device has the whole tower queued while we look.
                host = getattr(self, "_img_count_host", None)
                if host is None:
                    host = torch.empty(1, dtype=torch.int64).pin_memory()
                    object.__setattr__(self, "_img_count_host", host)
                host.copy_(is_img.sum().view(1), non_blocking=True)
                pending = torch.cuda.Event()
                pending.record()
            image_features = self.image_features(pixel_values, pixel_mask)                   # :254-262
            if validate_image_tokens:
                if pending is not None:
                    pending.synchronize()
                    n_tok = int(host[0])
                else:
                    n_tok = int(is_img.sum().item())
                n_feat = image_features.shape[0] * image_features.shape[1]
                if n_tok != n_feat:
                    raise ValueError(f"Image features and image tokens do not match: tokens: {n_tok}, features {n_feat}")
            if validate_image_tokens:    # (the count check above is what makes the row list's length known)
                inputs_embeds = scatter_image_rows(inputs_embeds, is_img, image_features.to(inputs_embeds.dtype))   # :272-283
            else:
                mask = is_img.unsqueeze(-1).expand_as(inputs_embeds)
                inputs_embeds = inputs_embeds.masked_scatter(mask, image_features.to(inputs_embeds.dtype))
        out = self.language_model(inputs_embeds=inputs_embeds, attention_mask=attention_mask, labels=labels,
                                  num_logits_to_keep=num_logits_to_keep, return_logits=return_logits)
        return AriaCausalLMOutputWithPast(loss=out.loss, logits=out.logits, image_hidden_states=image_features)
