"""Hugging Face surface of the native modules (SURVEY 8b seam B3): ``PreTrainedConfig`` / ``PreTrainedModel`` + ``GenerationMixin``
subclasses registered with ``AutoConfig`` / ``AutoModelForCausalLM`` under the reference's model types, so that
``AutoModelForCausalLM.from_pretrained(dir)``, ``save_pretrained``, HF ``generate()``, ``Trainer`` and peft's module walk accept them.

Mirrors aria/model/modeling_aria.py:38-58 (``AriaPretrainedModel``: ``config_class``, ``base_model_prefix = "model"``,
``_no_split_modules``, ``supports_gradient_checkpointing``, ``_supports_flash_attn_2 / _supports_sdpa / _supports_cache_class``),
``:125-192`` (the model class and its freeze_* / set_moe_* helpers) and ``:337-365`` (``prepare_inputs_for_generation``), and
aria/model/configuration_aria.py:31-114 / moe_lm.py:43-80 / vision_encoder.py:31-40 (the three configs).

The arithmetic is the native path's (``aria_amd.modeling_aria.AriaForConditionalGeneration``: HIP kernels behind the C ABI); this file
only adds the HF plumbing.  State-dict keys are the reference's (``vision_tower.*``, ``multi_modal_projector.*``, ``language_model.*``),
so checkpoints written by either class load into the other and into the reference.

KV cache (modeling_aria.py:43-58, 337-365): HF ``generate()`` runs WITH a cache.  ``past_key_values`` is an ``AriaStaticKVCache`` -- the
static bf16 cache ``[1, S_max, H * hd]`` per layer of the model's gptfast twin (gptfast/model.py:67-93) with the one-call-per-token decode
engine (csrc/decode.hip) behind it: the first forward prefills the prompt (ViT + projector run ONCE, tile kernels), every later forward
decodes the new token against the cache -- linear time, any ``GenerationConfig`` sampling option of the HF loop.  Any batch size: every row
of a (left-padded) batch owns a row of static caches and a pointer table of the decode engine, is prefilled from its own first real token
(positions count from 0 there, as HF's ``position_ids = cumsum(attention_mask) - 1`` has it) and decoded by the batch-1 engine -- a batch of B
rows is B batch-1 generations step for step, token for token; beam search re-orders (copies) rows.  Right-padded or hole-y masks, assisted
decoding, or ``use_cache=False`` take the cache-free path (every step re-runs the prefix: correct, quadratic).  ``.generate_fast`` is the
native sampling loop (gptfast/generate.py:112-177) on the same engine.
"""
from __future__ import annotations

from typing import Optional

import torch
from transformers import AutoConfig, AutoModelForCausalLM, GenerationMixin, PreTrainedModel

try:  # transformers >= 4.5x spells it PreTrainedConfig; the reference's pin (4.45 era) only has PretrainedConfig
    from transformers import PreTrainedConfig
except ImportError:  # pragma: no cover - depends on the installed transformers
    from transformers import PretrainedConfig as PreTrainedConfig
from transformers.modeling_outputs import CausalLMOutputWithPast

from . import modeling_aria as native
from .moe_lm import AriaMoELMConfig as NativeTextConfig
from .vision import AriaVisionConfig as NativeVisionConfig

_TEXT_KEYS = ("moe_intermediate_size", "moe_num_experts", "moe_topk", "moe_z_loss_coeff", "moe_aux_loss_coeff", "moe_num_shared_experts",
              "hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "vocab_size", "rms_norm_eps", "rope_theta",
              "max_position_embeddings", "pad_token_id")
_VISION_KEYS = ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size", "patch_size", "image_size", "num_channels",
                "layer_norm_eps")


class AriaMoELMHFConfig(PreTrainedConfig):
    """aria/model/moe_lm.py:43-80 (``AriaMoELMConfig(LlamaConfig)``), Aria-25.3B defaults."""

    model_type = "aria_moe_lm"

    def __init__(self, **kwargs):
        n = NativeTextConfig(**{k: kwargs[k] for k in _TEXT_KEYS if k in kwargs})
        for k in _TEXT_KEYS:
            kwargs[k] = getattr(n, k)
        pad = kwargs.pop("pad_token_id")
        for k, v in kwargs.items():
            if k in _TEXT_KEYS:
                setattr(self, k, v)
        super().__init__(**{k: v for k, v in kwargs.items() if k not in _TEXT_KEYS})
        self.pad_token_id = pad


class AriaVisionHFConfig(PreTrainedConfig):
    """aria/model/vision_encoder.py:31-40 (``AriaVisionConfig(SiglipVisionConfig)``)."""

    model_type = "aria_vision_model"

    def __init__(self, **kwargs):
        n = NativeVisionConfig(**{k: kwargs[k] for k in _VISION_KEYS if k in kwargs})
        for k in _VISION_KEYS:
            setattr(self, k, getattr(n, k))
        super().__init__(**{k: v for k, v in kwargs.items() if k not in _VISION_KEYS})


class AriaHFConfig(PreTrainedConfig):
    """aria/model/configuration_aria.py:31-114: ``vision_config`` / ``text_config`` sub-configs (dicts in config.json),
    ``projector_patch_to_query_dict``, ``ignore_index``, ``image_token_index``."""

    model_type = "aria"
    sub_configs = {"text_config": AriaMoELMHFConfig, "vision_config": AriaVisionHFConfig}
    has_no_defaults_at_init = True

    def __init__(self, vision_config=None, text_config=None, projector_patch_to_query_dict=None, ignore_index: int = -100,
                 image_token_index: int = 32000, **kwargs):
        def sub(cls, v):
            if isinstance(v, cls):
                return v
            if v is None:
                return cls()
            d = v if isinstance(v, dict) else {k: getattr(v, k) for k in (_TEXT_KEYS if cls is AriaMoELMHFConfig else _VISION_KEYS)}
            return cls(**{k: x for k, x in d.items() if k != "model_type"})

        self.vision_config = sub(AriaVisionHFConfig, vision_config)
        self.text_config = sub(AriaMoELMHFConfig, text_config)
        p2q = projector_patch_to_query_dict or {1225: 128, 4900: 256}
        self.projector_patch_to_query_dict = {int(k): int(v) for k, v in p2q.items()}
        self.ignore_index = ignore_index
        self.image_token_index = image_token_index
        kwargs.setdefault("tie_word_embeddings", False)
        super().__init__(**kwargs)

    def to_dict(self):
        d = super().to_dict()
        d["projector_patch_to_query_dict"] = {str(k): v for k, v in self.projector_patch_to_query_dict.items()}  # JSON keys are strings
        return d

    def to_native(self) -> native.AriaConfig:
        t = {k: getattr(self.text_config, k) for k in _TEXT_KEYS}
        v = {k: getattr(self.vision_config, k) for k in _VISION_KEYS}
        return native.AriaConfig(vision_config=v, text_config=t, projector_patch_to_query_dict=dict(self.projector_patch_to_query_dict),
                                 ignore_index=self.ignore_index, image_token_index=self.image_token_index)

    # the generation utilities look these up on the top-level config
    @property
    def vocab_size(self):
        return self.text_config.vocab_size

    @property
    def num_hidden_layers(self):
        return self.text_config.num_hidden_layers

    @property
    def hidden_size(self):
        return self.text_config.hidden_size


class _CacheRow:
    """One sequence of a cached batch: its static K/V tensors per layer and the decode engine whose pointer table names them."""

    def __init__(self, kv, engine=None):
        self.kv, self.engine, self.pads = kv, engine, 0


class AriaStaticKVCache:
    """``past_key_values`` of the native HF class: the gptfast twin's static KV cache (``setup_caches`` gptfast/model.py:113-166, ``KVCache``
    :67-93) and how many positions of it are filled.  Exposes what ``GenerationMixin`` asks of a cache object (``get_seq_length``,
    ``is_compileable``, ``get_max_cache_shape``, ``reorder_cache``, ``crop``).  Row 0 is the twin's own cache (``twin.llm.layers[i].attention
    .kv_cache``); further rows of a batch are added by ``ensure_rows`` (kept on the twin between calls of equal geometry).  ``seen`` counts
    COLUMNS of ``input_ids`` consumed (left padding included, what HF's ``cache_position`` counts); row r holds ``seen - rows[r].pads``
    positions."""

    is_compileable = False   # the decode step is already one native call per token; there is nothing for torch.compile to capture
    is_sliding = [False]

    def __init__(self, twin, max_len: int, batch: int = 1):
        self.twin, self.max_len, self.seen = twin, int(max_len), 0
        twin.setup_caches(1, self.max_len)
        llm = twin.llm
        pool = getattr(twin, "_hf_cache_rows", None)
        row0_kv = [b.attention.kv_cache for b in llm.layers]
        if pool is None or pool[0].kv[0] is not row0_kv[0]:          # new geometry (setup_caches re-allocated): rows of the old one are stale
            pool = [_CacheRow(row0_kv, llm._engine)]
            twin._hf_cache_rows = pool
        self.rows = pool
        self.ensure_rows(batch)
        for r in self.rows:
            r.pads = 0
        # Rows beyond the first are B - 1 full per-layer caches (19 GB each at 64K positions) + an engine each.  They stay pooled on the twin
        # for the next call of equal geometry only while they are small; a pool beyond ARIA_HF_CACHE_POOL_GB (default 8) is released when
        # this cache object is dropped.
        import os
        import weakref

        token = object()
        twin._hf_cache_owner = token
        limit = float(os.environ.get("ARIA_HF_CACHE_POOL_GB", "8")) * 1e9
        weakref.finalize(self, AriaStaticKVCache._trim_pool, weakref.ref(twin), token, limit)

    @staticmethod
    def _trim_pool(twin_ref, token, limit_bytes):
        twin = twin_ref()
        pool = getattr(twin, "_hf_cache_rows", None) if twin is not None else None
        if not pool or getattr(twin, "_hf_cache_owner", None) is not token:   # a newer cache object uses the pool now
            return
        extra = sum(kv.k.numel() * kv.k.element_size() * 2 for row in pool[1:] for kv in row.kv)
        if extra > limit_bytes:
            del pool[1:]

    def ensure_rows(self, batch: int):
        from . import gptfast as G

        llm = self.twin.llm
        c, a0 = llm.config, llm.layers[0].attention
        dev = llm.output.weight.device
        while len(self.rows) < batch:
            self.rows.append(_CacheRow([G.KVCache(1, llm.max_seq_length, c.n_head, a0.hdp, dev) for _ in llm.layers]))
        self.batch = batch

    def bind(self, r: int):
        """Context: the twin's attention layers (and its decode engine) point at row ``r``; row 0 is restored on exit."""
        cache = self

        class _Bind:
            def __enter__(self_b):
                cache._attach(cache.rows[r])

            def __exit__(self_b, *exc):
                cache.rows[r].engine = cache.twin.llm._engine     # (created lazily by the first single-token step of the row)
                cache._attach(cache.rows[0])
                return False

        return _Bind()

    def _attach(self, row):
        llm = self.twin.llm
        for b, kv in zip(llm.layers, row.kv):
            b.attention.kv_cache = kv
        llm._engine = row.engine

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.seen

    def get_max_cache_shape(self, layer_idx: int = 0) -> int:
        return self.max_len

    get_max_length = get_max_cache_shape

    def __len__(self):
        return len(self.twin.llm.layers)

    def reorder_cache(self, beam_idx):
        """Beam search: row i continues the hypothesis row ``beam_idx[i]`` held -- copy the filled part of the source's K/V (two rows may
        descend from one parent, so sources are snapshotted before anything is overwritten)."""
        idx = [int(i) for i in beam_idx.tolist()]
        if len(idx) != self.batch:
            raise ValueError(f"reorder_cache: {len(idx)} beams for a cache of {self.batch} rows")
        moves = [(dst, src) for dst, src in enumerate(idx) if dst != src]
        if not moves:
            return
        snap = {}
        for _, src in moves:
            if src not in snap:
                n = self.seen - self.rows[src].pads
                snap[src] = (self.rows[src].pads, [(kv.k[:, :n].clone(), kv.v[:, :n].clone()) for kv in self.rows[src].kv])
        for dst, src in moves:
            pads, layers = snap[src]
            self.rows[dst].pads = pads
            for kv, (k, v) in zip(self.rows[dst].kv, layers):
                kv.k[:, :k.shape[1]].copy_(k)
                kv.v[:, :v.shape[1]].copy_(v)

    def crop(self, max_length: int):
        """Forget positions >= max_length (assisted decoding rolls back rejected tokens): the rows stay in the buffer and are overwritten."""
        self.seen = min(self.seen, max_length if max_length >= 0 else self.seen + max_length)


class AriaPretrainedModel(PreTrainedModel):
    """modeling_aria.py:38-58."""

    config_class = AriaHFConfig
    config: AriaHFConfig
    base_model_prefix = "model"
    _no_split_modules = ["MoEDecoderLayer", "VisionEncoderLayer"]
    supports_gradient_checkpointing = True
    _skip_keys_device_placement = "past_key_values"
    _supports_flash_attn = True       # the attention IS a flash kernel (attn.hip); the flags only tell HF not to reject the config
    _supports_sdpa = True
    _supports_cache_class = True      # modeling_aria.py:43-58; the cache object is AriaStaticKVCache (module docstring)
    _supports_static_cache = True
    main_input_name = "input_ids"

    def _init_weights(self, module):
        """modeling_aria.py:60-88 (normal(0, initializer_range) for Linear / Embedding, ones for norms) + the tensors the reference
        leaves as ``torch.empty`` garbage (router / expert weights): N(0, 0.02) too."""
        std = 0.02
        for name, p in module.named_parameters(recurse=False):
            with torch.no_grad():
                if "norm" in type(module).__name__.lower() and name == "weight":
                    p.fill_(1.0)
                elif name == "bias":
                    p.zero_()
                else:
                    p.normal_(0.0, std)


class AriaForConditionalGeneration(AriaPretrainedModel, GenerationMixin):
    """modeling_aria.py:125-365 on the native modules.  Sub-module names, parameter names and shapes are the reference's."""

    def __init__(self, config: AriaHFConfig):
        super().__init__(config)
        ncfg = config.to_native()
        self.native_config = ncfg
        core = native.AriaForConditionalGeneration(ncfg)
        self.vision_tower = core.vision_tower
        self.multi_modal_projector = core.multi_modal_projector
        self.language_model = core.language_model
        self.vocab_size = ncfg.text_config.vocab_size
        self.post_init()

    # ---- the native implementation's methods, bound to this class (they only use the three sub-modules and .config fields)
    freeze_vit = native.AriaForConditionalGeneration.freeze_vit
    freeze_projector = native.AriaForConditionalGeneration.freeze_projector
    freeze_llm = native.AriaForConditionalGeneration.freeze_llm
    set_moe_z_loss_coeff = native.AriaForConditionalGeneration.set_moe_z_loss_coeff
    set_moe_aux_loss_coeff = native.AriaForConditionalGeneration.set_moe_aux_loss_coeff
    image_features = native.AriaForConditionalGeneration.image_features
    enable_expert_parallel = native.AriaForConditionalGeneration.enable_expert_parallel

    def get_input_embeddings(self):
        return self.language_model.get_input_embeddings()

    def get_output_embeddings(self):
        return self.language_model.lm_head

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None):
        self.native_config.text_config.gradient_checkpointing = True   # per-layer recompute inside the fused decoder node

    def gradient_checkpointing_disable(self):
        self.native_config.text_config.gradient_checkpointing = False

    @property
    def is_gradient_checkpointing(self) -> bool:
        return bool(self.native_config.text_config.gradient_checkpointing)

    def _core(self) -> native.AriaForConditionalGeneration:
        """A native model object sharing this module's sub-modules (no copy): the fast generate / gptfast bridge live on it."""
        core = getattr(self, "_native_core", None)
        if core is None:
            core = native.AriaForConditionalGeneration.__new__(native.AriaForConditionalGeneration)
            torch.nn.Module.__init__(core)
            core.config = self.native_config
            core.vision_tower, core.multi_modal_projector, core.language_model = self.vision_tower, self.multi_modal_projector, self.language_model
            core.vocab_size = self.vocab_size
            object.__setattr__(self, "_native_core", core)
        return core

    def generate_fast(self, *args, **kwargs):
        """The native generate (gptfast twin + one-call-per-token decode engine, gptfast/generate.py:112-177)."""
        return self._core().generate(*args, **kwargs)

    def _twin(self):
        """The gptfast twin behind the KV cache (``native.to_gptfast``: the reference's own checkpoint conversion), rebuilt when a parameter
        has been written since it was made (an optimizer step bumps the tensors' version counters)."""
        core = self._core()
        stamp = sum(p._version for p in self.parameters())
        if getattr(core, "_gptfast_twin", None) is None or getattr(self, "_twin_stamp", None) != stamp:
            object.__setattr__(core, "_gptfast_twin", core.to_gptfast())
            object.__setattr__(core, "_gptfast_decoder", None)
            object.__setattr__(self, "_twin_stamp", stamp)
        return core._gptfast_twin

    def make_cache(self, max_len: int, batch: int = 1) -> AriaStaticKVCache:
        """A fresh ``past_key_values`` object for ``batch`` sequences of up to ``max_len`` positions (prompt + new tokens) each."""
        return AriaStaticKVCache(self._twin(), max_len, batch)

    @staticmethod
    def left_pads(attention_mask: Optional[torch.Tensor], B: int, T: int):
        """Per-row count of leading padding columns, or None when the mask is not left padding (a zero after a one)."""
        if attention_mask is None:
            return [0] * B
        m = attention_mask[:, :T].ne(0)
        pads = (~m).sum(dim=1)
        ok = (m.long().cumsum(dim=1) == (torch.arange(T, device=m.device)[None, :] + 1 - pads[:, None]).clamp(min=0)).all()
        return [int(p) for p in pads.tolist()] if bool(ok) else None

    def _forward_cached(self, cache: AriaStaticKVCache, input_ids, pixel_values, pixel_mask, keep: int, attention_mask=None) -> torch.Tensor:
        """One step of cached generation (modeling_aria.py:337-365 + gptfast/generate.py:71-110): the first call prefills every row's prompt
        (image features merged once; a row identical to the one before it -- beams, ``num_return_sequences`` -- is copied, not recomputed),
        later calls feed the new token of every row to the decode engine."""
        twin = cache.twin
        if input_ids is None or input_ids.dim() != 2:
            raise NotImplementedError("cached generation takes input_ids [B, T]; pass use_cache=False for other inputs")
        B, T = input_ids.shape
        if cache.seen + T > cache.max_len:
            raise ValueError(f"AriaStaticKVCache of {cache.max_len} positions cannot take {cache.seen} + {T}")
        dev = input_ids.device
        V = self.vocab_size
        with torch.no_grad():
            if cache.seen == 0:                                   # prefill: ViT + projector once per row, its prompt positions into its cache
                pads = self.left_pads(attention_mask, B, T)
                if pads is None or max(pads) >= T:
                    raise NotImplementedError("cached generation serves unpadded or LEFT-padded prompts (HF's convention for decoder-only "
                                              "generation); pass use_cache=False for other masks")
                cache.ensure_rows(B)
                n_img = [0] * B
                if pixel_values is not None:
                    side = pixel_values.shape[-1] // self.native_config.vision_config.patch_size
                    q = self.native_config.projector_patch_to_query_dict[side * side]
                    n_img = [int(c) // q for c in (input_ids == self.config.image_token_index).sum(dim=1).tolist()]
                    if sum(n_img) != pixel_values.shape[0]:
                        raise ValueError("Image features and image tokens do not match")          # modeling_aria.py:267-271
                rows_logits, first = [], 0
                for r in range(B):
                    ids_r = input_ids[r:r + 1, pads[r]:]
                    pv = pixel_values[first:first + n_img[r]] if n_img[r] else None
                    pm = pixel_mask[first:first + n_img[r]] if (n_img[r] and pixel_mask is not None) else None
                    same = (r > 0 and pads[r] == pads[r - 1] and n_img[r] == n_img[r - 1] and torch.equal(ids_r, input_ids[r - 1:r, pads[r - 1]:])
                            and (pv is None or (torch.equal(pv, pixel_values[first - n_img[r]:first])
                                                and (pm is None or torch.equal(pm, pixel_mask[first - n_img[r]:first])))))
                    first += n_img[r]
                    cache.rows[r].pads = pads[r]
                    if same:                                      # an expanded copy of the previous row: its cache rows and logits are the same
                        n = T - pads[r]
                        for kv, src in zip(cache.rows[r].kv, cache.rows[r - 1].kv):
                            kv.k[:, :n].copy_(src.k[:, :n]), kv.v[:, :n].copy_(src.v[:, :n])
                        rows_logits.append(rows_logits[-1])
                        continue
                    with cache.bind(r):
                        emb = twin.prepare_embeddings(ids_r, pv, pm)
                        lg = twin(None, torch.arange(T - pads[r], device=dev), emb, last_only=(keep == 1))
                    if keep != 1 and pads[r]:                     # full-length logits: pad columns read zero
                        lg = torch.cat([lg.new_zeros((1, pads[r], V)), lg], dim=1)
                    rows_logits.append(lg)
                logits = torch.cat(rows_logits, dim=0)
            elif T == 1:                                          # decode: the native one-call-per-token engine, row by row
                if B != cache.batch:
                    raise ValueError(f"AriaStaticKVCache holds {cache.batch} rows, input_ids has {B}")
                out = torch.empty((B, 1, V), dtype=torch.bfloat16, device=dev)
                for r in range(B):
                    with cache.bind(r):
                        out[r] = twin(input_ids[r:r + 1], torch.tensor([cache.seen - cache.rows[r].pads], dtype=torch.int32, device=dev))[0]
                logits = out
            else:
                raise NotImplementedError("several new tokens against a filled cache (assisted / speculative decoding): use use_cache=False")
        cache.seen += T
        return logits if keep != 1 else logits[:, -1:]

    def forward(self, input_ids: Optional[torch.Tensor] = None, pixel_values: Optional[torch.Tensor] = None,
                pixel_mask: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None, position_ids=None,
                past_key_values=None, inputs_embeds: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
                use_cache: Optional[bool] = None, output_attentions=None, output_hidden_states=None, return_dict=None,
                num_logits_to_keep: int = 0, logits_to_keep: int = 0, cache_position=None, **kwargs) -> CausalLMOutputWithPast:
        """modeling_aria.py:194-335 (same keyword arguments).  ``past_key_values``: an ``AriaStaticKVCache`` (``make_cache`` / created by
        ``generate``) selects cached generation; None (or an empty HF cache object) the full forward."""
        if output_attentions or output_hidden_states:
            raise NotImplementedError("output_attentions / output_hidden_states: the flash kernels never materialise them")
        keep = int(num_logits_to_keep or logits_to_keep or 0)
        self.native_config.image_token_index = self.config.image_token_index
        if isinstance(past_key_values, AriaStaticKVCache):
            if labels is not None or inputs_embeds is not None or self.training:
                raise NotImplementedError("AriaStaticKVCache is an inference cache: no labels / inputs_embeds / training mode")
            logits = self._forward_cached(past_key_values, input_ids, pixel_values, pixel_mask, keep, attention_mask)
            return CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=past_key_values)
        if past_key_values is not None and not (hasattr(past_key_values, "get_seq_length") and past_key_values.get_seq_length() == 0):
            raise NotImplementedError("aria_amd HF surface: past_key_values must be an AriaStaticKVCache (model.make_cache(max_len))")
        out = native.AriaForConditionalGeneration.forward(self._core(), input_ids=input_ids, pixel_values=pixel_values, pixel_mask=pixel_mask,
                                                          attention_mask=attention_mask, inputs_embeds=inputs_embeds, labels=labels,
                                                          num_logits_to_keep=keep, return_logits=True if labels is None else None,
                                                          validate_image_tokens=kwargs.pop("validate_image_tokens", True))
        # (the Trainer scales the loss in place; the native loss is a view produced by a custom autograd node)
        return CausalLMOutputWithPast(loss=None if out.loss is None else out.loss.clone(), logits=out.logits, past_key_values=None)

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, pixel_values=None,
                                      pixel_mask=None, **kwargs):
        """modeling_aria.py:337-365: with a filled cache only the tokens it has not seen go in, and the pixel inputs only with the prompt;
        without a cache every step sees the whole sequence.  Always asks for the last position's logits only."""
        if isinstance(past_key_values, AriaStaticKVCache):
            seen = past_key_values.seen
            return {"input_ids": input_ids[:, seen:], "attention_mask": attention_mask, "past_key_values": past_key_values,
                    "pixel_values": pixel_values if seen == 0 else None, "pixel_mask": pixel_mask if seen == 0 else None,
                    "num_logits_to_keep": 1, "use_cache": True}
        return {"input_ids": input_ids, "attention_mask": attention_mask, "pixel_values": pixel_values, "pixel_mask": pixel_mask,
                "num_logits_to_keep": 1, "use_cache": False}

    def generate(self, *args, **kwargs):
        """HF ``generate`` with the static KV cache whenever the decode engine can serve the request (any batch, unpadded or left-padded
        prompts, sampling / greedy / beam search, eval mode); ``use_cache=False``, assisted decoding or other masks run cache-free."""
        use_cache = kwargs.pop("use_cache", True)
        input_ids = kwargs.get("input_ids", args[0] if args else kwargs.get("inputs"))
        am = kwargs.get("attention_mask")
        gc = kwargs.get("generation_config") or self.generation_config
        beams = kwargs.get("num_beams", getattr(gc, "num_beams", 1)) or 1
        servable = (use_cache is not False and kwargs.get("past_key_values") is None and torch.is_tensor(input_ids) and input_ids.dim() == 2
                    and input_ids.shape[0] >= 1 and not self.training and kwargs.get("assistant_model") is None)
        if servable and am is None:
            # no mask given: HF will INFER one from the pad tokens (GenerationMixin._prepare_attention_mask_for_generation: inputs != pad when a
            # pad token is set, occurs in the prompt and is not also an eos token) -- decide on that mask, not on "no mask" (a
            # right-padded prompt without a mask raised NotImplementedError in the middle of generation instead of running cache-free)
            pad = kwargs.get("pad_token_id", getattr(gc, "pad_token_id", None))
            eos = kwargs.get("eos_token_id", getattr(gc, "eos_token_id", None))
            eos = [] if eos is None else ([int(e) for e in eos] if isinstance(eos, (list, tuple)) else [int(eos)])
            if pad is not None and int(pad) not in eos and bool((input_ids == int(pad)).any()):
                am = input_ids.ne(int(pad)).long()
        if servable:
            pads = self.left_pads(am, *input_ids.shape)
            servable = pads is not None and max(pads) < input_ids.shape[1]
        if servable:
            T = input_ids.shape[1]
            nret = kwargs.get("num_return_sequences", getattr(gc, "num_return_sequences", 1)) or 1
            rows = input_ids.shape[0] * max(int(beams), 1) * (int(nret) if int(beams) == 1 else 1)
            max_new = kwargs.get("max_new_tokens", getattr(gc, "max_new_tokens", None))
            max_len = kwargs.get("max_length", getattr(gc, "max_length", None))
            total = T + int(max_new) if max_new is not None else max(int(max_len or 0), T + 1)
            kwargs["past_key_values"] = self.make_cache(total + 1, rows)
            kwargs["use_cache"] = True
        elif not isinstance(kwargs.get("past_key_values"), AriaStaticKVCache):
            kwargs["use_cache"] = False
        return super().generate(*args, **kwargs)

    def _supports_default_dynamic_cache(self) -> bool:  # GenerationMixin: do not build a DynamicCache for us (ours is AriaStaticKVCache)
        return False


def register() -> None:
    """``AutoConfig`` / ``AutoModelForCausalLM`` resolve the reference's model types to these classes.  transformers >= 4.48 ships its own
    port of Aria under the same ``model_type = "aria"`` (different module / parameter names: it cannot read the reference's checkpoints
    of this layout); importing ``aria_amd.hf`` overrides that mapping for the process, like the reference's ``trust_remote_code`` does."""
    for cfg in (AriaMoELMHFConfig, AriaVisionHFConfig, AriaHFConfig):
        AutoConfig.register(cfg.model_type, cfg, exist_ok=True)
    AutoModelForCausalLM.register(AriaHFConfig, AriaForConditionalGeneration, exist_ok=True)


register()
