"""Checkpoint / wire formats: real ``rhymes-ai/Aria`` weights load into the modules of this package,
both ways, without any layout surprise.

* HF layout  = the ``state_dict()`` of the reference's ``AriaForConditionalGeneration`` (``language_model.model.layers.{i}...``,
  ``vision_tower...``, ``multi_modal_projector...``), stored as sharded ``*.safetensors`` + ``model.safetensors.index.json`` (or
  ``pytorch_model*.bin`` + index) -- what ``aria_amd.modeling_aria.AriaForConditionalGeneration`` and ``aria_amd.train`` consume.
* gptfast layout = ``model.pth`` written by the reference's ``gptfast/scripts/convert_hf_checkpoint.py:90-162``: ``llm.*`` names, q/k rows
  permuted for interleaved-pair RoPE (:110-116), q/k/v fused into ``wqkv`` (:145-153), experts ``fc1 [E,K,2I]`` split into ``w1``/``w3``
  ``[E,I,K]`` and ``fc2 [E,I,D]`` stored as ``w2 [E,D,I]`` (:154-162); vision tower / projector keys unchanged -- what
  ``aria_amd.gptfast.Aria`` consumes (``load_model_pth``).

Everything here is pure layout (bit-exact, CPU tensors); the converters are each other's inverse and are pinned against the conversion
that the reference's own gptfast model accepted (tests/golden/gptfast.pt, tests/test_checkpoint.py).
"""
from __future__ import annotations

import json
import os
import re
from typing import Dict, Optional

import torch

_LAYER_MAP = {  # HF suffix -> gptfast suffix (inside "language_model.model.layers.{i}." / "llm.layers.{i}.")
    "self_attn.o_proj.weight": "attention.wo.weight",
    "mlp.router.weight": "feed_forward.gate.weight",
    "mlp.shared_experts.gate_proj.weight": "feed_forward.shared_ffn.w1.weight",
    "mlp.shared_experts.up_proj.weight": "feed_forward.shared_ffn.w3.weight",
    "mlp.shared_experts.down_proj.weight": "feed_forward.shared_ffn.w2.weight",
    "input_layernorm.weight": "attention_norm.weight",
    "post_attention_layernorm.weight": "ffn_norm.weight",
}
_TOP_MAP = {
    "language_model.model.embed_tokens.weight": "llm.tok_embeddings.weight",
    "language_model.model.norm.weight": "llm.norm.weight",
    "language_model.lm_head.weight": "llm.output.weight",
}
_HF_LAYER = re.compile(r"^language_model\.model\.layers\.(\d+)\.(.+)$")
_GF_LAYER = re.compile(r"^llm\.layers\.(\d+)\.(.+)$")


def _permute_qk(w: torch.Tensor, n_head: int, head_dim: int) -> torch.Tensor:
    """half-split RoPE rows -> interleaved-pair rows (convert_hf_checkpoint.py:110-116)"""
    return w.view(n_head, 2, head_dim // 2, w.shape[1]).transpose(1, 2).reshape(n_head * head_dim, w.shape[1])


def _unpermute_qk(w: torch.Tensor, n_head: int, head_dim: int) -> torch.Tensor:
    return w.view(n_head, head_dim // 2, 2, w.shape[1]).transpose(1, 2).reshape(n_head * head_dim, w.shape[1])


def hf_to_gptfast(sd: Dict[str, torch.Tensor], n_head: int, head_dim: int, n_kv_head: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """The reference's HF -> gptfast conversion for a complete Aria state dict (vision / projector keys pass through)."""
    n_kv_head = n_kv_head or n_head
    out: Dict[str, torch.Tensor] = {}
    layers: Dict[int, Dict[str, torch.Tensor]] = {}
    for key, val in sd.items():
        if key.startswith(("vision_tower.", "multi_modal_projector.")):
            out[key] = val
        elif key in _TOP_MAP:
            out[_TOP_MAP[key]] = val
        else:
            m = _HF_LAYER.match(key)
            if not m:
                raise KeyError(f"hf_to_gptfast: unexpected key {key}")
            if m.group(2).endswith("rotary_emb.inv_freq"):
                continue  # recomputed (:97)
            layers.setdefault(int(m.group(1)), {})[m.group(2)] = val
    for i, lw in layers.items():
        d = f"llm.layers.{i}."
        q = _permute_qk(lw.pop("self_attn.q_proj.weight"), n_head, head_dim)
        k = _permute_qk(lw.pop("self_attn.k_proj.weight"), n_kv_head, head_dim)
        out[d + "attention.wqkv.weight"] = torch.cat([q, k, lw.pop("self_attn.v_proj.weight")])
        w1, w3 = torch.chunk(lw.pop("mlp.experts.fc1.weight"), 2, dim=-1)
        out[d + "feed_forward.cond_ffn.w1"] = w1.transpose(1, 2).contiguous()
        out[d + "feed_forward.cond_ffn.w3"] = w3.transpose(1, 2).contiguous()
        out[d + "feed_forward.cond_ffn.w2"] = lw.pop("mlp.experts.fc2.weight").transpose(1, 2).contiguous()
        for suffix, val in lw.items():
            if suffix not in _LAYER_MAP:
                raise KeyError(f"hf_to_gptfast: unexpected key language_model.model.layers.{i}.{suffix}")
            out[d + _LAYER_MAP[suffix]] = val
    return out


def gptfast_to_hf(sd: Dict[str, torch.Tensor], n_head: int, head_dim: int, n_kv_head: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Inverse of :func:`hf_to_gptfast` (fine-tuned gptfast weights back into the HF / training layout)."""
    n_kv_head = n_kv_head or n_head
    inv_layer = {v: k for k, v in _LAYER_MAP.items()}
    inv_top = {v: k for k, v in _TOP_MAP.items()}
    out: Dict[str, torch.Tensor] = {}
    for key, val in sd.items():
        if key.startswith(("vision_tower.", "multi_modal_projector.")):
            out[key] = val
            continue
        if key in inv_top:
            out[inv_top[key]] = val
            continue
        m = _GF_LAYER.match(key)
        if not m:
            raise KeyError(f"gptfast_to_hf: unexpected key {key}")
        s, suffix = f"language_model.model.layers.{m.group(1)}.", m.group(2)
        if suffix == "attention.wqkv.weight":
            nq, nk = n_head * head_dim, n_kv_head * head_dim
            out[s + "self_attn.q_proj.weight"] = _unpermute_qk(val[:nq], n_head, head_dim).contiguous()
            out[s + "self_attn.k_proj.weight"] = _unpermute_qk(val[nq:nq + nk], n_kv_head, head_dim).contiguous()
            out[s + "self_attn.v_proj.weight"] = val[nq + nk:].contiguous()
        elif suffix == "feed_forward.cond_ffn.w1":
            w3 = sd[f"llm.layers.{m.group(1)}.feed_forward.cond_ffn.w3"]
            out[s + "mlp.experts.fc1.weight"] = torch.cat([val.transpose(1, 2), w3.transpose(1, 2)], dim=-1).contiguous()
        elif suffix == "feed_forward.cond_ffn.w3":
            pass  # consumed with w1
        elif suffix == "feed_forward.cond_ffn.w2":
            out[s + "mlp.experts.fc2.weight"] = val.transpose(1, 2).contiguous()
        elif suffix in inv_layer:
            out[s + inv_layer[suffix]] = val
        else:
            raise KeyError(f"gptfast_to_hf: unexpected key {key}")
    return out


# ------------------------------------------------------------------------------------------------ files
def load_checkpoint_dir(path: str) -> Dict[str, torch.Tensor]:
    """All tensors of an HF checkpoint directory: ``model.safetensors.index.json`` / ``pytorch_model.bin.index.json`` shards
    (convert_hf_checkpoint.py:63-88 looks for the same two), a single ``model.safetensors`` / ``pytorch_model.bin``, or a gptfast
    ``model.pth``."""
    def read(file):
        if file.endswith(".safetensors"):
            from safetensors.torch import load_file

            return load_file(file, device="cpu")
        return torch.load(file, map_location="cpu", mmap=True, weights_only=True)

    for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
        ipath = os.path.join(path, index)
        if os.path.exists(ipath):
            with open(ipath) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
            out: Dict[str, torch.Tensor] = {}
            for name in files:
                out.update(read(os.path.join(path, name)))
            return out
    for single in ("model.safetensors", "pytorch_model.bin", "model.pth"):
        if os.path.exists(os.path.join(path, single)):
            return read(os.path.join(path, single))
    raise FileNotFoundError(f"no checkpoint index or weight file under {path}")


def iter_checkpoint_shards(path: str):
    """The same files as :func:`load_checkpoint_dir`, one state-dict shard at a time (a 25 B-parameter checkpoint is ~50 GB: the loader
    below never holds more than one shard on the host)."""
    def read(file):
        if file.endswith(".safetensors"):
            from safetensors.torch import load_file

            return load_file(file, device="cpu")
        return torch.load(file, map_location="cpu", mmap=True, weights_only=True)

    for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
        ipath = os.path.join(path, index)
        if os.path.exists(ipath):
            with open(ipath) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
            for name in files:
                yield read(os.path.join(path, name))
            return
    for single in ("model.safetensors", "pytorch_model.bin", "model.pth"):
        if os.path.exists(os.path.join(path, single)):
            yield read(os.path.join(path, single))
            return
    raise FileNotFoundError(f"no checkpoint index or weight file under {path}")


def load_hf_dir_into(model: torch.nn.Module, path: str, strict: bool = True):
    """:func:`load_hf_into` shard by shard.  Returns (missing, unexpected)."""
    own = model.state_dict()
    seen, unexpected = set(), []
    with torch.no_grad():
        for shard in iter_checkpoint_shards(path):
            for k, v in shard.items():
                if k in own:
                    if own[k].shape != v.shape:
                        raise ValueError(f"{k}: checkpoint {tuple(v.shape)} vs module {tuple(own[k].shape)}")
                    own[k].copy_(v.to(own[k].dtype))
                    seen.add(k)
                elif not k.endswith("rotary_emb.inv_freq"):
                    unexpected.append(k)
            del shard
    missing = [k for k in own if k not in seen]
    if strict and (missing or unexpected):
        raise KeyError(f"load_hf_dir_into: missing {missing[:4]} unexpected {unexpected[:4]}")
    return missing, unexpected


def save_checkpoint_dir(sd: Dict[str, torch.Tensor], path: str, max_shard_bytes: int = 5 << 30) -> None:
    """Sharded safetensors + index in the HF convention (``model-0000k-of-0000n.safetensors``, ``metadata.total_size``)."""
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    shards, cur, cur_bytes = [], {}, 0
    for key in sorted(sd):
        t = sd[key].detach().cpu().contiguous()
        nbytes = t.numel() * t.element_size()
        if cur and cur_bytes + nbytes > max_shard_bytes:
            shards.append(cur)
            cur, cur_bytes = {}, 0
        cur[key] = t
        cur_bytes += nbytes
    if cur:
        shards.append(cur)
    weight_map, total = {}, 0
    for n, shard in enumerate(shards, 1):
        name = "model.safetensors" if len(shards) == 1 else f"model-{n:05d}-of-{len(shards):05d}.safetensors"
        save_file(shard, os.path.join(path, name), metadata={"format": "pt"})
        for key, t in shard.items():
            weight_map[key] = name
            total += t.numel() * t.element_size()
    with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2)


def convert_hf_checkpoint(checkpoint_dir: str, n_head: int = 20, head_dim: int = 128, n_kv_head: Optional[int] = None) -> str:
    """``python gptfast/scripts/convert_hf_checkpoint.py --checkpoint_dir D``: read the HF shards, write ``D/model.pth``."""
    out = hf_to_gptfast(load_checkpoint_dir(checkpoint_dir), n_head, head_dim, n_kv_head)
    target = os.path.join(checkpoint_dir, "model.pth")
    torch.save(out, target)
    return target


def load_hf_into(model: torch.nn.Module, sd: Dict[str, torch.Tensor], strict: bool = True):
    """Copy an HF-layout state dict into ``aria_amd.modeling_aria.AriaForConditionalGeneration`` (or the text model alone when the
    keys lack the ``language_model.`` prefix).  Returns (missing, unexpected) like ``load_state_dict``."""
    own = model.state_dict()
    ignorable = lambda k: k.endswith("rotary_emb.inv_freq")  # noqa: E731  (recomputed buffers)
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own and not ignorable(k)]
    if strict and (missing or unexpected):
        raise KeyError(f"load_hf_into: missing {missing[:4]} unexpected {unexpected[:4]}")
    with torch.no_grad():
        for k, v in own.items():
            if k in sd:
                if v.shape != sd[k].shape:
                    raise ValueError(f"{k}: checkpoint {tuple(sd[k].shape)} vs module {tuple(v.shape)}")
                v.copy_(sd[k].to(v.dtype))
    return missing, unexpected
