"""MI355X-native mirror of the reference's vision tower and projector.

``AriaVisionModel``  <- aria/model/vision_encoder.py:70-152 (Idefics2 NaViT-style SigLIP transformer without post-LN,
                        transformers/models/idefics2/modeling_idefics2.py:130-173, 203-278, 330-363)
``AriaProjector``    <- aria/model/projector.py:26-189 (cross-attention resampler + FFN, incl. the q/k/v_proj THEN
                        nn.MultiheadAttention in/out-proj double projection)

Parameter names follow the reference state dict (``vision_model.embeddings.patch_embedding.weight [hidden,3,14,14]``,
``vision_model.encoder.layers.{i}.{layer_norm1,self_attn.{q,k,v,out}_proj,layer_norm2,mlp.fc1,mlp.fc2}.{weight,bias}``,
``cross_attn.multihead_attn.in_proj_weight`` ...).  All arithmetic runs in libaria_hip.so.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import autograd as AG
from . import functional as Fn
from . import ops
from .moe_lm import Linear

bf16 = torch.bfloat16


class AriaVisionConfig:
    model_type = "aria_vision_model"

    def __init__(self, hidden_size=1152, num_hidden_layers=27, num_attention_heads=16, intermediate_size=4304, patch_size=14,
                 image_size=980, num_channels=3, layer_norm_eps=1e-6, **kwargs):
        self.hidden_size, self.num_hidden_layers, self.num_attention_heads = hidden_size, num_hidden_layers, num_attention_heads
        self.intermediate_size, self.patch_size, self.image_size = intermediate_size, patch_size, image_size
        self.num_channels, self.layer_norm_eps = num_channels, layer_norm_eps
        self.extra = kwargs


# --------------------------------------------------------------------------------------------- autograd pieces
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        y, mean, rstd = ops.layernorm(x, w, b, eps)
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dx, dw, db = ops.layernorm_bwd(AG._c(dy), x, w, mean, rstd)
        return dx, dw, db, None


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gelu_tanh(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_tanh_bwd(x, AG._c(dy))


class SdpaFn(torch.autograd.Function):
    """softmax(q k^T * scale + key mask) v on token-major [B*S, H*hd] tensors (non-causal; Sq may differ from Skv)."""

    @staticmethod
    def forward(ctx, q, k, v, B, Sq, Skv, H, hd, scale, key_mask):
        hdp = Fn._pad_hd(hd, need_bwd=any(ctx.needs_input_grad[:3]))
        if hdp != hd:
            q, k, v = (Fn._pad_heads(t, H, hd, hdp) for t in (q, k, v))
        o, lse = ops.attention_fwd(q, k, v, B, Sq, H, hdp, scale, False, key_mask=key_mask, Skv=Skv)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.meta = (B, Sq, Skv, H, hd, hdp, scale, key_mask)
        return Fn._unpad_heads(o, H, hd, hdp) if hdp != hd else o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        B, Sq, Skv, H, hd, hdp, scale, key_mask = ctx.meta
        do = AG._c(do)
        if hdp != hd:
            do = Fn._pad_heads(do, H, hd, hdp)
        dq, dk, dv = ops.attention_bwd(q, k, v, o, do, lse, B, Sq, H, hdp, scale, False, key_mask=key_mask, Skv=Skv)
        if hdp != hd:
            dq, dk, dv = (Fn._unpad_heads(t, H, hd, hdp) for t in (dq, dk, dv))
        return dq, dk, dv, None, None, None, None, None, None, None


class LayerNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, dtype=bf16))
        self.bias = nn.Parameter(torch.zeros(dim, dtype=bf16))
        self.eps = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shp = x.shape
        return LayerNormFn.apply(AG._c(x.reshape(-1, shp[-1])), self.weight, self.bias, self.eps).view(shp)


# --------------------------------------------------------------------------------------------- ViT
class PatchEmbedding(nn.Module):
    """Parameter holder shaped like nn.Conv2d(3, hidden, k=s=patch) (weight [hidden, 3, p, p], bias [hidden])."""

    def __init__(self, cfg: AriaVisionConfig):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cfg.hidden_size, cfg.num_channels, cfg.patch_size, cfg.patch_size, dtype=bf16))
        self.bias = nn.Parameter(torch.zeros(cfg.hidden_size, dtype=bf16))


class VisionEmbeddings(nn.Module):
    """Idefics2VisionEmbeddings: patch-embed conv as im2col + MFMA GEMM (K = 3*14*14 = 588 padded to 592), + bias,
    + bucketised position embedding (ids computed in fp32 on the device exactly like the reference's CPU path)."""

    def __init__(self, cfg: AriaVisionConfig):
        super().__init__()
        self.cfg = cfg
        self.patch_embedding = PatchEmbedding(cfg)
        self.num_patches_per_side = cfg.image_size // cfg.patch_size
        self.position_embedding = nn.Embedding(self.num_patches_per_side ** 2, cfg.hidden_size, dtype=bf16)
        self._bound = None

    def boundaries(self, device) -> torch.Tensor:
        if self._bound is None or self._bound.device != torch.device(device):
            n = self.num_patches_per_side
            self._bound = torch.arange(1 / n, 1.0, 1 / n, dtype=torch.float32).to(device)  # built on the host like the reference
        return self._bound

    def forward(self, pixel_values: torch.Tensor, patch_mask: torch.Tensor) -> torch.Tensor:
        cfg = self.cfg
        N, C, R, _ = pixel_values.shape
        K = C * cfg.patch_size * cfg.patch_size
        KP = (K + 7) // 8 * 8
        w = self.patch_embedding.weight.reshape(cfg.hidden_size, K)
        wp = torch.zeros((cfg.hidden_size, KP), dtype=bf16, device=w.device)
        wp[:, :K] = w
        px = pixel_values if pixel_values.dtype in (bf16, torch.float32) else pixel_values.float()
        patches = ops.vit_im2col(px.contiguous(), cfg.patch_size, KP)
        x = PatchEmbedFn.apply(patches, wp, self.patch_embedding.bias, self.patch_embedding.weight, K)
        ids = ops.vit_pos_ids(patch_mask, self.boundaries(x.device), self.num_patches_per_side)
        x = PosAddFn.apply(x, self.position_embedding.weight, ids.reshape(-1))
        return x.view(N, -1, cfg.hidden_size)


class PatchEmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, patches, wp, bias, w_orig, K):
        ctx.save_for_backward(patches, wp)
        ctx.K, ctx.wshape = K, w_orig.shape
        return ops.gemm(patches, wp, bias=bias)

    @staticmethod
    def backward(ctx, dy):
        patches, wp = ctx.saved_tensors
        dy = AG._c(dy)
        dwp = ops.gemm(dy, patches, a_oc=True, b_oc=True)
        return None, None, ops.colsum(dy), dwp[:, :ctx.K].reshape(ctx.wshape).contiguous(), None


class PosAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table, ids):
        ctx.save_for_backward(ids)
        ctx.tshape = table.shape
        return ops.gather_add_rows_(x.clone() if x.requires_grad else x, table, ids)

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        dt = torch.zeros(ctx.tshape, dtype=bf16, device=dy.device)
        ops.embedding_bwd(AG._c(dy), ids, dt)
        return dy, dt, None


class VisionAttention(nn.Module):
    def __init__(self, cfg: AriaVisionConfig):
        super().__init__()
        D = cfg.hidden_size
        self.num_heads, self.head_dim = cfg.num_attention_heads, D // cfg.num_attention_heads
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (Linear(D, D, bias=True) for _ in range(4))

    def forward(self, x: torch.Tensor, key_mask: Optional[torch.Tensor]) -> torch.Tensor:
        B, P, D = x.shape
        x2 = x.reshape(B * P, D)
        q, k, v = self.q_proj(x2), self.k_proj(x2), self.v_proj(x2)
        o = SdpaFn.apply(q, k, v, B, P, P, self.num_heads, self.head_dim, self.head_dim ** -0.5, key_mask)
        return self.out_proj(o).view(B, P, D)


class VisionMLP(nn.Module):
    def __init__(self, cfg: AriaVisionConfig):
        super().__init__()
        self.fc1 = Linear(cfg.hidden_size, cfg.intermediate_size, bias=True)
        self.fc2 = Linear(cfg.intermediate_size, cfg.hidden_size, bias=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shp = x.shape
        h = self.fc1(x.reshape(-1, shp[-1]))
        return self.fc2(GeluFn.apply(h)).view(shp)


class VisionEncoderLayer(nn.Module):
    def __init__(self, cfg: AriaVisionConfig):
        super().__init__()
        self.self_attn = VisionAttention(cfg)
        self.layer_norm1 = LayerNorm(cfg.hidden_size, cfg.layer_norm_eps)
        self.mlp = VisionMLP(cfg)
        self.layer_norm2 = LayerNorm(cfg.hidden_size, cfg.layer_norm_eps)

    def forward(self, x: torch.Tensor, key_mask: Optional[torch.Tensor]) -> torch.Tensor:
        x = AddFn.apply(x, self.self_attn(self.layer_norm1(x), key_mask))
        return AddFn.apply(x, self.mlp(self.layer_norm2(x)))

    def _fused_qkv(self):
        """[3D, D] weight / [3D] bias of the three projections, built once per weight version (frozen tower: once)."""
        a = self.self_attn
        ver = (a.q_proj.weight._version, a.k_proj.weight._version, a.v_proj.weight._version,
               a.q_proj.bias._version, a.k_proj.bias._version, a.v_proj.bias._version, a.q_proj.weight.data_ptr())
        if getattr(self, "_qkv_ver", None) != ver:
            self._qkv_w = torch.cat([a.q_proj.weight.detach(), a.k_proj.weight.detach(), a.v_proj.weight.detach()]).contiguous()
            self._qkv_b = torch.cat([a.q_proj.bias.detach(), a.k_proj.bias.detach(), a.v_proj.bias.detach()]).contiguous()
            self._qkv_ver = ver
        return self._qkv_w, self._qkv_b

    def forward_frozen(self, x2: torch.Tensor, B: int, P: int, key_mask: Optional[torch.Tensor]) -> torch.Tensor:
        """No-grad forward of the layer on a PRIVATE [B*P, D] buffer, updated in place (the recipe freezes the tower, config_full.yaml:39):
        one fused q/k/v GEMM, attention on strided views of its output, the two residual adds folded into the out_proj / fc2 GEMM
        epilogues (accumulate), GELU folded into the fc1 epilogue.  Same arithmetic as forward() except that `x + (h W^T + b)` is
        rounded to bf16 once instead of twice."""
        a, m = self.self_attn, self.mlp
        D, H, hd = x2.shape[1], a.num_heads, a.head_dim
        h, _, _ = ops.layernorm(x2, self.layer_norm1.weight, self.layer_norm1.bias, self.layer_norm1.eps, want_stats=False)
        wqkv, bqkv = self._fused_qkv()
        qkv = ops.gemm(h, wqkv, bias=bqkv)
        o, _ = ops.attention_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, P, H, hd, hd ** -0.5, False, key_mask=key_mask)
        ops.gemm(o, a.out_proj.weight, bias=a.out_proj.bias, out=x2, accumulate=True)
        h, _, _ = ops.layernorm(x2, self.layer_norm2.weight, self.layer_norm2.bias, self.layer_norm2.eps, want_stats=False)
        g = ops.gemm(h, m.fc1.weight, bias=m.fc1.bias, act="gelu_tanh")
        ops.gemm(g, m.fc2.weight, bias=m.fc2.bias, out=x2, accumulate=True)
        return x2


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(AG._c(a), AG._c(b))

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class VisionEncoder(nn.Module):
    def __init__(self, cfg: AriaVisionConfig):
        super().__init__()
        self.layers = nn.ModuleList([VisionEncoderLayer(cfg) for _ in range(cfg.num_hidden_layers)])


class VisionTransformer(nn.Module):
    """AriaVisionTransformer (vision_encoder.py:58-67): Idefics2VisionTransformer with post_layernorm = Identity."""

    def __init__(self, cfg: AriaVisionConfig):
        super().__init__()
        self.embeddings = VisionEmbeddings(cfg)
        self.encoder = VisionEncoder(cfg)


class AriaVisionModel(nn.Module):
    """forward(pixel_values [N,3,R,R], pixel_mask [N,R,R] bool) -> (last_hidden_state [N,P,hidden], image_atts [N,P] bool,
    True = padded patch) -- vision_encoder.py:94-152."""

    def __init__(self, cfg: AriaVisionConfig):
        super().__init__()
        self.config = cfg
        self.vision_model = VisionTransformer(cfg)

    def forward(self, pixel_values: torch.Tensor, pixel_mask: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        cfg = self.config
        N, _, R, _ = pixel_values.shape
        Hp = R // cfg.patch_size
        if pixel_mask is None:
            patch_mask = torch.ones((N, Hp, Hp), dtype=torch.uint8, device=pixel_values.device)
            key_mask, image_atts = None, None
        else:
            patch_mask = ops.vit_patch_mask(pixel_mask, cfg.patch_size)
            key_mask = patch_mask.view(N, Hp * Hp)
            image_atts = key_mask == 0
        x = self.vision_model.embeddings(pixel_values, patch_mask)
        layers = self.vision_model.encoder.layers
        frozen = not torch.is_grad_enabled() or not (x.requires_grad or any(p.requires_grad for p in layers.parameters()))
        if frozen and Fn._pad_hd(cfg.hidden_size // cfg.num_attention_heads, need_bwd=False) == cfg.hidden_size // cfg.num_attention_heads:
            Bn, P, D = x.shape
            x2 = x.detach().reshape(Bn * P, D).clone()  # private buffer: the layers update it in place
            for layer in layers:
                layer.forward_frozen(x2, Bn, P, key_mask)
            return x2.view(Bn, P, D), image_atts
        for layer in layers:
            x = layer(x, key_mask)
        return x, image_atts


# --------------------------------------------------------------------------------------------- projector
class FFN(nn.Module):
    """projector.py:26-45 (no biases, gelu_new)."""

    def __init__(self, embed_dim, ff_dim, output_dim):
        super().__init__()
        self.linear_in = Linear(embed_dim, ff_dim)
        self.linear_out = Linear(ff_dim, output_dim)

    def forward(self, x):
        shp = x.shape
        h = GeluFn.apply(self.linear_in(x.reshape(-1, shp[-1])))
        return self.linear_out(h).view(*shp[:-1], -1)


class MultiheadAttentionParams(nn.Module):
    """Parameter surface of nn.MultiheadAttention(embed_dim, num_heads): in_proj_weight [3E,E], in_proj_bias, out_proj."""

    def __init__(self, embed_dim: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim, dtype=bf16))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim, dtype=bf16))
        self.out_proj = Linear(embed_dim, embed_dim, bias=True)


class CrossAttention(nn.Module):
    """projector.py:48-102."""

    def __init__(self, kv_dim, embed_dim, num_heads, drop_out_rate=0):
        super().__init__()
        self.num_heads, self.embed_dim = num_heads, embed_dim
        self.q_proj = Linear(embed_dim, embed_dim)
        self.k_proj = Linear(kv_dim, embed_dim)
        self.v_proj = Linear(kv_dim, embed_dim)
        self.multihead_attn = MultiheadAttentionParams(embed_dim)
        self.linear = Linear(embed_dim, embed_dim, bias=True)
        self.layer_norm = LayerNorm(embed_dim)
        self.ln_kv = LayerNorm(kv_dim)

    def forward(self, x, hidden_states, key_mask=None, add_residual=False):
        B, P, _ = x.shape
        Q, E, H = hidden_states.shape[1], self.embed_dim, self.num_heads
        q = self.q_proj(self.layer_norm(hidden_states).reshape(B * Q, E))
        xn = self.ln_kv(x).reshape(B * P, -1)
        k, v = self.k_proj(xn), self.v_proj(xn)
        wi, bi = self.multihead_attn.in_proj_weight, self.multihead_attn.in_proj_bias
        q = AG.LinearFn.apply(q, wi[:E], bi[:E])
        k = AG.LinearFn.apply(k, wi[E:2 * E], bi[E:2 * E])
        v = AG.LinearFn.apply(v, wi[2 * E:], bi[2 * E:])
        hd = E // H
        o = SdpaFn.apply(q, k, v, B, Q, P, H, hd, hd ** -0.5, key_mask)
        o = self.linear(self.multihead_attn.out_proj(o)).view(B, Q, E)
        return hidden_states + o if add_residual else o


class AriaProjector(nn.Module):
    """projector.py:105-189.  forward(x [N,P,kv_dim], attn_mask [N,P] bool True = masked) -> [N, Q(P), output_dim]."""

    def __init__(self, patch_to_query_dict: Dict[int, int], embed_dim, num_heads, kv_dim, ff_dim, output_dim):
        super().__init__()
        self.patch_to_query_dict = {int(k): int(v) for k, v in patch_to_query_dict.items()}
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.query = nn.Parameter(torch.zeros(max(self.patch_to_query_dict.values()), embed_dim, dtype=bf16))
        self.cross_attn = CrossAttention(kv_dim, embed_dim, num_heads)
        self.ln_ffn = LayerNorm(embed_dim)
        self.ffn = FFN(embed_dim, ff_dim, output_dim)

    def forward(self, x: torch.Tensor, attn_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        bs, P = x.shape[0], x.shape[1]
        query_num = self.patch_to_query_dict.get(P, None)
        assert query_num is not None, f"Query number for {P} patches is not provided"
        queries = self.query[:query_num].unsqueeze(0).expand(bs, query_num, self.embed_dim).contiguous()
        key_mask = None if attn_mask is None else (~attn_mask).to(torch.uint8).contiguous()
        attention_out = self.cross_attn(x, queries, key_mask=key_mask)
        return self.ffn(self.ln_ffn(attention_out))
