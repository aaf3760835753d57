"""ctypes binding of ``libaria_hip.so`` (the C ABI declared in ``include/aria_hip.h``).

The library is the product: if it is missing, importing an op raises -- there is no
PyTorch / CPU fallback anywhere in this package (a silent fallback would void every
parity claim).  Build it with ``make`` or ``python -c "import __graft_entry__ as g; g.build()"``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaria_hip.so")

ERRORS = {
    1: "ARIA_ERR_INVALID (null pointer / negative size)",
    2: "ARIA_ERR_ALIGN (pointer or leading dimension not 16-byte aligned)",
    3: "ARIA_ERR_UNSUPPORTED (shape outside what the kernels implement)",
    4: "ARIA_ERR_LAUNCH (HIP launch failed)",
}

P, I64, I32, F32 = c_void_p, c_int64, c_int, c_float
RESTYPES = {"aria_gemm_workspace_bytes": c_int64, "aria_decode_scratch_bytes": c_int64, "aria_decode_graph_create": c_void_p,
            "aria_decode_graph_destroy": None, "aria_decode_attn_workspace_bytes": c_int64}  # everything else returns an int status

# name -> argtypes (all return int).  Kept in one table so tests can check that the shared
# library exports every symbol the header declares.
SIGNATURES = {
    "aria_abi_version": [],
    "aria_last_gemm_variant": [],
    "aria_last_attn_bwd_variant": [],
    "aria_last_attn_fwd_variant": [],
    "aria_decode_scratch_bytes": [P],
    "aria_decode_trace_layout": [P, P],
    "aria_decode_token": [P, P, F32, P],
    "aria_decode_route": [P, I64, I64, P, P, P],
    "aria_sample_topk": [P, P, I64, I64, F32, P, P],
    "aria_decode_attn_workspace_bytes": [I64, I64, I64],
    "aria_decode_attn": [P, P, P, P, P, P, I64, I64, I64, P, I64, P],
    "aria_decode_graph_create": [P, P, F32],
    "aria_decode_graph_launch": [P, P],
    "aria_decode_graph_destroy": [P],
    "aria_gemm_bf16": [P, P, P, P, I64, I64, I64, I32, I32, I64, I64, I64, I32, I32, P],
    "aria_gemm_bf16_ws": [P, P, P, P, I64, I64, I64, I32, I32, I64, I64, I64, I32, I32, P, I64, P],
    "aria_gemm_workspace_bytes": [I64, I64, I64, I32, I32],
    "aria_gemm_act_bf16": [P, P, P, P, I64, I64, I64, I32, I32, I64, I64, I64, I32, I32, I32, P, I64, P],
    "aria_grouped_gemm_bf16": [P, P, P, P, I64, I64, I64, I64, I32, I64, I64, I64, I64, P],
    "aria_grouped_gemm_wgrad_bf16": [P, P, P, P, I64, I64, I64, I64, I64, I32, I32, P],
    "aria_grouped_gemm_swiglu_bf16": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, P],
    "aria_gemm_swiglu_bf16": [P, P, P, P, I64, I64, I64, I64, I64, I64, I64, P],
    "aria_grouped_gemm_swiglu_split_bf16": [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, P],
    "aria_grouped_gemm_seg_bf16": [P, P, P, P, I64, I64, I64, I64, I64, I32, I64, I64, I64, I64, P],
    "aria_grouped_gemm_swiglu_seg_bf16": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, P],
    "aria_gemm_lora_bf16": [P, P, P, I64, I64, I64, I32, I64, I64, I64, P, P, I64, I64, I64, P],
    "aria_gemm_swiglu_lora_bf16": [P, P, P, P, I64, I64, I64, I64, I64, I64, I64, P, P, I64, I64, I64, P],
    "aria_grouped_gemm_lora_bf16": [P, P, P, P, I64, I64, I64, I64, I32, I64, I64, I64, I64, P, P, I64, I64, I64, I64, P],
    "aria_grouped_gemm_swiglu_lora_bf16": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, P, P, I64, I64, I64, I64, P],
    "aria_dropout_fwd_bf16": [P, P, P, I64, F32, ctypes.c_uint64, P],
    "aria_dropout_bwd_bf16": [P, P, P, I64, F32, I32, P],
    "aria_grouped_gemm_wgrad_gather_bf16": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, P],
    "aria_gemm_qkv_rope_hf_bf16": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, P],
    "aria_gemm_qkv_rope_cache_bf16": [P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, P],
    "aria_grouped_gemm_swiglu_gather_bf16": [P, P, I64, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, P],
    "aria_grouped_gemm_swiglu_split_gather_bf16": [P, P, I64, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, P],
    "aria_gemm_swiglu_split_bf16": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, P],
    "aria_grouped_gemm_dswiglu_bf16": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, P],
    "aria_gemm_dswiglu_bf16": [P, P, P, P, I64, I64, I64, I32, I64, I64, I64, I64, P],
    "aria_moe_route": [P, I32, P, P, P, I64, I64, I64, P],
    "aria_moe_router_fused": [P, P, P, P, P, P, I64, I64, I64, I64, I64, P],
    "aria_moe_sort": [P, P, P, P, P, P, I64, I64, I64, P],
    "aria_moe_permute": [P, P, P, I64, I64, I64, I64, P],
    "aria_moe_unpermute": [P, P, P, P, P, I64, I64, I64, P],
    "aria_moe_unpermute_res": [P, P, P, P, P, P, I64, I64, I64, P],
    "aria_moe_unpermute_bwd": [P, P, P, P, P, P, I64, I64, I64, P],
    "aria_moe_route_bwd": [P, P, P, P, P, P, I64, I64, I64, F32, F32, F32, P],
    "aria_embedding_bwd": [P, P, P, I64, I64, P],
    "aria_swiglu_fwd": [P, P, P, I64, I64, P],
    "aria_swiglu_bwd": [P, P, P, P, P, I64, I64, P],
    "aria_rmsnorm_fwd": [P, P, P, P, P, P, I64, I64, F32, P],
    "aria_rmsnorm_bwd": [P, P, P, P, P, P, P, I64, I64, I64, P],
    "aria_colsum_f32": [P, P, I64, I64, I32, P],
    "aria_rope_inplace": [P, P, P, I64, I64, I64, I64, I64, I32, P],
    "aria_rope_interleaved_inplace": [P, P, P, I64, I64, I64, I64, I64, P],
    "aria_adamw_step": [P, P, P, P, P, I64, F32, F32, F32, F32, F32, I64, F32, P],
    "aria_sumsq_bf16": [P, I64, P, I32, P, P],
    "aria_add_bf16": [P, P, P, I64, P],
    "aria_scale_bf16": [P, P, I64, P],
    "aria_attn_fwd": [P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, F32, I32, P],
    "aria_attn_bwd": [P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, F32, I32, P],
    "aria_attn_bwd_rope": [P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, F32, I32, P, P, I64, P],
    "aria_layernorm_fwd": [P, P, P, P, P, P, I64, I64, F32, P],
    "aria_layernorm_bwd": [P, P, P, P, P, P, P, P, I64, I64, I64, P],
    "aria_gelu_tanh_fwd": [P, P, I64, P],
    "aria_gelu_tanh_bwd": [P, P, P, I64, P],
    "aria_vit_patch_mask": [P, P, I64, I64, I64, P],
    "aria_vit_pos_ids": [P, P, P, I64, I64, I64, I64, P],
    "aria_vit_im2col": [P, I32, P, I64, I64, I64, I64, I64, P],
    "aria_gather_add_rows": [P, P, P, I64, I64, P],
    "aria_colsum_bf16": [P, P, I64, I64, I64, I64, P],
    "aria_cross_entropy": [P, P, P, P, P, F32, P, I64, I64, I64, P],
}


ABI_VERSION = 3   # include/aria_hip.h ARIA_ABI_VERSION this host code was written against


class AriaHipError(RuntimeError):
    pass


class HipLibrary:
    """A loaded C-ABI library; ``call(name, *args)`` raises on a non-zero status."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise AriaHipError(
                f"{path} not found: the HIP library has not been built (run `make` in the repo root). "
                "aria_amd has no fallback path."
            )
        self.path = path
        self.cdll = ctypes.CDLL(path)
        self.missing = []
        for name, argtypes in SIGNATURES.items():
            try:
                fn = getattr(self.cdll, name)
            except AttributeError:
                self.missing.append(name)
                continue
            fn.argtypes = argtypes
            fn.restype = RESTYPES.get(name, c_int)
        if "aria_abi_version" not in self.missing and self.cdll.aria_abi_version() != ABI_VERSION:
            raise AriaHipError(f"{path} has ABI version {self.cdll.aria_abi_version()}, aria_amd expects {ABI_VERSION}: rebuild it (`make`)")

    def call(self, name: str, *args) -> None:
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            raise AriaHipError(f"{name} failed: {ERRORS.get(rc, rc)}")


_LIB: Optional[HipLibrary] = None
_EMULATED = False  # set only by tests/emu (never by product code)


def get_lib() -> HipLibrary:
    global _LIB
    if _LIB is None:
        _LIB = HipLibrary(LIB_PATH)
        if _LIB.missing:
            raise AriaHipError(f"{LIB_PATH} lacks symbols {_LIB.missing}: rebuild it")
    return _LIB


def is_emulated() -> bool:
    return _EMULATED
