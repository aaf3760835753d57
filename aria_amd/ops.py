"""Functional (non-autograd) wrappers over the C ABI: torch tensors in, raw pointers out.

PyTorch is plumbing here (device memory + the current HIP stream); all arithmetic happens in
``libaria_hip.so``.  Every function requires CUDA(=HIP) tensors; there is no CPU path.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import hip

bf16 = torch.bfloat16


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(t: torch.Tensor):
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    if not hip.is_emulated():
        raise hip.AriaHipError("aria_amd ops need tensors on an MI355X (cuda) device; there is no CPU path")
    return None


def _chk(t: torch.Tensor, dtype=bf16, name="tensor"):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _rowmajor_2d(t: torch.Tensor, name="tensor") -> int:
    """Returns the leading dimension (elements) of a 2-D view whose last dim is contiguous."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: need a 2-D view with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# ----------------------------------------------------------------------------- GEMM
ACTIVATIONS = {None: 0, "gelu_tanh": 1}


def gemm(a: torch.Tensor, b: torch.Tensor, *, a_oc: bool = False, b_oc: bool = False, bias: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None, out_dtype=bf16, accumulate: bool = False, act: Optional[str] = None) -> torch.Tensor:
    """C[M,N] = op(A) op(B) (+bias) (+C).  a: [M,K] (a_oc=False) or [K,M] (a_oc=True);
    b: [N,K] (b_oc=False, i.e. nn.Linear weight) or [K,N] (b_oc=True)."""
    _chk(a, name="a"), _chk(b, name="b")
    lda, ldb = _rowmajor_2d(a, "a"), _rowmajor_2d(b, "b")
    M, K = (a.shape[1], a.shape[0]) if a_oc else (a.shape[0], a.shape[1])
    N, Kb = (b.shape[1], b.shape[0]) if b_oc else (b.shape[0], b.shape[1])
    if K != Kb:
        raise ValueError(f"gemm: reduction sizes differ ({K} vs {Kb})")
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an existing `out`")
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    ldc = _rowmajor_2d(out, "out")
    if out.shape != (M, N):
        raise ValueError(f"gemm: out shape {tuple(out.shape)} != {(M, N)}")
    c_f32 = {bf16: 0, torch.float32: 1}[out.dtype]
    if bias is not None:
        _chk(bias, name="bias")
    lib = hip.get_lib()
    need = lib.cdll.aria_gemm_workspace_bytes(M, N, K, int(a_oc), int(b_oc)) if GEMM_SPLIT_K else 0
    if act is not None:  # epilogue activation on bf16(acc + bias), bit-identical to the stand-alone elementwise kernel
        ws = _gemm_workspace(a.device, _stream(a), need) if need > 0 else None
        lib.call("aria_gemm_act_bf16", _p(a), _p(b), _p(out), _p(bias), M, N, K, int(a_oc), int(b_oc), lda, ldb, ldc, c_f32,
                 int(accumulate), ACTIVATIONS[act], _p(ws), ws.numel() if ws is not None else 0, _stream(a))
    elif need > 0:  # small outputs: let the library split the last round of tiles along K (fp32 slabs in a scratch buffer)
        ws = _gemm_workspace(a.device, _stream(a), need)
        lib.call("aria_gemm_bf16_ws", _p(a), _p(b), _p(out), _p(bias), M, N, K, int(a_oc), int(b_oc), lda, ldb, ldc,
                 c_f32, int(accumulate), _p(ws), ws.numel(), _stream(a))
    else:
        lib.call("aria_gemm_bf16", _p(a), _p(b), _p(out), _p(bias), M, N, K, int(a_oc), int(b_oc), lda, ldb, ldc,
                 c_f32, int(accumulate), _stream(a))
    return out


_WORKSPACES = {}
GEMM_SPLIT_K = True  # tools flip this off to compare kernels bit for bit (split-K changes the fp32 summation order)


def _gemm_workspace(device, stream, nbytes: int) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream): launches on one stream are ordered, so they can share it."""
    key = (str(device), int(stream or 0))
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
        _WORKSPACES[key] = ws
    return ws


def grouped_gemm(a: torch.Tensor, w: torch.Tensor, offsets: torch.Tensor, *, w_is_kn: bool = True,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """experts_gemm with device-side offsets (int32 [E+1]).  w: [E,K,N] (w_is_kn, forward) or [E,N,K]
    used as W_e^T (dgrad through [E,N,K]-shaped storage, i.e. the forward weight [E, K', N'] with K'=N, N'=K)."""
    _chk(a, name="a"), _chk(w, name="w"), _chk(offsets, torch.int32, "offsets")
    if w.dim() != 3 or not w.is_contiguous():
        raise ValueError("grouped_gemm: w must be a contiguous [E, ., .] tensor")
    lda = _rowmajor_2d(a, "a")
    M, K = a.shape
    E = w.shape[0]
    if w_is_kn:
        if w.shape[1] != K:
            raise ValueError("grouped_gemm: w.shape[1] != K")
        N = w.shape[2]
    else:
        if w.shape[2] != K:
            raise ValueError("grouped_gemm: w.shape[2] != K")
        N = w.shape[1]
    if out is None:
        out = torch.empty((M, N), dtype=bf16, device=a.device)
    hip.get_lib().call("aria_grouped_gemm_bf16", _p(a), _p(w), _p(out), _p(offsets), E, M, N, K, int(w_is_kn), lda,
                       w.shape[2], w.shape[1] * w.shape[2], _rowmajor_2d(out, "out"), _stream(a))
    return out


def swiglu_fusion_enabled() -> bool:
    import os

    return os.environ.get("ARIA_FUSE_SWIGLU", "1") != "0"


def glu_fusable(K: int, N2: int) -> bool:
    """Shapes the fused fc1 + SwiGLU launch takes (a tile = 128 gate + 128 up columns; ARIA_FUSE_SWIGLU=0 switches the fusion off)."""
    return swiglu_fusion_enabled() and N2 % 2 == 0 and (N2 // 2) % 128 == 0 and K >= 64 and K % 8 == 0


def grouped_gemm_swiglu(a: torch.Tensor, w: torch.Tensor, offsets: torch.Tensor, want_h: bool = True):
    """fc1 + glu of GroupedMLP.forward (moe_lm.py:505-507, 522-523) in one launch: -> (h [M, 2I] or None, act [M, I]).
    Bit-identical to ``swiglu(grouped_gemm(a, w, offsets))``."""
    _chk(a, name="a"), _chk(w, name="w"), _chk(offsets, torch.int32, "offsets")
    if w.dim() != 3 or not w.is_contiguous() or w.shape[1] != a.shape[1]:
        raise ValueError("grouped_gemm_swiglu: w must be a contiguous [E, K, 2I] tensor")
    lda = _rowmajor_2d(a, "a")
    M, K = a.shape
    E, _, N2 = w.shape
    h = torch.empty((M, N2), dtype=bf16, device=a.device) if want_h else None
    act = torch.empty((M, N2 // 2), dtype=bf16, device=a.device)
    hip.get_lib().call("aria_grouped_gemm_swiglu_bf16", _p(a), _p(w), _p(h) if want_h else None, _p(act), _p(offsets), E, M, N2, K, lda, N2,
                       K * N2, N2, N2 // 2, _stream(a))
    return h, act


def gemm_swiglu(x: torch.Tensor, w: torch.Tensor, want_h: bool = True):
    """act(gate_proj(x)) * up_proj(x) on the row-wise concatenation w = [gate_proj.weight; up_proj.weight] ([2I, K]) in one launch:
    -> (h [M, 2I] or None, act [M, I]); bit-identical to ``swiglu(gemm(x, w))``."""
    _chk(x, name="x"), _chk(w, name="w")
    lda, ldb = _rowmajor_2d(x, "x"), _rowmajor_2d(w, "w")
    M, K = x.shape
    N2 = w.shape[0]
    h = torch.empty((M, N2), dtype=bf16, device=x.device) if want_h else None
    act = torch.empty((M, N2 // 2), dtype=bf16, device=x.device)
    hip.get_lib().call("aria_gemm_swiglu_bf16", _p(x), _p(w), _p(h) if want_h else None, _p(act), M, N2, K, lda, ldb, N2, N2 // 2, _stream(x))
    return h, act


def dropout(x: torch.Tensor, p: float, seed: int):
    """nn.Dropout(p) of the LoRA layers (aria/lora/layers.py:83-85) in training mode: -> (x_dropped, mask uint8 [numel / 8]); the mask is a
    counter-based function of (seed, element index)."""
    _chk(x, name="x")
    if not x.is_contiguous() or x.numel() % 8:
        raise ValueError("dropout: contiguous tensor with numel % 8 == 0")
    out = torch.empty_like(x)
    mask = torch.empty((x.numel() // 8,), dtype=torch.uint8, device=x.device)
    hip.get_lib().call("aria_dropout_fwd_bf16", _p(x), _p(out), _p(mask), x.numel(), float(p), int(seed) & ((1 << 64) - 1), _stream(x))
    return out, mask


def dropout_bwd_(dx: torch.Tensor, term: torch.Tensor, mask: torch.Tensor, p: float, accumulate: bool = True) -> torch.Tensor:
    """dx (+)= mask * term / (1 - p), in place."""
    _chk(dx, name="dx"), _chk(term, name="term"), _chk(mask, torch.uint8, "mask")
    if not (dx.is_contiguous() and term.is_contiguous()) or dx.shape != term.shape or mask.numel() * 8 != dx.numel():
        raise ValueError("dropout_bwd_: contiguous dx / term of one shape, one mask byte per 8 elements")
    hip.get_lib().call("aria_dropout_bwd_bf16", _p(term), _p(mask), _p(dx), dx.numel(), float(p), int(accumulate), _stream(dx))
    return dx


# ------------------------------------------------------------------------------------------------ LoRA as a K-extension
def _lora_ext_ok(K: int, ext_k: int) -> bool:
    import os

    return K % 64 == 0 and K >= 64 and 0 < ext_k <= 64 and ext_k % 8 == 0 and os.environ.get("ARIA_FUSE_LORA", "1") != "0"


def _try_lora(name: str, *args) -> bool:
    """Run a fused base + adapter entry; False when the library says the shape is not one the 256 x 256 kernels take (the caller then runs
    base and adapter as two launches, the second accumulating) -- every other status raises."""
    rc = getattr(hip.get_lib().cdll, name)(*args)
    if rc == 3:   # ARIA_ERR_UNSUPPORTED
        return False
    if rc != 0:
        raise hip.AriaHipError(f"{name} failed: {hip.ERRORS.get(rc, rc)}")
    return True


def gemm_lora(a: torch.Tensor, b: torch.Tensor, ea: torch.Tensor, eb: torch.Tensor, *, b_oc: bool = False) -> torch.Tensor:
    """C = a b + ea eb in ONE launch (aria/lora/layers.py:129-139 for nn.Linear targets: base(x) + lora_B(scaling * lora_A(x))): a [M, K]; b
    [N, K] (b_oc False: the Linear's weight; eb [N, r] = lora_B.weight) or [K, N] (b_oc True: the dgrad form; eb [r, N] = lora_A.weight); ea
    [M, r] (scaling folded in by the caller).  Falls back to two launches (the second accumulating) on shapes the fused kernel does not take."""
    _chk(a, name="a"), _chk(b, name="b"), _chk(ea, name="ea"), _chk(eb, name="eb")
    lda, ldb = _rowmajor_2d(a, "a"), _rowmajor_2d(b, "b")
    M, K = a.shape
    N = b.shape[1] if b_oc else b.shape[0]
    r = ea.shape[1]
    if ea.shape[0] != M or eb.shape != ((r, N) if b_oc else (N, r)):
        raise ValueError(f"gemm_lora: ea {tuple(ea.shape)} / eb {tuple(eb.shape)} do not extend a [{M}, {K}] x [{N}] product")
    out = torch.empty((M, N), dtype=bf16, device=a.device)
    if _lora_ext_ok(K, r) and _try_lora("aria_gemm_lora_bf16", _p(a), _p(b), _p(out), M, N, K, int(b_oc), lda, ldb, N, _p(ea), _p(eb), r,
                                        _rowmajor_2d(ea, "ea"), _rowmajor_2d(eb, "eb"), _stream(a)):
        return out
    gemm(a, b, b_oc=b_oc, out=out)
    return gemm(ea, eb, b_oc=b_oc, out=out, accumulate=True)


def gemm_swiglu_lora(x: torch.Tensor, w: torch.Tensor, ea: torch.Tensor, eb: torch.Tensor, want_h: bool = True):
    """``gemm_swiglu`` on base + adapter: h = x w^T + ea eb^T ([2I, r] eb), act = glu(h); one launch where the fused kernel takes the shape."""
    _chk(x, name="x"), _chk(w, name="w"), _chk(ea, name="ea"), _chk(eb, name="eb")
    M, K = x.shape
    N2, r = w.shape[0], ea.shape[1]
    if glu_fusable(K, N2) and _lora_ext_ok(K, r):
        h = torch.empty((M, N2), dtype=bf16, device=x.device) if want_h else None
        act = torch.empty((M, N2 // 2), dtype=bf16, device=x.device)
        if _try_lora("aria_gemm_swiglu_lora_bf16", _p(x), _p(w), _p(h) if want_h else None, _p(act), M, N2, K, _rowmajor_2d(x, "x"),
                     _rowmajor_2d(w, "w"), N2, N2 // 2, _p(ea), _p(eb), r, _rowmajor_2d(ea, "ea"), _rowmajor_2d(eb, "eb"), _stream(x)):
            return h, act
    h = gemm(x, w)
    gemm(ea, eb, out=h, accumulate=True)
    return (h if want_h else None), swiglu(h)


def grouped_gemm_lora(a: torch.Tensor, w: torch.Tensor, offsets: torch.Tensor, ea: torch.Tensor, eb: torch.Tensor, *,
                      w_is_kn: bool = True) -> torch.Tensor:
    """experts_gemm on base + adapter (GroupedGemmLoraLayer.forward, aria/lora/layers.py:129-139): rows of expert e times (w[e] + the rank-r
    product): w [E, K, N] with eb [E, r, N] (w_is_kn: forward, eb = lora_B.weight) or w [E, N, K] used transposed with eb [E, N, r]
    (dgrad form, eb = lora_A.weight [E, K_in, r] read as [E, N, r]); ea [M, r]."""
    _chk(a, name="a"), _chk(w, name="w"), _chk(offsets, torch.int32, "offsets"), _chk(ea, name="ea"), _chk(eb, name="eb")
    if w.dim() != 3 or not w.is_contiguous() or eb.dim() != 3 or not eb.is_contiguous() or eb.shape[0] != w.shape[0]:
        raise ValueError("grouped_gemm_lora: w and eb must be contiguous [E, ., .] tensors")
    M, K = a.shape
    E = w.shape[0]
    N = w.shape[2] if w_is_kn else w.shape[1]
    r = ea.shape[1]
    if eb.shape[1:] != ((r, N) if w_is_kn else (N, r)) or ea.shape[0] != M:
        raise ValueError(f"grouped_gemm_lora: eb {tuple(eb.shape)} does not extend [{E}, {K}, {N}] by rank {r}")
    out = torch.empty((M, N), dtype=bf16, device=a.device)
    if _lora_ext_ok(K, r) and _try_lora("aria_grouped_gemm_lora_bf16", _p(a), _p(w), _p(out), _p(offsets), E, M, N, K, int(w_is_kn),
                                        _rowmajor_2d(a, "a"), w.shape[2], w.shape[1] * w.shape[2], N, _p(ea), _p(eb), r,
                                        _rowmajor_2d(ea, "ea"), eb.shape[2], eb.shape[1] * eb.shape[2], _stream(a)):
        return out
    grouped_gemm(a, w, offsets, w_is_kn=w_is_kn, out=out)
    return out.add_(grouped_gemm(ea, eb, offsets, w_is_kn=w_is_kn))


def grouped_gemm_swiglu_lora(a: torch.Tensor, w: torch.Tensor, offsets: torch.Tensor, ea: torch.Tensor, eb: torch.Tensor, want_h: bool = True):
    """``grouped_gemm_swiglu`` on base + adapter: fc1 + lora + glu in one launch (w [E, K, 2I], eb [E, r, 2I])."""
    _chk(a, name="a"), _chk(w, name="w"), _chk(offsets, torch.int32, "offsets"), _chk(ea, name="ea"), _chk(eb, name="eb")
    M, K = a.shape
    E, _, N2 = w.shape
    r = ea.shape[1]
    if eb.shape != (E, r, N2) or not eb.is_contiguous():
        raise ValueError("grouped_gemm_swiglu_lora: eb must be a contiguous [E, r, 2I] tensor")
    if glu_fusable(K, N2) and _lora_ext_ok(K, r):
        h = torch.empty((M, N2), dtype=bf16, device=a.device) if want_h else None
        act = torch.empty((M, N2 // 2), dtype=bf16, device=a.device)
        if _try_lora("aria_grouped_gemm_swiglu_lora_bf16", _p(a), _p(w), _p(h) if want_h else None, _p(act), _p(offsets), E, M, N2, K,
                     _rowmajor_2d(a, "a"), N2, K * N2, N2, N2 // 2, _p(ea), _p(eb), r, _rowmajor_2d(ea, "ea"), N2, r * N2, _stream(a)):
            return h, act
    h = grouped_gemm_lora(a, w, offsets, ea, eb)
    return (h if want_h else None), swiglu(h)


def glu_split_fusable(w_gate: torch.Tensor, w_up: torch.Tensor) -> bool:
    """gate / up weights as two [.., I, K] tensors ([N, K] form, the gptfast wire format): the fused launch reaches the up rows as a ROW
    offset from the gate rows, so both must be contiguous views of one allocation with the up tensor a whole number of rows behind the gate
    tensor (``gptfast.adjacent_pair`` builds that layout); I % 128 == 0, K % 64 == 0."""
    if not swiglu_fusion_enabled() or w_gate.shape != w_up.shape or not (w_gate.is_contiguous() and w_up.is_contiguous()):
        return False
    I, K = w_gate.shape[-2], w_gate.shape[-1]
    same_storage = w_gate.untyped_storage().data_ptr() == w_up.untyped_storage().data_ptr()
    return bool(same_storage and glu_split_offset_ok(w_up.data_ptr() - w_gate.data_ptr(), I, K, w_gate.numel()))


def glu_split_offset_ok(diff_bytes: int, I: int, K: int, numel: int) -> bool:
    """The limits of csrc/gemm.hip ``glu_split_rows`` on the byte distance between the gate and the up tensor, mirrored exactly (per-lane DMA
    offsets are 32-bit, the row index goes through a 24-bit multiply) + a margin of one 128-row tile on the byte limit."""
    if I % 128 or K % 64 or K < 64 or diff_bytes < 2 * numel or diff_bytes % (2 * K):
        return False
    rows = diff_bytes // (2 * K)
    return rows >= I and rows + I < (1 << 24) and (rows + I + 128) * 2 * K < (1 << 32)


def grouped_gemm_swiglu_split(a: torch.Tensor, w_gate: torch.Tensor, w_up: torch.Tensor, offsets: torch.Tensor, want_h: bool = False):
    """silu(a @ w_gate[e]^T) * (a @ w_up[e]^T) per expert in one launch (gptfast ConditionalFeedForward, model.py:278-325): w_gate / w_up
    [E, I, K]; -> (h [M, 2I] or None, act [M, I]); bit-identical to ``swiglu(grouped_gemm(a, w_gate, w_is_kn=False), grouped_gemm(a, w_up, ..))``."""
    _chk(a, name="a"), _chk(w_gate, name="w_gate"), _chk(w_up, name="w_up"), _chk(offsets, torch.int32, "offsets")
    E, I, K = w_gate.shape
    M = a.shape[0]
    h = torch.empty((M, 2 * I), dtype=bf16, device=a.device) if want_h else None
    act = torch.empty((M, I), dtype=bf16, device=a.device)
    hip.get_lib().call("aria_grouped_gemm_swiglu_split_bf16", _p(a), _p(w_gate), _p(w_up), _p(h) if want_h else None, _p(act), _p(offsets), E, M,
                       I, K, _rowmajor_2d(a, "a"), K, I * K, 2 * I, I, _stream(a))
    return h, act


def segments_supported(K: int) -> bool:
    """Grouped launches over the segments of an expert-parallel exchange (``expert_mod``): v3 kernels only; ARIA_EP_SEGMENTS=0 switches the
    expert-parallel layer back to re-ordering the received rows."""
    import os

    return K % 64 == 0 and K >= 64 and os.environ.get("ARIA_EP_SEGMENTS", "1") != "0"


def grouped_gemm_seg(a: torch.Tensor, w: torch.Tensor, offsets: torch.Tensor, *, w_is_kn: bool = True) -> torch.Tensor:
    """``grouped_gemm`` over segments: offsets int32 [n_seg + 1], n_seg a multiple of w.shape[0]; segment g uses w[g % w.shape[0]]."""
    _chk(a, name="a"), _chk(w, name="w"), _chk(offsets, torch.int32, "offsets")
    if w.dim() != 3 or not w.is_contiguous():
        raise ValueError("grouped_gemm_seg: w must be a contiguous [E_local, ., .] tensor")
    M, K = a.shape
    El, n_seg = w.shape[0], offsets.numel() - 1
    N = w.shape[2] if w_is_kn else w.shape[1]
    if (w.shape[1] if w_is_kn else w.shape[2]) != K or n_seg % El:
        raise ValueError("grouped_gemm_seg: reduction sizes differ or the segment count is not a multiple of the local experts")
    out = torch.empty((M, N), dtype=bf16, device=a.device)
    hip.get_lib().call("aria_grouped_gemm_seg_bf16", _p(a), _p(w), _p(out), _p(offsets), n_seg, El, M, N, K, int(w_is_kn), _rowmajor_2d(a, "a"),
                       w.shape[2], w.shape[1] * w.shape[2], N, _stream(a))
    return out


def grouped_gemm_swiglu_seg(a: torch.Tensor, w: torch.Tensor, offsets: torch.Tensor, want_h: bool = True):
    """``grouped_gemm_swiglu`` over segments (see ``grouped_gemm_seg``)."""
    _chk(a, name="a"), _chk(w, name="w"), _chk(offsets, torch.int32, "offsets")
    M, K = a.shape
    El, _, N2 = w.shape
    n_seg = offsets.numel() - 1
    if w.shape[1] != K or n_seg % El or not w.is_contiguous():
        raise ValueError("grouped_gemm_swiglu_seg: w must be a contiguous [E_local, K, 2I] tensor and the segments a multiple of E_local")
    h = torch.empty((M, N2), dtype=bf16, device=a.device) if want_h else None
    act = torch.empty((M, N2 // 2), dtype=bf16, device=a.device)
    hip.get_lib().call("aria_grouped_gemm_swiglu_seg_bf16", _p(a), _p(w), _p(h) if want_h else None, _p(act), _p(offsets), n_seg, El, M, N2, K,
                       _rowmajor_2d(a, "a"), N2, K * N2, N2, N2 // 2, _stream(a))
    return h, act


def grouped_gemm_wgrad_seg(a: torch.Tensor, dy: torch.Tensor, offsets: torch.Tensor, El: int) -> torch.Tensor:
    """dW[e] = sum over source ranks s of a[seg(s, e)]^T dy[seg(s, e)]: one grouped-K launch per source rank on that rank's El + 1 offsets;
    several ranks accumulate in fp32 and round once."""
    n_seg = offsets.numel() - 1
    W = n_seg // El
    if W == 1:
        return grouped_gemm_wgrad(a, dy, offsets, El)
    out = torch.empty((El, a.shape[1], dy.shape[1]), dtype=torch.float32, device=a.device)
    for src in range(W):
        grouped_gemm_wgrad(a, dy, offsets[src * El: (src + 1) * El + 1], El, out=out, accumulate=src > 0)
    return out.to(bf16)


def qkv_rope_cache_fusable(D: int, K: int, hd: int) -> bool:
    """The wqkv projection can carry RoPE and the KV-cache write as its epilogue (K7); ARIA_FUSE_QKV_ROPE=0: the three-step chain."""
    import os

    return D % 256 == 0 and K % 64 == 0 and K >= 64 and hd % 8 == 0 and D % hd == 0 and os.environ.get("ARIA_FUSE_QKV_ROPE", "1") != "0"


def gemm_qkv_rope_cache(x: torch.Tensor, wqkv: torch.Tensor, freqs_cis: torch.Tensor, pos32: Optional[torch.Tensor], k_cache: torch.Tensor,
                        v_cache: torch.Tensor, S: int, hd: int) -> torch.Tensor:
    """gptfast Attention.forward up to the attention call (model.py:413-435) in one launch: x [B*S, K], wqkv [3D, K], freqs_cis bf16
    [positions, hd/2, 2], pos32 int32 [B*S] (or None: t % S), k_cache / v_cache [B_max, S_cache, D] (rows (t // S, pos) are written).
    -> q [B*S, D] rotated.  Bit-identical to ``gemm`` + ``rope_interleaved_`` + the cache copies."""
    _chk(x, name="x"), _chk(wqkv, name="wqkv"), _chk(freqs_cis, name="freqs_cis"), _chk(k_cache, name="k_cache"), _chk(v_cache, name="v_cache")
    M, K = x.shape
    D = wqkv.shape[0] // 3
    if wqkv.shape != (3 * D, K) or k_cache.dim() != 3 or k_cache.shape != v_cache.shape or k_cache.shape[2] != D:
        raise ValueError("gemm_qkv_rope_cache: wqkv [3D, K], caches [B, S_cache, D]")
    if not (k_cache.is_contiguous() and v_cache.is_contiguous() and freqs_cis.is_contiguous()):
        raise ValueError("gemm_qkv_rope_cache: caches and freqs_cis must be contiguous")
    if M % S or M // S > k_cache.shape[0]:
        raise ValueError("gemm_qkv_rope_cache: rows must be whole sequences that fit the cache's batch")
    if S > k_cache.shape[1]:
        raise ValueError(f"gemm_qkv_rope_cache: {S} positions do not fit a cache of {k_cache.shape[1]} slots (setup_caches first)")
    if pos32 is not None:
        _chk(pos32, torch.int32, "pos32")
        assert pos32.is_contiguous() and pos32.numel() == M
    q = torch.empty((M, D), dtype=bf16, device=x.device)
    hip.get_lib().call("aria_gemm_qkv_rope_cache_bf16", _p(x), _p(wqkv), _p(q), _p(k_cache), _p(v_cache), _p(freqs_cis), _p(pos32), M, D, K, hd, S,
                       k_cache.shape[1], _rowmajor_2d(x, "x"), _rowmajor_2d(wqkv, "wqkv"), D, D, _stream(x))
    return q


def gather_fusable(K: int) -> bool:
    """The fused fc1 + SwiGLU launches can take the UN-permuted tokens and the dispatcher's row index (K2); ARIA_FUSE_GATHER=0 switches it
    off (the permuted copy is built and the un-gathered launch runs: bit-identical)."""
    import os

    return K % 64 == 0 and K >= 64 and os.environ.get("ARIA_FUSE_GATHER", "1") != "0"


def permuted_token_rows(sorted_src: torch.Tensor, k: int, pad: int = 64) -> torch.Tensor:
    """Token row of every permuted row (``sorted_indices // topk``, moe_lm.py:330): int32 [T * k] -- a view of a buffer with ``pad`` more entries
    (the weight gradient's gathered loader reads a whole K-tile of indices at the end of the last expert)."""
    n = sorted_src.numel()
    buf = torch.zeros((n + pad,), dtype=torch.int32, device=sorted_src.device)
    torch.div(sorted_src, k, rounding_mode="floor", out=buf[:n])
    return buf[:n]


def wgrad_gather_fusable(K: int) -> bool:
    """The fc1 weight gradient can take the UN-permuted tokens + the dispatcher's index (no permuted copy in the training step)."""
    import os

    return gather_fusable(K) and os.environ.get("ARIA_FUSE_WGRAD_GATHER", "1") != "0"


def grouped_gemm_wgrad_gather(x: torch.Tensor, rows: torch.Tensor, dy: torch.Tensor, offsets: torch.Tensor, E: int, out_dtype=bf16):
    """``grouped_gemm_wgrad(moe_permute(x, ..), dy, offsets, E)`` without the permuted copy: x [T, K], rows = ``permuted_token_rows`` (its
    padded buffer), dy [M, N] -> dW [E, K, N]; None when the library does not take the shape (the caller permutes and uses the plain entry)."""
    _chk(x, name="x"), _chk(dy, name="dy"), _chk(offsets, torch.int32, "offsets"), _chk(rows, torch.int32, "rows")
    if rows.untyped_storage().nbytes() - rows.storage_offset() * 4 < (rows.numel() + 64) * 4 or not rows.is_contiguous():
        raise ValueError("grouped_gemm_wgrad_gather: rows must come from permuted_token_rows (64 entries of padding behind it)")
    K, N = x.shape[1], dy.shape[1]
    out = torch.empty((E, K, N), dtype=out_dtype, device=x.device)
    if dy.shape[0] == 0:
        return out.zero_()
    rc = hip.get_lib().cdll.aria_grouped_gemm_wgrad_gather_bf16(_p(x), _p(rows), _p(dy), _p(out), _p(offsets), E, x.shape[0], K, N, _rowmajor_2d(x, "x"),
                                                                _rowmajor_2d(dy, "dy"), int(out_dtype == torch.float32), 0, _stream(x))
    if rc == 3:
        return None
    if rc != 0:
        raise hip.AriaHipError(f"aria_grouped_gemm_wgrad_gather_bf16 failed: {hip.ERRORS.get(rc, rc)}")
    return out


def grouped_gemm_swiglu_gather(x: torch.Tensor, rows: torch.Tensor, w: torch.Tensor, offsets: torch.Tensor, want_h: bool = False):
    """``grouped_gemm_swiglu(moe_permute(x, ..), w, offsets)`` without the permuted copy: x [T, K] tokens, rows int32 [M] (token row per
    permuted row), w [E, K, 2I].  -> (h [M, 2I] or None, act [M, I]); bit-identical to the two-step form."""
    _chk(x, name="x"), _chk(w, name="w"), _chk(offsets, torch.int32, "offsets"), _chk(rows, torch.int32, "rows")
    if w.dim() != 3 or not w.is_contiguous() or w.shape[1] != x.shape[1] or not rows.is_contiguous():
        raise ValueError("grouped_gemm_swiglu_gather: w must be a contiguous [E, K, 2I] tensor, rows contiguous int32")
    T, K = x.shape
    M = rows.numel()
    E, _, N2 = w.shape
    h = torch.empty((M, N2), dtype=bf16, device=x.device) if want_h else None
    act = torch.empty((M, N2 // 2), dtype=bf16, device=x.device)
    hip.get_lib().call("aria_grouped_gemm_swiglu_gather_bf16", _p(x), _p(rows), T, _p(w), _p(h) if want_h else None, _p(act), _p(offsets), E, M,
                       N2, K, _rowmajor_2d(x, "x"), N2, K * N2, N2, N2 // 2, _stream(x))
    return h, act


def grouped_gemm_swiglu_split_gather(x: torch.Tensor, rows: torch.Tensor, w_gate: torch.Tensor, w_up: torch.Tensor, offsets: torch.Tensor,
                                     want_h: bool = False):
    """The gptfast form (w_gate / w_up [E, I, K], ``glu_split_fusable``) of ``grouped_gemm_swiglu_gather``."""
    _chk(x, name="x"), _chk(w_gate, name="w_gate"), _chk(w_up, name="w_up"), _chk(offsets, torch.int32, "offsets"), _chk(rows, torch.int32, "rows")
    E, I, K = w_gate.shape
    T, M = x.shape[0], rows.numel()
    h = torch.empty((M, 2 * I), dtype=bf16, device=x.device) if want_h else None
    act = torch.empty((M, I), dtype=bf16, device=x.device)
    hip.get_lib().call("aria_grouped_gemm_swiglu_split_gather_bf16", _p(x), _p(rows), T, _p(w_gate), _p(w_up), _p(h) if want_h else None, _p(act),
                       _p(offsets), E, M, I, K, _rowmajor_2d(x, "x"), K, I * K, 2 * I, I, _stream(x))
    return h, act


def gemm_swiglu_split(x: torch.Tensor, w_gate: torch.Tensor, w_up: torch.Tensor, want_h: bool = False):
    """silu(x @ w_gate^T) * (x @ w_up^T) in one launch (gptfast FeedForward): w_gate / w_up [I, K] as in ``grouped_gemm_swiglu_split``."""
    _chk(x, name="x"), _chk(w_gate, name="w_gate"), _chk(w_up, name="w_up")
    I, K = w_gate.shape
    M = x.shape[0]
    h = torch.empty((M, 2 * I), dtype=bf16, device=x.device) if want_h else None
    act = torch.empty((M, I), dtype=bf16, device=x.device)
    hip.get_lib().call("aria_gemm_swiglu_split_bf16", _p(x), _p(w_gate), _p(w_up), _p(h) if want_h else None, _p(act), M, I, K,
                       _rowmajor_2d(x, "x"), K, 2 * I, I, _stream(x))
    return h, act


def sample_topk(logits: torch.Tensor, q: torch.Tensor, temperature: float, top_k: Optional[int], out: Optional[torch.Tensor] = None):
    """gptfast's ``sample`` (generate.py:35-58) for one token in one launch: logits [V] bf16, q [V] fp32 = Exp(1) draws (the caller's
    generator) -> int32 [1] = argmax softmax(top-k filtered logits / T) / q; ties at the k-th value are kept, as the tensor path keeps them."""
    _chk(logits, name="logits"), _chk(q, torch.float32, "q")
    assert logits.is_contiguous() and q.is_contiguous() and logits.numel() == q.numel()
    if out is None:
        out = torch.empty(1, dtype=torch.int32, device=logits.device)
    hip.get_lib().call("aria_sample_topk", _p(logits), _p(q), logits.numel(), int(top_k or 0), float(temperature), _p(out), _stream(logits))
    return out


def decode_route(logits: torch.Tensor, k: int):
    """The decode engine's routing of ONE token (csrc/decode.hip route_one_token) on its own: logits [E] bf16 -> (scores [k] bf16,
    idx [k] int32); the same function as ``moe_route`` on one row."""
    _chk(logits, name="logits")
    E = logits.numel()
    scores = torch.empty(k, dtype=bf16, device=logits.device)
    idx = torch.empty(k, dtype=torch.int32, device=logits.device)
    hip.get_lib().call("aria_decode_route", _p(logits), E, k, _p(scores), _p(idx), _stream(logits))
    return scores, idx


def dglu_fusable(I: int, K: int) -> bool:
    """Shapes the fused input-gradient + SwiGLU-backward launches take (128-column blocks of I are all-or-nothing, the v3 K loop wants
    K % 64 == 0; ARIA_FUSE_DSWIGLU=0 switches the fusion off: the two-step chain -- bit-identical for the grouped launches and for dense
    ones whose two-step GEMM does not select split-K; where it does (a partly filled last round with a workspace) the two agree to the GEMM
    tolerance only, last assertion of ``tests/kernel_cases.py::case_gemm_dswiglu_fused``)."""
    import os

    return os.environ.get("ARIA_FUSE_DSWIGLU", "1") != "0" and I % 128 == 0 and K >= 64 and K % 64 == 0


def grouped_gemm_dswiglu(dy: torch.Tensor, w: torch.Tensor, offsets: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    """Backward of glu behind experts.fc2's input gradient in one launch (moe_lm.py:505-507, 524): dy [M, K] (gradient of the fc2 output
    rows), w = fc2.weight [E, I, K], h [M, 2I] = the forward's [gate | up] -> d_h [M, 2I].
    Bit-identical to ``swiglu_bwd(h, grouped_gemm(dy, w, offsets, w_is_kn=False))``; the [M, I] product never visits HBM."""
    _chk(dy, name="dy"), _chk(w, name="w"), _chk(offsets, torch.int32, "offsets"), _chk(h, name="h")
    if w.dim() != 3 or not w.is_contiguous() or w.shape[2] != dy.shape[1]:
        raise ValueError("grouped_gemm_dswiglu: w must be a contiguous [E, I, K] tensor")
    E, I, K = w.shape
    M = dy.shape[0]
    if h.shape != (M, 2 * I) or not h.is_contiguous():
        raise ValueError("grouped_gemm_dswiglu: h must be a contiguous [M, 2I] tensor")
    dh = torch.empty_like(h)
    hip.get_lib().call("aria_grouped_gemm_dswiglu_bf16", _p(dy), _p(w), _p(h), _p(dh), _p(offsets), E, M, I, K, _rowmajor_2d(dy, "dy"), K,
                       I * K, 2 * I, 2 * I, _stream(dy))
    return dh


def gemm_dswiglu(dy: torch.Tensor, w: torch.Tensor, h: torch.Tensor, *, b_oc: bool = True) -> torch.Tensor:
    """The dense counterpart (SharedExpertMLP): dy [M, K], w = down_proj.weight [K, I] (``b_oc``) or [I, K], h [M, 2I] -> d_h [M, 2I];
    bit-identical to ``swiglu_bwd(h, gemm(dy, w, b_oc=b_oc))``."""
    _chk(dy, name="dy"), _chk(w, name="w"), _chk(h, name="h")
    M, K = dy.shape
    I = w.shape[1] if b_oc else w.shape[0]
    if (w.shape[0] if b_oc else w.shape[1]) != K or h.shape != (M, 2 * I) or not h.is_contiguous():
        raise ValueError("gemm_dswiglu: shapes")
    dh = torch.empty_like(h)
    hip.get_lib().call("aria_gemm_dswiglu_bf16", _p(dy), _p(w), _p(h), _p(dh), M, I, K, int(b_oc), _rowmajor_2d(dy, "dy"), _rowmajor_2d(w, "w"),
                       2 * I, 2 * I, _stream(dy))
    return dh


def grouped_gemm_wgrad(a: torch.Tensor, dy: torch.Tensor, offsets: torch.Tensor, E: int, *,
                       out: Optional[torch.Tensor] = None, out_dtype=bf16, accumulate: bool = False) -> torch.Tensor:
    """dW[e] = a[s_e:s_e+n_e]^T @ dy[s_e:s_e+n_e]  -> [E, K, N]."""
    _chk(a, name="a"), _chk(dy, name="dy"), _chk(offsets, torch.int32, "offsets")
    K, N = a.shape[1], dy.shape[1]
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an existing `out`")
        out = torch.empty((E, K, N), dtype=out_dtype, device=a.device)
    if not out.is_contiguous() or out.shape != (E, K, N):
        raise ValueError("grouped_gemm_wgrad: out must be contiguous [E,K,N]")
    c_f32 = {bf16: 0, torch.float32: 1}[out.dtype]
    if a.shape[0] == 0:  # no rows at all: every expert's gradient is zero
        return out if accumulate else out.zero_()
    hip.get_lib().call("aria_grouped_gemm_wgrad_bf16", _p(a), _p(dy), _p(out), _p(offsets), E, K, N, _rowmajor_2d(a, "a"),
                       _rowmajor_2d(dy, "dy"), c_f32, int(accumulate), _stream(a))
    return out


# ----------------------------------------------------------------------------- MoE routing / dispatch
def moe_route(logits: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> scores [T,k] (logits dtype), indices int32 [T,k], counts int32 [E]."""
    if logits.dtype not in (bf16, torch.float32) or not logits.is_contiguous() or logits.dim() != 2:
        raise ValueError("moe_route: logits must be contiguous [T,E] bf16/fp32")
    T, E = logits.shape
    scores = torch.empty((T, k), dtype=logits.dtype, device=logits.device)
    idx = torch.empty((T, k), dtype=torch.int32, device=logits.device)
    counts = torch.empty((E,), dtype=torch.int32, device=logits.device)
    hip.get_lib().call("aria_moe_route", _p(logits), int(logits.dtype == torch.float32), _p(scores), _p(idx), _p(counts), T, E,
                       k, _stream(logits))
    return scores, idx, counts


def router_fusable(D: int, E: int, k: int) -> bool:
    """Shapes the one-launch router takes (K1: logits GEMM + top-k + softmax + histogram); ARIA_FUSE_ROUTER=0 keeps the two-step form."""
    import os

    return E in (32, 64) and D % 256 == 0 and k <= 8 and os.environ.get("ARIA_FUSE_ROUTER", "1") != "0"


def moe_router_fused(x: torch.Tensor, w: torch.Tensor, k: int):
    """TopKRouter.forward in one launch: x [T, D], w [E, D] -> (logits [T, E] bf16, scores [T, k] bf16, indices int32 [T, k], counts int32 [E]);
    bit-identical to ``gemm(x, w)`` + ``moe_route``."""
    _chk(x, name="x"), _chk(w, name="w")
    T, D = x.shape
    E = w.shape[0]
    if w.shape[1] != D or not w.is_contiguous():
        raise ValueError("moe_router_fused: w must be a contiguous [E, D] tensor")
    logits = torch.empty((T, E), dtype=bf16, device=x.device)
    scores = torch.empty((T, k), dtype=bf16, device=x.device)
    idx = torch.empty((T, k), dtype=torch.int32, device=x.device)
    counts = torch.empty((E,), dtype=torch.int32, device=x.device)
    hip.get_lib().call("aria_moe_router_fused", _p(x), _p(w), _p(logits), _p(scores), _p(idx), _p(counts), T, D, E, k, _rowmajor_2d(x, "x"),
                       _stream(x))
    return logits, scores, idx, counts


def moe_sort(indices: torch.Tensor, counts: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> offsets int32 [E+1], sorted_src int32 [T*k] (== reference sorted_indices), inv int32 [T*k]."""
    _chk(indices, torch.int32, "indices"), _chk(counts, torch.int32, "counts")
    T, k = indices.shape
    E = counts.shape[0]
    M = T * k
    dev = indices.device
    offsets = torch.empty((E + 1,), dtype=torch.int32, device=dev)
    sorted_src = torch.empty((M,), dtype=torch.int32, device=dev)
    inv = torch.empty((M,), dtype=torch.int32, device=dev)
    nchunks = (M + 2047) // 2048
    ws = torch.empty((max(1, nchunks) * 128 + 64,), dtype=torch.int32, device=dev)
    hip.get_lib().call("aria_moe_sort", _p(indices), _p(counts), _p(offsets), _p(sorted_src), _p(inv), _p(ws), T, E, k,
                       _stream(indices))
    return offsets, sorted_src, inv


def moe_permute(x: torch.Tensor, sorted_src: torch.Tensor, k: int) -> torch.Tensor:
    _chk(x, name="x"), _chk(sorted_src, torch.int32, "sorted_src")
    ldx = _rowmajor_2d(x, "x")
    M, D = sorted_src.shape[0], x.shape[1]
    out = torch.empty((M, D), dtype=bf16, device=x.device)
    hip.get_lib().call("aria_moe_permute", _p(x), _p(sorted_src), _p(out), M, D, k, ldx, _stream(x))
    return out


def moe_unpermute(expert_out: torch.Tensor, inv: torch.Tensor, scores: Optional[torch.Tensor], k: int,
                  add: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(expert_out, name="expert_out"), _chk(inv, torch.int32, "inv")
    if not expert_out.is_contiguous():
        raise ValueError("moe_unpermute: expert_out must be contiguous")
    M, D = expert_out.shape
    T = M // k
    if scores is not None:
        _chk(scores, name="scores")
        assert scores.is_contiguous() and scores.shape == (T, k)
    if add is not None:
        _chk(add, name="add")
        assert add.is_contiguous() and add.shape == (T, D)
    out = torch.empty((T, D), dtype=bf16, device=expert_out.device)
    if residual is not None:   # the decoder layer's `h + moe(h)` as the kernel's last step (same two roundings as unpermute + add)
        _chk(residual, name="residual")
        assert residual.is_contiguous() and residual.shape == (T, D)
        lib = hip.get_lib()
        rc = lib.cdll.aria_moe_unpermute_res(_p(expert_out), _p(inv), _p(scores), _p(add), _p(residual), _p(out), T, D, k, _stream(expert_out))
        if rc == 3:   # ARIA_ERR_UNSUPPORTED: the generic-width kernel has no residual step
            lib.call("aria_moe_unpermute", _p(expert_out), _p(inv), _p(scores), _p(add), _p(out), T, D, k, _stream(expert_out))
            return globals()["add"](residual, out)
        if rc != 0:
            raise hip.AriaHipError(f"aria_moe_unpermute_res failed: {hip.ERRORS.get(rc, rc)}")
        return out
    hip.get_lib().call("aria_moe_unpermute", _p(expert_out), _p(inv), _p(scores), _p(add), _p(out), T, D, k, _stream(expert_out))
    return out


def moe_unpermute_bwd(dout: torch.Tensor, expert_out: torch.Tensor, inv: torch.Tensor, scores: torch.Tensor, k: int):
    _chk(dout, name="dout"), _chk(expert_out, name="expert_out"), _chk(scores, name="scores")
    assert dout.is_contiguous() and expert_out.is_contiguous() and scores.is_contiguous()
    T, D = dout.shape
    d_eo = torch.empty_like(expert_out)
    dscores = torch.empty_like(scores)
    hip.get_lib().call("aria_moe_unpermute_bwd", _p(dout), _p(expert_out), _p(inv), _p(scores), _p(d_eo), _p(dscores), T, D, k,
                       _stream(dout))
    return d_eo, dscores


def moe_route_bwd(logits, indices, scores, dscores, counts, z_coeff=0.0, aux_coeff=0.0, aux_scale=1.0) -> torch.Tensor:
    for t, n in ((logits, "logits"), (scores, "scores"), (dscores, "dscores")):
        _chk(t, name=n)
        assert t.is_contiguous()
    T, E = logits.shape
    k = indices.shape[1]
    dlogits = torch.empty_like(logits)
    hip.get_lib().call("aria_moe_route_bwd", _p(logits), _p(indices), _p(scores), _p(dscores), _p(counts), _p(dlogits), T, E, k,
                       float(z_coeff), float(aux_coeff), float(aux_scale), _stream(logits))
    return dlogits


def embedding_bwd(dy: torch.Tensor, ids: torch.Tensor, dw: torch.Tensor) -> torch.Tensor:
    """dw[ids[t]] += dy[t] (dw bf16 [V,D], zero-initialised by the caller)."""
    _chk(dy, name="dy"), _chk(ids, torch.int32, "ids"), _chk(dw, name="dw")
    assert dy.is_contiguous() and dw.is_contiguous() and ids.is_contiguous()
    hip.get_lib().call("aria_embedding_bwd", _p(dy), _p(ids), _p(dw), dy.shape[0], dy.shape[1], _stream(dy))
    return dw


def swiglu(h: torch.Tensor, h2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """h [M,2I] -> silu(h[:, :I]) * h[:, I:]   or   (gate [M,I], up [M,I])."""
    _chk(h, name="h")
    assert h.is_contiguous() and (h2 is None or (h2.is_contiguous() and h2.shape == h.shape))
    M = h.shape[0]
    I = h.shape[1] if h2 is not None else h.shape[1] // 2
    act = torch.empty((M, I), dtype=bf16, device=h.device)
    hip.get_lib().call("aria_swiglu_fwd", _p(h), _p(h2), _p(act), M, I, _stream(h))
    return act


def swiglu_bwd(h: torch.Tensor, dact: torch.Tensor, h2: Optional[torch.Tensor] = None):
    _chk(h, name="h"), _chk(dact, name="dact")
    assert h.is_contiguous() and dact.is_contiguous()
    M, I = dact.shape
    dh = torch.empty_like(h)
    dh2 = None if h2 is None else torch.empty_like(h2)
    hip.get_lib().call("aria_swiglu_bwd", _p(h), _p(h2), _p(dact), _p(dh), _p(dh2), M, I, _stream(h))
    return dh if h2 is None else (dh, dh2)


# ----------------------------------------------------------------------------- norm / rope / elementwise
def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, residual: Optional[torch.Tensor] = None, want_rstd: bool = True):
    """-> (y, h, rstd): h = x (+ residual) is the tensor that was normalised (new residual stream)."""
    _chk(x, name="x"), _chk(w, name="w")
    assert x.is_contiguous() and x.dim() == 2
    T, D = x.shape
    y = torch.empty_like(x)
    h = torch.empty_like(x) if residual is not None else x
    rstd = torch.empty((T,), dtype=torch.float32, device=x.device) if want_rstd else None
    hip.get_lib().call("aria_rmsnorm_fwd", _p(x), _p(residual), _p(w), _p(h) if residual is not None else None, _p(y), _p(rstd),
                       T, D, float(eps), _stream(x))
    return y, h, rstd


def rmsnorm_bwd(dy, h, w, rstd, dres: Optional[torch.Tensor] = None, dw_out: Optional[torch.Tensor] = None,
                accumulate: bool = False):
    """-> (dx, dw).  dx includes dres (gradient arriving through the residual stream) when given."""
    T, D = h.shape
    nblocks = max(1, min(512, (T + 3) // 4))
    dx = torch.empty_like(h)
    partial = torch.empty((nblocks, D), dtype=torch.float32, device=h.device)
    hip.get_lib().call("aria_rmsnorm_bwd", _p(dy), _p(h), _p(w), _p(rstd), _p(dres), _p(dx), _p(partial), nblocks, T, D,
                       _stream(h))
    if dw_out is None:
        dw_out = torch.empty((D,), dtype=bf16, device=h.device)
        accumulate = False
    hip.get_lib().call("aria_colsum_f32", _p(partial), _p(dw_out), nblocks, D, int(accumulate), _stream(h))
    return dx, dw_out


def scale_(x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """x *= s in place, s a one-element fp32 tensor ON THE DEVICE (an autograd upstream gradient): no host read, and no memory traffic when
    s == 1 (aria_scale_bf16).  Falls back to torch for shapes the kernel does not take (n % 8, other dtypes)."""
    if (x.dtype == bf16 and x.is_contiguous() and x.numel() % 8 == 0 and s.numel() == 1 and s.dtype == torch.float32 and s.device == x.device
            and x.data_ptr() % 16 == 0):
        hip.get_lib().call("aria_scale_bf16", _p(x), _p(s), x.numel(), _stream(x))
        return x
    return x.mul_(s)


def rope_(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, S: int, n_heads: int, hd: int, inverse: bool = False):
    """In-place half-split RoPE on the first n_heads*hd columns of the 2-D view x [T, >= n_heads*hd]."""
    _chk(x, name="x"), _chk(cos, name="cos"), _chk(sin, name="sin")
    ld = _rowmajor_2d(x, "x")
    assert cos.is_contiguous() and sin.is_contiguous() and cos.shape[1] == hd and cos.shape[0] >= S
    hip.get_lib().call("aria_rope_inplace", _p(x), _p(cos), _p(sin), x.shape[0], S, n_heads, hd, ld, int(inverse), _stream(x))
    return x


def gemm_qkv_rope(x: torch.Tensor, wqkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, S: int, hd: int,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LlamaAttention's q | k | v projections + apply_rotary_pos_emb (modeling_llama.py:243-281, 130-160) in one launch: x [T, K], wqkv [3D, K]
    (q rows, k rows, v rows) -> qkv [T, 3D] with the q and k columns rotated (half-split form, position t % S).  Bit-identical to ``gemm`` +
    ``rope_``; shapes the fused launch does not take (or ARIA_FUSE_QKV_ROPE=0) run exactly those two."""
    import os

    _chk(x, name="x"), _chk(wqkv, name="wqkv"), _chk(cos, name="cos"), _chk(sin, name="sin")
    T, K = x.shape
    D = wqkv.shape[0] // 3
    if wqkv.shape != (3 * D, K) or D % hd:
        raise ValueError("gemm_qkv_rope: wqkv [3D, K] with D a multiple of the head dim")
    if out is None:
        out = torch.empty((T, 3 * D), dtype=bf16, device=x.device)
    assert cos.is_contiguous() and sin.is_contiguous() and cos.shape[1] == hd and cos.shape[0] >= S and sin.shape == cos.shape
    if os.environ.get("ARIA_FUSE_QKV_ROPE", "1") != "0" and D % 256 == 0 and 256 % hd == 0 and hd % 16 == 0 and K % 64 == 0 and K >= 64:
        rc = hip.get_lib().cdll.aria_gemm_qkv_rope_hf_bf16(_p(x), _p(wqkv), _p(out), _p(cos), _p(sin), T, D, K, hd, S, _rowmajor_2d(x, "x"),
                                                           _rowmajor_2d(wqkv, "wqkv"), _rowmajor_2d(out, "out"), _stream(x))
        if rc == 0:
            return out
        if rc != 3:
            raise hip.AriaHipError(f"aria_gemm_qkv_rope_hf_bf16 failed: {hip.ERRORS.get(rc, rc)}")
    gemm(x, wqkv, out=out)
    rope_(out[:, :2 * D], cos, sin, S, 2 * D // hd, hd)
    return out


def rope_interleaved_(x: torch.Tensor, freqs_cis: torch.Tensor, n_heads: int, hd: int, pos: Optional[torch.Tensor] = None):
    """gptfast RoPE in place on the first n_heads*hd columns of x [T, >=n_heads*hd]; freqs_cis bf16 [S, hd/2, 2]."""
    _chk(x, name="x"), _chk(freqs_cis, name="freqs_cis")
    assert freqs_cis.is_contiguous() and freqs_cis.shape[1] * 2 == hd
    if pos is not None:
        _chk(pos, torch.int32, "pos")
        assert pos.numel() == x.shape[0]
    hip.get_lib().call("aria_rope_interleaved_inplace", _p(x), _p(freqs_cis), _p(pos), x.shape[0], freqs_cis.shape[0], n_heads, hd,
                       _rowmajor_2d(x, "x"), _stream(x))
    return x


def adamw_step_(param: torch.Tensor, grad: torch.Tensor, master: torch.Tensor, m: torch.Tensor, v: torch.Tensor, *, lr: float,
                beta1: float, beta2: float, eps: float, weight_decay: float, step: int, grad_scale: float = 1.0):
    """In-place AdamW on flat contiguous views (bf16 param/grad, fp32 master/m/v) of equal, even length."""
    _chk(param, name="param"), _chk(grad, name="grad")
    for t in (master, m, v):
        _chk(t, torch.float32, "state")
    n = param.numel()
    assert all(t.is_contiguous() and t.numel() == n for t in (param, grad, master, m, v))
    hip.get_lib().call("aria_adamw_step", _p(param), _p(grad), _p(master), _p(m), _p(v), n, float(lr), float(beta1), float(beta2),
                       float(eps), float(weight_decay), int(step), float(grad_scale), _stream(param))
    # the kernel wrote param / master / m / v behind autograd's back: bump the version counters so that anything keyed on them
    # (the ViT's cached fused q/k/v weight, saved-tensor checks) sees the update
    torch.autograd.graph.increment_version(param)


def sumsq_(x: torch.Tensor, out: torch.Tensor, workspace: torch.Tensor, accumulate: bool = True) -> None:
    """out[0] (+)= sum(x^2) in fp32 over a contiguous bf16 tensor (one term of the clipping norm); ``workspace``: fp32 [1024]."""
    _chk(x, name="x"), _chk(out, torch.float32, "out"), _chk(workspace, torch.float32, "workspace")
    assert x.is_contiguous() and workspace.numel() >= 1024 and out.numel() == 1
    hip.get_lib().call("aria_sumsq_bf16", _p(x), x.numel(), _p(out), int(accumulate), _p(workspace), _stream(x))


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _chk(a, name="a"), _chk(b, name="b")
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    out = torch.empty_like(a)
    hip.get_lib().call("aria_add_bf16", _p(a), _p(b), _p(out), a.numel(), _stream(a))
    return out


# ----------------------------------------------------------------------------- ViT / projector support
def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, want_stats: bool = True):
    _chk(x, name="x"), _chk(w, name="w"), _chk(b, name="b")
    assert x.is_contiguous() and x.dim() == 2
    T, D = x.shape
    y = torch.empty_like(x)
    mean = torch.empty((T,), dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty((T,), dtype=torch.float32, device=x.device) if want_stats else None
    hip.get_lib().call("aria_layernorm_fwd", _p(x), _p(w), _p(b), _p(y), _p(mean), _p(rstd), T, D, float(eps), _stream(x))
    return y, mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd):
    """-> (dx, dw bf16 [D], db bf16 [D])"""
    T, D = x.shape
    nblocks = max(1, min(512, (T + 3) // 4))
    dx = torch.empty_like(x)
    pw = torch.empty((nblocks, D), dtype=torch.float32, device=x.device)
    pb = torch.empty((nblocks, D), dtype=torch.float32, device=x.device)
    hip.get_lib().call("aria_layernorm_bwd", _p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(dx), _p(pw), _p(pb), nblocks, T, D,
                       _stream(x))
    dw = torch.empty((D,), dtype=bf16, device=x.device)
    db = torch.empty((D,), dtype=bf16, device=x.device)
    hip.get_lib().call("aria_colsum_f32", _p(pw), _p(dw), nblocks, D, 0, _stream(x))
    hip.get_lib().call("aria_colsum_f32", _p(pb), _p(db), nblocks, D, 0, _stream(x))
    return dx, dw, db


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    _chk(x, name="x")
    assert x.is_contiguous()
    y = torch.empty_like(x)
    hip.get_lib().call("aria_gelu_tanh_fwd", _p(x), _p(y), x.numel(), _stream(x))
    return y


def gelu_tanh_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    assert x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    hip.get_lib().call("aria_gelu_tanh_bwd", _p(x), _p(dy), _p(dx), x.numel(), _stream(x))
    return dx


def vit_patch_mask(pixel_mask: torch.Tensor, patch: int) -> torch.Tensor:
    """pixel_mask bool/uint8 [N,R,R] -> uint8 [N, R/patch, R/patch]"""
    pm = pixel_mask.to(torch.uint8).contiguous()
    N, R, _ = pm.shape
    out = torch.empty((N, R // patch, R // patch), dtype=torch.uint8, device=pm.device)
    hip.get_lib().call("aria_vit_patch_mask", _p(pm), _p(out), N, R, patch, _stream(pm))
    return out


def vit_pos_ids(patch_mask: torch.Tensor, boundaries: torch.Tensor, n_side: int) -> torch.Tensor:
    _chk(patch_mask, torch.uint8, "patch_mask"), _chk(boundaries, torch.float32, "boundaries")
    N, Hp, Wp = patch_mask.shape
    ids = torch.empty((N, Hp * Wp), dtype=torch.int32, device=patch_mask.device)
    hip.get_lib().call("aria_vit_pos_ids", _p(patch_mask), _p(boundaries), _p(ids), N, Hp, Wp, n_side, _stream(patch_mask))
    return ids


def vit_im2col(pixels: torch.Tensor, patch: int, KP: int) -> torch.Tensor:
    assert pixels.is_contiguous() and pixels.dtype in (bf16, torch.float32)
    N, C, R, _ = pixels.shape
    Hp = R // patch
    out = torch.empty((N * Hp * Hp, KP), dtype=bf16, device=pixels.device)
    hip.get_lib().call("aria_vit_im2col", _p(pixels), int(pixels.dtype == torch.float32), _p(out), N, C, R, patch, KP, _stream(pixels))
    return out


def gather_add_rows_(x: torch.Tensor, table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    _chk(x, name="x"), _chk(table, name="table"), _chk(ids, torch.int32, "ids")
    assert x.is_contiguous() and table.is_contiguous() and ids.is_contiguous()
    hip.get_lib().call("aria_gather_add_rows", _p(x), _p(table), _p(ids), x.shape[0], x.shape[1], _stream(x))
    return x


def colsum(x: torch.Tensor) -> torch.Tensor:
    """bf16 [T,D] -> bf16 [D] column sums (bias gradient), fp32 accumulation."""
    _chk(x, name="x")
    T, D = x.shape
    nparts = max(1, min(64, T // 64))
    partial = torch.empty((nparts, D), dtype=torch.float32, device=x.device)
    hip.get_lib().call("aria_colsum_bf16", _p(x), _p(partial), nparts, T, D, _rowmajor_2d(x, "x"), _stream(x))
    out = torch.empty((D,), dtype=bf16, device=x.device)
    hip.get_lib().call("aria_colsum_f32", _p(partial), _p(out), nparts, D, 0, _stream(x))
    return out


# ----------------------------------------------------------------------------- attention
def attention_fwd(q, k, v, B: int, S: int, H: int, hd: int, scale: float, causal: bool,
                  kv_len: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                  key_mask: Optional[torch.Tensor] = None, Skv: Optional[int] = None):
    """q: [B*S, >= H*hd], k,v: [B*Skv, >= H*hd] token-major views (head h at columns h*hd);
    key_mask uint8 [B, Skv] (1 = attend).  -> (o [B*S, H*hd], lse fp32 [B,H,S])."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, name=n)
    Skv = S if Skv is None else Skv
    if key_mask is not None:
        _chk(key_mask, torch.uint8, "key_mask")
        assert key_mask.is_contiguous() and key_mask.numel() == B * Skv
    if out is None:
        out = torch.empty((B * S, H * hd), dtype=bf16, device=q.device)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
    hip.get_lib().call("aria_attn_fwd", _p(q), _p(k), _p(v), _p(out), _p(lse), _p(kv_len), _p(key_mask), B, S, Skv, H, hd,
                       _rowmajor_2d(q, "q"), _rowmajor_2d(k, "k"), _rowmajor_2d(v, "v"), _rowmajor_2d(out, "o"), float(scale),
                       int(causal), _stream(q))
    return out, lse


def attention_bwd(q, k, v, o, do, lse, B: int, S: int, H: int, hd: int, scale: float, causal: bool,
                  kv_len: Optional[torch.Tensor] = None, dq=None, dk=None, dv=None, key_mask: Optional[torch.Tensor] = None,
                  Skv: Optional[int] = None, rope=None):
    """Two kernels (dK/dV per key block, dQ per query block: 7 GEMM units of S x S x hd per head, deterministic).  A single-pass form with
    fp32 atomic dQ accumulation was built and measured slower at every length (profiles/r03_attention_notes.md).
    ``rope`` = (cos, sin) bf16 [S_rope, hd] tables: q / k are the ROTATED tensors of LlamaAttention.forward and dq / dk leave with the inverse
    half-split rotation applied in the kernels' register epilogues (hd 128; bit-identical to ``rope_(.., inverse=True)`` afterwards)."""
    dev = q.device
    Skv = S if Skv is None else Skv
    if dq is None:
        dq = torch.empty((B * S, H * hd), dtype=bf16, device=dev)
    if dk is None:
        dk = torch.empty((B * Skv, H * hd), dtype=bf16, device=dev)
    if dv is None:
        dv = torch.empty((B * Skv, H * hd), dtype=bf16, device=dev)
    delta = torch.empty((B, H, S), dtype=torch.float32, device=dev)
    if rope is not None:
        cos, sin = rope
        _chk(cos, name="rope cos"), _chk(sin, name="rope sin")
        assert cos.is_contiguous() and sin.is_contiguous() and cos.shape == sin.shape and cos.dim() == 2 and cos.shape[1] == hd, "rope tables"
        hip.get_lib().call("aria_attn_bwd_rope", _p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv),
                           _p(kv_len), _p(key_mask), B, S, Skv, H, hd, _rowmajor_2d(q, "q"), _rowmajor_2d(k, "k"), _rowmajor_2d(v, "v"),
                           _rowmajor_2d(o, "o"), _rowmajor_2d(dq, "dq"), _rowmajor_2d(dk, "dk"), _rowmajor_2d(dv, "dv"), float(scale),
                           int(causal), _p(cos), _p(sin), cos.shape[0], _stream(q))
        return dq, dk, dv
    hip.get_lib().call("aria_attn_bwd", _p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), _p(kv_len),
                       _p(key_mask), B, S, Skv, H, hd, _rowmajor_2d(q, "q"), _rowmajor_2d(k, "k"), _rowmajor_2d(v, "v"),
                       _rowmajor_2d(o, "o"), _rowmajor_2d(dq, "dq"), _rowmajor_2d(dk, "dk"), _rowmajor_2d(dv, "dv"), float(scale),
                       int(causal), _stream(q))
    return dq, dk, dv


def attention_bwd_rope_fusable(hd: int, S: int, cos: torch.Tensor) -> bool:
    """the attention backward's inverse-RoPE epilogue: the decoder's head dim, positions = index in the sequence (the tables cover S rows),
    ARIA_FUSE_QKV_ROPE=0 switches it off together with the forward's fused rotation"""
    return hd == 128 and cos.dim() == 2 and cos.shape[0] >= S and cos.shape[1] == hd and os.environ.get("ARIA_FUSE_QKV_ROPE", "1") != "0"


def decode_attention(qkv: torch.Tensor, freqs_cis: torch.Tensor, pos: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                     n_heads: int, hd: int, splits: int = 1) -> torch.Tensor:
    """One new token against the static KV cache (aria_decode_attn): qkv bf16 [3*H*hd] (not modified), caches bf16 [S_max, H*hd]
    (row pos[0] is written), pos int32 [1] on the device -> out bf16 [H*hd].  splits > 1: flash-decoding over contiguous key ranges."""
    for t, n in ((qkv, "qkv"), (freqs_cis, "freqs_cis"), (k_cache, "k_cache"), (v_cache, "v_cache")):
        _chk(t, name=n)
        assert t.is_contiguous(), n
    _chk(pos, torch.int32, "pos")
    D = n_heads * hd
    assert qkv.numel() == 3 * D and k_cache.shape[-1] == D and v_cache.shape == k_cache.shape and freqs_cis.shape[0] >= k_cache.shape[0]
    out = torch.empty(D, dtype=bf16, device=qkv.device)
    lib = hip.get_lib()
    nbytes = int(lib.cdll.aria_decode_attn_workspace_bytes(n_heads, hd, splits))
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=qkv.device)
    lib.call("aria_decode_attn", _p(qkv), _p(freqs_cis), _p(pos), _p(k_cache), _p(v_cache), _p(out), n_heads, hd, splits, _p(ws), nbytes,
             _stream(qkv))
    return out


# ----------------------------------------------------------------------------- loss
def cross_entropy(logits: torch.Tensor, labels: torch.Tensor, grad_scale: Optional[float] = None,
                  dlogits: Optional[torch.Tensor] = None, count_in: Optional[torch.Tensor] = None):
    """labels int32 [T] already shifted/masked (-100 ignore).  -> (loss_sum fp32[1], count int32[1], dlogits|None).
    dlogits may alias logits.  grad_scale multiplies (softmax - onehot)."""
    _chk(logits, name="logits"), _chk(labels, torch.int32, "labels")
    T, V = logits.shape
    loss_sum = torch.zeros((1,), dtype=torch.float32, device=logits.device)
    count = torch.zeros((1,), dtype=torch.int32, device=logits.device)
    hip.get_lib().call("aria_cross_entropy", _p(logits), _p(labels), _p(loss_sum), _p(count), _p(dlogits),
                       float(grad_scale if grad_scale is not None else 0.0), _p(count_in), T, V, _rowmajor_2d(logits, "logits"), _stream(logits))
    return loss_sum, count, dlogits
