"""r06: where a tile's time goes in the SIX grouped launches of one MoE layer at the benchmark's shape (16 384 tokens x top-6 -> 98 304 routed
rows, 64 experts, D 2560, I 1664) -- wall-clock marks inside gemm3_kernel (library built with -DARIA_ABL=512 and copied over the product
library on the box: tools/gpu_session.sh `lib=build/abl/libaria_gemm3_512.so`).  Per launch: plain duration (HIP events, 10 launches), and
from the marks of up to 16 384 workgroups: entry -> first operands, K loop, K loop end -> parked, parked -> stores issued, store
acknowledgement, the shader clock inside the K loop, K-tiles per tile -- medians and p10 / p90 in microseconds -- plus the fitted line
tile time = fixed + per-K-tile x K-tiles over the launch's workgroups (the weight gradients' experts differ in K).  One JSON line."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

root = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, root)
from aria_amd import hip, ops  # noqa: E402

lib = hip.get_lib()
lib.cdll.aria_abl_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.cdll.aria_abl_hw.argtypes = [ctypes.c_void_p, ctypes.c_int]
NWG = 16384
dev, bf16 = "cuda", torch.bfloat16
T, D, I, E, k = 16384, 2560, 1664, 64, 6
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((T, D), generator=g, device=dev).to(bf16)
logits = torch.randn((T, E), generator=g, device=dev).to(bf16)
scores, idx, counts = ops.moe_route(logits, k)
off, sorted_src, inv = ops.moe_sort(idx, counts)
rows = ops.permuted_token_rows(sorted_src, k)
M = rows.numel()
w1 = (torch.randn((E, D, 2 * I), generator=g, device=dev) * 0.02).to(bf16)
w2 = (torch.randn((E, I, D), generator=g, device=dev) * 0.02).to(bf16)
h, act = ops.grouped_gemm_swiglu_gather(x, rows, w1, off, want_h=True)
dy = torch.randn((M, D), generator=g, device=dev).to(bf16)
dh = ops.grouped_gemm_dswiglu(dy, w2, off, h)
res = {"shape": {"T": T, "D": D, "I": I, "E": E, "topk": k, "rows": M}, "expert rows min/median/max": [int(counts.min()), int(counts.median()), int(counts.max())]}

launches = {
    "fc1 + SwiGLU, gathered rows  gemm3<rc,oc,8>": (lambda: ops.grouped_gemm_swiglu_gather(x, rows, w1, off, want_h=True), 2.0 * M * D * 2 * I),
    "fc2 forward  gemm3<rc,oc,3>": (lambda: ops.grouped_gemm(act, w2, off), 2.0 * M * I * D),
    "fc2 dgrad + dSwiGLU  gemm3<rc,rc,5>": (lambda: ops.grouped_gemm_dswiglu(dy, w2, off, h), 2.0 * M * I * D),
    "fc1 dgrad  gemm3<rc,rc,3>": (lambda: ops.grouped_gemm(dh, w1, off, w_is_kn=False), 2.0 * M * D * 2 * I),
    "fc1 wgrad, gathered reduction rows  gemm3<oc,oc,11>": (lambda: ops.grouped_gemm_wgrad_gather(x, rows, dh, off, E), 2.0 * M * D * 2 * I),
    "fc2 wgrad  gemm3<oc,oc,3>": (lambda: ops.grouped_gemm_wgrad(act, dy, off, E), 2.0 * M * I * D),
}


def pct(v):
    return [round(float(np.percentile(v, q)), 2) for q in (10, 50, 90)]


for name, (launch, flops) in launches.items():
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        launch()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100.0
    # marks of the LAST launch (every launch overwrites its workgroups' slots; stale slots of a larger earlier grid are filtered by entry time)
    zero = np.zeros(NWG * 8, dtype=np.uint64)
    launch()
    torch.cuda.synchronize()
    ts = np.zeros(NWG * 8, dtype=np.uint64)
    lib.cdll.aria_abl_ts(ts.ctypes.data, NWG * 8)
    hw = np.zeros(NWG, dtype=np.uint32)
    lib.cdll.aria_abl_hw(hw.ctypes.data, NWG)
    full = ts.reshape(NWG, 8).astype(np.int64)
    t_last = full[:, 0].max()
    keep = (full[:, 0] > t_last - 1_000_000) & (full[:, 5] > full[:, 0]) & (full[:, 2] > full[:, 1])   # entered within the last 10 ms, ran a tile
    full, hw = full[keep], hw[keep]
    rel = (full[:, :6] - full[:, 0].min()) * 0.01
    seg = {"entry -> first operands": rel[:, 1] - rel[:, 0], "K loop": rel[:, 2] - rel[:, 1], "K loop end -> parked": rel[:, 3] - rel[:, 2],
           "parked -> stores issued": rel[:, 4] - rel[:, 3], "store ack": rel[:, 5] - rel[:, 4], "tile total": rel[:, 5] - rel[:, 0]}
    clk = (full[:, 7] - full[:, 6]) / np.maximum((full[:, 2] - full[:, 1]) * 10.0, 1)
    r = {"launch us (10 launches)": round(us, 1), "TF/s": round(flops / us / 1e6, 1), "workgroups sampled": int(full.shape[0])}
    r.update({k2: pct(v) for k2, v in seg.items()})
    r["shader clock GHz inside the K loop p50"] = round(float(np.percentile(clk, 50)), 3)
    # full tiles only (K loop >= 60 % of the median): edge tiles skip MFMAs
    kl = seg["K loop"]
    fullt = kl > 0.6 * np.median(kl)
    r["full tiles: K loop / tile total p50"] = [round(float(np.median(kl[fullt])), 2), round(float(np.median(seg["tile total"][fullt])), 2)]
    r["edge tiles (K loop < 60 % of the median)"] = int((~fullt).sum())
    span = rel[:, 5].max() - rel[:, 0].min()
    busy = seg["tile total"].sum() / 256.0
    r["sampled span us / sum of tile totals per CU us"] = [round(float(span), 1), round(float(busy), 1)]
    r["share of tile time (sum over workgroups)"] = {k2: round(float(v.sum() / seg["tile total"].sum()), 3) for k2, v in seg.items() if k2 != "tile total"}
    # where the CU time outside any tile goes: per CU (XCC, SE, SH, CU of HW_ID) the tiles in entry order -> idle before the first tile, gaps
    # between a tile's store acknowledgement and the next tile's entry, idle behind the last tile until the launch's last acknowledgement
    cu_key = ((hw >> 16) & 0xf) * 4096 + ((hw >> 8) & 0xff)   # XCC | SE, SH, CU
    t0, t1 = rel[:, 0].min(), rel[:, 5].max()
    lead, gaps, tail, ntile = [], [], [], []
    for key in np.unique(cu_key):
        sel = np.where(cu_key == key)[0]
        o = sel[np.argsort(rel[sel, 0])]
        lead.append(rel[o[0], 0] - t0)
        tail.append(t1 - rel[o[-1], 5])
        gaps.extend((rel[o[1:], 0] - rel[o[:-1], 5]).tolist())
        ntile.append(len(o))
    xcc = (hw >> 16) & 0xf
    r["CUs seen"] = int(len(lead))
    r["tiles per CU min/median/max"] = [int(np.min(ntile)), int(np.median(ntile)), int(np.max(ntile))]
    r["idle before a CU's first tile us p10/50/90"] = pct(np.array(lead))
    r["gap between tiles on a CU (ack -> next entry) us p10/50/90"] = pct(np.array(gaps))
    r["idle behind a CU's last tile us p10/50/90"] = pct(np.array(tail))
    r["CU time outside tiles: lead / gaps / tail, us per CU"] = [round(float(np.mean(lead)), 1), round(float(np.sum(gaps) / len(lead)), 1), round(float(np.mean(tail)), 1)]
    r["last acknowledgement per XCC, us after the launch's first entry"] = [round(float(rel[xcc == xc, 5].max() - t0), 1) for xc in np.unique(xcc)]
    res[name] = r
print(json.dumps(res))
