"""Wall-clock marks inside gemm3_kernel (build with -DARIA_ABL=512): where one tile's time goes, per workgroup."""
import ctypes, json, os, sys, numpy as np, torch
root = os.environ.get("GRAFT_REPO_ROOT", ".")
os.environ["ARIA_GEMM_FORCE"] = "3"
lib = ctypes.CDLL(os.path.join(root, "build", "abl", os.environ.get("ABL_LIB", "libgemm_abl512.so")))
fn = lib.aria_gemm_bf16
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 3 + [ctypes.c_int] * 2 + [ctypes.c_int64] * 3 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
lib.aria_abl_ts.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev, bf16 = "cuda", torch.bfloat16
out_all = {}
for (M, N, K) in ((4096, 4096, 640), (4096, 4096, 2560), (16384, 8192, 640), (16384, 8192, 2560)):
    x = torch.randn(M, K, device=dev).to(bf16); w = (torch.randn(N, K, device=dev) * 0.02).to(bf16)
    out = torch.empty(M, N, dtype=bf16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        fn(x.data_ptr(), w.data_ptr(), out.data_ptr(), None, M, N, K, 0, 0, K, K, N, 0, 0, st)
    torch.cuda.synchronize()
    ntile = (M // 256) * (N // 256)
    n = min(ntile, 4096)
    ts = np.zeros(4096 * 8, dtype=np.uint64)
    lib.aria_abl_ts(ts.ctypes.data, 4096 * 8)
    full = ts.reshape(4096, 8)[:n].astype(np.int64)
    t = full[:, :6]
    clk = (full[:, 7] - full[:, 6]) / ((full[:, 2] - full[:, 1]) * 10.0)  # shader-counter ticks per ns inside the K loop
    t0 = t[:, 0].min()
    rel = (t - t0) * 0.01  # us
    seg = np.diff(rel, axis=1)
    names = ["entry->first data", "K loop", "pack+park+sync", "store issue", "store ack"]
    r = {"tiles": ntile, "kernel span us": round(float(rel[:, 5].max()), 2),
         "shader counter GHz inside the K loop p10/50/90": [round(float(np.percentile(clk, q)), 3) for q in (10, 50, 90)]}
    for i, nm in enumerate(names):
        r[nm] = [round(float(np.percentile(seg[:, i], q)), 2) for q in (10, 50, 90)]
    r["tile total (entry->ack) p10/50/90"] = [round(float(np.percentile(rel[:, 5] - rel[:, 0], q)), 2) for q in (10, 50, 90)]
    # rounds: sort by entry time
    order = np.argsort(rel[:, 0])
    ent = rel[order, 0]
    r["entry time of workgroup #0,255,256,511,512 (sorted)"] = [round(float(ent[i]), 2) for i in (0, 255, 256, 511, 512) if i < n]
    # gap between a tile's end on a CU and the next entry: approx = entry of k-th (k>=256) minus the (k-256)-th earliest ack
    ack = np.sort(rel[:, 5])
    if n > 256:
        gaps = ent[256:] - ack[: n - 256]
        r["turnover gap (next entry - matching ack) p10/50/90"] = [round(float(np.percentile(gaps, q)), 2) for q in (10, 50, 90)]
    out_all[f"{M}x{N}x{K}"] = r
print(json.dumps(out_all))
