"""Timing ablation of the v4 K loop (compile-time ARIA_ABL variants built into build/abl/libgemm_abl<N>.so by hand; results are garbage,
only the time matters): cycles per K-tile per CU at 2.0 GHz for 8192^3 rc,rc."""
import ctypes, glob, json, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ["ARIA_GEMM_FORCE"] = "3"
dev = "cuda"; bf16 = torch.bfloat16
M = N = K = 8192
x = torch.randn(M, K, device=dev).to(bf16); w = (torch.randn(N, K, device=dev) * 0.02).to(bf16); out = torch.empty(M, N, dtype=bf16, device=dev)
def run(lib, v4):
    os.environ["ARIA_GEMM_V4"] = v4
    fn = lib.aria_gemm_bf16
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 3 + [ctypes.c_int] * 2 + [ctypes.c_int64] * 3 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    def call():
        rc = fn(x.data_ptr(), w.data_ptr(), out.data_ptr(), None, M, N, K, 0, 0, K, K, N, 0, 0, st)
        assert rc == 0, rc
    for _ in range(3): call()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): call()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10 * 1e-3
    tiles_per_cu = (M // 256) * (N // 256) / 256
    cyc = t * 2.0e9 / (tiles_per_cu * (K // 64))
    return round(2 * M * N * K / t / 1e12, 1), round(cyc)
res = {}
root = os.environ.get("GRAFT_REPO_ROOT", ".")
base = ctypes.CDLL(os.path.join(root, "aria_amd", "libaria_hip.so"))
res["v3 full"] = run(base, "0"); res["v4 full"] = run(base, "1")
names = {3: "DMA + waits + barriers only", 35: "DMA issue + barriers only (no waits)", 34: "MFMA + DMA issue, no waits, no reads", 32: "full without vmcnt waits", 16: "DMA inside the MFMA section", 1: "no MFMA", 2: "no frag reads", 4: "no DMA", 8: "no barriers", 6: "MFMA + barriers only", 7: "barriers only", 14: "MFMA only, no barriers"}
for f in sorted(glob.glob(os.path.join(root, "build", "abl", "libgemm_abl*.so"))):
    n = int(f.split("abl")[-1].split(".")[0])
    res[f"v4 abl{n}: {names.get(n, n)}"] = run(ctypes.CDLL(f), "1")
print(json.dumps(res, indent=1))
