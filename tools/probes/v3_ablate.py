"""Timing ablation of the v3 K loop (build/abl/libgemm_abl<N>.so built with -DARIA_ABL=N; results are garbage, only the time matters):
shader cycles (2.28 GHz, the clock the in-kernel counter showed) per K-tile per CU at 8192^3, rc,rc and rc,oc."""
import ctypes, glob, json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ["ARIA_GEMM_FORCE"] = "3"
dev, bf16 = "cuda", torch.bfloat16
M = N = K = 8192
x = torch.randn(M, K, device=dev).to(bf16); w = (torch.randn(N, K, device=dev) * 0.02).to(bf16); out = torch.empty(M, N, dtype=bf16, device=dev)


def run(lib, b_oc):
    fn = lib.aria_gemm_bf16
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] * 3 + [ctypes.c_int] * 2 + [ctypes.c_int64] * 3 + [ctypes.c_int] * 2 + [ctypes.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    def call():
        rc = fn(x.data_ptr(), w.data_ptr(), out.data_ptr(), None, M, N, K, 0, b_oc, K, N if b_oc else K, N, 0, 0, st)
        assert rc == 0, rc
    for _ in range(3): call()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): call()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10 * 1e-3
    return round(t * 2.28e9 / (4 * (K // 64)))


names = {0: "full", 1: "no MFMA", 2: "no fragment reads", 4: "no DMA", 8: "no barriers", 6: "MFMA + barriers only", 5: "fragment reads + barriers only",
         3: "DMA + waits + barriers only", 2048: "every workgroup loads tile (0,0) (all L2 hits)", 12: "MFMA + fragment reads, no barriers", 14: "MFMA only, no barriers", 32: "full without vmcnt waits"}
root = os.environ.get("GRAFT_REPO_ROOT", ".")
res = {}
libs = [(0, os.path.join(root, "aria_amd", "libaria_hip.so"))] + sorted((int(f.split("abl")[-1].split(".")[0]), f) for f in glob.glob(os.path.join(root, "build", "abl", "libgemm_abl*.so")))
for n, f in libs:
    if n not in names: continue
    lib = ctypes.CDLL(f)
    res[f"{n}: {names[n]}"] = {"rc,rc": run(lib, 0), "rc,oc": run(lib, 1)}
print(json.dumps(res))
