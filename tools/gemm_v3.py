"""v3 (LDS-DMA, phase-scheduled) vs v2 GEMM on hardware: bit-identical results (same k order per accumulator), repeated
launches as a race screen, and timings.  Usage: python tools/gemm_v3.py [orders...]"""
import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from aria_amd import ops, hip
from tools.microbench import timeit
bf16 = torch.bfloat16; dev = "cuda"
res = {"check": {}, "time": {}}
ops.GEMM_SPLIT_K = False  # bit-exact comparison against v2 needs the same summation order

def run(force, fn):
    os.environ["ARIA_GEMM_FORCE"] = force
    out = fn()
    assert hip.get_lib().cdll.aria_last_gemm_variant() == int(force), (force, hip.get_lib().cdll.aria_last_gemm_variant())
    return out

torch.manual_seed(0)
for (M, N, K) in ((256, 256, 64), (264, 136, 192), (1000, 520, 1152), (2048, 3328, 2560), (4096, 4096, 4096)):
    for a_oc, b_oc in ((0, 0), (0, 1), (1, 1)):
        Mm = (M + 7) // 8 * 8 if a_oc else M
        A = torch.randn(Mm, K, device=dev).to(bf16); B = torch.randn(K, N, device=dev).to(bf16)
        aa = A.t().contiguous() if a_oc else A
        bb = B if b_oc else B.t().contiguous()
        ref = run("2", lambda: ops.gemm(aa, bb, a_oc=bool(a_oc), b_oc=bool(b_oc)))
        bad = 0
        for rep in range(5):
            got = run("3", lambda: ops.gemm(aa, bb, a_oc=bool(a_oc), b_oc=bool(b_oc)))
            bad += int(not torch.equal(got, ref))
        res["check"][f"{M}x{N}x{K}_{a_oc}{b_oc}"] = bad
# grouped, Aria fc1 shape slice
E, K, N = 64, 2560, 3328
counts = torch.tensor([37 * (i % 5) * 20 + (i * 7) % 11 for i in range(E)])
off = torch.zeros(E + 1, dtype=torch.int32); off[1:] = torch.cumsum(counts, 0); M = int(off[-1])
a = torch.randn(M, K, device=dev).to(bf16); w = (torch.randn(E, K, N, device=dev) * 0.05).to(bf16); offd = off.to(dev)
ref = run("2", lambda: ops.grouped_gemm(a, w, offd))
bad = sum(int(not torch.equal(run("3", lambda: ops.grouped_gemm(a, w, offd)), ref)) for _ in range(5))
res["check"][f"grouped_fc1_M{M}"] = bad
dy = torch.randn(M, N, device=dev).to(bf16)
ref = run("2", lambda: ops.grouped_gemm(dy, w, offd, w_is_kn=False))
bad = sum(int(not torch.equal(run("3", lambda: ops.grouped_gemm(dy, w, offd, w_is_kn=False)), ref)) for _ in range(5))
res["check"][f"grouped_dgrad_M{M}"] = bad
print(json.dumps(res["check"]), flush=True)

orders = sys.argv[1:] or ["4"]
def tf(flops, fn):
    t = timeit(fn, 10, 3)
    return [round(flops / t / 1e12), round(t * 1e6)]
for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 8192), (16384, 6912, 2560)):
    for a_oc, b_oc in ((0, 0), (0, 1), (1, 1)):
        A = torch.randn(M, K, device=dev).to(bf16); B = torch.randn(K, N, device=dev).to(bf16)
        aa = A.t().contiguous() if a_oc else A
        bb = B if b_oc else B.t().contiguous()
        row = {}
        os.environ["ARIA_GEMM_ORDER"] = "2"
        os.environ["ARIA_GEMM_FORCE"] = "2"
        row["v2"] = tf(2 * M * N * K, lambda: ops.gemm(aa, bb, a_oc=bool(a_oc), b_oc=bool(b_oc)))
        os.environ["ARIA_GEMM_FORCE"] = "3"
        for o in orders:
            os.environ["ARIA_GEMM_ORDER"] = o
            row[f"v3_o{o}"] = tf(2 * M * N * K, lambda: ops.gemm(aa, bb, a_oc=bool(a_oc), b_oc=bool(b_oc)))
        res["time"][f"{M}x{N}x{K}_{a_oc}{b_oc}"] = row
        print(json.dumps({f"{M}x{N}x{K}_{a_oc}{b_oc}": row}), flush=True)
        del A, B, aa, bb
# fc1 / fc2 of config #3: 53248 routed rows over 64 experts, uneven like a real router
g = torch.Generator().manual_seed(1)
counts = torch.bincount(torch.randint(0, E, (53248,), generator=g), minlength=E)
off = torch.zeros(E + 1, dtype=torch.int32); off[1:] = torch.cumsum(counts, 0); M = int(off[-1])
a = torch.randn(M, K, device=dev).to(bf16); offd = off.to(dev)
w1 = (torch.randn(E, K, N, device=dev) * 0.02).to(bf16)
w2 = (torch.randn(E, 1664, K, device=dev) * 0.02).to(bf16)
h = torch.randn(M, 1664, device=dev).to(bf16)
dy = torch.randn(M, N, device=dev).to(bf16)
for name, fn, flops in (("fc1", lambda: ops.grouped_gemm(a, w1, offd), 2 * M * N * K), ("fc2", lambda: ops.grouped_gemm(h, w2, offd), 2 * M * 1664 * K),
                        ("fc1_dgrad", lambda: ops.grouped_gemm(dy, w1, offd, w_is_kn=False), 2 * M * N * K)):
    row = {}
    os.environ["ARIA_GEMM_ORDER"] = "2"; os.environ["ARIA_GEMM_FORCE"] = "2"
    row["v2"] = tf(flops, fn)
    os.environ["ARIA_GEMM_FORCE"] = "3"
    for o in orders:
        os.environ["ARIA_GEMM_ORDER"] = o
        row[f"v3_o{o}"] = tf(flops, fn)
    res["time"]["grouped_" + name] = row
    print(json.dumps({"grouped_" + name: row}), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/gemm_v3.json", "w"), indent=1)
