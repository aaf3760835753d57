import sys
import torch
sys.path.insert(0, ".")
from aria_amd import ops, hip
bf16 = torch.bfloat16
dev = "cuda"
M, K, N = 128, 64, 128
a = torch.randn(M, K).to(bf16).to(dev); dy = torch.randn(M, N).to(bf16).to(dev)
ref = a.float().t() @ dy.float()
print("dense oc/oc f32", flush=True)
o = ops.gemm(a, dy, a_oc=True, b_oc=True, out_dtype=torch.float32); torch.cuda.synchronize(); print(" ok", (o - ref).abs().max().item(), flush=True)
off = torch.tensor([0, 128], dtype=torch.int32, device=dev)
print("off ptr", hex(off.data_ptr()), "a", hex(a.data_ptr()), flush=True)
lib = hip.get_lib()
out = torch.zeros(1, K, N, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
print("stream", st, flush=True)
for cf32, acc in ((0, 0), (1, 0)):
    o = torch.zeros(1, K, N, dtype=torch.float32 if cf32 else bf16, device=dev)
    print("grouped wgrad c_f32", cf32, flush=True)
    lib.call("aria_grouped_gemm_wgrad_bf16", a.data_ptr(), dy.data_ptr(), o.data_ptr(), off.data_ptr(), 1, K, N, K, N, cf32, acc, st)
    torch.cuda.synchronize(); print(" ok", (o[0].float() - ref).abs().max().item(), flush=True)
