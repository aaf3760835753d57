"""global_atomic_add_f32 on MI355X: rate and scope semantics (what the single-pass attention backward's dQ accumulation relies on).
  * placement: HW_REG_XCC_ID of workgroup b vs b % 8
  * correctness: region per XCD (b & 7), NO sc bits -> every float must end at (#workgroups per region) x iters
  * rates (GB/s of fp32 payload): XCD-local regions / private regions / one region for everybody, without and with sc1
Writes gpurun_out/l2_atomics.json."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes

# test infrastructure, not the product library: tests/probes/libaria_probe.so (make probes)
_cdll = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "probes", "libaria_probe.so"))
_cdll.aria_probe_atomic.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]


class _Lib:
    @staticmethod
    def call(name, *a):
        rc = getattr(_cdll, name)(*a)
        assert rc == 0, (name, rc)


lib = _Lib()
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
out = {}
NB = 2048


def run(region_floats, region_mode, scope, iters, row_stride=128, nb=NB):
    nreg = {0: 8, 1: nb, 2: 1}[region_mode]
    buf = torch.zeros(nreg * region_floats, dtype=torch.float32, device=dev)
    xcc = torch.full((nb,), -1, dtype=torch.int32, device=dev)
    lib.call("aria_probe_atomic", buf.data_ptr(), xcc.data_ptr(), nb, region_floats, region_mode, scope, 1, row_stride, st)  # warm
    torch.cuda.synchronize()
    buf.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    lib.call("aria_probe_atomic", buf.data_ptr(), xcc.data_ptr(), nb, region_floats, region_mode, scope, iters, row_stride, st)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    used = region_floats // row_stride * 128
    payload = nb * iters * used * 4
    v = buf.view(nreg, region_floats // row_stride, row_stride)[:, :, :128]
    want = nb // nreg * iters
    return {"ms": round(ms, 4), "GB_s": round(payload / ms / 1e6, 1), "expected_per_float": want, "min": float(v.min()), "max": float(v.max()),
            "exact": bool((v == want).all())}, xcc.cpu()


r, xcc = run(64 * 128, 0, 0, 4)
b = torch.arange(NB)
out["xcc_id_equals_block_mod_8"] = bool((xcc == (b % 8).int()).all())
out["xcc_id_histogram"] = torch.bincount(xcc.clamp(min=0).long(), minlength=8).tolist()
out["xcc_of_first_16_blocks"] = xcc[:16].tolist()
for name, rf, rm in (("xcd_region_32KB", 64 * 128, 0), ("xcd_region_1MB", 2048 * 128, 0), ("private_region_32KB", 64 * 128, 1), ("one_region_32KB", 64 * 128, 2),
                     ("one_region_1MB", 2048 * 128, 2)):
    for scope in (0, 1):
        iters = 16 if rf <= 64 * 128 else 1
        out[f"{name}.scope{scope}"] = run(rf, rm, scope, iters)[0]
# the attention backward's shape: row stride 2560 floats (dQ fp32 [S, H*hd], one head's 128 columns), XCD-local, 64-row tiles
out["xcd_region_64rows_stride2560.scope0"] = run(64 * 2560, 0, 0, 16, row_stride=2560)[0]
out["xcd_region_2048rows_stride2560.scope0"] = run(2048 * 2560, 0, 0, 1, row_stride=2560)[0]
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/l2_atomics.json", "w"), indent=1)
print(json.dumps(out, indent=1))
