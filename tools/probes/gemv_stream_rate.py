"""How fast a GEMV-shaped read streams on MI355X as a function of the bytes a CU keeps in flight (tests/probes/probe.hip
probe_stream_kernel): rows of 5120 bytes, R rows per wave in registers (+ RL rows through LDS-DMA), workgroups per CU limited by the
dynamic LDS size.  4 GiB of distinct rows per launch (16x the infinity cache).  The decode GEMVs (csrc/decode.hip) hold 4 rows per wave at
2 waves per SIMD: 160 KB in flight per CU."""
import ctypes
import json
import os

import torch

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cdll = ctypes.CDLL(os.path.join(root, "tests", "probes", "libaria_probe.so"))
cdll.aria_probe_stream.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
ROWS = 4 * (1 << 30) // 5120 // 96 * 96  # a multiple of every rows-per-workgroup used below
W = torch.empty(ROWS * 2560, dtype=torch.int16, device=dev).fill_(3)
out = torch.zeros(1024, dtype=torch.int32, device=dev)
res = {"rows": ROWS, "bytes_per_launch": ROWS * 5120, "runs": []}


def run(R, RL, wg_per_cu):
    lds = max(160 * 1024 // wg_per_cu // 256 * 256, 4 * RL * 5120)
    lds = min(lds, 160 * 1024)
    ts = []
    for rep in range(4):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = cdll.aria_probe_stream(W.data_ptr(), ROWS, R, RL, lds, out.data_ptr(), st)
        e.record()
        torch.cuda.synchronize()
        assert rc == 0, (R, RL, rc)
        ts.append(s.elapsed_time(e))
    t = min(ts[1:])
    res["runs"].append({"rows_in_registers": R, "rows_through_lds": RL, "workgroups_per_cu_by_lds": wg_per_cu, "lds_bytes": lds,
                        "kb_in_flight_per_cu": round(wg_per_cu * 4 * (R + RL) * 5120 / 1024, 1), "ms": round(t, 4),
                        "TB_s": round(ROWS * 5120 / t / 1e9, 3)})


for R, RL, occs in ((2, 0, (2, 4, 6, 8)), (4, 0, (1, 2, 3, 4, 5)), (6, 0, (2, 3, 4)), (8, 0, (1, 2, 3)), (4, 2, (2, 3, 4)), (4, 4, (2,)), (2, 2, (2, 3)),
                    (0, 4, (1, 2)), (8, 4, (1,))):
    for o in occs:
        run(R, RL, o)
print(json.dumps(res))
