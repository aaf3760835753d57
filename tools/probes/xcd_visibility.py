"""Which accesses make one workgroup's stores visible to a workgroup on ANOTHER XCD inside one launch (tests/probes/probe.hip
probe_visibility_kernel: the same 64-word buffer rewritten 2000 times, the reader's L2 holds the previous round's lines), and whether a
region read once is served faster the second time (infinity cache) with the decode GEMVs' non-temporal loads."""
import ctypes
import json
import os

import torch

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cdll = ctypes.CDLL(os.path.join(root, "tests", "probes", "libaria_probe.so"))
cdll.aria_probe_visibility.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
cdll.aria_probe_stream.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
res = {"visibility": {}, "reread": []}
W = ("plain stores + wait", "sc1 stores + wait", "plain stores + buffer_wbl2 sc1 + wait")
R = ("plain loads", "sc1 loads", "buffer_inv sc1 + plain loads", "sc0 sc1 loads")
for partner, pname in ((1, "other XCD"), (8, "same XCD")):
    for wm in range(3):
        for rm in range(4):
            buf = torch.zeros(64, dtype=torch.int32, device=dev)
            f1, f2, out = (torch.zeros(64, dtype=torch.int32, device=dev) for _ in range(3))
            rc = cdll.aria_probe_visibility(buf.data_ptr(), f1.data_ptr(), f2.data_ptr(), out.data_ptr(), 2000, partner, wm, rm, st)
            torch.cuda.synchronize()
            o = out.tolist()
            res["visibility"][f"{pname}: {W[wm]} -> {R[rm]}"] = {"rounds": o[0], "stale_words_of_128000": o[1], "timed_out": bool(o[2]), "rc": rc}
rows_total = 2 * (1 << 30) // 5120 // 96 * 96
Wt = torch.empty(rows_total * 2560, dtype=torch.int16, device=dev).fill_(3)
sink = torch.zeros(1024, dtype=torch.int32, device=dev)


def read(first_row, nrows):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = cdll.aria_probe_stream(Wt.data_ptr() + first_row * 5120, nrows, 4, 0, 80 * 1024, sink.data_ptr(), st)
    e.record()
    torch.cuda.synchronize()
    assert rc == 0
    return s.elapsed_time(e) * 1e3


for mb in (16, 32, 64, 128, 192, 384):
    nrows = mb * (1 << 20) // 5120 // 16 * 16
    read(rows_total // 2, rows_total // 2 // 16 * 16)   # flush: 1 GiB of other rows
    cold = read(0, nrows)
    again = [read(0, nrows) for _ in range(3)]
    res["reread"].append({"MB": mb, "cold_us": round(cold, 1), "again_us": [round(x, 1) for x in again],
                          "cold_TB_s": round(nrows * 5120 / cold / 1e6, 2), "again_TB_s": round(nrows * 5120 / min(again) / 1e6, 2)})
print(json.dumps(res, indent=1))
