"""What the cross-workgroup synchronisation of a one-launch decode schedule costs on MI355X (tests/probes/probe.hip probe_sync_kernel /
probe_pingpong_kernel): 78 176 workgroups of 256 threads (the streamed decode kernel's grid at Aria's shape) doing ONLY the selected pieces --
ticket atomics, completion-counter atomics, cache write-back / invalidate, polls, barriers -- and the one-way latency of a word bounced
between two workgroups on different / the same XCD.  Prints one JSON object."""
import ctypes
import json
import os

import torch

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cdll = ctypes.CDLL(os.path.join(root, "tests", "probes", "libaria_probe.so"))
cdll.aria_probe_sync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
cdll.aria_probe_pingpong.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
cnt = torch.zeros(4096, dtype=torch.int32, device=dev)
sink = torch.zeros(1 << 21, dtype=torch.int16, device=dev)
NB = 78176
MODES = [("nothing (launch + 78 176 workgroups ending)", 0),
         ("two bare barriers", 32),
         ("ticket: returning atomic, ONE word", 1),
         ("ticket: returning atomic, one word per XCD", 1 | 64),
         ("completion: returning atomic, a new word every 512 workgroups", 2),
         ("completion: non-returning atomic", 2 | 128),
         ("ticket + completion", 1 | 2),
         ("one sc1 poll per workgroup", 16),
         ("buffer_inv sc1 per wave", 8),
         ("buffer_wbl2 sc1 per wave", 4),
         ("one 2-byte sc1 store per wave + wait", 256),
         ("sc1 store + wait, poll, barriers, ticket, completion (the proposed protocol)", 256 | 16 | 32 | 1 | 2),
         ("inv + wbl2 + poll + barriers + ticket + completion (the round-4 first build)", 8 | 4 | 16 | 32 | 1 | 2)]
out = {"workgroups": NB, "sync_pieces": {}, "pingpong": {}}
for name, mode in MODES:
    times = []
    for rep in range(4):
        cnt.zero_()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = cdll.aria_probe_sync(cnt.data_ptr(), sink.data_ptr(), NB, mode, st)
        e.record()
        torch.cuda.synchronize()
        assert rc == 0
        times.append(s.elapsed_time(e) * 1e3)
    best = min(times[1:])
    out["sync_pieces"][name] = {"mode": mode, "us_per_launch": round(best, 1), "ns_per_workgroup": round(best * 1e3 / NB, 2)}
flag = torch.zeros(64, dtype=torch.int32, device=dev)
res = torch.zeros(2, dtype=torch.int32, device=dev)
ITERS = 2000
for how, hname in ((0, "sc1 store / sc1 load"), (1, "atomic exchange / sc1 load"), (2, "plain store + buffer_wbl2 sc1 / buffer_inv sc1 + plain load")):
    for partner, pname in ((1, "workgroups 0 and 1 (different XCDs)"), (8, "workgroups 0 and 8 (same XCD)")):
        flag.zero_()
        res.zero_()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = cdll.aria_probe_pingpong(flag.data_ptr(), res.data_ptr(), ITERS, partner, how, st)
        e.record()
        torch.cuda.synchronize()
        assert rc == 0
        r = res.tolist()
        out["pingpong"][f"{hname}; {pname}"] = {"round_trips": r[0], "lost": bool(r[1]),
                                               "one_way_ns": round(s.elapsed_time(e) * 1e6 / max(1, r[0]) / 2, 1)}
print(json.dumps(out, indent=1))
