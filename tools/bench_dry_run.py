"""Exercise bench.py's OWN code path (argument handling, timing hooks, roofline / JSON assembly) on CPU through the emulator, without
touching bench.py: this harness hands bench.py a torch proxy whose "cuda" is the CPU (fake events, no-op synchronisation) and shrinks the
model dimensions.  Prints bench.py's JSON line; the numbers mean nothing -- it only has to run and carry every contract field.

    python tools/bench_dry_run.py            # N = 1
    python tools/bench_dry_run.py --world 2  # the N > 1 branch: two gloo ranks (init_process_group is redirected from "nccl" to "gloo")
    python tools/bench_dry_run.py --world 2 --ep
"""
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from tests.emu import emu_lib  # noqa: E402

emu_lib.install()


class FakeEvent:
    def __init__(self, enable_timing=False):
        self.t = None

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class TorchProxy(types.ModuleType):
    """torch as bench.py sees it: device("cuda", i) and Generator(device="cuda") land on the CPU, torch.cuda is a stub"""

    def __init__(self):
        super().__init__("torch")
        self.cuda = types.SimpleNamespace(is_available=lambda: True, set_device=lambda i: None, synchronize=lambda *a: None, Event=FakeEvent,
                                          get_device_name=lambda *a: "cpu (dry run)", empty_cache=lambda: None,
                                          max_memory_allocated=lambda *a: 0, reset_peak_memory_stats=lambda *a: None)

    def __getattr__(self, name):
        return getattr(torch, name)

    def device(self, *a, **k):
        return torch.device("cpu")

    def Generator(self, device=None):
        return torch.Generator(device="cpu")


def run(world: int, rank: int, ep: bool):
    import aria_amd.moe_lm as moe_lm
    import aria_amd.vision as vision
    import bench

    bench.torch = TorchProxy()
    if world > 1:
        import torch.distributed as dist

        real_init = dist.init_process_group
        dist.init_process_group = lambda backend=None, **kw: real_init("gloo", rank=rank, world_size=world)
    RealLM, RealVis = moe_lm.AriaMoELMConfig, vision.AriaVisionConfig

    def tiny_lm(**kw):
        kw.update(hidden_size=64, num_attention_heads=1, vocab_size=512, moe_intermediate_size=16, moe_num_experts=8, moe_topk=2)
        return RealLM(**kw)

    def tiny_vis(**kw):
        return RealVis(hidden_size=64, num_attention_heads=1, intermediate_size=64, image_size=56, **kw)

    moe_lm.AriaMoELMConfig, vision.AriaVisionConfig = tiny_lm, tiny_vis
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1", "--layers", "2", "--vit-layers", "1", "--images", "0",
                "--batch", "2", "--seq", "64", "--no-cpu-baseline", "--long64k-seq", "96", "--long64k-images", "0", "--gen-new", "6", "--gen-image", "0", "--prefill-seq", "96", "--prefill-frames", "0", "--sub-record-repeats", "1"] + (["--ep"] if ep else []) + (["--time-grouped"] if world == 1 else []) + os.environ.get("ARIA_DRY_EXTRA", "").split()
    try:
        bench.main()
    finally:
        moe_lm.AriaMoELMConfig, vision.AriaVisionConfig = RealLM, RealVis


def _worker(rank, world, port, ep):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    run(world, rank, ep)


if __name__ == "__main__":
    world = int(sys.argv[sys.argv.index("--world") + 1]) if "--world" in sys.argv else 1
    ep = "--ep" in sys.argv
    if world == 1:
        run(1, 0, False)
    else:
        import socket

        import torch.multiprocessing as mp

        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        mp.spawn(_worker, args=(world, port, ep), nprocs=world, join=True)
