"""N > 1 path on CPU: two gloo ranks run the tiny golden LM (kernels through the SIMT emulator) on different batches with
aria_amd.parallel.GradSync (overlapped per-parameter all-reduce + packed small tensors); the synchronised gradients must
equal the average of the two single-process gradients."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "lm.pt")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(g):
    from aria_amd.moe_lm import AriaMoELMConfig, AriaMoELMForCausalLM, load_reference_state_dict

    lm = AriaMoELMForCausalLM(AriaMoELMConfig(**g["cfg"]))
    load_reference_state_dict(lm, g["weights"])
    return lm.train()


def _ids(rank, g):
    gen = torch.Generator().manual_seed(100 + rank)
    return torch.randint(1, g["cfg"]["vocab_size"], (2, 12), generator=gen)


def _worker(rank, world, port, outdir, overlap):
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu import emu_lib

    emu_lib.install()
    from aria_amd.parallel import GradSync

    g = torch.load(GOLDEN, map_location="cpu", weights_only=False)
    lm = _build(g)
    sync = GradSync(lm, overlap=overlap)
    ids = _ids(rank, g)
    out = lm(input_ids=ids, labels=ids, return_logits=False)
    out.loss.backward()
    sync.finish()
    torch.save({n: p.grad.float() for n, p in lm.named_parameters() if p.grad is not None}, os.path.join(outdir, f"g{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_two_rank_gradient_exchange_matches_average(overlap):
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d, overlap), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, f"g{r}.pt")) for r in range(world)]
    from tests.emu import emu_lib

    emu_lib.install()
    try:
        g = torch.load(GOLDEN, map_location="cpu", weights_only=False)
        singles = []
        for r in range(world):
            lm = _build(g)
            ids = _ids(r, g)
            lm(input_ids=ids, labels=ids, return_logits=False).loss.backward()
            singles.append({n: p.grad.float() for n, p in lm.named_parameters() if p.grad is not None})
    finally:
        emu_lib.uninstall()
    assert set(got[0]) == set(singles[0])
    for n in singles[0]:
        want = sum(s_[n] for s_ in singles) / world
        for r in range(world):
            err = (got[r][n] - want).abs().max()
            assert err <= 1e-2 * want.abs().max().clamp(min=1e-6) + 1e-6, (n, r, float(err))
        assert all(torch.equal(got[0][n], got[r_][n]) for r_ in range(1, world)), n  # replicas hold identical gradients after the exchange
