"""tests/dist_cases.py over gloo (two CPU ranks, kernels through the SIMT emulator): the same worker code that tests/test_gpu_rccl_multi.py
runs over RCCL as soon as two GPUs are visible -- so that what runs there first has already run here."""
import pytest

from tests import dist_cases as D
from tests.emu import emu_lib


@pytest.fixture(autouse=True)
def _emu():
    yield
    emu_lib.uninstall()


@pytest.mark.parametrize("mode,overlap", [("reduce_scatter", True), ("all_reduce", False)])
def test_dp_lm_exchange_norm_and_sharded_adamw(mode, overlap):
    got = D.run_dp_lm("gloo", 2, mode, overlap)
    assert got[0]["bytes"] > 0 and got[0]["bytes"] == got[1]["bytes"]


def test_ep_layer_two_ranks():
    D.run_ep_layer("gloo", 2)


def test_ep_layer_chunked_exchange(monkeypatch):
    """The exchange cut into two token chunks (chunk i + 1's dispatch under chunk i's GEMMs on hardware; here: the same code, one stream):
    outputs, input gradients and every weight gradient equal the local layer's -- the segment launches run once per chunk."""
    monkeypatch.setenv("ARIA_EP_CHUNKS", "2")
    monkeypatch.setenv("ARIA_EP_CHUNK_MIN", "8")
    D.run_ep_layer("gloo", 2, width=(128, 128, 8, 2))


def test_ep_layer_consumes_the_exchange_in_arrival_order():
    """Width the segment launches take (D 128, I 128, 8 experts top-2): fc1 + glu and fc2 run over (source rank, local expert) segments with
    weight index = segment mod local experts -- no row re-order around the grouped GEMMs; weight gradients summed over the source ranks."""
    D.run_ep_layer("gloo", 2, width=(128, 128, 8, 2))
