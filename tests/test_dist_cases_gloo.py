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
