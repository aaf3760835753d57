"""Re-verify on hardware the lane layouts the emulator assumes (-m gpu)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def probe_lib():
    """tests/probes/libaria_probe.so (make probes; built by __graft_entry__.build()): test infrastructure, not part of the product ABI."""
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libaria_probe.so")
    lib = ctypes.CDLL(path)
    lib.aria_probe_tr16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return lib


def test_ds_read_tr16_b64_semantics_dump():
    lib = probe_lib()
    for mode in (0, 1):
        out = torch.zeros(256, dtype=torch.int16, device="cuda")
        assert lib.aria_probe_tr16(out.data_ptr(), mode, torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        v = out.cpu().view(64, 4).tolist()
        print(f"tr16 mode {mode}:", v[:20], "...", v[32:36])
        # believed semantics: within each 16-lane group, out[l][j] = in64[lane 4*j + (l%16)//4].b16[l%4]
        for l in range(64):
            g, q = l // 16, l % 16
            for j in range(4):
                src_lane = g * 16 + 4 * j + q // 4
                off = src_lane * 4 if mode == 0 else ((src_lane & 15) * 64 + (src_lane >> 4) * 4)
                assert v[l][j] == off + (q % 4), (mode, l, j, v[l][j], off + q % 4)
