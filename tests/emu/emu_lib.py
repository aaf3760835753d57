"""TEST INFRASTRUCTURE: build + load the emulated C-ABI library (same kernel sources compiled for the host
against tests/emu/hip_emu.h) and point aria_amd.hip at it so the Python host code can be exercised with CPU
tensors.  Never used by product code; the product loader only ever opens aria_amd/libaria_hip.so."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EMU_SO = os.path.join(ROOT, "tests", "emu", "libaria_emu.so")


def build():
    subprocess.run(["make", "-s", "emu"], cwd=ROOT, check=True)
    return EMU_SO


def install():
    from aria_amd import hip

    build()
    lib = hip.HipLibrary(EMU_SO)
    hip._LIB = lib
    hip._EMULATED = True
    return lib


def uninstall():
    from aria_amd import hip

    hip._LIB = None
    hip._EMULATED = False
