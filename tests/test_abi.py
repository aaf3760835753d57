"""Host-side checks that need no GPU: the gfx950 library exists, loads, and exports every symbol include/aria_hip.h declares;
the ctypes table matches the header; the product path refuses to run without the library / on CPU tensors."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "aria_hip.h")).read()
    src = src.split("#ifdef __cplusplus\n}")[0]  # C declarations only
    return sorted(set(re.findall(r"^(?:int|int64_t|void\*|void) (aria_\w+)\(", src, flags=re.M)))


def test_header_and_ctypes_table_agree():
    from aria_amd import hip

    assert header_symbols() == sorted(hip.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from aria_amd import hip

    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = hip.HipLibrary(hip.LIB_PATH)
    assert lib.missing == []
    assert lib.cdll.aria_abi_version() == hip.ABI_VERSION == 3
    assert "#define ARIA_ABI_VERSION 3" in open(os.path.join(ROOT, "include", "aria_hip.h")).read()
    # the library's own `aria_*` exports are exactly the header's entry points (internal helpers such as aria_check_launch are hidden)
    import subprocess

    nm = subprocess.run(["nm", "-D", "--defined-only", hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1] for line in nm.splitlines() if re.search(r" [TW] _?(Z\d+)?aria_", line)})
    assert exported == header_symbols(), set(exported) ^ set(header_symbols())


def test_missing_library_fails_loudly(tmp_path):
    from aria_amd import hip

    with pytest.raises(hip.AriaHipError):
        hip.HipLibrary(str(tmp_path / "libaria_hip.so"))


def test_cpu_tensors_are_rejected_by_the_product_path():
    from aria_amd import hip, ops

    assert not hip.is_emulated()
    x = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(hip.AriaHipError):
        ops.gemm(x, x)


def test_nothing_in_the_package_imports_the_oracle_or_tests():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "aria_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", src, flags=re.M), f
