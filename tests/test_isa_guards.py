"""Guards on the COMPILED code of the hot kernels (CPU only: hipcc cross-compiles gfx950).

Three silent performance bugs of round 2 were wait instructions the compiler placed, none of them visible in the source
(profiles/r02_gemm_tile_timeline.md):
  * `s_waitcnt vmcnt(0)` in front of every transposing LDS read that follows an LDS-DMA in flight (GEMM K loop drained twice per K-tile),
  * `__shfl_xor(v, 1)` = `ds_bpermute_b32` + `lgkmcnt(0)`, 64 serialized LDS round trips per wave in the GEMM epilogues,
  * the wait for a fragment loaded in front of a loop landing at its first use INSIDE the loop (attention: the next tile's prefetch
    drained in front of the first MFMA of every tile).
These tests look at the ISA so that a compiler or source change that brings one of them back fails here and not in a benchmark."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")

_ISA = {}


def isa(src):
    if src not in _ISA:
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Iinclude", "-Iaria_amd/csrc", "-S", "--cuda-device-only",
                            os.path.join("aria_amd", "csrc", src), "-o", "-"], cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        _ISA[src] = r.stdout.split("\n")
    return _ISA[src]


def kernel_body(lines, mangled_fragment):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(mangled_fragment) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def loops(body):
    """[(header label, [lines])] for every innermost loop LLVM annotated."""
    out = []
    for i, l in enumerate(body):
        m = re.match(r"^\.(LBB\d+_\d+):.*Loop Header", l)
        if not m:
            continue
        tag = "Header=" + m.group(1)[1:]
        j = i + 1
        while j < len(body):
            if re.match(r"^\.LBB\d+_\d+:", body[j]) and tag not in body[j]:
                break
            j += 1
        ls = body[i:j]
        back = [k for k, x in enumerate(ls) if re.search(r"s_c?branch\w*\s+\." + m.group(1) + r"\b", x)]
        if back:  # cut behind the last back edge (the annotation of the exit block is not reliable)
            ls = ls[:back[-1] + 1]
        out.append((m.group(1), ls))
    return out


def code(lines):
    """instruction lines with a flag: inside an inline-assembly block or not"""
    in_asm, res = False, []
    for l in lines:
        s = l.strip()
        if "#ASMSTART" in s:
            in_asm = True
        elif "#ASMEND" in s:
            in_asm = False
        elif s and not s.startswith(";") and not s.startswith("."):
            res.append((s, in_asm))
    return res


GEMM3 = ["12gemm3_kernelILb0ELb0ELi3", "12gemm3_kernelILb0ELb1ELi3", "12gemm3_kernelILb1ELb1ELi3"]
GEMM3_DGLU = ["12gemm3_kernelILb0ELb0ELi5", "12gemm3_kernelILb0ELb1ELi5"]  # same K loop, SwiGLU-backward epilogue (round 3)
# r06: the gathered launches too -- the weight gradient with gathered reduction rows had NO straight-line steady loop (its per-K-tile index
# requests sat behind run-time tests) and its scalar index loads were waited for in full by the next phase's `s_waitcnt lgkmcnt(0)`
GEMM3_GATHER = ["12gemm3_kernelILb0ELb1ELi8", "12gemm3_kernelILb1ELb1ELi11"]


@pytest.mark.parametrize("kernel", GEMM3 + GEMM3_DGLU + GEMM3_GATHER)
def test_gemm3_steady_loop_has_no_compiler_waits_and_no_branches(kernel):
    body = kernel_body(isa("gemm3.hip"), kernel)
    steady = None
    for _, ls in loops(body):
        c = code(ls)
        n_mfma = sum(1 for s, _ in c if s.startswith("v_mfma"))
        n_br = sum(1 for s, _ in c if s.startswith("s_cbranch") or s.startswith("s_branch"))
        if n_mfma == 64 and n_br == 1:  # two K-tiles, only the back edge
            steady = c
    assert steady is not None, "no straight-line 64-MFMA loop: the steady-state K loop is gone"
    own = [s for s, a in steady if s.startswith("s_waitcnt") and not a]
    assert own == [], f"compiler-inserted waits in the steady K loop: {own}"
    assert sum(1 for s, _ in steady if s.startswith("global_load_lds_dwordx4")) == 16  # 4 half-tiles x 2 pieces per wave and K-tile
    assert sum(1 for s, _ in steady if s.startswith("s_barrier")) == 16
    assert not any(s.startswith("scratch_") for s, _ in steady), "register spills inside the steady K loop"
    # no scalar loads inside the loop (lgkmcnt is the fragment reads' counter: the next `lgkmcnt(0)` would wait for the load in full); the gathered
    # weight gradient's indices arrive as one 4-byte-per-lane LDS-DMA piece per K-tile and leave their slot through one ds_read2_b32
    assert not any(s.startswith("s_load") for s, _ in steady), "scalar loads inside the steady K loop"
    n_idx = sum(1 for s, _ in steady if s.startswith("global_load_lds_dword "))
    assert n_idx == (2 if kernel.endswith("Li11") else 0), n_idx


@pytest.mark.parametrize("kernel", GEMM3)
def test_gemm3_epilogues_park_without_a_lane_exchange(kernel):
    """r02: `__shfl_xor(v, 1)` = ds_bpermute_b32 + lgkmcnt(0), 64 serialized LDS round trips per wave; r02-r05: a DPP move + three selects
    per pair of values (320 vector instructions per wave and tile).  r06: the accumulators are held transposed (a lane owns four consecutive
    columns of a row), so the epilogues exchange NOTHING between lanes and park 8 bytes per LDS store."""
    c = code(kernel_body(isa("gemm3.hip"), kernel))
    n_perm = sum(1 for s, _ in c if s.startswith("ds_bpermute_b32"))
    n_dpp = sum(1 for s, _ in c if "quad_perm:[1,0,3,2]" in s)
    assert n_perm <= 24, f"{n_perm} ds_bpermute_b32 (only the grouped tile lookup's wave collectives may use it)"
    assert n_dpp == 0, "a pair exchange is back in an epilogue"
    assert sum(1 for s, _ in c if s.startswith("ds_write_b64") or s.startswith("ds_write2_b64")) >= 32


@pytest.mark.parametrize("kernel", GEMM3_DGLU)
def test_gemm3_dglu_epilogue_keeps_its_loads_in_flight_and_out_of_scratch(kernel):
    """The SwiGLU-backward epilogue reads gate / up of the forward from HBM with the MFMAs finished and nothing else to hide the latency:
    the 16 loads of a 128-row half must be issued back to back (no wait between them), and the epilogue must not spill (a first form that
    fetched all 32 pieces up front kept them live next to the accumulators: 49 spilled VGPRs)."""
    body = kernel_body(isa("gemm3.hip"), kernel)
    c = [s for s, _ in code(body)]
    last_mfma = max(i for i, s in enumerate(c) if s.startswith("v_mfma"))
    tail = c[last_mfma:]
    assert sum(1 for s in tail if s.startswith("scratch_store")) == 0, "the epilogue spills"
    runs, cur = [], 0
    for s in tail:
        if s.startswith("global_load_dwordx4"):
            cur += 1
        elif s.startswith("s_waitcnt") and "vmcnt" in s:
            if cur:
                runs.append(cur)
            cur = 0
    assert runs and max(runs) >= 16, runs


ATTN = ["16attn_fwd2_kernelILi72ELi12", "16attn_fwd2_kernelILi128ELi8", "21attn_bwd3_dkdv_kernelILi128", "19attn_bwd5_dq_kernelILi128"]


@pytest.mark.parametrize("kernel", ATTN)
def test_attention_tile_loop_does_not_drain_its_prefetch_before_the_first_mfma(kernel):
    body = kernel_body(isa("attn.hip"), kernel)
    main = max((ls for _, ls in loops(body)), key=lambda ls: sum(1 for s, _ in code(ls) if s.startswith("v_mfma")))
    c = [s for s, _ in code(main)]
    first_mfma = next(i for i, s in enumerate(c) if s.startswith("v_mfma"))
    loads = [i for i, s in enumerate(c[:first_mfma]) if s.startswith("global_load")]
    assert loads, "the next tile's loads are expected in front of the first MFMA"
    drains = [s for s in c[loads[0]:first_mfma] if s.startswith("s_waitcnt") and "vmcnt(0)" in s]
    assert drains == [], f"the tile loop waits for its own prefetch before its first MFMA: {drains}"


@pytest.mark.parametrize("kernel", ["21attn_bwd3_dkdv_kernelILi128", "19attn_bwd5_dq_kernelILi128"])
def test_attention_backward_tiles_arrive_by_lds_dma_without_compiler_drains(kernel):
    """Round 3: the hd-128 backward kernels stage the next tile with global_load_lds_dwordx4 issued as their OWN instruction (inline
    assembly): hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS read that follows a __builtin_amdgcn_global_load_lds, i.e. it would
    drain the prefetch at its first consumer.  In the tile loop: 4 DMA pieces per wave (2 tensors x 2 pieces), every one inside an asm block,
    no register-staged ds_write_b128 pass left, and no compiler-inserted vmcnt(0) between the pieces and the loop's last MFMA."""
    body = kernel_body(isa("attn.hip"), kernel)
    main = max((ls for _, ls in loops(body)), key=lambda ls: sum(1 for s, _ in code(ls) if s.startswith("v_mfma")))
    c = code(main)
    dma = [(i, a) for i, (s, a) in enumerate(c) if s.startswith("global_load_lds_dwordx4")]
    assert len(dma) == 4 and all(a for _, a in dma), dma
    n_w128 = sum(1 for s, _ in c if s.startswith("ds_write_b128"))  # dK/dV: role A publishes P with 8 of them; dQ v5 has no exchange at all
    assert n_w128 == (8 if "dkdv" in kernel else 0), f"{n_w128} ds_write_b128 in the loop: a register-staged tile store is back"
    last_mfma = max(i for i, (s, _) in enumerate(c) if s.startswith("v_mfma"))
    own = [s for s, a in c[dma[0][0]:last_mfma] if s.startswith("s_waitcnt") and "vmcnt(0)" in s and not a]
    assert own == [], f"compiler-inserted drains between the DMA issue and the last MFMA of the tile: {own}"


def test_router_reduction_loop_keeps_its_dma_pieces_in_flight():
    """Round 5: router_fused_kernel stages its operands by LDS-DMA, four stages deep: per 64-index chunk a wave issues 6 pieces (2 of the x tile + 4 of
    its weight tile) as its own instructions, waits with a COUNTED vmcnt (two newer chunks stay in flight), reads 8 fragments by ds_read_b128 and
    issues 4 MFMAs -- and the compiler must not put a drain (vmcnt(0)) into the loop."""
    body = kernel_body(isa("moe.hip"), "19router_fused_kernelILi2E")
    main = max((ls for _, ls in loops(body)), key=lambda ls: sum(1 for s, _ in code(ls) if s.startswith("v_mfma")))
    c = [s for s, _ in code(main)]
    assert sum(1 for s in c if s.startswith("v_mfma")) == 4
    assert sum(1 for s in c if s.startswith("global_load_lds_dwordx4")) == 6
    assert sum(1 for s in c if s.startswith("ds_read_b128")) == 8
    waits = [s for s in c if s.startswith("s_waitcnt") and "vmcnt" in s]
    assert waits and all("vmcnt(12)" in s for s in waits), waits
    assert not [s for s in c if s.startswith("global_load_dwordx4")], "an operand fragment is read straight from global memory again"
