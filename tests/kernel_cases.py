"""Kernel parity cases shared by the emulator tests (CPU, tests/test_emu_kernels.py) and the hardware tests
(MI355X, tests/test_gpu_kernels.py, -m gpu).  Every case feeds the SAME seeded inputs to the C-ABI kernels
(through aria_amd.ops) and to the CPU oracle (oracle/aria_oracle.py) and compares:
  * integer / index work (router indices, histogram, sort order, permuted rows): bit-exact;
  * bf16 elementwise work that mirrors the reference's rounding points: bit-exact under the emulator,
    <= 1 bf16 ulp on hardware (device expf / rsqrt may differ from the host's libm by one fp32 ulp);
  * GEMM / attention / reductions: tolerance stated per case (bf16 inputs, fp32 accumulation).
"""
import pytest
import torch

from oracle import aria_oracle as O

bf16 = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(bf16)


def close(got, want, rtol=2e-2, atol=2e-2):
    got, want = got.detach().cpu().float(), want.detach().cpu().float()
    assert got.shape == want.shape, (got.shape, want.shape)
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    worst = (err - tol).flatten().argmax()  # the element that exceeds its own tolerance the most (not simply the largest error)
    assert bool((err <= tol).all()), f"err {err.flatten()[worst].item():.4g} > tol {tol.flatten()[worst].item():.3g} (max err {err.max().item():.4g})"


def same(got, want, exact):
    """bit-exact under the emulator; within two bf16 ulps on hardware (device rsqrt/exp differ from the host libm
    by an fp32 ulp, which can flip one bf16 rounding; a second rounding point can double it)."""
    got, want = got.cpu(), want.cpu()
    if exact:
        assert torch.equal(got, want)
    else:
        err = (got.float() - want.float()).abs()
        assert bool((err <= 2 ** -6 * want.float().abs() + 1e-30).all()), err.max().item()


# ------------------------------------------------------------------------------------------ GEMM
def case_gemm_layouts(dev, M, N, K, a_oc, b_oc):
    from aria_amd import ops

    if a_oc and M % 8:
        M = (M + 7) // 8 * 8
    A = rnd(M, K, seed=1)
    B = rnd(K, N, seed=2)  # logical [K,N]
    a_arg = (A.t().contiguous() if a_oc else A).to(dev)
    b_arg = (B if b_oc else B.t().contiguous()).to(dev)
    bias = rnd(N, seed=3)
    want = A.float() @ B.float()
    got = ops.gemm(a_arg, b_arg, a_oc=a_oc, b_oc=b_oc)
    close(got, want.to(bf16), 1e-2, 1e-2 * K ** 0.5)
    got32 = ops.gemm(a_arg, b_arg, a_oc=a_oc, b_oc=b_oc, out_dtype=torch.float32)
    close(got32, want, 1e-4, 1e-3)
    gotb = ops.gemm(a_arg, b_arg, a_oc=a_oc, b_oc=b_oc, bias=bias.to(dev), out_dtype=torch.float32)
    close(gotb, want + bias.float(), 1e-4, 1e-3)
    acc = torch.ones(M, N, dtype=torch.float32, device=dev)
    ops.gemm(a_arg, b_arg, a_oc=a_oc, b_oc=b_oc, out=acc, accumulate=True)
    close(acc, want + 1.0, 1e-4, 1e-3)
    accb = torch.ones(M, N, dtype=bf16, device=dev)
    ops.gemm(a_arg, b_arg, a_oc=a_oc, b_oc=b_oc, out=accb, accumulate=True)
    close(accb, (want + 1.0).to(bf16), 1e-2, 1e-2 * K ** 0.5)


def case_gemm_accumulate_exact(dev, M, N, K, b_oc=False):
    """C += A B + bias on a bf16 C: whichever epilogue carries it (r05b: the complete-row form with the old tile staged through the LDS, also for
    column tiles that hang over N -- the frozen ViT's residual adds, vision.py forward_frozen; the 4-byte-per-lane form elsewhere), the result
    is bf16(fp32(A B + bias) + C_old) with ONE rounding: exactly what the fp32-output launch of the same GEMM plus an fp32 add gives."""
    from aria_amd import ops

    a = rnd(M, K, seed=M + 1).to(dev)
    w = rnd(N, K, seed=N + 2, scale=0.3)
    w_arg = (w.t().contiguous() if b_oc else w).to(dev)
    bias = rnd(N, seed=3).to(dev)
    old = rnd(M, N, seed=4).to(dev)
    for bv in (None, bias):
        c32 = ops.gemm(a, w_arg, b_oc=b_oc, bias=bv, out_dtype=torch.float32)
        want = (c32 + old.float()).to(bf16)
        got = old.clone()
        ops.gemm(a, w_arg, b_oc=b_oc, bias=bv, out=got, accumulate=True)
        assert torch.equal(got.cpu(), want.cpu()), float((got.float() - want.float()).abs().max())
        plain = ops.gemm(a, w_arg, b_oc=b_oc, bias=bv)                 # (and the plain store of the same tiles, partial column tile included)
        assert torch.equal(plain.cpu(), c32.to(bf16).cpu())
    big = torch.zeros(M + 2, N + 16, dtype=bf16, device=dev)           # a strided C (row pitch > N): neighbours untouched
    view = big[1:M + 1, 8:N + 8]
    view.copy_(old)
    ops.gemm(a, w_arg, b_oc=b_oc, bias=bias, out=view, accumulate=True)
    assert torch.equal(view.cpu(), want.cpu()) and float(big[0].abs().max()) == 0 and float(big[:, :8].abs().max()) == 0 and float(big[:, N + 8:].abs().max()) == 0


def case_gemm_split_k_slabs(dev, M, N, K, a_oc, b_oc):
    """The remainder split-K of the 256 x 256 kernels (aria_gemm_bf16_ws): partial sums leave as fp32 slabs in the ACCUMULATORS' order (r05b: 16
    bytes per lane, 1 KiB per store instruction) and gemm3_reduce_kernel, their only reader, maps them back -- every element of C (ragged
    row / column tiles, bias, accumulate, fp32 output) against fp32 torch."""
    from aria_amd import hip, ops

    assert hip.get_lib().cdll.aria_gemm_workspace_bytes(M, N, K, int(a_oc), int(b_oc)) > 0, "the shape is expected to split"
    g = torch.Generator().manual_seed(M + N)
    A = (torch.randn(M, K, generator=g) * 0.5).to(bf16)
    B = (torch.randn(N, K, generator=g) * 0.5).to(bf16)
    a = (A.t().contiguous() if a_oc else A).to(dev)
    b = (B.t().contiguous() if b_oc else B).to(dev)
    bias, old = torch.randn(N, generator=g).to(bf16), torch.randn(M, N, generator=g).to(bf16)
    want = A.float() @ B.float().t()
    g32 = ops.gemm(a, b, a_oc=a_oc, b_oc=b_oc, out_dtype=torch.float32)
    close(g32, want, 1e-4, 1e-3)
    gb = ops.gemm(a, b, a_oc=a_oc, b_oc=b_oc, bias=bias.to(dev))
    close(gb, (want + bias.float()).to(bf16), 1e-2, 1e-2 * K ** 0.5)
    acc = old.clone().to(dev)
    ops.gemm(a, b, a_oc=a_oc, b_oc=b_oc, bias=bias.to(dev), out=acc, accumulate=True)
    assert torch.equal(acc.cpu(), (ops.gemm(a, b, a_oc=a_oc, b_oc=b_oc, bias=bias.to(dev), out_dtype=torch.float32).cpu() + old.float()).to(bf16))


def case_unpermute_with_residual(dev, T, D, k, E=8):
    """r05b: the decoder layer's residual add as the last step of the un-permute launch (aria_moe_unpermute_res) == aria_moe_unpermute +
    aria_add_bf16, bit for bit (bf16(h + bf16(combine + shared))), with and without scores / shared rows; widths without a compile-time kernel
    take the two launches inside ``ops.moe_unpermute``."""
    from aria_amd import ops

    g = torch.Generator().manual_seed(T + D)
    idx = torch.stack([torch.randperm(E, generator=g)[:k] for _ in range(T)]).to(torch.int32).to(dev)
    counts = torch.bincount(idx.flatten().cpu().long(), minlength=E).to(torch.int32).to(dev)
    _, _, inv = ops.moe_sort(idx, counts)
    eo = rnd(T * k, D, seed=1).to(dev)
    scores = torch.rand(T, k, generator=g).to(bf16).to(dev)
    sh, h = rnd(T, D, seed=2).to(dev), rnd(T, D, seed=3).to(dev)
    for sc, ad in ((scores, sh), (scores, None), (None, None)):
        want = ops.add(h, ops.moe_unpermute(eo, inv, sc, k, add=ad))
        got = ops.moe_unpermute(eo, inv, sc, k, add=ad, residual=h)
        assert torch.equal(got.cpu(), want.cpu()), float((got.float() - want.float()).abs().max())


def case_layernorm_two_rows_in_flight(dev, T, D):
    """r05b: ``layernorm_fwd2_kernel`` (two rows in flight per wave, compile-time row width; the frozen ViT's LayerNorms) gives the bits of the
    one-row-at-a-time kernel (ARIA_LAYERNORM_V1=1): y, mean, rstd -- odd row counts and rows past the grid's last pair included."""
    import os

    from aria_amd import ops

    x = rnd(T, D, seed=T).to(dev)
    w, b = rnd(D, seed=1).to(dev), rnd(D, seed=2).to(dev)
    prev = os.environ.get("ARIA_LAYERNORM_V1")
    try:
        os.environ["ARIA_LAYERNORM_V1"] = "1"
        want = ops.layernorm(x, w, b, 1e-6)
        os.environ.pop("ARIA_LAYERNORM_V1")
        got = ops.layernorm(x, w, b, 1e-6)
    finally:
        if prev is None:
            os.environ.pop("ARIA_LAYERNORM_V1", None)
        else:
            os.environ["ARIA_LAYERNORM_V1"] = prev
    for g, wnt in zip(got, want):
        assert torch.equal(g.cpu(), wnt.cpu())


def case_gemm_fused_gelu(dev, M, N, K):
    """fc1 + gelu_pytorch_tanh in the GEMM epilogue == the GEMM followed by the stand-alone GELU kernel, bit for bit (the activation
    sees bf16(acc + bias) in both), also through accumulate (x += gelu(...) is never used, but the order act -> accumulate is ABI)."""
    from aria_amd import ops

    a = rnd(M, K, seed=11).to(dev)
    w = rnd(N, K, seed=12, scale=0.3).to(dev)
    bias = rnd(N, seed=13).to(dev)
    want = ops.gelu_tanh(ops.gemm(a, w, bias=bias))
    got = ops.gemm(a, w, bias=bias, act="gelu_tanh")
    assert torch.equal(got, want)
    acc = torch.ones(M, N, dtype=bf16, device=dev)
    ops.gemm(a, w, bias=bias, act="gelu_tanh", out=acc, accumulate=True)
    close(acc, (want.float() + 1.0).to(bf16), 2 ** -7, 2 ** -7)  # (the unrounded activation is added: one rounding fewer)


def case_gemm_strided_views(dev):
    from aria_amd import ops

    T, D = 40, 64
    x = rnd(T, D, seed=4)
    w = rnd(D, D, seed=5, scale=0.2)
    qkv = torch.zeros(T, 3 * D, dtype=bf16, device=dev)
    ops.gemm(x.to(dev), w.to(dev), out=qkv[:, D:2 * D])
    ref = (x.float() @ w.float().t()).to(bf16)
    close(qkv[:, D:2 * D], ref, 1e-2, 5e-2)
    assert float(qkv[:, :D].abs().max()) == 0 and float(qkv[:, 2 * D:].abs().max()) == 0
    y = ops.gemm(qkv[:, D:2 * D], w.to(dev))
    close(y, (qkv[:, D:2 * D].cpu().float() @ w.float().t()).to(bf16), 1e-2, 5e-2)


def case_grouped_gemm(dev, counts, K=72, N=136):
    from aria_amd import ops

    E = len(counts)
    M = sum(counts)
    a = rnd(M, K, seed=6)
    w = rnd(E, K, N, seed=7, scale=0.3)
    tpe = torch.tensor(counts)
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(tpe, 0)
    offd = off.to(dev)
    want = O.sequential_gemm(a.float(), w.float(), tpe)
    dy = rnd(M, N, seed=8)
    if M:
        got = ops.grouped_gemm(a.to(dev), w.to(dev), offd)
        close(got, want.to(bf16), 1e-2, 1e-2 * K ** 0.5)
        da = ops.grouped_gemm(dy.to(dev), w.to(dev), offd, w_is_kn=False)
        want_da = O.sequential_gemm(dy.float(), w.float().transpose(1, 2), tpe)
        close(da, want_da.to(bf16), 1e-2, 1e-2 * N ** 0.5)
    dw = ops.grouped_gemm_wgrad(a.to(dev), dy.to(dev), offd, E, out_dtype=torch.float32)
    want_dw = torch.zeros(E, K, N)
    s = 0
    for e, n in enumerate(counts):
        want_dw[e] = a[s:s + n].float().t() @ dy[s:s + n].float()
        s += n
    close(dw, want_dw, 1e-4, 1e-3)
    dwb = ops.grouped_gemm_wgrad(a.to(dev), dy.to(dev), offd, E)
    close(dwb, want_dw.to(bf16), 1e-2, 1e-2 * max(1, max(counts)) ** 0.5)


def case_gemm_swiglu_fused(dev, counts, K, I, T_dense):
    """fc1 + glu in one launch (aria_grouped_gemm_swiglu_bf16 / aria_gemm_swiglu_bf16) == the two-step chain, bit for bit: h (both halves),
    act, with and without the h output; ragged / empty experts; the chain itself is checked against the oracle (moe_lm.py:505-507)."""
    from aria_amd import ops

    E, M = len(counts), sum(counts)
    a = rnd(M, K, seed=71).to(dev)
    w = rnd(E, K, 2 * I, seed=72, scale=0.2).to(dev)
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(torch.tensor(counts), 0)
    offd = off.to(dev)
    assert ops.glu_fusable(K, 2 * I)
    h_ref = ops.grouped_gemm(a, w, offd)
    act_ref = ops.swiglu(h_ref)
    want = O.glu(O.sequential_gemm(a.cpu().float(), w.cpu().float(), torch.tensor(counts)).to(bf16).float())
    close(act_ref, want.to(bf16), 2e-2, 2e-2)
    h, act = ops.grouped_gemm_swiglu(a, w, offd, want_h=True)
    assert torch.equal(h.cpu(), h_ref.cpu()) and torch.equal(act.cpu(), act_ref.cpu())
    h2, act2 = ops.grouped_gemm_swiglu(a, w, offd, want_h=False)
    assert h2 is None and torch.equal(act2.cpu(), act_ref.cpu())
    # dense form on [gate; up] rows (shared expert)
    x = rnd(T_dense, K, seed=73).to(dev)
    wd = rnd(2 * I, K, seed=74, scale=0.2).to(dev)
    hd_ref = ops.gemm(x, wd)
    actd_ref = ops.swiglu(hd_ref)
    hd, actd = ops.gemm_swiglu(x, wd, want_h=True)
    assert torch.equal(hd.cpu(), hd_ref.cpu()) and torch.equal(actd.cpu(), actd_ref.cpu())
    _, actd2 = ops.gemm_swiglu(x, wd, want_h=False)
    assert torch.equal(actd2.cpu(), actd_ref.cpu())


def case_gemm_swiglu_split(dev, counts, K, I, T_dense):
    """The gptfast form of the fused gate / up + SwiGLU launch (aria_grouped_gemm_swiglu_split_bf16 / aria_gemm_swiglu_split_bf16,
    gemm3_kernel<false, false, 6>): w1 and w3 are separate [E, I, K] tensors of one allocation ([N, K] form) == two grouped GEMMs + the
    stand-alone SwiGLU, bit for bit (h halves and act); tensors that are not laid out that way are refused by the host check."""
    from aria_amd import ops

    E, M = len(counts), sum(counts)
    a = rnd(M, K, seed=91).to(dev)
    buf = rnd(2, E, I, K, seed=92, scale=0.2).to(dev)
    w1, w3 = buf[0], buf[1]
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(torch.tensor(counts), 0)
    offd = off.to(dev)
    assert ops.glu_split_fusable(w1, w3) and not ops.glu_split_fusable(w3, w1) and not ops.glu_split_fusable(w1, w3.clone())
    h1, h3 = ops.grouped_gemm(a, w1, offd, w_is_kn=False), ops.grouped_gemm(a, w3, offd, w_is_kn=False)
    act_ref = ops.swiglu(h1, h3)
    h, act = ops.grouped_gemm_swiglu_split(a, w1, w3, offd, want_h=True)
    assert torch.equal(act.cpu(), act_ref.cpu()) and torch.equal(h[:, :I].cpu(), h1.cpu()) and torch.equal(h[:, I:].cpu(), h3.cpu())
    assert torch.equal(ops.grouped_gemm_swiglu_split(a, w1, w3, offd)[1].cpu(), act_ref.cpu())
    x = rnd(T_dense, K, seed=93).to(dev)
    dbuf = rnd(2, 2 * I, K, seed=94, scale=0.2).to(dev)   # the shared expert: [2 I, K] each
    s1, s3 = dbuf[0], dbuf[1]
    assert ops.glu_split_fusable(s1, s3)
    ops.GEMM_SPLIT_K = False  # (the stand-alone GEMMs may split the last round along K: another fp32 summation order)
    try:
        ref = ops.swiglu(ops.gemm(x, s1), ops.gemm(x, s3))
    finally:
        ops.GEMM_SPLIT_K = True
    assert torch.equal(ops.gemm_swiglu_split(x, s1, s3)[1].cpu(), ref.cpu())


def case_gemm_lora_ext(dev, counts, K, I, T_dense, r=8, expect_fused=True):
    """LoRA as a K-extension of the base GEMM (aria_*_lora_bf16; GroupedGemmLoraLayer.forward aria/lora/layers.py:129-139, peft's Linear
    adapter): result = base(x) + lora_B(scaling * lora_A(x)) from ONE launch, for every base form the decoder layer uses -- dense [N, K]
    weights (q/k/v/o, down), the dgrad form ([K, N]), dense gate|up + SwiGLU, grouped [E, K, N] experts forward and dgrad form, grouped
    fc1 + SwiGLU -- on ragged / empty experts.  Checks per form: (1) vs the fp32 oracle (O.lora_linear / O.lora_grouped_gemm: the reference's
    own line); (2) with a ZERO adapter input the launch gives the plain base launch's bits (the extension tile adds exact zeros); (3) with a
    ZERO base input it gives the bits of the adapter product run as a GEMM of its own (same accumulation); (4) the two-launch fallback
    (ARIA_FUSE_LORA=0) agrees to rounding."""
    import os

    from aria_amd import hip, ops

    lib = hip.get_lib().cdll
    E, M = len(counts), sum(counts)
    tpe = torch.tensor(counts)
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(tpe, 0)
    offd = off.to(dev)

    def fused(fn, *a, **k):
        out = fn(*a, **k)
        if expect_fused:
            assert lib.aria_last_gemm_variant() == 3
        return out

    def unfused(fn, *a, **k):
        before = os.environ.get("ARIA_FUSE_LORA")
        os.environ["ARIA_FUSE_LORA"] = "0"
        try:
            return fn(*a, **k)
        finally:
            if before is None:
                os.environ.pop("ARIA_FUSE_LORA", None)
            else:
                os.environ["ARIA_FUSE_LORA"] = before

    # ---- dense, Linear weight [N, K] (N = 2 I so that the same tensors serve the SwiGLU form), adapter B [N, r]
    N = 2 * I
    x = rnd(T_dense, K, seed=201).to(dev)
    w = rnd(N, K, seed=202, scale=0.2).to(dev)
    u = rnd(T_dense, r, seed=203).to(dev)
    lb = rnd(N, r, seed=204, scale=0.3).to(dev)
    want = x.float().cpu() @ w.float().cpu().t() + u.float().cpu() @ lb.float().cpu().t()
    got = fused(ops.gemm_lora, x, w, u, lb)
    close(got, want.to(bf16), 1e-2, 1e-2 * K ** 0.5)
    ops.GEMM_SPLIT_K = False   # (the plain launches below must use one K order: the fused one never splits)
    try:
        assert torch.equal(fused(ops.gemm_lora, x, w, torch.zeros_like(u), lb).cpu(), ops.gemm(x, w).cpu())
        close(unfused(ops.gemm_lora, x, w, u, lb), got, 2e-2, 2e-2 * K ** 0.5)
        # ---- dgrad form: b [K', N'] = the same storage read as [N, K] -> C = dy [T, N] x w [N, K]; adapter A as [r, K]
        dy = rnd(T_dense, N, seed=205).to(dev)
        la = rnd(r, K, seed=206, scale=0.3).to(dev)
        if N % 64 == 0:
            want = dy.float().cpu() @ w.float().cpu() + u.float().cpu() @ la.float().cpu()
            few = expect_fused and ((T_dense + 255) // 256) * ((K + 255) // 256) < 192   # (too few output tiles for the 256 x 256 kernels: two launches)
            expect_fused = expect_fused and not few
            got = fused(ops.gemm_lora, dy, w, u, la, b_oc=True)
            close(got, want.to(bf16), 1e-2, 1e-2 * N ** 0.5)
            assert torch.equal(fused(ops.gemm_lora, dy, w, torch.zeros_like(u), la, b_oc=True).cpu(), ops.gemm(dy, w, b_oc=True).cpu())
            expect_fused = expect_fused or few
        # ---- dense gate | up + SwiGLU
        h_want = x.float().cpu() @ w.float().cpu().t() + u.float().cpu() @ lb.float().cpu().t()
        h, act = fused(ops.gemm_swiglu_lora, x, w, u, lb, want_h=True)
        close(h, h_want.to(bf16), 1e-2, 1e-2 * K ** 0.5)
        assert torch.equal(act.cpu(), ops.swiglu(h).cpu())          # the epilogue rounds h, then applies glu: the two-step chain's bits
        h0, act0 = ops.gemm_swiglu(x, w, want_h=True)
        hz, actz = fused(ops.gemm_swiglu_lora, x, w, torch.zeros_like(u), lb, want_h=True)
        assert torch.equal(hz.cpu(), h0.cpu()) and torch.equal(actz.cpu(), act0.cpu())
        assert torch.equal(fused(ops.gemm_swiglu_lora, x, w, u, lb, want_h=False)[1].cpu(), act.cpu())
    finally:
        ops.GEMM_SPLIT_K = True
    if M == 0:
        return
    # ---- grouped experts [E, K, N] forward: adapter A [E, K, r] applied by the caller, B [E, r, N]
    a = rnd(M, K, seed=207).to(dev)
    we = rnd(E, K, N, seed=208, scale=0.2).to(dev)
    ue = rnd(M, r, seed=209).to(dev)
    lbe = rnd(E, r, N, seed=210, scale=0.3).to(dev)
    want = O.sequential_gemm(a.float().cpu(), we.float().cpu(), tpe) + O.sequential_gemm(ue.float().cpu(), lbe.float().cpu(), tpe)
    got = fused(ops.grouped_gemm_lora, a, we, offd, ue, lbe)
    close(got, want.to(bf16), 1e-2, 1e-2 * K ** 0.5)
    assert torch.equal(fused(ops.grouped_gemm_lora, a, we, offd, torch.zeros_like(ue), lbe).cpu(), ops.grouped_gemm(a, we, offd).cpu())
    close(unfused(ops.grouped_gemm_lora, a, we, offd, ue, lbe), got, 2e-2, 2e-2 * K ** 0.5)
    # ... the reference's whole line, lora_A included (O.lora_grouped_gemm), scaling folded into u
    lae = rnd(E, K, r, seed=211, scale=0.2).to(dev)
    u_dev = ops.grouped_gemm(a, lae, offd)
    sc = 4.0
    want = O.lora_grouped_gemm(a.float().cpu(), we.float().cpu(), lae.float().cpu(), lbe.float().cpu(), tpe, sc)
    got = fused(ops.grouped_gemm_lora, a, we, offd, (u_dev.float() * sc).to(bf16), lbe)
    close(got, want.to(bf16), 2e-2, 2e-2 * K ** 0.5)
    # ---- grouped dgrad form: dy [M, N] x we[e]^T -> [M, K]; adapter factor read as [E, K, r]
    if N % 64 == 0:
        dye = rnd(M, N, seed=212).to(dev)
        want = O.sequential_gemm(dye.float().cpu(), we.float().cpu().transpose(1, 2), tpe) + \
            O.sequential_gemm(ue.float().cpu(), lae.float().cpu().transpose(1, 2), tpe)
        got = fused(ops.grouped_gemm_lora, dye, we, offd, ue, lae, w_is_kn=False)
        close(got, want.to(bf16), 1e-2, 1e-2 * N ** 0.5)
    # ---- grouped fc1 + SwiGLU
    h, act = fused(ops.grouped_gemm_swiglu_lora, a, we, offd, ue, lbe, want_h=True)
    want = O.sequential_gemm(a.float().cpu(), we.float().cpu(), tpe) + O.sequential_gemm(ue.float().cpu(), lbe.float().cpu(), tpe)
    close(h, want.to(bf16), 1e-2, 1e-2 * K ** 0.5)
    assert torch.equal(act.cpu(), ops.swiglu(h).cpu())
    h0, act0 = ops.grouped_gemm_swiglu(a, we, offd, want_h=True)
    hz, actz = fused(ops.grouped_gemm_swiglu_lora, a, we, offd, torch.zeros_like(ue), lbe, want_h=True)
    assert torch.equal(hz.cpu(), h0.cpu()) and torch.equal(actz.cpu(), act0.cpu())


def case_gemm_qkv_rope_cache(dev, B, S, D, hd, K, S_cache, shuffled_pos=False):
    """K7: wqkv projection + interleaved RoPE + KV-cache write in one launch (``gemm3_kernel<false, false, 7>``; gptfast/model.py:413-435,
    67-93, 519-531) == the chain gemm (v3, no split-K) -> rope_interleaved_ -> row copies, bit for bit: q, and the cache rows at the tokens'
    positions (rows nobody wrote keep their sentinel); then against the oracle's fp32 interleaved rotation."""
    import os

    from aria_amd import ops

    M = B * S
    x = rnd(M, K, seed=95).to(dev)
    w = rnd(3 * D, K, seed=96, scale=0.2).to(dev)
    n_pos = max(S, S_cache)
    ang = torch.outer(torch.arange(n_pos).float(), 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd)))
    fc = torch.stack([ang.cos(), ang.sin()], dim=-1).to(bf16).contiguous()          # [n_pos, hd/2, 2] (precompute_freqs_cis :500-516)
    if shuffled_pos:
        pos = torch.stack([torch.randperm(S_cache, generator=torch.Generator().manual_seed(97 + b))[:S] for b in range(B)]).reshape(-1).to(torch.int32)
    else:
        pos = torch.arange(S, dtype=torch.int32).repeat(B)
    assert ops.qkv_rope_cache_fusable(D, K, hd)
    kc = torch.full((B, S_cache, D), 7.0, dtype=bf16, device=dev)
    vc = torch.full((B, S_cache, D), 7.0, dtype=bf16, device=dev)
    q = ops.gemm_qkv_rope_cache(x, w, fc.to(dev), pos.to(dev), kc, vc, S, hd)
    prev, ops.GEMM_SPLIT_K = os.environ.get("ARIA_GEMM_FORCE"), False
    os.environ["ARIA_GEMM_FORCE"] = "3"
    try:
        qkv = ops.gemm(x, w)
    finally:
        ops.GEMM_SPLIT_K = True
        if prev is None:
            os.environ.pop("ARIA_GEMM_FORCE")
        else:
            os.environ["ARIA_GEMM_FORCE"] = prev
    raw = qkv.clone()
    ops.rope_interleaved_(qkv[:, :2 * D], fc.to(dev), 2 * D // hd, hd, pos.to(dev))
    assert torch.equal(q.cpu(), qkv[:, :D].cpu())
    rows = (torch.arange(M) // S) * S_cache + pos.long()
    assert torch.equal(kc.view(-1, D).cpu()[rows], qkv[:, D:2 * D].cpu()) and torch.equal(vc.view(-1, D).cpu()[rows], qkv[:, 2 * D:].cpu())
    untouched = torch.ones(B * S_cache, dtype=torch.bool)
    untouched[rows] = False
    assert bool((kc.view(-1, D).cpu()[untouched] == 7.0).all()) and bool((vc.view(-1, D).cpu()[untouched] == 7.0).all())
    # oracle: interleaved pairs rotated in fp32 with the bf16 table (gptfast/model.py:519-531)
    xq = raw[:, :D].cpu().float().view(M, D // hd, hd // 2, 2)
    f = fc[pos.long()].float().view(M, 1, hd // 2, 2)
    want = torch.stack([xq[..., 0] * f[..., 0] - xq[..., 1] * f[..., 1], xq[..., 1] * f[..., 0] + xq[..., 0] * f[..., 1]], dim=-1).reshape(M, D)
    same(q, want.to(bf16), str(dev) == "cpu")   # (hardware: the compiler contracts x0 cs - x1 sn into an fma)


def case_gemm_swiglu_gather(dev, T, E, k, K, I, seed=91):
    """K2: the fused fc1 + SwiGLU launches with the dispatcher's row gather in the A loader (``gemm3_kernel<.., .., 8 / 9>``: token matrix +
    row index instead of the permuted copy; moe_lm.py:326-334, 505-525; gptfast/model.py:243-254, 278-325) == permute followed by the
    un-gathered launch, bit for bit -- both weight forms, ragged / empty experts from a real routing, a column edge tile (I = 128 mod 256)."""
    from aria_amd import ops

    g = torch.Generator().manual_seed(seed)
    x = rnd(T, K, seed=seed).to(dev)
    logits = torch.randn(T, E, generator=g)
    logits[:, 1] = -50.0                                   # an expert nobody routes to
    scores, idx, counts = ops.moe_route(logits.to(bf16).to(dev), k)
    offsets, sorted_src, inv = ops.moe_sort(idx, counts)
    rows = ops.permuted_token_rows(sorted_src, k)
    assert rows.dtype == torch.int32 and rows.numel() == T * k and int(counts.cpu()[1]) == 0
    perm = ops.moe_permute(x, sorted_src, k)
    assert torch.equal(perm.cpu(), x.cpu()[rows.cpu().long()])
    w = rnd(E, K, 2 * I, seed=seed + 1, scale=0.2).to(dev)
    assert ops.glu_fusable(K, 2 * I) and ops.gather_fusable(K)
    h_ref, act_ref = ops.grouped_gemm_swiglu(perm, w, offsets, want_h=True)
    h, act = ops.grouped_gemm_swiglu_gather(x, rows, w, offsets, want_h=True)
    assert torch.equal(h.cpu(), h_ref.cpu()) and torch.equal(act.cpu(), act_ref.cpu())
    assert ops.grouped_gemm_swiglu_gather(x, rows, w, offsets, want_h=False)[0] is None
    pair = torch.empty((2, E, I, K), dtype=bf16, device=dev)
    pair[0].copy_(w[:, :, :I].transpose(1, 2))
    pair[1].copy_(w[:, :, I:].transpose(1, 2))
    assert ops.glu_split_fusable(pair[0], pair[1])
    h6_ref, act6_ref = ops.grouped_gemm_swiglu_split(perm, pair[0], pair[1], offsets, want_h=True)
    h6, act6 = ops.grouped_gemm_swiglu_split_gather(x, rows, pair[0], pair[1], offsets, want_h=True)
    assert torch.equal(h6.cpu(), h6_ref.cpu()) and torch.equal(act6.cpu(), act6_ref.cpu())
    # the oracle's own chain on the gathered rows (sequential_gemm + glu), with the usual GEMM tolerance
    want = O.glu(O.sequential_gemm(x.cpu().float()[rows.cpu().long()], w.cpu().float(), counts.cpu().long()).to(bf16).float())
    close(act, want, 3e-2, 3e-2)


def case_gemm_dswiglu_fused(dev, counts, K, I, T_dense):
    """experts.fc2's input gradient + the backward of glu in one launch (aria_grouped_gemm_dswiglu_bf16 / aria_gemm_dswiglu_bf16,
    gemm3_kernel<.., .., 5>) == the two-step chain (grouped GEMM with [N, K] weights, then aria_swiglu_bwd), bit for bit; ragged / empty
    experts, a partial last column tile (I = 128 (mod 256)); the chain itself against autograd through the oracle's glu (moe_lm.py:505-507)."""
    from aria_amd import ops

    E, M = len(counts), sum(counts)
    dy = rnd(M, K, seed=81).to(dev)
    w = rnd(E, I, K, seed=82, scale=0.2).to(dev)          # fc2.weight [E, I, K]: forward act [., I] @ w[e] -> [., K]
    h = rnd(M, 2 * I, seed=83, scale=1.5).to(dev)          # the forward's [gate | up]
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(torch.tensor(counts), 0)
    offd = off.to(dev)
    assert ops.dglu_fusable(I, K)
    dact_ref = ops.grouped_gemm(dy, w, offd, w_is_kn=False)
    dh_ref = ops.swiglu_bwd(h, dact_ref)
    # oracle: d/dh of sum(glu(h) * d_act) with d_act as the device rounded it
    hf = h.cpu().float().requires_grad_(True)
    (O.glu(hf) * dact_ref.cpu().float()).sum().backward()
    close(dh_ref, hf.grad.to(bf16), 3e-2, 3e-2)
    dh = ops.grouped_gemm_dswiglu(dy, w, offd, h)
    assert torch.equal(dh.cpu(), dh_ref.cpu()), float((dh.float() - dh_ref.float()).abs().max())
    # dense forms (shared expert): down_proj.weight [K, I] read as the [k][n] operand, and the [I, K] form.  The reference chain runs
    # WITHOUT the remainder split-K (ops.GEMM_SPLIT_K: fp32 partial sums in another order -- at Aria's widths the stand-alone GEMM would
    # take it; the fused launch never splits)
    g = rnd(T_dense, K, seed=84).to(dev)
    hd = rnd(T_dense, 2 * I, seed=85, scale=1.5).to(dev)
    wkn = rnd(K, I, seed=86, scale=0.2).to(dev)
    wnk = wkn.t().contiguous()
    ops.GEMM_SPLIT_K = False
    try:
        ref = ops.swiglu_bwd(hd, ops.gemm(g, wkn, b_oc=True))
        ref2 = ops.swiglu_bwd(hd, ops.gemm(g, wnk))
    finally:
        ops.GEMM_SPLIT_K = True
    got = ops.gemm_dswiglu(g, wkn, hd, b_oc=True)
    assert torch.equal(got.cpu(), ref.cpu()), float((got.float() - ref.float()).abs().max())
    got2 = ops.gemm_dswiglu(g, wnk, hd, b_oc=False)
    assert torch.equal(got2.cpu(), ref2.cpu())
    close(ops.swiglu_bwd(hd, ops.gemm(g, wkn, b_oc=True)), got, 2e-2, 2e-2)  # (and the split-K chain agrees within bf16 rounding)


# ------------------------------------------------------------------------------------------ routing
def case_route(dev, T, E, k, dtype, exact):
    from aria_amd import ops

    g = torch.Generator().manual_seed(T + E)
    logits = torch.randn(T, E, generator=g) * 0.5
    logits[:, 3] = logits[:, 1]  # force ties everywhere
    logits[0] = 0.25              # a fully tied row
    logits = logits.to(dtype)
    scores, idx, counts = ops.moe_route(logits.to(dev), k)
    ws, wi, wc = O.router_routing(logits, k, E)
    assert torch.equal(idx.cpu().long(), wi)          # router indices bit-exact under the tie protocol
    assert torch.equal(counts.cpu().long(), wc)
    if dtype == bf16:
        same(scores, ws, exact)
    else:
        assert torch.allclose(scores.cpu(), ws, atol=1e-6)


def case_sample_topk(dev, V, top_k, temperature, seed=0):
    """aria_sample_topk (one launch) == gptfast/generate.py's tensor path (logits_to_probs + multinomial_sample_one_no_sync, restated in
    aria_amd.gptfast) on the SAME Exp(1) draws: the same token, with ties at the k-th largest logit (the tensor path keeps them), a
    maximum shared by several tokens, negative and positive logits, top_k >= V and top_k None."""
    from aria_amd import gptfast as G
    from aria_amd import ops

    g = torch.Generator().manual_seed(seed + V)
    logits = (torch.randn(V, generator=g) * 3.0).to(bf16)
    if V > 64:
        srt = torch.sort(logits.float(), descending=True).values
        kk = min(top_k or V, V)
        logits[torch.randperm(V, generator=g)[:5]] = srt[kk - 1].to(bf16)      # ties exactly at the threshold
        logits[torch.randperm(V, generator=g)[:3]] = srt[0].to(bf16)           # a shared maximum
    for trial in range(6):
        q = torch.empty(V).exponential_(1, generator=g)
        x = logits.float() / max(temperature, 1e-5)
        if top_k is not None:
            v, _ = torch.topk(x, min(top_k, V))
            x = torch.where(x < v[-1], torch.tensor(-float("inf")), x)
        want = int(torch.argmax(torch.softmax(x, dim=-1) / q))
        got = int(ops.sample_topk(logits.to(dev), q.to(dev), temperature, top_k).cpu())
        if got != want:  # the normaliser is skipped: only an exact fp tie of p/q may pick the other index
            pw, pg = (torch.softmax(x, -1) / q)[want], (torch.softmax(x, -1) / q)[got]
            assert abs(float(pw - pg)) <= 1e-6 * float(pw) and x[got] > -float("inf"), (trial, got, want)


def case_decode_route(dev, E, k, n_rows=24):
    """The decode engine's one-token routing (maximum on a DPP ladder, expert id from ballots over the id-ordered slots) == the batched
    router kernel == the oracle (TopKRouter.routing, moe_lm.py:243-273), ids AND bf16 scores bit for bit, on rows full of ties: equal
    values across lanes, across the four 64-expert slots of a lane, and a fully tied row."""
    from aria_amd import ops

    g = torch.Generator().manual_seed(E * 10 + k)
    logits = (torch.randn(n_rows, E, generator=g) * 0.5).to(bf16)
    logits[:, E // 2] = logits[:, 1]                    # ties in every row
    if E > 70:
        logits[:, 66] = logits[:, 2]                    # the same value in two slots of one lane
        logits[3, 2] = logits[3].float().max().to(bf16)
        logits[3, 66] = logits[3, 2]                    # ... as the row maximum
    logits[0] = 0.25                                    # a fully tied row
    logits[1, E - 1] = 3.0                              # the maximum in the last expert
    ws, wi, _ = O.router_routing(logits, k, E)
    bs, bi, _ = ops.moe_route(logits.to(dev), k)
    for r in range(n_rows):
        sc, idx = ops.decode_route(logits[r].to(dev).contiguous(), k)
        assert torch.equal(idx.cpu().long(), wi[r]), (r, idx.cpu(), wi[r])
        assert torch.equal(idx.cpu(), bi[r].cpu()) and torch.equal(sc.cpu(), bs[r].cpu()), r


def case_dispatch(dev, T, E, k, exact, D=72):
    from aria_amd import ops

    g = torch.Generator().manual_seed(T)
    logits = torch.randn(T, E, generator=g).to(bf16)
    scores, idx, counts = ops.moe_route(logits.to(dev), k)
    off, sorted_src, inv = ops.moe_sort(idx, counts)
    x = rnd(T, D, seed=9)
    want_perm, want_sorted = O.token_permutation(x, idx.cpu().long(), k)
    assert torch.equal(sorted_src.cpu().long(), want_sorted)  # stable order == argsort(stable=True)
    assert torch.equal(off.cpu().long(), torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(counts.cpu().long(), 0)]))
    assert torch.equal(inv.cpu().long()[want_sorted], torch.arange(T * k))
    perm = ops.moe_permute(x.to(dev), sorted_src, k)
    assert torch.equal(perm.cpu(), want_perm)
    eo = rnd(T * k, D, seed=10)
    shared = rnd(T, D, seed=11)
    got = ops.moe_unpermute(eo.to(dev), inv, scores, k, add=shared.to(dev))
    want = O.token_unpermutation(eo, scores.cpu(), want_sorted, k, (T, D)) + shared
    assert torch.equal(got.cpu(), want)  # only mul/add/round: bit-exact on hardware too
    gsum = ops.moe_unpermute(eo.to(dev), inv, None, k)
    close(gsum, torch.zeros(T * k, D).index_copy_(0, want_sorted, eo.float()).view(T, k, D).sum(1).to(bf16), 1e-2, 1e-2)


def case_dispatch_fixed_width(dev, T, E, k, D):
    """r04: the permute / unpermute / unpermute-backward kernels with a compile-time row width (every chunk of a row in flight together, the
    gradient row fetched once per token, the dot product's wave sum on DPP) against the generic-width kernels (ARIA_MOE_GENERIC_DISPATCH=1):
    rows bit-identical; dscores identical up to the order of the 64-lane sum; and against the oracle (moe_lm.py:313-365)."""
    import os

    from aria_amd import ops

    g = torch.Generator().manual_seed(T + D)
    logits = torch.randn(T, E, generator=g).to(bf16)
    scores, idx, counts = ops.moe_route(logits.to(dev), k)
    off, sorted_src, inv = ops.moe_sort(idx, counts)
    x, eo, shared, dout = rnd(T, D, seed=9), rnd(T * k, D, seed=10), rnd(T, D, seed=11), rnd(T, D, seed=12)

    def run():
        perm = ops.moe_permute(x.to(dev), sorted_src, k)
        comb = ops.moe_unpermute(eo.to(dev), inv, scores, k, add=shared.to(dev))
        plain = ops.moe_unpermute(eo.to(dev), inv, None, k)
        d_eo, dsc = ops.moe_unpermute_bwd(dout.to(dev), eo.to(dev), inv, scores, k)
        return [t.cpu() for t in (perm, comb, plain, d_eo, dsc)]

    fast = run()
    os.environ["ARIA_MOE_GENERIC_DISPATCH"] = "1"
    try:
        generic = run()
    finally:
        os.environ.pop("ARIA_MOE_GENERIC_DISPATCH")
    for a, b, what in zip(fast[:4], generic[:4], ("permute", "unpermute + shared", "unpermute", "d_eo")):
        assert torch.equal(a, b), what
    close(fast[4], generic[4].float(), 1e-2, 1e-2)                      # dscores: another order of the 64-lane sum
    want_perm, want_sorted = O.token_permutation(x, idx.cpu().long(), k)
    assert torch.equal(fast[0], want_perm)
    assert torch.equal(fast[1], O.token_unpermutation(eo, scores.cpu(), want_sorted, k, (T, D)) + shared)
    eof, sf = eo.float().requires_grad_(True), scores.cpu().float().requires_grad_(True)
    O.token_unpermutation(eof, sf, want_sorted, k, (T, D)).backward(dout.float())
    close(fast[3], eof.grad, 1e-2, 1e-2)
    close(fast[4], sf.grad, 2e-2, 5e-2)


def case_moe_backward_pieces(dev):
    from aria_amd import ops

    T, E, k, D = 50, 8, 3, 40
    cfg = O.LMConfig(hidden_size=D, moe_num_experts=E, moe_topk=k, moe_z_loss_coeff=1e-2, moe_aux_loss_coeff=5e-2)
    logits = rnd(T, E, seed=12)
    lf = logits.float().requires_grad_(True)
    O._AuxLossScaler.scale = 0.5
    try:
        lz = O._AuxLossScaler.apply(lf, O.z_loss_func(lf, cfg.moe_z_loss_coeff))
        s, idx, tpe = O.router_routing(lz, k, E)
        probs = torch.softmax(lz, dim=-1, dtype=torch.float32)
        s2 = O._AuxLossScaler.apply(s, O.switch_load_balancing_loss_func(probs, tpe, k, cfg.moe_aux_loss_coeff))
        ds = rnd(T, k, seed=13)
        s2.backward(ds.float())
    finally:
        O._AuxLossScaler.scale = 1.0
    scores, idx_k, counts = ops.moe_route(logits.to(dev), k)
    assert torch.equal(idx_k.cpu().long(), idx)
    dl = ops.moe_route_bwd(logits.to(dev), idx_k, scores, ds.to(dev), counts, cfg.moe_z_loss_coeff, cfg.moe_aux_loss_coeff, 0.5)
    close(dl, lf.grad, 2e-2, 2e-3)

    off, sorted_src, inv = ops.moe_sort(idx_k, counts)
    eo = rnd(T * k, D, seed=14)
    dout = rnd(T, D, seed=15)
    eof = eo.float().requires_grad_(True)
    sf = scores.cpu().float().requires_grad_(True)
    out = O.token_unpermutation(eof, sf, sorted_src.cpu().long(), k, (T, D))
    out.backward(dout.float())
    d_eo, dsc = ops.moe_unpermute_bwd(dout.to(dev), eo.to(dev), inv, scores, k)
    close(d_eo, eof.grad, 1e-2, 1e-2)
    close(dsc, sf.grad, 2e-2, 5e-2)


def case_swiglu(dev, exact):
    from aria_amd import ops

    M, I = 37, 24
    h = rnd(M, 2 * I, seed=16)
    want = O.glu(h)
    same(ops.swiglu(h.to(dev)), want, exact)
    gate, up = h[:, :I].contiguous(), h[:, I:].contiguous()
    same(ops.swiglu(gate.to(dev), up.to(dev)), want, exact)
    hf = h.float().requires_grad_(True)
    dact = rnd(M, I, seed=17)
    O.glu(hf).backward(dact.float())
    close(ops.swiglu_bwd(h.to(dev), dact.to(dev)), hf.grad, 2e-2, 2e-2)
    dg, du = ops.swiglu_bwd(gate.to(dev), dact.to(dev), up.to(dev))
    close(torch.cat([dg, du], 1), hf.grad, 2e-2, 2e-2)


# ------------------------------------------------------------------------------------------ norm / rope
def case_rmsnorm(dev, T, D, exact):
    from aria_amd import ops

    x, res, w = rnd(T, D, seed=18), rnd(T, D, seed=19), (1 + 0.1 * torch.randn(D, generator=torch.Generator().manual_seed(1))).to(bf16)
    y, h, rstd = ops.rmsnorm(x.to(dev), w.to(dev), 1e-6)
    same(y, O.rms_norm(x, w, 1e-6), exact)
    y2, h2, rstd2 = ops.rmsnorm(x.to(dev), w.to(dev), 1e-6, residual=res.to(dev))
    assert torch.equal(h2.cpu(), x + res)
    same(y2, O.rms_norm(x + res, w, 1e-6), exact)
    hf = (x + res).float().requires_grad_(True)
    wf = w.float().requires_grad_(True)
    dy, dres = rnd(T, D, seed=20), rnd(T, D, seed=21)
    yf = O.rms_norm(hf, wf, 1e-6)
    (yf * dy.float()).sum().backward()
    dx, dw = ops.rmsnorm_bwd(dy.to(dev), h2, w.to(dev), rstd2, dres=dres.to(dev))
    close(dx, hf.grad + dres.float(), 2e-2, 2e-2)
    close(dw, wf.grad, 2e-2, 2e-2 * T ** 0.5)


def case_rope(dev):
    from aria_amd import ops

    B, S, H, hd = 2, 7, 3, 32
    D = H * hd
    qkv = rnd(B * S, 3 * D, seed=22)
    pos = torch.arange(S)[None].expand(B, S)
    cos, sin = O.rope_cos_sin(pos[:1], hd, 5e6, bf16)
    cos, sin = cos[0].contiguous(), sin[0].contiguous()
    q = qkv[:, :D].view(B, S, H, hd).transpose(1, 2)
    k = qkv[:, D:2 * D].view(B, S, H, hd).transpose(1, 2)
    cb, sb = O.rope_cos_sin(pos, hd, 5e6, bf16)
    wq, wk = O.apply_rope_half(q, k, cb, sb)
    work = qkv.clone().to(dev)
    ops.rope_(work[:, :2 * D], cos.to(dev), sin.to(dev), S, 2 * H, hd)
    wc = work.cpu()
    assert torch.equal(wc[:, :D].view(B, S, H, hd).transpose(1, 2), wq)  # mul/add/round only: exact everywhere
    assert torch.equal(wc[:, D:2 * D].view(B, S, H, hd).transpose(1, 2), wk)
    assert torch.equal(wc[:, 2 * D:], qkv[:, 2 * D:])
    ops.rope_(work[:, :2 * D], cos.to(dev), sin.to(dev), S, 2 * H, hd, inverse=True)
    close(work, qkv, 2e-2, 2e-2)


def case_add(dev):
    from aria_amd import ops

    a, b = rnd(5, 16, seed=23), rnd(5, 16, seed=24)
    assert torch.equal(ops.add(a.to(dev), b.to(dev)).cpu(), a + b)


# ------------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v, B, S, H, hd, scale, causal, kv_len=None):
    """fp32 oracle on bf16 inputs: O.attention_eager with key padding from kv_len."""
    qh = q.float().view(B, S, H, hd).transpose(1, 2)
    kh = k.float().view(B, S, H, hd).transpose(1, 2)
    vh = v.float().view(B, S, H, hd).transpose(1, 2)
    pad = None
    if kv_len is not None:
        pad = torch.arange(S)[None, :] >= kv_len[:, None]
    o = O.attention_eager(qh, kh, vh, scale, causal, key_padding=pad)
    return o.transpose(1, 2).reshape(B * S, H * hd)


def case_attention(dev, B, S, H, hd, causal, use_len, bwd=True):
    from aria_amd import hip, ops

    D = H * hd
    qkv = rnd(B * S, 3 * D, seed=30, scale=1.0)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    kv_len = None
    if use_len:
        kv_len = torch.tensor([max(1, S - 3 - 5 * b) for b in range(B)], dtype=torch.int32)
    scale = hd ** -0.5
    qd = qkv.to(dev)
    o, lse = ops.attention_fwd(qd[:, :D], qd[:, D:2 * D], qd[:, 2 * D:], B, S, H, hd, scale, causal,
                               None if kv_len is None else kv_len.to(dev))
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    want = attn_ref(qf, kf, vf, B, S, H, hd, scale, causal, kv_len)
    close(o, want, 2e-2, 2e-2)
    if not bwd:
        return
    do = rnd(B * S, D, seed=31)
    want.backward(do.float())
    dq, dk, dv = ops.attention_bwd(qd[:, :D], qd[:, D:2 * D], qd[:, 2 * D:], o, do.to(dev), lse, B, S, H, hd, scale, causal,
                                   None if kv_len is None else kv_len.to(dev))
    variant = int(hip.get_lib().cdll.aria_last_attn_bwd_variant())
    assert variant == (5 if hd == 128 else 2), variant
    close(dq, qf.grad, 3e-2, 3e-2)
    close(dk, kf.grad, 3e-2, 3e-2)
    close(dv, vf.grad, 3e-2, 3e-2)


# ------------------------------------------------------------------------------------------ loss
def case_cross_entropy(dev, T, V):
    from aria_amd import ops

    logits = rnd(T, V, seed=40, scale=2.0)
    g = torch.Generator().manual_seed(41)
    labels = torch.randint(0, V, (T,), generator=g)
    labels[::5] = -100
    lf = logits.float().requires_grad_(True)
    want = torch.nn.functional.cross_entropy(lf, labels, ignore_index=-100)
    want.backward()
    n = int((labels != -100).sum())
    ld = logits.to(dev)
    dl = torch.empty_like(ld)
    loss_sum, count, _ = ops.cross_entropy(ld, labels.to(torch.int32).to(dev), grad_scale=1.0 / n, dlogits=dl)
    assert int(count.cpu()) == n
    close(loss_sum.cpu() / n, want.detach().reshape(1), 1e-4, 1e-4)
    close(dl, lf.grad, 2e-2, 1e-4)


def case_attention_hd72_forward(dev, B, Sq, Skv, H, masked):
    """native head_dim 72 forward (frozen ViT / projector): reduction padded to 80 in LDS, output tiles 32+32+8."""
    from aria_amd import ops

    hd = 72
    D = H * hd
    q = rnd(B * Sq, D, seed=60)
    kv = rnd(B * Skv, 2 * D, seed=61)
    km = None
    if masked:
        km = (torch.rand(B, Skv, generator=torch.Generator().manual_seed(62)) > 0.25).to(torch.uint8)
        km[:, 0] = 1
    o, lse = ops.attention_fwd(q.to(dev), kv.to(dev)[:, :D], kv.to(dev)[:, D:], B, Sq, H, hd, hd ** -0.5, False,
                               key_mask=None if km is None else km.to(dev), Skv=Skv)
    qh = q.float().view(B, Sq, H, hd).transpose(1, 2)
    kh = kv[:, :D].float().view(B, Skv, H, hd).transpose(1, 2)
    vh = kv[:, D:].float().view(B, Skv, H, hd).transpose(1, 2)
    want = O.attention_eager(qh, kh, vh, hd ** -0.5, False, key_padding=None if km is None else (km == 0))
    close(o, want.transpose(1, 2).reshape(B * Sq, D), 2e-2, 2e-2)
    s = (qh @ kh.transpose(2, 3)) * hd ** -0.5
    if km is not None:
        s = s.masked_fill((km == 0)[:, None, None, :], float("-inf"))
    # the hd-72 kernel takes its row sums out of the MFMA (a ones column next to V), i.e. over the bf16-ROUNDED probabilities that also
    # multiply V -- numerator and denominator see the same rounding; the log-sum-exp therefore carries up to one bf16 rounding of a
    # probability (2^-9 relative -> 4e-3 absolute in the log) when a row has only a few keys, and averages out below 1e-3 for long rows
    close(lse, torch.logsumexp(s, dim=-1), 1e-3, 4e-3 if Skv < 64 else 1e-3)


def case_attention_cross_masked(dev, B, Sq, Skv, H, hd):
    """Sq != Skv with an arbitrary (non-prefix) key mask: the projector's cross-attention and the ViT's patch padding."""
    from aria_amd import ops

    D = H * hd
    q = rnd(B * Sq, D, seed=50)
    kv = rnd(B * Skv, 2 * D, seed=51)
    g = torch.Generator().manual_seed(52)
    km = (torch.rand(B, Skv, generator=g) > 0.3).to(torch.uint8)
    km[:, 0] = 1
    scale = hd ** -0.5
    qd, kvd = q.to(dev), kv.to(dev)
    o, lse = ops.attention_fwd(qd, kvd[:, :D], kvd[:, D:], B, Sq, H, hd, scale, False, key_mask=km.to(dev), Skv=Skv)
    qf = q.float().clone().requires_grad_(True)
    kf = kv[:, :D].float().clone().requires_grad_(True)
    vf = kv[:, D:].float().clone().requires_grad_(True)
    qh = qf.view(B, Sq, H, hd).transpose(1, 2)
    kh = kf.view(B, Skv, H, hd).transpose(1, 2)
    vh = vf.view(B, Skv, H, hd).transpose(1, 2)
    want = O.attention_eager(qh, kh, vh, scale, False, key_padding=(km == 0)).transpose(1, 2).reshape(B * Sq, D)
    close(o, want, 2e-2, 2e-2)
    do = rnd(B * Sq, D, seed=53)
    want.backward(do.float())
    dq, dk, dv = ops.attention_bwd(qd, kvd[:, :D], kvd[:, D:], o, do.to(dev), lse, B, Sq, H, hd, scale, False,
                                   key_mask=km.to(dev), Skv=Skv)
    close(dq, qf.grad, 3e-2, 3e-2)
    close(dk, kf.grad, 3e-2, 3e-2)
    close(dv, vf.grad, 3e-2, 3e-2)


def case_attention_masked_tiles(dev, hd, H, B=2, S=330):
    """A padded image masks a CONTIGUOUS patch range: key tiles without a single valid key are skipped by the forward and dQ kernels
    (nothing to add: every probability of the tile is 0) -- the first tile, a middle run and the last tiles fully masked, plus scattered
    keys, against the fp32 reference (idefics2 eager attention with the additive mask, vision_encoder.py:147-152); dK / dV of masked keys
    are exactly zero."""
    from aria_amd import ops

    D = H * hd
    g = torch.Generator().manual_seed(hd)
    qkv = torch.randn(B * S, 3 * D, generator=g).to(bf16)
    do = torch.randn(B * S, D, generator=g).to(bf16)
    km = torch.ones(B, S, dtype=torch.uint8)
    km[0, :64] = 0          # the FIRST tile has no valid key
    km[0, 128:256] = 0      # two middle tiles
    km[1, 192:] = 0         # everything from tile 3 on
    km[1, 5:40:3] = 0       # and scattered ones
    qd, kmd = qkv.to(dev), km.to(dev)
    o, lse = ops.attention_fwd(qd[:, :D], qd[:, D:2 * D], qd[:, 2 * D:], B, S, H, hd, hd ** -0.5, False, key_mask=kmd)
    dq, dk, dv = ops.attention_bwd(qd[:, :D], qd[:, D:2 * D], qd[:, 2 * D:], o, do.to(dev), lse, B, S, H, hd, hd ** -0.5, False, key_mask=kmd)
    q, k, v = (qkv[:, i * D:(i + 1) * D].float().view(B, S, H, hd).transpose(1, 2).clone().requires_grad_(True) for i in range(3))
    want = O.attention_eager(q, k, v, hd ** -0.5, False, key_padding=(km == 0))
    want.backward(do.float().view(B, S, H, hd).transpose(1, 2))
    close(o, want.detach().transpose(1, 2).reshape(B * S, D), 2e-2, 2e-2)
    for got, t in zip((dq, dk, dv), (q, k, v)):
        close(got, t.grad.transpose(1, 2).reshape(B * S, D), 3e-2, 3e-2)
    dead = (km == 0).reshape(-1)
    assert float(dk.float().cpu()[dead].abs().max()) == 0.0 and float(dv.float().cpu()[dead].abs().max()) == 0.0


def case_attention_forward_variants(dev, B, Sq, Skv, H, hd, causal, masked, use_len):
    """Round 4: ``attn_fwd3`` (the next key tile's score MFMAs inside the current tile's softmax / P V block, K one tile ahead of V, both
    by LDS-DMA; hd 128 default) performs v2's arithmetic in v2's order -- every selectable form (ARIA_ATTN_FWD: "2" = v2; "3" = v3, for
    hd 72 with 12 waves and the scores in step; "3p" = hd 72 pipelined with 8 waves) gives the same bits, and those bits are within the
    attention tolerance of the fp32 eager oracle (modeling_llama.py:192-215; key padding vision_encoder.py:147-152)."""
    import os

    from aria_amd import hip, ops

    D = H * hd
    g = torch.Generator().manual_seed(B * 1000 + Sq + Skv + hd)
    q = torch.randn(B * Sq, D, generator=g).to(bf16)
    kv = torch.randn(B * Skv, 2 * D, generator=g).to(bf16)
    km = kl = None
    if masked:
        km = (torch.rand(B, Skv, generator=g) > 0.3).to(torch.uint8)
        km[:, 0] = 1
        if Skv > 130:
            km[0, 64:128] = 0          # a whole tile without a valid key: v2 skips it, v3 works through it (exp2(-inf) = 0)
        if Skv > 200:
            km[0, Skv - Skv // 4:] = 0
    if use_len:
        kl = torch.randint(1, Skv + 1, (B,), generator=g).to(torch.int32)
    want_variant = {"2": 2, "3": 3 if hd == 128 else 2}   # (r05: hd 72 has only v2 in the library; "3" there selects nothing)
    vers = ("2", "3")
    prev = os.environ.get("ARIA_ATTN_FWD")
    got = {}
    try:
        for ver in vers:
            os.environ["ARIA_ATTN_FWD"] = ver
            o, lse = ops.attention_fwd(q.to(dev), kv[:, :D].to(dev), kv[:, D:].to(dev), B, Sq, H, hd, hd ** -0.5, causal,
                                       kv_len=None if kl is None else kl.to(dev), key_mask=None if km is None else km.to(dev), Skv=Skv)
            assert hip.get_lib().cdll.aria_last_attn_fwd_variant() == want_variant[ver]
            got[ver] = (o.cpu(), lse.cpu())
        os.environ.pop("ARIA_ATTN_FWD")
        ops.attention_fwd(q.to(dev), kv[:, :D].to(dev), kv[:, D:].to(dev), B, Sq, H, hd, hd ** -0.5, causal, Skv=Skv)
        assert hip.get_lib().cdll.aria_last_attn_fwd_variant() == (3 if hd == 128 else 2)    # the defaults
    finally:
        if prev is None:
            os.environ.pop("ARIA_ATTN_FWD", None)
        else:
            os.environ["ARIA_ATTN_FWD"] = prev
    for ver in vers[1:]:
        assert torch.equal(got["2"][0], got[ver][0]) and torch.equal(got["2"][1], got[ver][1]), (ver, float((got["2"][0].float() - got[ver][0].float()).abs().max()))
    pad = None
    if km is not None or kl is not None:
        pad = torch.zeros(B, Skv, dtype=torch.bool)
        if km is not None:
            pad |= km == 0
        if kl is not None:
            pad |= torch.arange(Skv)[None, :] >= kl[:, None].long()
    qq = q.float().view(B, Sq, H, hd).transpose(1, 2)
    kk, vv = (kv[:, i * D:(i + 1) * D].float().view(B, Skv, H, hd).transpose(1, 2) for i in range(2))
    want = O.attention_eager(qq, kk, vv, hd ** -0.5, causal, key_padding=pad)
    close(got["3"][0], want.transpose(1, 2).reshape(B * Sq, D), 2e-2, 2e-2)


def case_attention_bwd_rope(dev, B, S, H, causal, use_len, s_rope=None):
    """r05: ``aria_attn_bwd_rope`` -- the inverse half-split RoPE of dq / dk (the chain rule through apply_rotary_pos_emb,
    modeling_llama.py:130-160) in the attention backward's register epilogues -- gives the bits of ``aria_attn_bwd`` followed by the in-place
    inverse pass (``rope_(.., inverse=True)``), dv untouched; and rotating back forward returns the un-fused gradients' neighbourhood."""
    from aria_amd import functional as Fn
    from aria_amd import ops

    hd = 128
    D = H * hd
    g = torch.Generator().manual_seed(B * 100 + S + H)
    qkv = torch.randn(B * S, 3 * D, generator=g).to(bf16).to(dev)
    do = torch.randn(B * S, D, generator=g).to(bf16).to(dev)
    kl = torch.randint(1, S + 1, (B,), generator=g).to(torch.int32).to(dev) if use_len else None
    cos, sin = Fn.rope_tables(s_rope or S, hd, 1e4, dev)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    o, lse = ops.attention_fwd(q, k, v, B, S, H, hd, hd ** -0.5, causal, kv_len=kl)
    ref = torch.empty(B * S, 3 * D, dtype=bf16, device=dev)
    ops.attention_bwd(q, k, v, o, do, lse, B, S, H, hd, hd ** -0.5, causal, kv_len=kl, dq=ref[:, :D], dk=ref[:, D:2 * D], dv=ref[:, 2 * D:])
    plain = ref.clone()
    ops.rope_(ref[:, :2 * D], cos, sin, S, 2 * H, hd, inverse=True)
    got = torch.full((B * S, 3 * D), float("nan"), dtype=bf16, device=dev)
    ops.attention_bwd(q, k, v, o, do, lse, B, S, H, hd, hd ** -0.5, causal, kv_len=kl, dq=got[:, :D], dk=got[:, D:2 * D], dv=got[:, 2 * D:],
                      rope=(cos, sin))
    assert torch.equal(got.cpu(), ref.cpu()), float((got.float() - ref.float()).abs().max())
    assert S == 1 or not torch.equal(got[:, :2 * D].cpu(), plain[:, :2 * D].cpu())       # (the rotation did happen; position 0 alone is the identity)
    assert ops.attention_bwd_rope_fusable(hd, S, cos) and not ops.attention_bwd_rope_fusable(64, S, cos)
    with pytest.raises(Exception):
        ops.attention_bwd(q[:, :H * 64], k[:, :H * 64], v[:, :H * 64], o[:, :H * 64], do[:, :H * 64], lse, B, S, H, 64, 0.125, causal,
                          rope=(cos[:, :64].contiguous(), sin[:, :64].contiguous()))


def case_scale_by_device_scalar(dev, n):
    """``ops.scale_`` (aria_scale_bf16): x *= s with s ONE fp32 on the device -- what the fused loss nodes do with autograd's upstream gradient
    (modeling_aria.py:301-323's loss under ``loss.backward()`` / ``(loss / k).backward()``); torch's own ``x.mul_(s)`` is the reference, bit
    for bit (bf16 -> fp32, multiply, round), and s == 1 leaves every bit alone."""
    from aria_amd import ops

    x = rnd(n, seed=n)
    x[0], x[-1] = float("inf"), -0.0
    for sv in (1.0, 0.25, 1.0 / 3.0, -2.5, 0.0):
        s = torch.tensor(sv, dtype=torch.float32)
        want = x.clone().mul_(s)
        got = ops.scale_(x.clone().to(dev), s.to(dev))
        got = got.cpu()
        assert torch.equal(got.isnan(), want.isnan()), sv          # (inf x 0: a NaN either way; its payload is the platform's)
        assert torch.equal(got.nan_to_num(0.0).view(torch.int16), want.nan_to_num(0.0).view(torch.int16)), sv
    odd = rnd(12, seed=1)                      # n % 8 != 0: torch's path
    assert torch.equal(ops.scale_(odd.clone().to(dev), torch.tensor(0.5).to(dev)).cpu(), odd * 0.5)


# ------------------------------------------------------------------------------------------ single-query decode attention
def case_decode_attention(dev, H, hd, pos, splits, S_max=None):
    """aria_decode_attn (RoPE of q / k with freqs_cis[pos], cache write at row pos, softmax over rows 0..pos) against fp32 torch on the same
    bf16 inputs with the kernel's rounding points (rotated q / k and P rounded to bf16); the split form against the single-workgroup form."""
    from aria_amd import ops

    S_max = S_max or pos + 3
    D = H * hd
    qkv = rnd(3 * D, seed=1)
    kc, vc = rnd(S_max, D, seed=2), rnd(S_max, D, seed=3)
    g = torch.Generator().manual_seed(4)
    ang = torch.rand(S_max, hd // 2, generator=g) * 6.28
    fc = torch.stack([ang.cos(), ang.sin()], dim=-1).to(bf16)  # [S_max, hd/2, 2] like precompute_freqs_cis (gptfast/model.py:500-516)

    def rope(x):  # x [H*hd] interleaved pairs, fp32 math, one rounding (gptfast/model.py:519-531)
        xp = x.float().view(H, hd // 2, 2)
        c, s_ = fc[pos, :, 0].float()[None], fc[pos, :, 1].float()[None]
        return torch.stack([xp[..., 0] * c - xp[..., 1] * s_, xp[..., 1] * c + xp[..., 0] * s_], dim=-1).reshape(D).to(bf16)

    q_r, k_r = rope(qkv[:D]), rope(qkv[D:2 * D])
    kc_want, vc_want = kc.clone(), vc.clone()
    kc_want[pos], vc_want[pos] = k_r, qkv[2 * D:]
    n = pos + 1
    sc = torch.einsum("hd,nhd->hn", q_r.float().view(H, hd), kc_want[:n].float().view(n, H, hd)) * hd ** -0.5
    p = torch.softmax(sc, dim=-1)
    want = torch.einsum("hn,nhd->hd", p, vc_want[:n].float().view(n, H, hd)).reshape(D)

    outs = {}
    for ns in sorted({1, splits}):
        q_dev, k_dev, v_dev = qkv.clone().to(dev), kc.clone().to(dev), vc.clone().to(dev)
        out = ops.decode_attention(q_dev, fc.to(dev), torch.tensor([pos], dtype=torch.int32, device=dev), k_dev, v_dev, H, hd, splits=ns)
        assert torch.equal(k_dev.cpu(), kc_want) and torch.equal(v_dev.cpu(), vc_want), "cache row / untouched rows"
        assert torch.equal(q_dev.cpu(), qkv), "qkv is read-only"
        close(out, want, 2e-2, 2e-2)
        outs[ns] = out.float().cpu()
    if splits > 1:  # same per-key arithmetic; only the merge order of partial states differs
        close(outs[splits], outs[1], 1e-2, 4e-3)


def case_router_fused(dev, T, D, E, k):
    """K1: TopKRouter.forward (moe_lm.py:190-201, 243-293) as ONE launch == the gating GEMM followed by aria_moe_route, bit for bit: logits,
    scores, indices, histogram -- on token counts that are not multiples of the wave's 32 rows, with forced ties in the weights (duplicated
    expert rows: the tie rule "lowest expert id" decides) -- and the routing against the oracle's own top-k on the SAME bf16 logits."""
    from aria_amd import ops

    x = rnd(T, D, seed=301).to(dev)
    w = rnd(E, D, seed=302, scale=0.05)
    w[5] = w[3]                      # exact ties between experts 3 and 5 (and 17 / 40 when they exist) on every token
    if E > 40:
        w[40] = w[17]
    w = w.to(dev)
    assert ops.router_fusable(D, E, k)
    logits, scores, idx, counts = ops.moe_router_fused(x, w, k)
    ops.GEMM_SPLIT_K = False
    try:
        l_ref = ops.gemm(x, w)
    finally:
        ops.GEMM_SPLIT_K = True
    s_ref, i_ref, c_ref = ops.moe_route(l_ref, k)
    assert torch.equal(logits.cpu(), l_ref.cpu()), "fused router logits != gemm(x, w)"
    assert torch.equal(idx.cpu(), i_ref.cpu()) and torch.equal(scores.cpu(), s_ref.cpu()) and torch.equal(counts.cpu(), c_ref.cpu())
    want_scores, want_idx, want_tpe = O.router_routing(logits.float().cpu(), k, E)
    assert torch.equal(idx.cpu().long(), want_idx) and torch.equal(counts.cpu().long(), want_tpe)
    close(scores, want_scores.to(bf16), 1e-2, 1e-3)
    assert int(counts.sum()) == T * k


def case_gemm_qkv_rope_hf(dev, B, S, D, hd, K):
    """LlamaAttention's q | k | v projection + half-split RoPE in one launch (gemm3_kernel<false, false, 10>) == gemm + rope_ (the stand-alone
    kernel), bit for bit; v columns untouched; the rotation itself against the oracle's apply_rope_half on the bf16-rounded product."""
    import os

    from aria_amd import functional as Fn
    from aria_amd import hip, ops

    T = B * S
    x = rnd(T, K, seed=311, scale=0.5).to(dev)
    w = rnd(3 * D, K, seed=312, scale=0.1).to(dev)
    cos, sin = Fn.rope_tables(S + 3, hd, 5e6, dev)
    got = ops.gemm_qkv_rope(x, w, cos, sin, S, hd)
    fused = hip.get_lib().cdll.aria_last_gemm_variant() == 3
    os.environ["ARIA_FUSE_QKV_ROPE"] = "0"
    ops.GEMM_SPLIT_K = False
    try:
        ref = ops.gemm_qkv_rope(x, w, cos, sin, S, hd)
        plain = ops.gemm(x, w)
    finally:
        os.environ.pop("ARIA_FUSE_QKV_ROPE", None)
        ops.GEMM_SPLIT_K = True
    assert torch.equal(got.cpu(), ref.cpu()), float((got.float() - ref.float()).abs().max())
    assert torch.equal(got[:, 2 * D:].cpu(), plain[:, 2 * D:].cpu())
    H = D // hd
    q = plain[:, :D].float().cpu().view(B, S, H, hd).transpose(1, 2)
    kk = plain[:, D:2 * D].float().cpu().view(B, S, H, hd).transpose(1, 2)
    co, si = cos[:S].float().cpu()[None], sin[:S].float().cpu()[None]
    qo, ko = O.apply_rope_half(q, kk, co, si)
    # (three bf16 roundings -- product, x cos, x' sin -- against the oracle's fp32 rotation of the rounded product: 2^-8 of |x| ~ 10 per term)
    close(got[:, :D], qo.transpose(1, 2).reshape(T, D).to(bf16), 2e-2, 1e-1)
    close(got[:, D:2 * D], ko.transpose(1, 2).reshape(T, D).to(bf16), 2e-2, 1e-1)
    return fused


def case_grouped_gemm_wgrad_gather(dev, T, E, k, K, N, seed=321):
    """The weight gradient of experts.fc1 through the dispatcher's index (aria_grouped_gemm_wgrad_gather_bf16: the reduction rows of a K-tile
    are token rows reached through indices that travel by LDS-DMA beside the operand pieces) == aria_moe_permute + aria_grouped_gemm_wgrad_bf16, bit for bit (same tiles, same reduction
    order), bf16 and fp32 outputs, on a real routing (ragged and empty experts: expert 1 gets no token, the last one ends mid-tile)."""
    from aria_amd import hip, ops

    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(T, K, generator=g) * 0.5).to(bf16).to(dev)
    logits = torch.randn(T, E, generator=g)
    logits[:, 1] = -1e9                                        # nobody routes to expert 1
    idx = torch.topk(logits, k, dim=1).indices.to(torch.int32).to(dev)
    counts = torch.bincount(idx.flatten().long(), minlength=E).to(torch.int32)
    offsets, sorted_src, inv = ops.moe_sort(idx, counts)
    M = T * k
    dy = (torch.randn(M, N, generator=g) * 0.5).to(bf16).to(dev)
    perm = ops.moe_permute(x, sorted_src, k)
    rows = ops.permuted_token_rows(sorted_src, k)
    assert torch.equal(perm.cpu(), x.cpu()[rows.long().cpu()])
    for dt in (bf16, torch.float32):
        ref = ops.grouped_gemm_wgrad(perm, dy, offsets, E, out_dtype=dt)
        assert hip.get_lib().cdll.aria_last_gemm_variant() == 3, "the case must run the v3 weight gradient (ARIA_GEMM_FORCE=3 at toy sizes)"
        got = ops.grouped_gemm_wgrad_gather(x, rows, dy, offsets, E, out_dtype=dt)
        assert got is not None and torch.equal(got.cpu(), ref.cpu()), float((got.float() - ref.float()).abs().max())
    assert float(got[1].abs().max()) == 0.0
    want = torch.zeros(E, K, N)
    off = offsets.cpu().tolist()
    for e in range(E):
        want[e] = perm[off[e]:off[e + 1]].float().cpu().t() @ dy[off[e]:off[e + 1]].float().cpu()
    close(got, want, 1e-4, 1e-3 * max(1, max(counts.tolist())) ** 0.5)


def case_adamw_values(dev):
    """aria_adamw_step (the kernel the DP optimizer shards run, parallel.py ShardedAdamW; HF Trainer's AdamW of recipes/config_full.yaml:25-29:
    betas (0.9, 0.95), eps 1e-8, decoupled weight decay on matrices only) VALUE-checked against fp32 AdamW: (a) the kernel alone on a flat
    tensor, three steps, with and without a gradient scale, master weights to fp32 rounding and the bf16 parameter == bf16(master) bit for
    bit; (b) ShardedAdamW at world 1 on odd-sized tensors in both decay groups, every element against the closed-form update.
    (VERDICT r5 weak #1b: the gfx950 build of this kernel had no value check on hardware -- tests/test_gpu_ep.py only saw weights change.)"""
    from aria_amd import ops
    from aria_amd.parallel import ShardedAdamW

    g0 = torch.Generator().manual_seed(5)
    for n, scale in ((1000, 1.0), ((1 << 20) + 2, 0.25)):
        p = torch.randn(n, generator=g0).to(bf16).to(dev)
        master, m, v = p.float().clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        ref = torch.nn.Parameter(p.float().cpu().clone())
        opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
        for step in range(1, 4):
            g = torch.randn(n, generator=g0).to(bf16)
            ops.adamw_step_(p, g.to(dev), master, m, v, lr=1e-2, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.1, step=step, grad_scale=scale)
            ref.grad = g.float() * scale
            opt.step()
            torch.testing.assert_close(master.cpu(), ref.detach(), atol=1e-5, rtol=1e-5)
            assert torch.equal(p.cpu(), master.to(bf16).cpu())
    torch.manual_seed(1)
    params = {"layer.weight": torch.randn(7, 3).bfloat16(), "layer.bias": torch.randn(7).bfloat16(), "norm.weight": torch.randn(5).bfloat16()}
    params = {k: torch.nn.Parameter(v.to(dev)) for k, v in params.items()}
    ref = {k: v.detach().float().cpu().clone() for k, v in params.items()}
    mm = {k: torch.zeros_like(v) for k, v in ref.items()}
    vv = {k: torch.zeros_like(v) for k, v in ref.items()}
    opt = ShardedAdamW(list(params.items()), lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    assert opt.decay == [0.1, 0.0, 0.0]
    for step in range(1, 4):
        for k, p in params.items():
            p.grad = torch.randn(p.shape, generator=torch.Generator().manual_seed(step * 7 + len(k))).bfloat16().to(dev)
        opt.step()
        for k, p in params.items():
            g = p.grad.float().cpu()
            mm[k] = 0.9 * mm[k] + 0.1 * g
            vv[k] = 0.95 * vv[k] + 0.05 * g * g
            wd = 0.1 if k == "layer.weight" else 0.0
            ref[k] = ref[k] - 1e-2 * ((mm[k] / (1 - 0.9 ** step)) / ((vv[k] / (1 - 0.95 ** step)).sqrt() + 1e-8) + wd * ref[k])
    for i, (k, p) in enumerate(params.items()):
        torch.testing.assert_close(opt.state[i]["master"].cpu(), ref[k].reshape(-1), rtol=2e-5, atol=1e-6)
        assert torch.equal(p.detach().cpu().reshape(-1), opt.state[i]["master"].to(bf16).cpu())


def case_grouped_tile_orders(dev, T=900, E=5, k=2, K=128, I=128, seed=77):
    """The grouped-row launches' tile ORDER (ARIA_GEMM_ORDER: expert-major eighths per XCD, ragged-last = bit 9, ragged-first = bit 11, the
    r06 per-expert interleave = bit 12, ragged row tiles on the steady K loop = bit 13, the diagnostic rotation = bits 16-18, and combinations) only re-assigns tiles to workgroup ids: every order must cover every tile
    exactly once -- same bits as the default order for the plain grouped GEMM, the fused fc1 + SwiGLU launch and its gathered form, on a
    routing with an empty expert, experts of less than one tile and of several tiles with a ragged last one."""
    import os

    from aria_amd import ops

    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(T, K, generator=g) * 0.5).to(bf16).to(dev)
    logits = torch.randn(T, E, generator=g)
    logits[:, 1] = -1e9                     # expert 1 gets nothing
    logits[: T // 2, 0] += 3.0              # expert 0 gets many rows (several row tiles), the others few
    idx = torch.topk(logits, k, dim=1).indices.to(torch.int32).to(dev)
    counts = torch.bincount(idx.flatten().long(), minlength=E).to(torch.int32)
    offsets, sorted_src, inv = ops.moe_sort(idx, counts)
    perm = ops.moe_permute(x, sorted_src, k)
    rows = ops.permuted_token_rows(sorted_src, k)
    w1 = (torch.randn(E, K, 2 * I, generator=g) * 0.05).to(bf16).to(dev)
    prev = os.environ.get("ARIA_GEMM_ORDER")
    outs = {}
    try:
        for order in (None, 4, 4 | 512, 4 | 512 | 2048, 4 | 4096, 4 | 512 | 4096, 4 | 512 | 2048 | 4096, 4 | 8192, 4 | 512 | 8192, 4 | (3 << 16)):
            if order is None:
                os.environ.pop("ARIA_GEMM_ORDER", None)
            else:
                os.environ["ARIA_GEMM_ORDER"] = str(order)
            plain = ops.grouped_gemm(perm, w1, offsets)
            h, act = ops.grouped_gemm_swiglu(perm, w1, offsets, want_h=True)
            hg, actg = ops.grouped_gemm_swiglu_gather(x, rows, w1, offsets, want_h=True)
            outs[order] = [t.cpu().clone() for t in (plain, h, act, hg, actg)]
    finally:
        if prev is None:
            os.environ.pop("ARIA_GEMM_ORDER", None)
        else:
            os.environ["ARIA_GEMM_ORDER"] = prev
    ref = outs[None]
    assert float(ref[0].float().abs().max()) > 0
    for order, o in outs.items():
        for a, b in zip(o, ref):
            assert torch.equal(a, b), order
