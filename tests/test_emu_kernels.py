"""CPU-side parity of the HIP kernels' *source* through the SIMT emulator (tests/emu): the same .hip files
compiled for the host.  Checks index arithmetic, masking and barrier placement against the oracle at small
sizes; the real parity tests (-m gpu) run the gfx950 build through the same Python wrappers."""
import pytest
import torch

from oracle import aria_oracle as O

bf16 = torch.bfloat16


@pytest.fixture(scope="module", autouse=True)
def emu():
    from tests.emu import emu_lib

    emu_lib.install()
    yield
    emu_lib.uninstall()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(bf16)


def assert_close(got, want, rtol=2e-2, atol=2e-2):
    got, want = got.float(), want.float()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    assert bool((err <= tol).all()), f"max err {err.max().item():.4g} (tol {tol.max().item():.3g}) at {err.argmax().item()}"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 72), (1, 8, 8), (130, 264, 200)])
@pytest.mark.parametrize("a_oc,b_oc", [(False, False), (False, True), (True, True)])
def test_gemm_layouts(M, N, K, a_oc, b_oc):
    from aria_amd import ops

    if a_oc and M % 8:
        M = (M + 7) // 8 * 8
    A = rnd(M, K, seed=1)
    B = rnd(K, N, seed=2)  # logical [K,N]
    a_arg = A.t().contiguous() if a_oc else A
    b_arg = B if b_oc else B.t().contiguous()
    bias = rnd(N, seed=3)
    want = A.float() @ B.float()
    got = ops.gemm(a_arg, b_arg, a_oc=a_oc, b_oc=b_oc)
    assert_close(got, want.to(bf16), 1e-2, 1e-2 * K ** 0.5)
    got32 = ops.gemm(a_arg, b_arg, a_oc=a_oc, b_oc=b_oc, out_dtype=torch.float32)
    assert_close(got32, want, 1e-4, 1e-3)
    gotb = ops.gemm(a_arg, b_arg, a_oc=a_oc, b_oc=b_oc, bias=bias, out_dtype=torch.float32)
    assert_close(gotb, want + bias.float(), 1e-4, 1e-3)
    acc = torch.ones(M, N, dtype=torch.float32)
    ops.gemm(a_arg, b_arg, a_oc=a_oc, b_oc=b_oc, out=acc, accumulate=True)
    assert_close(acc, want + 1.0, 1e-4, 1e-3)


def test_gemm_strided_output_and_input_views():
    from aria_amd import ops

    T, D = 40, 64
    x = rnd(T, D, seed=4)
    w = rnd(D, D, seed=5, scale=0.2)
    qkv = torch.zeros(T, 3 * D, dtype=bf16)
    ops.gemm(x, w, out=qkv[:, D:2 * D])
    assert_close(qkv[:, D:2 * D], (x.float() @ w.float().t()).to(bf16), 1e-2, 5e-2)
    assert float(qkv[:, :D].abs().max()) == 0 and float(qkv[:, 2 * D:].abs().max()) == 0
    y = ops.gemm(qkv[:, D:2 * D], w)
    assert_close(y, (qkv[:, D:2 * D].float() @ w.float().t()).to(bf16), 1e-2, 5e-2)


@pytest.mark.parametrize("counts", [[3, 0, 130, 5, 0, 0, 70, 1], [0, 0, 0, 0], [256], [1, 1, 1]])
def test_grouped_gemm_fwd_dgrad_wgrad(counts):
    from aria_amd import ops

    E, K, N = len(counts), 72, 136
    M = sum(counts)
    a = rnd(M, K, seed=6)
    w = rnd(E, K, N, seed=7, scale=0.3)
    tpe = torch.tensor(counts)
    off = torch.zeros(E + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(tpe, 0)
    want = O.sequential_gemm(a.float(), w.float(), tpe)
    if M:
        got = ops.grouped_gemm(a, w, off)
        assert_close(got, want.to(bf16), 1e-2, 0.1)
        dy = rnd(M, N, seed=8)
        da = ops.grouped_gemm(dy, w, off, w_is_kn=False)
        want_da = O.sequential_gemm(dy.float(), w.float().transpose(1, 2), tpe)
        assert_close(da, want_da.to(bf16), 1e-2, 0.1)
    else:
        dy = rnd(0, N)
    dw = ops.grouped_gemm_wgrad(a, dy, off, E, out_dtype=torch.float32)
    want_dw = torch.zeros(E, K, N)
    s = 0
    for e, n in enumerate(counts):
        want_dw[e] = a[s:s + n].float().t() @ dy[s:s + n].float()
        s += n
    assert_close(dw, want_dw, 1e-4, 1e-3)


@pytest.mark.parametrize("T,E,k", [(33, 8, 3), (257, 64, 6), (5, 200, 4)])
@pytest.mark.parametrize("dtype", [bf16, torch.float32])
def test_route_bit_exact_with_ties(T, E, k, dtype):
    from aria_amd import ops

    g = torch.Generator().manual_seed(T + E)
    logits = (torch.randn(T, E, generator=g) * 0.5)
    logits[:, 3] = logits[:, 1]  # force ties everywhere
    logits[0] = 0.25              # a fully tied row
    logits = logits.to(dtype)
    scores, idx, counts = ops.moe_route(logits, k)
    ws, wi, wc = O.router_routing(logits, k, E)
    assert torch.equal(idx.long(), wi)
    assert torch.equal(counts.long(), wc)
    if dtype == bf16:
        assert (scores.float() - ws.float()).abs().max() <= 2 ** -8  # 1 bf16 ulp of a value < 1
    else:
        assert torch.allclose(scores, ws, atol=1e-6)


@pytest.mark.parametrize("T,E,k", [(33, 8, 3), (700, 64, 6), (1, 4, 2)])
def test_sort_permute_unpermute_match_reference_order(T, E, k):
    from aria_amd import ops

    D = 72
    g = torch.Generator().manual_seed(T)
    logits = torch.randn(T, E, generator=g).to(bf16)
    scores, idx, counts = ops.moe_route(logits, k)
    off, sorted_src, inv = ops.moe_sort(idx, counts)
    x = rnd(T, D, seed=9)
    want_perm, want_sorted = O.token_permutation(x, idx.long(), k)
    assert torch.equal(sorted_src.long(), want_sorted)  # stable order == reference's argsort(stable=True)
    assert torch.equal(off.long(), torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(counts.long(), 0)]))
    assert torch.equal(inv.long()[want_sorted], torch.arange(T * k))
    perm = ops.moe_permute(x, sorted_src, k)
    assert torch.equal(perm, want_perm)
    eo = rnd(T * k, D, seed=10)
    shared = rnd(T, D, seed=11)
    got = ops.moe_unpermute(eo, inv, scores, k, add=shared)
    want = O.token_unpermutation(eo, scores, want_sorted, k, (T, D)) + shared
    assert torch.equal(got, want)  # bf16 rounding points mirrored exactly
    # plain-sum mode == backward of the gather
    gsum = ops.moe_unpermute(eo, inv, None, k)
    assert_close(gsum, torch.zeros(T * k, D).index_copy_(0, want_sorted, eo.float()).view(T, k, D).sum(1).to(bf16), 1e-2, 1e-2)


def test_moe_backward_pieces_against_autograd():
    from aria_amd import ops

    T, E, k, D = 50, 8, 3, 40
    cfg = O.LMConfig(hidden_size=D, moe_num_experts=E, moe_topk=k, moe_z_loss_coeff=1e-2, moe_aux_loss_coeff=5e-2)
    logits = rnd(T, E, seed=12)
    lf = logits.float().requires_grad_(True)
    # oracle: scores + aux losses in fp32 from the same bf16 logits
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
    scores, idx_k, counts = ops.moe_route(logits, k)
    assert torch.equal(idx_k.long(), idx)
    dl = ops.moe_route_bwd(logits, idx_k, scores, ds, counts, cfg.moe_z_loss_coeff, cfg.moe_aux_loss_coeff, 0.5)
    assert_close(dl, lf.grad, 2e-2, 2e-3)

    # unpermute backward
    off, sorted_src, inv = ops.moe_sort(idx_k, counts)
    eo = rnd(T * k, D, seed=14)
    dout = rnd(T, D, seed=15)
    eof = eo.float().requires_grad_(True)
    sf = scores.float().requires_grad_(True)
    out = O.token_unpermutation(eof, sf, sorted_src.long(), k, (T, D))
    out.backward(dout.float())
    d_eo, dsc = ops.moe_unpermute_bwd(dout, eo, inv, scores, k)
    assert_close(d_eo, eof.grad, 1e-2, 1e-2)
    assert_close(dsc, sf.grad, 2e-2, 5e-2)


def test_swiglu_fwd_bwd():
    from aria_amd import ops

    M, I = 37, 24
    h = rnd(M, 2 * I, seed=16)
    want = O.glu(h)
    assert torch.equal(ops.swiglu(h), want)
    gate, up = h[:, :I].contiguous(), h[:, I:].contiguous()
    assert torch.equal(ops.swiglu(gate, up), want)
    hf = h.float().requires_grad_(True)
    dact = rnd(M, I, seed=17)
    O.glu(hf).backward(dact.float())
    assert_close(ops.swiglu_bwd(h, dact), hf.grad, 2e-2, 2e-2)
    dg, du = ops.swiglu_bwd(gate, dact, up)
    assert_close(torch.cat([dg, du], 1), hf.grad, 2e-2, 2e-2)


@pytest.mark.parametrize("T,D", [(9, 64), (130, 2560), (3, 1152)])
def test_rmsnorm_fwd_bwd(T, D):
    from aria_amd import ops

    x, res, w = rnd(T, D, seed=18), rnd(T, D, seed=19), (1 + 0.1 * torch.randn(D)).to(bf16)
    y, h, rstd = ops.rmsnorm(x, w, 1e-6)
    assert torch.equal(y, O.rms_norm(x, w, 1e-6))
    y2, h2, rstd2 = ops.rmsnorm(x, w, 1e-6, residual=res)
    assert torch.equal(h2, x + res)
    assert torch.equal(y2, O.rms_norm(x + res, w, 1e-6))
    hf = h2.float().requires_grad_(True)
    wf = w.float().requires_grad_(True)
    dy, dres = rnd(T, D, seed=20), rnd(T, D, seed=21)
    yf = O.rms_norm(hf, wf, 1e-6)
    (yf * dy.float()).sum().backward()
    dx, dw = ops.rmsnorm_bwd(dy, h2, w, rstd2, dres=dres)
    assert_close(dx, hf.grad + dres.float(), 2e-2, 2e-2)
    assert_close(dw, wf.grad, 2e-2, 2e-2 * T ** 0.5)


def test_rope_matches_hf_half_split_and_inverse():
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
    work = qkv.clone()
    ops.rope_(work[:, :2 * D], cos, sin, S, 2 * H, hd)
    assert torch.equal(work[:, :D].view(B, S, H, hd).transpose(1, 2), wq)
    assert torch.equal(work[:, D:2 * D].view(B, S, H, hd).transpose(1, 2), wk)
    assert torch.equal(work[:, 2 * D:], qkv[:, 2 * D:])
    ops.rope_(work[:, :2 * D], cos, sin, S, 2 * H, hd, inverse=True)
    assert_close(work, qkv, 2e-2, 2e-2)


def test_add():
    from aria_amd import ops

    a, b = rnd(5, 16, seed=23), rnd(5, 16, seed=24)
    assert torch.equal(ops.add(a, b), a + b)
