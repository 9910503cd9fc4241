"""Randomised parity sweep on the GPU: random shapes across the kernel-family boundaries, every
result (forward, dX, dCore) against the float64 CPU oracle.  Seeds are fixed, so failures repeat."""
import random

import pytest
import torch

import dctn_amd
from dctn_amd.conv_sbs import ConvSBS, DumbNormalInitialization, matrix_core_sweep
from dctn_amd.conv_sbs_spec import SBSSpecCore, SBSSpecString
from dctn_amd.eps import eps, keep_gemm_result
from dctn_amd.logmatmulexp import logmatmulexp, logmatmulexp_batched, logmatmulexp_fold
from dctn_amd.pos2d import Pos2D
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL = {torch.float64: (1e-9, 1e-11), torch.float32: (3e-4, 3e-5), torch.bfloat16: (3e-2, 3e-2)}


def check(got, want, dtype, what):
    want = want.double()
    got = got.detach().cpu().double()
    rtol, atol = TOL[dtype]
    scale = float(want.abs().max()) or 1.0
    err = float((got - want).abs().max())
    bound = atol * scale + rtol * scale
    assert err <= bound, f"{what}: err {err:.3e} > {bound:.3e} (scale {scale:.3e})"


def eps_cases():
    rng = random.Random(2026)
    cases = []
    while len(cases) < 28:
        C = rng.choice([1, 1, 1, 2, 3])
        K = rng.choice([1, 2, 2, 3, 3, 4])
        Q = rng.choice([2, 2, 2, 3, 4, 4, 5, 8])
        N = K * K * C
        if Q**N > 2**18 or N > 18:
            continue
        O = rng.choice([1, 2, 3, 4, 5, 6, 8, 10, 16, 20])
        if Q**N * O > 2**21:
            continue
        B = rng.choice([1, 2, 3, 5, 17])
        H, W = K + rng.randrange(0, 9), K + rng.randrange(0, 9)
        if B * (H - K + 1) * (W - K + 1) * Q**N * O > 3e9:
            continue
        dtype = rng.choice([torch.float32, torch.float32, torch.float64, torch.bfloat16])
        cases.append((C, K, Q, O, B, H, W, dtype, rng.random() < 0.3))
    return cases


@pytest.mark.parametrize("case", eps_cases(), ids=lambda c: "C%dK%dQ%dO%dB%d_%dx%d_%s%s" % (
    c[0], c[1], c[2], c[3], c[4], c[5], c[6], str(c[7]).split(".")[1], "_strided" if c[8] else ""))
def test_eps_random(case):
    C, K, Q, O, B, H, W, dtype, strided = case
    torch.manual_seed(hash(case[:7]) % (2**31))
    N = K * K * C
    x = torch.randn(C, B, H, W, Q).to(dtype)
    core = (torch.randn(*(Q,) * N, O) * Q ** (-N / 4)).to(dtype)
    xd = x.to(DEV)
    if strided:  # same values behind a non-contiguous view
        xd = xd.permute(0, 1, 3, 2, 4).contiguous().permute(0, 1, 3, 2, 4)
    xd = xd.requires_grad_(True)
    cd = core.to(DEV).requires_grad_(True)
    y = eps(cd, xd)
    want = R.eps_4step(core.double(), x.double())
    check(y, want, dtype, f"forward [{dctn_amd.last_kernel()}]")
    dy = torch.randn(*want.shape).to(dtype)
    y.backward(dy.to(DEV))
    dcore, dx = R.grads(R.eps_4step, [core.double(), x.double()], dy.double())
    check(xd.grad, dx, dtype, "dX")
    check(cd.grad, dcore, dtype, "dCore")


def sbs_cases():
    rng = random.Random(77)
    cases = []
    for _ in range(16):
        kh, kw = rng.choice([(2, 2), (3, 3), (1, 3), (2, 3)])
        pos = [(h, w) for h in range(kh) for w in range(kw)]
        rng.shuffle(pos)
        n = rng.randrange(2, len(pos) + 1)
        pos = pos[:n]
        mh, mw = min(p[0] for p in pos), min(p[1] for p in pos)
        pos = [(h - mh, w - mw) for h, w in pos]
        ring = rng.random() < 0.35
        uniform = rng.random() < 0.5
        r = rng.choice([2, 3, 4, 8])
        bonds = [(r if uniform else rng.randrange(1, 6)) for _ in range(n)]
        if not ring:
            bonds[0] = 1
        outs = [1] * n
        for _ in range(rng.choice([0, 1, 1, 2])):
            outs[rng.randrange(n)] = rng.choice([2, 3])
        C, q = rng.choice([(1, 2), (1, 3), (2, 2), (1, 4)])
        dtype = rng.choice([torch.float32, torch.float32, torch.float64])
        cases.append((tuple(pos), tuple(bonds), tuple(outs), C, q, dtype))
    return cases


@pytest.mark.parametrize("case", sbs_cases(), ids=lambda c: "n%d_b%s_o%s_C%dq%d_%s" % (
    len(c[0]), "".join(map(str, c[1])), "".join(map(str, c[2])), c[3], c[4], str(c[5]).split(".")[1]))
def test_convsbs_random(case):
    pos, bonds, outs, C, q, dtype = case
    torch.manual_seed(len(pos) * 100 + sum(bonds))
    spec = SBSSpecString(tuple(SBSSpecCore(Pos2D(h, w), o) for (h, w), o in zip(pos, outs)), bonds, C, q)
    m = ConvSBS(spec, DumbNormalInitialization(0.6)).to(dtype).to(DEV)
    B, H, W = 3, spec.max_height_pos + 4, spec.max_width_pos + 5
    x = torch.randn(C, B, H, W, q, dtype=dtype, device=DEV, requires_grad=True)
    y = m(x)
    cores64 = [c.detach().cpu().double() for c in m.cores]
    want = R.convsbs_forward(cores64, list(pos), x.detach().cpu().double())
    check(y, want, dtype, f"forward [{dctn_amd.last_kernel()}]")
    dy = torch.randn_like(y)
    y.backward(dy)
    gr = R.grads(lambda xx, *cc: R.convsbs_forward(cc, list(pos), xx), [x.detach().cpu().double()] + cores64,
                 dy.cpu().double())
    check(x.grad, gr[0], dtype, "dX")
    for i, (c, gc) in enumerate(zip(m.cores, gr[1:])):
        check(c.grad, gc, dtype, f"dCore{i}")


# (C, K, Q, O, B, H, W, strided): float64 on the f64 matrix cores (two-halves GEMM path): power-of-two and odd Q,
# odd numbers of factors (n0 != n1), O that does not divide the tile, ragged window counts, strided input
F64_CASES = [(1, 3, 2, 4, 5, 9, 8, False), (2, 2, 3, 3, 3, 7, 9, True), (1, 2, 8, 5, 7, 6, 6, False),
             (1, 4, 2, 2, 2, 10, 9, False), (3, 2, 2, 6, 3, 8, 5, True), (1, 3, 3, 1, 2, 8, 9, False)]


@pytest.mark.parametrize("keep", [True, False], ids=["savedz", "recompute"])
@pytest.mark.parametrize("case", F64_CASES, ids=lambda c: "C%dK%dQ%dO%dB%d_%dx%d%s" % (c[:7] + ("_strided" if c[7] else "",)))
def test_eps_f64_matrix_core_path(case, keep):
    """keep: the training forward leaves both Khatri-Rao halves and the GEMM result for the backward
    (`dctn_eps_fwd_save` / `dctn_eps_bwd_saved`), or the backward rebuilds them."""
    with keep_gemm_result(keep):
        _eps_f64_matrix_core_path(case, "_saving" if keep else "", "_savedz" if keep else "")


def _eps_f64_matrix_core_path(case, fsuf, bsuf):
    C, K, Q, O, B, H, W, strided = case
    torch.manual_seed(sum(case[:7]))
    N = K * K * C
    x = torch.randn(C, B, H, W, Q, dtype=torch.float64)
    core = torch.randn(*(Q,) * N, O, dtype=torch.float64) * Q ** (-N / 4)
    xd = x.to(DEV)
    if strided:
        xd = xd.permute(0, 1, 3, 2, 4).contiguous().permute(0, 1, 3, 2, 4)
    xd = xd.requires_grad_(True)
    cd = core.to(DEV).requires_grad_(True)
    y = eps(cd, xd)
    assert dctn_amd.last_kernel() == "eps_fwd_mfma_f64_halves" + fsuf
    want = R.eps_4step(core, x)
    check(y, want, torch.float64, "forward")
    dy = torch.randn(*want.shape, dtype=torch.float64)
    y.backward(dy.to(DEV))
    assert dctn_amd.last_kernel() == "eps_bwd_mfma_f64_halves" + bsuf
    dcore, dx = R.grads(R.eps_4step, [core, x], dy)
    check(xd.grad, dx, torch.float64, "dX")
    check(cd.grad, dcore, torch.float64, "dCore")
    # each gradient alone (other buffers absent)
    x3 = xd.detach().clone().requires_grad_(True)
    eps(cd.detach(), x3).backward(dy.to(DEV))
    check(x3.grad, dx, torch.float64, "dX alone")
    c3 = cd.detach().clone().requires_grad_(True)
    eps(c3, xd.detach()).backward(dy.to(DEV))
    check(c3.grad, dcore, torch.float64, "dCore alone")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32, torch.bfloat16])
def test_eps_halves_path_in_several_window_chunks(dtype, monkeypatch):
    """The two-halves path bounds its per-chunk buffers (1 GiB); with the bound lowered to 256 KiB the 1 000
    windows below run in 16 chunks of 64 windows (dCore accumulated over the chunks, per-chunk columns of the factor
    gradients) — same numbers as the oracle, and as the single-chunk run."""
    C, K, Q, O, B, H, W = 1, 3, 3, 2, 10, 12, 12      # N = 9, Q = 3: halves of 81 and 243 entries, 1 000 windows
    if dtype == torch.bfloat16:                        # bf16 MFMA instantiation: halves must be multiples of 32 / 64
        C, K, Q, O, B, H, W = 1, 3, 4, 3, 3, 10, 11   # N = 9, Q = 4: halves of 256 and 1024 entries, 216 windows, odd O
    torch.manual_seed(77)
    N = K * K * C
    x = torch.randn(C, B, H, W, Q).to(dtype)
    core = (torch.randn(*(Q,) * N, O) * Q ** (-N / 4)).to(dtype)
    dy = torch.randn(B, H - K + 1, W - K + 1, O).to(dtype)

    def run():
        xd, cd = x.to(DEV).requires_grad_(True), core.to(DEV).requires_grad_(True)
        y = eps(cd, xd)
        assert "halves" in dctn_amd.last_kernel()
        y.backward(dy.to(DEV))
        return y.detach(), xd.grad, cd.grad

    from dctn_amd import _lib as L

    whole = run()
    with L.options(L.OPT_SMALL_CHUNKS):   # 256 KiB chunk buffers: the chunk loop runs many times
        chunked = run()
    want = R.eps_4step(core.double(), x.double())
    dcore, dx = R.grads(R.eps_4step, [core.double(), x.double()], dy.double())
    for got in (whole, chunked):
        check(got[0], want, dtype, "forward")
        check(got[1], dx, dtype, "dX")
        check(got[2], dcore, dtype, "dCore")
    tol = {torch.float64: 1e-12, torch.float32: 1e-5, torch.bfloat16: 2e-2}[dtype]
    for a, b in zip(whole, chunked):
        assert float((a.float() - b.float()).abs().max()) <= tol * float(a.float().abs().max())


# float32 shapes the bf16-register and bigcore families leave (odd Q): the two-halves GEMM path on v_mfma_f32_16x16x4_f32
F32_HALVES_CASES = [(2, 2, 3, 3, 3, 7, 9, True), (1, 2, 5, 4, 4, 8, 8, False), (1, 3, 3, 2, 2, 9, 9, False),
                    (1, 2, 6, 8, 5, 7, 6, False)]


@pytest.mark.parametrize("case", F32_HALVES_CASES, ids=lambda c: "C%dK%dQ%dO%dB%d_%dx%d%s" % (c[:7] + ("_strided" if c[7] else "",)))
@pytest.mark.parametrize("keep", [True, False], ids=["savedz", "recompute"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eps_f32_halves_path_for_odd_q(case, dtype, keep):
    with keep_gemm_result(keep):
        _eps_f32_halves_path_for_odd_q(case, dtype, "_saving" if keep else "", "_savedz" if keep else "")


def _eps_f32_halves_path_for_odd_q(case, dtype, fsuf, bsuf):
    C, K, Q, O, B, H, W, strided = case
    torch.manual_seed(sum(case[:7]) + 1)
    N = K * K * C
    x = torch.randn(C, B, H, W, Q).to(dtype)
    core = (torch.randn(*(Q,) * N, O) * Q ** (-N / 4)).to(dtype)
    xd = x.to(DEV)
    if strided:
        xd = xd.permute(0, 1, 3, 2, 4).contiguous().permute(0, 1, 3, 2, 4)
    xd = xd.requires_grad_(True)
    cd = core.to(DEV).requires_grad_(True)
    y = eps(cd, xd)
    assert dctn_amd.last_kernel() == "eps_fwd_mfma_f32_halves" + fsuf and y.dtype == dtype   # bf16: storage only
    want = R.eps_4step(core.double(), x.double())
    check(y, want, dtype, "forward")
    dy = torch.randn(*want.shape).to(dtype)
    y.backward(dy.to(DEV))
    assert dctn_amd.last_kernel() == "eps_bwd_mfma_f32_halves" + bsuf
    dcore, dx = R.grads(R.eps_4step, [core.double(), x.double()], dy.double())
    check(xd.grad, dx, dtype, "dX")
    check(cd.grad, dcore, dtype, "dCore")


@pytest.mark.parametrize("T,Rr,I,dtype", [(1, 1, 1, torch.float64), (7, 3, 5, torch.float32), (33, 65, 17, torch.float64),
                                          (256, 256, 256, torch.float32), (5, 128, 3, torch.float32)])
def test_logmatmulexp_random(T, Rr, I, dtype):
    torch.manual_seed(T * 7 + I)
    a, b = torch.randn(T, Rr, dtype=dtype) * 3, torch.randn(Rr, I, dtype=dtype) * 3
    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = logmatmulexp(ad, bd)
    want = R.logmatmulexp(a.double(), b.double())
    check(y, want, dtype, "forward")
    dy = torch.randn(T, I, dtype=dtype)
    y.backward(dy.to(DEV))
    ga, gb = R.grads(R.logmatmulexp, [a.double(), b.double()], dy.double())
    check(ad.grad, ga, dtype, "dA")
    check(bd.grad, gb, dtype, "dB")


@pytest.mark.parametrize("Wn,L,D,dtype", [(1, 1, 16, torch.float32), (5, 2, 16, torch.float32), (130, 9, 16, torch.float32),
                                          (9, 4, 7, torch.float64), (3, 3, 32, torch.float32), (64, 9, 16, torch.float64)])
def test_logmatmulexp_fold_random(Wn, L, D, dtype):
    torch.manual_seed(Wn + L + D)
    m = torch.randn(Wn, L, D, D, dtype=dtype) * 2
    md = m.to(DEV).requires_grad_(True)
    y = logmatmulexp_fold(md)
    want = R.logmatmulexp_fold_batched(m.double())
    check(y, want, dtype, f"forward [{dctn_amd.last_kernel()}]")
    dy = torch.randn(Wn, D, D, dtype=dtype)
    y.backward(dy.to(DEV))
    (g,) = R.grads(R.logmatmulexp_fold_batched, [m.double()], dy.double())
    check(md.grad, g, dtype, "dMats")


# (C, K, size, B, O, Cout): fused EPS + head backward across its shape family — position groups with
# idle lanes (P < 64, P not a multiple of 64), fewer samples than waves, odd batches, both core
# splits (N = 9: LDS transpose path, N = 8: identity-MFMA path), several class counts
HEAD_CASES = [(1, 3, 10, 1, 4, 10), (1, 3, 12, 7, 2, 4), (2, 2, 9, 13, 4, 16), (1, 3, 12, 70, 4, 2),
              (1, 3, 70, 2, 4, 10), (2, 2, 13, 33, 2, 10), (1, 3, 30, 19, 4, 6), (1, 3, 28, 9, 2, 16),
              (1, 3, 70, 250, 4, 10)]   # 11 samples per wave: two groups of the matrix-core dY / dW products (8 + 3)


@pytest.mark.parametrize("C,K,size,B,O,Cout", HEAD_CASES)
def test_eps_head_fused_backward_random(C, K, size, B, O, Cout):
    from dctn_amd.eps_plus_linear import _EpsLinearHeadFunction

    torch.manual_seed(C * 1000 + K * 100 + size + B + O + Cout)
    N = K * K * C
    side = size - K + 1
    F = side * side * O
    assert F % 8 == 0
    core = (torch.randn(*(2,) * N, O) * 2.0 ** (-N / 2) * 4).to(torch.bfloat16)
    u = torch.rand(C, B, size, size)
    x = torch.stack([torch.sin(u * 1.5707963) ** 2, torch.cos(u * 1.5707963) ** 2], dim=-1).to(torch.bfloat16)
    w = (torch.randn(Cout, F) * F ** -0.5 * 4).to(torch.bfloat16)
    bias = (torch.randn(Cout) * 0.1).to(torch.bfloat16)
    g = torch.randn(B, Cout).to(torch.bfloat16)
    cd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (core, w, bias))
    xd = x.to(DEV)
    assert _EpsLinearHeadFunction.supported(cd, xd, wd, bd)
    out = _EpsLinearHeadFunction.apply(cd, xd, wd, bd)
    out.backward(g.to(DEV))
    assert dctn_amd.last_kernel() == "eps_head_bwd_mfma_q2reg"
    c64, w64, b64 = (t.double().requires_grad_(True) for t in (core, w, bias))
    want = R.eps_plus_linear_forward([c64], w64, b64, x.double())
    check(out, want.detach(), torch.bfloat16, "logits")
    want.backward(g.double())
    check(cd.grad, c64.grad, torch.bfloat16, "dCore")
    check(wd.grad, w64.grad, torch.bfloat16, "dWeight")
    check(bd.grad, b64.grad, torch.bfloat16, "dBias")


# (C, K, H, W, B, O, Cout, strided): the register-resident exact-float32 family (eps_q2f32.hip) across its shape family -
# rectangular images, position groups with idle lanes, fewer steps than waves, more than four samples per workgroup (several
# groups: B > 4 * 256 on a small image), every out size 1..4 (3: zero-padded core rows, element stores; 1: half-empty tiles),
# class counts 1..16, both window modes plus the generic one (strided input), images too large for the one-kernel forward
Q2F32_CASES = [(1, 3, 10, 10, 1, 4, 10, False), (1, 3, 5, 17, 7, 2, 4, False), (2, 2, 9, 6, 13, 4, 16, False), (1, 3, 12, 12, 70, 4, 1, False),
               (1, 3, 7, 9, 1100, 4, 10, False), (2, 2, 13, 5, 33, 2, 10, True), (1, 3, 30, 11, 19, 3, 6, False), (1, 3, 28, 28, 9, 1, 16, False),
               (1, 3, 9, 9, 5, 4, 7, True), (1, 3, 40, 40, 3, 4, 10, False), (2, 2, 4, 4, 3, 3, 5, False), (1, 3, 3, 3, 2, 4, 10, False)]


@pytest.mark.parametrize("C,K,H,W,B,O,Cout,strided", Q2F32_CASES)
def test_eps_float32_register_family_random(C, K, H, W, B, O, Cout, strided):
    from dctn_amd import _lib
    from dctn_amd.eps_plus_linear import _EpsLinearHeadFunction, _LinearHeadFunction

    torch.manual_seed(C * 1000 + K * 100 + H * 7 + W + B + O + Cout)
    N = K * K * C
    F = (H - K + 1) * (W - K + 1) * O
    core = torch.randn(*(2,) * N, O) * 2.0 ** (-N / 2) * 4
    u = torch.rand(C, B, H, W)
    x = torch.stack([torch.sin(u * 1.5707963) ** 2, torch.cos(u * 1.5707963) ** 2], dim=-1)
    if strided:   # (C, B, W, H, 2) permuted back: same values, pixel rows no longer contiguous -> one 4-byte load per feature
        x = x.permute(0, 1, 3, 2, 4).contiguous().permute(0, 1, 3, 2, 4)
    w = torch.randn(Cout, F) * F ** -0.5 * 4
    bias = torch.randn(Cout) * 0.1
    g = torch.randn(B, Cout)
    cd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (core, w, bias))
    xd = x.to(DEV) if not strided else x.permute(0, 1, 3, 2, 4).to(DEV).permute(0, 1, 3, 2, 4)
    assert _lib.lib().dctn_eps_family(C, B, H, W, 2, K, O, _lib.F32, 0) == 4
    fused = _EpsLinearHeadFunction.supported(cd, xd, wd, bd)
    assert fused == (O in (2, 4))
    if fused:
        out = _EpsLinearHeadFunction.apply(cd, xd, wd, bd)
    else:
        feat = eps(cd, xd)
        assert dctn_amd.last_kernel() == "eps_fwd_q2f32"
        out = _LinearHeadFunction.apply(feat.reshape(B, -1), wd, bd)
    out.backward(g.to(DEV))
    if fused:
        assert dctn_amd.last_kernel() == "eps_head_bwd_q2f32"
    c64, w64, b64 = (t.double().requires_grad_(True) for t in (core, w, bias))
    want = R.eps_plus_linear_forward([c64], w64, b64, x.double())
    check(out, want.detach(), torch.float32, "logits")
    want.backward(g.double())
    check(cd.grad, c64.grad, torch.float32, "dCore")
    check(wd.grad, w64.grad, torch.float32, "dWeight")
    check(bd.grad, b64.grad, torch.float32, "dBias")


# (C, B, H, W, Q, K, O): large-core exact-f32 MFMA family with out sizes that are NOT powers of two
# (rows in memory order, o outermost in the transposed GEMMs' k, exact dCore columns), row halves of
# 16 .. 256 entries, one and two input channels
BIGCORE_XO_CASES = [(1, 2, 7, 7, 4, 2, 5), (1, 2, 6, 6, 2, 3, 6), (1, 2, 9, 9, 4, 3, 6), (2, 3, 6, 6, 2, 2, 7),
                    (1, 1, 5, 5, 8, 2, 3), (1, 2, 6, 6, 4, 2, 12), (1, 5, 11, 8, 2, 4, 3)]


@pytest.mark.parametrize("keep", [True, False], ids=["savedz", "recompute"])
@pytest.mark.parametrize("C,B,H,W,Q,K,O", BIGCORE_XO_CASES)
def test_eps_bigcore_exact_out_size(C, B, H, W, Q, K, O, keep):
    with keep_gemm_result(keep):
        _eps_bigcore_exact_out_size(C, B, H, W, Q, K, O, "_saving" if keep else "")


def _eps_bigcore_exact_out_size(C, B, H, W, Q, K, O, fsuf):
    torch.manual_seed(C + 10 * B + 100 * H + Q + K + O)
    N = K * K * C
    core = torch.randn(*(Q,) * N, O) * Q ** (-N / 2) * 3
    x = torch.rand(C, B, H, W, Q) + 0.1
    cd, xd = core.to(DEV).requires_grad_(True), x.to(DEV).requires_grad_(True)
    y = eps(cd, xd)
    assert dctn_amd.last_kernel() == "eps_fwd_mfma_bigcore_f32" + fsuf   # x needs a gradient: the GEMM result can be kept
    want = R.eps_4step(core.double(), x.double())
    check(y, want, torch.float32, "forward")
    dy = torch.randn(y.shape)
    y.backward(dy.to(DEV))
    dcore, dx = R.grads(R.eps_4step, [core.double(), x.double()], dy.double())
    check(cd.grad, dcore, torch.float32, "dCore")
    check(xd.grad, dx, torch.float32, "dX")


@pytest.mark.parametrize("Wn,L", [(7, 3), (33, 5), (20, 6), (9, 10), (5, 16), (4, 17), (257, 9), (9001, 2), (4700, 4), (11, 13), (6, 7), (6, 8), (5, 11),
                                  (9, 12), (5, 14), (7, 15)])
def test_logmatmulexp_fold16_all_chain_lengths(Wn, L):
    """D = 16 float32 fold: chain lengths on both sides of every occupancy step of the per-length kernels (2..16), the
    exact recomputing kernel beyond 16, windows noted / flagged for the exact path in the middle of a batch and in a later
    turn of a persistent wave's loop."""
    torch.manual_seed(Wn * 31 + L)
    m = torch.randn(Wn, L, 16, 16) * 1.5
    if Wn > 4:
        m[2] *= 40.0                                  # range far beyond what the factorisation accepts
        m[Wn - 1, L // 2, 3, :] = -float("inf")       # a -inf row in one factor
    if Wn > 4096:                                     # more windows than resident waves: windows noted in a LATER turn of a wave's loop
        m[4096 + 5] *= 40.0
        m[Wn - 3, 0, :, 2] += 300.0                   # a column the one-shift-per-matrix backward flushes: its second tier takes it
    md = m.to(DEV).requires_grad_(True)
    y = logmatmulexp_fold(md)
    want = R.logmatmulexp_fold_batched(m.double())
    yc = y.detach().cpu().double()
    fin = torch.isfinite(want)
    assert torch.equal(torch.isfinite(yc), fin)
    assert float(((yc[fin] - want[fin]).abs() / (1.0 + want[fin].abs())).max()) < 5e-5
    dy = torch.randn(Wn, 16, 16)
    y.backward(dy.to(DEV))
    assert ("mfma16" in dctn_amd.last_kernel()) == (L <= 16)
    (g,) = R.grads(R.logmatmulexp_fold_batched, [m.double()], dy.double())
    got = md.grad.cpu().double()
    assert torch.isfinite(got).all()
    for w in range(Wn):
        scale = float(g[w].abs().max().clamp_min(1.0))
        assert float((got[w] - g[w]).abs().max()) < 3e-4 * scale, w


@pytest.mark.parametrize("seed", range(6))
def test_logmatmulexp_fold16_dynamic_ranges(seed):
    """D = 16 float32 fold with the prefix carried in scaled form (logmatmulexp.hip): windows of very different dynamic
    range in one batch - spreads the scaled form represents (accepted steps), spreads it flushes (the window restarts in the
    log domain / is left to the exact backward kernel), isolated -inf entries, columns that differ by hundreds of nats -
    forward and gradient against the float64 oracle (dctn/logmatmulexp.py:5-14 folded as the benchmark does, :30)."""
    rng = random.Random(1000 + seed)
    torch.manual_seed(1000 + seed)
    L = rng.choice([2, 4, 7, 9, 12, 16, 19])
    Wn = rng.randrange(3, 70)
    m = torch.randn(Wn, L, 16, 16)
    for w in range(Wn):
        kind = rng.randrange(6)
        if kind == 0:
            m[w] *= rng.choice([5.0, 20.0, 60.0])                      # wide spread everywhere
        elif kind == 1:
            m[w, rng.randrange(L)] *= 80.0                              # one factor with a huge range
        elif kind == 2:
            m[w, rng.randrange(L), :, rng.randrange(16)] += rng.choice([-300.0, 300.0])   # one column far away from the others
        elif kind == 3:
            m[w, rng.randrange(L), rng.randrange(16), rng.randrange(16)] = -float("inf")  # an isolated -inf entry
        elif kind == 4:
            # a common shift: the scaled form carries it in the row shifts (error 3e-7 at any L <= 16); beyond 16 factors the
            # backward is the log-domain recomputing kernel, whose float32 prefixes of magnitude L * shift lose what torch's
            # float32 logsumexp loses (5e-4 at 500), so the shift stays small there
            m[w] += rng.choice([-1.0, 1.0]) * (500.0 if L <= 16 else 20.0)
    md = m.to(DEV).requires_grad_(True)
    y = logmatmulexp_fold(md)
    want = R.logmatmulexp_fold_batched(m.double())
    yc = y.detach().cpu().double()
    fin = torch.isfinite(want)
    assert torch.equal(torch.isfinite(yc), fin)
    assert float(((yc[fin] - want[fin]).abs() / (1.0 + want[fin].abs())).max()) < 5e-5
    dy = torch.randn(Wn, 16, 16)
    y.backward(dy.to(DEV))
    (g,) = R.grads(R.logmatmulexp_fold_batched, [m.double()], dy.double())
    got = md.grad.cpu().double()
    assert torch.isfinite(got).all()
    for w in range(Wn):
        scale = float(g[w].abs().max().clamp_min(1.0))
        assert float((got[w] - g[w]).abs().max()) < 3e-4 * scale, w


def sbs_mfma_cases():
    """Open chains with a uniform bond in {4, 8, 16}: the MFMA sweep family.  String lengths on both sides of the
    9-core specialisation (register accumulators up to 9 cores, LDS accumulators beyond), the one two-output core at
    any position or absent, one or two channels, ragged window counts (not a multiple of the 32-window group)."""
    rng = random.Random(77)
    cases = []
    for n, r, C, q in [(3, 16, 1, 3), (4, 8, 2, 2), (5, 4, 1, 2), (6, 16, 2, 2), (9, 8, 1, 4), (9, 16, 1, 2), (9, 4, 2, 2),
                       (10, 4, 1, 3), (12, 8, 1, 2), (7, 16, 1, 3)]:
        walk, seen = [(0, 0)], {(0, 0)}
        while len(walk) < n:   # a self-avoiding walk on a 4 x 4 grid (any order of positions is legal)
            h, w = walk[-1]
            opts = [(h + dh, w + dw) for dh, dw in ((0, 1), (1, 0), (0, -1), (-1, 0))
                    if 0 <= h + dh < 4 and 0 <= w + dw < 4 and (h + dh, w + dw) not in seen]
            if not opts:
                opts = [(a, b) for a in range(4) for b in range(4) if (a, b) not in seen]
            nxt = rng.choice(opts)
            walk.append(nxt)
            seen.add(nxt)
        mh, mw = min(p[0] for p in walk), min(p[1] for p in walk)
        pos = tuple((h - mh, w - mw) for h, w in walk)
        outs = [1] * n
        if rng.random() < 0.8:
            outs[rng.randrange(1, n - 1)] = 2
        cases.append((pos, (1,) + (r,) * (n - 1), tuple(outs), C, q))
    # one many-valued middle core (the final string of the reference's ConvSBS classifier, mnist.py:214-224: ten labels on
    # core 4 of a snake over two channels): slices of two outputs on the same kernels; an odd count leaves a one-output slice
    snake = ((0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2))
    for r, C, q, where, om in [(4, 2, 2, 4, 10), (16, 1, 2, 4, 5), (8, 2, 2, 2, 3)]:
        outs = [1] * 9
        outs[where] = om
        cases.append((snake, (1,) + (r,) * 8, tuple(outs), C, q))
    # rings (trace_edge=True, mnist.py:229): one open-chain launch per value of the closing bond; with and without a
    # two-valued / many-valued core
    for r, C, q, outs in [(4, 1, 2, (1, 1, 1, 1, 2, 1, 1, 1, 1)), (8, 2, 2, (1,) * 9), (4, 2, 2, (1, 1, 1, 1, 10, 1, 1, 1, 1)),
                          (16, 1, 3, (1, 1, 2, 1, 1))]:
        cases.append((snake[:len(outs)], (r,) * len(outs), tuple(outs), C, q))
    # bonds between the kernels' tile sizes (the reference script's default is 2): zero-padded packs, real entries only
    # in the gradients; open chains, a ring, a many-valued core
    for r, C, q, outs, ring in [(2, 1, 2, (1, 1, 1, 1, 2, 1, 1, 1, 1), False), (3, 2, 2, (1,) * 9, False),
                                (6, 1, 3, (1, 1, 1, 1, 2, 1, 1, 1, 1), True), (12, 2, 2, (1, 1, 1, 1, 5, 1, 1, 1, 1), False),
                                (5, 1, 2, (1, 2, 1, 1), False), (2, 2, 2, (1, 1, 1, 1, 10, 1, 1, 1, 1), True)]:
        cases.append((snake[:len(outs)], ((r if ring else 1),) + (r,) * (len(outs) - 1), tuple(outs), C, q))
    # unequal bonds (every core padded to the tile of the largest), several cores with outputs, outputs on the end cores:
    # slices over views of the cores - the ring of the reference's conversion test, tests/test_conversion_of_convsbs_to_eps.py:25-28
    cases.append((snake, (1, 3, 5, 8, 6, 2, 7, 4, 8), (1, 1, 1, 1, 2, 1, 1, 1, 1), 1, 2))
    cases.append((snake, (1, 16, 12, 16, 9, 16, 16, 10, 16), (1,) * 9, 1, 3))
    cases.append((((0, 0), (0, 1), (1, 1), (1, 0)), (3, 4, 5, 6), (1, 3, 2, 4), 2, 2))
    cases.append((((0, 1), (0, 0), (1, 0), (1, 1)), (3, 4, 5, 6), (4, 2, 1, 3), 2, 2))
    cases.append((snake[:5], (1, 4, 6, 6, 4), (2, 1, 3, 1, 2), 1, 3))
    cases.append((snake[:6], (1, 4, 4, 4, 4, 4), (1, 2, 1, 2, 1, 1), 1, 2))
    return cases


def sbs_reg_family_takes(pos, bonds, outs, C, q):
    """Strings the register-resident small-bond sweep (convsbs_reg.hip) takes by default: float32 open chains of at most
    9 cores, every bond <= 4, at most one two-valued core or exactly one many-valued core, q^C <= 4."""
    prod = 1
    for o in outs:
        prod *= o
    many = [o for o in outs if o != 1]
    shape_ok = len(pos) <= 9 and bonds[0] == 1 and 2 <= max(bonds) <= 4 and ((C == 1 and 2 <= q <= 4) or (C == 2 and q == 2))
    # at most one two-valued core - or exactly one many-valued core (3..16 values: the classifier's ten-label string)
    return shape_ok and ((all(o in (1, 2) for o in outs) and prod <= 2) or (len(many) == 1 and 3 <= many[0] <= 16))


def sbs_band_family_takes(pos, bonds, outs, C, q):
    """Strings whose BACKWARD runs on the band-owning kernels (convsbs_band.hip): float32 open chains of at most 9 cores,
    largest bond 5..16 (two state values per lane up to 8, four above), at most one two-valued core - a middle one -,
    q^C <= 4."""
    prod = 1
    for o in outs:
        prod *= o
    return (3 <= len(pos) <= 9 and bonds[0] == 1 and 4 < max(bonds[1:]) <= 16 and all(o in (1, 2) for o in outs) and prod <= 2
            and outs[0] == 1 and outs[-1] == 1 and ((C == 1 and 2 <= q <= 4) or (C == 2 and q == 2)))


@pytest.mark.parametrize("family", ["default", "matrix_cores"])
@pytest.mark.parametrize("case", sbs_mfma_cases(), ids=lambda c: "n%d_r%s_o%s_C%dq%d" % (
    len(c[0]), c[1][1] if len(set(c[1][1:])) == 1 else "".join("%x" % b for b in c[1]), "".join(map(str, c[2])), c[3], c[4]))
def test_convsbs_mfma_family_random(case, family):
    """family: small-bond strings run on the register-resident sweep by default; `matrix_core_sweep()` sends them to the
    matrix-core sweep, so both families meet the oracle on the same strings."""
    reg = sbs_reg_family_takes(*case)
    small_band = sbs_band_family_takes(*case) and max(case[1]) <= 8
    if family == "matrix_cores":
        if not (reg or small_band):
            pytest.skip("the default already is the matrix-core sweep")
        with matrix_core_sweep():
            _convsbs_family_case(case, "mfma", matrix_cores=True)
    else:
        _convsbs_family_case(case, "reg" if reg else "mfma")


def _convsbs_family_case(case, fam, matrix_cores=False):
    pos, bonds, outs, C, q = case
    torch.manual_seed(len(pos) * 10 + bonds[1])
    spec = SBSSpecString(tuple(SBSSpecCore(Pos2D(h, w), o) for (h, w), o in zip(pos, outs)), bonds, C, q)
    m = ConvSBS(spec, DumbNormalInitialization((q ** C * bonds[1]) ** -0.5)).to(DEV)
    B, H, W = 3, spec.max_height_pos + 6, spec.max_width_pos + 7      # 126 windows: three full groups and a ragged one
    x = torch.randn(C, B, H, W, q, device=DEV, requires_grad=True)
    y = m(x)
    if sbs_band_family_takes(*case) and not (matrix_cores and max(bonds) <= 8):
        fam = "band"   # bonds 5..16: forward and backward of convsbs_band.hip
    assert dctn_amd.last_kernel() == f"convsbs_fwd_{fam}_f32"
    cores64 = [c.detach().cpu().double() for c in m.cores]
    want = R.convsbs_forward(cores64, list(pos), x.detach().cpu().double())
    check(y, want, torch.float32, "forward")
    dy = torch.randn_like(y)
    y.backward(dy)
    assert dctn_amd.last_kernel() == f"convsbs_bwd_{fam}_f32"
    gr = R.grads(lambda xx, *cc: R.convsbs_forward(cc, list(pos), xx), [x.detach().cpu().double()] + cores64, dy.cpu().double())
    check(x.grad, gr[0], torch.float32, "dX")
    for i, (c, gc) in enumerate(zip(m.cores, gr[1:])):
        check(c.grad, gc, torch.float32, f"dCore{i}")
    if len(pos) > 9:
        return   # longer strings accumulate in LDS with float atomics (arrival order)
    # deterministic: a workgroup's waves join in a fixed order and the per-workgroup records are summed in a fixed order
    g1 = [c.grad.clone() for c in m.cores]
    for c in m.cores:
        c.grad = None
    x.grad = None
    m(x).backward(dy)
    assert all(torch.equal(a, c.grad) for a, c in zip(g1, m.cores))


# ------------------------------------------------------------------ register-resident small-bond sweep (convsbs_reg.hip)
SNAKE9 = ((0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2))
REG_CASES = [
    # (positions, bonds, outs, C, q, B, H, W, x_grad, core_grad, strided)
    (SNAKE9, (1,) + (4,) * 8, (1, 1, 1, 1, 2, 1, 1, 1, 1), 1, 3, 5, 32, 32, True, True, False),    # cfg4 geometry (2 bands of 16 rows at B = 128; six bands with halo rows here)
    (SNAKE9, (1,) + (4,) * 8, (1, 1, 1, 1, 2, 1, 1, 1, 1), 1, 3, 130, 20, 20, True, True, False),  # B * bands >= 256: two bands per image
    (SNAKE9, (1, 2, 3, 4, 4, 3, 2, 4, 2), (1,) * 9, 1, 2, 3, 9, 40, True, True, False),            # unequal bonds, one output, wide image
    (SNAKE9, (1,) + (2,) * 8, (2, 1, 1, 1, 1, 1, 1, 1, 1), 2, 2, 4, 8, 9, True, True, True),       # bond 2, two channels, two-valued FIRST core, strided x
    (SNAKE9, (1,) + (3,) * 8, (1, 1, 1, 1, 1, 1, 1, 1, 2), 1, 4, 2, 7, 7, True, True, False),      # q = 4, two-valued LAST core
    (((0, 0), (0, 1), (1, 1), (1, 0)), (1, 4, 4, 4), (1, 2, 1, 1), 1, 3, 6, 6, 5, True, True, False),   # 2 x 2 window, 4 cores
    (SNAKE9, (1,) + (4,) * 8, (1, 1, 1, 1, 2, 1, 1, 1, 1), 2, 2, 3, 10, 10, False, True, False),   # x without gradient
    (SNAKE9, (1,) + (4,) * 8, (1, 1, 1, 1, 2, 1, 1, 1, 1), 1, 2, 3, 10, 10, True, False, False),   # cores without gradient
    (SNAKE9, (1,) + (4,) * 8, (1, 1, 1, 1, 2, 1, 1, 1, 1), 1, 3, 2, 6, 300, True, True, False),    # more windows per band than lanes in a workgroup
]


@pytest.mark.parametrize("case", REG_CASES, ids=lambda c: "n%d_b%s_o%s_C%dq%d_B%d_%dx%d%s%s%s" % (
    len(c[0]), max(c[1]), "".join(map(str, c[2])), c[3], c[4], c[5], c[6], c[7], "" if c[8] else "_nodx", "" if c[9] else "_nodcore",
    "_strided" if c[10] else ""))
def test_convsbs_reg_family_shapes(case):
    """Lane = window, chain state and every core's input state in registers, dX written by the kernel that owns the
    band of pixel rows, dCore through per-workgroup records: bands, halo rows, ragged waves, looping lanes, unequal
    bonds, both channel modes - against the oracle, and bit-reproducible."""
    pos, bonds, outs, C, q, B, H, W, x_grad, core_grad, strided = case
    assert sbs_reg_family_takes(pos, bonds, outs, C, q)
    torch.manual_seed(B + H + W)
    spec = SBSSpecString(tuple(SBSSpecCore(Pos2D(h, w), o) for (h, w), o in zip(pos, outs)), bonds, C, q)
    m = ConvSBS(spec, DumbNormalInitialization((q ** C * max(bonds)) ** -0.5 * 1.2)).to(DEV)
    for c in m.cores:
        c.requires_grad_(core_grad)
    x0 = torch.randn(C, B, H, W, q)
    x = x0.to(DEV)
    if strided:
        x = x.permute(0, 1, 3, 2, 4).contiguous().permute(0, 1, 3, 2, 4)
    x.requires_grad_(x_grad)
    y = m(x)
    assert dctn_amd.last_kernel() == "convsbs_fwd_reg_f32"
    cores64 = [c.detach().cpu().double() for c in m.cores]
    want = R.convsbs_forward(cores64, list(pos), x0.double())
    check(y, want, torch.float32, "forward")
    dy = torch.randn_like(y)
    y.backward(dy)
    assert dctn_amd.last_kernel() == "convsbs_bwd_reg_f32"
    gr = R.grads(lambda xx, *cc: R.convsbs_forward(cc, list(pos), xx), [x0.double()] + cores64, dy.cpu().double())
    if x_grad:
        check(x.grad, gr[0], torch.float32, "dX")
    else:
        assert x.grad is None
    for i, (c, gc) in enumerate(zip(m.cores, gr[1:])):
        if core_grad:
            check(c.grad, gc, torch.float32, f"dCore{i}")
        else:
            assert c.grad is None
    first = ([c.grad.clone() for c in m.cores] if core_grad else []) + ([x.grad.clone()] if x_grad else [])
    for c in m.cores:
        c.grad = None
    x.grad = None
    m(x).backward(dy)
    again = ([c.grad for c in m.cores] if core_grad else []) + ([x.grad] if x_grad else [])
    assert all(torch.equal(a, b) for a, b in zip(first, again))


MV_CASES = [
    # pos, bonds, outs, C, q, B, H, W, x needs grad, cores need grad, strided x
    (SNAKE9, (1,) + (4,) * 8, (1, 1, 1, 1, 10, 1, 1, 1, 1), 2, 2, 3, 10, 10, True, True, False),   # the classifier's final string, bond 4
    (SNAKE9, (1,) + (2,) * 8, (1, 1, 1, 1, 10, 1, 1, 1, 1), 2, 2, 130, 8, 8, True, True, False),   # ... at the reference's default bond, many images
    (SNAKE9, (1,) + (4,) * 8, (1, 1, 1, 1, 10, 1, 1, 1, 1), 1, 3, 2, 12, 40, True, True, True),    # one channel, q = 3, wide rows, strided x
    (SNAKE9, (1,) + (3,) * 8, (16, 1, 1, 1, 1, 1, 1, 1, 1), 1, 2, 2, 7, 9, True, True, False),     # the many-valued core first (no prefix)
    (SNAKE9, (1,) + (4,) * 8, (1, 1, 1, 1, 1, 1, 1, 1, 3), 1, 4, 2, 6, 6, True, True, False),      # ... last (no suffix), q = 4
    (SNAKE9, (1, 2, 4, 3, 4, 2, 4, 3, 4), (1, 1, 5, 1, 1, 1, 1, 1, 1), 2, 2, 3, 9, 8, True, True, False),   # unequal bonds
    (((0, 0), (0, 1), (1, 1), (1, 0)), (1, 4, 4, 4), (1, 7, 1, 1), 1, 3, 6, 6, 5, True, True, False),       # four cores
    (SNAKE9, (1,) + (4,) * 8, (1, 1, 1, 1, 10, 1, 1, 1, 1), 2, 2, 3, 10, 10, False, True, False),  # x without gradient
    (SNAKE9, (1,) + (4,) * 8, (1, 1, 1, 1, 10, 1, 1, 1, 1), 2, 2, 3, 10, 10, True, False, False),  # cores without gradient
]


@pytest.mark.parametrize("case", MV_CASES, ids=lambda c: "n%d_b%s_o%s_C%dq%d_B%d_%dx%d%s%s%s" % (
    len(c[0]), max(c[1]), "x".join(map(str, c[2])), c[3], c[4], c[5], c[6], c[7], "" if c[8] else "_nodx", "" if c[9] else "_nodcore",
    "_strided" if c[10] else ""))
def test_convsbs_reg_family_many_valued_core(case):
    """One many-valued core (the ten-label final string of the reference's classifier, mnist.py:213-223) on the register
    sweep: prefix state, suffix vector, one dot product per output; the many-valued core at either end and in the
    middle, both channel modes, unequal bonds, bands - against the oracle, and bit-reproducible."""
    pos, bonds, outs, C, q, B, H, W, x_grad, core_grad, strided = case
    assert sbs_reg_family_takes(pos, bonds, outs, C, q)
    torch.manual_seed(B + H + W)
    spec = SBSSpecString(tuple(SBSSpecCore(Pos2D(h, w), o) for (h, w), o in zip(pos, outs)), bonds, C, q)
    m = ConvSBS(spec, DumbNormalInitialization((q ** C * max(bonds)) ** -0.5 * 1.2)).to(DEV)
    for c in m.cores:
        c.requires_grad_(core_grad)
    x0 = torch.randn(C, B, H, W, q)
    x = x0.to(DEV)
    if strided:
        x = x.permute(0, 1, 3, 2, 4).contiguous().permute(0, 1, 3, 2, 4)
    x.requires_grad_(x_grad)
    y = m(x)
    assert dctn_amd.last_kernel() == "convsbs_fwd_reg_f32"
    cores64 = [c.detach().cpu().double() for c in m.cores]
    want = R.convsbs_forward(cores64, list(pos), x0.double())
    check(y, want, torch.float32, "forward")
    dy = torch.randn_like(y)
    y.backward(dy)
    assert dctn_amd.last_kernel() == "convsbs_bwd_reg_f32"
    gr = R.grads(lambda xx, *cc: R.convsbs_forward(cc, list(pos), xx), [x0.double()] + cores64, dy.cpu().double())
    if x_grad:
        check(x.grad, gr[0], torch.float32, "dX")
    else:
        assert x.grad is None
    for i, (c, gc) in enumerate(zip(m.cores, gr[1:])):
        if core_grad:
            check(c.grad, gc, torch.float32, f"dCore{i}")
        else:
            assert c.grad is None
    first = ([c.grad.clone() for c in m.cores] if core_grad else []) + ([x.grad.clone()] if x_grad else [])
    for c in m.cores:
        c.grad = None
    x.grad = None
    m(x).backward(dy)
    again = ([c.grad for c in m.cores] if core_grad else []) + ([x.grad] if x_grad else [])
    assert all(torch.equal(a, b) for a, b in zip(first, again))


BAND_CASES = [
    # pos, bonds, outs, C, q, B, H, W, x needs grad, cores need grad, strided x
    (SNAKE9, (1,) + (16,) * 8, (1, 1, 1, 1, 2, 1, 1, 1, 1), 1, 3, 2, 32, 32, True, True, False),   # BASELINE cfg4 r = 16 geometry, two images
    (SNAKE9, (1,) + (16,) * 8, (1, 1, 1, 1, 2, 1, 1, 1, 1), 1, 3, 130, 8, 9, True, True, False),   # more images than CUs: one band per image
    (SNAKE9, (1,) + (16,) * 8, (1, 1, 1, 1, 2, 1, 1, 1, 1), 2, 2, 3, 14, 13, True, True, False),   # the classifier's second layer: two channels
    (SNAKE9, (1,) + (16,) * 8, (1,) * 9, 1, 2, 2, 9, 40, True, True, True),                         # one output, q = 2, strided x, wide rows
    (SNAKE9, (1,) + (16,) * 8, (1, 2, 1, 1, 1, 1, 1, 1, 1), 1, 4, 1, 11, 7, True, True, False),    # two states from the second core on, q = 4
    (SNAKE9, (1,) + (16,) * 8, (1, 1, 1, 1, 1, 1, 1, 2, 1), 1, 3, 5, 7, 7, True, True, False),     # the two-valued core next to the end
    (SNAKE9, (1, 9, 16, 11, 12, 10, 16, 13, 15), (1, 1, 1, 2, 1, 1, 1, 1, 1), 1, 3, 3, 10, 12, True, True, False),   # unequal bonds
    (SNAKE9[:4], (1, 12, 16, 9), (1, 1, 2, 1), 2, 2, 4, 6, 9, True, True, False),                   # four cores
    (((0, 0), (0, 1), (0, 2)), (1, 16, 16), (1, 2, 1), 1, 3, 7, 5, 21, True, True, False),          # one row of pixels: no shared rows
    (((0, 0), (1, 0), (2, 0), (3, 0), (4, 0)), (1, 10, 10, 10, 10), (1,) * 5, 1, 2, 2, 23, 6, True, True, False),   # a column: four shared rows per boundary
    (SNAKE9, (1,) + (16,) * 8, (1, 1, 1, 1, 2, 1, 1, 1, 1), 1, 3, 3, 10, 10, False, True, False),  # x without gradient
    (SNAKE9, (1,) + (16,) * 8, (1, 1, 1, 1, 2, 1, 1, 1, 1), 2, 2, 3, 10, 10, True, False, False),  # cores without gradient
]


@pytest.mark.parametrize("case", BAND_CASES, ids=lambda c: "n%d_b%s_o%s_C%dq%d_B%d_%dx%d%s%s%s" % (
    len(c[0]), max(c[1]), "".join(map(str, c[2])), c[3], c[4], c[5], c[6], c[7], "" if c[8] else "_nodx", "" if c[9] else "_nodcore",
    "_strided" if c[10] else ""))
def test_convsbs_band_family_shapes(case):
    """Bonds 5..16, backward by band-owning workgroups with two roles per SIMD (chain waves and gradient waves): bands of
    one image and the pixel rows they share, ragged tiles, chain waves without a tile, every position of the two-valued
    core, unequal bonds (zero-padded packs), both channel modes, every q - against the oracle, and bit-reproducible."""
    pos, bonds, outs, C, q, B, H, W, x_grad, core_grad, strided = case
    assert sbs_band_family_takes(pos, bonds, outs, C, q)
    torch.manual_seed(B + H + W)
    spec = SBSSpecString(tuple(SBSSpecCore(Pos2D(h, w), o) for (h, w), o in zip(pos, outs)), bonds, C, q)
    m = ConvSBS(spec, DumbNormalInitialization((q ** C * max(bonds)) ** -0.5 * 1.2)).to(DEV)
    for c in m.cores:
        c.requires_grad_(core_grad)
    x0 = torch.randn(C, B, H, W, q)
    x = x0.to(DEV)
    if strided:
        x = x.permute(0, 1, 3, 2, 4).contiguous().permute(0, 1, 3, 2, 4)
    x.requires_grad_(x_grad)
    y = m(x)
    assert dctn_amd.last_kernel() == "convsbs_fwd_band_f32"
    cores64 = [c.detach().cpu().double() for c in m.cores]
    want = R.convsbs_forward(cores64, list(pos), x0.double())
    check(y, want, torch.float32, "forward")
    dy = torch.randn_like(y)
    y.backward(dy)
    assert dctn_amd.last_kernel() == "convsbs_bwd_band_f32"
    gr = R.grads(lambda xx, *cc: R.convsbs_forward(cc, list(pos), xx), [x0.double()] + cores64, dy.cpu().double())
    if x_grad:
        check(x.grad, gr[0], torch.float32, "dX")
    else:
        assert x.grad is None
    for i, (c, gc) in enumerate(zip(m.cores, gr[1:])):
        if core_grad:
            check(c.grad, gc, torch.float32, f"dCore{i}")
        else:
            assert c.grad is None
    first = ([c.grad.clone() for c in m.cores] if core_grad else []) + ([x.grad.clone()] if x_grad else [])
    for c in m.cores:
        c.grad = None
    x.grad = None
    m(x).backward(dy)
    again = ([c.grad for c in m.cores] if core_grad else []) + ([x.grad] if x_grad else [])
    assert all(torch.equal(a, b) for a, b in zip(first, again))


SNAKE9B = ((0, 0), (1, 0), (2, 0), (2, 1), (1, 1), (0, 1), (0, 2), (1, 2), (2, 2))   # mnist.py:201-210


@pytest.mark.parametrize("bond,C,q,outs_a,outs_b,B,H,W", [
    (4, 1, 2, (1, 1, 1, 1, 2, 1, 1, 1, 1), (1, 1, 1, 1, 2, 1, 1, 1, 1), 5, 12, 11),     # the classifier's first layer
    (2, 2, 2, (1, 1, 1, 1, 2, 1, 1, 1, 1), (1, 1, 1, 1, 2, 1, 1, 1, 1), 130, 10, 10),   # its second layer, bond 2, several bands
    (3, 1, 3, (2, 1, 1, 1, 1, 1, 1, 1, 1), (1, 1, 1, 1, 1, 1, 1, 1, 2), 3, 9, 30),      # bond 3, q = 3, different two-valued cores
    (4, 1, 4, (1,) * 9, (1,) * 9, 2, 8, 8),                                              # one output each
    # bonds 9..16: the band family, blockIdx.y = string (the bond-16 two-string layer of mnist.py:224-242)
    (16, 1, 3, (1, 1, 1, 1, 2, 1, 1, 1, 1), (1, 1, 1, 1, 2, 1, 1, 1, 1), 5, 12, 11),
    (16, 2, 2, (1, 1, 1, 1, 2, 1, 1, 1, 1), (1, 1, 2, 1, 1, 1, 1, 1, 1), 130, 10, 10),   # two channels, many images: several bands
    (12, 1, 2, (1,) * 9, (1,) * 9, 3, 9, 30),
    (16, 1, 4, (1,) * 9, (1,) * 9, 2, 8, 8),
])
def test_many_convsbs_strings_in_one_launch(bond, C, q, outs_a, outs_b, B, H, W):
    """`ManyConvSBS.forward` (dctn/conv_sbs.py:367-370) for a layer of two nine-core strings over the same 3 x 3 window:
    one forward launch, one backward launch (dX written once, summed over the strings) - against the oracle per string and
    against the same layer run string by string; only one of the two outputs used (the other's gradient arrives as None)."""
    from dctn_amd.conv_sbs import ManyConvSBS, matrix_core_sweep

    torch.manual_seed(bond * 100 + B)
    specs = (tuple(SBSSpecCore(Pos2D(h, w), o) for (h, w), o in zip(SNAKE9, outs_a)),
             tuple(SBSSpecCore(Pos2D(h, w), o) for (h, w), o in zip(SNAKE9B, outs_b)))
    init = DumbNormalInitialization((q ** C * bond) ** -0.5 * 1.2)
    many = ManyConvSBS(C, q, bond, False, specs, (init, init)).to(DEV)
    x0 = torch.randn(C, B, H, W, q)
    x = x0.to(DEV).requires_grad_(True)
    fam = "band" if bond > 4 else "reg"
    ya, yb = many(x)
    assert dctn_amd.last_kernel() == f"convsbs_many_fwd_{fam}_f32"
    dya, dyb = torch.randn_like(ya), torch.randn_like(yb)
    ((ya * dya).sum() + (yb * dyb).sum()).backward()
    assert dctn_amd.last_kernel() == f"convsbs_many_bwd_{fam}_f32"
    dx_sum = torch.zeros_like(x0, dtype=torch.float64)
    for string, pos, y, dy in ((many.strings[0], SNAKE9, ya, dya), (many.strings[1], SNAKE9B, yb, dyb)):
        cores64 = [c.detach().cpu().double() for c in string.cores]
        check(y, R.convsbs_forward(cores64, list(pos), x0.double()), torch.float32, "forward")
        gr = R.grads(lambda xx, *cc: R.convsbs_forward(cc, list(pos), xx), [x0.double()] + cores64, dy.cpu().double())
        dx_sum += gr[0]
        for i, (c, gc) in enumerate(zip(string.cores, gr[1:])):
            check(c.grad, gc, torch.float32, f"dCore{i}")
    check(x.grad, dx_sum, torch.float32, "dX")
    # the same layer string by string (the matrix-core family: launches per string) gives the same numbers
    first = [x.grad.clone()] + [p.grad.clone() for p in many.parameters()]
    x.grad = None
    for p in many.parameters():
        p.grad = None
    with matrix_core_sweep():
        za, zb = many(x)
        assert "many" not in dctn_amd.last_kernel()
        ((za * dya).sum() + (zb * dyb).sum()).backward()
    for a, b in zip(first, [x.grad] + [p.grad for p in many.parameters()]):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-5 * float(a.abs().max()))
    # only the second output is used
    x.grad = None
    for p in many.parameters():
        p.grad = None
    _, yb2 = many(x)
    (yb2 * dyb).sum().backward()
    assert all(float(c.grad.abs().max()) == 0.0 for c in many.strings[0].cores)
    cores64 = [c.detach().cpu().double() for c in many.strings[1].cores]
    gr = R.grads(lambda xx, *cc: R.convsbs_forward(cc, list(SNAKE9B), xx), [x0.double()] + cores64, dyb.cpu().double())
    check(x.grad, gr[0], torch.float32, "dX (one output used)")
