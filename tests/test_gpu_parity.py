"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the Python
host layer and the C-ABI, against (a) the golden vectors produced by the reference itself and
(b) the CPU oracle on seeded inputs.  Tolerances are stated per dtype below.

Nothing here reads /root/reference.
"""
import functools
import glob
import itertools
import os

import numpy as np
import pytest
import torch

import dctn_amd
from dctn_amd import _lib
from dctn_amd.conv_sbs import ConvSBS, DumbNormalInitialization, ManyConvSBS
from dctn_amd.conv_sbs_spec import SBSSpecCore, SBSSpecString
from dctn_amd.eps import eps, eps_one_by_one, keep_gemm_result, transform_in_slices
from dctn_amd.epses_composition import contract_with_input
from dctn_amd.logmatmulexp import logmatmulexp, logmatmulexp_batched, logmatmulexp_fold, logmatmulexp_lowmem
from dctn_amd.pos2d import Pos2D
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = torch.device("cuda:0")

# float64: different summation order only.  float32: fp32 accumulation of up to 2^16 products.
TOL = {torch.float64: dict(rtol=1e-9, atol=1e-11), torch.float32: dict(rtol=2e-4, atol=2e-5)}


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def dev(a, dtype=None, grad=False):
    t = torch.from_numpy(np.asarray(a)).to(DEV)
    if dtype is not None:
        t = t.to(dtype)
    return t.requires_grad_(grad)


def close(got, want, dtype, scale=None):
    want = torch.as_tensor(want).to(torch.float64)
    got = got.detach().cpu().to(torch.float64)
    tol = dict(TOL[dtype])
    if scale is None:
        scale = float(want.abs().max()) or 1.0
    tol["atol"] = tol["atol"] * max(scale, 1e-30)
    ok = torch.allclose(got, want, **tol)
    if not ok:
        print("max abs err", float((got - want).abs().max()), "scale", scale)
    return ok


EPS_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "eps_c*.npz")))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("name", EPS_CASES)
def test_eps_golden(name, dtype):
    g = load(name)
    x, core = dev(g["x"], dtype, True), dev(g["core"], dtype, True)
    y = eps(core, x)
    assert "eps_fwd" in dctn_amd.last_kernel()
    assert y.shape == g["y"].shape and y.dtype == dtype
    assert close(y, g["y"], dtype)
    assert close(eps_one_by_one(core, x), g["y_one_by_one"], dtype)
    y.backward(dev(g["dy"], dtype))
    assert "eps_bwd" in dctn_amd.last_kernel()
    assert close(x.grad, g["dx"], dtype)
    assert close(core.grad, g["dcore"], dtype)


def test_eps_single_pixel_output():  # reference test: tests/test_eps.py:9-26
    x = torch.randn((2, 3, 2, 2, 2), dtype=torch.float64)
    core = torch.rand((*(2 for _ in range(8)), 4), dtype=torch.float64)
    got = eps_one_by_one(core.to(DEV), x.to(DEV)).reshape(3, 4)
    want = torch.einsum("abcdefgho,ia,ib,ic,id,ie,if,ig,ih->io", core, x[0, :, 0, 0], x[1, :, 0, 0], x[0, :, 0, 1],
                        x[1, :, 0, 1], x[0, :, 1, 0], x[1, :, 1, 0], x[0, :, 1, 1], x[1, :, 1, 1])
    assert torch.allclose(got.cpu(), want)


def test_eps_two_pixels_output():  # reference test: tests/test_eps.py:29-61
    x = torch.randn((1, 1, 4, 3, 2), dtype=torch.float64)
    core = torch.rand((*(2 for _ in range(9)), 4), dtype=torch.float64)
    got = eps_one_by_one(core.to(DEV), x.to(DEV)).cpu()
    assert got.shape == (1, 2, 1, 4)
    for row in (0, 1):
        pix = [x[0, 0, row + dh, dw] for dh in range(3) for dw in range(3)]
        want = torch.einsum("abcdefghio,a,b,c,d,e,f,g,h,i->o", core, *pix)
        assert torch.allclose(got[0, row, 0], want)


@pytest.mark.parametrize(
    "C,B,H,W,Q,K,O,dtype",
    [
        (1, 5, 28, 28, 2, 3, 4, torch.float32),   # BASELINE cfg2 geometry, reduced batch
        (1, 2, 28, 28, 2, 4, 2, torch.float64),   # cfg1 geometry (eps2d benchmark), reduced batch
        (1, 2, 12, 11, 2, 4, 4, torch.float32),   # cfg3a layer 1 core
        (1, 2, 6, 7, 4, 3, 6, torch.float32),     # cfg3a layer 2 core (6 MiB), reduced image
        (1, 3, 9, 9, 8, 2, 8, torch.float32),     # cfg3b layer 2
        (2, 2, 7, 6, 2, 2, 3, torch.float64),     # two channels
        (1, 2, 7, 7, 3, 2, 20, torch.float32),    # out_size > 8 (several o-tiles)
        (3, 2, 4, 4, 3, 1, 5, torch.float64),     # K = 1
        (1, 70, 5, 5, 2, 3, 4, torch.float32),    # several workgroups, ragged tail
    ],
)
def test_eps_vs_oracle_seeded(C, B, H, W, Q, K, O, dtype):
    torch.manual_seed(C * 1000 + B * 100 + K * 10 + Q)
    N = K * K * C
    x = torch.randn(C, B, H, W, Q, dtype=dtype)
    core = torch.randn(*(Q,) * N, O, dtype=dtype) * Q ** (-N / 4)
    xd, cd = x.to(DEV).requires_grad_(True), core.to(DEV).requires_grad_(True)
    y = eps(cd, xd)
    want = R.eps_4step(core.double(), x.double())
    assert close(y, want, dtype)
    dy = torch.randn(*want.shape, dtype=dtype)
    y.backward(dy.to(DEV))
    dcore, dx = R.grads(R.eps_4step, [core.double(), x.double()], dy.double())
    assert close(xd.grad, dx, dtype)
    assert close(cd.grad, dcore, dtype)


def test_eps_noncontiguous_input_and_needs_input_grad():
    torch.manual_seed(3)
    big = torch.randn(1, 9, 8, 8, 2, dtype=torch.float64, device=DEV)
    core = (torch.randn(*(2,) * 9, 4, dtype=torch.float64, device=DEV) / 4).requires_grad_(True)
    part = big.split(4, dim=1)[1]                      # dctn/eps.py:136 passes such slices
    assert not part.is_contiguous() or part.storage_offset() != 0
    perm = big.permute(0, 1, 3, 2, 4)                  # genuinely strided
    for v in (part, perm):
        want = R.eps_4step(core.detach().cpu(), v.cpu())
        assert close(eps(core, v), want, torch.float64)
    y = eps(core, perm)                                # input without grad: only dCore is produced
    y.backward(torch.ones_like(y))
    assert core.grad is not None
    frozen = core.detach()                             # new_runner.py:443-444 freezes cores
    xg = perm.clone().requires_grad_(True)
    eps(frozen, xg).sum().backward()
    dcore, dx = R.grads(R.eps_4step, [frozen.cpu(), perm.cpu()], torch.ones(y.shape, dtype=torch.float64))
    assert close(xg.grad, dx, torch.float64)
    assert close(core.grad, dcore, torch.float64)
    out = transform_in_slices(frozen, big, 4)
    assert out.shape == (1, 9, 6, 6, 4) and not out.requires_grad


def test_epses_composition_golden_and_eps_plus_linear():
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd

    g = load("epses_composition_33_25")
    for dtype in (torch.float64, torch.float32):
        x, e1, e2 = dev(g["x"], dtype, True), dev(g["e1"], dtype, True), dev(g["e2"], dtype, True)
        y = contract_with_input((e1, e2), x)
        assert close(y, g["y"], dtype)
        y.backward(dev(g["dy"], dtype))
        assert close(x.grad, g["dx"], dtype) and close(e1.grad, g["de1"], dtype) and close(e2.grad, g["de2"], dtype)
    torch.manual_seed(5)
    m = EPSesPlusLinear(((3, 4), (2, 3)), UnitTheoreticalOutputStd(), 1.0, DEV, torch.float32, image_size=10)
    x = torch.rand(1, 6, 10, 10, 2, device=DEV)
    out = m(x)
    assert out.shape == (6, 10)
    want = R.eps_plus_linear_forward([c.detach().cpu().double() for c in m.epses], m.linear.weight.detach().cpu().double(),
                                     m.linear.bias.detach().cpu().double(), x.cpu().double())
    assert close(out, want, torch.float32)
    out.logsumexp(1).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    # component dropout (eps_plus_linear.py:139-143) with a PINNED mask: logits and parameter gradients must be those of
    # the oracle applied to the masked, rescaled cores  core * mask / p
    m.train()
    m.load_state_dict({**m.state_dict(), "p": torch.tensor(0.5)})     # the host copy of p follows the state_dict
    assert m._p_float == 0.5
    gen = torch.Generator().manual_seed(11)
    masks = [torch.bernoulli(torch.full(c.shape, 0.5), generator=gen) for c in m.epses]
    queue = list(masks)
    m.dropout_mask = lambda core, p: queue.pop(0).to(core)
    for prm in m.parameters():
        prm.grad = None
    out = m(x)
    assert not queue and out.shape == (6, 10)
    out.logsumexp(1).sum().backward()
    cores64 = [c.detach().cpu().double().requires_grad_(True) for c in m.epses]
    w64 = m.linear.weight.detach().cpu().double()
    want = R.eps_plus_linear_forward([c * mk.double() / 0.5 for c, mk in zip(cores64, masks)], w64,
                                     m.linear.bias.detach().cpu().double(), x.cpu().double())
    assert close(out, want.detach(), torch.float32)
    want.logsumexp(1).sum().backward()
    for got, ref, mk in zip(m.epses, cores64, masks):
        assert close(got.grad, ref.grad, torch.float32)
        assert bool((got.grad.cpu()[mk == 0] == 0).all())          # dropped components get no gradient
    del m.dropout_mask
    assert m(x).shape == (6, 10)                                    # and the random-mask path runs


# ------------------------------------------------------------------ ConvSBS
SBS_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "sbs_*.npz")))


def sbs_from_golden(g, dtype):
    spec = SBSSpecString(
        tuple(SBSSpecCore(Pos2D(int(h), int(w)), int(o)) for (h, w), o in zip(g["positions"], g["out_sizes"])),
        tuple(int(b) for b in g["bond_sizes"]), int(g["C"]), int(g["q"]),
    )
    m = ConvSBS(spec).to(dtype)
    with torch.no_grad():
        for i, c in enumerate(m.cores):
            c.copy_(torch.from_numpy(g[f"core{i}"]))
    return m.to(DEV)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("name", SBS_CASES)
def test_convsbs_golden(name, dtype):
    g = load(name)
    m = sbs_from_golden(g, dtype)
    x = dev(g["x"], dtype, True)
    y = m(x)
    assert "convsbs_fwd" in dctn_amd.last_kernel()
    assert close(y, g["y"], dtype)
    y.backward(dev(g["dy"], dtype))
    assert close(x.grad, g["dx"], dtype)
    for i, c in enumerate(m.cores):
        assert close(c.grad, g[f"dcore{i}"], dtype), f"dcore{i}"
    # tuple-of-channels input (conv_sbs.py:258-262) gives the same result
    y2 = m(tuple(ch for ch in x.detach()))
    assert torch.equal(y2, y.detach())


def test_conversion_all_24_permutations():  # reference test: tests/test_conversion_of_convsbs_to_eps.py:13-56
    cores = (SBSSpecCore(Pos2D(0, 0), 1), SBSSpecCore(Pos2D(0, 1), 3), SBSSpecCore(Pos2D(1, 0), 2), SBSSpecCore(Pos2D(1, 1), 4))
    torch.manual_seed(24)
    for perm in itertools.permutations(cores):
        spec = SBSSpecString(perm, (3, 4, 5, 6), 2, 2)
        m = ConvSBS(spec).double().to(DEV)
        with torch.no_grad():
            eps_tensor = m.as_eps()
        assert eps_tensor.shape == (2,) * 8 + (24,)
        x = torch.randn(2, 3, 4, 5, 2, dtype=torch.float64, device=DEV, requires_grad=True)
        ys = m(x)
        dy = torch.randn_like(ys)
        ys.backward(dy)
        g_sbs = x.grad.clone()
        x.grad.zero_()
        ye = eps(eps_tensor, x)
        assert torch.allclose(ye, ys)
        ye.backward(dy)
        assert torch.allclose(x.grad, g_sbs)
        # ... and neither side is only checked against the other HIP path: the oracle's own `as_eps`, forward and input
        # gradient (oracle/ref_cpu.py: dctn/conv_sbs.py:226-304 restated) for every one of the 24 orders
        cores64 = [c.detach().cpu() for c in m.cores]
        pos = [(c.position.h, c.position.w) for c in perm]
        want_eps = R.convsbs_as_eps(cores64, pos)
        assert close(eps_tensor, want_eps, torch.float64)
        xc = x.detach().cpu().requires_grad_(True)
        want_y = R.convsbs_forward(cores64, pos, xc)
        want_y.backward(dy.cpu())
        assert close(ys, want_y.detach(), torch.float64) and close(ye, want_y.detach(), torch.float64)
        assert close(g_sbs, xc.grad, torch.float64) and close(x.grad, xc.grad, torch.float64)
        assert close(R.eps_4step(want_eps, x.detach().cpu()), want_y.detach(), torch.float64)   # the oracle agrees with itself


def test_conversion_ring_float32_runs_on_the_matrix_core_sweep():
    """The ring of the reference's conversion test (bonds (3, 4, 5, 6), outputs (1, 3, 2, 4), every order of the cores;
    tests/test_conversion_of_convsbs_to_eps.py:13-56) in float32: unequal bonds are padded to one tile, the closing bond and
    the cores with several outputs run as slices - forward and backward both on the matrix-core sweep, against the oracle."""
    cores = (SBSSpecCore(Pos2D(0, 0), 1), SBSSpecCore(Pos2D(0, 1), 3), SBSSpecCore(Pos2D(1, 0), 2), SBSSpecCore(Pos2D(1, 1), 4))
    torch.manual_seed(25)
    for perm in itertools.permutations(cores):
        m = ConvSBS(SBSSpecString(perm, (3, 4, 5, 6), 2, 2)).to(DEV)
        x = torch.randn(2, 3, 4, 5, 2, device=DEV, requires_grad=True)
        ys = m(x)
        assert dctn_amd.last_kernel() == "convsbs_fwd_mfma_f32"
        dy = torch.randn_like(ys)
        ys.backward(dy)
        assert dctn_amd.last_kernel() == "convsbs_bwd_mfma_f32"
        cores64 = [c.detach().cpu().double() for c in m.cores]
        pos = [(c.position.h, c.position.w) for c in perm]
        gr = R.grads(lambda xx, *cc: R.convsbs_forward(cc, pos, xx), [x.detach().cpu().double()] + cores64, dy.cpu().double())
        assert close(ys, R.convsbs_forward(cores64, pos, x.detach().cpu().double()), torch.float32)
        assert close(x.grad, gr[0], torch.float32)
        for c, gc in zip(m.cores, gr[1:]):
            assert close(c.grad, gc, torch.float32)


@pytest.mark.parametrize("r,q,C,B,HW", [(4, 3, 1, 3, 12), (8, 3, 1, 2, 10), (16, 3, 1, 2, 8), (16, 2, 2, 2, 7)])
def test_convsbs_vs_oracle_mnist_snake(r, q, C, B, HW):
    """BASELINE cfg4: the 9-core snake of mnist.py:190-199 on the CIFAR colour layout (q=3) and its
    second-layer variant (C=2, q=2), float32."""
    snake = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
    torch.manual_seed(r * 10 + q)
    many = ManyConvSBS(C, q, r, False, (tuple(SBSSpecCore(Pos2D(*p), 2 if i == 4 else 1) for i, p in enumerate(snake)),),
                       (DumbNormalInitialization((q**C * r) ** -0.5),))
    m = many.strings[0].to(DEV)
    x = torch.randn(C, B, HW, HW, q, device=DEV, requires_grad=True)
    (y,) = many(x)
    # bond <= 4: lane-per-window register sweep; 5..16: the band family (convsbs_band.hip; two state values per lane up to 8)
    fam = "reg" if r <= 4 else "band"
    assert dctn_amd.last_kernel() == f"convsbs_fwd_{fam}_f32"
    cores64 = [c.detach().cpu().double() for c in m.cores]
    want = R.convsbs_forward(cores64, snake, x.detach().cpu().double())
    assert close(y, want, torch.float32)
    dy = torch.randn_like(y)
    y.backward(dy)
    assert dctn_amd.last_kernel() == f"convsbs_bwd_{fam}_f32"
    gr = R.grads(lambda xx, *cc: R.convsbs_forward(cc, snake, xx), [x.detach().cpu().double()] + cores64, dy.cpu().double())
    assert close(x.grad, gr[0], torch.float32)
    for c, gc in zip(m.cores, gr[1:]):
        assert close(c.grad, gc, torch.float32)


def test_convsbs_forward_reports_whether_it_wrote_the_saved_states():
    """`dctn_convsbs_fwd` returns DCTN_SAVED only when the matrix-core sweep wrote the forward states; a string the size
    query accepts but the sweep's LDS plan declines (bond 16, long strings) runs the generic forward, the buffer stays
    uninitialised and must never reach `dctn_convsbs_bwd_saved` - checked with a poisoned allocator and the oracle."""
    lib = _lib.lib()
    seen = set()
    for rows, cols, r, q in ((2, 5, 8, 2), (5, 5, 16, 2), (4, 6, 16, 2)):   # (ten cores and more: outside the band family)
        pos = [(h, w if h % 2 == 0 else cols - 1 - w) for h in range(rows) for w in range(cols)]   # boustrophedon snake
        n = len(pos)
        torch.manual_seed(n)
        outs = tuple(2 if i == n // 2 else 1 for i in range(n))
        spec = SBSSpecString(tuple(SBSSpecCore(Pos2D(h, w), o) for (h, w), o in zip(pos, outs)), (1,) + (r,) * (n - 1), 1, q)
        m = ConvSBS(spec, DumbNormalInitialization((q * r) ** -0.5 * 1.3)).to(DEV)
        B, HW = 2, max(rows, cols) + 3
        x = torch.randn(1, B, HW, HW, q, device=DEV, requires_grad=True)
        plan_args = (n, _lib.int_array(outs), _lib.int_array(spec.bond_sizes), 1, B, HW, HW, q,
                     _lib.int_array([p[0] for p in pos]), _lib.int_array([p[1] for p in pos]), _lib.F32)
        nstates = lib.dctn_convsbs_saved_states_bytes(*plan_args)
        assert nstates > 0
        # raw call: the return code says whether the buffer was written
        cores_c = [c.detach().contiguous() for c in m.cores]
        states = torch.full((nstates,), 0xFF, dtype=torch.uint8, device=DEV)
        out = torch.empty(B, HW - rows + 1, HW - cols + 1, 2, device=DEV)
        rc = lib.dctn_convsbs_fwd(x.data_ptr(), _lib.strides5(x), _lib.ptr_array(cores_c), out.data_ptr(), n, plan_args[1],
                                  plan_args[2], plan_args[8], plan_args[9], 1, B, HW, HW, q, states.data_ptr(), states.numel(),
                                  _lib.F32, _lib.stream_ptr(DEV))
        torch.cuda.synchronize()
        assert rc in (0, _lib.SAVED), (rows, cols, r, rc)
        seen.add(rc)
        if rc == _lib.SAVED:
            assert dctn_amd.last_kernel() == "convsbs_fwd_mfma_f32" and not bool((states == 0xFF).all())
        else:
            assert dctn_amd.last_kernel() == "convsbs_fwd_generic" and bool((states == 0xFF).all())   # untouched
        del states
        torch.empty(nstates + (1 << 20), dtype=torch.uint8, device=DEV).fill_(0xFF)   # poison what the module allocates next
        y = m(x)
        cores64 = [c.detach().cpu().double() for c in m.cores]
        want = R.convsbs_forward(cores64, pos, x.detach().cpu().double())
        assert close(y, want, torch.float32)
        dy = torch.randn_like(y)
        y.backward(dy)
        gr = R.grads(lambda xx, *cc: R.convsbs_forward(cc, pos, xx), [x.detach().cpu().double()] + cores64, dy.cpu().double())
        assert close(x.grad, gr[0], torch.float32)
        for c, gc in zip(m.cores, gr[1:]):
            assert close(c.grad, gc, torch.float32)
    assert _lib.SAVED in seen   # (whether a string is declined depends on the LDS plan; the ten-core bond-8 snake always saves)


# ------------------------------------------------------------------ logmatmulexp
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_logmatmulexp_golden(dtype):
    g = load("logmatmulexp")
    for i in (0, 1):
        A, B = dev(g[f"A{i}"], dtype, True), dev(g[f"B{i}"], dtype, True)
        y = logmatmulexp(A, B)
        assert close(y, g[f"y{i}"], dtype)
        y.backward(dev(g[f"dy{i}"], dtype))
        assert close(A.grad, g[f"dA{i}"], dtype) and close(B.grad, g[f"dB{i}"], dtype)
        assert torch.equal(logmatmulexp_lowmem(A.detach(), B.detach()), y.detach())
    y2 = logmatmulexp(dev(g["A2"], dtype), dev(g["B2"], dtype)).cpu()
    want = torch.from_numpy(g["y2"])
    assert torch.equal(torch.isinf(y2), torch.isinf(want)) and torch.all(y2[2] == -float("inf"))
    fin = torch.isfinite(want)
    assert close(y2[fin], want[fin], dtype)
    with pytest.raises(AssertionError):  # dctn/logmatmulexp.py:10
        logmatmulexp(dev(g["A0"], dtype), dev(g["A0"], dtype))


def test_logmatmulexp_fold_and_batched():
    g = load("logmatmulexp")
    mats = dev(g["fold_mats"], torch.float32)
    y = functools.reduce(logmatmulexp, list(mats))       # logmatmulexp_benchmark/benchmark.py:30
    assert torch.allclose(y.cpu(), torch.from_numpy(g["fold_y"]), rtol=1e-5, atol=1e-4)
    torch.manual_seed(9)
    m = torch.randn(37, 9, 16, 16, dtype=torch.float64)  # BASELINE cfg5 geometry: 9 16x16 matrices / window
    md = m.to(DEV).requires_grad_(True)
    yf = logmatmulexp_fold(md)
    want = R.logmatmulexp_fold_batched(m)
    assert close(yf, want, torch.float64)
    dy = torch.randn_like(want)
    yf.backward(dy.to(DEV))
    (gm,) = R.grads(R.logmatmulexp_fold_batched, [m], dy)
    assert close(md.grad, gm, torch.float64)
    a, b = torch.randn(5, 4, 6, dtype=torch.float64), torch.randn(1, 6, 3, dtype=torch.float64)
    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    yb = logmatmulexp_batched(ad, bd)
    ref = lambda aa, bb: torch.logsumexp(aa.unsqueeze(3) + bb.unsqueeze(1), dim=2)
    assert close(yb, ref(a, b), torch.float64)
    dyb = torch.randn(5, 4, 3, dtype=torch.float64)
    yb.backward(dyb.to(DEV))
    ga, gb = R.grads(ref, [a, b], dyb)
    assert close(ad.grad, ga, torch.float64) and close(bd.grad, gb, torch.float64)


# ------------------------------------------------------------------ bf16 MFMA family (q2-reg)
# bf16 operands (8-bit mantissa), f32 accumulation: every Khatri-Rao product, core entry and output is rounded to
# bf16 once (relative error <= 2^-9 each).  Three bounds, all must hold:
#   * max |err| < 2e-2 of the output scale (catches gross errors),
#   * rms(err) < 6e-3 * rms(want) (the roundings are independent: small-magnitude outputs are constrained too),
#   * where the caller supplies mag[e] = sum over the terms of |term| (the same contraction on |operands|):
#     per element |err| <= 2^-7 * mag + 2^-8 * |want|, i.e. four roundings of slack on the worst case.
BF16_TOL = 2e-2
BF16_RMS_TOL = 6e-3


def bf16_close(got, want, mag=None):
    want = want.double()
    got = got.detach().cpu().double()
    diff = (got - want).abs()
    scale = float(want.abs().max()) or 1.0
    err = float(diff.max()) / scale
    rms = float(diff.square().mean().sqrt()) / (float(want.square().mean().sqrt()) or 1.0)
    ok = err < BF16_TOL and rms < BF16_RMS_TOL
    if mag is not None:
        bound = 2.0 ** -7 * mag.double().abs() + 2.0 ** -8 * want.abs() + 1e-300
        ok = ok and bool((diff <= bound).all())
        if not bool((diff <= bound).all()):
            print("bf16 per-element bound exceeded by factor", float((diff / bound).max()))
    if not ok:
        print("bf16 rel max err", err, "rel rms err", rms)
    return ok


@pytest.mark.parametrize(
    "C,B,H,W,K,O",
    [(1, 7, 28, 28, 3, 4), (1, 70, 8, 8, 3, 4), (2, 5, 9, 7, 2, 4), (1, 3, 10, 10, 3, 1), (1, 3, 10, 10, 3, 2),
     (1, 3, 10, 10, 3, 3), (1, 3, 10, 10, 3, 6), (1, 3, 10, 10, 3, 8), (1, 2, 9, 9, 3, 10), (2, 3, 6, 6, 2, 16)],
)
def test_eps_bf16_mfma_vs_oracle(C, B, H, W, K, O):
    torch.manual_seed(100 * K + O)
    N = K * K * C
    u = torch.rand(C, B, H, W)
    x = torch.stack((torch.sin(u * torch.pi / 2) ** 2, torch.cos(u * torch.pi / 2) ** 2), dim=-1).bfloat16()
    core = (torch.randn(*(2,) * N, O) * 2 ** (-N / 4)).bfloat16()
    xd, cd = x.to(DEV), core.to(DEV).requires_grad_(True)
    y = eps(cd, xd)
    assert dctn_amd.last_kernel() == "eps_fwd_mfma_q2reg"
    assert y.dtype == torch.bfloat16
    want = R.eps_4step(core.double(), x.double())
    assert bf16_close(y, want, mag=R.eps_4step(core.double().abs(), x.double().abs()))
    dy = torch.randn(*want.shape).bfloat16()
    y.backward(dy.to(DEV))
    assert dctn_amd.last_kernel() == "eps_bwd_mfma_q2reg"
    dcore, dx = R.grads(R.eps_4step, [core.double(), x.double()], dy.double())
    dcore_mag, _ = R.grads(R.eps_4step, [core.double().abs(), x.double().abs()], dy.double().abs())
    assert bf16_close(cd.grad, dcore, mag=dcore_mag)
    # with an input gradient too: dCore on the MFMA family, dX on the generic kernels
    xg = xd.clone().requires_grad_(True)
    cd.grad = None
    eps(cd, xg).backward(dy.to(DEV))
    assert bf16_close(cd.grad, dcore) and bf16_close(xg.grad, dx)


def test_eps_bf16_mfma_strided_input_and_f32_policy():
    torch.manual_seed(8)
    x = torch.rand(1, 6, 12, 12, 2).bfloat16().to(DEV)
    core = (torch.randn(*(2,) * 9, 4) / 4).bfloat16().to(DEV)
    xt = x.permute(0, 1, 3, 2, 4)  # strided: scalar-load path of the same kernel
    assert bf16_close(eps(core, xt), R.eps_4step(core.cpu().double(), xt.cpu().double()))
    assert dctn_amd.last_kernel() == "eps_fwd_mfma_q2reg"
    # float32 tensors keep exact f32 arithmetic unless the caller opts in to bf16 operands
    xf, cf = x.float(), core.float().requires_grad_(True)
    y = eps(cf, xf)
    assert dctn_amd.last_kernel() == "eps_fwd_q2f32"  # exact f32 on v_mfma_f32_32x32x2_f32, register-resident core (round 5)
    assert close(y, R.eps_4step(cf.detach().cpu().double(), xf.cpu().double()), torch.float32)
    dctn_amd.set_float32_matmul_precision("bf16")
    try:
        y2 = eps(cf, xf)
        assert dctn_amd.last_kernel() == "eps_fwd_mfma_q2reg" and y2.dtype == torch.float32
        assert bf16_close(y2, y.detach().cpu())
        y2.sum().backward()
        assert dctn_amd.last_kernel() == "eps_bwd_mfma_q2reg"
        dcore, _ = R.grads(R.eps_4step, [cf.detach().cpu().double(), xf.cpu().double()], torch.ones(y.shape).double())
        assert bf16_close(cf.grad, dcore)
    finally:
        dctn_amd.set_float32_matmul_precision("exact")


# ------------------------------------------------------------------ f32 MFMA family "bigcore"
@pytest.mark.parametrize(
    "C,B,H,W,Q,K,O",
    [
        (1, 3, 9, 10, 2, 4, 4),    # cfg3a layer 1 core (1 MiB), small image
        (1, 2, 8, 8, 2, 4, 8),     # cfg3b layer 1 (O = 8: outputs split over the lane halves)
        (1, 2, 7, 7, 2, 4, 2),     # cfg1 core in f32
        (1, 2, 6, 7, 4, 3, 6),     # cfg3a layer 2 core (6 MiB), O padded 6 -> 8
        (1, 3, 9, 9, 8, 2, 8),     # cfg3b layer 2
        (1, 40, 6, 6, 4, 2, 5),    # Q=4 K=2, O padded 5 -> 8, several workgroups
        (2, 3, 6, 5, 4, 2, 3),     # two channels, N=8 factors of size 4
        (1, 2, 6, 6, 2, 3, 16),    # O = 16
        (1, 2, 7, 7, 16, 2, 4),    # Q = 16
    ],
)
def test_eps_f32_bigcore_vs_oracle(C, B, H, W, Q, K, O):
    torch.manual_seed(7 * Q + K + O)
    N = K * K * C
    x = torch.randn(C, B, H, W, Q)
    core = torch.randn(*(Q,) * N, O) * Q ** (-N / 4)
    xd, cd = x.to(DEV).requires_grad_(True), core.to(DEV).requires_grad_(True)
    y = eps(cd, xd)
    if Q < 16:  # Q = 16 exceeds the family's LDS budget and falls back to the generic kernels
        assert dctn_amd.last_kernel() == "eps_fwd_mfma_bigcore_f32_saving"   # x needs a gradient: the GEMM result is kept
    want = R.eps_4step(core.double(), x.double())
    assert close(y, want, torch.float32)
    dy = torch.randn(*want.shape)
    y.backward(dy.to(DEV))
    if Q < 16:
        assert dctn_amd.last_kernel() == "eps_bwd_mfma_bigcore_f32_savedz"
    dcore, dx = R.grads(R.eps_4step, [core.double(), x.double()], dy.double())
    assert close(xd.grad, dx, torch.float32)
    assert close(cd.grad, dcore, torch.float32)
    if Q < 16:
        # bit-reproducible: the window chunks of the dCore product are written as slices and summed in a fixed order
        # (float atomics in arrival order until round 2), dX comes from a deterministic gather
        first = (xd.grad.clone(), cd.grad.clone())
        xd.grad = cd.grad = None
        eps(cd, xd).backward(dy.to(DEV))
        assert torch.equal(cd.grad, first[1]) and torch.equal(xd.grad, first[0])


@pytest.mark.parametrize(
    "C,B,H,W,Q,K,O",
    [
        (1, 3, 10, 10, 2, 4, 4),    # cfg3a layer 1 shape: n1 = 8, four Z rows per b
        (1, 3, 7, 7, 4, 3, 6),      # cfg3a layer 2 shape: exact O = 6 (rows (b, o) in memory order, 3 quads = 2 b)
        (1, 4, 6, 7, 8, 2, 8),      # cfg3b layer 2: n1 = 2
        (2, 3, 6, 6, 2, 2, 16),     # two channels, sixteen Z rows per b
        (1, 3, 9, 9, 2, 3, 5),      # odd O: run-time row -> (b, o) walk
        (1, 70, 6, 6, 2, 3, 2),     # two b per row quad (N = 9: halves of 5 and 4 factors), more windows than one workgroup's 64
    ],
)
def test_eps_f32_bigcore_saved_gemm_result(C, B, H, W, Q, K, O):
    """Training forward that keeps the GEMM result Z for the backward (`dctn_eps_fwd_save` / `dctn_eps_bwd_saved`, the
    reference's autograd saves it too: dctn/eps.py:25-30): same output bit for bit, input gradient against the oracle
    and against the recomputing backward, buffer poisoned beforehand, HIP-graph replay."""
    from dctn_amd.eps import _EpsFunction

    torch.manual_seed(7)
    N = K * K * C
    x0 = torch.rand(C, B, H, W, Q, dtype=torch.float64) + 0.25
    c0 = torch.randn(*([Q] * N), O, dtype=torch.float64) * (Q ** N) ** -0.5
    dy0 = torch.randn(B, H - K + 1, W - K + 1, O, dtype=torch.float64)
    x64, c64 = x0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
    R.eps_4step(c64, x64).backward(dy0)
    lib = _lib.lib()
    nsaved = lib.dctn_eps_saved_bytes(C, B, H, W, Q, K, O, _lib.F32, 0)
    assert nsaved >= B * (H - K + 1) * (W - K + 1) * Q ** (N // 2) * O * 4

    def run(keep):
        x, core = dev(x0, torch.float32, True), dev(c0, torch.float32, True)
        y = _EpsFunction.apply(core, x, keep)
        fwd = dctn_amd.last_kernel()
        y.backward(dev(dy0, torch.float32))
        return y.detach(), x.grad, core.grad, fwd, dctn_amd.last_kernel()

    torch.empty(nsaved + (1 << 20), dtype=torch.uint8, device=DEV).fill_(0xFF)   # poison what the allocator hands out next
    y1, dx1, dc1, f1, b1 = run(True)
    y0, dx0, dc0, f0, b0 = run(False)
    assert f1 == "eps_fwd_mfma_bigcore_f32_saving" and b1 == "eps_bwd_mfma_bigcore_f32_savedz", (f1, b1)
    f32 = torch.float32
    if lib.dctn_eps_family(C, B, H, W, Q, K, O, _lib.F32, 0) == 4:
        # Q = 2 with 9 factors and a small out size: without the buffer the register-resident float32 family takes the
        # forward and dCore (other kernels, another summation order), the input gradient stays on the large-core family
        assert f0 == "eps_fwd_q2f32" and b0 == "eps_bwd_mfma_bigcore_f32", (f0, b0)
        assert close(y0, y1.cpu(), f32) and close(dc0, c64.grad, f32)
    else:
        assert f0 == "eps_fwd_mfma_bigcore_f32" and b0 == "eps_bwd_mfma_bigcore_f32", (f0, b0)
        assert torch.equal(y1, y0) and torch.equal(dc1, dc0)
    assert close(dx1, x64.grad, f32) and close(dx0, x64.grad, f32)
    assert close(dc1, c64.grad, f32)
    # replayed from a HIP graph (the saved buffer comes from the graph's pool), with dirtied outputs in between
    xs, cs = dev(x0, f32, True), dev(c0, f32, True)
    dys = dev(dy0, f32)
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for _ in range(2):
            gx, gc = torch.autograd.grad(_EpsFunction.apply(cs, xs, True), (xs, cs), dys)
    torch.cuda.current_stream().wait_stream(stream)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gx, gc = torch.autograd.grad(_EpsFunction.apply(cs, xs, True), (xs, cs), dys)
    for _ in range(3):
        gx.fill_(float("nan")); gc.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(gx, dx1) and torch.equal(gc, dc1)


@pytest.mark.parametrize("which", ["eps_bigcore_f32", "eps_generic_q3", "convsbs_generic_bond3", "convsbs_ring_many_bond4",
                                   "convsbs_saved_states_bond8", "many_convsbs_band_bond16"])
def test_graph_replays_start_their_accumulators_from_zero(which):
    """Kernels that accumulate into a zero-filled buffer (atomics, slices) must zero it with a kernel of their own: a
    `hipMemsetAsync` recorded into a torch HIP graph filled with garbage from the second replay on
    (tools/memset_capture_check.py).  Forward + backward captured once, replayed three times with the gradient
    buffers dirtied in between, compared with the eager result."""
    torch.manual_seed(3)
    if which.startswith("eps"):
        if which == "eps_bigcore_f32":
            C, B, H, W, Q, K, O = 1, 3, 6, 7, 4, 3, 6
        else:
            C, B, H, W, Q, K, O = 1, 5, 6, 6, 3, 2, 3
        N = K * K * C
        x = torch.randn(C, B, H, W, Q, device=DEV, requires_grad=True)
        params = [(torch.randn(*(Q,) * N, O, device=DEV) * Q ** (-N / 4)).requires_grad_(True)]
        run = lambda: eps(params[0], x)
    elif which == "many_convsbs_band_bond16":
        # the bond-16 two-string layer of the reference's classifier (mnist.py:224-242): one forward launch, the backward
        # kernel + its tail for both strings (dX summed over the strings by the tail)
        from dctn_amd.conv_sbs import ManyConvSBS
        snake_a = ((0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2))
        snake_b = ((0, 0), (1, 0), (2, 0), (2, 1), (1, 1), (0, 1), (0, 2), (1, 2), (2, 2))
        outs = (1, 1, 1, 1, 2, 1, 1, 1, 1)
        specs = tuple(tuple(SBSSpecCore(Pos2D(h, w), o) for (h, w), o in zip(sn, outs)) for sn in (snake_a, snake_b))
        init = DumbNormalInitialization((4 * 16) ** -0.5 * 1.2)
        many = ManyConvSBS(2, 2, 16, False, specs, (init, init)).to(DEV)
        x = torch.randn(2, 3, 7, 8, 2, device=DEV, requires_grad=True)
        params = list(many.parameters())
        run = lambda: torch.cat(many(x), dim=-1)
        run()
        assert dctn_amd.last_kernel() == "convsbs_many_fwd_band_f32"
    else:
        snake = ((0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2))
        if which == "convsbs_generic_bond3":
            bonds, outs = (1,) + (3,) * 8, (1, 1, 1, 1, 2, 1, 1, 1, 1)
        elif which == "convsbs_saved_states_bond8":   # the training forward leaves its states for the backward
            bonds, outs = (1,) + (8,) * 8, (1, 1, 1, 1, 2, 1, 1, 1, 1)
        else:
            bonds, outs = (4,) * 9, (1, 1, 1, 1, 5, 1, 1, 1, 1)
        spec = SBSSpecString(tuple(SBSSpecCore(Pos2D(h, w), o) for (h, w), o in zip(snake, outs)), bonds, 2, 2)
        m = ConvSBS(spec, DumbNormalInitialization(0.4)).to(DEV)
        x = torch.randn(2, 3, 7, 8, 2, device=DEV, requires_grad=True)
        params = list(m.cores)
        run = lambda: m(x)

    def fwd_bwd():
        x.grad = None
        for p in params:
            p.grad = None
        y = run()
        y.backward(torch.ones_like(y))
        return y

    y_e = fwd_bwd().detach().clone()
    want = [x.grad.clone()] + [p.grad.clone() for p in params]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y_g = fwd_bwd()
    for rep in range(3):
        for t in [x] + params:
            t.grad.fill_(7.0)
        y_g.fill_(7.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.allclose(y_g, y_e, rtol=1e-5, atol=1e-6 * float(y_e.abs().max())), rep
        for t, w in zip([x] + params, want):
            assert torch.allclose(t.grad, w, rtol=1e-4, atol=1e-5 * float(w.abs().max())), (rep, tuple(w.shape))


# ------------------------------------------------------------------ linear head (bf16, skinny)
def test_eps_f32_large_core_under_the_bf16_policy():
    """set_float32_matmul_precision("bf16") lets float32 tensors with a large core use the bf16 matrix cores
    (operands rounded to bf16, float32 accumulate); the default policy keeps the exact-f32 family."""
    torch.manual_seed(32)
    C, B, H, W, Q, K, O = 1, 4, 9, 8, 2, 4, 4
    N = K * K * C
    x = torch.randn(C, B, H, W, Q)
    core = torch.randn(*(Q,) * N, O) * Q ** (-N / 4)
    want = R.eps_4step(core.double(), x.double())
    xd, cd = x.to(DEV).requires_grad_(True), core.to(DEV).requires_grad_(True)
    y = eps(cd, xd)
    assert dctn_amd.last_kernel() == "eps_fwd_mfma_bigcore_f32_saving"   # x needs a gradient: the GEMM result is kept
    dctn_amd.set_float32_matmul_precision("bf16")
    try:
        y2 = eps(cd, xd)
        assert y2.dtype == torch.float32 and dctn_amd.last_kernel() == "eps_fwd_mfma_bf16_halves_saving"
        # the policy's definition: the OPERANDS are rounded to bf16 (16 factors of 2^-9 each on a window product), so
        # the oracle is evaluated on the rounded operands; against the unrounded ones only the gross bound holds
        cb, xb = core.bfloat16().double(), x.bfloat16().double()
        assert bf16_close(y2, R.eps_4step(cb, xb))
        assert float((y2.detach().cpu().double() - want).abs().max()) < BF16_TOL * float(want.abs().max())
        dy = torch.randn(*want.shape).bfloat16().float()
        y2.backward(dy.to(DEV))
        assert cd.grad.dtype == torch.float32 and xd.grad.dtype == torch.float32
        gc, gx = R.grads(R.eps_4step, [cb, xb], dy.double())
        assert bf16_close(cd.grad, gc) and bf16_close(xd.grad, gx)
    finally:
        dctn_amd.set_float32_matmul_precision("exact")
    assert float((y.detach().cpu().double() - want).abs().max()) < 3e-4 * float(want.abs().max())


@pytest.mark.parametrize("keep", [True, False], ids=["savedz", "recompute"])
def test_eps_bf16_large_core_runs_on_the_matrix_cores(keep):
    """bf16 tensors with a core outside the bf16 register family (a deeper / wider layer): two-halves GEMMs on
    v_mfma_f32_16x16x32_bf16 (bf16 P0 / P1 / T, float32 accumulate) — not the generic kernels.  keep: the training forward
    leaves P0, P1 and Z' (bf16) for the backward, whose dP1 is then one pass over Z' instead of a GEMM."""
    with keep_gemm_result(keep):
        _eps_bf16_large_core(("_saving", "_savedz") if keep else ("", ""))


def _eps_bf16_large_core(suf):
    torch.manual_seed(31)
    # halves of 256 x 256, 256 x 1024 (odd O) and 64 x 64 (a 128-column tile spans two outputs; 5 outputs)
    for (C, B, H, W, Q, K, O) in ((1, 5, 9, 8, 2, 4, 4), (1, 3, 7, 7, 4, 3, 6), (1, 7, 9, 10, 8, 2, 5)):
        N = K * K * C
        x = torch.randn(C, B, H, W, Q).to(torch.bfloat16)
        core = (torch.randn(*(Q,) * N, O) * Q ** (-N / 4)).to(torch.bfloat16)
        xd, cd = x.to(DEV).requires_grad_(True), core.to(DEV).requires_grad_(True)
        y = eps(cd, xd)
        assert y.dtype == torch.bfloat16 and dctn_amd.last_kernel() == "eps_fwd_mfma_bf16_halves" + suf[0]
        want = R.eps_4step(core.double(), x.double())
        assert bf16_close(y, want)
        dy = torch.randn(*want.shape).to(torch.bfloat16)
        y.backward(dy.to(DEV))
        assert dctn_amd.last_kernel() == "eps_bwd_mfma_bf16_halves" + suf[1]
        assert xd.grad.dtype == torch.bfloat16 and cd.grad.dtype == torch.bfloat16
        gc, gx = R.grads(R.eps_4step, [core.double(), x.double()], dy.double())
        assert bf16_close(cd.grad, gc) and bf16_close(xd.grad, gx)


@pytest.mark.parametrize("B,F,C", [(1024, 2704, 10), (37, 3176, 10), (5, 64, 3), (130, 200, 16)])
def test_linear_head_vs_torch_reference(B, F, C):
    """bf16 operands, f32 accumulation; reference = the same op in float64 on the rounded inputs.
    Tolerance 2e-2 of the output scale (bf16 output rounding 2^-8 plus bf16 products)."""
    from dctn_amd.eps_plus_linear import _LinearHeadFunction

    torch.manual_seed(B + F)
    feat = torch.randn(B, F).bfloat16()
    w = (torch.randn(C, F) / F**0.5).bfloat16()
    bias = torch.randn(C).bfloat16()
    fd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (feat, w, bias))
    assert _LinearHeadFunction.supported(fd, wd, bd)
    out = _LinearHeadFunction.apply(fd, wd, bd)
    assert dctn_amd.last_kernel() == "linear_head_fwd_mfma"
    f64, w64, b64 = (t.double().requires_grad_(True) for t in (feat, w, bias))
    want = torch.nn.functional.linear(f64, w64, b64)
    assert bf16_close(out, want.detach())
    g = torch.randn(B, C).bfloat16()
    want.backward(g.double())
    for mode in ("blas", "hip"):  # library-GEMM backward and the HIP backward kernels (the default)
        import dctn_amd.eps_plus_linear as EPL
        assert EPL.HEAD_BWD == "hip"
        EPL.HEAD_BWD = mode
        try:
            for t in (fd, wd, bd):
                t.grad = None
            _LinearHeadFunction.apply(fd, wd, bd).backward(g.to(DEV))
            if mode == "hip":
                assert dctn_amd.last_kernel() == "linear_head_bwd"
        finally:
            EPL.HEAD_BWD = "hip"
        assert bf16_close(fd.grad, f64.grad) and bf16_close(wd.grad, w64.grad) and bf16_close(bd.grad, b64.grad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16])
@pytest.mark.parametrize("B,F,C", [(128, 3174, 10), (37, 201, 10), (5, 7, 3), (1, 64, 16), (70, 1000, 1)])
def test_linear_head_scalar_kernels_any_dtype_any_feature_count(B, F, C, dtype):
    """float32 / float64 heads (the reference's own arithmetic) and feature counts that are not multiples of 8 (cfg3a:
    23 x 23 x 6 = 3174) on the scalar streaming kernels - no library GEMM on the path; needs_input_grad honoured."""
    from dctn_amd.eps_plus_linear import _LinearHeadFunction

    torch.manual_seed(B + F + C)
    feat = torch.randn(B, F).to(dtype)
    w = (torch.randn(C, F) / F**0.5).to(dtype)
    bias = torch.randn(C).to(dtype)
    fd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (feat, w, bias))
    assert _LinearHeadFunction.supported(fd, wd, bd)
    out = _LinearHeadFunction.apply(fd, wd, bd)
    assert dctn_amd.last_kernel() == ("linear_head_fwd_mfma" if dtype == torch.bfloat16 and F % 8 == 0 else "linear_head_fwd_generic")
    f64, w64, b64 = (t.double().requires_grad_(True) for t in (feat, w, bias))
    want = torch.nn.functional.linear(f64, w64, b64)
    g = torch.randn(B, C).to(dtype)
    want.backward(g.double())
    out.backward(g.to(DEV))
    assert dctn_amd.last_kernel() in ("linear_head_bwd_generic", "linear_head_bwd")
    pairs = ((out, want.detach()), (fd.grad, f64.grad), (wd.grad, w64.grad), (bd.grad, b64.grad))
    if dtype == torch.bfloat16:
        assert all(bf16_close(a, b) for a, b in pairs)
    else:
        assert all(close(a, b, dtype) for a, b in pairs)
    # only dWeight / dBias (a frozen earlier layer), only dFeat (a frozen head)
    f2, w2, b2 = fd.detach().clone(), wd.detach().clone().requires_grad_(True), bd.detach().clone().requires_grad_(True)
    _LinearHeadFunction.apply(f2, w2, b2).backward(g.to(DEV))
    assert f2.grad is None and torch.equal(w2.grad, wd.grad) and torch.equal(b2.grad, bd.grad)
    f3 = fd.detach().clone().requires_grad_(True)
    _LinearHeadFunction.apply(f3, wd.detach(), bd.detach()).backward(g.to(DEV))
    assert torch.equal(f3.grad, fd.grad)


@pytest.mark.parametrize("K,O,size,B", [(3, 4, 28, 5), (3, 4, 8, 19), (3, 2, 12, 9), (3, 4, 40, 3)])
def test_eps_plus_linear_fused_head_backward(K, O, size, B):
    """bf16 single-EPS model (BASELINE config 2's shape family): the last EPS + flatten + linear head
    run as one autograd node whose backward (`dctn_eps_head_bwd`) forms dY on the fly and produces
    dCore, dWeight and dBias in one pass; checked against the float64 oracle and against the unfused
    composition (library GEMMs + dctn_eps_bwd)."""
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd

    torch.manual_seed(100 * K + 10 * O + size)
    m = EPSesPlusLinear(((K, O),), UnitTheoreticalOutputStd(), 1.0, DEV, torch.bfloat16, image_size=size)
    with torch.no_grad():
        m.linear.weight.mul_(8.0)
    u = torch.rand(B, size, size)
    x = torch.stack([torch.sin(u * 1.5707963) ** 2, torch.cos(u * 1.5707963) ** 2], dim=-1)[None].to(torch.bfloat16).to(DEV)
    g = torch.randn(B, 10).to(torch.bfloat16)

    def run(fused):
        import dctn_amd.eps_plus_linear as EPL
        EPL.FUSED_HEAD = fused
        try:
            for prm in m.parameters():
                prm.grad = None
            out = m(x)
            out.backward(g.to(DEV))
            return out.detach().cpu(), [prm.grad.detach().cpu().double() for prm in (m.epses[0], m.linear.weight, m.linear.bias)], dctn_amd.last_kernel()
        finally:
            EPL.FUSED_HEAD = True

    out_f, grads_f, kern_f = run(True)
    out_u, grads_u, kern_u = run(False)
    assert kern_f == "eps_head_bwd_mfma_q2reg" and kern_u != kern_f
    # same features; the fused forward sums the head's products in another order than the stand-alone head kernel
    assert float((out_f.float() - out_u.float()).abs().max()) <= 2 ** -7 * float(out_u.float().abs().max())
    core64 = m.epses[0].detach().cpu().double().requires_grad_(True)
    w64 = m.linear.weight.detach().cpu().double().requires_grad_(True)
    b64 = m.linear.bias.detach().cpu().double().requires_grad_(True)
    want = R.eps_plus_linear_forward([core64], w64, b64, x.cpu().double())
    assert bf16_close(out_f, want.detach())
    want.backward(g.double())
    for name, got_f, got_u, ref in zip(("dCore", "dWeight", "dBias"), grads_f, grads_u, (core64.grad, w64.grad, b64.grad)):
        assert bf16_close(got_f, ref), name
        assert bf16_close(got_u, ref), name + " (unfused)"


@pytest.mark.parametrize("C,K,O,size,B,cout", [(1, 3, 4, 28, 5, 10), (1, 3, 4, 8, 19, 10), (1, 3, 2, 12, 9, 16), (1, 3, 3, 11, 4, 10),
                                                (2, 2, 4, 9, 5, 6), (2, 2, 2, 7, 3, 1), (1, 3, 4, 40, 3, 10), (1, 3, 4, 28, 70, 10)])
def test_eps_plus_linear_float32_register_family(C, K, O, size, B, cout):
    """The headline model in the reference's own arithmetic (float32: new_runner.py:417): layer + flatten + head as one
    autograd node on the register-resident exact-f32 family (`eps_q2f32.hip`: v_mfma_f32_32x32x2_f32, dY formed in the
    dCore kernel, dW / dBias by the finishing kernel), against the float64 oracle at the float32 tolerance and
    against the unfused composition; out sizes that are no power of two (3) take the unfused node on the same family."""
    import dctn_amd.eps_plus_linear as EPL

    torch.manual_seed(1000 * C + 100 * K + 10 * O + size)
    core = (torch.randn(*(2,) * (K * K * C), O) * 2.0 ** (-K * K * C / 2 + 1)).to(DEV).requires_grad_(True)
    side = size - K + 1
    w = (torch.randn(cout, side * side * O) * 0.05).to(DEV).requires_grad_(True)
    bias = torch.randn(cout).to(DEV).requires_grad_(True)
    u = torch.rand(C, B, size, size)
    x = torch.stack([torch.sin(u * 1.5707963) ** 2, torch.cos(u * 1.5707963) ** 2], dim=-1).to(DEV)
    g = torch.randn(B, cout)

    def run(fused):
        EPL.FUSED_HEAD = fused
        try:
            for t in (core, w, bias):
                t.grad = None
            if EPL._EpsLinearHeadFunction.supported(core, x, w, bias):
                out = EPL._EpsLinearHeadFunction.apply(core, x, w, bias)
            else:
                feat = eps(core, x)
                out = EPL._LinearHeadFunction.apply(feat.reshape(B, -1), w, bias)
            out.backward(g.to(DEV))
            return out.detach().cpu(), [t.grad.detach().cpu() for t in (core, w, bias)], dctn_amd.last_kernel()
        finally:
            EPL.FUSED_HEAD = True

    out_f, grads_f, kern_f = run(True)
    out_u, grads_u, kern_u = run(False)
    if O in (2, 4):
        assert kern_f == "eps_head_bwd_q2f32", kern_f
    assert kern_u != "eps_head_bwd_q2f32"
    assert _lib.lib().dctn_eps_family(C, B, size, size, 2, K, O, _lib.F32, 0) == 4
    c64, w64, b64 = (t.detach().cpu().double().requires_grad_(True) for t in (core, w, bias))
    want = R.eps_plus_linear_forward([c64], w64, b64, x.cpu().double())
    want.backward(g.double())
    assert close(out_f, want.detach(), torch.float32) and close(out_u, want.detach(), torch.float32)
    for name, got_f, got_u, ref in zip(("dCore", "dWeight", "dBias"), grads_f, grads_u, (c64.grad, w64.grad, b64.grad)):
        assert close(got_f, ref, torch.float32), name
        assert close(got_u, ref, torch.float32), name + " (unfused)"
    # a second call reproduces the first bit for bit (fixed-order sums, no float atomics)
    out_2, grads_2, _ = run(True)
    assert torch.equal(out_2, out_f) and all(torch.equal(a, b) for a, b in zip(grads_2, grads_f))


def test_eps_float32_register_family_strided_input_and_input_gradient():
    """The same family behind `eps()`: a strided batch slice (dctn/eps.py:136 hands `x.split(batch, dim=1)` pieces) and a
    permuted input (generic window loads), with the input's gradient taken by the large-core family."""
    torch.manual_seed(7)
    core = (torch.randn(*(2,) * 9, 4) * 0.1).to(DEV).requires_grad_(True)
    big = torch.rand(1, 9, 12, 12, 2).to(DEV)
    for x, keep in ((big[:, 2:7], False), (big.permute(0, 1, 3, 2, 4)[:, 1:4], False), (big[:, 2:7], True)):
        x = x.detach().requires_grad_(True)
        core.grad = None
        with keep_gemm_result(keep):
            y = eps(core, x)
        # a training forward that keeps its GEMM result for the input gradient stays on the large-core family (whose
        # backward reads the buffer); without the buffer the register family takes the forward and dCore
        assert dctn_amd.last_kernel() == ("eps_fwd_mfma_bigcore_f32_saving" if keep else "eps_fwd_q2f32")
        dy = torch.randn_like(y)
        y.backward(dy)
        c64, x64 = core.detach().cpu().double().requires_grad_(True), x.detach().cpu().double().requires_grad_(True)
        want = R.eps_4step(c64, x64)
        want.backward(dy.cpu().double())
        assert close(y, want.detach(), torch.float32)
        assert close(core.grad, c64.grad, torch.float32) and close(x.grad, x64.grad, torch.float32)


def test_logmatmulexp_fold16_factored_mfma_and_exact_fallback():
    """D = 16 float32 fold: factored exp -> MFMA -> log forward with the exact path taken per step
    when the dynamic range is unsafe (large magnitudes, -inf entries)."""
    torch.manual_seed(16)
    m = torch.randn(300, 9, 16, 16)
    m[7] *= 60.0                      # ranges far above 40: exact fallback every step
    m[11, 3, 5, :] = -float("inf")    # a -inf row in one factor
    m[13, 0] = -float("inf")          # a whole -inf matrix: result -inf
    md = m.to(DEV)
    y = logmatmulexp_fold(md)
    assert dctn_amd.last_kernel() == "logmatmulexp_fold_fwd_mfma16"
    want = R.logmatmulexp_fold_batched(m.double())
    yc = y.cpu().double()
    assert torch.equal(torch.isinf(yc), torch.isinf(want))
    fin = torch.isfinite(want)
    err = ((yc[fin] - want[fin]).abs() / (1.0 + want[fin].abs())).max()
    assert float(err) < 5e-5, float(err)
    # backward: factored MFMA kernel; windows it cannot represent (range > 40, -inf) are flagged and
    # redone by the exact recomputing kernel
    m2 = torch.randn(70, 9, 16, 16)
    m2[5] *= 30.0
    m2[9, 4, 2, :] = -float("inf")
    m2[33, 8] *= 25.0
    m2d = m2.to(DEV).requires_grad_(True)
    y2 = logmatmulexp_fold(m2d)
    dy = torch.randn(70, 16, 16)
    y2.backward(dy.to(DEV))
    assert dctn_amd.last_kernel() == "logmatmulexp_fold_bwd_mfma16"
    (gm,) = R.grads(R.logmatmulexp_fold_batched, [m2.double()], dy.double())
    got = m2d.grad.cpu().double()
    assert torch.isfinite(got).all()
    for wdw in range(70):
        scale = gm[wdw].abs().max().clamp_min(1.0)
        assert float((got[wdw] - gm[wdw]).abs().max() / scale) < 2e-4, wdw


@pytest.mark.parametrize("nb,T,Rr,I", [(1, 256, 256, 256), (3, 100, 70, 130), (2, 64, 16, 64), (5, 33, 40, 65)])
def test_logmatmulexp_factored_gemm_and_exact_fallback(nb, T, Rr, I):
    """float32 products that are not tiny: exp -> MFMA GEMM -> log.  Ragged tiles, -inf rows / columns,
    dynamic ranges the factorisation cannot hold (output tiles / batch elements redone by the direct
    kernels) — torch.logsumexp semantics throughout."""
    from dctn_amd.logmatmulexp import logmatmulexp_batched
    torch.manual_seed(nb * 1000 + T)
    a, b = torch.randn(nb, T, Rr) * 3, torch.randn(nb, Rr, I) * 3
    a[0, 5, 3] = -float("inf")              # isolated -inf entries
    b[0, 4, 7] = -float("inf")
    if nb > 1:
        a[1, :, 0] += 300.0                  # one dominant term per row, far beyond the factored range
        b[1, 0, :] -= 300.0
        b[1, 1, ::2] += 250.0
    a64, b64 = a.double(), b.double()

    def reference(a64, b64):
        return torch.stack([R.logmatmulexp(a64[n], b64[n]) for n in range(nb)])

    # forward, with a -inf row and a -inf column on top (-inf outputs; their gradient is NaN in the
    # reference as well, so the backward comparison below runs without them)
    af, bf = a.clone(), b.clone()
    af[0, 1, :] = -float("inf")
    bf[0, :, 2] = -float("inf")
    yf = logmatmulexp_batched(af.to(DEV), bf.to(DEV)).cpu().double()
    assert dctn_amd.last_kernel() == "logmatmulexp_fwd_mfma_gemm"
    wf = reference(af.double(), bf.double())
    assert torch.equal(torch.isinf(yf), torch.isinf(wf)) and not torch.isnan(yf).any()
    fin = torch.isfinite(wf)
    assert float(((yf[fin] - wf[fin]).abs() / (1.0 + wf[fin].abs())).max()) < 2e-6
    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = logmatmulexp_batched(ad, bd)
    want = reference(a64, b64)
    assert float(((y.detach().cpu().double() - want).abs() / (1.0 + want.abs())).max()) < 2e-6
    dy = torch.randn(nb, T, I)
    y.backward(dy.to(DEV))
    assert dctn_amd.last_kernel() == "logmatmulexp_bwd_mfma_gemm"
    # float64 gradients from the closed form
    wgt = torch.exp(a64.unsqueeze(3) + b64.unsqueeze(1) - want.unsqueeze(2)) * dy.double().unsqueeze(2)
    ga, gb = wgt.sum(3), wgt.sum(1)
    for name, got, ref in (("dA", ad.grad, ga), ("dB", bd.grad, gb)):
        got = got.cpu().double()
        assert torch.isfinite(got).all(), name
        for n in range(nb):
            assert float((got[n] - ref[n]).abs().max()) < 2e-5 * float(ref[n].abs().max().clamp_min(1.0)), (name, n)
    # NaN propagates like logsumexp's: only the outputs that see it
    a2 = a.clone()
    a2[0, 9, 4] = float("nan")
    y2 = logmatmulexp_batched(a2.to(DEV), bd.detach()).cpu()
    assert torch.isnan(y2[0, 9]).all() and int(torch.isnan(y2).sum()) == I


# ------------------------------------------------------------------ window statistics
def test_window_statistics_kernel_against_reference_fixture_and_oracle():
    """`dctn_window_stats` (one pass over the images, no window tensor) against the numbers the
    reference's make_windows + RankOneTensorsBatch produced, then dtype / stride variants against the
    oracle, then the defining property of calc_scaling_factor."""
    from dctn_amd.align import make_windows
    from dctn_amd.window_stats import (apply_feature_map, calc_scaling_factor, window_mean_var, window_sums)

    g = load("window_stats")
    for tag in "abcd":
        x, K = dev(g[f"x_{tag}"]), int(g[f"K_{tag}"])
        sums = window_sums(x, K).cpu()
        assert dctn_amd.last_kernel() == "window_stats"
        assert np.isclose(float(sums[0]), float(g[f"sum_{tag}"]), rtol=1e-12)
        assert np.isclose(float(sums[1]), float(g[f"sq_{tag}"]), rtol=1e-12)
        mean, var = window_mean_var(x, K)
        assert np.isclose(float(mean), float(g[f"mean_{tag}"]), rtol=1e-11)
        assert np.isclose(float(var), float(g[f"var_{tag}"]), rtol=1e-9)
        assert np.isclose(calc_scaling_factor(x, K), float(g[f"factor_{tag}"]), rtol=1e-10)
    torch.manual_seed(33)
    images = torch.rand(300, 28, 28)
    x = apply_feature_map(images)                               # (1, 300, 28, 28, 2), the reference's layout
    assert x.shape == (1, 300, 28, 28, 2)
    for K in (3, 4):
        want = R.window_mean_var_factor(x, K)
        for dtype, rtol in ((torch.float64, 1e-11), (torch.float32, 1e-6), (torch.bfloat16, 2e-2)):
            xd = x.to(dtype).to(DEV)
            mean, var = window_mean_var(xd, K)
            ref = R.window_mean_var_factor(x.to(dtype), K) if dtype != torch.float64 else want
            assert np.isclose(float(mean), float(ref[0]), rtol=rtol) and np.isclose(float(var), float(ref[1]), rtol=10 * rtol)
        # strided view (every other sample, a crop): strides go through the C-ABI
        big = torch.rand(2, 9, 12, 13, 3, dtype=torch.float64)
        view = big[:, ::2, 1:11, 2:12]
        s_view = window_sums(big.to(DEV)[:, ::2, 1:11, 2:12], K)
        t, q = R.window_sums(view, K)
        assert np.isclose(float(s_view[0]), float(t), rtol=1e-12) and np.isclose(float(s_view[1]), float(q), rtol=1e-12)
        # after scaling, the windows' rank-one tensors have mean^2 + variance == 1
        f = calc_scaling_factor(x, K, DEV)
        m2, v2 = window_mean_var((x.double() * f).to(DEV), K)
        assert abs(float(m2**2 + v2) - 1.0) < 1e-9
        # and the materialising host mirror agrees (K*K copies of the data)
        w = make_windows(x.double(), K)
        assert np.isclose(float(w.mean_over_batch()), float(want[0]), rtol=1e-11)
    # a CPU tensor is staged to the device and the sums come back on the CPU (same kernel, same numbers)
    s_cpu = window_sums(x.double(), 3)
    assert s_cpu.device.type == "cpu"   # (float64 atomics: the summation order varies from launch to launch)
    assert torch.allclose(s_cpu, window_sums(x.double().to(DEV), 3).cpu(), rtol=1e-12, atol=0)


def test_feature_map_and_its_window_statistics_on_the_device():
    """phi_cos_sin_squared_1 (dctn/dataset_loading.py:33-36) on the device - inside the statistics kernel
    (`dctn_phi_window_stats`: raw images in, the two sums out, no expanded tensor, no stacked windows) and as one
    write of the scaled data-set tensor in the model's dtype (`dctn_phi_expand`) - against the reference's own float32
    lambdas and `calc_scaling_factor` on their output."""
    from dctn_amd.window_stats import (apply_feature_map, apply_feature_map_on_device, calc_scaling_factor,
                                       calc_scaling_factor_from_images, window_mean_var)

    torch.manual_seed(34)
    images = torch.rand(300, 28, 28)
    images[0, 0, :4] = torch.tensor([0.0, 1.0, 0.5, 0.25])
    x_ref = apply_feature_map(images)                 # the reference's lambdas, float32 on the CPU
    for dtype, tol in ((torch.float32, 2e-6), (torch.float64, 2e-6), (torch.bfloat16, 2 ** -7)):
        x = apply_feature_map_on_device(images, 0.75, dtype, DEV)
        assert dctn_amd.last_kernel() == "phi_expand" and x.shape == (1, 300, 28, 28, 2) and x.dtype == dtype
        assert float((x.cpu().double() - 0.75 * x_ref.double()).abs().max()) <= tol * 2.0
    for K in (2, 3, 4):
        want = calc_scaling_factor(x_ref, K, DEV)     # float32 phi values, float64 statistics (dataset_loading.py:82)
        got = calc_scaling_factor_from_images(images, K, DEV)
        assert dctn_amd.last_kernel() == "phi_window_stats"
        assert abs(got / want - 1.0) < 1e-6, (K, got, want)
        m2, v2 = window_mean_var(apply_feature_map_on_device(images, got, torch.float64, DEV), K)
        assert abs(float(m2**2 + v2) - 1.0) < 1e-5
    # an image too large for the kernel's per-pixel table takes the expanding fallback, same number
    big = torch.rand(3, 120, 110)
    assert abs(calc_scaling_factor_from_images(big, 3, DEV) / calc_scaling_factor(apply_feature_map(big), 3, DEV) - 1.0) < 1e-6


def test_log_intermediate_reps_stats_reports_the_reference_window_statistics(caplog):
    """EPSesPlusLinear.log_intermediate_reps_stats (dctn/eps_plus_linear.py:161-196): the w_n lines carry the
    mean / std of the K x K windows as rank-one tensors — here from the one-pass kernel, checked against the
    RankOneTensorsBatch formulas on the stacked windows."""
    import logging
    import re

    from dctn_amd.align import make_windows
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd

    torch.manual_seed(41)
    model = EPSesPlusLinear(((2, 3), (2, 4)), UnitTheoreticalOutputStd(), 1.0, DEV, torch.float32, image_size=9)
    u = torch.rand(1, 6, 9, 9)
    x = torch.stack((torch.sin(u * torch.pi / 2) ** 2, torch.cos(u * torch.pi / 2) ** 2), dim=-1).to(DEV)
    with caplog.at_level(logging.INFO):
        model.log_intermediate_reps_stats(x, batch_size=4)
    lines = [r.getMessage() for r in caplog.records]
    names = [ln.split(":")[0] for ln in lines[1:]]
    assert names == ["x_0", "w_0", "x_1", "w_1", "x_2", "output_of_linear_without_bias", "output_of_linear_with_bias"]
    w0 = next(ln for ln in lines if ln.startswith("w_0"))
    mu, sigma = (float(v) for v in re.findall(r"(?:mu|sigma)=([-+0-9.e]+)", w0)[:2])
    ref = make_windows(x.cpu().double(), 2)
    assert abs(mu - float(ref.mean_over_batch())) < 1e-6 * abs(mu)
    assert abs(sigma - float(ref.std_over_batch())) < 1e-6 * sigma
    assert "batch_shape=(6, 8, 8)" in w0 and "num_factors=4" in w0 and "num_coordinates_in_one_factor=2" in w0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_train_step_end_to_end_loss_goes_down(dtype):
    """The reference's training iteration (dctn/training.py:77-84) on the HIP path: forward, CE loss +
    L2 regulariser, backward, optimizer step — a few iterations on a separable synthetic task must
    reduce the loss (float32: large-core family; bf16: register family with the fused head backward)."""
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd
    from dctn_amd.evaluation import score
    from dctn_amd.training import train_step
    from dctn_amd.window_stats import apply_feature_map, calc_scaling_factor

    torch.manual_seed(7)
    n = 256
    y = torch.randint(0, 2, (n,))
    images = torch.rand(n, 12, 12) * 0.2 + y[:, None, None] * 0.6       # two brightness classes
    x = apply_feature_map(images)
    x = (x * calc_scaling_factor(x, 3, DEV)).to(dtype).to(DEV)
    yd = y.to(DEV)
    model = EPSesPlusLinear(((3, 4),), UnitTheoreticalOutputStd(), 1.0, DEV, dtype, image_size=12)
    params = [p for p in model.parameters()]
    master = [p.detach().float().clone().requires_grad_(True) for p in params] if dtype == torch.bfloat16 else None
    opt = torch.optim.Adam(master if master is not None else params, lr=3e-3 if dtype == torch.float32 else 1e-2)

    def step():
        if master is None:
            return train_step(model, x, yd, torch.nn.functional.cross_entropy, opt,
                              reg_fn=lambda m: m.epswise_l2_regularizer(), reg_coeff=1e-4)
        # bf16 weights with float32 master copies (plain mixed-precision recipe)
        res = train_step(model, x, yd, torch.nn.functional.cross_entropy, _NoStep(), reg_fn=None)
        for m, p in zip(master, params):
            m.grad = p.grad.float()
        opt.step()
        with torch.no_grad():
            for m, p in zip(master, params):
                p.copy_(m.to(p.dtype))
        return res

    class _NoStep:
        def zero_grad(self, set_to_none=True):
            for p in params:
                p.grad = None

        def step(self):
            pass

    first = float(step()["loss"])
    for _ in range(40):
        last = float(step()["loss"])
    assert last < 0.7 * first, (first, last)
    loss, acc = score(model, [(x, yd, None)], DEV)
    assert acc > 0.8 and abs(loss - last) < 0.5


def test_graphed_train_step_matches_eager():
    """`GraphedTrainStep` (forward, CE, regulariser, backward, SGD step replayed from one HIP graph)
    leaves the parameters where the eager iteration leaves them."""
    import copy

    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd
    from dctn_amd.training import GraphedTrainStep, train_step

    torch.manual_seed(11)
    a = EPSesPlusLinear(((3, 4),), UnitTheoreticalOutputStd(), 1.0, DEV, torch.float32, image_size=10)
    b = copy.deepcopy(a)
    xs = [torch.rand(1, 32, 10, 10, 2, device=DEV) for _ in range(6)]
    ys = [torch.randint(0, 10, (32,), device=DEV) for _ in range(6)]
    reg = lambda m: m.epswise_l2_regularizer()
    oa = torch.optim.SGD(a.parameters(), lr=0.05, momentum=0.9)
    ob = torch.optim.SGD(b.parameters(), lr=0.05, momentum=0.9)
    # the graphed step warms up with three real iterations on the example batch: give model `a` the same
    graphed = GraphedTrainStep(b, xs[0], ys[0], torch.nn.functional.cross_entropy, ob, reg_fn=reg, reg_coeff=1e-3,
                               warmup=3)
    for _ in range(3):
        train_step(a, xs[0], ys[0], torch.nn.functional.cross_entropy, oa, reg_fn=reg, reg_coeff=1e-3)
    for x, y in zip(xs, ys):
        ra = train_step(a, x, y, torch.nn.functional.cross_entropy, oa, reg_fn=reg, reg_coeff=1e-3)
        rb = graphed(x, y)
        assert torch.allclose(ra["loss"], rb["loss"], rtol=1e-5, atol=1e-6)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("bond,ring", [(4, False), (4, True), (2, False)])
def test_convsbs_classifier_step_eager_and_graphed(bond, ring):
    """The reference's ConvSBS classifier (mnist.py:170-262: two-string layers, a final string with ten labels on its
    middle core): open chains run on the register sweep - the ten-label string too -, rings on the matrix-core sweep (ring
    slices, slices of the many-valued core); the whole training iteration is capturable (no synchronisation, no host read-back on
    the path) and the graphed iteration leaves the parameters where the eager one leaves them."""
    import copy

    from dctn_amd.training import GraphedTrainStep, train_step

    A = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
    Bs = [(0, 0), (1, 0), (2, 0), (2, 1), (1, 1), (0, 1), (0, 2), (1, 2), (2, 2)]

    def string(pos, mid):
        return tuple(SBSSpecCore(Pos2D(*p), mid if i == 4 else 1) for i, p in enumerate(pos))

    class Classifier(torch.nn.Module):
        def __init__(self):
            super().__init__()
            init = DumbNormalInitialization((2 * bond) ** -0.5 * 1.3)
            two = (string(A, 2), string(Bs, 2))
            self.layers = torch.nn.ModuleList([
                ManyConvSBS(1, 2, bond, ring, two, (init,) * 2),
                ManyConvSBS(2, 2, bond, ring, two, (init,) * 2),
                ManyConvSBS(2, 2, bond, ring, (string(A, 10),), (init,)),
            ])

            self.scales = [1.0, 1.0, 1.0]   # fixed per-layer output scales (calibrated below): 27 cores would underflow

        def forward(self, x):   # x: (1, B, H, W, 2)
            inter = (x[0],)
            for layer, scale in zip(self.layers, self.scales):   # (tanh: products of 9 random factors are heavy-tailed)
                inter = tuple(torch.tanh(o * scale) for o in layer(inter))
            (out,) = inter
            return out.reshape(out.shape[0], -1, out.shape[-1]).mean(1)   # (B, 10)

        def calibrate(self, x):
            with torch.no_grad():
                inter = (x[0],)
                for k, layer in enumerate(self.layers):
                    outs = layer(inter)
                    # (products of many random factors are heavy-tailed: the median of |o| is the typical size)
                    self.scales[k] = 1.0 / float(torch.cat([o.reshape(-1) for o in outs]).abs().median())
                    inter = tuple(torch.tanh(o * self.scales[k]) for o in outs)

    torch.manual_seed(5)
    a = Classifier().to(DEV)
    xs = [torch.rand(1, 8, 8, 8, 2, device=DEV) for _ in range(4)]
    ys = [torch.randint(0, 10, (8,), device=DEV) for _ in range(4)]
    a.calibrate(xs[0])
    b = copy.deepcopy(a)
    y0 = a(xs[0])
    # the last string to run is the ten-label one: open chains of bond <= 4 take it on the register sweep (prefix state,
    # suffix vector, one dot product per label: convsbs_reg.hip); rings run as slices of the matrix-core sweep
    want_family = "convsbs_fwd_mfma_f32" if ring else "convsbs_fwd_reg_f32"
    assert dctn_amd.last_kernel() == want_family and y0.shape == (8, 10)
    assert 1e-3 < float(y0.detach().abs().median()) < 1e3, float(y0.detach().abs().median())   # a live model
    # (a small step: the gradients of a 27-core product are large and an unnormalised ring model diverges quickly; the
    # point here is that both forms of the iteration compute the same thing)
    oa = torch.optim.SGD(a.parameters(), lr=1e-6)
    ob = torch.optim.SGD(b.parameters(), lr=1e-6)
    ce = torch.nn.functional.cross_entropy
    graphed = GraphedTrainStep(b, xs[0], ys[0], ce, ob, warmup=3)
    for _ in range(3):
        train_step(a, xs[0], ys[0], ce, oa)
    for x, y in zip(xs, ys):
        ra = train_step(a, x, y, ce, oa)
        rb = graphed(x, y)
        assert torch.isfinite(ra["loss"]) and torch.allclose(ra["loss"], rb["loss"], rtol=1e-4, atol=1e-6)
    moved = 0.0
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.isfinite(pa).all() and torch.allclose(pa, pb, rtol=1e-4, atol=1e-7)
        moved = max(moved, float(pa.grad.abs().max()))
    assert moved > 0.0   # the iterations did train something


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_training_tail_matches_torch(dtype):
    """`fused_cross_entropy` against F.cross_entropy (value and gradient), and `FlatSGD` (momentum + the
    L2 regulariser folded into the update, one kernel over a flat parameter buffer) against
    torch.optim.SGD driven by loss + l2 * epswise_l2_regularizer through autograd."""
    import copy

    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd
    from dctn_amd.training import FlatSGD, fused_cross_entropy, train_step

    torch.manual_seed(21)
    logits = (torch.randn(77, 10) * 3).to(dtype).to(DEV).requires_grad_(True)
    labels = torch.randint(0, 10, (77,), device=DEV)
    loss = fused_cross_entropy(logits, labels)
    (loss * 1.7).backward()
    ref_logits = logits.detach().float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_logits, labels)
    (ref * 1.7).backward()
    assert loss.dtype == torch.float32 and abs(float(loss.detach()) - float(ref.detach())) < 1e-5 * max(1.0, abs(float(ref.detach())))
    tol = 1e-6 if dtype == torch.float32 else 2e-2 * float(ref_logits.grad.abs().max())
    assert float((logits.grad.float() - ref_logits.grad).abs().max()) <= tol

    # the registered constant-1 seed (what GraphedTrainStep passes): the forward's unit gradient comes back as is;
    # and a batch beyond the single-workgroup bound (fill + atomics form of the forward kernel)
    from dctn_amd.training import unit_seed
    for rows in (77, 9000):
        lg = (torch.randn(rows, 10) * 3).to(dtype).to(DEV).requires_grad_(True)
        lb = torch.randint(0, 10, (rows,), device=DEV)
        ls = fused_cross_entropy(lg, lb)
        ls.backward(unit_seed(DEV, torch.float32))
        rl = lg.detach().float().requires_grad_(True)
        rf = torch.nn.functional.cross_entropy(rl, lb)
        rf.backward()
        assert abs(float(ls.detach()) - float(rf.detach())) < 2e-5 * max(1.0, abs(float(rf.detach())))
        tol2 = 1e-6 if dtype == torch.float32 else 2e-2 * float(rl.grad.abs().max())
        assert float((lg.grad.float() - rl.grad).abs().max()) <= tol2

    # rows labelled -100 (F.cross_entropy's default ignore_index): skipped, zero gradient rows, not counted in the mean -
    # in the one-workgroup form and in the several-workgroups form; all rows ignored: NaN, as torch
    for rows in (50, 9000):
        lg = (torch.randn(rows, 10) * 3).to(dtype).to(DEV).requires_grad_(True)
        lb = torch.randint(0, 10, (rows,), device=DEV)
        lb[::3] = -100
        ls = fused_cross_entropy(lg, lb)
        (ls * 0.5).backward()
        rl = lg.detach().float().requires_grad_(True)
        rf = torch.nn.functional.cross_entropy(rl, lb)
        (rf * 0.5).backward()
        assert abs(float(ls.detach()) - float(rf.detach())) < 2e-5 * max(1.0, abs(float(rf.detach())))
        tol3 = 1e-6 if dtype == torch.float32 else 2e-2 * float(rl.grad.abs().max())
        assert float((lg.grad.float() - rl.grad).abs().max()) <= tol3
        assert float(lg.grad[::3].abs().max()) == 0.0
    assert torch.isnan(fused_cross_entropy(lg.detach()[:4], torch.full((4,), -100, device=DEV)))

    if dtype == torch.bfloat16:
        return   # the optimizer comparison below needs float32 weights on the torch side
    a = EPSesPlusLinear(((3, 4),), UnitTheoreticalOutputStd(), 1.0, DEV, dtype, image_size=10)
    b = copy.deepcopy(a)
    l2, lr, mom = 1e-2, 0.05, 0.9
    oa = torch.optim.SGD(a.parameters(), lr=lr, momentum=mom)
    ob = FlatSGD(list(b.epses) + [b.linear.weight], [b.linear.bias], lr=lr, momentum=mom, l2=l2)
    for it in range(5):
        x = torch.rand(1, 16, 10, 10, 2, device=DEV)
        y = torch.randint(0, 10, (16,), device=DEV)
        ra = train_step(a, x, y, torch.nn.functional.cross_entropy, oa, reg_fn=lambda m: m.epswise_l2_regularizer(),
                        reg_coeff=l2)
        rb = train_step(b, x, y, fused_cross_entropy, ob)
        assert abs(float(ra["loss"]) - float(rb["loss"])) < 1e-4
        assert abs(float(ra["reg_term"]) * l2 - float(ob.reg_value())) < 1e-4 * max(1.0, float(ra["reg_term"]) * l2)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=2e-4, atol=2e-6)


def test_fused_cross_entropy_invalid_label_poisons_loss_and_gradient():
    """F.cross_entropy raises (device-asserts) on a label outside [0, C); the fused kernels cannot raise, so they
    return NaN for the loss and for that sample's gradient row instead of a plausible number."""
    from dctn_amd.training import fused_cross_entropy

    logits = torch.randn(6, 10, device=DEV, requires_grad=True)
    labels = torch.tensor([1, 2, 3, 10, 0, 5], device=DEV)
    loss = fused_cross_entropy(logits, labels)
    assert torch.isnan(loss)
    loss.backward()
    bad = torch.isnan(logits.grad).all(dim=1).cpu()
    assert bad.tolist() == [False, False, False, True, False, False]
    # an incoming gradient other than the registered unit seed goes through dctn_ce_loss_bwd: same poisoning
    logits.grad = None
    (2.0 * fused_cross_entropy(logits, labels)).backward()
    assert torch.isnan(logits.grad[3]).all() and torch.isfinite(logits.grad[[0, 1, 2, 4, 5]]).all()


@pytest.mark.parametrize("C,K,O,size,B,cout", [(1, 3, 4, 28, 37, 10), (1, 3, 4, 28, 1024, 10), (1, 3, 2, 12, 9, 16), (2, 2, 4, 9, 5, 6),
                                               (1, 3, 4, 30, 3, 10), (1, 3, 4, 40, 2, 10)])
def test_fused_forward_of_layer_and_head_matches_the_two_kernels(C, K, O, size, B, cout):
    """`dctn_eps_head_fwd` (layer + flatten + linear head in one kernel, a workgroup = all positions of a few samples)
    against dctn_eps_fwd + dctn_linear_head_fwd: the stored features must be bit-identical (same arithmetic), the
    logits equal up to the order of the float32 sums; shapes with more than 768 positions fall back (40x40)."""
    from dctn_amd import _lib as L

    torch.manual_seed(1000 * K + 10 * O + size + B)
    N = K * K * C
    u = torch.rand(C, B, size, size)
    x = torch.stack((torch.sin(u * torch.pi / 2) ** 2, torch.cos(u * torch.pi / 2) ** 2), dim=-1).bfloat16().to(DEV)
    core = (torch.randn(*(2,) * N, O) * 2 ** (-N / 4)).bfloat16().to(DEV)
    Ho = size - K + 1
    F = Ho * Ho * O
    w = (torch.randn(cout, F) * F ** -0.5).bfloat16().to(DEV)
    bias = torch.randn(cout).bfloat16().to(DEV)
    lib, code = L.lib(), L.dtype_code(x)
    feat_a, feat_b = torch.empty(B, F, dtype=torch.bfloat16, device=DEV), torch.zeros(B, F, dtype=torch.bfloat16, device=DEV)
    log_a, log_b = torch.empty(B, cout, dtype=torch.bfloat16, device=DEV), torch.zeros(B, cout, dtype=torch.bfloat16, device=DEV)
    ws = L.workspace(lib.dctn_eps_fwd_workspace_bytes(C, B, size, size, 2, K, O, code, 0), DEV)
    L.check(lib.dctn_eps_fwd(x.data_ptr(), L.strides5(x), core.data_ptr(), feat_a.data_ptr(), ws.data_ptr(), ws.numel(),
                             C, B, size, size, 2, K, O, code, 0, L.stream_ptr(DEV)), "fwd")
    rc = lib.dctn_eps_head_fwd(x.data_ptr(), L.strides5(x), core.data_ptr(), w.data_ptr(), bias.data_ptr(), feat_b.data_ptr(),
                               log_b.data_ptr(), C, B, size, size, 2, K, O, cout, code, 0, L.stream_ptr(DEV))
    if Ho * Ho > 768:
        assert rc == L.ERR_UNSUPPORTED
        return
    L.check(rc, "fused forward")
    assert dctn_amd.last_kernel() == "eps_head_fwd_mfma_q2reg"
    torch.cuda.synchronize()
    assert torch.equal(feat_a, feat_b)
    want = torch.nn.functional.linear(feat_a.double(), w.double(), bias.double())
    assert bf16_close(log_b, want.cpu())
    if F % 8 == 0:   # the stand-alone head kernel's own shape constraint
        L.check(lib.dctn_linear_head_fwd(feat_a.data_ptr(), w.data_ptr(), bias.data_ptr(), log_a.data_ptr(), B, F, cout, code,
                                         L.stream_ptr(DEV)), "head")
        assert float((log_a.float() - log_b.float()).abs().max()) <= 2 ** -7 * float(want.abs().max())


def test_flat_sgd_repointed_parameters_at_odd_offsets_still_run():
    """FlatSGD re-points every parameter into one flat buffer; a core whose byte size is not a multiple of 16 leaves
    `linear.weight` misaligned for the 16-byte loads of the head kernels: the model must then take the library GEMM
    (not raise), with the same numbers."""
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd
    from dctn_amd.training import FlatSGD

    torch.manual_seed(4)
    m = EPSesPlusLinear(((3, 4),), UnitTheoreticalOutputStd(), 1.0, DEV, torch.bfloat16, image_size=10, Q_0=3)
    x = torch.rand(1, 5, 10, 10, 3, device=DEV).bfloat16()
    before = m(x).detach().float()
    FlatSGD(list(m.epses) + [m.linear.weight], [m.linear.bias], lr=0.0)
    assert m.linear.weight.data_ptr() % 16 != 0        # 3^9 * 4 * 2 bytes = 157 464: the case the review named
    after = m(x)
    assert torch.allclose(after.detach().float(), before, rtol=2e-2, atol=1e-3)
    after.float().sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad.float()).all() for p in m.parameters())
    # the cfg2 family (Q_0 = 2) with a misaligned weight: the fused node is skipped, the separate kernels / GEMM run
    m2 = EPSesPlusLinear(((3, 2),), UnitTheoreticalOutputStd(), 1.0, DEV, torch.bfloat16, image_size=9)
    x2 = torch.rand(1, 4, 9, 9, 2, device=DEV).bfloat16()
    ref = m2(x2).detach().float()
    flat = torch.empty(m2.linear.weight.numel() + 8, dtype=torch.bfloat16, device=DEV)
    view = flat[1 : 1 + m2.linear.weight.numel()].view_as(m2.linear.weight)
    view.copy_(m2.linear.weight.data)
    m2.linear.weight.data = view
    assert m2.linear.weight.data_ptr() % 16 != 0
    out = m2(x2)
    assert torch.allclose(out.detach().float(), ref, rtol=2e-2, atol=1e-3)
    out.float().sum().backward()
