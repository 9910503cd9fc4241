"""GPU parity at BASELINE.json's FULL sizes (run with `-m gpu` on an MI355X).

The CPU oracle cannot evaluate these sizes in seconds, so every configuration is checked through
size-independent properties of the domain plus an oracle spot check on a few samples of the same
tensors:

* batch independence: windows (hence samples) do not interact, so the result of a slice of the
  batch equals the slice of the result — bit for bit, whatever wave / tile a window lands in;
* homogeneity in the core: scaling a core by a power of two scales the output exactly;
* additivity of the parameter gradients over batch slices (up to the summation order);
* fused vs unfused backward of the classifier tail (two different kernel sets, same numbers);
* logmatmulexp: associativity of the fold and exact shift equivariance.

Nothing here reads /root/reference.
"""
import os

import pytest
import torch

import dctn_amd
from dctn_amd.conv_sbs import DumbNormalInitialization, ManyConvSBS
from dctn_amd.conv_sbs_spec import SBSSpecCore
from dctn_amd.eps import eps
from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd
from dctn_amd.logmatmulexp import logmatmulexp_batched, logmatmulexp_fold
from dctn_amd.pos2d import Pos2D
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def mnist_like(batch, size, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(1, batch, size, size, generator=g)
    return torch.stack((torch.sin(u * torch.pi / 2) ** 2, torch.cos(u * torch.pi / 2) ** 2), dim=-1).to(dtype).to(DEV)


def rel_err(got, want):
    want = want.double()
    got = got.detach().cpu().double()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))


def test_cfg1_full_batch_64_f64():
    """BASELINE configs[0]: the reference's own EPS micro-benchmark shape (small_experiments/eps2d_benchmark/
    benchmark.py:45-64): B = 64, 28x28, K = 4, Q = 2, O = 2, float64, core and input both with gradients."""
    torch.manual_seed(1)
    x = mnist_like(64, 28, torch.float64, 11)
    core = (torch.randn(*(2,) * 16, 2, dtype=torch.float64) * 2.0 ** -4).to(DEV)
    y = eps(core, x)
    assert y.shape == (64, 25, 25, 2) and dctn_amd.last_kernel() == "eps_fwd_mfma_f64_halves"
    # batch independence (the windows of a sample never meet those of another one; GEMM tiles do change)
    for lo, hi in ((0, 1), (5, 40), (63, 64)):
        assert rel_err(eps(core, x[:, lo:hi]), y[lo:hi].cpu()) < 1e-13
    assert torch.equal(eps(core * 4, x), y * 4)                      # exact homogeneity (powers of two)
    idx = [0, 31, 63]
    want = R.eps_4step(core.cpu(), x[:, idx].cpu())
    assert rel_err(y[idx], want) < 1e-12
    # gradients: oracle on a slice of the batch, additivity of dCore over batch slices, dX per sample
    dy = torch.randn(64, 25, 25, 2, dtype=torch.float64, generator=torch.Generator().manual_seed(12)).to(DEV)

    def grads(lo, hi):
        c, xs = core.clone().requires_grad_(True), x[:, lo:hi].clone().requires_grad_(True)
        eps(c, xs).backward(dy[lo:hi])
        assert dctn_amd.last_kernel() == "eps_bwd_mfma_f64_halves_savedz"   # the forward kept P0, P1 and Z
        return c.grad.cpu(), xs.grad.cpu()

    dc_whole, dx_whole = grads(0, 64)
    parts = [grads(0, 3), grads(3, 40), grads(40, 64)]
    assert rel_err(sum(p[0] for p in parts), dc_whole) < 1e-12
    assert rel_err(torch.cat([p[1] for p in parts], dim=1), dx_whole) < 1e-12
    gc, gx = R.grads(R.eps_4step, [core.cpu(), x[:, 0:3].cpu()], dy[0:3].cpu())
    assert rel_err(parts[0][0], gc) < 1e-11 and rel_err(parts[0][1], gx) < 1e-11


def test_cfg2_full_batch_1024_bf16():
    """BASELINE configs[1]: EPSesPlusLinear(((3,4),)), x (1,1024,28,28,2) bf16 — the bench workload."""
    torch.manual_seed(2)
    model = EPSesPlusLinear(((3, 4),), UnitTheoreticalOutputStd(), 1.0, DEV, torch.bfloat16)
    with torch.no_grad():
        model.linear.weight.mul_(8.0)
    x = mnist_like(1024, 28, torch.bfloat16, 1)
    core = model.epses[0].detach()
    feat = eps(core, x)
    assert feat.shape == (1024, 26, 26, 4) and dctn_amd.last_kernel() == "eps_fwd_mfma_q2reg"
    # batch independence, bit for bit (slices start at odd offsets: other waves, other lanes)
    for lo, hi in ((0, 1), (3, 260), (517, 1024)):
        assert torch.equal(eps(core, x[:, lo:hi]), feat[lo:hi])
    # exact homogeneity
    assert torch.equal(eps(core * 4, x), feat * 4)
    # oracle spot check on 6 samples of the same tensors
    idx = [0, 1, 511, 512, 777, 1023]
    want = R.eps_4step(core.cpu().double(), x[:, idx].cpu().double())
    assert rel_err(feat[idx], want) < 1e-2
    # whole model: fused backward == unfused backward == sum over batch slices; oracle on the logits
    g = (torch.randn(1024, 10, generator=torch.Generator().manual_seed(3)) * 0.1).to(torch.bfloat16).to(DEV)

    def grads(xs, gs, fused):
        import dctn_amd.eps_plus_linear as EPL
        EPL.FUSED_HEAD = fused
        try:
            for prm in model.parameters():
                prm.grad = None
            out = model(xs)
            out.backward(gs)
            return out.detach(), [prm.grad.detach().float().cpu() for prm in model.parameters()], dctn_amd.last_kernel()
        finally:
            EPL.FUSED_HEAD = True

    out_f, gf, kf = grads(x, g, True)
    out_u, gu, ku = grads(x, g, False)
    assert kf == "eps_head_bwd_mfma_q2reg" and ku != kf
    # same features; the fused forward sums the head's products in another order than the stand-alone head kernel
    assert float((out_f.float() - out_u.float()).abs().max()) <= 2 ** -7 * float(out_u.float().abs().max())
    w64, b64 = model.linear.weight.detach().cpu().double(), model.linear.bias.detach().cpu().double()
    want_logits = R.eps_plus_linear_forward([core.cpu().double()], w64, b64, x[:, idx].cpu().double())
    assert rel_err(out_f[idx], want_logits) < 2e-2
    parts = [grads(x[:, lo:hi], g[lo:hi], True)[1] for lo, hi in ((0, 300), (300, 301), (301, 1024))]
    for name, a, b, c in zip(("dCore", "dWeight", "dBias"), gf, gu, [sum(p[i] for p in parts) for i in range(3)]):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) < 2e-2 * scale, name + ": fused vs unfused"
        assert float((a - c).abs().max()) < 2e-2 * scale, name + ": whole batch vs sum of slices"


def test_cfg2_full_batch_1024_f32():
    """BASELINE configs[1] in the reference's own arithmetic (float32, new_runner.py:417): EPSesPlusLinear(((3,4),)),
    x (1,1024,28,28,2) float32 on the register-resident exact-f32 family (eps_q2f32.hip) - bench config cfg2_f32."""
    torch.manual_seed(2)
    model = EPSesPlusLinear(((3, 4),), UnitTheoreticalOutputStd(), 1.0, DEV, torch.float32)
    with torch.no_grad():
        model.linear.weight.mul_(8.0)
    x = mnist_like(1024, 28, torch.float32, 1)
    core = model.epses[0].detach()
    feat = eps(core, x)
    assert feat.shape == (1024, 26, 26, 4) and dctn_amd.last_kernel() == "eps_fwd_q2f32"
    # batch independence, bit for bit (slices start at odd offsets: other workgroups, other waves)
    for lo, hi in ((0, 1), (3, 260), (517, 1024)):
        assert torch.equal(eps(core, x[:, lo:hi]), feat[lo:hi])
    assert torch.equal(eps(core * 4, x), feat * 4)   # exact homogeneity
    idx = [0, 1, 511, 512, 777, 1023]
    want = R.eps_4step(core.cpu().double(), x[:, idx].cpu().double())
    assert rel_err(feat[idx], want) < 2e-5
    g = (torch.randn(1024, 10, generator=torch.Generator().manual_seed(3)) * 0.1).to(DEV)

    def grads(xs, gs, fused):
        import dctn_amd.eps_plus_linear as EPL
        EPL.FUSED_HEAD = fused
        try:
            for prm in model.parameters():
                prm.grad = None
            out = model(xs)
            kfw = dctn_amd.last_kernel()
            out.backward(gs)
            return out.detach(), [prm.grad.detach().cpu() for prm in model.parameters()], (kfw, dctn_amd.last_kernel())
        finally:
            EPL.FUSED_HEAD = True

    out_f, gf, kf = grads(x, g, True)
    out_u, gu, ku = grads(x, g, False)
    assert kf == ("eps_head_fwd_q2f32", "eps_head_bwd_q2f32") and ku[1] != kf[1]
    # the one-kernel forward stores the same features and sums the head's products in another order
    assert torch.equal(model(x), out_f)
    assert float((out_f - out_u).abs().max()) <= 2e-5 * float(out_u.abs().max())
    w64, b64 = model.linear.weight.detach().cpu().double(), model.linear.bias.detach().cpu().double()
    want_logits = R.eps_plus_linear_forward([core.cpu().double()], w64, b64, x[:, idx].cpu().double())
    assert rel_err(out_f[idx], want_logits) < 2e-5
    parts = [grads(x[:, lo:hi], g[lo:hi], True)[1] for lo, hi in ((0, 300), (300, 301), (301, 1024))]
    for name, a, b, c in zip(("dCore", "dWeight", "dBias"), gf, gu, [sum(p[i] for p in parts) for i in range(3)]):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) < 2e-4 * scale, name + ": fused vs unfused"      # the float32 tolerance (rtol 2e-4)
        assert float((a - c).abs().max()) < 2e-4 * scale, name + ": whole batch vs sum of slices"
    # oracle on the gradients of a slice of the batch (the same tensors)
    lo, hi = 300, 316
    c64 = core.cpu().double().requires_grad_(True)
    w64g, b64g = w64.clone().requires_grad_(True), b64.clone().requires_grad_(True)
    R.eps_plus_linear_forward([c64], w64g, b64g, x[:, lo:hi].cpu().double()).backward(g[lo:hi].cpu().double())
    got = grads(x[:, lo:hi], g[lo:hi], True)[1]
    for name, a, ref in zip(("dCore", "dWeight", "dBias"), got, (c64.grad, w64g.grad, b64g.grad)):
        assert rel_err(a, ref) < 2e-5, name


def test_cfg3a_full_batch_128_f32():
    """BASELINE cfg3 (reference-canonical spec): EPSesPlusLinear(((4,4),(3,6))), f32, B = 128."""
    torch.manual_seed(3)
    model = EPSesPlusLinear(((4, 4), (3, 6)), UnitTheoreticalOutputStd(), 1.0, DEV, torch.float32)
    x = mnist_like(128, 28, torch.float32, 4)
    e1, e2 = model.epses[0].detach(), model.epses[1].detach()
    y1 = eps(e1, x)
    assert y1.shape == (128, 25, 25, 4) and dctn_amd.last_kernel() == "eps_fwd_mfma_bigcore_f32"
    y2 = eps(e2, y1.unsqueeze(0))
    assert y2.shape == (128, 23, 23, 6)
    # batch independence; not bit for bit here: the number of row-group slices the k-sum is split
    # into depends on the number of windows, i.e. the f32 summation order changes with the batch
    for lo, hi in ((0, 1), (5, 70), (127, 128)):
        assert rel_err(eps(e1, x[:, lo:hi]), y1[lo:hi].cpu()) < 2e-6
        assert rel_err(eps(e2, y1[lo:hi].unsqueeze(0)), y2[lo:hi].cpu()) < 2e-6
    # homogeneity (not bit for bit at these magnitudes: outputs are ~1e-32, single products fall into
    # the denormal range and are flushed or not depending on the scale)
    assert rel_err(eps(e2 * 2, y1.unsqueeze(0)), (y2 * 2).cpu()) < 1e-5
    idx = [0, 127]
    w1 = R.eps_4step(e1.cpu().double(), x[:, idx].cpu().double())
    assert rel_err(y1[idx], w1) < 1e-5
    w2 = R.eps_4step(e2.cpu().double(), w1.unsqueeze(0))
    assert rel_err(y2[idx], w2) < 1e-5
    # parameter gradients: whole batch == sum over slices (all four GEMM kernels + the dCore kernel)
    g = torch.randn(128, 10, generator=torch.Generator().manual_seed(5)).to(DEV)

    def grads(lo, hi):
        for prm in model.parameters():
            prm.grad = None
        model(x[:, lo:hi]).backward(g[lo:hi])
        return [prm.grad.detach().double().cpu() for prm in model.parameters()]

    whole = grads(0, 128)
    parts = [grads(0, 50), grads(50, 51), grads(51, 128)]
    for i, a in enumerate(whole):
        c = sum(p[i] for p in parts)
        assert float((a - c).abs().max()) < 2e-4 * float(a.abs().max()), i


@pytest.mark.parametrize("r", [16, 8, 4])
def test_cfg4_convsbs_full_batch_128(r):
    """BASELINE cfg4: the mnist.py snake string on the CIFAR colour layout, B = 128: r = 16 and r = 8 (band family: two bands
    of window rows per image, 256 workgroups, the pixel rows they share summed by the tail kernel; four / two state values per
    lane) and r = 4 (register-resident sweep: two bands of pixel rows per image, halo rows computed twice)."""
    snake = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
    spec = (tuple(SBSSpecCore(Pos2D(*p), 2 if i == 4 else 1) for i, p in enumerate(snake)),)
    torch.manual_seed(4)
    many = ManyConvSBS(1, 3, r, False, spec, (DumbNormalInitialization((3 * r) ** -0.5),)).to(DEV)
    x = torch.randn(1, 128, 32, 32, 3, generator=torch.Generator().manual_seed(6)).to(DEV).requires_grad_(True)
    (y,) = many(x)
    assert y.shape == (128, 30, 30, 2) and ("band" if r >= 8 else "reg") in dctn_amd.last_kernel()
    for lo, hi in ((0, 1), (7, 100), (127, 128)):
        assert rel_err(many(x[:, lo:hi].detach())[0], y[lo:hi].detach().cpu()) < 2e-6
    cores = [c.detach().cpu().double() for c in many.strings[0].cores]
    idx = [0, 64]
    want = R.convsbs_forward(cores, snake, x[:, idx].detach().cpu().double())
    assert rel_err(y[idx], want) < 1e-4
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7)).to(DEV)
    y.backward(dy)
    whole = [c.grad.detach().double().cpu() for c in many.strings[0].cores]
    dx_whole = x.grad.detach().clone()
    # oracle gradients on two samples (dX is per sample; dCore of the pair by additivity below)
    xo = x[:, idx].detach().cpu().double().requires_grad_(True)
    R.convsbs_forward(cores, snake, xo).backward(dy[idx].cpu().double())
    assert rel_err(dx_whole[:, idx], xo.grad) < 1e-4
    acc = [torch.zeros_like(w) for w in whole]
    for lo, hi in ((0, 40), (40, 41), (41, 128)):
        for c in many.strings[0].cores:
            c.grad = None
        xs = x[:, lo:hi].detach().requires_grad_(True)
        many(xs)[0].backward(dy[lo:hi])
        for a, c in zip(acc, many.strings[0].cores):
            a += c.grad.detach().double().cpu()
        assert rel_err(xs.grad, dx_whole[:, lo:hi].cpu()) < 1e-5     # dX is per sample
    for a, w in zip(acc, whole):
        assert float((a - w).abs().max()) < 2e-4 * float(w.abs().max())


def test_cfg5_logmatmulexp_fold_full_692224_windows():
    """BASELINE cfg5: left fold of 9 log-matrices (16x16) for every window of a 1024-sample batch."""
    Wn = 692224
    m = torch.randn(Wn, 9, 16, 16, device=DEV, generator=torch.Generator(device=DEV).manual_seed(8))
    md = m.requires_grad_(True)
    m = m.detach()
    y = logmatmulexp_fold(md)
    assert y.shape == (Wn, 16, 16) and dctn_amd.last_kernel() == "logmatmulexp_fold_fwd_mfma16"
    idx = torch.tensor([0, 1, 63, 64, 345678, Wn - 1], device=DEV)
    m_sub = m[idx].cpu().double()
    want = R.logmatmulexp_fold_batched(m_sub)
    assert float((y[idx].detach().cpu().double() - want).abs().max()) < 5e-5
    # associativity: fold(9) == lme(fold(first 5), fold(last 4))
    left, right = logmatmulexp_fold(md[:, :5].detach().contiguous()), logmatmulexp_fold(md[:, 5:].detach().contiguous())
    both = logmatmulexp_batched(left, right)
    assert float((both - y.detach()).abs().max()) < 1e-4
    # shift equivariance: adding c to one factor adds c to the result
    shifted = md.detach().clone()
    shifted[:, 3] += 2.5
    assert float((logmatmulexp_fold(shifted) - (y.detach() + 2.5)).abs().max()) < 2e-5
    # backward at full size: oracle on the same subset of windows
    dy = torch.randn(Wn, 16, 16, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9))
    y.backward(dy)
    assert dctn_amd.last_kernel() == "logmatmulexp_fold_bwd_mfma16"
    (gm,) = R.grads(R.logmatmulexp_fold_batched, [m_sub], dy[idx].cpu().double())
    got = md.grad[idx].cpu().double()
    assert float((got - gm).abs().max()) < 2e-4 * float(gm.abs().max().clamp_min(1.0))
    # gradient mass conservation: for every window the gradient wrt the last factor sums to sum(dy)
    # (the softmax weights of each output entry sum to one)
    assert float((md.grad[:, 8].sum((1, 2)) - dy.sum((1, 2))).abs().max()) < 2e-3
