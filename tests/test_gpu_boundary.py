"""The drop-in boundary with CPU tensors (SURVEY 8b "dtype/device": f32 and f64 on CPU must be accepted).

The reference's own tests build CPU float64 tensors and call ``dctn.eps`` / ``dctn.conv_sbs`` directly
(/root/reference/tests/test_eps.py:9-61, tests/test_conversion_of_convsbs_to_eps.py:13-56,
tests/test_epses_composition.py:7-41).  Here their assertions are RESTATED (not copied) and run through the alias
package ``dctn`` exactly as a user of the reference would: CPU tensors in, CPU tensors out, gradients on the CPU
leaves — with the arithmetic on the MI355X (CPU tensors are staged to the device, `dctn_amd._lib.placement`; the
last-kernel name proves the HIP path ran).  Nothing here touches the oracle: the expected values are the
independent einsum definitions the reference's tests use.
"""
import itertools
import string

import pytest
import torch

pytestmark = pytest.mark.gpu


def _definition(core, factors):
    """sum over i_0..i_{N-1} of core[i_0, .., i_{N-1}, o] * prod_n factors[n][..., i_n]: one einsum, the formulation
    the reference's tests hand to opt_einsum ("01234567θ,b0,...,b7->bθ")."""
    n = len(factors)
    letters = string.ascii_lowercase[:n]
    batched = factors[0].ndim == 2
    lhs = ",".join(("z" + l) if batched else l for l in letters)
    return torch.einsum(f"{letters}y,{lhs}->{'zy' if batched else 'y'}", core, *factors)


def test_eps_single_pixel_output_cpu_f64():
    # restated from tests/test_eps.py:9-26: C=2, K=2 on a 2x2 image, factor index = position * C + channel
    import dctn.eps
    import dctn_amd

    x = torch.randn((2, 3, 2, 2, 2), dtype=torch.float64)
    core = torch.rand((2,) * 8 + (4,), dtype=torch.float64)
    got = dctn.eps.eps_one_by_one(core, x)
    assert got.device.type == "cpu" and got.dtype == torch.float64 and got.shape == (3, 1, 1, 4)
    assert dctn_amd.last_kernel().startswith("eps_fwd")   # the HIP library computed it
    factors = [x[ch, :, h, w] for h in range(2) for w in range(2) for ch in range(2)]
    assert torch.allclose(got.reshape(3, 4), _definition(core, factors))
    assert torch.allclose(dctn.eps.eps(core, x), got)


def test_eps_two_pixels_output_cpu_f64():
    # restated from tests/test_eps.py:29-61: K=3 on a 4x3 image -> two windows, one below the other
    import dctn.eps

    x = torch.randn((1, 1, 4, 3, 2), dtype=torch.float64)
    core = torch.rand((2,) * 9 + (4,), dtype=torch.float64)
    got = dctn.eps.eps_one_by_one(core, x)
    assert got.shape == (1, 2, 1, 4) and got.device.type == "cpu"
    for top in (0, 1):
        factors = [x[0, 0, top + dh, dw] for dh in range(3) for dw in range(3)]
        assert torch.allclose(got[0, top, 0], _definition(core, factors))


def test_cpu_gradients_arrive_on_the_cpu_leaves():
    import dctn.eps

    x = torch.randn((1, 2, 5, 5, 3), dtype=torch.float32, requires_grad=True)
    core = torch.randn((3,) * 4 + (5,), dtype=torch.float32, requires_grad=True)
    out = dctn.eps.eps(core, x)
    out.square().sum().backward()
    assert x.grad is not None and x.grad.device.type == "cpu" and core.grad.device.type == "cpu"
    factors = [x.detach()[0, :, h, w] for h in range(2) for w in range(2)]   # the top-left window of each sample
    assert torch.allclose(out.detach()[:, 0, 0], _definition(core.detach(), factors), rtol=1e-4, atol=1e-5)


def test_convsbs_equals_eps_for_all_orders_cpu_f64():
    # restated from tests/test_conversion_of_convsbs_to_eps.py:13-56: ring bonds (3,4,5,6), outs (1,3,2,4), every
    # order of the four cores of a 2x2 window; CPU float64 module and input, as the reference builds them
    from dctn.conv_sbs import ConvSBS
    from dctn.conv_sbs_spec import SBSSpecCore, SBSSpecString
    from dctn.eps import eps
    from dctn.pos2d import Pos2D
    import dctn_amd

    cores = (SBSSpecCore(Pos2D(0, 0), 1), SBSSpecCore(Pos2D(0, 1), 3), SBSSpecCore(Pos2D(1, 0), 2),
             SBSSpecCore(Pos2D(1, 1), 4))
    for order in itertools.permutations(cores):
        sbs = ConvSBS(SBSSpecString(order, (3, 4, 5, 6), 2, 2)).double()
        with torch.no_grad():
            dense = sbs.as_eps()
        assert dense.shape == (2,) * 8 + (24,)
        assert torch.all(dense == sbs.as_eps())
        x = torch.randn(2, 3, 4, 5, 2, dtype=torch.float64, requires_grad=True)
        y_sbs = sbs(x)
        assert y_sbs.device.type == "cpu" and dctn_amd.last_kernel().startswith("convsbs_fwd")
        seed = torch.randn_like(y_sbs)
        y_sbs.backward(seed)
        g_sbs = x.grad.clone()
        x.grad.zero_()
        y_eps = eps(dense, x)
        assert torch.allclose(y_eps, y_sbs)
        y_eps.backward(seed)
        assert torch.allclose(x.grad, g_sbs)


def test_model_on_cpu_runs_on_the_device():
    """EPSesPlusLinear built with device=cpu (what `EPS.__init__` / a CPU-side user gets): forward and backward
    are staged once per call; parameters' gradients land on the CPU parameters."""
    from dctn.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd

    torch.manual_seed(3)
    cpu_model = EPSesPlusLinear(((2, 3), (2, 4)), UnitTheoreticalOutputStd(), 1.0, torch.device("cpu"), torch.float64,
                                image_size=6)
    gpu_model = EPSesPlusLinear(((2, 3), (2, 4)), UnitTheoreticalOutputStd(), 1.0, torch.device("cuda"), torch.float64,
                                image_size=6)
    gpu_model.load_state_dict(cpu_model.state_dict())
    x = torch.rand(1, 4, 6, 6, 2, dtype=torch.float64)
    out = cpu_model(x)
    assert out.device.type == "cpu" and out.shape == (4, 10)
    out.logsumexp(1).sum().backward()
    ref = gpu_model(x.cuda())
    ref.logsumexp(1).sum().backward()
    assert torch.allclose(out, ref.cpu(), rtol=1e-12, atol=1e-14)
    for a, b in zip(cpu_model.parameters(), gpu_model.parameters()):
        assert a.grad.device.type == "cpu" and torch.allclose(a.grad, b.grad.cpu(), rtol=1e-10, atol=1e-13)


def test_logmatmulexp_cpu_inputs():
    from dctn.logmatmulexp import logmatmulexp, logmatmulexp_lowmem

    a = torch.randn(5, 7, dtype=torch.float64, requires_grad=True)
    b = torch.randn(7, 3, dtype=torch.float64, requires_grad=True)
    got = logmatmulexp(a, b)
    want = (a.detach().exp() @ b.detach().exp()).log()   # the reference's docstring definition (logmatmulexp.py:6-7)
    assert got.device.type == "cpu" and torch.allclose(got, want)
    assert torch.allclose(logmatmulexp_lowmem(a, b), want)
    got.sum().backward()
    assert a.grad.device.type == "cpu" and b.grad.device.type == "cpu"


def test_mixed_placement_raises():
    import dctn.eps

    with pytest.raises(RuntimeError, match="one device"):
        dctn.eps.eps(torch.randn(2, 2, 2, 2, 3), torch.randn(1, 2, 4, 4, 2, device="cuda"))
