"""SURVEY 8(f) rows f1 and f2 on the device.

f1 - the tensor-network inner product of two EPS stacks (dctn/epses_composition.py:21-58), evaluated and
differentiated every training iteration: `dctn_fiber_gram` + `dctn_mode_product` (dctn_amd/csrc/tn_inner.hip) against
the reference's closed forms (restated from /root/reference/tests/test_epses_composition.py:7-41 and
tests/test_eps.py:64-73), the reference's own values and autograd gradients (tests/golden/inner_product.npz) and
the oracle on seeded stacks at the size of BASELINE config 3a.

f2 - the empirical-output-std initialisation (dctn/eps.py:163-181, dctn/epses_composition.py:91-105) through
`dctn_eps_fwd_stats` (sums as an epilogue of the forward, nothing materialised) against the cores the reference
itself returns for a fixed seed (tests/golden/empirical_std_init.npz) and against the oracle.
"""
import os

import numpy as np
import pytest
import torch

import dctn_amd
from dctn_amd import _lib as L
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


# ------------------------------------------------------------------------------------------------ f1
def test_inner_product_closed_forms_through_the_alias_package():
    # restated from tests/test_epses_composition.py:7-41 and tests/test_eps.py:64-73; CPU float32 tensors as there
    from dctn.eps import contract_on_input_dims
    from dctn.epses_composition import inner_product

    a = torch.einsum("oi,j->ijo", torch.eye(3), 2.0 * torch.ones(3))
    assert torch.allclose(contract_on_input_dims(a, a), 12.0 * torch.eye(3))
    assert dctn_amd.last_kernel() == "tn_fiber_gram"
    a4 = torch.einsum("oi,j->ijo", 2.0 * torch.eye(4), torch.tensor([1.0, 2.0, 3.0, 4.0]))
    b4 = torch.einsum("pj,i->ijp", 3.0 * torch.eye(4), torch.ones(4))
    assert torch.allclose(contract_on_input_dims(a4, b4),
                          torch.einsum("o,p->op", 2.0 * torch.ones(4), torch.tensor([3.0, 6.0, 9.0, 12.0])))
    a = torch.einsum("oi,j->ijo", torch.eye(3), torch.ones(3))
    assert torch.allclose(inner_product((a,), (a,)), torch.tensor(9.0))
    assert torch.allclose(inner_product((a, a), (a, a)), torch.tensor(3.0**4))
    assert torch.allclose(inner_product((a, a, a), (a, a, a)), torch.tensor(3.0**8))
    green = torch.einsum("oj,i->ijo", torch.eye(6)[:4], torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0]))
    black = torch.einsum("oi,j->ijo", torch.eye(4)[:3], torch.tensor([1.5, 0.0, 0.0, 0.0]))
    orange = torch.einsum("oi,j->ijo", torch.eye(6)[:4], torch.ones(6))
    red = torch.einsum("oi,j->ijo", torch.eye(4)[1:], torch.tensor([1.0, 0.0, 0.0, 1.0]))
    got = inner_product((green, black), (orange, red))
    assert got.device.type == "cpu" and torch.allclose(got, torch.tensor((2 + 3 + 4) * 5 * 1.5))


@pytest.mark.parametrize("tag", ["s1", "s2", "s3"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_inner_product_value_and_gradients_match_the_reference(tag, dtype):
    from dctn_amd.epses_composition import epswise_squared_fro_norm, inner_product

    g = load("inner_product")
    n = int(g[f"{tag}_n"])
    a = [torch.from_numpy(g[f"{tag}_a{i}"]).to(dtype).to(DEV).requires_grad_(True) for i in range(n)]
    b = [torch.from_numpy(g[f"{tag}_b{i}"]).to(dtype).to(DEV).requires_grad_(True) for i in range(n)]
    rtol = 1e-10 if dtype == torch.float64 else 2e-4

    def close(got, want):
        want = torch.from_numpy(np.asarray(want)).double()
        scale = float(want.abs().max()) or 1.0
        return float((got.detach().cpu().double() - want).abs().max()) <= rtol * scale

    val = inner_product(a, b)
    assert val.shape == () and val.dtype == dtype and close(val, g[f"{tag}_value"])
    val.backward()
    assert dctn_amd.last_kernel() in ("tn_mode_product", "tn_fiber_gram")
    for i in range(n):
        assert close(a[i].grad, g[f"{tag}_da{i}"]) and close(b[i].grad, g[f"{tag}_db{i}"])
    # the regulariser's own use: a stack with itself (both uses of every core collect gradient)
    if dtype == torch.float32 and abs(float(g[f"{tag}_self_value"])) > 1e30:
        return   # the three-layer fixture's self inner product (1.6e48) does not fit float32
    for t in a:
        t.grad = None
    self_val = inner_product(a, a)
    assert close(self_val, g[f"{tag}_self_value"])
    self_val.backward()
    for i in range(n):
        assert close(a[i].grad, g[f"{tag}_self_da{i}"])
    assert close(epswise_squared_fro_norm(a), g[f"{tag}_sqfro"])


def test_inner_product_cfg3a_size_against_the_oracle_and_in_the_model():
    """BASELINE config 3a's cores (65 536 x 4 and 262 144 x 6): the 6 MiB core goes through nine mode products."""
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd

    torch.manual_seed(7)
    m = EPSesPlusLinear(((4, 4), (3, 6)), UnitTheoreticalOutputStd(), 1.0, DEV, torch.float32)
    # (with the theoretical-std cores the Gram matrix of layer 1 is close to the identity and the value is O(1))
    reg = m.epses_composition_l2_regularizer()
    cores64 = [c.detach().cpu().double().requires_grad_(True) for c in m.epses]
    w64 = m.linear.weight.detach().cpu().double()
    want = (w64 * w64).sum() + R.epses_inner_product(cores64, cores64)
    assert abs(float(reg) - float(want)) <= 2e-4 * abs(float(want))
    reg.backward()
    want.backward()
    for got, ref in zip(m.epses, cores64):
        assert float((got.grad.cpu().double() - ref.grad).abs().max()) <= 3e-4 * float(ref.grad.abs().max())
    # bf16 storage, f32 accumulation
    ab = [c.detach().bfloat16().requires_grad_(True) for c in m.epses]
    from dctn_amd.epses_composition import inner_product

    val = inner_product(ab, ab)
    ref = R.epses_inner_product([c.detach().cpu().double() for c in ab], [c.detach().cpu().double() for c in ab])
    assert abs(float(val) - float(ref)) <= 3e-2 * abs(float(ref))


def test_mode_product_and_fiber_gram_shapes_fuzz():
    """The two primitives against einsum for leg sizes 1..32, pre / post of 1 and of non-multiples of the tile."""
    from dctn_amd import tn_inner

    gen = torch.Generator().manual_seed(3)
    for pre, q, q2, post in [(1, 3, 5, 1), (7, 2, 2, 9), (300, 4, 4, 1), (5, 32, 17, 3), (1, 6, 1, 700), (513, 1, 8, 2),
                             (2, 24, 24, 130)]:
        x = torch.randn(pre, q, post, dtype=torch.float64, generator=gen)
        M = torch.randn(q, q2, dtype=torch.float64, generator=gen)
        y = torch.randn(pre, q2, post, dtype=torch.float64, generator=gen)
        got = tn_inner._mode_product(x.to(DEV).reshape(-1), M.to(DEV), pre, q, q2, post).cpu().reshape(pre, q2, post)
        assert torch.allclose(got, torch.einsum("aib,ij->ajb", x, M), rtol=1e-11, atol=1e-12)
        gram = tn_inner._fiber_gram(x.to(DEV).reshape(-1), y.to(DEV).reshape(-1), pre, q, q2, post).cpu()
        assert torch.allclose(gram, torch.einsum("aib,ajb->ij", x, y), rtol=1e-11, atol=1e-11)
    lib = L.lib()
    assert lib.dctn_mode_product(8, 8, 8, 4, 33, 2, 1, 0, None) == L.ERR_UNSUPPORTED      # leg size beyond 32
    assert lib.dctn_mode_product(8, 8, 8, 0, 3, 2, 1, 0, None) == L.ERR_BAD_SHAPE
    assert lib.dctn_fiber_gram(8, 8, 8, None, 0, 4, 2, 2, 1, 0, None) == L.ERR_WORKSPACE


# ------------------------------------------------------------------------------------------------ f2
def test_empirical_std_init_reproduces_the_reference_cores():
    from dctn_amd import eps as E
    from dctn_amd import epses_composition as EC

    g = load("empirical_std_init")
    x, batch, seed = torch.from_numpy(g["x"]), int(g["batch_size"]), int(g["seed"])
    for device in (DEV, torch.device("cpu")):     # a CPU-side caller is staged
        torch.manual_seed(seed)
        core = E.make_eps_unit_empirical_output_std(3, 4, x, device, torch.float64, batch)
        assert core.device.type == device.type
        assert torch.allclose(core.cpu(), torch.from_numpy(g["one_core"]), rtol=1e-11, atol=0)
    assert dctn_amd.last_kernel().startswith("eps_fwd")
    torch.manual_seed(seed)
    cores = EC.make_epses_composition_unit_empirical_output_std(((3, 3), (2, 4)), x, DEV, torch.float64, batch)
    assert torch.allclose(cores[0].cpu(), torch.from_numpy(g["stack_core0"]), rtol=1e-11, atol=0)
    assert torch.allclose(cores[1].cpu(), torch.from_numpy(g["stack_core1"]), rtol=1e-10, atol=0)
    # float32: same draw (randn in float32 differs from the float64 draw), so compare with the oracle's scale
    torch.manual_seed(seed)
    raw = torch.randn(*(2,) * 9, 4, dtype=torch.float32)
    torch.manual_seed(seed)
    core32 = E.make_eps_unit_empirical_output_std(3, 4, x, DEV, torch.float32, batch)
    scale = R.unit_empirical_output_std_scale(raw.double(), x, batch)
    assert torch.allclose(core32.cpu().double(), raw.double() * scale, rtol=1e-5, atol=0)


@pytest.mark.parametrize("dtype,K,Q,O,family", [(torch.bfloat16, 3, 2, 4, "q2reg"), (torch.float32, 4, 2, 4, "bigcore"),
                                                (torch.float64, 4, 2, 2, "halves"), (torch.float32, 2, 3, 5, "generic")])
def test_forward_statistics_epilogue_equals_the_materialised_output(dtype, K, Q, O, family):
    """`dctn_eps_fwd_stats` on every kernel family: (count, sum, sum of squares) equal those of the stored output of
    the same slices; the register-resident family does it inside the forward kernel and asks for no scratch."""
    from dctn_amd.eps import eps, output_sums_in_slices

    torch.manual_seed(11)
    n = 37
    x = torch.rand(1, n, 12, 11, Q).to(dtype).to(DEV)
    core = (torch.randn(*(Q,) * (K * K), O) * Q ** (-K * K / 2.5)).to(dtype).to(DEV)
    count, sums = output_sums_in_slices(core, x, 16)          # slices of 16, 16, 5
    y = torch.cat([eps(core, part) for part in x.split(16, dim=1)]).double()
    assert family in dctn_amd.last_kernel()
    assert count == y.numel()
    tol = {torch.float64: 1e-12, torch.float32: 1e-6, torch.bfloat16: 1e-6}[dtype]
    assert abs(float(sums[0]) - float(y.sum())) <= tol * float(y.abs().sum())
    assert abs(float(sums[1]) - float((y * y).sum())) <= tol * float((y * y).sum())
    code = L.dtype_code(x)
    ws = L.lib().dctn_eps_fwd_stats_workspace_bytes(1, 16, 12, 11, Q, K, O, code, 0)
    assert (ws <= 512) == (family == "q2reg")


def test_composition_regulariser_inside_the_graphed_training_iteration():
    """f1 wired into the iteration: `GraphedTrainStep` with reg_fn = epses_composition_l2_regularizer (Gram, the mode
    products, the dot and all their backward launches captured in the one HIP graph) follows the eager loop."""
    import copy

    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd
    from dctn_amd.training import GraphedTrainStep, train_step

    torch.manual_seed(9)
    a = EPSesPlusLinear(((2, 3), (2, 4)), UnitTheoreticalOutputStd(), 1.0, DEV, torch.float32, image_size=8)
    b = copy.deepcopy(a)
    xs = [torch.rand(1, 16, 8, 8, 2, device=DEV) for _ in range(4)]
    ys = [torch.randint(0, 10, (16,), device=DEV) for _ in range(4)]
    reg = lambda m: m.epses_composition_l2_regularizer()
    ce = torch.nn.functional.cross_entropy
    oa = torch.optim.SGD(a.parameters(), lr=0.05, momentum=0.9)
    ob = torch.optim.SGD(b.parameters(), lr=0.05, momentum=0.9)
    graphed = GraphedTrainStep(b, xs[0], ys[0], ce, ob, reg_fn=reg, reg_coeff=1e-2, warmup=2)
    for _ in range(2):
        train_step(a, xs[0], ys[0], ce, oa, reg_fn=reg, reg_coeff=1e-2)
    for x, y in zip(xs, ys):
        ra = train_step(a, x, y, ce, oa, reg_fn=reg, reg_coeff=1e-2)
        rb = graphed(x, y)
        assert torch.allclose(ra["loss"], rb["loss"], rtol=1e-5, atol=1e-6)
        assert torch.allclose(ra["reg_term"], rb["reg_term"].float(), rtol=1e-5, atol=1e-6)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-4, atol=1e-6)
