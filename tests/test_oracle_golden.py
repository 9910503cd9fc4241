"""The CPU oracle (oracle/ref_cpu.py) against the golden vectors produced by the reference
itself (tests/golden/make_golden.py).  Runs without a GPU."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = dict(rtol=1e-10, atol=1e-12)  # float64 vs float64, different pairwise order only


def load(name):
    return {k: v for k, v in np.load(os.path.join(GOLDEN, name + ".npz")).items()}


def t(a):
    return torch.from_numpy(np.asarray(a))


def test_pos2d_table_bit_exact():
    tab = load("pos2d")["table"]
    for max_w, idx, h, w, back in tab:
        assert R.index_to_pos(int(max_w), int(idx)) == (int(h), int(w))
        assert R.pos_to_index(int(max_w), (int(h), int(w))) == int(back) == int(idx)


@pytest.mark.parametrize("K", [2, 3, 4])
def test_align_index_map_bit_exact(K):
    g = load(f"align_k{K}")
    src = R.window_source_indices(int(g["H"]), int(g["W"]), R.standard_positions(K))
    assert src.dtype == np.int64 and np.array_equal(src, g["src"])


def test_align_with_positions_index_map_bit_exact():
    g = load("align_snake")
    pos = [tuple(int(v) for v in p) for p in g["positions"]]
    assert np.array_equal(R.window_source_indices(int(g["H"]), int(g["W"]), pos), g["src"])
    ramp = torch.arange(int(g["H"]) * int(g["W"]), dtype=torch.float64).reshape(1, 1, int(g["H"]), int(g["W"]), 1)
    views = torch.stack(R.align_with_positions(ramp, pos))[:, 0, :, :, 0].to(torch.int64)
    assert np.array_equal(views.numpy(), g["src"])


EPS_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "eps_c*.npz")))


@pytest.mark.parametrize("name", EPS_CASES)
def test_eps_forward_backward(name):
    g = load(name)
    x, core = t(g["x"]), t(g["core"])
    y = R.eps_4step(core, x)
    assert torch.allclose(y, t(g["y"]), **TOL)
    assert torch.allclose(R.eps_one_by_one(core, x), t(g["y_one_by_one"]), **TOL)
    dx, dcore = R.grads(R.eps_4step, [core, x], t(g["dy"]))[::-1]
    assert torch.allclose(dx, t(g["dx"]), **TOL)
    assert torch.allclose(dcore, t(g["dcore"]), **TOL)


@pytest.mark.parametrize("name", ["eps_c1_k3_q2_o4", "eps_c2_k2_q2_o4", "eps_c1_k2_q3_o5", "eps_c2_k1_q3_o2"])
def test_eps_definition_loops(name):
    g = load(name)
    y = R.eps_definition_numpy(g["core"], g["x"])
    assert np.allclose(y, g["y"], rtol=1e-10, atol=1e-12)


def test_epses_composition():
    g = load("epses_composition_33_25")
    x, e1, e2 = t(g["x"]), t(g["e1"]), t(g["e2"])
    assert torch.allclose(R.contract_with_input((e1, e2), x), t(g["y"]), **TOL)
    dx, de1, de2 = R.grads(lambda a, b, c: R.contract_with_input((b, c), a), [x, e1, e2], t(g["dy"]))
    assert torch.allclose(dx, t(g["dx"]), **TOL)
    assert torch.allclose(de1, t(g["de1"]), **TOL)
    assert torch.allclose(de2, t(g["de2"]), **TOL)


SBS_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "sbs_*.npz")))


def sbs_inputs(g):
    n = len(g["out_sizes"])
    cores = [t(g[f"core{i}"]) for i in range(n)]
    pos = [tuple(int(v) for v in p) for p in g["positions"]]
    return cores, pos, t(g["x"])


@pytest.mark.parametrize("name", SBS_CASES)
def test_convsbs_forward_backward(name):
    g = load(name)
    cores, pos, x = sbs_inputs(g)
    shapes = R.sbs_core_shapes(g["out_sizes"].tolist(), g["bond_sizes"].tolist(), int(g["C"]), int(g["q"]))
    assert [tuple(c.shape) for c in cores] == [tuple(s) for s in shapes]
    assert np.array_equal(np.array([s[:3] for s in shapes]), g["shapes"])
    y = R.convsbs_forward(cores, pos, x)
    assert torch.allclose(y, t(g["y"]), **TOL)
    gr = R.grads(lambda xx, *cc: R.convsbs_forward(cc, pos, xx), [x] + cores, t(g["dy"]))
    assert torch.allclose(gr[0], t(g["dx"]), **TOL)
    for i, gc in enumerate(gr[1:]):
        assert torch.allclose(gc, t(g[f"dcore{i}"]), **TOL)


@pytest.mark.parametrize("name", ["sbs_2x2_ring_perm0", "sbs_2x2_ring_perm1"])
def test_convsbs_as_eps(name):
    g = load(name)
    cores, pos, x = sbs_inputs(g)
    assert torch.allclose(R.convsbs_as_explicit_tensor(cores), t(g["explicit"]), **TOL)
    as_eps = R.convsbs_as_eps(cores, pos)
    assert as_eps.shape == (2,) * 8 + (24,)
    assert torch.allclose(as_eps, t(g["as_eps"]), **TOL)
    # tests/test_conversion_of_convsbs_to_eps.py:52: ConvSBS.forward == eps(as_eps, x)
    assert torch.allclose(R.eps_4step(as_eps, x), t(g["y"]), rtol=1e-9, atol=1e-11)


def test_logmatmulexp():
    g = load("logmatmulexp")
    for i in (0, 1):
        A, B = t(g[f"A{i}"]), t(g[f"B{i}"])
        assert torch.allclose(R.logmatmulexp(A, B), t(g[f"y{i}"]), **TOL)
        dA, dB = R.grads(R.logmatmulexp, [A, B], t(g[f"dy{i}"]))
        assert torch.allclose(dA, t(g[f"dA{i}"]), **TOL)
        assert torch.allclose(dB, t(g[f"dB{i}"]), **TOL)
    y2 = R.logmatmulexp(t(g["A2"]), t(g["B2"]))
    assert torch.equal(torch.isinf(y2), torch.isinf(t(g["y2"])))
    assert torch.all(y2[2] == -float("inf"))
    fin = torch.isfinite(y2)
    assert torch.allclose(y2[fin], t(g["y2"])[fin], **TOL)
    mats = t(g["fold_mats"])
    assert torch.allclose(R.logmatmulexp_fold(list(mats)), t(g["fold_y"]), rtol=1e-5, atol=1e-5)
    assert torch.allclose(R.logmatmulexp_fold_batched(mats[None])[0], t(g["fold_y"]), rtol=1e-5, atol=1e-5)


def test_window_statistics_against_the_reference_fixture():
    """make_windows + RankOneTensorsBatch of the reference vs the oracle's loops."""
    g = dict(np.load(os.path.join(GOLDEN, "window_stats.npz")))
    for tag in "abcd":
        x, K = torch.from_numpy(g[f"x_{tag}"]), int(g[f"K_{tag}"])
        total, sq = R.window_sums(x, K)
        assert np.isclose(float(total), float(g[f"sum_{tag}"]), rtol=1e-12)
        assert np.isclose(float(sq), float(g[f"sq_{tag}"]), rtol=1e-12)
        mean, var, factor = R.window_mean_var_factor(x, K)
        assert np.isclose(float(mean), float(g[f"mean_{tag}"]), rtol=1e-11)
        assert np.isclose(float(var), float(g[f"var_{tag}"]), rtol=1e-9)
        assert np.isclose(float(factor), float(g[f"factor_{tag}"]), rtol=1e-10)
