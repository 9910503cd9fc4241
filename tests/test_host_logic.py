"""CPU tests (no GPU): host-side mirror of the reference interface, C-ABI exports, and the
loud failure of the product path without a device.  The assertions marked "reference test"
restate /root/reference/tests/*.py against this package."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import dctn_amd
from dctn_amd import _lib
from dctn_amd.align import align, align_with_positions
from dctn_amd.contraction_path_cache import ContractionPathCache, contract
from dctn_amd.conv_sbs import ConvSBS, DumbNormalInitialization, KhrulkovNormalInitialization, ManyConvSBS
from dctn_amd.conv_sbs_spec import SBSCoreShape, SBSSpecCore, SBSSpecString
from dctn_amd.eps import EPS, eps, is_eps, matrix_shape
from oracle import ref_cpu as R
from dctn_amd.epses_composition import specs_to_full_specs
from dctn_amd.pos2d import Pos2D, index_to_pos, pos_to_index

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


# ------------------------------------------------------------------ C-ABI
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "dctn_amd.h")).read()
    declared = set(re.findall(r"\b(dctn_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), f"{name} not exported by {_lib.LIB_PATH}"
    assert _lib.lib().dctn_version() >= 100
    assert _lib.lib().dctn_strerror(-1) == b"inconsistent shape"


def test_argument_validation_without_gpu():
    """Validation happens on the host before any launch, so it is checkable here."""
    lib = _lib.lib()
    s = (ctypes.c_int64 * 5)(1, 1, 1, 1, 1)
    assert lib.dctn_eps_fwd(None, s, None, None, None, 0, 1, 1, 4, 4, 2, 3, 4, 0, 0, None) == _lib.ERR_NULL
    assert lib.dctn_eps_fwd(8, s, 8, 8, None, 0, 1, 1, 2, 2, 2, 3, 4, 0, 0, None) == _lib.ERR_BAD_SHAPE  # H < K
    assert lib.dctn_eps_fwd(8, s, 8, 8, None, 0, 1, 1, 4, 4, 2, 3, 4, 7, 0, None) == _lib.ERR_BAD_DTYPE
    assert lib.dctn_eps_bwd_workspace_bytes(1, 2, 8, 8, 2, 3, 4, 0, 0, 1, 1) > 0
    assert lib.dctn_logmatmulexp_fwd(8, 8, 8, None, 0, 1, 0, 3, 3, 0, 0, 0, None) == _lib.ERR_BAD_SHAPE


def test_eps_family_query_matches_the_dispatch_table():
    """dctn_eps_family (host-only): which kernel family a shape runs on (BASELINE configs)."""
    lib = _lib.lib()
    BF16, F32, F64 = (_lib._DTYPE_CODE[t] for t in (torch.bfloat16, torch.float32, torch.float64))
    assert lib.dctn_eps_family(1, 1024, 28, 28, 2, 3, 4, BF16, 0) == 1      # cfg2: bf16 register family
    assert lib.dctn_eps_family(1, 128, 28, 28, 2, 4, 4, F32, 0) == 2        # cfg3a layer 1: exact-f32 bigcore
    assert lib.dctn_eps_family(1, 128, 25, 25, 4, 3, 6, F32, 0) == 2        # cfg3a layer 2
    assert lib.dctn_eps_family(1, 64, 28, 28, 2, 4, 2, F64, 0) == 3         # cfg1: f64 matrix cores
    assert lib.dctn_eps_family(1, 128, 28, 28, 2, 4, 4, BF16, 0) == 3       # bf16 big core: two-halves GEMMs on the bf16 matrix cores
    assert lib.dctn_eps_family(2, 3, 7, 9, 3, 2, 3, BF16, 0) == 0           # bf16, odd Q: eps() routes it through float32
    assert lib.dctn_eps_family(1, 2, 5, 5, 3, 2, 2, F64, 0) == 0            # tiny: generic
    assert lib.dctn_eps_family(2, 3, 7, 9, 3, 2, 3, F32, 0) == 3            # odd Q in float32: two-halves GEMMs
    assert lib.dctn_eps_family(1, 2, 5, 5, 3, 9, 2, F32, 0) == -1           # kernel larger than the image


@pytest.mark.skipif(torch.cuda.is_available(), reason="with a GPU, CPU tensors are staged to it (tests/test_gpu_boundary.py)")
def test_product_path_has_no_cpu_fallback():
    """CPU tensors are accepted at the boundary by STAGING them to the GPU (`_lib.placement`); on a machine
    without a GPU there is nothing to stage to and no CPU implementation to fall back on."""
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        eps(torch.randn(2, 2, 2, 2, 3), torch.randn(1, 2, 4, 4, 2))
    from dctn_amd.logmatmulexp import logmatmulexp

    with pytest.raises(RuntimeError, match="no CPU implementation"):
        logmatmulexp(torch.randn(3, 4), torch.randn(4, 5))
    spec = SBSSpecString((SBSSpecCore(Pos2D(0, 0), 1), SBSSpecCore(Pos2D(0, 1), 2)), (1, 3), 1, 2)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        ConvSBS(spec)(torch.randn(1, 2, 3, 3, 2))
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd

    model = EPSesPlusLinear(((2, 3),), UnitTheoreticalOutputStd(), 1.0, torch.device("cpu"), torch.float32, image_size=5)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        model(torch.rand(1, 2, 5, 5, 2))
    # the autograd nodes themselves never take CPU tensors
    from dctn_amd.eps import _EpsFunction

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _EpsFunction.apply(torch.randn(2, 2, 2, 2, 3), torch.randn(1, 2, 4, 4, 2))


def test_product_code_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dctn_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"
    for f in os.listdir(os.path.join(ROOT, "dctn")):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "dctn", f)).read()


def test_shape_errors_are_assertion_errors():
    # reference: dctn/eps.py:22 asserts the core's input dims equal in_size
    with pytest.raises(AssertionError):
        eps(torch.randn(3, 3, 3, 3, 4), torch.randn(1, 2, 4, 4, 2))


# ------------------------------------------------------------------ pos2d / align (bit-exact)
def test_pos_index_conversion():  # reference test: tests/test_pos2d.py:4-31
    for max_w, cases in ((3, ((Pos2D(0, 0), 0), (Pos2D(1, 0), 4), (Pos2D(1, 1), 5), (Pos2D(2, 3), 11))),
                         (0, ((Pos2D(0, 0), 0), (Pos2D(1, 0), 1), (Pos2D(2, 0), 2), (Pos2D(3, 0), 3)))):
        for pos, index in cases:
            assert pos_to_index(max_w, pos) == index
            assert index_to_pos(max_w, index) == pos


def test_pos2d_golden_table():
    for max_w, idx, h, w, back in load("pos2d")["table"]:
        assert index_to_pos(int(max_w), int(idx)) == Pos2D(int(h), int(w))
        assert pos_to_index(int(max_w), Pos2D(int(h), int(w))) == int(back)


@pytest.mark.parametrize("K", [2, 3, 4])
def test_align_golden_index_map(K):
    g = load(f"align_k{K}")
    H, W = int(g["H"]), int(g["W"])
    ramp = torch.arange(H * W, dtype=torch.float64).reshape(1, 1, H, W, 1)
    views = torch.stack(list(align(ramp, K)))[:, 0, :, :, 0].to(torch.int64).numpy()
    assert np.array_equal(views, g["src"])


def test_align_with_positions_golden_and_asserts():
    g = load("align_snake")
    H, W = int(g["H"]), int(g["W"])
    ramp = torch.arange(H * W, dtype=torch.float64).reshape(1, 1, H, W, 1)
    pos = tuple(Pos2D(int(h), int(w)) for h, w in g["positions"])
    views = torch.stack(list(align_with_positions(ramp, pos)))[:, 0, :, :, 0].to(torch.int64).numpy()
    assert np.array_equal(views, g["src"])
    with pytest.raises(AssertionError):  # dctn/align.py:18-19
        list(align_with_positions(ramp, (Pos2D(1, 0), Pos2D(1, 1))))


# ------------------------------------------------------------------ spec
def test_all_dangling_dim_names():  # reference test: tests/test_conv_sbs_spec.py:5-35
    spec = SBSSpecString(
        (SBSSpecCore(Pos2D(0, 0), 1), SBSSpecCore(Pos2D(0, 1), 1), SBSSpecCore(Pos2D(1, 1), 2), SBSSpecCore(Pos2D(1, 0), 1)),
        bond_sizes=(5, 5, 5, 5), in_num_channels=3, in_quantum_dim_size=100,
    )
    expect = tuple(f"in_quantum_{c}_{k}" for k in range(4) for c in range(3)) + tuple(f"out_quantum_{k}" for k in range(4))
    assert spec.all_dangling_dim_names == expect


def test_spec_shapes_and_validators():
    cores = (SBSSpecCore(Pos2D(0, 0), 1), SBSSpecCore(Pos2D(0, 1), 3), SBSSpecCore(Pos2D(1, 0), 2), SBSSpecCore(Pos2D(1, 1), 4))
    spec = SBSSpecString(cores, (3, 4, 5, 6), 2, 2)
    assert [s.as_tuple() for s in spec.shapes] == [(1, 3, 4, 2, 2), (3, 4, 5, 2, 2), (2, 5, 6, 2, 2), (4, 6, 3, 2, 2)]
    assert spec.out_total_quantum_dim_size == 24 and len(spec) == 4
    assert spec.nelement == (1 * 4) * (3 * 4) * (2 * 4) * (4 * 4)
    assert spec.get_indices_wrt_standard_order() == (0, 1, 2, 3)
    assert spec.get_dim_names(3) == ("out_quantum_3", "bond_3", "bond_0", "in_quantum_0_3", "in_quantum_1_3")
    assert SBSCoreShape(2, 3, 4, 2, 5).dimensions_names == ("out_quantum", "bond_left", "bond_right", "in_quantum_0", "in_quantum_1")
    with pytest.raises(ValueError):
        SBSSpecString((SBSSpecCore(Pos2D(1, 0), 1), SBSSpecCore(Pos2D(1, 1), 1)), (1, 2), 1)
    with pytest.raises(ValueError):
        SBSSpecString(cores, (1, 2, 3), 1)
    g = load("sbs_2x2_ring_perm0")
    assert np.array_equal(np.array([s.as_tuple()[:3] for s in spec.shapes]), g["shapes"])


# ------------------------------------------------------------------ path cache
def test_contraction_path_cache_formats_bit_identical():  # reference test: tests/test_contraction_path_cache.py:6-26
    cache = ContractionPathCache()
    a, b = torch.randn(3, 4), torch.randn(4, 5)
    ab0 = cache.contract("ij,jk->ijk", a, b)
    for ab in (cache.contract("ij,jk->ijk", a, b), cache.contract(a, "ij", b, "jk", "ijk"),
               cache.contract(a, (0, 1), b, (1, 2), (0, 1, 2))):
        assert torch.all(ab == ab0)
    assert ContractionPathCache() is cache
    n = len(cache.paths)
    contract("ij,jk->ijk", a, b)
    assert len(cache.paths) == n  # memoised on shapes + subscripts
    assert torch.allclose(contract("ij,jk->ik", a, b), a @ b, atol=1e-6)
    assert torch.allclose(contract(a, ("x", "y"), ()), a.sum(), atol=1e-5)


# ------------------------------------------------------------------ parameter-only contractions
# (the product functions of these run on the device: tests/test_gpu_regulariser_init.py; here the ORACLE's restatement
# is pinned by the reference's closed forms and by the reference's own numbers)
def test_oracle_contract_on_inner_dims():  # reference test: tests/test_eps.py:64-73
    a = torch.einsum("oi,j->ijo", torch.eye(3), 2.0 * torch.ones(3))
    assert torch.allclose(R.contract_on_input_dims(a, a), 12.0 * torch.eye(3))
    a = torch.einsum("oi,j->ijo", 2.0 * torch.eye(4), torch.tensor([1.0, 2.0, 3.0, 4.0]))
    b = torch.einsum("pj,i->ijp", 3.0 * torch.eye(4), torch.ones(4))
    assert torch.allclose(R.contract_on_input_dims(a, b),
                          torch.einsum("o,p->op", 2.0 * torch.ones(4), torch.tensor([3.0, 6.0, 9.0, 12.0])))


def test_oracle_epses_inner_product_closed_forms():  # reference test: tests/test_epses_composition.py:7-41
    inner_product = R.epses_inner_product
    a = torch.einsum("oi,j->ijo", torch.eye(3), torch.ones(3))
    assert torch.allclose(inner_product((a,), (a,)), torch.tensor(9.0))
    assert torch.allclose(inner_product((a, a), (a, a)), torch.tensor(3.0**4))
    assert torch.allclose(inner_product((a, a, a), (a, a, a)), torch.tensor(3.0**8))
    green = torch.einsum("oj,i->ijo", torch.eye(6)[:4], torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0]))
    black = torch.einsum("oi,j->ijo", torch.eye(4)[:3], torch.tensor([1.5, 0.0, 0.0, 0.0]))
    orange = torch.einsum("oi,j->ijo", torch.eye(6)[:4], torch.ones(6))
    red = torch.einsum("oi,j->ijo", torch.eye(4)[1:], torch.tensor([1.0, 0.0, 0.0, 1.0]))
    assert torch.allclose(inner_product((green, black), (orange, red)), torch.tensor((2 + 3 + 4) * 5 * 1.5))


def test_oracle_inner_product_and_init_match_the_reference_fixtures():
    g = load("inner_product")
    for tag in ("s1", "s2", "s3"):
        n = int(g[f"{tag}_n"])
        a = [torch.from_numpy(g[f"{tag}_a{i}"]).requires_grad_(True) for i in range(n)]
        b = [torch.from_numpy(g[f"{tag}_b{i}"]).requires_grad_(True) for i in range(n)]
        val = R.epses_inner_product(a, b)
        assert torch.allclose(val, torch.from_numpy(g[f"{tag}_value"]), rtol=1e-12)
        grads = torch.autograd.grad(val, a + b)
        for i in range(n):
            assert torch.allclose(grads[i], torch.from_numpy(g[f"{tag}_da{i}"]), rtol=1e-10, atol=1e-13)
            assert torch.allclose(grads[n + i], torch.from_numpy(g[f"{tag}_db{i}"]), rtol=1e-10, atol=1e-13)
        assert torch.allclose(R.epses_inner_product(a, a), torch.from_numpy(g[f"{tag}_self_value"]), rtol=1e-12)
    g = load("empirical_std_init")
    x, batch = torch.from_numpy(g["x"]), int(g["batch_size"])
    torch.manual_seed(int(g["seed"]))
    raw = torch.randn(*(2,) * 9, 4, dtype=torch.float64)
    scale = R.unit_empirical_output_std_scale(raw, x, batch)
    assert torch.allclose(scale, torch.from_numpy(g["one_inverse_output_std"]), rtol=1e-12)
    assert torch.allclose(raw * scale, torch.from_numpy(g["one_core"]), rtol=1e-12)


@pytest.mark.parametrize("name", ["sbs_2x2_ring_perm0", "sbs_2x2_ring_perm1"])
def test_convsbs_parameter_contractions_match_reference(name):
    g = load(name)
    spec = SBSSpecString(
        tuple(SBSSpecCore(Pos2D(int(h), int(w)), int(o)) for (h, w), o in zip(g["positions"], g["out_sizes"])),
        tuple(int(b) for b in g["bond_sizes"]), int(g["C"]), int(g["q"]),
    )
    m = ConvSBS(spec).double()
    with torch.no_grad():
        for i, c in enumerate(m.cores):
            c.copy_(torch.from_numpy(g[f"core{i}"]))
        assert m.as_eps().shape == (2,) * 8 + (24,)
        assert torch.all(m.as_eps() == m.as_eps())  # deterministic (tests/test_conversion...:33)
        assert torch.allclose(m.as_eps(), torch.from_numpy(g["as_eps"]), rtol=1e-10, atol=1e-12)
        assert torch.allclose(m.as_explicit_tensor(), torch.from_numpy(g["explicit"]), rtol=1e-10, atol=1e-12)
        assert torch.allclose(m.sum(), torch.from_numpy(g["tt_sum"]), rtol=1e-10)
        assert torch.allclose(m.squared_fro_norm(), torch.from_numpy(g["tt_sqnorm"]), rtol=1e-10)
        assert torch.allclose(m.var(), torch.from_numpy(g["tt_var"]), rtol=1e-9)
        explicit = m.as_explicit_tensor()  # reference test: tests/test_conv_sbs.py:52-57
        assert torch.allclose(m.var(), explicit.var(), rtol=1e-9)
        assert torch.allclose(m.mean(), explicit.mean(), rtol=1e-9)
        assert torch.allclose(m.fro_norm(), explicit.norm(), rtol=1e-9)


def test_khrulkov_init_reaches_requested_std():  # reference test: tests/test_conv_sbs.py:10-50 (reduced draws)
    spec = SBSSpecString(tuple(SBSSpecCore(Pos2D(0, i), 2 if i == 1 else 1) for i in range(4)), (1, 3, 3, 3), 1, 2)
    stds = [float(ConvSBS(spec, KhrulkovNormalInitialization(0.5)).as_explicit_tensor().std()) for _ in range(300)]
    assert abs(np.sqrt(np.mean(np.square(stds))) - 0.5) / 0.5 < 0.3


def test_module_surfaces():
    e = EPS(3, 1, 2, 4)
    assert e.core.shape == (2,) * 9 + (4,) and e.core.dtype == torch.float32 and e.core.device.type == "cpu"
    assert e.matrix_shape == (4, 512) and is_eps(e.core) and matrix_shape(e.core) == (4, 512)
    assert specs_to_full_specs(((4, 4), (3, 6)), 2) == (
        dict(kernel_size=4, in_num_channels=1, in_size=2, out_size=4),
        dict(kernel_size=3, in_num_channels=1, in_size=4, out_size=6),
    )
    many = ManyConvSBS(1, 2, 3, False, ((SBSSpecCore(Pos2D(0, 0), 1), SBSSpecCore(Pos2D(0, 1), 2)),) * 2)
    assert [tuple(c.shape) for c in many.strings[0].cores] == [(1, 1, 3, 2), (2, 3, 1, 2)]
    ring = ManyConvSBS(2, 2, 3, True, ((SBSSpecCore(Pos2D(0, 0), 1), SBSSpecCore(Pos2D(0, 1), 2)),),
                       (DumbNormalInitialization(0.5),))
    assert [tuple(c.shape) for c in ring.strings[0].cores] == [(1, 3, 3, 2, 2), (2, 3, 3, 2, 2)]
    import dctn.eps as alias  # drop-in module path

    assert alias.eps is eps


def test_runner_import_lines_resolve_against_the_alias_package():
    """new_runner.py:14-59 imports these names from the package's modules on the path (the runner's other imports -
    dataset_loading, tb_logging, libcrap, ignite - are out of scope); dctn/utils.py:20-31,50-51."""
    from dctn.utils import (  # noqa: F401  (new_runner.py:51-59, the line as the runner has it)
        implies,
        xor,
        exactly_one_true,
        ZeroCenteredNormalInitialization,
        ZeroCenteredUniformInitialization,
        FromFileInitialization,
        OneTensorInitialization,
    )
    from dctn.utils import raise_exception, transform_dataset, id_assert_shape_matches  # noqa: F401
    from dctn.eps_plus_linear import (  # noqa: F401  (new_runner.py:25-31)
        EPSesPlusLinear,
        UnitEmpiricalOutputStd,
        UnitTheoreticalOutputStd,
        ManuallyChosenInitialization,
    )
    from dctn.evaluation import score  # noqa: F401
    from dctn.training import train, every_n_iters_intervals  # noqa: F401

    assert implies(False, False) and implies(False, True) and implies(True, True) and not implies(True, False)
    assert xor() is False and xor(True) is True and xor(True, True) is False and xor(True, False, False) is True
    assert xor(True, True, True) is True   # parity, as the reference's reduce
    assert exactly_one_true(False, True, False) and not exactly_one_true(True, True) and not exactly_one_true()
    with pytest.raises(AssertionError):
        exactly_one_true(1, 0)   # dctn/utils.py:29: genuine bools only
    with pytest.raises(KeyError):
        raise_exception(KeyError("x"))


def test_eps_plus_linear_ctor_state_dict_and_manual_init():  # cf. reference tests/test_eps_plus_linear.py:13-36
    from dctn_amd.eps_plus_linear import EPSesPlusLinear, ManuallyChosenInitialization, UnitTheoreticalOutputStd
    from dctn_amd.utils import ZeroCenteredNormalInitialization, ZeroCenteredUniformInitialization

    for dtype in (torch.float32, torch.float64):
        m = EPSesPlusLinear(((3, 4), (2, 3)), UnitTheoreticalOutputStd(), 1.0, torch.device("cpu"), dtype, image_size=10)
        sd = m.state_dict()
        assert list(sd) == ["p", "epses.0", "epses.1", "linear.weight", "linear.bias"]
        assert sd["epses.0"].shape == (2,) * 9 + (4,) and sd["epses.1"].shape == (4,) * 4 + (3,)
        assert sd["linear.weight"].shape == (10, 7 * 7 * 3) and sd["linear.weight"].dtype == dtype
    for p in (1e-3, 0.4, 1.0):
        m = EPSesPlusLinear(
            ((2, 3),),
            ManuallyChosenInitialization((ZeroCenteredUniformInitialization(0.25),), ZeroCenteredNormalInitialization(0.1),
                                         ZeroCenteredUniformInitialization(0.5)),
            p, torch.device("cpu"), torch.float32, image_size=6)
        assert m.epses[0].abs().max() <= 0.25 and m.linear.bias.abs().max() <= 0.5
        assert float(m.p) == pytest.approx(p)
    with pytest.raises(ValueError):
        EPSesPlusLinear(((2, 3),), object(), 1.0, torch.device("cpu"), torch.float32)


def test_rank_one_tensors_batch_and_make_windows():
    """Host mirror of dctn/rank_one_tensor.py and dctn/align.py:49-61 against the fixture produced by
    the reference (incl. the values of its own tests/test_rank_one_tensor.py:8-45)."""
    import numpy as np

    from dctn.rank_one_tensor import RankOneTensorsBatch      # alias package path
    from dctn_amd.align import make_windows

    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "window_stats.npz")))
    b = RankOneTensorsBatch(torch.from_numpy(g["basic_array"]), factors_dim=1, coordinates_dim=2)
    assert b.batch_shape == (2, 1) and b.ntensors == 2 and b.ncoordinates == 4
    assert torch.allclose(b.sum_per_tensor(), torch.from_numpy(g["basic_sum_per_tensor"]))
    assert torch.allclose(b.squared_fro_norm_per_tensor(), torch.from_numpy(g["basic_sq_per_tensor"]))
    assert torch.allclose(b.sum_over_batch(), torch.tensor(-93.0)) and torch.allclose(b.mean_over_batch(), torch.tensor(-11.625))
    assert torch.allclose(b.var_over_batch(), torch.from_numpy(g["basic_var"])) and torch.allclose(b.std_over_batch(), torch.from_numpy(g["basic_std"]))
    assert torch.allclose(b.std_over_batch(unbiased=False), b.std_over_batch())    # as in the reference
    with pytest.raises(AssertionError):
        RankOneTensorsBatch(torch.zeros(2, 2), 1, 1)
    for tag in "abcd":
        x, K = torch.from_numpy(g[f"x_{tag}"]), int(g[f"K_{tag}"])
        w = make_windows(x, K)
        assert w.array.shape[0] == K * K * x.shape[0] and w.factors_dim == 0 and w.coordinates_dim == 4
        assert np.isclose(float(w.mean_over_batch()), float(g[f"mean_{tag}"]), rtol=1e-12)
        assert np.isclose(float(w.var_over_batch()), float(g[f"var_{tag}"]), rtol=1e-10)


def test_training_hooks_mirror_the_reference(tmp_path):  # dctn/training.py:87-248
    """every_n_iters_intervals gating, the two checkpointers (names, retention), the early stopper, the
    iteration-count stopper and the NaN stopper — model-agnostic host logic, exercised on a CPU model."""
    from dctn_amd import training as T

    calls = []
    gated = T.every_n_iters_intervals((4, 2), (6, 3), (None, 5))(lambda sx, si: calls.append(si["num_iters_done"]))
    for n in range(21):
        gated({}, {"num_iters_done": n})
    assert calls == [0, 2, 6, 9, 10, 15, 20]   # period 2 on [0,4), 3 on [4,10), 5 from 10 on
    calls.clear()
    always_after = T.every_n_iters_intervals((3, 2))(lambda sx, si: calls.append(si["num_iters_done"]))
    for n in range(6):
        always_after({}, {"num_iters_done": n})
    assert calls == [0, 2, 3, 4, 5]

    model = torch.nn.Linear(3, 2)
    st_x = {"model": model}

    def it(n, vacc, vmce):
        return {"num_iters_done": n, "train_acc": 0.5, "val_acc": vacc, "train_mean_ce": 1.25, "val_mean_ce": vmce}

    last = T.LastModelsCheckpointer(str(tmp_path), 2)
    best = T.BestModelCheckpointer(str(tmp_path), "val_acc", low_is_good=False)
    for n, (vacc, vmce) in enumerate([(0.1, 3.0), (0.3, 2.0), (0.2, 2.5), (0.25, 2.6)]):
        last(st_x, it(n, vacc, vmce))
        best(st_x, it(n, vacc, vmce))
    assert sorted(os.listdir(tmp_path)) == [
        "model_best_val_acc_nitd=0000001_tracc=0.5000_vacc=0.3000_trmce=1.2500_vmce=2.0000.pth",
        "model_nitd=0000002_tracc=0.5000_vacc=0.2000_trmce=1.2500_vmce=2.5000.pth",
        "model_nitd=0000003_tracc=0.5000_vacc=0.2500_trmce=1.2500_vmce=2.6000.pth"]
    state = torch.load(os.path.join(tmp_path, sorted(os.listdir(tmp_path))[0]))
    assert set(state) == {"weight", "bias"}

    stopper = T.ValuesNotImprovingEarlyStopper(2, (("val_acc", False), ("val_mean_ce", True)))
    history = [(0.1, 3.0), (0.2, 3.1), (0.2, 3.0), (0.2, 3.0), (0.19, 2.9), (0.1, 3.0), (0.1, 3.0), (0.1, 3.0)]
    stopped_at = None
    for n, (vacc, vmce) in enumerate(history):
        st_it = dict(it(n, vacc, vmce), stop=False)
        stopper(st_x, st_it)
        if st_it["stop"]:
            stopped_at = n
            break
    assert stopped_at == 7        # improvements at 0, 1, 4; calls 5, 6, 7 bring none: the third exceeds patience 2

    # the loop itself: hooks in order, stop after n, NaN stop with the dump of the offending batch
    x, y = torch.randn(6, 3), torch.randint(0, 2, (6,))
    order = []
    st_x2, st_it2 = T.train([(x, y, torch.arange(6))], model, torch.optim.SGD(model.parameters(), lr=0.1), torch.device("cpu"),
                            torch.nn.functional.cross_entropy, lambda sx, si: torch.zeros(()), 0.0,
                            at_iter_start=[lambda sx, si: order.append("start")],
                            after_back=[lambda sx, si: order.append("back")],
                            after_param_upd=[lambda sx, si: order.append("upd"), T.make_stopper_after_n_iters(2)])
    assert st_it2["num_iters_done"] == 2 and order == ["start", "back", "upd"] * 3
    assert set(st_it2) >= {"x", "y", "indices", "output", "loss", "reg_term", "stop"} and st_x2["model"] is model
    nan_dir = tmp_path / "nan"
    nan_dir.mkdir()
    _, st_it3 = T.train([(x, y, torch.arange(6))], model, torch.optim.SGD(model.parameters(), lr=0.1), torch.device("cpu"),
                        lambda out, yy: out.sum() * float("nan"), lambda sx, si: torch.zeros(()), 0.0,
                        at_iter_start=[], after_back=[T.make_stopper_on_nan_loss(str(nan_dir), False)], after_param_upd=[])
    assert st_it3["num_iters_done"] == 0 and st_it3["stop"]
    assert sorted(os.listdir(nan_dir / "nan_loss_stop"))[:3] == ["indices.pth", "model_nitd=0_loss=nan_reg_term=0.000.pth", "output.pth"]


@pytest.mark.skipif(torch.cuda.is_available(), reason="on a GPU box the probe is exercised by tests/test_gpu_ddp.py")
def test_allreduce_capture_probe_says_no_without_a_gpu():
    """The probe child cannot come up without a GPU: the parent must get a plain False (and survive), which sends
    the data-parallel step down the eager-collective path."""
    from dctn_amd import ddp

    assert ddp.probe_allreduce_capture(timeout=120.0) is False
    assert ddp.all_ranks_agree(True) is True and ddp.all_ranks_agree(False) is False


def test_allreduce_capture_probe_child_opens_the_launchers_gpu(monkeypatch):
    """Every rank calls the probe BEFORE `torch.cuda.set_device(local_rank)` (bench.py): the child's GPU must then be the
    launcher's LOCAL_RANK - not `torch.cuda.current_device()`, which is 0 on every rank at that point (all N children on
    cuda:0: RCCL refuses the duplicate GPU and the multi-GPU run silently loses the graphed collective) - and asking
    must not create a GPU context in the parent."""
    import subprocess

    from dctn_amd import ddp

    seen = {}

    class _Child:
        def wait(self, timeout=None):
            return 0

    def fake_popen(cmd, cwd=None, env=None, **kw):
        seen.update(env)
        return _Child()

    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    for rank in (0, 1, 5):
        monkeypatch.setenv("RANK", str(rank))
        monkeypatch.setenv("LOCAL_RANK", str(rank))
        monkeypatch.setenv("WORLD_SIZE", "8")
        monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
        monkeypatch.setenv("MASTER_PORT", "29511")
        assert ddp.probe_allreduce_capture(timeout=1.0) is True
        assert seen["LOCAL_RANK"] == str(rank) and seen["RANK"] == str(rank) and seen["WORLD_SIZE"] == "8"
        assert seen["MASTER_PORT"] == str(29511 + 53)
    assert ddp.probe_allreduce_capture(timeout=1.0, local_rank=3) is True and seen["LOCAL_RANK"] == "3"
    monkeypatch.delenv("LOCAL_RANK")
    assert ddp._probe_child_device() == (torch.cuda.current_device() if torch.cuda.is_initialized() else 0)
    if not torch.cuda.is_available():
        assert not torch.cuda.is_initialized()


def test_eps_plus_linear_model_is_picklable_and_refreshes_p_on_load():
    """The dropout gate's host copy of `p` is refreshed by a module-level load_state_dict hook (a lambda stored on the
    module made `torch.save(model)` / spawn arguments fail)."""
    import pickle

    from dctn_amd.eps_plus_linear import EPSesPlusLinear, UnitTheoreticalOutputStd

    m = EPSesPlusLinear(((2, 2),), UnitTheoreticalOutputStd(), 0.5, torch.device("cpu"), torch.float32, image_size=6)
    m2 = pickle.loads(pickle.dumps(m))
    sd = m.state_dict()
    sd["p"] = torch.tensor(0.25)
    m2.load_state_dict(sd)
    assert m2._p_float == 0.25 and m._p_float == 0.5
