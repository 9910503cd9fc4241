#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (authoring container only).

Usage (needs /root/reference, which never travels to the GPU box):
    python tests/golden/make_golden.py

What runs: the reference's own ``dctn`` package imported from /root/reference.
``dctn.eps`` / ``dctn.conv_sbs`` import two third-party packages that are not installed
here (``opt_einsum``, ``more_itertools``).  Neither carries arithmetic of its own:
``opt_einsum`` only sequences pairwise torch einsum calls along a path.  This script
therefore registers two small sequencing modules of OUR OWN (below) under those names
before importing the reference: the explicit 4-step path of dctn/eps.py:25-30 is executed
step by step exactly as given; for ``optimize="auto-hq"`` expressions the operands are
contracted left to right (pairwise order may differ from real opt_einsum's choice, which
changes fp rounding only).  The fixtures are therefore "reference glue + torch arithmetic".

Output: tests/golden/*.npz (inputs + expected outputs, float64 unless stated).
"""
import itertools
import os
import sys
import types
import zlib

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- sequencing modules
def _parse(args):
    """Returns (tensors_or_shapes, input_subscripts(list of tuples), output_subscripts(tuple))."""
    if isinstance(args[0], str):
        expr = args[0].replace(" ", "")
        lhs, rhs = expr.split("->")
        subs = [tuple(s) for s in lhs.split(",")]
        return list(args[1:]), subs, tuple(rhs)
    ops, subs = [], []
    rest = list(args)
    while len(rest) >= 2:
        ops.append(rest.pop(0))
        subs.append(tuple(rest.pop(0)))
    assert len(rest) == 1, "interleaved format needs an explicit output"
    return ops, subs, tuple(rest[0])


def _einsum_named(ops, subs, out):
    names = {}
    for s in list(subs) + [out]:
        for n in s:
            names.setdefault(n, len(names))
    flat = []
    for t, s in zip(ops, subs):
        flat += [t, [names[n] for n in s]]
    flat.append([names[n] for n in out])
    return torch.einsum(*flat)


def _run(ops, subs, out, optimize):
    ops, subs = list(ops), list(subs)
    if isinstance(optimize, (tuple, list)):
        for step in optimize:
            idx = sorted(step, reverse=True)
            t_ops = [ops.pop(i) for i in idx][::-1]
            t_subs = [subs.pop(i) for i in idx][::-1]
            keep = set(out)
            for s in subs:
                keep |= set(s)
            new_sub = []
            for s in t_subs:
                for n in s:
                    if n in keep and n not in new_sub:
                        new_sub.append(n)
            ops.append(_einsum_named(t_ops, t_subs, tuple(new_sub)))
            subs.append(tuple(new_sub))
        if len(ops) == 1 and subs[0] == tuple(out):
            return ops[0]
        return _einsum_named(ops, subs, out)
    # string strategies ("auto-hq"): pairwise, left to right
    while len(ops) > 1:
        a, b = ops.pop(0), ops.pop(0)
        sa, sb = subs.pop(0), subs.pop(0)
        keep = set(out)
        for s in subs:
            keep |= set(s)
        new_sub = [n for n in dict.fromkeys(sa + sb) if n in keep]
        ops.insert(0, _einsum_named([a, b], [sa, sb], tuple(new_sub)))
        subs.insert(0, tuple(new_sub))
    return _einsum_named(ops, subs, out)


def _install_sequencers():
    torch.backends.opt_einsum.enabled = False
    oe = types.ModuleType("opt_einsum")

    def contract(*args, optimize="auto", **kw):
        ops, subs, out = _parse(args)
        return _run(ops, subs, out, optimize)

    class ContractExpression:
        def __init__(self, args, optimize):
            self.args, self.optimize = args, optimize

        def __call__(self, *tensors):
            _, subs, out = _parse(self.args)
            return _run(tensors, subs, out, self.optimize)

    def contract_expression(*args, optimize="auto", **kw):
        return ContractExpression(args, optimize)

    oe.contract = contract
    oe.contract_expression = contract_expression
    oe_contract = types.ModuleType("opt_einsum.contract")
    oe_contract.ContractExpression = ContractExpression
    oe.__path__ = []
    sys.modules["opt_einsum"] = oe
    sys.modules["opt_einsum.contract"] = oe_contract

    mi = types.ModuleType("more_itertools")

    def chunked(it, n):
        it = iter(it)
        while chunk := list(itertools.islice(it, n)):
            yield chunk

    def last(it):
        x = None
        for x in it:
            pass
        return x

    def intersperse(e, it):
        first = True
        for x in it:
            if not first:
                yield e
            first = False
            yield x

    mi.chunked, mi.last, mi.intersperse = chunked, last, intersperse
    mi.ilen = lambda it: sum(1 for _ in it)
    sys.modules["more_itertools"] = mi


# --------------------------------------------------------------------------- fixtures
def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: tuple(np.shape(v)) for k, v in out.items()})


def phi(u):
    """dataset_loading.py:33-36 feature map with nu=1."""
    return torch.stack((torch.sin(u * torch.pi / 2) ** 2, torch.cos(u * torch.pi / 2) ** 2), dim=-1)


def main():
    assert os.path.isdir(REF), "reference not mounted: this script only runs in the authoring container"
    _install_sequencers()
    sys.path.insert(0, REF)
    from dctn import align as r_align
    from dctn import conv_sbs as r_sbs
    from dctn import eps as r_eps
    from dctn import epses_composition as r_comp
    from dctn import logmatmulexp as r_lme
    from dctn.conv_sbs_spec import SBSSpecCore, SBSSpecString
    from dctn.pos2d import Pos2D, index_to_pos, pos_to_index

    f64 = torch.float64

    # ---- pos2d / align index maps (bit-exact)
    rows = []
    for max_w in (0, 1, 3, 7):
        for idx in range(0, 3 * (max_w + 1)):
            p = index_to_pos(max_w, idx)
            rows.append((max_w, idx, p.h, p.w, pos_to_index(max_w, p)))
    npz("pos2d", table=np.array(rows, dtype=np.int64))

    H, W = 6, 7
    ramp = torch.arange(H * W, dtype=f64).reshape(1, 1, H, W, 1)  # value == flat pixel index
    for K in (2, 3, 4):
        views = list(r_align.align(ramp, K))
        npz(f"align_k{K}", src=torch.stack(views)[:, 0, :, :, 0].to(torch.int64), H=H, W=W, K=K)
    snake = [(0, 0), (0, 1), (0, 2), (1, 2), (1, 1), (1, 0), (2, 0), (2, 1), (2, 2)]
    views = list(r_align.align_with_positions(ramp, tuple(Pos2D(*p) for p in snake)))
    npz("align_snake", src=torch.stack(views)[:, 0, :, :, 0].to(torch.int64), H=H, W=W,
        positions=np.array(snake, dtype=np.int64))

    # ---- EPS cases: (name, C, B, H, W, Q, K, O)
    eps_cases = [
        ("eps_c1_k3_q2_o4", 1, 3, 8, 8, 2, 3, 4),      # cfg2 shape (MNIST 3x3, out 4)
        ("eps_c1_k4_q2_o2", 1, 2, 7, 7, 2, 4, 2),      # cfg1 / eps2d benchmark shape
        ("eps_c1_k3_q3_o2", 1, 2, 5, 5, 3, 3, 2),      # N=9 with odd Q (cfg3a layer-2 family, reduced Q,O)
        ("eps_c1_k2_q4_o6", 1, 2, 5, 5, 4, 2, 6),      # Q=4, O=6 as cfg3a layer 2, reduced K
        ("eps_c1_k2_q8_o8", 1, 2, 5, 5, 8, 2, 8),      # cfg3b layer 2
        ("eps_c2_k2_q2_o4", 2, 3, 4, 5, 2, 2, 4),      # tests/test_eps.py:9 shape (C=2)
        ("eps_c1_k2_q3_o5", 1, 2, 5, 4, 3, 2, 5),      # odd sizes
        ("eps_c2_k1_q3_o2", 2, 2, 3, 3, 3, 1, 2),      # K=1, two channels (N=2)
    ]
    for name, C, B, Hh, Ww, Q, K, O in eps_cases:
        torch.manual_seed(zlib.crc32(name.encode()))
        x = torch.randn(C, B, Hh, Ww, Q, dtype=f64, requires_grad=True)
        core = (torch.randn(*(Q,) * (K * K * C), O, dtype=f64) * Q ** (-K * K * C / 4)).requires_grad_(True)
        y = r_eps.eps(core, x)
        dy = torch.randn_like(y)
        dx, dcore = torch.autograd.grad(y, (x, core), dy)
        y1 = r_eps.eps_one_by_one(core, x)
        npz(name, x=x, core=core, y=y, y_one_by_one=y1, dy=dy, dx=dx, dcore=dcore)

    # ---- stacked EPSes (cfg3-like, reduced so the fixture stays small): (3,3),(2,5) on 7x7
    torch.manual_seed(1234)
    x = phi(torch.rand(1, 2, 7, 7, dtype=f64)).requires_grad_(True)
    e1 = (torch.randn(*(2,) * 9, 3, dtype=f64) * 2 ** -2.25).requires_grad_(True)
    e2 = (torch.randn(*(3,) * 4, 5, dtype=f64) * 3 ** -1.0).requires_grad_(True)
    y = r_comp.contract_with_input((e1, e2), x)
    dy = torch.randn_like(y)
    dx, de1, de2 = torch.autograd.grad(y, (x, e1, e2), dy)
    npz("epses_composition_33_25", x=x, e1=e1, e2=e2, y=y, dy=dy, dx=dx, de1=de1, de2=de2)

    # ---- ConvSBS
    def run_sbs(name, cores_spec, bond_sizes, C, q, B, Hh, Ww, seed, std=0.7, with_eps=False):
        torch.manual_seed(seed)
        spec = SBSSpecString(
            tuple(SBSSpecCore(Pos2D(*p), o) for p, o in cores_spec), tuple(bond_sizes), C, q
        )
        m = r_sbs.ConvSBS(spec, r_sbs.DumbNormalInitialization(std)).double()
        x = torch.randn(C, B, Hh, Ww, q, dtype=f64, requires_grad=True)
        y = m(x)
        dy = torch.randn_like(y)
        g = torch.autograd.grad(y, (x, *m.cores), dy)
        arrays = dict(
            x=x, y=y, dy=dy, dx=g[0],
            positions=np.array([p for p, _ in cores_spec], dtype=np.int64),
            out_sizes=np.array([o for _, o in cores_spec], dtype=np.int64),
            bond_sizes=np.array(bond_sizes, dtype=np.int64), C=C, q=q,
        )
        for i, (c, gc) in enumerate(zip(m.cores, g[1:])):
            arrays[f"core{i}"] = c
            arrays[f"dcore{i}"] = gc
        arrays["shapes"] = np.array([s.as_tuple()[:3] for s in spec.shapes], dtype=np.int64)
        if with_eps:
            with torch.no_grad():
                arrays["as_eps"] = m.as_eps()
                arrays["explicit"] = m.as_explicit_tensor()
                arrays["tt_sum"] = m.sum()
                arrays["tt_sqnorm"] = m.squared_fro_norm()
                arrays["tt_var"] = m.var()
        npz(name, **arrays)

    snake_spec = [(p, 2 if i == 4 else 1) for i, p in enumerate(snake)]  # mnist.py:190-199
    snake2 = [(0, 0), (1, 0), (2, 0), (2, 1), (1, 1), (0, 1), (0, 2), (1, 2), (2, 2)]  # mnist.py:201-210
    snake2_spec = [(p, 2 if i == 4 else 1) for i, p in enumerate(snake2)]
    run_sbs("sbs_snake_r4_c1_q2", snake_spec, (1,) + (4,) * 8, 1, 2, 2, 6, 6, 11)
    run_sbs("sbs_snake_r2_c1_q3", snake_spec, (1,) + (2,) * 8, 1, 3, 2, 5, 6, 12)     # CIFAR colour as q=3
    run_sbs("sbs_snake2_r4_c2_q2", snake2_spec, (1,) + (4,) * 8, 2, 2, 2, 5, 5, 13, std=0.5)  # 2nd layer C=2
    run_sbs("sbs_snake_ring_r3_c1_q2", snake_spec, (3,) * 9, 1, 2, 2, 5, 5, 14, std=0.6)  # trace_edge=True
    # the ring case of tests/test_conversion_of_convsbs_to_eps.py:13-56, two of the 24 permutations
    base = [((0, 0), 1), ((0, 1), 3), ((1, 0), 2), ((1, 1), 4)]
    run_sbs("sbs_2x2_ring_perm0", base, (3, 4, 5, 6), 2, 2, 3, 4, 5, 15, with_eps=True)
    perm = [base[2], base[0], base[3], base[1]]
    run_sbs("sbs_2x2_ring_perm1", perm, (3, 4, 5, 6), 2, 2, 3, 4, 5, 16, with_eps=True)

    # ---- logmatmulexp (dctn/logmatmulexp.py is importable as-is)
    torch.manual_seed(77)
    cases = {}
    A = torch.randn(5, 7, dtype=f64, requires_grad=True)
    Bm = torch.randn(7, 3, dtype=f64, requires_grad=True)
    y = r_lme.logmatmulexp(A, Bm)
    dy = torch.randn_like(y)
    dA, dB = torch.autograd.grad(y, (A, Bm), dy)
    cases.update(A0=A, B0=Bm, y0=y, dy0=dy, dA0=dA, dB0=dB)
    # large magnitudes as in small_experiments/logmatmulexp_old.py:149-152
    A = (torch.randn(16, 16, dtype=f64) * 150).requires_grad_(True)
    Bm = (torch.randn(16, 16, dtype=f64) * 150).requires_grad_(True)
    y = r_lme.logmatmulexp(A, Bm)
    dy = torch.randn_like(y)
    dA, dB = torch.autograd.grad(y, (A, Bm), dy)
    cases.update(A1=A, B1=Bm, y1=y, dy1=dy, dA1=dA, dB1=dB)
    # -inf entries: one full -inf row of A, scattered -inf in B (forward only)
    A = torch.randn(6, 5, dtype=f64)
    A[2] = -float("inf")
    A[0, 1] = -float("inf")
    Bm = torch.randn(5, 4, dtype=f64)
    Bm[3, 2] = -float("inf")
    cases.update(A2=A, B2=Bm, y2=r_lme.logmatmulexp(A, Bm))
    # float32 left fold of 6 square matrices (logmatmulexp_benchmark/benchmark.py:23-30), dim 32
    mats = [torch.randn(32, 32, dtype=torch.float32) for _ in range(6)]
    import functools
    cases.update(fold_mats=torch.stack(mats), fold_y=functools.reduce(r_lme.logmatmulexp, mats))
    npz("logmatmulexp", **cases)


def window_stats_fixture():
    """Window statistics: the reference's own make_windows (dctn/align.py:49-61) and
    RankOneTensorsBatch (dctn/rank_one_tensor.py) on seeded inputs; the scaling factor is the last
    line of calc_scaling_factor (dctn/dataset_loading.py:94) applied to those two numbers (the function
    itself needs torchvision, which is not installed)."""
    assert os.path.isdir(REF)
    sys.path.insert(0, REF)
    from dctn.align import make_windows
    from dctn.rank_one_tensor import RankOneTensorsBatch

    torch.manual_seed(zlib.crc32(b"window_stats"))
    out = {}
    for tag, (C, B, H, W, Q, K) in {"a": (1, 6, 7, 7, 2, 2), "b": (1, 5, 8, 9, 2, 3), "c": (2, 4, 6, 6, 3, 2),
                                    "d": (1, 3, 9, 9, 2, 4)}.items():
        if Q == 2 and C == 1:
            x = 2 * phi(torch.rand(B, H, W, dtype=torch.float64)).unsqueeze(0)   # the reference's feature map
        else:
            x = torch.rand(C, B, H, W, Q, dtype=torch.float64) + 0.25
        r1 = make_windows(x, K)
        mean, var = r1.mean_over_batch(), r1.var_over_batch()
        out.update({f"x_{tag}": x, f"K_{tag}": K, f"mean_{tag}": mean, f"var_{tag}": var,
                    f"sum_{tag}": r1.sum_over_batch(), f"sq_{tag}": r1.squared_fro_norm_over_batch(),
                    f"factor_{tag}": (mean**2 + var) ** (-1 / (2 * K**2))})
    basic = RankOneTensorsBatch(
        array=torch.tensor([[[[1.0], [2.0]], [[2.0], [-3.0]]], [[[4.0], [2.0]], [[-5.0], [-10.0]]]]),
        factors_dim=1, coordinates_dim=2)
    out.update(basic_array=basic.array, basic_sum_per_tensor=basic.sum_per_tensor(),
               basic_sq_per_tensor=basic.squared_fro_norm_per_tensor(), basic_var=basic.var_over_batch(),
               basic_std=basic.std_over_batch(), basic_mean=basic.mean_over_batch())
    npz("window_stats", **out)


def regulariser_and_init_fixture():
    """the reference's own `epses_composition.inner_product` (value and, through autograd, its
    gradients - the regulariser is differentiated every iteration) on seeded random stacks, and the cores its
    empirical-output-std initialisers return for a fixed seed and data set (the random core is drawn from torch's
    global generator: the same seed gives the same draw in the build's mirror)."""
    assert os.path.isdir(REF)
    _install_sequencers()
    sys.path.insert(0, REF)
    from dctn import eps as r_eps
    from dctn import epses_composition as r_comp

    f64 = torch.float64
    out = {}
    torch.manual_seed(zlib.crc32(b"inner_product"))
    stacks = {
        "s2": [(2,) * 4 + (3,), (3,) * 4 + (5,)],                   # two layers, K=2
        "s3": [(2,) * 9 + (4,), (4,) * 4 + (3,), (3,) * 4 + (2,)],  # three layers, first one K=3
        "s1": [(3,) * 4 + (6,)],                                    # one layer: plain dot product
    }
    for tag, shapes in stacks.items():
        e1 = [(torch.randn(*sh, dtype=f64) * 0.7).requires_grad_(True) for sh in shapes]
        e2 = [(torch.randn(*sh, dtype=f64) * 0.7).requires_grad_(True) for sh in shapes]
        val = r_comp.inner_product(e1, e2)
        g = torch.autograd.grad(val, e1 + e2)
        self_val = r_comp.inner_product(e1, e1)
        self_g = torch.autograd.grad(self_val, e1)
        out[f"{tag}_n"] = len(shapes)
        out[f"{tag}_value"], out[f"{tag}_self_value"] = val, self_val
        for i in range(len(shapes)):
            out[f"{tag}_a{i}"], out[f"{tag}_b{i}"] = e1[i], e2[i]
            out[f"{tag}_da{i}"], out[f"{tag}_db{i}"] = g[i], g[len(shapes) + i]
            out[f"{tag}_self_da{i}"] = self_g[i]
        out[f"{tag}_sqfro"] = r_comp.epswise_squared_fro_norm(e1)
    npz("inner_product", **out)

    out = {}
    x = phi(torch.rand(1, 20, 8, 8, dtype=f64, generator=torch.Generator().manual_seed(5)))   # (1, 20, 8, 8, 2)
    out["x"] = x
    seed, batch = 20261004, 7                                        # 20 samples in slices of 7, 7, 6
    torch.manual_seed(seed)
    one = r_eps.make_eps_unit_empirical_output_std(3, 4, x, torch.device("cpu"), f64, batch)
    torch.manual_seed(seed)
    raw = torch.randn(*(2,) * 9, 4, dtype=f64)                      # the draw the function starts from
    out.update(seed=seed, batch_size=batch, one_core=one, one_inverse_output_std=(one.reshape(-1)[0] / raw.reshape(-1)[0]))
    torch.manual_seed(seed)
    cores = r_comp.make_epses_composition_unit_empirical_output_std(((3, 3), (2, 4)), x, torch.device("cpu"), f64, batch)
    out.update(stack_core0=cores[0], stack_core1=cores[1])
    out["stack_output_std"] = r_eps.transform_in_slices(cores[1], r_eps.transform_in_slices(cores[0], x, batch), batch).std(unbiased=False)
    npz("empirical_std_init", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "window_stats":
        window_stats_fixture()
    elif len(sys.argv) > 1 and sys.argv[1] == "regulariser_init":
        regulariser_and_init_fixture()
    else:
        main()
        window_stats_fixture()
        regulariser_and_init_fixture()
