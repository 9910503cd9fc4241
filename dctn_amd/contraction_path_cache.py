"""Process-wide cache of pairwise contraction plans for general einsum expressions.

Mirror of dctn/contraction_path_cache.py:15-35 (``ContractionPathCache`` singleton with a
``paths`` dict keyed on shapes + subscripts, and the module-level ``contract``).  The reference
delegates plan search and execution to the third-party ``opt_einsum``; here the plan is found by
a greedy smallest-intermediate search of our own and executed as pairwise ``torch.einsum`` calls
(rocBLAS on device).  This serves the parameter-only contractions (TT statistics, ``as_eps``,
the composition inner product); the per-window hot path does NOT go through here — it is the
HIP kernels behind ``eps`` / ``ConvSBS.forward``.

Accepted formats (all three give bit-identical results because they share one canonical plan):
    contract("ij,jk->ik", a, b)
    contract(a, "ij", b, "jk", "ik")          # interleaved, any hashable names
    contract(a, (0, 1), b, (1, 2), (0, 2))
"""
from __future__ import annotations

import math
from typing import Dict, Hashable, List, Sequence, Tuple, Union

import torch
from torch import Tensor

from .singleton import Singleton

ContractArgs = Tuple[Union[Hashable, Tensor], ...]
ContractExpressionArgs = Tuple[Hashable, ...]


def tensors_to_shapes(*args) -> ContractExpressionArgs:
    return tuple(tuple(x.shape) if isinstance(x, Tensor) else x for x in args)


def _split(args) -> Tuple[List, List[Tuple], Tuple]:
    """-> (operands-or-shapes, per-operand index names, output index names)."""
    if isinstance(args[0], str):
        lhs, rhs = args[0].replace(" ", "").split("->")
        return list(args[1:]), [tuple(s) for s in lhs.split(",")], tuple(rhs)
    ops, subs = list(args[0:-1:2]), [tuple(s) for s in args[1:-1:2]]
    assert len(args) % 2 == 1, "interleaved format needs an explicit output subscript"
    return ops, subs, tuple(args[-1])


class ContractExpression:
    """A frozen pairwise contraction plan for fixed shapes."""

    def __init__(self, shapes: Sequence[Tuple[int, ...]], subs: Sequence[Tuple[int, ...]], out: Tuple[int, ...]):
        self.subs, self.out = [tuple(s) for s in subs], tuple(out)
        size: Dict[int, int] = {}
        for shape, sub in zip(shapes, subs):
            assert len(shape) == len(sub)
            for n, d in zip(shape, sub):
                assert size.setdefault(d, n) == n
        self.steps: List[Tuple[int, int, Tuple[int, ...]]] = []
        live = list(self.subs)
        while len(live) > 1:
            best = None
            for i in range(len(live)):
                for j in range(i + 1, len(live)):
                    others = set(self.out)
                    for k, s in enumerate(live):
                        if k != i and k != j:
                            others.update(s)
                    merged = tuple(dict.fromkeys(live[i] + live[j]))
                    keep = tuple(d for d in merged if d in others)
                    shared = bool(set(live[i]) & set(live[j]))
                    cost = (not shared, math.prod(size[d] for d in keep), math.prod(size[d] for d in merged))
                    if best is None or cost < best[0]:
                        best = (cost, i, j, keep)
            _, i, j, keep = best
            self.steps.append((i, j, keep))
            live = [s for k, s in enumerate(live) if k not in (i, j)] + [keep]

    @staticmethod
    def _pair(a: Tensor, sa, b: Tensor, sb, so) -> Tensor:
        local = {d: n for n, d in enumerate(dict.fromkeys(tuple(sa) + tuple(sb)))}
        return torch.einsum(a, [local[d] for d in sa], b, [local[d] for d in sb], [local[d] for d in so])

    def __call__(self, *tensors: Tensor) -> Tensor:
        ops, subs = list(tensors), list(self.subs)
        for i, j, keep in self.steps:
            t = self._pair(ops[i], subs[i], ops[j], subs[j], keep)
            ops = [o for k, o in enumerate(ops) if k not in (i, j)] + [t]
            subs = [s for k, s in enumerate(subs) if k not in (i, j)] + [keep]
        (res,), (sub,) = ops, subs
        if sub != self.out:  # single operand, or a final permutation / trace
            local = {d: n for n, d in enumerate(dict.fromkeys(sub))}
            res = torch.einsum(res, [local[d] for d in sub], [local[d] for d in self.out])
        return res


class ContractionPathCache(metaclass=Singleton):
    def __init__(self):
        self.paths: Dict[ContractExpressionArgs, ContractExpression] = {}

    def contract(self, *args) -> Tensor:
        ops, subs, out = _split(args)
        names: Dict[Hashable, int] = {}
        for s in subs + [out]:
            for d in s:
                names.setdefault(d, len(names))
        isubs = tuple(tuple(names[d] for d in s) for s in subs)
        iout = tuple(names[d] for d in out)
        key = (tuple(tuple(o.shape) for o in ops), isubs, iout)
        expr = self.paths.get(key)
        if expr is None:
            expr = self.paths[key] = ContractExpression(key[0], isubs, iout)
        return expr(*ops)


def contract(*args) -> Tensor:
    return ContractionPathCache().contract(*args)
