"""Initialisation tags and the small helpers the models and the runner import (reference: dctn/utils.py:10-59;
new_runner.py:51-59 does ``from dctn.utils import implies, xor, exactly_one_true, ...``)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Sequence, Union

import torch
from torch import Tensor


@dataclass(frozen=True)
class ZeroCenteredNormalInitialization:
    std: float


@dataclass(frozen=True)
class ZeroCenteredUniformInitialization:
    maximum: float


@dataclass(frozen=True)
class FromFileInitialization:
    path: str


OneTensorInitialization = Union[
    ZeroCenteredNormalInitialization, ZeroCenteredUniformInitialization, FromFileInitialization
]


@torch.no_grad()
def transform_dataset(f: Callable[[Tensor], Tensor], x: Tensor, batch_size: int = 64) -> Tensor:
    """Applies an ``eps``-like ``f`` to ``x`` (channel, sample, height, width, quantum) slice by
    slice along the sample dim; returns (1, sample, height', width', quantum')."""
    return torch.cat([f(part) for part in x.split(batch_size, dim=1)]).unsqueeze(0)


def implies(x: bool, y: bool) -> bool:
    """Material implication (dctn/utils.py:20-21)."""
    return (not x) or y


def xor(*args: bool) -> bool:
    """True when an odd number of the arguments is true; ``xor()`` is False (dctn/utils.py:24-25)."""
    result = False
    for arg in args:
        result = result != bool(arg)
    return result


def exactly_one_true(*args: bool) -> bool:
    """dctn/utils.py:28-30: the arguments must be genuine bools (``AssertionError`` otherwise)."""
    assert all(isinstance(arg, bool) for arg in args)
    return sum(args) == 1


def raise_exception(exception):
    """``raise`` as an expression (dctn/utils.py:50-51)."""
    raise exception


def id_assert_shape_matches(tensor: Tensor, shape: Sequence[int]) -> Tensor:
    assert tuple(tensor.shape) == tuple(shape)
    return tensor
