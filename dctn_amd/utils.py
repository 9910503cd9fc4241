"""Initialisation tags and small helpers shared by the models (reference: dctn/utils.py:10-59)."""
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
    return (not x) or y


def xor(*args: bool) -> bool:
    return sum(bool(a) for a in args) % 2 == 1


def exactly_one_true(*args: bool) -> bool:
    assert all(isinstance(a, bool) for a in args)
    return sum(args) == 1


def raise_exception(exception: BaseException):
    raise exception


def id_assert_shape_matches(tensor: Tensor, shape: Sequence[int]) -> Tensor:
    assert tuple(tensor.shape) == tuple(shape)
    return tensor
