"""Initialisation tags and the two helpers the models on the path use (reference: dctn/utils.py:10-17,20-36,54-59)."""
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


def id_assert_shape_matches(tensor: Tensor, shape: Sequence[int]) -> Tensor:
    assert tuple(tensor.shape) == tuple(shape)
    return tensor
