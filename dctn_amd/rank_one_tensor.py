"""Batches of rank-one tensors kept as their tensor-product factors.

Mirror of the reference's dctn/rank_one_tensor.py:14-110 (same class, attributes, properties and
methods): for every index combination of the batch dims the slice of ``array`` is a 2-D array whose
fibres along ``coordinates_dim`` are the factors (one per index of ``factors_dim``) of the rank-one
tensor  T = factor_0 (x) factor_1 (x) ...  .  All statistics follow from two identities,
``sum(T) = prod_f sum(factor_f)`` and ``||T||^2 = prod_f ||factor_f||^2``; nothing of size
``coordinates ** factors`` is ever formed.

On the device the window statistics the reference computes through this class
(``calc_scaling_factor``) run as one HIP kernel: see dctn_amd/window_stats.py.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Tuple

import torch
from torch import Tensor


@dataclass(frozen=True)
class RankOneTensorsBatch:
    array: Tensor
    factors_dim: int
    coordinates_dim: int

    def __post_init__(self) -> None:
        assert self.factors_dim != self.coordinates_dim
        assert 0 <= self.factors_dim < self.array.ndim and 0 <= self.coordinates_dim < self.array.ndim

    @property
    def batch_shape(self) -> Tuple[int, ...]:
        return tuple(n for i, n in enumerate(self.array.shape) if i not in (self.factors_dim, self.coordinates_dim))

    @property
    def ncoordinates(self) -> int:
        """Elements of ONE tensor of the batch."""
        return self.array.shape[self.coordinates_dim] ** self.array.shape[self.factors_dim]

    @property
    def ntensors(self) -> int:
        return math.prod(self.batch_shape)

    def _per_tensor(self, per_factor: Tensor) -> Tensor:
        """prod over the factors of a per-factor scalar (``per_factor`` keeps both dims with size 1 /
        size factors), squeezed to the batch shape."""
        out = per_factor.prod(dim=self.factors_dim, keepdim=True)
        hi, lo = max(self.factors_dim, self.coordinates_dim), min(self.factors_dim, self.coordinates_dim)
        return out.squeeze(hi).squeeze(lo)

    def sum_per_tensor(self) -> Tensor:
        return self._per_tensor(self.array.sum(dim=self.coordinates_dim, keepdim=True))

    def sum_over_batch(self) -> Tensor:
        return self.sum_per_tensor().sum()

    def mean_per_tensor(self) -> Tensor:
        return self.sum_per_tensor() / self.ncoordinates

    def mean_over_batch(self) -> Tensor:
        return self.sum_over_batch() / (self.ntensors * self.ncoordinates)

    def squared_fro_norm_per_tensor(self) -> Tensor:
        return self._per_tensor((self.array**2).sum(dim=self.coordinates_dim, keepdim=True))

    def squared_fro_norm_over_batch(self) -> Tensor:
        return self.squared_fro_norm_per_tensor().sum()

    def var_over_batch(self, unbiased: bool = True) -> Tensor:
        """Empirical variance over every element of every tensor (Bessel's correction iff
        ``unbiased``): (sum x^2 - 2 mean sum x + n mean^2) / divisor."""
        total, mean = self.sum_over_batch(), self.mean_over_batch()
        n = self.ntensors * self.ncoordinates
        divisor = n - 1 if unbiased else n
        return self.squared_fro_norm_over_batch() / divisor - 2 * total / divisor * mean + n / divisor * mean**2

    def std_over_batch(self, unbiased: bool = True) -> Tensor:
        """As in the reference (rank_one_tensor.py:106-109) the variance is taken with its default
        Bessel correction whatever ``unbiased`` says."""
        return self.var_over_batch() ** 0.5
