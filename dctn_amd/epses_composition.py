"""A stack of EPS cores applied one after another, its tensor-network inner product and its
initialisers.  Mirror of dctn/epses_composition.py:21-146."""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch
from torch import Tensor

from . import eps, tn_inner
from .contraction_path_cache import contract
from .utils import (
    FromFileInitialization,
    OneTensorInitialization,
    ZeroCenteredNormalInitialization,
    ZeroCenteredUniformInitialization,
    id_assert_shape_matches,
)


def contract_with_input(epses: Sequence[Tensor], input: Tensor) -> Tensor:
    """``input``: (channels, batch, height, width, in_size).  Every EPS but the last feeds the
    next one as a single-channel input.  Returns (batch, height', width', out_size of the last)."""
    assert all(eps.is_eps(core) for core in epses)
    x = input
    for core in epses[:-1]:
        x = eps.eps(core, x).unsqueeze(0)
    return eps.eps(epses[-1], x)


def inner_product(epses1: Sequence[Tensor], epses2: Sequence[Tensor]) -> Tensor:
    """Inner product of the two linear maps the stacks represent.  The first pair is contracted
    over its inputs into an (out x out) matrix, which is then absorbed into every input leg of
    the next core of stack 1; recurse on the shortened stacks.  On the device: `dctn_fiber_gram` for the Gram
    matrix and the closing dot product, `dctn_mode_product` per absorbed leg (dctn_amd/tn_inner.py), forward and
    backward."""
    epses1, epses2 = tuple(epses1), tuple(epses2)
    assert len(epses1) == len(epses2)
    for a, b in zip(epses1, epses2):
        assert a.shape == b.shape
        assert eps.is_eps(a)
    if len(epses1) == 1:
        return eps.inner_product(epses1[0], epses2[0])
    gram = eps.contract_on_input_dims(epses1[0], epses2[0])  # (out of stack1[0], out of stack2[0])
    nxt = epses1[1]
    if gram.dtype == nxt.dtype and tn_inner.covers(*gram.shape):
        absorbed = tn_inner.absorb_into_input_legs(nxt, gram)   # N mode products on the device
    else:   # sizes beyond the kernels' 32: the planner + library einsums
        n_in = nxt.ndim - 1
        args = [nxt, tuple(f"in{i}" for i in range(n_in)) + ("out",)]
        for i in range(n_in):
            args += [gram, (f"in{i}", f"newin{i}")]
        args.append(tuple(f"newin{i}" for i in range(n_in)) + ("out",))
        absorbed = contract(*args)
    assert eps.is_eps(absorbed)
    return inner_product((absorbed,) + epses1[2:], epses2[1:])


def epswise_squared_fro_norm(epses: Sequence[Tensor]) -> Tensor:
    """Sum of the squared Frobenius norms of the cores: one `dctn_fiber_gram` (a dot product) per core."""
    assert all(eps.is_eps(core) for core in epses)
    return sum(tn_inner.dot(core, core) for core in epses)


def specs_to_full_specs(epses_specs: Tuple[Tuple[int, int], ...], initial_in_size: int) -> Tuple[Dict[str, int], ...]:
    """Each spec is (kernel_size, out_size); in_size chains from ``initial_in_size``."""
    full, in_size = [], initial_in_size
    for kernel_size, out_size in epses_specs:
        full.append(dict(kernel_size=kernel_size, in_num_channels=1, in_size=in_size, out_size=out_size))
        in_size = out_size
    return tuple(full)


def make_epses_composition_unit_theoretical_output_std(
    epses_specs: Tuple[Tuple[int, int], ...], initial_in_size: int, device: torch.device, dtype: torch.dtype
) -> Tuple[Tensor, ...]:
    return tuple(
        eps.make_eps_unit_theoretical_output_std(**spec, device=device, dtype=dtype)
        for spec in specs_to_full_specs(epses_specs, initial_in_size)
    )


def make_epses_composition_unit_empirical_output_std(
    epses_specs: Tuple[Tuple[int, int], ...], input: Tensor, device: torch.device, dtype: torch.dtype,
    batch_size: int = 128,
) -> Tuple[Tensor, ...]:
    cores = []
    for kernel_size, out_size in epses_specs:
        core = eps.make_eps_unit_empirical_output_std(kernel_size, out_size, input, device, dtype, batch_size)
        input = eps.transform_in_slices(core, input.to(device, dtype), batch_size)
        cores.append(core)
    return tuple(cores)


def make_epses_composition_manually_chosen_inializations(
    epses_specs: Tuple[Tuple[int, int], ...], initializations: Tuple[OneTensorInitialization, ...],
    initial_in_size: int, device: torch.device, dtype: torch.dtype,
) -> Tuple[Tensor, ...]:
    assert len(epses_specs) == len(initializations)
    cores = []
    for spec, init in zip(specs_to_full_specs(epses_specs, initial_in_size), initializations):
        shape = eps.spec_to_shape(**spec)
        if isinstance(init, ZeroCenteredNormalInitialization):
            cores.append(torch.randn(shape, dtype=dtype).to(device) * init.std)
        elif isinstance(init, ZeroCenteredUniformInitialization):
            cores.append(torch.rand(shape, dtype=dtype).to(device) * (2 * init.maximum) - init.maximum)
        elif isinstance(init, FromFileInitialization):
            cores.append(id_assert_shape_matches(torch.load(init.path, device).to(dtype=dtype), shape))
        else:
            raise ValueError()
    return tuple(cores)
