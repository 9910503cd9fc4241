"""Specification of one "snake" string of a ConvSBS layer: core positions, bond sizes and the
derived core shapes / einsum dimension names.

Mirror of the reference's dctn/conv_sbs_spec.py:10-158 (same class names, constructor arguments,
properties and ValueError behaviour).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Tuple

from .pos2d import Pos2D, pos_to_index


@dataclass(frozen=True)
class SBSSpecCore:
    position: Pos2D
    out_quantum_dim_size: int


@dataclass(frozen=True)
class SBSCoreShape:
    out_quantum_dim_size: int
    bond_left_size: int
    bond_right_size: int
    in_num_channels: int
    in_quantum_dim_size: int

    def as_tuple(self) -> Tuple[int, ...]:
        """(out, bond_left, bond_right, q, ..., q) with one q per input channel."""
        head = (self.out_quantum_dim_size, self.bond_left_size, self.bond_right_size)
        return head + (self.in_quantum_dim_size,) * self.in_num_channels

    @property
    def dimensions_names(self) -> Tuple[str, ...]:
        names = ["out_quantum", "bond_left", "bond_right"]
        names += [f"in_quantum_{c}" for c in range(self.in_num_channels)]
        return tuple(names)

    @property
    def total_dangling_dimensions_size(self) -> int:
        return self.out_quantum_dim_size * self.in_quantum_dim_size**self.in_num_channels


@dataclass(frozen=True)
class SBSSpecString:
    cores: Tuple[SBSSpecCore, ...]
    bond_sizes: Tuple[int, ...]
    in_num_channels: int
    in_quantum_dim_size: int = 2

    def __post_init__(self):
        if min(c.position.h for c in self.cores) != 0 or min(c.position.w for c in self.cores) != 0:
            raise ValueError("Positions of cores are invalid")
        if len(self.bond_sizes) != len(self.cores):
            raise ValueError(
                f"len(bond_sizes)={len(self.bond_sizes)}, it must be equal to len(cores)={len(self.cores)}"
            )

    def __len__(self) -> int:
        return len(self.cores)

    @property
    def shapes(self) -> Tuple[SBSCoreShape, ...]:
        """Core i sits between bond i (left) and bond i+1 (right); the last core closes the
        ring on bond 0."""
        n = len(self.cores)
        return tuple(
            SBSCoreShape(
                core.out_quantum_dim_size,
                self.bond_sizes[i],
                self.bond_sizes[(i + 1) % n],
                self.in_num_channels,
                self.in_quantum_dim_size,
            )
            for i, core in enumerate(self.cores)
        )

    @property
    def positions(self) -> Tuple[Pos2D, ...]:
        return tuple(core.position for core in self.cores)

    @property
    def max_height_pos(self) -> int:
        return max(core.position.h for core in self.cores)

    @property
    def max_width_pos(self) -> int:
        return max(core.position.w for core in self.cores)

    def get_indices_wrt_standard_order(self) -> Tuple[int, ...]:
        """For a string that fills a rectangle: row-major index of every core's position."""
        assert len(self) == (self.max_width_pos + 1) * (self.max_height_pos + 1)
        return tuple(pos_to_index(self.max_width_pos, p) for p in self.positions)

    @property
    def out_total_quantum_dim_size(self) -> int:
        return math.prod(core.out_quantum_dim_size for core in self.cores)

    @property
    def nelement(self) -> int:
        """Number of elements of the tensor the string represents."""
        return math.prod(shape.total_dangling_dimensions_size for shape in self.shapes)

    def get_dim_names(self, core_index: int, /) -> Tuple[str, ...]:
        """einsum names of the dims of core ``core_index``; only bond names are shared."""
        right = core_index + 1 if core_index < len(self) - 1 else 0
        return (
            f"out_quantum_{core_index}",
            f"bond_{core_index}",
            f"bond_{right}",
        ) + tuple(f"in_quantum_{c}_{core_index}" for c in range(self.in_num_channels))

    @property
    def all_dim_names(self) -> Tuple[Tuple[str, ...], ...]:
        return tuple(self.get_dim_names(i) for i in range(len(self)))

    def get_all_dim_names_add_suffix_to_bonds(self, suffix: str, /) -> Tuple[Tuple[str, ...], ...]:
        return tuple(
            tuple(n + suffix if n.startswith("bond_") else n for n in names)
            for names in self.all_dim_names
        )

    @property
    def all_dangling_dim_names(self) -> Tuple[str, ...]:
        """All input dims (core-major, channel-minor) followed by all output dims."""
        ins = [n for names in self.all_dim_names for n in names[3:]]
        outs = [names[0] for names in self.all_dim_names]
        return tuple(ins + outs)
