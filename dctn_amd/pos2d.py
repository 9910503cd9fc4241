"""2-D positions inside a window and their row-major enumeration.

Mirror of the reference's dctn/pos2d.py:4-23 (same names, argument order and integer results).
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Pos2D:
    h: int
    w: int


def pos_to_index(max_w: int, pos: Pos2D) -> int:
    """Index of ``pos`` when positions are listed row by row with ``w`` in ``0..max_w``."""
    assert pos.w <= max_w
    return (max_w + 1) * pos.h + pos.w


def index_to_pos(max_w: int, index: int) -> Pos2D:
    """Inverse of ``pos_to_index(max_w, .)``."""
    h, w = divmod(index, max_w + 1)
    return Pos2D(h, w)
