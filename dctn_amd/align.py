"""Shifted views of the input that line up the positions of a window.

Mirror of dctn/align.py:11-46.  The HIP kernels do this indexing themselves (the views below are
never materialised on the hot path); these functions exist for API parity and for host-side
helpers, and their integer indexing is pinned bit-exactly by tests/golden/align_*.npz.
"""
from __future__ import annotations

from typing import Iterator, Sequence, Tuple, Union

from torch import Tensor

from .pos2d import Pos2D


def align_with_positions(
    input: Union[Tensor, Sequence[Tensor]], positions: Tuple[Pos2D, ...]
) -> Iterator[Tensor]:
    """For every position (in the given order) and every channel (inner loop) yield the view
    ``input[ch][:, p.h : H-(max_h-p.h), p.w : W-(max_w-p.w)]`` of shape (B, H-max_h, W-max_w, Q)."""
    _, height, width, _ = input[0].shape
    hs = [p.h for p in positions]
    ws = [p.w for p in positions]
    assert min(hs) == 0
    assert min(ws) == 0
    out_h, out_w = height - max(hs), width - max(ws)
    for p in positions:
        for ch in range(len(input)):
            yield input[ch][:, p.h : p.h + out_h, p.w : p.w + out_w]


def align(input: Tensor, kernel_size: int) -> Iterator[Tensor]:
    """Positions in row-major order: 0 1 2 / 3 4 5 / 6 7 8 for ``kernel_size == 3``."""
    return align_with_positions(
        input, tuple(Pos2D(i // kernel_size, i % kernel_size) for i in range(kernel_size**2))
    )


def make_windows(x: Tensor, kernel_size: int):
    """``x``: (num_channels, batch, height, width, in_size).  The K*K*channels aligned views stacked
    on a new leading dim, as a batch of rank-one tensors (factors on dim 0, coordinates on dim 4):
    the reference's dctn/align.py:49-61.  This MATERIALISES K*K copies of ``x`` and exists for API
    parity and the tests; the statistics the reference reads off it come from
    ``dctn_amd.window_stats`` without any copy."""
    import torch

    from .rank_one_tensor import RankOneTensorsBatch

    stacked = torch.cat(
        tuple(torch.stack(tuple(align(part, kernel_size)), dim=0) for part in x.split(128, dim=1)), dim=1
    )
    return RankOneTensorsBatch(stacked, factors_dim=0, coordinates_dim=4)
