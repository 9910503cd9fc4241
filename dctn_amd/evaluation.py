"""Scoring a model on a data loader (mirror of the reference's dctn/evaluation.py:7-22), with the two
sums reduced over the data-parallel ranks when a process group is up: every rank scores its own
shard and all ranks return the same global numbers."""
from __future__ import annotations

from typing import Iterable, Tuple

import torch
import torch.nn.functional as F

from . import ddp


def score(model, dl: Iterable, device) -> Tuple[float, float]:
    """Mean cross-entropy and accuracy over all batches of ``dl`` (items ``(x, y, indices)`` with ``x``
    in the (channels, batch, height, width, features) layout).  ``model(x)`` returns unnormalised
    log-probabilities (batch, classes)."""
    num_samples = torch.zeros((), dtype=torch.float64, device=device)
    num_correct = torch.zeros((), dtype=torch.float64, device=device)
    sum_loss = torch.zeros((), dtype=torch.float64, device=device)
    with torch.no_grad():
        for x, y, _ in iter(dl):
            y = y.to(device)
            out = model(x.to(device)).float()
            num_samples += len(y)
            sum_loss += F.cross_entropy(out, y, reduction="sum").double()
            num_correct += (out.argmax(dim=1) == y).sum().double()
    sum_loss, num_correct, num_samples = ddp.all_reduce_scalar_sums(sum_loss, num_correct, num_samples)
    return float(sum_loss / num_samples), float(num_correct / num_samples)
