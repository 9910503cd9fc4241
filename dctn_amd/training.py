"""One training iteration, data-parallel.

The reference's loop (dctn/training.py:62-84) is: forward, ``loss_fn(output, y)``, a regulariser
term scaled by ``reg_coeff``, ``optimizer.zero_grad()``, backward of the sum, ``optimizer.step()``,
with callback hooks around it.  Here the same sequence is one function; between backward and the
optimizer step the parameter gradients are averaged over the ranks (``ddp.FlatGradAllReducer``: in
place on the fused backward's flat gradient buffer when the model provides one).  The callback /
logging / checkpoint machinery of the reference is out of scope.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch
from torch import Tensor

from . import ddp


def train_step(
    model: torch.nn.Module,
    x: Tensor,
    y: Tensor,
    loss_fn: Callable[[Tensor, Tensor], Tensor],
    optimizer: torch.optim.Optimizer,
    reg_fn: Optional[Callable[[torch.nn.Module], Tensor]] = None,
    reg_coeff: float = 0.0,
    reducer: Optional[ddp.FlatGradAllReducer] = None,
) -> Dict[str, Tensor]:
    """``x``: this rank's shard (channels, batch, height, width, features); ``y``: its labels.
    Returns the detached ``output``, ``loss`` and ``reg_term`` of this rank."""
    model.train()
    output = model(x)
    loss = loss_fn(output.float(), y)
    reg_term = reg_fn(model) if reg_fn is not None else output.new_zeros((), dtype=torch.float32)
    optimizer.zero_grad(set_to_none=True)
    (loss + reg_term.float() * reg_coeff).backward()
    if reducer is not None:
        reducer()
    optimizer.step()
    return {"output": output.detach(), "loss": loss.detach(), "reg_term": reg_term.detach()}
