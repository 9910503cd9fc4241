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
    loss = loss_fn(output if getattr(loss_fn, "accepts_low_precision", False) else output.float(), y)
    reg_term = reg_fn(model) if reg_fn is not None else output.new_zeros((), dtype=torch.float32)
    optimizer.zero_grad(set_to_none=True)
    (loss + reg_term.float() * reg_coeff).backward()
    if reducer is not None:
        reducer()
    optimizer.step()
    return {"output": output.detach(), "loss": loss.detach(), "reg_term": reg_term.detach()}


class GraphedTrainStep:
    """The same iteration replayed from captured HIP graphs (the step of the small models is launch
    bound: ~40 kernels of a few microseconds each once loss, regulariser and optimizer are counted).

    Single process: one graph holds forward, loss, regulariser, backward and the optimizer step.
    Data parallel: forward + backward are one graph, the gradient all-reduce runs eagerly on the same
    stream (RCCL), the optimizer step is a second graph.  Inputs are copied into static buffers, so
    every call must use the batch shape of the example; the optimizer must be capturable
    (``torch.optim.SGD``, or ``Adam(..., capturable=True)``).
    """

    def __init__(self, model: torch.nn.Module, example_x: Tensor, example_y: Tensor,
                 loss_fn: Callable[[Tensor, Tensor], Tensor], optimizer: torch.optim.Optimizer,
                 reg_fn: Optional[Callable[[torch.nn.Module], Tensor]] = None, reg_coeff: float = 0.0,
                 reducer: Optional[ddp.FlatGradAllReducer] = None, warmup: int = 3):
        self.model, self.optimizer, self.reducer = model, optimizer, reducer
        self.x, self.y = example_x.clone(), example_y.clone()
        dev = example_x.device
        split = reducer is not None and reducer.world > 1

        def fwd_bwd():
            model.train()
            out = model(self.x)
            loss = loss_fn(out if getattr(loss_fn, "accepts_low_precision", False) else out.float(), self.y)
            reg = reg_fn(model) if reg_fn is not None else out.new_zeros((), dtype=torch.float32)
            optimizer.zero_grad(set_to_none=True)
            (loss + reg.float() * reg_coeff).backward()
            return out, loss, reg

        assert warmup >= 1, "capture needs at least one eager iteration first (lazy optimizer state, kernel attributes)"
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fwd_bwd()
                if split:
                    reducer()
                optimizer.step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.g_main = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_main, capture_error_mode="thread_local"):
            self.out, self.loss, self.reg = fwd_bwd()
            if not split:
                optimizer.step()
        self.g_opt = None
        if split:
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, capture_error_mode="thread_local"):
                optimizer.step()

    def __call__(self, x: Tensor, y: Tensor) -> Dict[str, Tensor]:
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        self.g_main.replay()
        if self.g_opt is not None:
            self.reducer()
            self.g_opt.replay()
        return {"output": self.out, "loss": self.loss, "reg_term": self.reg}


# ------------------------------------------------------------------------------- fused iteration tail
from . import _lib as L  # noqa: E402


class _FusedCrossEntropy(torch.autograd.Function):
    """``F.cross_entropy(logits, labels)`` (mean reduction) as one forward and one backward HIP kernel
    (`dctn_ce_loss_fwd/bwd`) instead of cast + log-softmax + nll and their three backward launches."""

    @staticmethod
    def forward(ctx, logits: Tensor, labels: Tensor) -> Tensor:
        dev = L.require_device(logits, labels)
        lg, lb = logits.contiguous(), labels.contiguous().long()
        assert lg.ndim == 2 and lb.shape == (lg.shape[0],)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        L.check(L.lib().dctn_ce_loss_fwd(lg.data_ptr(), lb.data_ptr(), loss.data_ptr(), lg.shape[0], lg.shape[1],
                                         L.dtype_code(lg), L.stream_ptr(dev)), "cross-entropy forward")
        ctx.save_for_backward(lg, lb)
        return loss

    @staticmethod
    def backward(ctx, d_loss: Tensor):
        lg, lb = ctx.saved_tensors
        dev = lg.device
        g = d_loss.to(torch.float32).contiguous()
        d_logits = torch.empty_like(lg)
        L.check(L.lib().dctn_ce_loss_bwd(lg.data_ptr(), lb.data_ptr(), g.data_ptr(), d_logits.data_ptr(), lg.shape[0],
                                         lg.shape[1], L.dtype_code(lg), L.stream_ptr(dev)), "cross-entropy backward")
        return d_logits, None


def fused_cross_entropy(logits: Tensor, labels: Tensor) -> Tensor:
    """Mean cross-entropy of (batch, classes) logits (float32 or bfloat16) as a float32 scalar."""
    return _FusedCrossEntropy.apply(logits, labels)


fused_cross_entropy.accepts_low_precision = True   # train_step / GraphedTrainStep skip their float32 cast


class FlatSGD:
    """SGD with momentum plus the reference's L2 regulariser, as ONE kernel per step over one flat
    parameter buffer (`dctn_sgd_l2_step`).

    ``regularised``: the parameters whose squared Frobenius norms the regulariser sums (for
    EPSesPlusLinear.epswise_l2_regularizer: every core and ``linear.weight``,
    dctn/eps_plus_linear.py:149-153); ``others``: the rest (``linear.bias``).  Adding
    ``l2 * sum ||w||^2`` to the loss and letting autograd differentiate it is the same update as adding
    ``2 * l2 * w`` to the gradient, which is what the kernel does; ``reg_value()`` returns the term's
    value (before the update) for logging.  The parameters are moved into one buffer (their ``.data``
    become views of it, in the order regularised + others); when the gradients already sit back to back
    in that order (the fused EPS + head backward allocates them so) the step reads them in place,
    otherwise they are gathered first.  Semantics of torch.optim.SGD(momentum, dampening = 0).
    """

    def __init__(self, regularised, others=(), lr: float = 1e-3, momentum: float = 0.0, l2: float = 0.0):
        self.reg_params = [p for p in regularised]
        self.params = self.reg_params + [p for p in others]
        assert self.params and len({p.dtype for p in self.params}) == 1 and len({p.device for p in self.params}) == 1
        self.lr, self.momentum, self.l2 = float(lr), float(momentum), float(l2)
        ref = self.params[0]
        self.n = sum(p.numel() for p in self.params)
        self.n_reg = sum(p.numel() for p in self.reg_params)
        self.flat = torch.empty(self.n, dtype=ref.dtype, device=ref.device)
        off = 0
        with torch.no_grad():
            for p in self.params:
                view = self.flat[off : off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
                off += p.numel()
        self.buf = torch.zeros(self.n, dtype=torch.float32, device=ref.device)
        self.sq_sum = torch.zeros((), dtype=torch.float32, device=ref.device)
        self.flat_grad = torch.zeros(self.n, dtype=ref.dtype, device=ref.device)
        self._steps = 0

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _grads(self) -> Tensor:
        grads = [p.grad for p in self.params]
        base, off, st = None, 0, None
        ok = all(g is not None and g.is_contiguous() for g in grads)
        if ok:
            g0 = grads[0]
            base, st = g0.data_ptr(), g0.untyped_storage()
            for g in grads:
                if g.untyped_storage().data_ptr() != st.data_ptr() or g.data_ptr() != base + off * g.element_size():
                    ok = False
                    break
                off += g.numel()
        if ok:
            return grads[0].new_empty(0).set_(st, grads[0].storage_offset(), (self.n,), (1,))
        torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(grads, self.params)],
                  out=self.flat_grad)
        return self.flat_grad

    @torch.no_grad()
    def step(self) -> None:
        g = self._grads()
        dev = self.flat.device
        L.check(L.lib().dctn_sgd_l2_step(self.flat.data_ptr(), g.data_ptr(), self.buf.data_ptr(), self.sq_sum.data_ptr(),
                                         self.n, self.n_reg, self.lr, self.momentum, self.l2,
                                         1 if self._steps == 0 else 0, L.dtype_code(self.flat), L.stream_ptr(dev)),
                "fused SGD step")
        self._steps += 1

    def reg_value(self) -> Tensor:
        """l2 * sum of squared Frobenius norms of the regularised parameters, as of the last step."""
        return self.sq_sum * self.l2
