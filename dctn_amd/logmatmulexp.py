"""log(exp(A) @ exp(B)) in log space, on the MI355X.

Mirror of dctn/logmatmulexp.py:5-22: ``logmatmulexp(log_A, log_B, /)`` (strictly 2-D, asserts the
inner sizes match) and ``logmatmulexp_lowmem`` (the checkpointed variant of the reference; here
both are the same kernels, which never materialise the (Theta, R, I) tensor).  Additions of this
build: ``logmatmulexp_batched`` and ``logmatmulexp_fold`` (BASELINE config 5: the left fold of
small_experiments/logmatmulexp_benchmark/benchmark.py:30, one window per workgroup).
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib as L


def _workspace(dev, nb, T, R, I, sA, sB, code):
    """Scratch for the factored (exp -> MFMA GEMM -> log) kernels; (None, 0) where the direct kernels run."""
    nbytes = L.lib().dctn_logmatmulexp_workspace_bytes(nb, T, R, I, sA, sB, code)
    if nbytes == 0:
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes


class _LME(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_A: Tensor, log_B: Tensor) -> Tensor:
        dev = L.require_device(log_A, log_B)
        assert log_A.dtype == log_B.dtype
        batch, T, R = log_A.shape
        assert log_B.shape[1] == R
        I = log_B.shape[2]
        nb = max(batch, log_B.shape[0])
        a, b = log_A.contiguous(), log_B.contiguous()
        sA = 0 if (a.shape[0] == 1 and nb > 1) else T * R
        sB = 0 if (b.shape[0] == 1 and nb > 1) else R * I
        out = torch.empty((nb, T, I), dtype=a.dtype, device=dev)
        ws, nbytes = _workspace(dev, nb, T, R, I, sA, sB, L.dtype_code(a))
        L.check(
            L.lib().dctn_logmatmulexp_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(),
                                          None if ws is None else ws.data_ptr(), nbytes, nb, T, R, I, sA, sB,
                                          L.dtype_code(a), L.stream_ptr(dev)),
            "logmatmulexp forward",
        )
        ctx.save_for_backward(a, b, out)
        ctx.dims = (nb, T, R, I, sA, sB)
        return out

    @staticmethod
    def backward(ctx, d_out: Tensor):
        a, b, out = ctx.saved_tensors
        nb, T, R, I, sA, sB = ctx.dims
        dev = a.device
        g = d_out.contiguous()
        dA = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        dB = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        ws, nbytes = _workspace(dev, nb, T, R, I, sA, sB, L.dtype_code(a))
        L.check(
            L.lib().dctn_logmatmulexp_bwd(
                a.data_ptr(), b.data_ptr(), out.data_ptr(), g.data_ptr(),
                None if dA is None else dA.data_ptr(), None if dB is None else dB.data_ptr(),
                None if ws is None else ws.data_ptr(), nbytes, nb, T, R, I, sA, sB, L.dtype_code(a), L.stream_ptr(dev)),
            "logmatmulexp backward",
        )
        return dA, dB


def logmatmulexp(log_A: Tensor, log_B: Tensor, /) -> Tensor:
    """Given log_A (Theta x R) and log_B (R x I) returns (log_A.exp() @ log_B.exp()).log(),
    numerically stable, forward and backward."""
    Theta, R = log_A.shape
    I = log_B.shape[1]
    assert log_B.shape == (R, I)
    return L.on_device(_LME.apply, log_A.unsqueeze(0), log_B.unsqueeze(0)).squeeze(0)


def logmatmulexp_lowmem(log_A: Tensor, log_B: Tensor, /) -> Tensor:
    """Same result; the reference needs activation checkpointing to avoid saving a
    (Theta, R, I) tensor — the kernels here never create one, so this is ``logmatmulexp``."""
    return logmatmulexp(log_A, log_B)


def logmatmulexp_batched(log_A: Tensor, log_B: Tensor, /) -> Tensor:
    """(batch, Theta, R) x (batch, R, I) -> (batch, Theta, I); a batch size of 1 on either side
    broadcasts (its gradient is summed over the batch)."""
    assert log_A.ndim == 3 and log_B.ndim == 3 and log_A.shape[2] == log_B.shape[1]
    assert log_A.shape[0] == log_B.shape[0] or 1 in (log_A.shape[0], log_B.shape[0])
    return L.on_device(_LME.apply, log_A, log_B)


class _Fold(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mats: Tensor) -> Tensor:
        dev = L.require_device(mats)
        Wn, Ln, D, D2 = mats.shape
        assert D == D2
        m = mats.contiguous()
        out = torch.empty((Wn, D, D), dtype=m.dtype, device=dev)
        L.check(
            L.lib().dctn_logmatmulexp_fold_fwd(m.data_ptr(), out.data_ptr(), Wn, Ln, D, L.dtype_code(m),
                                               L.stream_ptr(dev)),
            "logmatmulexp fold forward",
        )
        ctx.save_for_backward(m)
        return out

    @staticmethod
    def backward(ctx, d_out: Tensor):
        (m,) = ctx.saved_tensors
        Wn, Ln, D, _ = m.shape
        dev = m.device
        g = d_out.contiguous()
        dm = torch.empty_like(m)
        nbytes = L.lib().dctn_logmatmulexp_fold_workspace_bytes(Wn, Ln, D, L.dtype_code(m), 1)
        ws = L.workspace(nbytes, dev)
        L.check(
            L.lib().dctn_logmatmulexp_fold_bwd(m.data_ptr(), g.data_ptr(), dm.data_ptr(), ws.data_ptr(),
                                               ws.numel(), Wn, Ln, D, L.dtype_code(m), L.stream_ptr(dev)),
            "logmatmulexp fold backward",
        )
        return dm


def logmatmulexp_fold(mats: Tensor, /) -> Tensor:
    """mats (windows, L, D, D): per window ``reduce(logmatmulexp, mats[w])`` -> (windows, D, D)."""
    assert mats.ndim == 4 and mats.shape[2] == mats.shape[3]
    return L.on_device(_Fold.apply, mats)
