"""The two device primitives behind the tensor-network inner product of EPS stacks (the regulariser the reference
evaluates every iteration: dctn/epses_composition.py:21-58, dctn/eps.py:106-123), with their autograd:

    mode_product(x, M, pre, q, post)   out[pre, j, post] = sum_i x[pre, i, post] M[i, j]
    fiber_gram(a, b, pre, qa, qb, post) out[i, j] = sum_(pre, post) a[pre, i, post] b[pre, j, post]

Kernels: dctn_amd/csrc/tn_inner.hip through `dctn_mode_product` / `dctn_fiber_gram`.  The backward of either is made
of the same two calls (no saved intermediates beyond the operands).  CPU tensors are staged like everywhere else.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor

from . import _lib as L

MAX_Q = 32


def _mode_product(x: Tensor, M: Tensor, pre: int, q: int, q2: int, post: int) -> Tensor:
    dev = L.require_device(x, M)
    out = torch.empty(pre * q2 * post, dtype=x.dtype, device=dev)
    L.check(L.lib().dctn_mode_product(x.data_ptr(), M.data_ptr(), out.data_ptr(), pre, q, q2, post, L.dtype_code(x),
                                      L.stream_ptr(dev)), "mode product")
    return out


def _fiber_gram(a: Tensor, b: Tensor, pre: int, qa: int, qb: int, post: int) -> Tensor:
    dev = L.require_device(a, b)
    code = L.dtype_code(a)
    out = torch.empty((qa, qb), dtype=a.dtype, device=dev)
    ws = L.workspace(L.lib().dctn_fiber_gram_workspace_bytes(pre, qa, qb, post, code), dev)
    L.check(L.lib().dctn_fiber_gram(a.data_ptr(), b.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), pre, qa, qb,
                                    post, code, L.stream_ptr(dev)), "fiber gram")
    return out


class _ModeProduct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, M: Tensor, pre: int, post: int) -> Tensor:
        q, q2 = M.shape
        assert x.numel() == pre * q * post and x.dtype == M.dtype
        xc, Mc = x.contiguous().reshape(-1), M.contiguous()
        ctx.save_for_backward(xc, Mc)
        ctx.dims = (pre, q, q2, post)
        ctx.x_shape = x.shape
        return _mode_product(xc, Mc, pre, q, q2, post)

    @staticmethod
    def backward(ctx, d_out: Tensor):
        xc, Mc = ctx.saved_tensors
        pre, q, q2, post = ctx.dims
        g = d_out.contiguous().reshape(-1)
        d_x = (_mode_product(g, Mc.t().contiguous(), pre, q2, q, post).reshape(ctx.x_shape)
               if ctx.needs_input_grad[0] else None)
        d_M = _fiber_gram(xc, g, pre, q, q2, post) if ctx.needs_input_grad[1] else None
        return d_x, d_M, None, None


class _FiberGram(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor, pre: int, qa: int, qb: int, post: int) -> Tensor:
        assert a.numel() == pre * qa * post and b.numel() == pre * qb * post and a.dtype == b.dtype
        ac, bc = a.contiguous().reshape(-1), b.contiguous().reshape(-1)
        ctx.save_for_backward(ac, bc)
        ctx.dims = (pre, qa, qb, post)
        ctx.shapes = (a.shape, b.shape)
        return _fiber_gram(ac, bc, pre, qa, qb, post)

    @staticmethod
    def backward(ctx, d_out: Tensor):
        ac, bc = ctx.saved_tensors
        pre, qa, qb, post = ctx.dims
        g = d_out.contiguous()
        # d a[pre, i, post] = sum_j b[pre, j, post] g[i, j]  ->  mode product of b with g^T (qb x qa)
        d_a = (_mode_product(bc, g.t().contiguous(), pre, qb, qa, post).reshape(ctx.shapes[0])
               if ctx.needs_input_grad[0] else None)
        d_b = _mode_product(ac, g, pre, qa, qb, post).reshape(ctx.shapes[1]) if ctx.needs_input_grad[1] else None
        return d_a, d_b, None, None, None, None


def covers(*sizes: int) -> bool:
    """Whether the kernels take these leg / output sizes (the reference's models: 2 ... 24)."""
    return all(1 <= s <= MAX_Q for s in sizes)


def gram_over_input_dims(a: Tensor, b: Tensor) -> Tensor:
    """(out of a, out of b): both cores contracted over all their input legs (dctn/eps.py:106-112)."""
    rows = math.prod(a.shape[:-1])
    assert math.prod(b.shape[:-1]) == rows
    return L.on_device(lambda a_, b_: _FiberGram.apply(a_, b_, rows, a_.shape[-1], b_.shape[-1], 1), a, b)


def dot(a: Tensor, b: Tensor) -> Tensor:
    """sum(a * b) as a 0-dim tensor (dctn/eps.py:120-123)."""
    assert a.shape == b.shape
    return L.on_device(lambda a_, b_: _FiberGram.apply(a_, b_, a_.numel(), 1, 1, 1).reshape(()), a, b)


def absorb_into_input_legs(core: Tensor, gram: Tensor) -> Tensor:
    """new[j_0..j_{N-1}, o] = sum_i core[i_0..i_{N-1}, o] prod_n gram[i_n, j_n]: the N-operand einsum of
    dctn/epses_composition.py:46-56, one leg (one `dctn_mode_product`) at a time."""
    n_in, q, q2 = core.ndim - 1, gram.shape[0], gram.shape[1]
    assert all(s == q for s in core.shape[:-1])
    out_size = core.shape[-1]

    def run(core_: Tensor, gram_: Tensor) -> Tensor:
        x = core_
        for n in range(n_in):   # legs before n already have size q2, legs after it still q
            x = _ModeProduct.apply(x, gram_, q2**n, q ** (n_in - 1 - n) * out_size)
        return x.reshape((q2,) * n_in + (out_size,))

    return L.on_device(run, core, gram)
