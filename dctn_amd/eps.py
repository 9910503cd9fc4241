"""EPS ("entangled plaquette state") layer: one dense core of order K*K*C + 1 contracted with
every K x K window of the input.

Mirror of the reference's dctn/eps.py:19-187 (same function / class names, argument order and
error behaviour).  ``eps`` and ``eps_one_by_one`` are the hot path: they run as hand-written HIP
kernels (dctn_amd/csrc/eps_*.hip) through the C-ABI of include/dctn_amd.h — forward and backward,
with no Khatri-Rao half, GEMM result or aligned view ever materialised in HBM.
"""
from __future__ import annotations

import math
from logging import getLogger
from typing import Tuple

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib as L


class _EpsFunction(torch.autograd.Function):
    """``keep``: the training forward may leave its GEMM result for the backward (`dctn_eps_fwd_save`) - what torch's
    autograd does on the reference's path (dctn/eps.py:25-30: the result of step (0,1) is saved and the backward runs
    two GEMMs, not three).  Only the input gradient uses it, so it is kept when the input requires grad."""

    @staticmethod
    def forward(ctx, core: Tensor, input: Tensor, keep: bool = True) -> Tensor:
        C, B, H, W, Q = input.shape
        K = math.isqrt((core.ndim - 1) // C)
        O = core.shape[-1]
        dev = L.require_device(core, input)
        if core.dtype != input.dtype:
            raise TypeError(f"eps: core is {core.dtype} but input is {input.dtype}")
        core_c = core.contiguous()
        out = torch.empty((B, H - K + 1, W - K + 1, O), dtype=input.dtype, device=dev)
        prec = L.precision()
        code = L.dtype_code(input)
        lib = L.lib()
        ws = L.workspace(lib.dctn_eps_fwd_workspace_bytes(C, B, H, W, Q, K, O, code, prec), dev)
        saved = None
        if keep and ctx.needs_input_grad[1]:
            nsaved = lib.dctn_eps_saved_bytes(C, B, H, W, Q, K, O, code, prec)
            if nsaved > 0:
                saved = torch.empty(nsaved, dtype=torch.uint8, device=dev)
        if saved is None:
            L.check(
                lib.dctn_eps_fwd(input.data_ptr(), L.strides5(input), core_c.data_ptr(), out.data_ptr(),
                                 ws.data_ptr(), ws.numel(), C, B, H, W, Q, K, O, code, prec, L.stream_ptr(dev)),
                "eps forward",
            )
        else:
            rc = L.check(
                lib.dctn_eps_fwd_save(input.data_ptr(), L.strides5(input), core_c.data_ptr(), out.data_ptr(),
                                      saved.data_ptr(), saved.numel(), ws.data_ptr(), ws.numel(),
                                      C, B, H, W, Q, K, O, code, prec, L.stream_ptr(dev)),
                "eps forward",
            )
            if rc != L.SAVED:   # the forward says whether it wrote the buffer; an untouched one never reaches the backward
                saved = None
        # the kept GEMM result (up to 416 MB at cfg3a layer 2) is an autograd-managed saved tensor: freed when the
        # backward has run (not when the last reference to the graph dies), visible to saved_tensors_hooks / checkpointing
        if saved is None:
            ctx.save_for_backward(core_c, input)
        else:
            ctx.save_for_backward(core_c, input, saved)
        ctx.dims = (C, B, H, W, Q, K, O, prec)
        return out

    @staticmethod
    def backward(ctx, d_out: Tensor):
        core_c, input, *kept = ctx.saved_tensors
        C, B, H, W, Q, K, O, prec = ctx.dims
        need_dcore, need_dx = ctx.needs_input_grad[:2]
        dev = input.device
        g = d_out.contiguous()
        d_core = torch.empty_like(core_c) if need_dcore else None
        d_x = torch.empty((C, B, H, W, Q), dtype=input.dtype, device=dev) if need_dx else None
        code = L.dtype_code(input)
        nbytes = L.lib().dctn_eps_bwd_workspace_bytes(C, B, H, W, Q, K, O, code, prec, int(need_dx), int(need_dcore))
        ws = L.workspace(nbytes, dev)
        saved = kept[0] if kept else None
        common = (None if d_x is None else d_x.data_ptr(), None if d_core is None else d_core.data_ptr(),
                  ws.data_ptr(), ws.numel(), C, B, H, W, Q, K, O, code, prec, L.stream_ptr(dev))
        if saved is not None and need_dx:
            rc = L.lib().dctn_eps_bwd_saved(input.data_ptr(), L.strides5(input), core_c.data_ptr(), g.data_ptr(),
                                            saved.data_ptr(), saved.numel(), *common)
        else:
            rc = L.lib().dctn_eps_bwd(input.data_ptr(), L.strides5(input), core_c.data_ptr(), g.data_ptr(), *common)
        L.check(rc, "eps backward")
        return d_core, d_x, None


_keep_gemm_result = True


class keep_gemm_result:
    """``with keep_gemm_result(False): ...`` - forwards inside the block keep nothing for their backward, which then
    recomputes (the pre-round-3 behaviour: less memory held between forward and backward - cfg3a layer 2: 416 MB -, one
    more GEMM in the backward).  Default True, as torch's autograd does on the reference's path."""

    def __init__(self, keep: bool):
        self.keep = bool(keep)

    def __enter__(self):
        global _keep_gemm_result
        self._saved = _keep_gemm_result
        _keep_gemm_result = self.keep
        return self

    def __exit__(self, *exc):
        global _keep_gemm_result
        _keep_gemm_result = self._saved
        return False


def _check_core(core: Tensor, input: Tensor) -> None:
    num_channels, batch_size, height, width, in_size = input.shape
    kernel_size = math.isqrt((core.ndim - 1) // num_channels)
    assert core.shape[:-1] == tuple(in_size for _ in range(kernel_size**2 * num_channels))


def eps(core: Tensor, input: Tensor) -> Tensor:
    """``input``: (channels, batch, height, width, in_size); ``core``: (in_size,)*(K*K*channels) +
    (out_size,) with factor index = window position (row-major) * channels + channel.
    Returns (batch, height-K+1, width-K+1, out_size)."""
    _check_core(core, input)
    return L.on_device(_eps_on_device, core, input)


def _eps_on_device(core: Tensor, input: Tensor) -> Tensor:
    if _bf16_through_f32(core, input):
        return _EpsFunction.apply(core.float(), input.float(), _keep_gemm_result).to(torch.bfloat16)
    if _f32_through_bf16(core, input):
        return _EpsFunction.apply(core.bfloat16(), input.bfloat16(), _keep_gemm_result).float()
    return _EpsFunction.apply(core, input, _keep_gemm_result)


def _f32_through_bf16(core: Tensor, input: Tensor) -> bool:
    """float32 tensors under ``set_float32_matmul_precision("bf16")`` ("operands rounded to bf16, float32
    accumulate") whose core is too large for the bf16 register family: the two-halves GEMMs on the bf16 matrix
    cores take them (cfg3a: 2.8 ms instead of 10.9 ms per step); with the default 'exact' policy nothing changes."""
    if core.dtype != torch.float32 or not core.is_cuda or (L.precision() & L.PREC_MASK) != L.PREC_BF16:
        return False
    C, B, H, W, Q = input.shape
    K = math.isqrt((core.ndim - 1) // C)
    args = (C, B, H, W, Q, K, core.shape[-1])
    lib = L.lib()
    return (lib.dctn_eps_family(*args, L._DTYPE_CODE[torch.float32], L.PREC_BF16) != 1
            and lib.dctn_eps_family(*args, L._DTYPE_CODE[torch.bfloat16], L.PREC_BF16) == 3)


def _bf16_through_f32(core: Tensor, input: Tensor) -> bool:
    """bf16 tensors whose core is outside the bf16 register family (deeper layers, Q > 2) would land on the
    generic kernels; the exact-f32 matrix-core families (bigcore, two-halves GEMMs) take them instead: bf16 storage, f32 arithmetic
    (the casts are three small elementwise kernels next to millisecond GEMMs; autograd casts the gradients
    back)."""
    if core.dtype != torch.bfloat16 or not core.is_cuda:
        return False
    C, B, H, W, Q = input.shape
    K = math.isqrt((core.ndim - 1) // C)
    args = (C, B, H, W, Q, K, core.shape[-1])
    lib, prec = L.lib(), L.precision()
    return (lib.dctn_eps_family(*args, L.dtype_code(core), prec) == 0
            and lib.dctn_eps_family(*args, L._DTYPE_CODE[torch.float32], prec) in (2, 3))


def eps_one_by_one(core: Tensor, input: Tensor) -> Tensor:
    """Same contraction, evaluated another way.  In the reference this is a second, factor-by-factor evaluation order
    (dctn/eps.py:43-63) that its tests hold against ``eps``; here it runs - forward and backward - on the generic
    kernels (`DCTN_OPT_GENERIC_KERNELS`: one lane per window walks the core rows digit by digit, no matrix cores, no
    Khatri-Rao halves, nothing shared with the MFMA families ``eps`` dispatches to), so the two names cross-check each
    other on the device as they do in the reference.  Like there it is the slow one: tests only."""
    _check_core(core, input)
    with L.options(L.OPT_GENERIC_KERNELS):
        out = L.on_device(_EpsFunction.apply, core, input)
    num_channels, batch_size, height, width, _ = input.shape
    kernel_size = math.isqrt((core.ndim - 1) // num_channels)
    assert out.shape == (batch_size, height - kernel_size + 1, width - kernel_size + 1, core.shape[-1])
    return out


def calc_eps_shape(kernel_size: int, in_num_channels: int, in_size: int, out_size: int) -> Tuple[int, ...]:
    return (in_size,) * (kernel_size**2 * in_num_channels) + (out_size,)


spec_to_shape = calc_eps_shape


def total_in_dim_size(kernel_size: int, in_num_channels: int, in_size: int) -> int:
    return in_size ** (in_num_channels * kernel_size**2)


def is_eps(a: Tensor) -> bool:
    """Whether ``a`` can be an EPS core judging by its shape: all dims but the last are equal."""
    return a.ndim >= 2 and all(d == a.shape[0] for d in a.shape[:-1])


def matrix_shape(eps_core: Tensor) -> Tuple[int, int]:
    assert is_eps(eps_core)
    return eps_core.shape[-1], math.prod(eps_core.shape[:-1])


def contract_on_input_dims(a: Tensor, b: Tensor) -> Tensor:
    """(out dim of a, out dim of b): both cores contracted over all their input dims (`dctn_fiber_gram`)."""
    from . import tn_inner

    assert is_eps(a)
    assert is_eps(b)
    if a.dtype == b.dtype and tn_inner.covers(a.shape[-1], b.shape[-1]):
        return tn_inner.gram_over_input_dims(a, b)
    return a.reshape(-1, a.shape[-1]).T @ b.reshape(-1, b.shape[-1])   # out sizes beyond the kernels' 32: library GEMM


def inner_product(a: Tensor, b: Tensor) -> Tensor:
    from . import tn_inner

    assert a.shape == b.shape
    assert is_eps(a)
    return tn_inner.dot(a, b)


@torch.no_grad()
def transform_in_slices(eps_core: Tensor, x: Tensor, batch_size: int) -> Tensor:
    """Applies ``eps`` to ``x`` (channels, dataset_size, H, W, in_size) slice by slice along the
    dataset dim, without autograd; returns (1, dataset_size, H', W', out_size)."""
    assert is_eps(eps_core)
    return torch.cat([eps(eps_core, part) for part in x.split(batch_size, dim=1)]).unsqueeze(0)


@torch.no_grad()
def output_sums_in_slices(eps_core: Tensor, x: Tensor, batch_size: int) -> Tuple[int, Tensor]:
    """(count, float64 tensor [sum y, sum y^2]) over every value of ``eps(eps_core, x)``, slice by slice along the
    data-set dim like `transform_in_slices` - but through `dctn_eps_fwd_stats`: the sums are an epilogue of the
    forward (in-kernel for the register-resident family, one reduction pass over a cache-resident slice for the
    others), the (dataset_size, H', W', out) output is never materialised."""
    assert is_eps(eps_core)
    _check_core(eps_core, x)
    dev, staged = L.placement(eps_core, x)
    core = eps_core.to(dev).contiguous()
    C, _, H, W, Q = x.shape
    K = math.isqrt((core.ndim - 1) // C)
    O = core.shape[-1]
    lib, code, pol = L.lib(), L.dtype_code(core), L.precision()
    stats = torch.zeros(2, dtype=torch.float64, device=dev)
    count = 0
    for part in x.split(batch_size, dim=1):
        part = part.to(dev)
        if part.dtype != core.dtype:
            raise TypeError(f"eps: core is {core.dtype} but input is {part.dtype}")
        B = part.shape[1]
        ws = L.workspace(lib.dctn_eps_fwd_stats_workspace_bytes(C, B, H, W, Q, K, O, code, pol), dev)
        L.check(lib.dctn_eps_fwd_stats(part.data_ptr(), L.strides5(part), core.data_ptr(), stats.data_ptr(), ws.data_ptr(),
                                       ws.numel(), C, B, H, W, Q, K, O, code, pol, L.stream_ptr(dev)), "eps forward statistics")
        count += B * (H - K + 1) * (W - K + 1) * O
    return count, (stats.cpu() if staged else stats)


def make_eps_unit_theoretical_output_std(
    kernel_size: int, in_num_channels: int, in_size: int, out_size: int, device: torch.device, dtype: torch.dtype
) -> Tensor:
    """randn scaled by (in_size^(K*K*C))^-1/2, which keeps the std of a unit-variance window."""
    std = total_in_dim_size(kernel_size, in_num_channels, in_size) ** -0.5
    getLogger(f"{__name__}.make_eps_unit_theoretical_output_std").info(
        f"Multiplying the output of randn by {std:.30e}"
    )
    shape = calc_eps_shape(kernel_size, in_num_channels, in_size, out_size)
    return std * torch.randn(*shape, dtype=dtype).to(device)


def make_eps_unit_empirical_output_std(
    kernel_size: int, out_size: int, input: Tensor, device: torch.device, dtype: torch.dtype, batch_size: int
) -> Tensor:
    """randn core rescaled so that its output over ``input`` has unit (biased) std.  Under
    ``torch.distributed`` every rank passes its shard of the dataset: the random core is rank 0's and the
    std is that of all shards together, so all ranks end with the same core a single process would get."""
    from . import ddp

    num_channels, dataset_size, height, width, in_size = input.shape
    core = torch.randn(*(in_size,) * (kernel_size**2 * num_channels), out_size, dtype=dtype).to(device)
    ddp.broadcast_parameters([core])
    # the output over the data set is only needed for its std: count, sum and sum of squares come out of the
    # forward's epilogue (`dctn_eps_fwd_stats`), the (dataset, H', W', out) tensor of eps.py:172 never exists
    count, sums = output_sums_in_slices(core, input.to(device, dtype), batch_size)
    inverse_output_std = ddp.global_biased_std_from_sums(count, sums.to(core.device)).to(core.dtype) ** -1
    logger = getLogger(f"{__name__}.make_eps_unit_empirical_output_std")
    logger.info(f"Multiplying the output of randn by {inverse_output_std:.30e}")
    core *= inverse_output_std
    logger.info(f"Initialized an EPS with empirical std = {core.std(unbiased=False):.30e}")
    return core


class EPS(nn.Module):
    def __init__(self, kernel_size: int, in_num_channels: int, in_size: int, out_size: int):
        super().__init__()
        self.kernel_size = kernel_size
        self.in_num_channels = in_num_channels
        self.in_size = in_size
        self.out_size = out_size
        self.core = nn.Parameter(
            make_eps_unit_theoretical_output_std(
                kernel_size, in_num_channels, in_size, out_size, torch.device("cpu"), torch.float32
            )
        )

    @property
    def matrix_shape(self) -> Tuple[int, int]:
        return matrix_shape(self.core)

    def forward(self, input: Tensor) -> Tensor:
        return eps(self.core, input)
