"""EPSesPlusLinear: a stack of EPS layers followed by a linear classifier head.

Mirror of the reference's dctn/eps_plus_linear.py:30-196: constructor arguments
(``epses_specs, initialization, p, device, dtype, image_size=28, Q_0=2``), parameters
``epses`` (ParameterList) / ``linear`` / buffer ``p`` (state_dict keys ``p``, ``epses.i``,
``linear.weight``, ``linear.bias``), forward, the two L2 regularisers.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from logging import getLogger
from typing import Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import _lib as L
from . import eps, epses_composition
from .utils import OneTensorInitialization, ZeroCenteredNormalInitialization, ZeroCenteredUniformInitialization


@dataclass(frozen=True)
class UnitEmpiricalOutputStd:
    input: Tensor
    batch_size: int = 128


class UnitTheoreticalOutputStd:
    pass


@dataclass(frozen=True)
class ManuallyChosenInitialization:
    epses: Tuple[OneTensorInitialization, ...]
    linear_weight: OneTensorInitialization
    linear_bias: OneTensorInitialization


Initialization = Union[UnitEmpiricalOutputStd, UnitTheoreticalOutputStd, ManuallyChosenInitialization]

# Host-side routing switches (plain module attributes; tests flip them to compare the paths):
#   FUSED_HEAD  - last EPS layer + flatten + linear head as one autograd node (`_EpsLinearHeadFunction`)
#   HEAD_BWD    - backward of the stand-alone linear head: "hip" (`dctn_linear_head_bwd`, the default) or "blas" (library GEMMs)
#   FUSED_HEAD_FWD - inside that node: forward of layer + head as one kernel (`dctn_eps_head_fwd`) or as two
FUSED_HEAD = True
FUSED_HEAD_FWD = True
HEAD_BWD = "hip"


class _LinearHeadFunction(torch.autograd.Function):
    """`F.linear(feat, weight, bias)` for a skinny output (<= 16 classes) on the HIP kernels of
    dctn_amd/csrc/linear_head.hip, forward and backward (the three library GEMMs of a 10-class head cost more than the
    EPS contraction at batch 1024): float32, float64 and bfloat16, any feature count (bf16 with a multiple of 8 features
    and 16-byte aligned tensors on the vectorised matrix-core kernels, everything else on the scalar streaming ones)."""

    @staticmethod
    def supported(feat: Tensor, weight: Tensor, bias) -> bool:
        return (
            feat.is_cuda and bias is not None and feat.dtype in (torch.bfloat16, torch.float32, torch.float64)
            and weight.dtype == feat.dtype and bias.dtype == feat.dtype and feat.ndim == 2 and weight.shape[0] <= 16
        )

    @staticmethod
    def forward(ctx, feat: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
        dev = L.require_device(feat, weight, bias)
        f, w, b = feat.contiguous(), weight.contiguous(), bias.contiguous()
        B, F_ = f.shape
        C = w.shape[0]
        out = torch.empty((B, C), dtype=f.dtype, device=dev)
        L.check(L.lib().dctn_linear_head_fwd(f.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), B, F_, C,
                                             L.dtype_code(f), L.stream_ptr(dev)), "linear head forward")
        ctx.save_for_backward(f, w)
        return out

    @staticmethod
    def backward(ctx, d_out: Tensor):
        """`dctn_linear_head_bwd`: dFeat and the dWeight slices in one launch, a second one sums the slices (fixed
        order) and emits dBias.  `eps_plus_linear.HEAD_BWD = "blas"` selects three library GEMM / reduction launches
        instead.  Device time per call, replayed from a HIP graph (tools/time_head_bwd.py, round 4, 10 classes):
        bf16 B = 1024, F = 2704: 20.2 us here against 19.9 through the library; bf16 B = 128, F = 5000: 13.5 against 17.7;
        float32 B = 128, F = 5000 (the cfg3 heads): 9.8 against 16.3; float32 B = 1024, F = 2704 (no BASELINE config): 32.6
        against 20.0 - the scalar streaming kernel loses there, and "blas" remains the switch for it."""
        f, w = ctx.saved_tensors
        need_f, need_w, need_b = ctx.needs_input_grad
        return _head_backward(f, w, d_out.contiguous(), need_f, need_w, need_b)


def _head_backward(f: Tensor, w: Tensor, g: Tensor, need_f: bool, need_w: bool, need_b: bool):
    """(dFeat, dWeight, dBias) of ``f @ w.T + bias`` for the incoming gradient ``g`` (contiguous)."""
    if HEAD_BWD != "hip" or w.shape[0] > 16:
        return (g @ w if need_f else None), (g.t() @ f if need_w else None), (g.sum(0) if need_b else None)
    dev = f.device
    B, F_ = f.shape
    C = w.shape[0]
    d_f = torch.empty_like(f) if need_f else None
    d_w = torch.empty_like(w) if (need_w or need_b) else None
    d_b = torch.empty((C,), dtype=w.dtype, device=dev) if (need_w or need_b) else None
    code = L.dtype_code(f)
    ws = L.workspace(L.lib().dctn_linear_head_bwd_workspace_bytes(B, F_, C, code), dev)
    L.check(
        L.lib().dctn_linear_head_bwd(
            f.data_ptr(), w.data_ptr(), g.data_ptr(), None if d_f is None else d_f.data_ptr(),
            None if d_w is None else d_w.data_ptr(), None if d_b is None else d_b.data_ptr(),
            ws.data_ptr(), ws.numel(), B, F_, C, code, L.stream_ptr(dev)),
        "linear head backward",
    )
    return d_f, (d_w if need_w else None), (d_b if need_b else None)


class _EpsLinearHeadFunction(torch.autograd.Function):
    """The last EPS layer, the flatten and the linear head as ONE autograd node (reference:
    dctn/eps_plus_linear.py:144-147).  Forward is the EPS kernel followed by the head kernel; in the
    backward the gradient of the features is never materialised: `dctn_eps_head_bwd` forms it inside
    the dCore kernel from dLogits and the head weight and accumulates dWeight / dBias in the same pass
    (two kernels instead of three library GEMM / reduction launches + two EPS kernels, and no write +
    read of the (B, H'*W'*O) gradient).  Used when the layer's input needs no gradient (it is
    the dataset tensor for a single-EPS model) and the shape is in one of the two register-resident
    families (bf16: eps_mfma.hip; float32, the reference's own dtype: eps_q2f32.hip); anything else takes
    the two separate nodes."""

    @staticmethod
    def supported(core: Tensor, x: Tensor, weight: Tensor, bias) -> bool:
        if not (x.is_cuda and bias is not None and not x.requires_grad):
            return False
        if not (x.dtype == core.dtype == weight.dtype == bias.dtype and x.dtype in (torch.bfloat16, torch.float32)):
            return False
        n, o, cout = core.ndim - 1, core.shape[-1], weight.shape[0]
        if not (x.shape[-1] == 2 and n in (8, 9) and o in (2, 4) and cout <= 16 and weight.data_ptr() % 16 == 0 and FUSED_HEAD):
            return False
        if x.dtype == torch.float32:   # the exact-f32 register family (eps_q2f32.hip) under the default policy
            return (L.precision() & L.PREC_MASK) == L.PREC_EXACT
        return cout % 2 == 0 and weight.shape[1] % 8 == 0

    @staticmethod
    def forward(ctx, core: Tensor, x: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
        dev = L.require_device(core, x, weight, bias)
        C, B, H, W, Q = x.shape
        K = math.isqrt((core.ndim - 1) // C)
        O = core.shape[-1]
        core_c, w, b = core.contiguous(), weight.contiguous(), bias.contiguous()
        prec, code = L.precision(), L.dtype_code(x)
        feat = torch.empty((B, (H - K + 1) * (W - K + 1) * O), dtype=x.dtype, device=dev)
        assert w.shape[1] == feat.shape[1]
        cout = w.shape[0]
        out = torch.empty((B, cout), dtype=x.dtype, device=dev)
        # one kernel for the layer and the head (`dctn_eps_head_fwd`); shapes / layouts it does not take run as two
        rc = L.ERR_UNSUPPORTED
        if FUSED_HEAD_FWD:
            rc = L.lib().dctn_eps_head_fwd(x.data_ptr(), L.strides5(x), core_c.data_ptr(), w.data_ptr(), b.data_ptr(),
                                           feat.data_ptr(), out.data_ptr(), C, B, H, W, Q, K, O, cout, code, prec,
                                           L.stream_ptr(dev))
        if rc == L.ERR_UNSUPPORTED:
            ws = L.workspace(L.lib().dctn_eps_fwd_workspace_bytes(C, B, H, W, Q, K, O, code, prec), dev)
            L.check(
                L.lib().dctn_eps_fwd(x.data_ptr(), L.strides5(x), core_c.data_ptr(), feat.data_ptr(), ws.data_ptr(),
                                     ws.numel(), C, B, H, W, Q, K, O, code, prec, L.stream_ptr(dev)),
                "eps forward",
            )
            L.check(L.lib().dctn_linear_head_fwd(feat.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), B,
                                                 feat.shape[1], cout, code, L.stream_ptr(dev)), "linear head forward")
        else:
            L.check(rc, "eps + linear head forward")
        ctx.save_for_backward(core_c, x, feat, w)
        ctx.dims = (C, B, H, W, Q, K, O, prec)
        return out

    @staticmethod
    def backward(ctx, d_out: Tensor):
        core_c, x, feat, w = ctx.saved_tensors
        C, B, H, W, Q, K, O, prec = ctx.dims
        need_core, _, need_w, need_b = ctx.needs_input_grad
        dev, code = x.device, L.dtype_code(x)
        g = d_out.contiguous()
        if not need_core:
            _, d_w, d_b = _head_backward(feat, w, g, False, need_w, need_b)
            return None, None, d_w, d_b
        cout = w.shape[0]
        # the three gradients are carved out of ONE buffer, in parameter order (epses[-1], linear.weight,
        # linear.bias): ddp.FlatGradAllReducer then all-reduces that buffer in place, with no gather /
        # scatter kernels around the collective
        flat = torch.empty(core_c.numel() + w.numel() + cout, dtype=w.dtype, device=dev)
        d_core = flat[: core_c.numel()].view_as(core_c)
        d_w = flat[core_c.numel() : core_c.numel() + w.numel()].view_as(w) if need_w else None
        d_b = flat[core_c.numel() + w.numel() :] if need_b else None
        nbytes = max(L.lib().dctn_eps_head_bwd_workspace_bytes(C, B, H, W, Q, K, O, cout, code, prec),
                     L.lib().dctn_eps_bwd_workspace_bytes(C, B, H, W, Q, K, O, code, prec, 0, 1))
        ws = L.workspace(nbytes, dev)
        rc = L.lib().dctn_eps_head_bwd(
            x.data_ptr(), L.strides5(x), feat.data_ptr(), g.data_ptr(), w.data_ptr(), d_core.data_ptr(),
            None if d_w is None else d_w.data_ptr(), None if d_b is None else d_b.data_ptr(), ws.data_ptr(),
            ws.numel(), C, B, H, W, Q, K, O, cout, code, prec, L.stream_ptr(dev))
        if rc == L.ERR_UNSUPPORTED:   # outside the fused family: the head's own backward + the plain EPS backward
            d_feat, d_w, d_b = _head_backward(feat, w, g, True, need_w, need_b)
            rc = L.lib().dctn_eps_bwd(x.data_ptr(), L.strides5(x), core_c.data_ptr(), d_feat.data_ptr(), None,
                                      d_core.data_ptr(), ws.data_ptr(), ws.numel(), C, B, H, W, Q, K, O, code,
                                      prec, L.stream_ptr(dev))
        L.check(rc, "eps + linear head backward")
        return d_core, None, d_w, d_b


def _refresh_p_after_load(module, _incompatible_keys) -> None:
    module._refresh_p()


class EPSesPlusLinear(nn.Module):
    def __init__(
        self,
        epses_specs: Tuple[Tuple[int, int], ...],
        initialization: Initialization,
        p: float,
        device: torch.device,
        dtype: torch.dtype,
        image_size: int = 28,
        Q_0: int = 2,
    ):
        """``epses_specs``: (kernel_size, out_size) per layer; ``p``: probability of KEEPING a
        component of a core during training (1 = no dropout)."""
        assert 0.0 < p <= 1
        super().__init__()
        if isinstance(initialization, UnitEmpiricalOutputStd):
            assert initialization.input.shape[2] == image_size
            assert initialization.input.shape[3] == image_size
            cores = epses_composition.make_epses_composition_unit_empirical_output_std(
                epses_specs, initialization.input, device, dtype, initialization.batch_size
            )
        elif isinstance(initialization, UnitTheoreticalOutputStd):
            cores = epses_composition.make_epses_composition_unit_theoretical_output_std(
                epses_specs, Q_0, device, dtype
            )
        elif isinstance(initialization, ManuallyChosenInitialization):
            cores = epses_composition.make_epses_composition_manually_chosen_inializations(
                epses_specs, initialization.epses, Q_0, device, dtype
            )
        else:
            raise ValueError(f"initialization={initialization} is not {Initialization}")
        self.epses = nn.ParameterList(nn.Parameter(core) for core in cores)

        side = image_size - sum(k for k, _ in epses_specs) + len(epses_specs)
        self.linear = nn.Linear(side * side * eps.matrix_shape(self.epses[-1])[0], 10, bias=True).to(dtype)
        if isinstance(initialization, ManuallyChosenInitialization):
            for param, init in (
                (self.linear.weight, initialization.linear_weight),
                (self.linear.bias, initialization.linear_bias),
            ):
                if isinstance(init, ZeroCenteredNormalInitialization):
                    param.data.copy_(torch.randn_like(param) * init.std)
                elif isinstance(init, ZeroCenteredUniformInitialization):
                    param.data.copy_(torch.rand_like(param) * (2 * init.maximum) - init.maximum)
                else:
                    raise ValueError(f"initialization={initialization} must be {ManuallyChosenInitialization}")
        else:
            logger = getLogger(f"{__name__}.EPSesPlusLinear.__init__")
            weight_std = self.linear.in_features**-0.5 / 4.0
            self.linear.weight.data.copy_(torch.randn_like(self.linear.weight) * weight_std)
            logger.info(f"Initialized linear.weight as randn * {weight_std:.30e}")
            bias_max = self.linear.in_features**-0.5
            self.linear.bias.data.copy_(torch.rand_like(self.linear.bias) * (2 * bias_max) - bias_max)
            logger.info(f"Initialized linear.bias from Uniform[{-bias_max:.30e}, {bias_max:.30e}]")
        self.linear.to(device)
        self.register_buffer("p", torch.tensor(p, device=device, dtype=dtype))
        # Host copy of `p` for the dropout gate (reading the buffer every forward would be a device
        # synchronisation, impossible under graph capture).  `p` is part of the state_dict, so the copy is
        # refreshed whenever a checkpoint is loaded.
        self._p_float = float(p)
        self.register_load_state_dict_post_hook(_refresh_p_after_load)   # a module-level function: the model stays picklable

    def _refresh_p(self) -> None:
        """Call after changing the ``p`` buffer in place (loading a state_dict does it by itself)."""
        self._p_float = float(self.p)

    def forward(self, input: Tensor) -> Tensor:
        """``input``: (channels, batch, height, width, Q_0) -> logits (batch, 10).  A model and input that live
        on the CPU are staged to the GPU once for the whole forward (`_lib.placement`)."""
        dev, staged = L.placement(input, *self.epses, self.linear.weight, self.linear.bias)
        if staged:
            return self._forward_on_device(
                input.to(dev), tuple(core.to(dev) for core in self.epses), self.linear.weight.to(dev),
                self.linear.bias.to(dev), self.p.to(dev)).cpu()
        return self._forward_on_device(input, tuple(self.epses), self.linear.weight, self.linear.bias, self.p)

    def _forward_on_device(self, input: Tensor, cores: Tuple[Tensor, ...], weight: Tensor, bias: Tensor,
                           p: Tensor) -> Tensor:
        if self._p_float < 1.0 and self.training:   # component dropout, dctn/eps_plus_linear.py:139-143
            cores = tuple(self.dropout_mask(core, p) * core / p for core in cores)
        x = input
        for core in cores[:-1]:   # as epses_composition.contract_with_input
            x = eps.eps(core, x).unsqueeze(0)
        if _EpsLinearHeadFunction.supported(cores[-1], x, weight, bias):
            eps._check_core(cores[-1], x)
            return _EpsLinearHeadFunction.apply(cores[-1], x, weight, bias)
        features = eps.eps(cores[-1], x)
        flat = features.reshape(features.shape[0], -1)
        if _LinearHeadFunction.supported(flat, weight, bias):
            return _LinearHeadFunction.apply(flat, weight, bias)
        return F.linear(flat, weight, bias)

    @staticmethod
    def dropout_mask(core: Tensor, p: Tensor) -> Tensor:
        """Bernoulli(p) keep-mask of one core (the reference draws ``self.p.expand_as(core).bernoulli()``); a
        method of its own so that a test can pin the mask."""
        return p.expand_as(core).bernoulli()

    def epswise_l2_regularizer(self) -> Tensor:
        """||linear.weight||^2 + sum of squared Frobenius norms of the cores (bias excluded)."""
        return self.linear.weight.norm(p="fro") ** 2 + epses_composition.epswise_squared_fro_norm(self.epses)

    def epses_composition_l2_regularizer(self) -> Tensor:
        return self.linear.weight.norm(p="fro") ** 2 + epses_composition.inner_product(self.epses, self.epses)

    @torch.no_grad()
    def log_intermediate_reps_stats(self, x: Tensor, batch_size: int = 128) -> None:
        """Logs mean / std of every intermediate representation and of the K x K windows in front of every
        EPS, as rank-one tensors (dctn/eps_plus_linear.py:161-196), as if in eval mode.  The window
        statistics come from the one-pass ``dctn_window_stats`` kernel instead of K*K stacked copies of the
        representation; like the reference's ``std_over_batch`` (rank_one_tensor.py:107-110, which drops its
        ``unbiased`` argument) the window sigma carries Bessel's correction."""
        from .window_stats import window_mean_var

        logger = getLogger(f"{__name__}.EPSesPlusLinear.log_intermediate_reps_stats")
        logger.info("Logging intermediate reps stats as if self.training == False")

        def log_one(t: Tensor, name: str) -> None:
            mu, sigma = t.mean(), t.std(unbiased=False)
            logger.info(f"{name}: mu={mu:.7e}, sigma={sigma:.7e}, mu^2+sigma^2={mu**2+sigma**2:.7e}, shape={tuple(t.shape)}")

        for n, core in enumerate(self.epses):
            log_one(x, f"x_{n}")
            kernel_size = math.isqrt(core.ndim - 1)
            assert kernel_size**2 == core.ndim - 1
            mu, var = window_mean_var(x, kernel_size)
            sigma = var**0.5
            C, B, H, W, Q = x.shape
            logger.info(
                f"w_{n}: mu={mu:.7e}, sigma={sigma:.7e}, mu^2+sigma^2={mu**2+sigma**2:.7e}, "
                f"batch_shape={(B, H - kernel_size + 1, W - kernel_size + 1)}, "
                f"num_factors={kernel_size**2 * C}, num_coordinates_in_one_factor={Q}"
            )
            x = eps.transform_in_slices(core, x, batch_size)
        flat = x.reshape(x.shape[1], -1)
        log_one(flat, f"x_{len(self.epses)}")
        log_one(F.linear(flat, self.linear.weight), "output_of_linear_without_bias")
        log_one(self.linear(flat), "output_of_linear_with_bias")
