"""ConvSBS: a tensor-train / tensor-ring ("snake" string) of per-pixel cores laid over a window
and slid over the image; ManyConvSBS: several strings over the same input.

Mirror of the reference's dctn/conv_sbs.py:27-370 (same class names, constructor arguments,
``cores`` ParameterList, methods).  ``ConvSBS.forward`` is the hot path: one HIP kernel sweeps the
string per window (dctn_amd/csrc/convsbs_*.hip) instead of materialising one (B,H',W',o,l,r)
tensor per core in HBM; backward is the adjoint sweep.  The parameter-only contractions
(``sum``, ``squared_fro_norm``, ``as_explicit_tensor``, ``as_eps``) are tiny and go through the
cached pairwise-einsum plans of ``contraction_path_cache``.
"""
from __future__ import annotations

import logging
import math
from dataclasses import dataclass
from itertools import chain
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib as L
from .contraction_path_cache import contract
from .conv_sbs_spec import SBSSpecCore, SBSSpecString


# --- initialisation tags (dctn/conv_sbs.py:27-43)
@dataclass(frozen=True)
class DumbNormalInitialization:
    std_of_elements_of_cores: float


@dataclass(frozen=True)
class KhrulkovNormalInitialization:
    std_of_elements_of_matrix: Optional[float]


class NormalPreservingOutputStdInitialization:
    pass


@dataclass(frozen=True)
class MinRandomEyeInitialization:
    base_std: float


Initialization = Union[
    DumbNormalInitialization, KhrulkovNormalInitialization, NormalPreservingOutputStdInitialization,
    MinRandomEyeInitialization,
]


class _StringPlan:
    """What a call needs from a spec, computed once per spec (the frozen dataclass hashes by value): core shapes, the
    int arrays of the C-ABI, output size.  Rebuilding them on every call was a third of the host time of a small
    layer's forward."""

    __slots__ = ("n", "shape_tuples", "outs", "bonds", "ph", "pw", "max_h", "max_w", "out_total", "C", "q", "ws",
                 "outs_list", "bonds_list", "ph_list", "pw_list", "many_ok")

    def __init__(self, spec: SBSSpecString):
        shapes = spec.shapes
        self.n = len(spec)
        self.shape_tuples = tuple(s.as_tuple() for s in shapes)
        self.outs_list = [s.out_quantum_dim_size for s in shapes]
        self.bonds_list = list(spec.bond_sizes)
        self.ph_list = [p.h for p in spec.positions]
        self.pw_list = [p.w for p in spec.positions]
        self.outs = L.int_array(self.outs_list)
        self.bonds = L.int_array(self.bonds_list)
        self.ph = L.int_array(self.ph_list)
        self.pw = L.int_array(self.pw_list)
        self.max_h, self.max_w = spec.max_height_pos, spec.max_width_pos
        self.out_total = spec.out_total_quantum_dim_size
        self.C, self.q = spec.in_num_channels, spec.in_quantum_dim_size
        self.ws = {}   # (B, H, W, dtype code, backward) -> workspace bytes
        self.many_ok = {}   # ((B, H, W), partner plans) -> workspace bytes of the several-strings-per-launch path (0: unsupported)


_PLANS: dict = {}


def _plan(spec: SBSSpecString) -> _StringPlan:
    plan = _PLANS.get(spec)
    if plan is None:
        plan = _PLANS[spec] = _StringPlan(spec)
    return plan


def _workspace_bytes(plan: _StringPlan, B: int, H: int, W: int, code: int, backward: int) -> int:
    """backward: 0 forward scratch, 1 backward scratch, 2 the forward states a training forward leaves for the backward
    (0 bytes: this string's backward recomputes them)."""
    key = (B, H, W, code, backward)
    nbytes = plan.ws.get(key)
    if nbytes is None:
        if backward == 2:
            nbytes = L.lib().dctn_convsbs_saved_states_bytes(plan.n, plan.outs, plan.bonds, plan.C, B, H, W, plan.q, plan.ph,
                                                             plan.pw, code)
        else:
            nbytes = L.lib().dctn_convsbs_workspace_bytes(plan.n, plan.outs, plan.bonds, plan.C, B, H, W, plan.q,
                                                          plan.ph, plan.pw, code, backward)
        plan.ws[key] = nbytes
    return nbytes


_sbs_flags = 0


class matrix_core_sweep:
    """``with matrix_core_sweep(): ...`` - strings whose bonds are all <= 4 run on the matrix-core sweep
    (convsbs_mfma.hip) instead of the register-resident sweep (convsbs_reg.hip) that is their default: the tests use it
    to hold both kernel families against the same expected values on the same strings (`DCTN_SBS_MATRIX_CORE_SWEEP`)."""

    def __enter__(self):
        global _sbs_flags
        self._saved = _sbs_flags
        _sbs_flags |= L.SBS_MATRIX_CORE_SWEEP
        return self

    def __exit__(self, *exc):
        global _sbs_flags
        _sbs_flags = self._saved
        return False


class _ConvSBSFunction(torch.autograd.Function):
    """x: (C, B, H, W, q) any strides; cores in string order."""

    @staticmethod
    def forward(ctx, x: Tensor, spec: SBSSpecString, *cores: Tensor) -> Tensor:
        dev = L.require_device(x, *cores)
        C, B, H, W, q = x.shape
        plan = _plan(spec)
        n = plan.n
        for core, shape in zip(cores, plan.shape_tuples):
            assert tuple(core.shape) == shape
            if core.dtype != x.dtype:
                raise TypeError(f"ConvSBS: core is {core.dtype} but input is {x.dtype}")
        assert C == plan.C and q == plan.q
        cores_c = [c.contiguous() for c in cores]
        Ho, Wo = H - plan.max_h, W - plan.max_w
        out = torch.empty((B, Ho, Wo, plan.out_total), dtype=x.dtype, device=dev)
        code = L.dtype_code(x) | _sbs_flags
        # a forward that will be differentiated leaves its forward states (a buffer of its own, alive until the backward)
        # instead of having the backward's first quarter recompute them
        saved_bytes = _workspace_bytes(plan, B, H, W, code, 2) if any(ctx.needs_input_grad) else 0
        if saved_bytes > 0:
            states = torch.empty(saved_bytes, dtype=torch.uint8, device=dev)
            ws = states
        else:
            states = None
            ws = L.workspace(_workspace_bytes(plan, B, H, W, code, 0), dev)
        rc = L.check(
            L.lib().dctn_convsbs_fwd(x.data_ptr(), L.strides5(x), L.ptr_array(cores_c), out.data_ptr(), n,
                                     plan.outs, plan.bonds, plan.ph, plan.pw, C, B, H, W, q, ws.data_ptr(), ws.numel(), code,
                                     L.stream_ptr(dev)),
            "ConvSBS forward",
        )
        if rc != L.SAVED:   # the forward says whether it wrote the states: an untouched buffer never reaches the backward
            states = None
        if states is None:
            ctx.save_for_backward(x, *cores_c)
        else:
            ctx.save_for_backward(x, *cores_c, states)
        ctx.meta = (plan, C, B, H, W, q, code, states is not None)
        return out

    @staticmethod
    def backward(ctx, d_out: Tensor):
        plan, C, B, H, W, q, code, has_states = ctx.meta
        if has_states:
            x, *cores_c, states = ctx.saved_tensors
        else:
            (x, *cores_c), states = ctx.saved_tensors, None
        n = plan.n
        dev = x.device
        need_dx = ctx.needs_input_grad[0]
        need_dcores = any(ctx.needs_input_grad[2:])
        g = d_out.contiguous()
        d_x = torch.empty((C, B, H, W, q), dtype=x.dtype, device=dev) if need_dx else None
        d_cores = None
        if need_dcores:
            # one flat buffer for the gradients of all cores of the string: the library zero-fills it with a single
            # launch, and a data-parallel reducer can all-reduce it in place
            flat = torch.empty(sum(c.numel() for c in cores_c), dtype=cores_c[0].dtype, device=dev)
            d_cores, off = [], 0
            for c in cores_c:
                d_cores.append(flat[off : off + c.numel()].view_as(c))
                off += c.numel()
        ws = L.workspace(_workspace_bytes(plan, B, H, W, code, 1), dev)
        L.check(
            L.lib().dctn_convsbs_bwd_saved(
                x.data_ptr(), L.strides5(x), L.ptr_array(cores_c), g.data_ptr(),
                None if d_x is None else d_x.data_ptr(), None if d_cores is None else L.ptr_array(d_cores),
                n, plan.outs, plan.bonds, plan.ph, plan.pw, C, B, H, W, q, ws.data_ptr(), ws.numel(),
                None if states is None else states.data_ptr(), 0 if states is None else states.numel(), code,
                L.stream_ptr(dev)),
            "ConvSBS backward",
        )
        grads = [None] * n if d_cores is None else [
            dc if need else None for dc, need in zip(d_cores, ctx.needs_input_grad[2:])
        ]
        return (d_x, None, *grads)


class _ManyConvSBSFunction(torch.autograd.Function):
    """All strings of one `ManyConvSBS` layer (dctn/conv_sbs.py:367-370) in one launch each way, for the layers the
    library takes that way (`dctn_convsbs_many_fwd`: nine-core strings of one bond <= 4 over the same window positions, the
    reference's two-snake layers of mnist.py:189-252): the input is read by one kernel, and the backward writes dX once,
    already summed over the strings.  `supported` asks the library (a workspace query: 0 = take the strings one by one)."""

    @staticmethod
    def supported(x: Tensor, specs) -> bool:
        if not (x.is_cuda and x.dtype == torch.float32 and 2 <= len(specs) <= 2 and _sbs_flags == 0):
            return False
        n = len(specs[0])
        if any(len(sp) != n for sp in specs):
            return False
        C, B, H, W, q = x.shape
        plans = [_plan(sp) for sp in specs]
        if any(C != p.C or q != p.q for p in plans):   # the single-string path raises the reference's AssertionError for it
            return False
        key = (B, H, W)
        cache = plans[0].many_ok
        tag = (key, tuple(id(p) for p in plans[1:]))
        if tag not in cache:
            args = _many_arrays(plans)
            cache[tag] = L.lib().dctn_convsbs_many_workspace_bytes(len(specs), n, args[0], args[1], C, B, H, W, q, args[2], args[3],
                                                                    L.F32)
        return cache[tag] > 0

    @staticmethod
    def forward(ctx, x: Tensor, specs, *cores: Tensor):
        dev = L.require_device(x, *cores)
        C, B, H, W, q = x.shape
        plans = [_plan(sp) for sp in specs]
        ns, n = len(plans), plans[0].n
        assert all(C == p.C and q == p.q for p in plans)   # (the library derives q^C from x: never index cores of another size)
        cores_c = [c.contiguous() for c in cores]
        for core, shape in zip(cores_c, [t for p in plans for t in p.shape_tuples]):
            assert tuple(core.shape) == shape
            if core.dtype != x.dtype:
                raise TypeError(f"ConvSBS: core is {core.dtype} but input is {x.dtype}")
        outs_arr, bonds_arr, ph_arr, pw_arr = _many_arrays(plans)
        outs = [torch.empty((B, H - p.max_h, W - p.max_w, p.out_total), dtype=x.dtype, device=dev) for p in plans]
        L.check(
            L.lib().dctn_convsbs_many_fwd(x.data_ptr(), L.strides5(x), L.ptr_array(cores_c), L.ptr_array(outs), ns, n, outs_arr,
                                          bonds_arr, ph_arr, pw_arr, C, B, H, W, q, L.F32, L.stream_ptr(dev)),
            "ManyConvSBS forward",
        )
        ctx.save_for_backward(x, *cores_c)
        ctx.meta = (plans, C, B, H, W, q)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *d_outs):
        plans, C, B, H, W, q = ctx.meta
        x, *cores_c = ctx.saved_tensors
        dev = x.device
        ns, n = len(plans), plans[0].n
        need_dx = ctx.needs_input_grad[0]
        need_dcores = any(ctx.needs_input_grad[2:])
        gs = []
        for p, g in zip(plans, d_outs):   # an output nobody used arrives as None
            gs.append(torch.zeros((B, H - p.max_h, W - p.max_w, p.out_total), dtype=x.dtype, device=dev) if g is None
                      else g.contiguous())
        d_x = torch.empty((C, B, H, W, q), dtype=x.dtype, device=dev) if need_dx else None
        d_cores = None
        if need_dcores:
            # one flat buffer per string (what a single string's backward produces too: a per-string reducer all-reduces in place)
            d_cores, k = [], 0
            for p in plans:
                cs = cores_c[k : k + n]
                flat = torch.empty(sum(c.numel() for c in cs), dtype=x.dtype, device=dev)
                off = 0
                for c in cs:
                    d_cores.append(flat[off : off + c.numel()].view_as(c))
                    off += c.numel()
                k += n
        outs_arr, bonds_arr, ph_arr, pw_arr = _many_arrays(plans)
        nbytes = L.lib().dctn_convsbs_many_workspace_bytes(ns, n, outs_arr, bonds_arr, C, B, H, W, q, ph_arr, pw_arr, L.F32)
        ws = L.workspace(nbytes, dev)
        L.check(
            L.lib().dctn_convsbs_many_bwd(
                x.data_ptr(), L.strides5(x), L.ptr_array(cores_c), L.ptr_array(gs), None if d_x is None else d_x.data_ptr(),
                None if d_cores is None else L.ptr_array(d_cores), ns, n, outs_arr, bonds_arr, ph_arr, pw_arr, C, B, H, W, q,
                ws.data_ptr(), ws.numel(), L.F32, L.stream_ptr(dev)),
            "ManyConvSBS backward",
        )
        grads = [None] * (ns * n) if d_cores is None else [
            dc if need else None for dc, need in zip(d_cores, ctx.needs_input_grad[2:])
        ]
        return (d_x, None, *grads)


def _many_arrays(plans):
    """(out sizes, bond sizes, pos_h, pos_w) of several strings, string-major, as C int arrays."""
    cat = lambda name: L.int_array([v for p in plans for v in getattr(p, name)])
    return cat("outs_list"), cat("bonds_list"), cat("ph_list"), cat("pw_list")


class ConvSBS(nn.Module):
    def __init__(self, spec: SBSSpecString, initialization: Initialization = DumbNormalInitialization(0.9)):
        super().__init__()
        logger = logging.getLogger(f"{__name__}.ConvSBS.__init__")
        self.spec = spec
        scale = (
            initialization.std_of_elements_of_cores
            if isinstance(initialization, DumbNormalInitialization)
            else float("nan")
        )
        self.cores = nn.ParameterList(
            nn.Parameter(scale * torch.randn(*shape.as_tuple())) for shape in spec.shapes
        )
        if isinstance(initialization, KhrulkovNormalInitialization):
            self.init_khrulkov_normal(initialization.std_of_elements_of_matrix)
        elif isinstance(initialization, NormalPreservingOutputStdInitialization):
            logger.info("Using normal preserving output std initialization")
            self.init_normal_preserving_output_std()
        elif isinstance(initialization, MinRandomEyeInitialization):
            self.init_min_random_eye(initialization.base_std)
        logger.info(f"self.var()**0.5={self.var()**0.5}")

    # ------------------------------------------------------------------ initialisers
    @property
    def tt_matrix_num_columns(self) -> int:
        return self.spec.in_quantum_dim_size ** (self.spec.in_num_channels * len(self.spec.cores))

    def init_khrulkov_normal(self, std_of_elements_of_matrix: Optional[float] = None) -> None:
        """i.i.d. normal cores whose variance makes the elements of the represented matrix have
        the requested std (default: Glorot-like 2/(cols+rows), Khrulkov et al., TT embeddings)."""
        logger = logging.getLogger(f"{__name__}.ConvSBS.init_khrulkov_normal")
        if std_of_elements_of_matrix is not None:
            var_matrix = std_of_elements_of_matrix**2
        else:
            var_matrix = 2 / (self.tt_matrix_num_columns + self.spec.out_total_quantum_dim_size)
        n = len(self.cores)
        prod_of_ranks = math.prod(self.spec.bond_sizes)
        var_cores = var_matrix ** (1 / n) / prod_of_ranks ** (1 / n)
        logger.info(f"bond_sizes={self.spec.bond_sizes}, prod_of_ranks={prod_of_ranks}, var_of_cores_elements={var_cores}")
        for core in self.cores:
            nn.init.normal_(core, std=math.sqrt(var_cores))

    def init_normal_preserving_output_std(self) -> None:
        """If a window's coordinates are i.i.d. with mean mu and std sigma, every output
        coordinate gets std sqrt(sigma^2 + mu^2)."""
        self.init_khrulkov_normal(self.tt_matrix_num_columns**-0.5)

    def init_min_random_eye(self, base_std: float) -> None:
        """Truncated scaled identities plus gaussian noise, so that the layer initially
        averages its window.  Needs an open chain with equal bonds and a single output core."""
        spec = self.spec
        assert spec.bond_sizes[0] == 1
        assert all(b == spec.bond_sizes[1] for b in spec.bond_sizes[1:])
        bond = spec.bond_sizes[1]
        assert spec.out_total_quantum_dim_size == max(s.out_quantum_dim_size for s in spec.shapes)
        out_dim = spec.out_total_quantum_dim_size
        total_in = spec.in_quantum_dim_size**spec.in_num_channels
        k = min(bond, out_dim)
        eye = torch.zeros(bond, bond)
        eye[:k, :k] = torch.eye(k) / total_in
        eye = eye.reshape((1, bond, bond) + (1,) * spec.in_num_channels)
        for core in list(self.cores)[1:-1]:
            core.data.zero_()
            core.data += eye.expand_as(core)
            core.data += torch.randn_like(core) * base_std / total_in
        for core in (self.cores[0], self.cores[-1]):
            core.data.zero_()
            core.data[0, 0, 0] = 1 / total_in
            assert torch.allclose(core.data.sum(), torch.tensor(1.0))
            core.data += torch.randn_like(core) * base_std / total_in

    # ------------------------------------------------------------------ parameter-only contractions
    def _cores_with(self, names):
        return chain.from_iterable((core, dims) for core, dims in zip(self.cores, names))

    def sum(self) -> Tensor:
        """Sum of all elements of the represented tensor."""
        return contract(*self._cores_with(self.spec.all_dim_names), ())

    def mean(self) -> Tensor:
        return self.sum() / float(self.spec.nelement)

    def squared_fro_norm(self) -> Tensor:
        return contract(
            *self._cores_with(self.spec.get_all_dim_names_add_suffix_to_bonds("_a")),
            *self._cores_with(self.spec.get_all_dim_names_add_suffix_to_bonds("_b")),
            (),
        )

    def fro_norm(self) -> Tensor:
        return self.squared_fro_norm() ** 0.5

    def var(self, unbiased: bool = True) -> Tensor:
        """Empirical variance of the elements (Bessel-corrected iff ``unbiased``)."""
        total = self.sum()
        n = self.spec.nelement
        mean = total / n
        divisor = n - 1 if unbiased else n
        return self.squared_fro_norm() / divisor - 2 * total / divisor * mean + n / divisor * mean**2

    def as_explicit_tensor(self) -> Tensor:
        """The represented tensor as one array, dims ordered like ``spec.all_dangling_dim_names``."""
        return contract(*self._cores_with(self.spec.all_dim_names), self.spec.all_dangling_dim_names)

    def as_eps(self) -> Tensor:
        """For a string that fills a square: the equivalent dense EPS core (out dims merged,
        input dims moved to the row-major position order EPS uses)."""
        spec = self.spec
        assert spec.max_height_pos == spec.max_width_pos
        C, n, q = spec.in_num_channels, len(spec), spec.in_quantum_dim_size
        t = self.as_explicit_tensor().reshape((q,) * (C * n) + (-1,))
        std_index = spec.get_indices_wrt_standard_order()
        perm = [0] * (C * n)
        for s, target in enumerate(std_index):
            for c in range(C):
                perm[target * C + c] = s * C + c
        return t.permute(*perm, C * n)

    # ------------------------------------------------------------------ hot path
    def forward(self, input: Union[Tensor, Tuple[Tensor, ...]], /) -> Tensor:
        """``input``: a tensor whose first dim is channels, or a tuple of per-channel tensors
        (batch, height, width, q).  Returns (batch, height', width', prod of out sizes)."""
        x = input if isinstance(input, Tensor) else torch.stack(tuple(input))
        spec = self.spec
        return L.on_device(lambda x_, *cores: _ConvSBSFunction.apply(x_, spec, *cores), x, *self.cores)

    def multiply_by_scalar(self, scalar: float, /):
        """Multiplies the represented tensor by ``scalar`` in place (spread over the cores)."""
        for core in self.cores:
            core.data *= scalar ** (1 / len(self.cores))
        return self


class ManyConvSBS(nn.Module):
    def __init__(
        self,
        in_num_channels: int,
        in_quantum_dim_size: int,
        bond_dim_size: int,
        trace_edge: bool,
        cores_specs: Tuple[Tuple[SBSSpecCore, ...], ...],
        initializations: Optional[Tuple[Initialization, ...]] = None,
    ):
        """``initializations is None`` -> ConvSBS's default initialisation."""
        super().__init__()
        if initializations is not None:
            assert len(initializations) == len(cores_specs)
        specs = tuple(
            SBSSpecString(
                cores_spec,
                (bond_dim_size if trace_edge else 1,) + (bond_dim_size,) * (len(cores_spec) - 1),
                in_num_channels,
                in_quantum_dim_size,
            )
            for cores_spec in cores_specs
        )
        out_sizes = [s.out_total_quantum_dim_size for s in specs]
        assert all(o == out_sizes[0] for o in out_sizes[1:])
        if initializations is None:
            self.strings = nn.ModuleList([ConvSBS(s) for s in specs])
        else:
            self.strings = nn.ModuleList([ConvSBS(s, i) for s, i in zip(specs, initializations)])

    def forward(self, channels: Union[Tensor, Tuple[Tensor, ...]], /) -> Tuple[Tensor, ...]:
        x = channels if isinstance(channels, Tensor) else torch.stack(tuple(channels))
        dev, staged = L.placement(x, *self.parameters())
        if staged:   # CPU module and input: the input crosses to the GPU once, not once per string
            xd = x.to(dev)
            return tuple(
                _ConvSBSFunction.apply(xd, s.spec, *(c.to(dev) for c in s.cores)).cpu() for s in self.strings
            )
        specs = tuple(string.spec for string in self.strings)
        if _ManyConvSBSFunction.supported(x, specs):   # the strings of the layer in one launch each way
            return _ManyConvSBSFunction.apply(x, specs, *(c for string in self.strings for c in string.cores))
        return tuple(string(x) for string in self.strings)
