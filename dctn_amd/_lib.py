"""ctypes binding of the C-ABI kernel library (include/dctn_amd.h).

The library is the product path.  There is deliberately NO fallback: if the shared object is
missing, or no MI355X is visible, the call raises (CPU tensors are staged to the GPU, see `placement`).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdctn_amd.so")
CSRC = os.path.join(_HERE, "csrc")

F32, F64, BF16 = 0, 1, 2
PREC_EXACT, PREC_BF16, PREC_MASK = 0, 1, 0xFF
OPT_F32_PREFER_HALVES, OPT_SMALL_CHUNKS, OPT_MAIN_KERNEL_ONLY, OPT_GENERIC_KERNELS = 1 << 8, 1 << 9, 1 << 10, 1 << 11
ERR_BAD_SHAPE, ERR_BAD_DTYPE, ERR_UNSUPPORTED, ERR_WORKSPACE, ERR_LAUNCH, ERR_NULL = -1, -2, -3, -4, -5, -6
SAVED, PARTIAL = 1, 2   # positive success codes (include/dctn_amd.h)
SBS_MATRIX_CORE_SWEEP = 1 << 8   # OR-ed into the dtype argument of the dctn_convsbs_* calls

_DTYPE_CODE = {torch.float32: F32, torch.float64: F64, torch.bfloat16: BF16}

c_int, c_i64, c_void, c_size = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
_I64x5 = ctypes.POINTER(c_i64)
_IntP = ctypes.POINTER(c_int)
_PtrP = ctypes.POINTER(c_void)

# name -> (restype, argtypes); mirrors include/dctn_amd.h one to one
SIGNATURES = {
    "dctn_version": (c_int, []),
    "dctn_strerror": (ctypes.c_char_p, [c_int]),
    "dctn_last_kernel": (ctypes.c_char_p, []),
    "dctn_eps_family": (c_int, [c_int] * 9),
    "dctn_eps_fwd_workspace_bytes": (c_size, [c_int] * 7 + [c_int] * 2),
    "dctn_eps_fwd": (c_int, [c_void, _I64x5, c_void, c_void, c_void, c_size] + [c_int] * 7 + [c_int, c_int, c_void]),
    "dctn_eps_saved_bytes": (c_size, [c_int] * 7 + [c_int] * 2),
    "dctn_eps_fwd_save": (c_int, [c_void, _I64x5, c_void, c_void, c_void, c_size, c_void, c_size] + [c_int] * 7 + [c_int, c_int, c_void]),
    "dctn_eps_bwd_saved": (c_int, [c_void, _I64x5, c_void, c_void, c_void, c_size, c_void, c_void, c_void, c_size]
                           + [c_int] * 7 + [c_int, c_int, c_void]),
    "dctn_eps_fwd_stats_workspace_bytes": (c_size, [c_int] * 7 + [c_int] * 2),
    "dctn_eps_fwd_stats": (c_int, [c_void, _I64x5, c_void, c_void, c_void, c_size] + [c_int] * 7 + [c_int, c_int, c_void]),
    "dctn_eps_bwd_workspace_bytes": (c_size, [c_int] * 7 + [c_int] * 4),
    "dctn_eps_bwd": (c_int, [c_void, _I64x5, c_void, c_void, c_void, c_void, c_void, c_size]
                     + [c_int] * 7 + [c_int, c_int, c_void]),
    "dctn_eps_head_fwd": (c_int, [c_void, _I64x5, c_void, c_void, c_void, c_void, c_void] + [c_int] * 8 + [c_int, c_int, c_void]),
    "dctn_eps_head_bwd_workspace_bytes": (c_size, [c_int] * 8 + [c_int, c_int]),
    "dctn_eps_head_bwd": (c_int, [c_void, _I64x5, c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_size]
                          + [c_int] * 8 + [c_int, c_int, c_void]),
    "dctn_ce_loss_fwd": (c_int, [c_void, c_void, c_void, c_i64, c_int, c_int, c_void]),
    "dctn_ce_loss_fwd_grad": (c_int, [c_void, c_void, c_void, c_void, c_i64, c_int, c_int, c_void]),
    "dctn_ce_loss_bwd": (c_int, [c_void, c_void, c_void, c_void, c_i64, c_int, c_int, c_void]),
    "dctn_sgd_l2_num_partials": (c_int, [c_i64]),
    "dctn_sgd_l2_step": (c_int, [c_void, c_void, c_void, c_void, c_i64, c_i64, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_float, c_int, c_int, c_void]),
    "dctn_window_stats": (c_int, [c_void, _I64x5, c_void] + [c_int] * 6 + [c_int, c_void]),
    "dctn_phi_window_stats": (c_int, [c_void, c_void, c_int, c_int, c_int, c_int, c_void]),
    "dctn_phi_expand": (c_int, [c_void, c_void, c_i64, ctypes.c_float, c_int, c_void]),
    "dctn_convsbs_workspace_bytes": (c_size, [c_int, _IntP, _IntP] + [c_int] * 5 + [_IntP, _IntP, c_int, c_int]),
    "dctn_convsbs_fwd": (c_int, [c_void, _I64x5, _PtrP, c_void, c_int, _IntP, _IntP, _IntP, _IntP]
                         + [c_int] * 5 + [c_void, c_size, c_int, c_void]),
    "dctn_convsbs_bwd": (c_int, [c_void, _I64x5, _PtrP, c_void, c_void, _PtrP, c_int, _IntP, _IntP, _IntP, _IntP]
                         + [c_int] * 5 + [c_void, c_size, c_int, c_void]),
    "dctn_convsbs_saved_states_bytes": (c_size, [c_int, _IntP, _IntP] + [c_int] * 5 + [_IntP, _IntP, c_int]),
    "dctn_convsbs_bwd_saved": (c_int, [c_void, _I64x5, _PtrP, c_void, c_void, _PtrP, c_int, _IntP, _IntP, _IntP, _IntP]
                               + [c_int] * 5 + [c_void, c_size, c_void, c_size, c_int, c_void]),
    "dctn_convsbs_many_workspace_bytes": (c_size, [c_int, c_int, _IntP, _IntP] + [c_int] * 5 + [_IntP, _IntP, c_int]),
    "dctn_convsbs_many_fwd": (c_int, [c_void, _I64x5, _PtrP, _PtrP, c_int, c_int, _IntP, _IntP, _IntP, _IntP] + [c_int] * 5 + [c_int, c_void]),
    "dctn_convsbs_many_bwd": (c_int, [c_void, _I64x5, _PtrP, _PtrP, c_void, _PtrP, c_int, c_int, _IntP, _IntP, _IntP, _IntP]
                              + [c_int] * 5 + [c_void, c_size, c_int, c_void]),
    "dctn_logmatmulexp_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int, c_i64, c_i64, c_int]),
    "dctn_logmatmulexp_fwd": (c_int, [c_void, c_void, c_void, c_void, c_size, c_i64, c_int, c_int, c_int, c_i64, c_i64, c_int, c_void]),
    "dctn_logmatmulexp_bwd": (c_int, [c_void] * 7 + [c_size, c_i64, c_int, c_int, c_int, c_i64, c_i64, c_int, c_void]),
    "dctn_logmatmulexp_fold_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int, c_int]),
    "dctn_logmatmulexp_fold_fwd": (c_int, [c_void, c_void, c_i64, c_int, c_int, c_int, c_void]),
    "dctn_linear_head_fwd": (c_int, [c_void, c_void, c_void, c_void, c_i64, c_int, c_int, c_int, c_void]),
    "dctn_linear_head_bwd_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_int]),
    "dctn_linear_head_bwd": (c_int, [c_void] * 7 + [c_size, c_i64, c_int, c_int, c_int, c_void]),
    "dctn_mode_product": (c_int, [c_void, c_void, c_void, c_i64, c_int, c_int, c_i64, c_int, c_void]),
    "dctn_fiber_gram_workspace_bytes": (c_size, [c_i64, c_int, c_int, c_i64, c_int]),
    "dctn_fiber_gram": (c_int, [c_void, c_void, c_void, c_void, c_size, c_i64, c_int, c_int, c_i64, c_int, c_void]),
    "dctn_logmatmulexp_fold_bwd": (c_int, [c_void, c_void, c_void, c_void, c_size, c_i64, c_int, c_int, c_int, c_void]),
    "dctn_ar_handle_bytes": (c_size, []),
    "dctn_ar_create": (c_int, [c_int, c_int, c_size, _PtrP]),
    "dctn_ar_export": (c_int, [c_void, c_void]),
    "dctn_ar_connect": (c_int, [c_void, c_void]),
    "dctn_ar_allreduce": (c_int, [c_void, c_void, c_i64, c_int, c_int, c_void]),
    "dctn_ar_allreduce_algo": (c_int, [c_void, c_void, c_i64, c_int, c_int, c_int, c_void]),
    "dctn_ar_status": (c_int, [c_void]),
    "dctn_ar_destroy": (c_int, [c_void]),
}

_lib: Optional[ctypes.CDLL] = None


def build(verbose: bool = False) -> str:
    """Compile dctn_amd/csrc/*.hip for gfx950 into dctn_amd/libdctn_amd.so (hipcc cross-compiles
    without a GPU)."""
    res = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise RuntimeError("building libdctn_amd.so failed")
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP kernel library is the only implementation of the "
                "contraction path (no fallback).  Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C dctn_amd/csrc`."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export it
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def last_kernel() -> str:
    return lib().dctn_last_kernel().decode()


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPE_CODE[t.dtype]
    except KeyError:
        raise TypeError(f"dctn_amd supports float32, float64 and bfloat16 tensors, got {t.dtype}") from None


def require_device(*tensors: torch.Tensor) -> torch.device:
    """Device of a kernel-library call: every tensor must already live on the same MI355X device (the
    autograd Functions are only ever reached through `on_device`, which stages CPU tensors first)."""
    dev = tensors[0].device
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                "dctn_amd: the contraction path runs on an MI355X device only; got a tensor on "
                f"'{t.device}'.  There is deliberately no CPU fallback."
            )
        if t.device != dev:
            raise RuntimeError(f"dctn_amd: tensors on different devices ({dev} vs {t.device})")
    return dev


def placement(*tensors: torch.Tensor):
    """(device to compute on, staged?).  "Device follows the tensors" (SURVEY 8b): tensors that all live on
    one GPU compute there; tensors that all live on the CPU - how the reference's own tests build them
    (tests/test_eps.py:10-13) - are STAGED to the current GPU, run through the same HIP kernels and the
    results come back as CPU tensors of the same dtype.  Without a GPU there is nothing to stage to, and
    no CPU implementation to fall back on: RuntimeError.  Mixed placements raise like torch does."""
    devs = {t.device for t in tensors}
    if len(devs) > 1:
        raise RuntimeError(f"dctn_amd: expected all tensors on one device, got {sorted(str(d) for d in devs)}")
    dev = tensors[0].device
    if dev.type == "cuda":
        return dev, False
    if dev.type != "cpu":
        raise RuntimeError(f"dctn_amd: unsupported device '{dev}'")
    if not torch.cuda.is_available():
        raise RuntimeError(
            "dctn_amd: CPU tensors are staged to an MI355X for the contraction, but no GPU is visible. "
            "There is deliberately no CPU implementation of this path."
        )
    return torch.device("cuda", torch.cuda.current_device()), True


def on_device(fn, *tensors: torch.Tensor):
    """``fn(*tensors)`` with CPU tensors staged to the GPU and the result(s) brought back (see `placement`).
    The copies are ordinary differentiable ``.to()`` calls, so gradients arrive on the CPU leaves."""
    dev, staged = placement(*tensors)
    if not staged:
        return fn(*tensors)
    out = fn(*(t.to(dev) for t in tensors))
    if isinstance(out, torch.Tensor):
        return out.cpu()
    return type(out)(o.cpu() for o in out)


def check(rc: int, what: str) -> int:
    """Raises for the negative DCTN_ERR_* codes; returns the (non-negative) success code: DCTN_OK, DCTN_SAVED, DCTN_PARTIAL."""
    if rc >= 0:
        return rc
    msg = lib().dctn_strerror(rc).decode()
    if rc == ERR_BAD_SHAPE:
        raise AssertionError(f"{what}: {msg}")  # the reference signals shape errors with `assert`
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg} (code {rc})")


def stream_ptr(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def strides5(t: torch.Tensor):
    return (c_i64 * 5)(*t.stride())


def int_array(values: Sequence[int]):
    return (c_int * len(values))(*[int(v) for v in values])


def ptr_array(tensors: Sequence[Optional[torch.Tensor]]):
    return (c_void * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


_ws_cache: dict = {}


def workspace(nbytes: int, dev: torch.device) -> torch.Tensor:
    """Scratch for one kernel-library call.  Small requests come from the caching allocator; large
    ones (ConvSBS backward states: tens of MB) reuse one grow-only buffer per (device, stream) —
    calls on a stream are ordered, so consecutive calls can share it — because re-requesting big
    blocks every call makes the allocator release and re-map them.  Under graph capture the
    allocation must come from the graph's pool, so the cache is bypassed."""
    nbytes = max(int(nbytes), 256)
    if nbytes < (8 << 20) or torch.cuda.is_current_stream_capturing():
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _ws_cache[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return buf


# precision policy for float32 tensors on the MFMA paths (bf16 tensors always use bf16 MFMA)
_precision = PREC_EXACT


def set_float32_matmul_precision(mode: str) -> None:
    """'exact' (default): f32 in / f32 accumulate.  'bf16': operands rounded to bf16, f32 accumulate."""
    global _precision
    _precision = {"exact": PREC_EXACT, "highest": PREC_EXACT, "bf16": PREC_BF16}[mode]


_options = 0


def precision() -> int:
    """The `policy` argument of the EPS entry points: precision | option flags (include/dctn_amd.h)."""
    return _precision | _options


class options:
    """``with L.options(L.OPT_SMALL_CHUNKS): ...`` - DCTN_OPT_* flags OR-ed into the policy of every EPS call inside
    the block (tests and measurement tools; the flags travel to the library as an explicit argument)."""

    def __init__(self, flags: int):
        self.flags = int(flags)

    def __enter__(self):
        global _options
        self._saved = _options
        _options |= self.flags
        return self

    def __exit__(self, *exc):
        global _options
        _options = self._saved
        return False
