"""dctn_amd — MI355X-native implementation of dctn's windowed tensor-network contraction path
(EPS / ConvSBS / logmatmulexp) behind the reference's own Python surface.

Module names mirror the reference package ``dctn`` (``dctn_amd.eps`` <-> ``dctn.eps`` ...); the
top-level package ``dctn`` in this repository aliases them so the reference's runner and tests
import unchanged.
"""
from ._lib import build, last_kernel, set_float32_matmul_precision  # noqa: F401

__all__ = ["build", "last_kernel", "set_float32_matmul_precision"]
