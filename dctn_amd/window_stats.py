"""Feature map of the input images and the window statistics that scale it — the step just before
the EPS path.

Reference: ``φ_cos_sin_squared_1`` (dctn/dataset_loading.py:33-36), ``calc_scaling_factor``
(dctn/dataset_loading.py:79-94), which materialises all K x K windows of 10 880 samples
(``make_windows``: K*K stacked copies of the data) to read two numbers off them.  Here the two sums
come from one HIP kernel over the images as they are (``dctn_window_stats``); the formulas that
turn them into a mean, a variance and the scaling factor are the reference's.
"""
from __future__ import annotations

from math import pi
from typing import Tuple, Union

import torch
from torch import Tensor

from . import _lib as L

# (x -> 2 sin^2(pi x / 2), x -> 2 cos^2(pi x / 2)): the two features of a pixel intensity in [0, 1]
φ_cos_sin_squared_1 = (
    lambda X: 2 * (X * pi / 2.0).sin() ** 2,
    lambda X: 2 * (X * pi / 2.0).cos() ** 2,
)
phi_cos_sin_squared_1 = φ_cos_sin_squared_1


def apply_feature_map(images: Tensor, φ=φ_cos_sin_squared_1) -> Tensor:
    """(samples, height, width) intensities -> (1, samples, height, width, len(φ)), the layout of
    ``MNISTLikeQuantumIndexedDataset.x`` (dataset_loading.py:63-64)."""
    return torch.stack(tuple(f(images) for f in φ), dim=3).unsqueeze(0)


def window_sums(x: Tensor, kernel_size: int) -> Tensor:
    """``x``: (channels, batch, height, width, in_size) on the device.  Returns a float64 tensor
    [sum_w sum(T_w), sum_w ||T_w||^2] over the rank-one tensors T_w of all K x K windows."""
    return L.on_device(lambda x_: _window_sums_on_device(x_, kernel_size), x)


def _window_sums_on_device(x: Tensor, kernel_size: int) -> Tensor:
    dev = L.require_device(x)
    C, B, H, W, Q = x.shape
    assert H >= kernel_size and W >= kernel_size
    sums = torch.empty(2, dtype=torch.float64, device=dev)
    L.check(
        L.lib().dctn_window_stats(x.data_ptr(), L.strides5(x), sums.data_ptr(), C, B, H, W, Q, kernel_size,
                                  L.dtype_code(x), L.stream_ptr(dev)),
        "window statistics",
    )
    return sums


def window_mean_var(x: Tensor, kernel_size: int, unbiased: bool = True) -> Tuple[Tensor, Tensor]:
    """Mean and variance over every element of every window's rank-one tensor: what
    ``make_windows(x, K).mean_over_batch()`` / ``.var_over_batch()`` return in the reference."""
    C, B, H, W, Q = x.shape
    ntensors = B * (H - kernel_size + 1) * (W - kernel_size + 1)
    n = ntensors * float(Q) ** (kernel_size * kernel_size * C)
    total, sq = window_sums(x, kernel_size).unbind(0)
    mean = total / n
    divisor = n - 1 if unbiased else n
    var = sq / divisor - 2 * total / divisor * mean + n / divisor * mean**2
    return mean, var


def calc_scaling_factor(ds: Union[Tensor, object], kernel_size: int, device=None) -> float:
    """The number the data set's ``x`` must be multiplied by so that its K x K windows, as rank-one
    tensors, have mean^2 + variance == 1.  ``ds``: the (1, samples, h, w, φ) tensor or an object with
    such an ``x`` attribute; the first 10 880 samples are used, in float64, as in the reference."""
    x = ds if isinstance(ds, Tensor) else ds.x
    x = x[:, :10880]
    if device is not None:
        x = x.to(device)
    mean, var = window_mean_var(x.double(), kernel_size)
    return float((mean**2 + var) ** (-1 / (2 * kernel_size**2)))


# ------------------------------------------------------------------------------------------------ phi on the device
def _raw_images_on_device(images: Tensor, device) -> Tensor:
    dev = torch.device(device) if device is not None else (images.device if images.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    return images.to(dev, torch.float32).contiguous()


def calc_scaling_factor_from_images(images: Tensor, kernel_size: int, device=None) -> float:
    """`calc_scaling_factor` (dctn/dataset_loading.py:79-94) from the RAW (samples, height, width) intensities: the
    feature map is applied inside the statistics kernel (`dctn_phi_window_stats`), so neither the expanded
    (1, samples, h, w, 2) tensor nor the K*K stacked window copies are ever built.  First 10 880 samples, like there."""
    img = _raw_images_on_device(images[:10880], device)
    B, H, W = img.shape
    sums = torch.empty(2, dtype=torch.float64, device=img.device)
    rc = L.lib().dctn_phi_window_stats(img.data_ptr(), sums.data_ptr(), B, H, W, kernel_size, L.stream_ptr(img.device))
    if rc == L.ERR_UNSUPPORTED:   # image too large for the per-pixel table in LDS: expand on the device, then the general kernel
        return calc_scaling_factor(apply_feature_map_on_device(img, 1.0, torch.float64), kernel_size)
    L.check(rc, "window statistics of the feature map")
    n = B * (H - kernel_size + 1) * (W - kernel_size + 1) * 2.0 ** (kernel_size * kernel_size)
    total, sq = sums.unbind(0)
    mean = total / n
    var = sq / (n - 1) - 2 * total / (n - 1) * mean + n / (n - 1) * mean**2
    return float((mean**2 + var) ** (-1 / (2 * kernel_size**2)))


def apply_feature_map_on_device(images: Tensor, scale: float = 1.0, dtype: torch.dtype = torch.float32, device=None) -> Tensor:
    """(samples, height, width) intensities -> scale * phi, shape (1, samples, height, width, 2), written once on the
    device in ``dtype`` (`dctn_phi_expand`): the data-set tensor of dataset_loading.py:63 with the autoscale factor of
    the runner folded in, without a float32 intermediate."""
    img = _raw_images_on_device(images, device)
    B, H, W = img.shape
    x = torch.empty((1, B, H, W, 2), dtype=dtype, device=img.device)
    L.check(L.lib().dctn_phi_expand(img.data_ptr(), x.data_ptr(), B * H * W, float(scale), L.dtype_code(x), L.stream_ptr(img.device)),
            "feature map")
    return x
