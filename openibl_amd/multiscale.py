"""Multi-scale descriptor extraction (BASELINE.json configs[4]).

EXTENSION: the reference has no multi-scale path (SURVEY.md §8d "Config 5 caveat"), so this module
defines one and the oracle restates the definition (`oracle.descriptor.multiscale_descriptor`):

    for every scale s:  x_s = bilinear_resize(x, (round(H s), round(W s)))   (s = 1: x itself)
                        d_s = extract_cnn_feature(model, x_s)                 (unit-norm rows)
    descriptor = normalize(d_s0 + d_s1 + ...)                                 (scales in order)

with the resize arithmetic of F.interpolate(mode="bilinear", align_corners=False).  The resize,
the per-scale forwards and the fusion all run in the HIP library.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

from . import ops

DEFAULT_SCALES: Tuple[float, ...] = (1.0, 2.0 ** -0.5, 0.5)


def scaled_size(H: int, W: int, s: float) -> Tuple[int, int]:
    """Image size at scale s: rounded, never below one conv5 cell (16 pixels)."""
    return max(16, int(round(H * s))), max(16, int(round(W * s)))


def extract_multiscale(model, x: torch.Tensor, scales: Sequence[float] = DEFAULT_SCALES,
                       vlad: bool = True) -> torch.Tensor:
    """float32 [N][3][H][W] on the GPU -> unit-norm [N][d] multi-scale descriptors."""
    if x.dtype != torch.float32 or x.dim() != 4:
        raise ValueError("extract_multiscale expects a float32 [N][3][H][W] batch")
    if len(scales) == 0:
        raise ValueError("extract_multiscale: no scales")
    H, W = int(x.shape[2]), int(x.shape[3])
    per_scale = []
    for s in scales:
        size = scaled_size(H, W, float(s))
        xs = x if size == (H, W) else ops.resize_bilinear(x, size)
        out = model(xs)
        if isinstance(out, (list, tuple)):
            out = out[1] if vlad else out[0]
        per_scale.append(ops.l2_normalize(out.float().contiguous()))
    return ops.sum_l2_normalize(torch.stack(per_scale))
