"""Losses of the PreSight model (camera-only configs): ns/models/PreSight/nerfacto_nusc_ms.py:558-645,
ns/model_components/PreSight/losses.py:106-206, ns/model_components/losses.py:130-149.

Per-ray O(S^2) losses (distortion, z-anti-aliased interlevel) run in fused HIP kernels (csrc/losses.hip) that return the
loss and its gradient w.r.t. the weights in one pass; the per-ray scalar losses (rgb / sky / semantic) are [R,<=64]
element-wise reductions."""
from __future__ import annotations

from typing import List, Sequence

import torch
from torch import Tensor

from . import ops
from ._lib import check, lib
from .ops import _f32, _p, _stream


def sky_loss(accumulation: Tensor, sky_mask: Tensor, eps: float = 1e-7) -> Tensor:
    target = 1.0 - sky_mask
    a = torch.clip(accumulation, min=eps, max=1 - eps)
    return torch.nn.functional.binary_cross_entropy(a, target, reduction="none").mean()


def semantic_loss(pred: Tensor, target: Tensor, clip: bool = True) -> Tensor:
    if clip:
        target = torch.clip(target, min=0.0, max=1.0)
    return torch.nn.functional.mse_loss(pred, target, reduction="none").mean()


class _Distortion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sbins, w):
        sbins, w = _f32(sbins), _f32(w)
        R, S = w.shape
        per_ray = torch.empty(R, device=w.device)
        dw = torch.empty_like(w)
        check(lib().ps_distortion_loss(_p(sbins), _p(w), R, S, _p(per_ray), _p(dw), _stream()), "ps_distortion_loss")
        ctx.save_for_backward(dw)
        ctx.R = R
        return per_ray.mean()

    @staticmethod
    def backward(ctx, g):
        (dw,) = ctx.saved_tensors
        return None, dw * (g / ctx.R)


def distortion_loss(weights_list: Sequence[Tensor], ray_samples_list) -> Tensor:
    """mip-NeRF 360 distortion of the final level (ns/model_components/losses.py:130-149)."""
    w = weights_list[-1]
    w = w[..., 0] if w.dim() == 3 else w
    return _Distortion.apply(ray_samples_list[-1].sbins, w)


class _Interlevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c, w, cp, wp, pulse_width):
        c, w, cp, wp = _f32(c), _f32(w), _f32(cp), _f32(wp)
        R, S = w.shape
        Sp = wp.shape[1]
        per_ray = torch.empty(R, device=w.device)
        dwp = torch.empty_like(wp)
        check(lib().ps_interlevel_loss(_p(c), _p(w), _p(cp), _p(wp), R, S, Sp, float(pulse_width), _p(per_ray), _p(dwp), _stream()),
              "ps_interlevel_loss")
        ctx.save_for_backward(dwp)
        ctx.n = R * Sp
        return per_ray.sum() / ctx.n

    @staticmethod
    def backward(ctx, g):
        (dwp,) = ctx.saved_tensors
        return None, None, None, dwp * (g / ctx.n), None


def z_anti_aliasing_interlevel_loss(weights_list: Sequence[Tensor], ray_samples_list, pulse_width: Sequence[float]) -> Tensor:
    """Zip-NeRF anti-aliased proposal loss (ns/model_components/PreSight/losses.py:166-206); the main-level histogram
    is detached, gradients flow to the proposal weights only."""
    c = ray_samples_list[-1].sbins.detach()
    w = weights_list[-1].detach()
    w = w[..., 0] if w.dim() == 3 else w
    total = 0.0
    for i, (rs, wp) in enumerate(zip(ray_samples_list[:-1], weights_list[:-1])):
        wp = wp[..., 0] if wp.dim() == 3 else wp
        total = total + _Interlevel.apply(c, w, rs.sbins, wp, pulse_width[i])
    return total
