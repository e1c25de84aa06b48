"""Losses of the PreSight model (camera-only configs): ns/models/PreSight/nerfacto_nusc_ms.py:558-645,
ns/model_components/PreSight/losses.py:106-206, ns/model_components/losses.py:130-149.

Per-ray O(S^2) losses (distortion, z-anti-aliased interlevel) run in fused HIP kernels (csrc/losses.hip) that return the
loss and its gradient w.r.t. the weights in one pass; the per-ray scalar losses (rgb / sky / semantic) are [R,<=64]
element-wise reductions."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch import Tensor

from . import ops
from ._lib import check, lib
from .ops import _f32, _p, _stream


def _finish(terms: Tensor, denom: float, scale: float, keep: Optional[Tensor] = None, want_inv: bool = False):
    """scale * sum(terms) / denom (or / sum(keep)) as a 0-dim tensor in ONE launch (+ the device scalar scale / denominator)"""
    out = torch.empty(2 if want_inv else 1, device=terms.device)
    inv = out[1:] if want_inv else None
    check(lib().ps_loss_finish(_p(terms), terms.numel(), _p(keep), 0 if keep is None else keep.numel(), float(denom), float(scale),
                               _p(out), _p(inv), _stream()), "ps_loss_finish")
    return out[0], inv


def _chain(grad: Tensor, g: Tensor, host_scale: float, factor: Optional[Tensor] = None) -> Tensor:
    """grad * g * host_scale (* factor): the backward of a scalar loss term in one launch; g is autograd's device scalar"""
    g = g if (g.dtype == torch.float32 and g.is_cuda) else g.to(device=grad.device, dtype=torch.float32)
    out = torch.empty_like(grad)
    check(lib().ps_scale_grad(_p(grad), grad.numel(), _p(g), _p(factor), float(host_scale), _p(out), _stream()), "ps_scale_grad")
    return out


class _ScalarLoss(torch.autograd.Function):
    """mean-reduced element-wise loss: one launch for the terms + gradient of the mean, one for the (scaled) value"""

    @staticmethod
    def forward(ctx, pred, other, kind, arg, scale):
        pred, other = _f32(pred), _f32(other)
        if pred.shape != other.shape:
            other = other.expand_as(pred).contiguous() if other.numel() != pred.numel() else other.reshape(pred.shape)
        n = pred.numel()
        partial = torch.empty(lib().ps_loss_partials(n), device=pred.device)
        grad = torch.empty_like(pred)
        if kind == "mse":
            check(lib().ps_mse_loss(_p(pred), _p(other), n, int(arg), _p(partial), _p(grad), _stream()), "ps_mse_loss")
        else:
            check(lib().ps_sky_bce_loss(_p(pred), _p(other), n, float(arg), _p(partial), _p(grad), _stream()), "ps_sky_bce_loss")
        ctx.save_for_backward(grad)
        ctx.scale = float(scale)
        return _finish(partial, n, scale)[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return _chain(grad, g, ctx.scale), None, None, None, None


# Every loss takes `scale`: the *_loss_mult the reference multiplies the term with afterwards (nerfacto_nusc_ms.py:558-645);
# folded into the value / gradient launches it costs nothing, as a separate scalar multiply it is two more launches per term.
def mse_loss(target: Tensor, pred: Tensor, scale: float = 1.0) -> Tensor:
    """nn.MSELoss()(target, pred) of the reference (nerfacto_nusc_ms.py:568); the gradient flows to `pred` only."""
    return _ScalarLoss.apply(pred, target.detach(), "mse", 0, scale)


class MSELoss(torch.nn.Module):
    def forward(self, a: Tensor, b: Tensor, scale: float = 1.0) -> Tensor:
        # the reference calls rgb_loss(gt, pred); keep nn.MSELoss's symmetry by differentiating whichever side needs it
        if a.requires_grad and not b.requires_grad:
            return _ScalarLoss.apply(a, b.detach(), "mse", 0, scale)
        return _ScalarLoss.apply(b, a.detach(), "mse", 0, scale)


def sky_loss(accumulation: Tensor, sky_mask: Tensor, eps: float = 1e-7, scale: float = 1.0) -> Tensor:
    """ns/model_components/PreSight/losses.py:106-115: BCE(clip(acc, eps, 1-eps), 1 - sky_mask), mean"""
    return _ScalarLoss.apply(accumulation, sky_mask.detach(), "bce", eps, scale)


def semantic_loss(pred: Tensor, target: Tensor, clip: bool = True, scale: float = 1.0) -> Tensor:
    """ns/model_components/PreSight/losses.py:117-125: MSE against the (clipped) feature target, mean"""
    return _ScalarLoss.apply(pred, target.detach(), "mse", int(clip), scale)


class _Distortion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sbins, w, scale):
        sbins, w = _f32(sbins), _f32(w)
        R, S = w.shape
        per_ray = torch.empty(R, device=w.device)
        dw = torch.empty_like(w)
        check(lib().ps_distortion_loss(_p(sbins), _p(w), R, S, _p(per_ray), _p(dw), _stream()), "ps_distortion_loss")
        ctx.save_for_backward(dw)
        ctx.k = float(scale) / R
        return _finish(per_ray, R, scale)[0]

    @staticmethod
    def backward(ctx, g):
        (dw,) = ctx.saved_tensors
        return None, _chain(dw, g, ctx.k), None


def distortion_loss(weights_list: Sequence[Tensor], ray_samples_list, scale: float = 1.0) -> Tensor:
    """mip-NeRF 360 distortion of the final level (ns/model_components/losses.py:130-149)."""
    w = weights_list[-1]
    w = w.reshape(w.shape[0], w.shape[1]) if w.dim() == 3 else w  # a view (x[..., 0] costs a zero-fill + copy in backward)
    return _Distortion.apply(ray_samples_list[-1].sbins, w, scale)


class _Interlevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c, w, cp, wp, pulse_width, scale):
        c, w, cp, wp = _f32(c), _f32(w), _f32(cp), _f32(wp)
        R, S = w.shape
        Sp = wp.shape[1]
        per_ray = torch.empty(R, device=w.device)
        dwp = torch.empty_like(wp)
        check(lib().ps_interlevel_loss(_p(c), _p(w), _p(cp), _p(wp), R, S, Sp, float(pulse_width), _p(per_ray), _p(dwp), _stream()),
              "ps_interlevel_loss")
        ctx.save_for_backward(dwp)
        ctx.k = float(scale) / (R * Sp)
        return _finish(per_ray, R * Sp, scale)[0]

    @staticmethod
    def backward(ctx, g):
        (dwp,) = ctx.saved_tensors
        return None, None, None, _chain(dwp, g, ctx.k), None, None


def z_anti_aliasing_interlevel_loss(weights_list: Sequence[Tensor], ray_samples_list, pulse_width: Sequence[float],
                                    scale: float = 1.0) -> Tensor:
    """Zip-NeRF anti-aliased proposal loss (ns/model_components/PreSight/losses.py:166-206); the main-level histogram
    is detached, gradients flow to the proposal weights only."""
    c = ray_samples_list[-1].sbins.detach()
    w = weights_list[-1].detach()
    w = w.reshape(w.shape[0], w.shape[1]) if w.dim() == 3 else w  # a view (x[..., 0] costs a zero-fill + copy in backward)
    total = None
    for i, (rs, wp) in enumerate(zip(ray_samples_list[:-1], weights_list[:-1])):
        wp = wp.reshape(wp.shape[0], wp.shape[1]) if wp.dim() == 3 else wp
        term = _Interlevel.apply(c, w, rs.sbins, wp, pulse_width[i], scale)
        total = term if total is None else total + term
    return total if total is not None else torch.zeros((), device=w.device)


class _LineOfSight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, ebins, depth, sky, sigma, upper_bound, pose_scale, scale):
        w, ebins, depth = _f32(w), _f32(ebins), _f32(depth)
        sky = None if sky is None else _f32(sky)
        R, S = w.shape
        per_ray, keep, dw = torch.empty(R, device=w.device), torch.empty(R, device=w.device), torch.empty_like(w)
        check(lib().ps_line_of_sight_loss(_p(w), _p(ebins), _p(depth), _p(sky), R, S, float(sigma), float(upper_bound),
                                          float(pose_scale), _p(per_ray), _p(dw), _p(keep), _stream()), "ps_line_of_sight_loss")
        value, inv = _finish(per_ray, 0.0, scale, keep=keep, want_inv=True)  # NaN when no ray qualifies, like torch.mean of an empty selection
        ctx.save_for_backward(dw, inv)
        return value

    @staticmethod
    def backward(ctx, g):
        dw, inv = ctx.saved_tensors
        return _chain(dw, g, 1.0, inv), None, None, None, None, None, None, None


def line_of_sight_loss(weights: Tensor, termination_depth: Tensor, ray_samples, sigma: float, sky_mask: Optional[Tensor] = None,
                       upper_bound: float = 75.0, pose_scale_factor: float = 1.0, scale: float = 1.0) -> Tensor:
    """URF line-of-sight loss (ns/model_components/PreSight/losses.py:28-65).  Takes the RaySamples (bin edges in scene
    units) + pose_scale_factor instead of the pre-divided `steps` tensor: the midpoints are formed inside the kernel."""
    w = weights.reshape(weights.shape[0], weights.shape[1]) if weights.dim() == 3 else weights
    sky = None if sky_mask is None else sky_mask.reshape(-1)
    return _LineOfSight.apply(w, ray_samples.ebins, termination_depth.reshape(-1), sky, sigma, upper_bound, pose_scale_factor, scale)


class _ExpectedDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, pred, sky, upper_bound, inverse, pose_scale, scale):
        depth, pred = _f32(depth), _f32(pred)
        sky = None if sky is None else _f32(sky)
        R = depth.shape[0]
        per_ray, keep, dpred = (torch.empty(R, device=depth.device) for _ in range(3))
        check(lib().ps_expected_depth_loss(_p(depth), _p(pred), _p(sky), R, float(upper_bound), int(bool(inverse)), float(pose_scale),
                                           _p(per_ray), _p(dpred), _p(keep), _stream()), "ps_expected_depth_loss")
        value, inv = _finish(per_ray, 0.0, scale, keep=keep, want_inv=True)
        ctx.save_for_backward(dpred, inv)
        return value

    @staticmethod
    def backward(ctx, g):
        dpred, inv = ctx.saved_tensors
        return None, _chain(dpred, g, 1.0, inv), None, None, None, None, None


def expected_depth_loss(termination_depth: Tensor, predicted_depth: Tensor, upper_bound: float = 75.0,
                        pose_scale_factor: float = 1.0, scale: float = 1.0) -> Tensor:
    """ns/model_components/PreSight/losses.py:67-81; predicted_depth in scene units (divided by pose_scale_factor inside)."""
    return _ExpectedDepth.apply(termination_depth.reshape(-1), predicted_depth.reshape(-1), None, upper_bound, False, pose_scale_factor,
                                scale)


def expected_monodepth_loss(termination_depth: Tensor, predicted_depth: Tensor, sky_mask: Tensor, upper_bound: float = 50.0,
                            inverse: bool = False, pose_scale_factor: float = 1.0, scale: float = 1.0) -> Tensor:
    """ns/model_components/PreSight/losses.py:83-103."""
    return _ExpectedDepth.apply(termination_depth.reshape(-1), predicted_depth.reshape(-1), sky_mask.reshape(-1), upper_bound, inverse,
                                pose_scale_factor, scale)
