"""Losses of the PreSight model (camera-only configs): ns/models/PreSight/nerfacto_nusc_ms.py:558-645,
ns/model_components/PreSight/losses.py:106-206, ns/model_components/losses.py:130-149.

Per-ray O(S^2) losses (distortion, z-anti-aliased interlevel) run in fused HIP kernels (csrc/losses.hip) that return the
loss and its gradient w.r.t. the weights in one pass; the per-ray scalar losses (rgb / sky / semantic) are [R,<=64]
element-wise reductions."""
from __future__ import annotations

import contextlib
import ctypes
import threading
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


# ---- the seed of the backward pass, known before the forward runs --------------------------------------------------------------
# Every loss term is LINEAR in the gradient autograd hands to its node, and in a training step that gradient is the trainer's loss
# scale (ns/engine/trainer.py:481: grad_scaler.scale(loss).backward(), a constant device scalar).  The trainer registers that scalar
# here before the forward pass; the fused loss kernels then store their gradients already multiplied with it, and a loss node whose
# backward receives exactly that tensor (same storage: torch's sum / add nodes hand the incoming gradient through as views) returns
# the stored gradient without launching anything.  Any other gradient (a user's own backward call, a different scale) takes the
# general path: one ps_scale_grad launch per stored array with g / seed.
_SEED_HINT: Optional[tuple] = None  # (data_ptr of the seed tensor, its value as python float, the tensor itself -- kept alive)


def set_seed_hint(seed: Optional[Tensor], value: Optional[float] = None) -> None:
    """seed: the 0-dim device tensor the caller will pass to `loss.backward(gradient=seed)`, value: its value; None clears the hint"""
    global _SEED_HINT
    _SEED_HINT = None if seed is None else (seed.data_ptr(), float(value), seed)


def _seed() -> tuple:
    """(pointer, value) of the registered seed, (None, 1.0) without one"""
    return (None, 1.0) if _SEED_HINT is None else _SEED_HINT[:2]


def _is_seed(g: Tensor, ptr) -> bool:
    return ptr is not None and g is not None and g.numel() == 1 and g.is_cuda and g.dtype == torch.float32 and g.data_ptr() == ptr


def _unseed(grad: Tensor, g: Tensor, seed_value: float) -> Tensor:
    """general path: the stored gradient carries `seed_value`, the incoming gradient is g"""
    return _chain(grad, g, 1.0 / seed_value)


def _f32mul(a: float, b: float) -> float:
    """a * b rounded as the device's fp32 product of the fp32 values (what ps_scale_grad formed as g[0] * host_scale)"""
    import numpy as np

    return float(np.float32(a) * np.float32(b))


_TICKETS: dict = {}


def _ticket(device) -> Tensor:
    """one zero-initialised uint32 per device: the "last workgroup adds the outputs" counter of ps_finish_losses (the launch resets it)"""
    t = _TICKETS.get(str(device))
    if t is None:
        t = torch.zeros(1, device=device, dtype=torch.int32)
        _TICKETS[str(device)] = t
    return t


# ---- the scalar values: one launch behind all loss kernels -----------------------------------------------------------------------
# A loss kernel leaves per-ray (or per-workgroup) terms; its value scale * sum / count used to be one ps_loss_finish launch per term
# (seven per training step) plus a concat + reduction for their sum.  Inside `with deferred_finish() as fin:` the loss nodes only
# RECORD their reductions; `fin.flush(order)` (called by the model at the end of get_loss_dict) then forms every value and the sum of
# the terms in ONE launch (ps_finish_losses: the bits of ps_loss_finish per term).  Outside such a block a loss finishes at once.
_DEFER = threading.local()


class _Deferred:
    def __init__(self):
        self.pending: List[tuple] = []  # (out [1] tensor, [(terms tensor, n, denom, scale), ...])
        self.total: Optional[Tensor] = None
        self.order_ptrs: Optional[tuple] = None

    def flush(self, order: Optional[Sequence[Tensor]] = None) -> Optional[Tensor]:
        """finish every recorded reduction; with `order` (the loss terms in the order of their sum) also -> their sum as a 0-dim tensor:
        ((t0 + t1) + t2) + ..., where terms that are not pending here (finished elsewhere) enter by value"""
        pend, self.pending = self.pending, []
        by_ptr = {o.data_ptr(): (o, descs) for o, descs in pend}
        groups, seen = [], set()
        for t in (order or ()):
            ent = by_ptr.get(t.data_ptr())
            if ent is not None and t.data_ptr() not in seen:
                groups.append(ent)
                seen.add(t.data_ptr())
            else:
                groups.append((None, [(t.detach().reshape(1), 1, 1.0, 1.0)]))  # a finished scalar: 1 * (t / 1)
        rest = [e for ptr, e in by_ptr.items() if ptr not in seen]
        total = None
        if groups:
            total = _finish_launch(groups, want_total=True)
        if rest:
            _finish_launch(rest, want_total=False)
        if order is not None:
            self.total, self.order_ptrs = total, tuple(t.data_ptr() for t in order)
        return total


def _finish_launch(groups: List[tuple], want_total: bool) -> Optional[Tensor]:
    """groups: [(out [1] tensor | None, descriptors)]; None: a scratch output (the value only enters the total)"""
    dev = groups[0][1][0][0].device
    n_desc = sum(len(d) for _, d in groups)
    if n_desc > 16 or len(groups) > 16:  # (never in the model's own calls: 5-7 terms)
        half = len(groups) // 2
        a, b = _finish_launch(groups[:half], want_total), _finish_launch(groups[half:], want_total)
        return (a + b) if want_total else None
    scratch = torch.empty(len(groups) + 1, device=dev)
    outs, terms, ns, denoms, scales, out_of, keep = [], [], [], [], [], [], []
    for o, (out, descs) in enumerate(groups):
        outs.append(scratch[o:o + 1].data_ptr() if out is None else out.data_ptr())
        for tt, n, denom, scale in descs:
            tt = _f32(tt)
            keep.append(tt)
            terms.append(tt.data_ptr())
            ns.append(int(n))
            denoms.append(float(denom))
            scales.append(float(scale))
            out_of.append(o)
    total = scratch[len(groups):] if want_total else None
    check(lib().ps_finish_losses((ctypes.c_void_p * n_desc)(*terms), (ctypes.c_int64 * n_desc)(*ns), (ctypes.c_float * n_desc)(*denoms),
                                 (ctypes.c_float * n_desc)(*scales), (ctypes.c_int * n_desc)(*out_of), n_desc,
                                 (ctypes.c_void_p * len(groups))(*outs), len(groups), _p(total), _p(_ticket(dev)), _stream()),
          "ps_finish_losses")
    return None if total is None else total[0]


@contextlib.contextmanager
def deferred_finish():
    prev = getattr(_DEFER, "cur", None)
    cur = _Deferred()
    _DEFER.cur = cur
    try:
        yield cur
    finally:
        _DEFER.cur = prev
        if cur.pending:  # (never leave a value unwritten: a block that raised or forgot to flush finishes here)
            cur.flush()


def _finish_groups(groups: List[tuple]) -> None:
    """groups: [(out [1] tensor, [(terms, n, denom, scale), ...])]: out[0] = sum over its descriptors of scale * (sum(terms[:n]) / denom)
    -- now (one launch for all groups), or at the enclosing deferred_finish block's flush"""
    if not groups:
        return
    cur = getattr(_DEFER, "cur", None)
    if cur is None:
        _finish_launch(list(groups), want_total=False)
    else:
        cur.pending.extend(groups)


def _finish_into(out: Tensor, descs: List[tuple]) -> None:
    _finish_groups([(out, descs)])


class LossDict(dict):
    """the loss dictionary of a training step: a plain dict whose sum has already been formed by the launch that finished the terms
    (`total`, valid while the values are the ones it was formed from)"""
    total: Optional[Tensor] = None
    _total_of: Optional[tuple] = None


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
    """per-ray values + gradient in one launch (ps_distortion_loss_scaled): the stored gradient already carries scale / R and the
    registered seed of the backward pass; the value scale * mean is finished with the step's other terms (deferred_finish)"""

    @staticmethod
    def forward(ctx, sbins, w, scale):
        sbins, w = _f32(sbins), _f32(w)
        R, S = w.shape
        per_ray = torch.empty(R, device=w.device)
        dw = torch.empty_like(w)
        out = torch.empty(1, device=w.device)
        ctx.seed_ptr, ctx.seed_value = _seed()
        check(lib().ps_distortion_loss_scaled(_p(sbins), _p(w), R, S, _p(per_ray), _p(dw), _f32mul(float(scale) / R, ctx.seed_value), _stream()),
              "ps_distortion_loss_scaled")
        _finish_into(out, [(per_ray, R, float(R), float(scale))])
        ctx.save_for_backward(dw)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (dw,) = ctx.saved_tensors
        return None, (dw if _is_seed(g, ctx.seed_ptr) else _unseed(dw, g, ctx.seed_value)), None


def distortion_loss(weights_list: Sequence[Tensor], ray_samples_list, scale: float = 1.0) -> Tensor:
    """mip-NeRF 360 distortion of the final level (ns/model_components/losses.py:130-149)."""
    w = weights_list[-1]
    w = w.reshape(w.shape[0], w.shape[1]) if w.dim() == 3 else w  # a view (x[..., 0] costs a zero-fill + copy in backward)
    return _Distortion.apply(ray_samples_list[-1].sbins, w, scale)


class _Interlevel(torch.autograd.Function):
    """all proposal levels in one node: one launch per level (ps_interlevel_loss_scaled); the value term_0 + term_1 (as the reference's
    python sum forms it) is finished with the step's other terms (deferred_finish)"""

    @staticmethod
    def forward(ctx, c, w, pulse_width, scale, *cp_wp):
        c, w = _f32(c), _f32(w)
        R, S = w.shape
        out = torch.empty(1, device=w.device)
        ctx.seed_ptr, ctx.seed_value = _seed()
        grads, descs = [], []
        for i in range(len(cp_wp) // 2):
            cp, wp = _f32(cp_wp[2 * i]), _f32(cp_wp[2 * i + 1])
            Sp = wp.shape[1]
            per_ray = torch.empty(R, device=w.device)
            dwp = torch.empty_like(wp)
            check(lib().ps_interlevel_loss_scaled(_p(c), _p(w), _p(cp), _p(wp), R, S, Sp, float(pulse_width[i]), _p(per_ray), _p(dwp),
                                                  _f32mul(float(scale) / (R * Sp), ctx.seed_value), _stream()), "ps_interlevel_loss_scaled")
            grads.append(dwp)
            descs.append((per_ray, R, float(R * Sp), float(scale)))
        _finish_into(out, descs)  # term_0 + term_1 in this order
        ctx.save_for_backward(*grads)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        fast = _is_seed(g, ctx.seed_ptr)
        out = [None, None, None, None]
        for dwp in ctx.saved_tensors:
            out += [None, dwp if fast else _unseed(dwp, g, ctx.seed_value)]
        return tuple(out)


def z_anti_aliasing_interlevel_loss(weights_list: Sequence[Tensor], ray_samples_list, pulse_width: Sequence[float],
                                    scale: float = 1.0) -> Tensor:
    """Zip-NeRF anti-aliased proposal loss (ns/model_components/PreSight/losses.py:166-206); the main-level histogram
    is detached, gradients flow to the proposal weights only."""
    c = ray_samples_list[-1].sbins.detach()
    w = weights_list[-1].detach()
    w = w.reshape(w.shape[0], w.shape[1]) if w.dim() == 3 else w  # a view (x[..., 0] costs a zero-fill + copy in backward)
    args = []
    for rs, wp in zip(ray_samples_list[:-1], weights_list[:-1]):
        args += [rs.sbins, wp.reshape(wp.shape[0], wp.shape[1]) if wp.dim() == 3 else wp]
    if not args:
        return torch.zeros((), device=w.device)
    return _Interlevel.apply(c, w, tuple(float(p) for p in pulse_width), scale, *args)


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


# ------------------------------------------------------------------------------------------------ fused blend + per-ray losses
class _BlendLosses(torch.autograd.Function):
    """sky blending (ops.sky_blend) + rgb MSE + sky BCE + semantic MSE as ONE launch (ps_blend_losses): the blended outputs, the three
    finished loss values and -- because the terms are linear in the seed of the backward pass, which the trainer registers before the
    forward (set_seed_hint) -- the gradients w.r.t. every input of the blend.  Backward on the seeded path launches nothing; any other
    incoming gradient re-derives the result from the separate operators (same values, the pre-round-6 launches)."""

    @staticmethod
    def forward(ctx, rgb_f, acc_raw, sem_f, sky_rgb, sky_sem, rgb_t, sky_t, sem_t, mults, bce_eps, clip_sem):
        rgb_f, acc_raw = _f32(rgb_f), _f32(acc_raw)
        R, dev = rgb_f.shape[0], rgb_f.device
        opt = lambda t: None if t is None else _f32(t)  # noqa: E731
        sem_f, sky_rgb, sky_sem = opt(sem_f), opt(sky_rgb), opt(sky_sem)
        if sem_f is None:
            sky_sem, sem_t = None, None
        C = 0 if sem_f is None else sem_f.shape[1]
        fit = lambda t, shape: None if t is None else _f32(t.detach()).reshape(shape)  # noqa: E731
        rgb_t, sky_t, sem_t = fit(rgb_t, (R, 3)), fit(sky_t, (R,)), fit(sem_t, (R, C))
        rgb, acc = torch.empty(R, 3, device=dev), torch.empty(R, 1, device=dev)
        sem = torch.empty(R, C, device=dev) if sem_f is not None else None
        d_rgb, d_acc_raw = torch.empty(R, 3, device=dev), torch.empty_like(acc_raw)
        d_sem = torch.empty(R, C, device=dev) if sem_f is not None else None
        d_sky_rgb = torch.empty_like(sky_rgb) if sky_rgb is not None else None
        d_sky_sem = torch.empty_like(sky_sem) if sky_sem is not None else None
        n_part = lib().ps_blend_losses_partials(R)
        partial = torch.empty(3, n_part, device=dev)
        l_rgb, l_sky, l_sem = (torch.empty(1, device=dev) for _ in range(3))
        ctx.set_materialize_grads(False)  # (outputs nobody differentiates -- the blended rgb / accumulation / semantics -- arrive as None)
        ctx.seed_ptr, ctx.seed_value = _seed()
        m_rgb, m_sky, m_sem = (float(m) for m in mults)
        k = lambda m, n: _f32mul(m / n, ctx.seed_value)  # noqa: E731
        check(lib().ps_blend_losses(_p(rgb_f), _p(acc_raw), _p(sem_f), _p(sky_rgb), _p(sky_sem), _p(rgb_t), _p(sky_t), _p(sem_t), R, C,
                                    int(bool(clip_sem)), float(bce_eps), k(2.0 * m_rgb, 3 * R), k(m_sky, R), k(2.0 * m_sem, R * max(C, 1)),
                                    _p(rgb), _p(acc), _p(sem), _p(d_rgb), _p(d_sem), _p(d_acc_raw), _p(d_sky_rgb), _p(d_sky_sem), _p(partial),
                                    _stream()), "ps_blend_losses")
        # values: mult * (sum of the per-workgroup partial sums) / count, finished with the step's other terms
        groups = [(l_rgb, [(partial[0], n_part, float(3 * R), m_rgb)])] if rgb_t is not None else []
        if sky_t is not None:
            groups.append((l_sky, [(partial[1], n_part, float(R), m_sky)]))
        if sem_t is not None:
            groups.append((l_sem, [(partial[2], n_part, float(R * max(C, 1)), m_sem)]))
        _finish_groups(groups)
        ctx.save_for_backward(d_rgb, d_acc_raw, d_sem, d_sky_rgb, d_sky_sem, rgb_f, acc_raw, sem_f, sky_rgb, sky_sem, rgb_t, sky_t, sem_t)
        ctx.cfg = (mults, bce_eps, clip_sem)
        empty = torch.empty(0, device=dev)
        return l_rgb[0], l_sky[0], l_sem[0], rgb, acc, (sem if sem is not None else empty)

    @staticmethod
    def backward(ctx, g_rgb_l, g_sky_l, g_sem_l, g_rgb, g_acc, g_sem):
        (d_rgb, d_acc_raw, d_sem, d_sky_rgb, d_sky_sem, rgb_f, acc_raw, sem_f, sky_rgb, sky_sem, rgb_t, sky_t, sem_t) = ctx.saved_tensors
        present = [g for g, t in ((g_rgb_l, rgb_t), (g_sky_l, sky_t), (g_sem_l, sem_t)) if t is not None]
        absent = [g for g, t in ((g_rgb_l, rgb_t), (g_sky_l, sky_t), (g_sem_l, sem_t)) if t is None]
        if (g_rgb is None and g_acc is None and g_sem is None and all(g is None for g in absent)
                and all(_is_seed(g, ctx.seed_ptr) for g in present)):
            return d_rgb, d_acc_raw, d_sem, d_sky_rgb, d_sky_sem, None, None, None, None, None, None
        # general path: the separate operators, differentiated by autograd (any mixture of incoming gradients)
        from . import ops

        mults, bce_eps, clip_sem = ctx.cfg
        with torch.enable_grad():
            leaves = [None if t is None else t.detach().requires_grad_(True) for t in (rgb_f, acc_raw, sem_f, sky_rgb, sky_sem)]
            rgb, acc, sem = ops.sky_blend(*leaves)
            outs, gouts = [], []
            for val, g in ((None if rgb_t is None else mse_loss(rgb_t, rgb, mults[0]), g_rgb_l),
                           (None if sky_t is None else sky_loss(acc.view(-1, 1), sky_t.view(-1, 1), bce_eps, mults[1]), g_sky_l),
                           (None if (sem_t is None or sem is None) else semantic_loss(sem, sem_t, clip_sem, mults[2]), g_sem_l),
                           (rgb, g_rgb), (acc, g_acc), (sem, g_sem)):
                if val is not None and g is not None:
                    outs.append(val)
                    gouts.append(g.reshape(val.shape) if g.shape != val.shape else g)
            live = [t for t in leaves if t is not None]
            grads = iter(torch.autograd.grad(outs, live, gouts, allow_unused=True)) if outs else iter([None] * len(live))
        res = [None if t is None else next(grads) for t in leaves]
        return (*res, None, None, None, None, None, None)


def blend_losses(rgb_f: Tensor, acc_raw: Tensor, sem_f: Optional[Tensor], sky_rgb: Optional[Tensor], sky_sem: Optional[Tensor],
                 rgb_target: Optional[Tensor], sky_mask: Optional[Tensor], sem_target: Optional[Tensor], rgb_mult: float = 1.0,
                 sky_mult: float = 1.0, sem_mult: float = 1.0, eps: float = 1e-7, clip: bool = True):
    """-> ((rgb_loss | None, sky_loss | None, semantic_loss | None), (rgb [R,3], accumulation [R,1], semantics [R,C] | None)):
    ops.sky_blend + MSELoss()(gt, rgb) + sky_loss(accumulation, sky_mask) + semantic_loss(semantics, features) of
    ns/models/PreSight/nerfacto_nusc_ms.py:512-533,558-575 in one launch; a term whose target is None is not formed."""
    l0, l1, l2, rgb, acc, sem = _BlendLosses.apply(rgb_f, acc_raw, sem_f, sky_rgb, sky_sem, rgb_target, sky_mask, sem_target,
                                                   (float(rgb_mult), float(sky_mult), float(sem_mult)), float(eps), bool(clip))
    return ((l0 if rgb_target is not None else None, l1 if sky_mask is not None else None,
             l2 if (sem_target is not None and sem_f is not None) else None), (rgb, acc, sem if sem_f is not None else None))


class _Holder:
    def __init__(self, value):
        self.value = value


class _LossSum(torch.autograd.Function):
    """functools.reduce(torch.add, loss_dict.values()) (ns/engine/trainer.py:478).  `total`: the sum when the launch that finished the
    terms has already formed it (LossDict), else None -> one ps_finish_losses launch.  The backward hands the incoming gradient to
    every term AS IS (the same tensor: the terms' nodes recognise the registered seed by its storage)."""

    @staticmethod
    def forward(ctx, total, *terms):
        ctx.n = len(terms)
        if total.value is not None:  # (a holder object, not a tensor argument: the sum is an OUTPUT of this node, never one of its inputs)
            return total.value.detach().reshape(())
        ts = [t if (t.dtype == torch.float32 and t.is_cuda) else t.to(device=terms[0].device, dtype=torch.float32) for t in terms]
        return _finish_launch([(None, [(t.detach().reshape(1), 1, 1.0, 1.0)]) for t in ts], want_total=True).reshape(())

    @staticmethod
    def backward(ctx, g):
        return (None,) + (g,) * ctx.n


def loss_sum(terms) -> Tensor:
    """the sum of 0-dim loss terms, ((t0 + t1) + t2) + ..., as a 0-dim tensor; `terms`: a sequence, or the loss dictionary itself (a
    LossDict carries the sum its finishing launch formed)"""
    total = None
    if isinstance(terms, dict):
        vals = list(terms.values())
        if isinstance(terms, LossDict) and terms.total is not None and terms._total_of == tuple(v.data_ptr() for v in vals):
            total = terms.total
        terms = vals
    terms = list(terms)
    if len(terms) == 1:
        return terms[0]
    if any(t.numel() != 1 or not t.is_cuda for t in terms):
        return torch.stack([t.reshape(()) for t in terms]).sum()
    return _LossSum.apply(_Holder(total), *terms)
