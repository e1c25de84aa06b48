"""CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch CPU restatement (torch-CPU fp32 tensors + int64 index maths) of the
reference's ray-sampling -> multires hash-grid -> tiny-MLP -> volumetric-rendering
path, i.e. of what PreSight's vendored nerfstudio executes when tinycudann is absent
(`implementation="torch"`).  It is the checker for the HIP kernels in
`presight_amd/csrc/`.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this file; the product path never does.

Parity status: PINNED.  Every function below is checked against fixtures produced by
importing the reference itself in the build container (tests/golden/make_golden.py
-> tests/golden/*.npz; tests/test_oracle_golden.py), plus the known-answer vectors
of SURVEY.md Appendix A.

Citations are `file:line` into /root/reference/nerfstudio-0.3.3/nerfstudio (`ns/`).
Everything is written as plain functions over tensors and a flat parameter dict whose
keys are the reference's own state-dict names (e.g.
`field.fields.0.mlp_base_grid.hash_table`), so the same parameter set can be pushed
into the reference, the oracle and the HIP modules.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

PRIME_Y = 2654435761  # ns/field_components/encodings.py:336
PRIME_Z = 805459861


# --------------------------------------------------------------------------------------
# a7  multiresolution hash grid (torch fallback semantics)
# --------------------------------------------------------------------------------------
def hash_scalings(num_levels: int, min_res: int, max_res: int) -> Tensor:
    """Per-level integer resolutions as float32.  ns/field_components/encodings.py:281-284.

    The growth factor is a NumPy float64 scalar, but `g ** arange` is evaluated by torch
    in float32, which is what decides e.g. 2047 vs 2048 for the last level; we therefore
    use the same two-step evaluation instead of a closed form.
    """
    lv = torch.arange(num_levels)
    g = np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)) if num_levels > 1 else 1
    return torch.floor(min_res * g**lv).to(torch.float32)


def hash_index(ix: Tensor, iy: Tensor, iz: Tensor, level_offset: Tensor, table_size: int) -> Tensor:
    """Instant-NGP spatial hash in int64.  ns/field_components/encodings.py:324-341."""
    h = ix.to(torch.int64) ^ (iy.to(torch.int64) * PRIME_Y) ^ (iz.to(torch.int64) * PRIME_Z)
    return h % table_size + level_offset


# corner order used by the reference (x, y, z each 'c'eil or 'f'loor), encodings.py:354-361
_CORNERS = ("ccc", "cfc", "ffc", "fcc", "ccf", "cff", "fff", "fcf")


def hash_encode(x: Tensor, table: Tensor, scalings: Tensor, log2_T: int, return_indices: bool = False):
    """x [N,3] -> [N, L*F].  ns/field_components/encodings.py:343-384.

    ceil/floor corners, no +0.5 offset, every level hashed, trilinear weights applied in
    the order x, y, z with weight `offset` on the ceil corner (so an exactly-integer
    coordinate puts all weight on that corner).
    """
    assert x.shape[-1] == 3
    T = 1 << log2_T
    L = scalings.numel()
    scaled = x[:, None, :] * scalings.view(1, L, 1)  # [N,L,3]
    c = torch.ceil(scaled).to(torch.int32)
    f = torch.floor(scaled).to(torch.int32)
    o = scaled - f
    off = (torch.arange(L, dtype=torch.int64) * T).view(1, L)
    pick = {"c": c, "f": f}
    idx = [hash_index(pick[k[0]][..., 0], pick[k[1]][..., 1], pick[k[2]][..., 2], off, T) for k in _CORNERS]
    v = [table[i] for i in idx]  # each [N,L,F]
    ox, oy, oz = o[..., 0:1], o[..., 1:2], o[..., 2:3]
    f03 = v[0] * ox + v[3] * (1 - ox)
    f12 = v[1] * ox + v[2] * (1 - ox)
    f56 = v[5] * ox + v[6] * (1 - ox)
    f47 = v[4] * ox + v[7] * (1 - ox)
    f0312 = f03 * oy + f12 * (1 - oy)
    f4756 = f47 * oy + f56 * (1 - oy)
    out = (f0312 * oz + f4756 * (1 - oz)).flatten(-2)
    if return_indices:
        return out, torch.stack(idx, dim=-1)  # [N,L,8] int64
    return out


# --------------------------------------------------------------------------------------
# a6  AABB normalisation + L-inf scene contraction + selector
# --------------------------------------------------------------------------------------
def normalize_contract(p: Tensor, aabb: Tensor, contract: bool = True) -> Tuple[Tensor, Tensor]:
    """world p [M,3] -> (u in [0,1]^3 with out-of-range rows zeroed, selector bool [M]).

    ns/fields/PreSight/utils.py:6-10, ns/field_components/spatial_distortions.py:66-69
    (order=inf), ns/fields/PreSight/ingp_field.py:169-177.
    """
    q = (p - aabb[0]) / (aabb[1] - aabb[0])
    if contract:
        q = q * 2 - 1
        mag = q.abs().amax(dim=-1, keepdim=True)  # linalg.norm(ord=inf)
        q = torch.where(mag < 1, q, (2 - (1 / mag)) * (q / mag))
        q = (q + 2.0) / 4.0
    sel = ((q > 0.0) & (q < 1.0)).all(dim=-1)
    return q * sel[:, None], sel


# --------------------------------------------------------------------------------------
# a10  degree-4 real spherical harmonics, evaluated on (d+1)/2 like the torch path does
# --------------------------------------------------------------------------------------
def sh4(v: Tensor) -> Tensor:
    """v [M,3] -> [M,16].  ns/utils/math.py:27-79 (levels=4)."""
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    xx, yy, zz = x * x, y * y, z * z
    out = torch.zeros(v.shape[0], 16, dtype=v.dtype)
    out[:, 0] = 0.28209479177387814
    out[:, 1] = 0.4886025119029199 * y
    out[:, 2] = 0.4886025119029199 * z
    out[:, 3] = 0.4886025119029199 * x
    out[:, 4] = 1.0925484305920792 * x * y
    out[:, 5] = 1.0925484305920792 * y * z
    out[:, 6] = 0.9461746957575601 * zz - 0.31539156525251999
    out[:, 7] = 1.0925484305920792 * x * z
    out[:, 8] = 0.5462742152960396 * (xx - yy)
    out[:, 9] = 0.5900435899266435 * y * (3 * xx - yy)
    out[:, 10] = 2.890611442640554 * x * y * z
    out[:, 11] = 0.4570457994644658 * y * (5 * zz - 1)
    out[:, 12] = 0.3731763325901154 * z * (5 * zz - 3)
    out[:, 13] = 0.4570457994644658 * x * (5 * zz - 1)
    out[:, 14] = 1.445305721320277 * z * (xx - yy)
    out[:, 15] = 0.5900435899266435 * x * (xx - 3 * yy)
    return out


def sh4_of_direction(d: Tensor) -> Tensor:
    """ns/fields/base_field.py:136-142 then encodings.py:711-714: SH of (d+1)/2, no grad."""
    with torch.no_grad():
        return sh4((d + 1.0) / 2.0)


# --------------------------------------------------------------------------------------
# a8 / a9  MLP and trunc_exp
# --------------------------------------------------------------------------------------
def mlp_forward(x: Tensor, layers: Sequence[Tuple[Tensor, Tensor]], out_act: Optional[str] = None,
                keep_hidden: bool = False):
    """Chain of y = x W^T + b with ReLU between layers.  ns/field_components/mlp.py:155-174."""
    hidden = []
    n = len(layers)
    for i, (W, b) in enumerate(layers):
        x = torch.nn.functional.linear(x, W, b)
        if i < n - 1:
            x = torch.relu(x)
            hidden.append(x)
    if out_act == "sigmoid":
        x = torch.sigmoid(x)
    elif out_act is not None:
        raise ValueError(out_act)
    return (x, hidden) if keep_hidden else x


class _TruncExp(torch.autograd.Function):
    """exp forward, gradient g*exp(clamp(x,-15,15)).  ns/field_components/activations.py:28-42."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


# --------------------------------------------------------------------------------------
# a12  density -> weights
# --------------------------------------------------------------------------------------
def weights_from_density(deltas: Tensor, sigma: Tensor) -> Tensor:
    """deltas, sigma [R,S] -> weights [R,S].  ns/cameras/rays.py:128-150."""
    dd = deltas * sigma
    alpha = 1 - torch.exp(-dd)
    acc = torch.cumsum(dd[:, :-1], dim=-1)
    acc = torch.cat([torch.zeros_like(acc[:, :1]), acc], dim=-1)
    return torch.nan_to_num(alpha * torch.exp(-acc))


# --------------------------------------------------------------------------------------
# a3  piecewise spaced sampler
# --------------------------------------------------------------------------------------
def spacing_fn(x: Tensor, thr: float) -> Tensor:
    """ns/models/PreSight/nerfacto_nusc_ms.py:314."""
    return torch.where(x < thr, x / (2 * thr), 1 - 1 / (2 * x / thr))


def spacing_fn_inv(x: Tensor, thr: float) -> Tensor:
    """ns/models/PreSight/nerfacto_nusc_ms.py:315."""
    return torch.where(x < 0.5, x * (2 * thr), thr / (2 - 2 * x))


def s_to_euclid(s: Tensor, nears: Tensor, fars: Tensor, thr: float) -> Tensor:
    """ns/model_components/ray_samplers.py:113-116.  s [R,K], nears/fars [R,1]."""
    s_near, s_far = spacing_fn(nears, thr), spacing_fn(fars, thr)
    return spacing_fn_inv(s * s_far + (1 - s) * s_near, thr)


def spaced_bins(num_rays: int, num_samples: int, jitter: Optional[Tensor]) -> Tensor:
    """Normalised bin edges [R,S+1].  ns/model_components/ray_samplers.py:100-111.

    jitter: one U[0,1) per ray ([R,1], single_jitter) for training, None for eval.
    """
    bins = torch.linspace(0.0, 1.0, num_samples + 1)[None, :]
    if jitter is None:
        return bins.expand(num_rays, -1).contiguous()
    centers = (bins[:, 1:] + bins[:, :-1]) / 2.0
    upper = torch.cat([centers, bins[:, -1:]], -1)
    lower = torch.cat([bins[:, :1], centers], -1)
    return lower + (upper - lower) * jitter


# --------------------------------------------------------------------------------------
# a13  PDF resampling
# --------------------------------------------------------------------------------------
def pdf_resample(weights: Tensor, bins_s: Tensor, num_new: int, jitter: Optional[Tensor],
                 pad: float = 0.01, eps: float = float(torch.finfo(torch.float32).eps)) -> Tensor:
    """weights [R,S], existing normalised bins [R,S+1] -> new normalised bins [R,num_new+1].

    ns/model_components/ray_samplers.py:305-360 with include_original=False.  `jitter` is the
    single per-ray U[0,1) draw ([R,1]) in training, None in eval.
    """
    nb = num_new + 1
    w = weights + pad
    wsum = w.sum(-1, keepdim=True)
    padding = torch.relu(eps - wsum)
    w = w + padding / w.shape[-1]
    wsum = wsum + padding
    pdf = w / wsum
    cdf = torch.minimum(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], dim=-1)
    u = torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb)
    if jitter is None:
        u = (u + 1.0 / (2 * nb)).expand(cdf.shape[0], nb)
    else:
        u = u.expand(cdf.shape[0], nb) + jitter / nb
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, side="right")
    hi = bins_s.shape[-1] - 1
    below = torch.clamp(inds - 1, 0, hi)
    above = torch.clamp(inds, 0, hi)
    c0, c1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(bins_s, -1, below), torch.gather(bins_s, -1, above)
    t = torch.clip(torch.nan_to_num((u - c0) / (c1 - c0), 0), 0, 1)
    return (b0 + t * (b1 - b0)).detach()


# --------------------------------------------------------------------------------------
# a15  renderers
# --------------------------------------------------------------------------------------
def threshold_depth(weights: Tensor, steps: Tensor, threshold: float = 0.5) -> Tensor:
    """First sample whose inclusive cumsum reaches `threshold`.  ns/model_components/renderers.py:352-362."""
    cw = torch.cumsum(weights, dim=-1)
    split = torch.full((weights.shape[0], 1), threshold)
    i = torch.searchsorted(cw, split, side="left").clamp(0, steps.shape[-1] - 1)
    return torch.gather(steps, -1, i)


def expected_depth(weights: Tensor, steps: Tensor, clip_lo=None, clip_hi=None) -> Tensor:
    """ns/model_components/renderers.py:363-381; clip bounds default to the batch-global min/max."""
    d = (weights * steps).sum(-1, keepdim=True) / (weights.sum(-1, keepdim=True) + 1e-10)
    lo = steps.min() if clip_lo is None else clip_lo
    hi = steps.max() if clip_hi is None else clip_hi
    return torch.clip(d, lo, hi)


# --------------------------------------------------------------------------------------
# a16  losses
# --------------------------------------------------------------------------------------
def sky_loss(acc: Tensor, sky_mask: Tensor, eps: float = 1e-7) -> Tensor:
    """ns/model_components/PreSight/losses.py:106-115."""
    target = 1.0 - sky_mask
    a = torch.clip(acc, min=eps, max=1 - eps)
    return torch.nn.functional.binary_cross_entropy(a, target, reduction="none").mean()


def semantic_loss(pred: Tensor, target: Tensor) -> Tensor:
    """ns/model_components/PreSight/losses.py:117-125."""
    return torch.nn.functional.mse_loss(pred, torch.clip(target, 0.0, 1.0), reduction="none").mean()


def distortion_loss(bins_s: Tensor, w: Tensor) -> Tensor:
    """ns/model_components/losses.py:130-149 (mip-NeRF 360)."""
    ut = (bins_s[:, 1:] + bins_s[:, :-1]) / 2
    dut = (ut[:, :, None] - ut[:, None, :]).abs()
    inter = (w * (w[:, None, :] * dut).sum(-1)).sum(-1)
    intra = (w**2 * (bins_s[:, 1:] - bins_s[:, :-1])).sum(-1) / 3
    return (inter + intra).mean()


def line_of_sight_loss(w: Tensor, depth: Tensor, steps: Tensor, sigma: float, sky_mask: Optional[Tensor] = None,
                       upper_bound: float = 75.0) -> Tensor:
    """URF line-of-sight loss, ns/model_components/PreSight/losses.py:28-65.  w, steps [R,S]; depth, sky_mask [R]
    (metres: the caller divides steps by pose_scale_factor, nerfacto_nusc_ms.py:576-629).  Mean over the rays with
    1 < depth < upper_bound (and not sky); NaN when no ray qualifies (torch.mean of an empty selection)."""
    keep = (depth > 1.0) & (depth < upper_bound)
    if sky_mask is not None:
        keep = keep & (sky_mask == 0.0)
    steps = steps.detach()
    d = depth[:, None]
    s = sigma / 3.0  # URF_SIGMA_SCALE_FACTOR
    x = steps - d
    target = torch.exp(-(x * x) / (2 * s * s) - math.log(s) - 0.5 * math.log(2 * math.pi))
    near = (steps <= d + sigma) & (steps >= d - sigma)
    empty = steps < d - sigma
    per_ray = (near * (w - target) ** 2).sum(-1) + (empty * w**2).sum(-1)
    return per_ray[keep].mean()


def normalize_depth(depth: Tensor, upper_bound: float) -> Tensor:
    """ns/model_components/PreSight/losses.py:25-26."""
    return torch.clip(depth / upper_bound, 0.0, 1.0)


def expected_depth_loss(depth: Tensor, pred: Tensor, upper_bound: float = 75.0, sky_mask: Optional[Tensor] = None,
                        inverse: bool = False) -> Tensor:
    """expected_depth_loss (lidar: sky_mask None, inverse False) and expected_monodepth_loss,
    ns/model_components/PreSight/losses.py:67-103.  depth, pred, sky_mask [R] in metres."""
    keep = (depth > 1.0) & (depth < upper_bound)
    if sky_mask is not None:
        keep = keep & (sky_mask == 0.0)
    if inverse:
        t, p = 1 / (depth + 5), 1 / (pred + 5)
    else:
        t, p = normalize_depth(depth, upper_bound), normalize_depth(pred, upper_bound)
    return ((t - p) ** 2)[keep].mean()


def line_of_sight_sigma(step: int, start_step=1000, end_step=30000, max_sigma=5.0, min_sigma=2.0) -> float:
    """nerfacto_nusc_ms.py:387-396."""
    frac = min(max((step - start_step) / (end_step - start_step), 0.0), 1.0)
    return max_sigma - frac * (max_sigma - min_sigma)


def line_of_sight_mult(step: int, mult=0.1, start_step=1000, decay_steps=5000) -> float:
    """nerfacto_nusc_ms.py:398-403."""
    if step <= start_step:
        return 0.0
    return mult / (2.0 ** (step // decay_steps))


def _blur_stepfun(x: Tensor, y: Tensor, r: float):
    """ns/model_components/PreSight/losses.py:127-139."""
    xr, order = torch.sort(torch.cat([x - r, x + r], dim=-1))
    y1 = (torch.cat([y, torch.zeros_like(y[:, :1])], -1) - torch.cat([torch.zeros_like(y[:, :1]), y], -1)) / (2 * r)
    y2 = torch.cat([y1, -y1], dim=-1).take_along_dim(order[:, :-1], dim=-1)
    yr = torch.cumsum((xr[:, 1:] - xr[:, :-1]) * torch.cumsum(y2, dim=-1), dim=-1).clamp_min(0)
    return xr, torch.cat([torch.zeros_like(yr[:, :1]), yr], dim=-1)


def _sorted_interp_quad(x: Tensor, xp: Tensor, fpdf: Tensor, fcdf: Tensor) -> Tensor:
    """ns/model_components/PreSight/losses.py:141-164."""
    mask = x[:, None, :] >= xp[:, :, None]  # [R, len(xp), len(x)]

    def interval(v, want_idx=False):
        v0, i0 = torch.max(torch.where(mask, v[:, :, None], v[:, :1, None]), -2)
        v1, i1 = torch.min(torch.where(~mask, v[:, :, None], v[:, -1:, None]), -2)
        return (v0, v1, i0, i1) if want_idx else (v0, v1)

    c0, c1, i0, i1 = interval(fcdf, True)
    p0 = fpdf.take_along_dim(i0, dim=-1)
    p1 = fpdf.take_along_dim(i1, dim=-1)
    x0, x1 = interval(xp)
    t = torch.clip(torch.nan_to_num((x - x0) / (x1 - x0), 0), 0, 1)
    return c0 + (x - x0) * (p0 + p1 * t + p0 * (1 - t)) / 2


def interlevel_loss_zaa(weights_list: List[Tensor], bins_list: List[Tensor], pulse_width: Sequence[float]) -> Tensor:
    """Zip-NeRF anti-aliased interlevel loss.  ns/model_components/PreSight/losses.py:166-206.

    weights_list[i] [R,S_i], bins_list[i] normalised bins [R,S_i+1]; last entry is the main level.
    """
    c = bins_list[-1].detach()
    w = weights_list[-1].detach()
    wn = w / (c[:, 1:] - c[:, :-1])
    total = 0.0
    for i, (cp, wp) in enumerate(zip(bins_list[:-1], weights_list[:-1])):
        ci, wi = _blur_stepfun(c, wn, pulse_width[i])
        area = 0.5 * (wi[:, 1:] + wi[:, :-1]) * (ci[:, 1:] - ci[:, :-1])
        cdf = torch.cat([torch.zeros_like(area[:, :1]), torch.cumsum(area, dim=-1)], dim=-1)
        ws = torch.diff(_sorted_interp_quad(cp, ci, wi, cdf), dim=-1)
        total = total + ((ws - wp).clamp_min(0) ** 2 / (wp + 1e-5)).mean()
    return total


# --------------------------------------------------------------------------------------
# a1  pinhole ray generation
# --------------------------------------------------------------------------------------
def generate_rays(ray_indices: Tensor, c2w: Tensor, fx: Tensor, fy: Tensor, cx: Tensor, cy: Tensor):
    """ray_indices int64 [R,3] (cam,row,col); c2w [C,3,4]; intrinsics [C].

    ns/model_components/ray_generators.py:43-61 + ns/cameras/cameras.py:292-318 (pixel centre
    +0.5), :614-616, :773-778 (perspective), :841-870.
    Returns origins [R,3], directions [R,3], pixel_area [R,1], directions_norm [R,1].
    """
    cam = ray_indices[:, 0]
    y = ray_indices[:, 1].to(torch.float32) + 0.5
    x = ray_indices[:, 2].to(torch.float32) + 0.5
    fx_, fy_, cx_, cy_ = fx[cam], fy[cam], cx[cam], cy[cam]
    base = torch.stack([(x - cx_) / fx_, -(y - cy_) / fy_], -1)
    offx = torch.stack([(x - cx_ + 1) / fx_, -(y - cy_) / fy_], -1)
    offy = torch.stack([(x - cx_) / fx_, -(y - cy_ + 1) / fy_], -1)
    co = torch.stack([base, offx, offy], dim=0)  # [3,R,2]
    dirs = torch.cat([co, -torch.ones_like(co[..., :1])], dim=-1)  # [3,R,3]
    rot = c2w[cam][:, :3, :3]  # [R,3,3]
    dirs = (dirs[:, :, None, :] * rot[None]).sum(-1)
    norm = torch.maximum(torch.linalg.vector_norm(dirs, dim=-1, keepdim=True), torch.tensor([1e-8]))
    dirs = dirs / norm
    d = dirs[0]
    dx = torch.sqrt(((d - dirs[1]) ** 2).sum(-1))
    dy = torch.sqrt(((d - dirs[2]) ** 2).sum(-1))
    return c2w[cam][:, :3, 3], d, (dx * dy)[:, None], norm[0]


# --------------------------------------------------------------------------------------
# a5  nearest-centroid router
# --------------------------------------------------------------------------------------
def route(points: Tensor, centroids: Tensor) -> Tensor:
    """argmin_k ||p - c_k||_2, int64 [M].  ns/fields/PreSight/ingp_field_ms.py:97."""
    return torch.cdist(points, centroids).argmin(dim=1)


# --------------------------------------------------------------------------------------
# parameter helpers (flat dict keyed by the reference's state-dict names)
# --------------------------------------------------------------------------------------
def _mlp_layers(P: Dict[str, Tensor], prefix: str) -> List[Tuple[Tensor, Tensor]]:
    out, i = [], 0
    while f"{prefix}.layers.{i}.weight" in P:
        out.append((P[f"{prefix}.layers.{i}.weight"], P[f"{prefix}.layers.{i}.bias"]))
        i += 1
    assert out, prefix
    return out


def default_config() -> dict:
    """BASELINE.json cfg 2 ("Boston-Seaport sub-tile"): iNGPField ctor defaults
    (ns/fields/PreSight/ingp_field.py:74-84) + proposal nets of
    ns/models/PreSight/nerfacto_nusc_ms.py:114-121 + the camera-dino method config
    (ns/configs/method_configs.py:145-151)."""
    return dict(
        num_fields=1,
        main=dict(num_levels=16, features_per_level=2, log2_hashmap_size=19, base_res=16, max_res=2048,
                  hidden_dim=64, hidden_dim_color=64, geo_feat_dim=15, semantic_dim=64),
        props=[dict(num_levels=8, features_per_level=1, log2_hashmap_size=20, base_res=16, max_res=1024, hidden_dim=64),
               dict(num_levels=8, features_per_level=1, log2_hashmap_size=20, base_res=16, max_res=4096, hidden_dim=64)],
        sky=dict(width=32, num_layers=3),
        appearance_embed_dim=4, video_embed_dim=12, num_cameras=1440, num_videos=6,
        num_proposal_samples=(128, 64), num_nerf_samples=64,
        near=0.005, far=50.0, thr=5.0,
        pulse_width=(0.03, 0.003),
        interlevel_loss_mult=1.0, distortion_loss_mult=0.002, sky_loss_mult=0.001, semantic_loss_mult=0.5,
    )


def tiny_config() -> dict:
    """BASELINE.json cfg 1: 64^3 tile, 2-level hash grid, 2x32 MLPs."""
    c = default_config()
    c["main"] = dict(num_levels=2, features_per_level=2, log2_hashmap_size=15, base_res=16, max_res=64,
                     hidden_dim=32, hidden_dim_color=32, geo_feat_dim=15, semantic_dim=64)
    c["props"] = [dict(num_levels=2, features_per_level=1, log2_hashmap_size=15, base_res=16, max_res=32, hidden_dim=32),
                  dict(num_levels=2, features_per_level=1, log2_hashmap_size=15, base_res=16, max_res=64, hidden_dim=32)]
    c["num_cameras"], c["num_videos"] = 12, 2
    return c


def prod_shaped_config(K: int = 8, log2_hashmap_size: int = 9) -> dict:
    """Production-SHAPED tile with K routed sub-fields: L10 F4 main grids up to resolution 16384, L8 F1 proposal grids, 64-wide
    MLPs (ns/models/PreSight/nerfacto_nusc_ms.py:88-121), small tables so that fixtures stay small (tests/golden/model_k8.npz)."""
    cfg = default_config()
    cfg["num_fields"] = K
    cfg["main"] = dict(num_levels=10, features_per_level=4, log2_hashmap_size=log2_hashmap_size, base_res=16, max_res=16384, hidden_dim=64,
                       hidden_dim_color=64, geo_feat_dim=15, semantic_dim=64)
    cfg["props"] = [dict(num_levels=8, features_per_level=1, log2_hashmap_size=log2_hashmap_size, base_res=16, max_res=1024, hidden_dim=64),
                    dict(num_levels=8, features_per_level=1, log2_hashmap_size=log2_hashmap_size, base_res=16, max_res=4096, hidden_dim=64)]
    cfg["num_cameras"], cfg["num_videos"] = 48, 2
    return cfg


def _linear_init(gen: torch.Generator, out_f: int, in_f: int) -> Tuple[Tensor, Tensor]:
    # same distribution family as torch.nn.Linear's default (U(-1/sqrt(in), 1/sqrt(in))), own RNG stream
    k = 1.0 / math.sqrt(in_f)
    W = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * k
    b = (torch.rand(out_f, generator=gen) * 2 - 1) * k
    return W, b


def _add_mlp(P, gen, prefix, dims):
    for i in range(len(dims) - 1):
        W, b = _linear_init(gen, dims[i + 1], dims[i])
        P[f"{prefix}.layers.{i}.weight"], P[f"{prefix}.layers.{i}.bias"] = W, b


def make_params(cfg: dict, seed: int = 42, table_scale: float = 1e-3) -> Dict[str, Tensor]:
    """Deterministic synthetic parameter set with the reference's state-dict key names
    (SURVEY.md 8b 'Names that must not change').  Hash tables U(-s, s) like
    ns/field_components/encodings.py:312-314."""
    gen = torch.Generator().manual_seed(seed)
    P: Dict[str, Tensor] = {}
    m = cfg["main"]
    app = cfg["appearance_embed_dim"] + cfg["video_embed_dim"]
    for k in range(cfg["num_fields"]):
        pre = f"field.fields.{k}"
        T = 1 << m["log2_hashmap_size"]
        P[f"{pre}.mlp_base_grid.hash_table"] = (torch.rand(T * m["num_levels"], m["features_per_level"], generator=gen) * 2 - 1) * table_scale
        nin = m["num_levels"] * m["features_per_level"]
        _add_mlp(P, gen, f"{pre}.mlp_base_mlp", [nin, m["hidden_dim"], 1 + m["geo_feat_dim"] + m["semantic_dim"]])
        _add_mlp(P, gen, f"{pre}.semantic_head", [m["semantic_dim"], 64, 64, m["semantic_dim"]])
        _add_mlp(P, gen, f"{pre}.rgb_head", [16 + m["geo_feat_dim"] + app, m["hidden_dim_color"], m["hidden_dim_color"], 3])
        for i, pc in enumerate(cfg["props"]):
            ppre = f"proposal_networks.{i}.fields.{k}"
            Tp = 1 << pc["log2_hashmap_size"]
            P[f"{ppre}.encoding.hash_table"] = (torch.rand(Tp * pc["num_levels"], pc["features_per_level"], generator=gen) * 2 - 1) * table_scale
            _add_mlp(P, gen, f"{ppre}.mlp_base.1", [pc["num_levels"] * pc["features_per_level"], pc["hidden_dim"], 1])
        s = cfg["sky"]
        _add_mlp(P, gen, f"sky_model.fields.{k}.rgb_head", [16 + app] + [s["width"]] * (s["num_layers"] - 1) + [3])
        _add_mlp(P, gen, f"sky_model.fields.{k}.semantic_head", [16] + [s["width"]] * (s["num_layers"] - 1) + [m["semantic_dim"]])
    P["appearance_embedding.embedding.weight"] = torch.randn(cfg["num_cameras"], cfg["appearance_embed_dim"], generator=gen)
    P["video_embedding.embedding.weight"] = torch.randn(cfg["num_videos"], cfg["video_embed_dim"], generator=gen)
    return P


def make_scene(cfg: dict, seed: int = 7):
    """Synthetic nuScenes-shaped rig: 6 pinhole cameras 1600x900 on a polyline,
    poses scaled by 0.05 and mean-centred; K centroids on the polyline, AABBs = +-15 m * 0.05."""
    gen = torch.Generator().manual_seed(seed)
    n_frames = cfg["num_cameras"] // 6
    scale = 0.05
    t = torch.arange(n_frames, dtype=torch.float32) * 0.5
    heading = 0.3 * torch.sin(t / 40.0)
    pos = torch.stack([torch.cumsum(0.5 * torch.cos(heading), 0), torch.cumsum(0.5 * torch.sin(heading), 0),
                       torch.full_like(t, 1.5)], -1)
    yaws = torch.tensor([0.0, 55.0, -55.0, 180.0, 110.0, -110.0]) * math.pi / 180
    c2w = torch.zeros(n_frames, 6, 3, 4)
    for j in range(6):
        a = heading + yaws[j]
        fwd = torch.stack([torch.cos(a), torch.sin(a), torch.zeros_like(a)], -1)  # camera looks along -z_cam
        up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
        right = torch.linalg.cross(fwd, up)
        c2w[:, j, :, 0], c2w[:, j, :, 1], c2w[:, j, :, 2], c2w[:, j, :, 3] = right, up, -fwd, pos
    c2w = c2w.reshape(-1, 3, 4)
    c2w[:, :, 3] = (c2w[:, :, 3] - c2w[:, :, 3].mean(0)) * scale
    C = c2w.shape[0]
    fx = torch.full((C,), 1266.0) + torch.rand(C, generator=gen)
    fy = fx.clone()
    cx, cy = torch.full((C,), 800.0), torch.full((C,), 450.0)
    K = cfg["num_fields"]
    sel = torch.linspace(0, C - 1, K + 2)[1:-1].long() if K > 1 else torch.tensor([C // 2])
    centroids = c2w[sel, :, 3].clone()
    ext = 15.0 * scale
    if K == 1:
        lo = c2w[:, :, 3].quantile(0.02, dim=0) - ext
        hi = c2w[:, :, 3].quantile(0.98, dim=0) + ext
        aabbs = torch.stack([lo, hi])[None]
    else:
        aabbs = torch.stack([torch.stack([c - 3 * ext, c + 3 * ext]) for c in centroids])
    sd = cfg["main"]["semantic_dim"]
    dino_to_rgb = dict(reduction_matrix=torch.randn(sd, 3, generator=gen) / math.sqrt(sd), rgb_min=torch.full((3,), -0.4),
                       rgb_max=torch.full((3,), 0.5), mean=torch.rand(sd, generator=gen))
    return dict(c2w=c2w, fx=fx, fy=fy, cx=cx, cy=cy, centroids=centroids, aabbs=aabbs, H=900, W=1600,
                frames_per_video=max(1, C // cfg["num_videos"]), dino_to_rgb=dino_to_rgb)


def make_batch(cfg: dict, scene: dict, num_rays: int, step: int = 0):
    """Uniform-random ray indices + random targets, `manual_seed(1234+step)`."""
    g = torch.Generator().manual_seed(1234 + step)
    C = scene["c2w"].shape[0]
    idx = torch.stack([torch.randint(0, C, (num_rays,), generator=g), torch.randint(0, scene["H"], (num_rays,), generator=g),
                       torch.randint(0, scene["W"], (num_rays,), generator=g)], -1)
    return dict(
        ray_indices=idx,
        video_ids=torch.clamp(idx[:, 0] // scene["frames_per_video"], max=cfg["num_videos"] - 1),
        rgb=torch.rand(num_rays, 3, generator=g),
        features=torch.rand(num_rays, cfg["main"]["semantic_dim"], generator=g),
        sky=(torch.rand(num_rays, generator=g) < 0.15).float(),
        jitter=torch.rand(3, num_rays, 1, generator=g),  # spaced sampler + 2 pdf samplers
    )


# --------------------------------------------------------------------------------------
# a11  fields
# --------------------------------------------------------------------------------------
def prop_density(P, cfg, i: int, k: int, pos: Tensor, aabb: Tensor) -> Tensor:
    """ns/fields/PreSight/prop_density_field.py:129-153 -> density [M]."""
    pc = cfg["props"][i]
    pre = f"proposal_networks.{i}.fields.{k}"
    u, sel = normalize_contract(pos, aabb)
    sc = hash_scalings(pc["num_levels"], pc["base_res"], pc["max_res"])
    enc = hash_encode(u, P[f"{pre}.encoding.hash_table"], sc, pc["log2_hashmap_size"])
    raw = mlp_forward(enc, _mlp_layers(P, f"{pre}.mlp_base.1"))[:, 0]
    return trunc_exp(raw) * sel


def main_density(P, cfg, k: int, pos: Tensor, aabb: Tensor):
    """ns/fields/PreSight/ingp_field.py:168-191 -> (density [M], embedding [M, geo+sem])."""
    m = cfg["main"]
    pre = f"field.fields.{k}"
    u, sel = normalize_contract(pos, aabb)
    sc = hash_scalings(m["num_levels"], m["base_res"], m["max_res"])
    enc = hash_encode(u, P[f"{pre}.mlp_base_grid.hash_table"], sc, m["log2_hashmap_size"])
    h = mlp_forward(enc, _mlp_layers(P, f"{pre}.mlp_base_mlp"))
    return trunc_exp(h[:, 0]) * sel, h[:, 1:]


def main_heads(P, cfg, k: int, dirs: Tensor, emb: Tensor, app: Optional[Tensor]):
    """ns/fields/PreSight/ingp_field.py:193-237 -> (rgb [M,3], semantics [M,64])."""
    m = cfg["main"]
    pre = f"field.fields.{k}"
    geo, sem_in = emb[:, : m["geo_feat_dim"]], emb[:, m["geo_feat_dim"]:]
    sem = mlp_forward(sem_in, _mlp_layers(P, f"{pre}.semantic_head"))
    parts = [sh4_of_direction(dirs), geo] + ([app] if app is not None else [])
    rgb = mlp_forward(torch.cat(parts, -1), _mlp_layers(P, f"{pre}.rgb_head"), out_act="sigmoid")
    return rgb, sem


def sky_outputs(P, k: int, dirs: Tensor, app: Optional[Tensor]):
    """ns/fields/PreSight/sky_field.py:95-110 -> (rgb [R,3], semantics [R,64])."""
    d = sh4_of_direction(dirs)
    x = torch.cat([d, app], -1) if app is not None else d
    rgb = mlp_forward(x, _mlp_layers(P, f"sky_model.fields.{k}.rgb_head"), out_act="sigmoid")
    sem = mlp_forward(d, _mlp_layers(P, f"sky_model.fields.{k}.semantic_head"))
    return rgb, sem


def _routed(fn_per_field, assign: Tensor, K: int, outs_dims: Sequence[int]):
    """Masked gather -> per-sub-field evaluation -> masked scatter, as in
    ns/fields/PreSight/ingp_field_ms.py:100-126.  fn_per_field(k, mask) returns a tuple of [m,dim]/[m] tensors."""
    M = assign.shape[0]
    outs = None
    for k in range(K):
        mask = assign == k
        if not bool(mask.any()):
            continue
        vals = fn_per_field(k, mask)
        if outs is None:
            outs = [torch.zeros((M,) + v.shape[1:], dtype=v.dtype) for v in vals]
        # index_put keeps autograd history (functional form of `out[mask] = v`)
        outs = [o.masked_scatter(mask.view((-1,) + (1,) * (v.dim() - 1)).expand_as(o), v) for o, v in zip(outs, vals)]
    return outs


# --------------------------------------------------------------------------------------
# a14 + a17  whole-model forward (training or eval), losses, one training step
# --------------------------------------------------------------------------------------
def model_forward(P, cfg, scene, batch, training: bool = True, anneal: float = 1.0, prop_requires_grad: bool = True,
                  main_override=None, prop_from_main: bool = False):
    """NerfactoNuscMSModel.forward (collider + get_outputs).
    ns/models/base_model.py:131-142, ns/models/PreSight/nerfacto_nusc_ms.py:452-546,
    ns/model_components/ray_samplers.py:572-614.
    main_override(static_eval, pos [R*S,3], dir_s, app_s, R, S) -> (sigma, rgb_s, sem_s, extras): hook used by
    oracle/dual_oracle.py (BASELINE cfg 4) to put a second field next to the static one; `static_eval()` evaluates the
    reference's field exactly as below.
    prop_from_main: the proposal levels are sampled from the MAIN field's density (the "teacher" of the learnable synthetic
    scene places its samples by its own density: teacher_targets below)."""
    K = cfg["num_fields"]
    cent, aabbs = scene["centroids"], scene["aabbs"]
    o, d, _, _ = generate_rays(batch["ray_indices"], scene["c2w"], scene["fx"], scene["fy"], scene["cx"], scene["cy"])
    R = o.shape[0]
    nears = torch.full((R, 1), cfg["near"] if training else 0.0)  # scene_colliders.py:182-187
    fars = torch.full((R, 1), cfg["far"])
    thr = cfg["thr"]
    jit = batch["jitter"] if training else [None, None, None]

    def positions(eu):  # rays.py:49-58
        mid = (eu[:, :-1] + eu[:, 1:]) / 2
        return (o[:, None, :] + d[:, None, :] * mid[:, :, None]).reshape(-1, 3)

    bins_list, weights_list, eu_list = [], [], []
    S = list(cfg["num_proposal_samples"]) + [cfg["num_nerf_samples"]]
    bins = spaced_bins(R, S[0], jit[0])
    w = None
    for lvl in range(len(S)):
        if lvl > 0:
            bins = pdf_resample(torch.pow(w, anneal), bins, S[lvl], jit[lvl])
        eu = s_to_euclid(bins, nears, fars, thr)
        pos = positions(eu)
        deltas = eu[:, 1:] - eu[:, :-1]
        if lvl < len(S) - 1:
            assign = route(pos, cent)
            ctx = torch.enable_grad() if (training and prop_requires_grad) else torch.no_grad()
            with ctx:
                if prop_from_main:
                    (sigma,) = _routed(lambda k, msk: (main_density(P, cfg, k, pos[msk], aabbs[k])[0],), assign, K, [1])
                else:
                    (sigma,) = _routed(lambda k, msk, i=lvl: (prop_density(P, cfg, i, k, pos[msk], aabbs[k]),), assign, K, [1])
            w = weights_from_density(deltas, sigma.view(R, -1))
            bins_list.append(bins)
            weights_list.append(w)
            eu_list.append(eu)

    # appearance embedding: nerfacto_nusc_ms.py:455-489
    cam = batch["ray_indices"][:, 0]
    if training:
        app = torch.cat([P["appearance_embedding.embedding.weight"][cam],
                         P["video_embedding.embedding.weight"][batch["video_ids"]]], -1)
    else:
        app = torch.cat([P["appearance_embedding.embedding.weight"].mean(0),
                         P["video_embedding.embedding.weight"].mean(0)])[None].expand(R, -1)
    Sm = S[-1]
    app_s = app[:, None, :].expand(R, Sm, app.shape[-1]).reshape(R * Sm, -1)
    dir_s = d[:, None, :].expand(R, Sm, 3).reshape(-1, 3)
    assign = route(pos, cent)

    def main_eval(k, msk):
        sg, emb = main_density(P, cfg, k, pos[msk], aabbs[k])
        rgb, sem = main_heads(P, cfg, k, dir_s[msk], emb, app_s[msk])
        return sg, rgb, sem

    extras = {}
    if main_override is None:
        sigma, rgb_s, sem_s = _routed(main_eval, assign, K, [1, 3, 64])
    else:
        sigma, rgb_s, sem_s, extras = main_override(lambda: _routed(main_eval, assign, K, [1, 3, 64]), pos, dir_s, app_s, R, Sm)
    w = weights_from_density(deltas, sigma.view(R, Sm))
    weights_list.append(w)
    bins_list.append(bins)
    eu_list.append(eu)
    steps = (eu[:, :-1] + eu[:, 1:]) / 2
    rgb = (w[:, :, None] * rgb_s.view(R, Sm, 3)).sum(1)
    acc = torch.clamp(w.sum(-1, keepdim=True), 0.0, 1.0)
    with torch.no_grad():
        depth = threshold_depth(w, steps)
    exp_depth = expected_depth(w, steps)
    sem = (w[:, :, None] * sem_s.view(R, Sm, -1)).sum(1)
    # sky model routed by ray origin: sky_field_ms.py:97-114
    sky_assign = route(o, cent)
    sky_rgb, sky_sem = _routed(lambda k, msk: sky_outputs(P, k, d[msk], app[msk]), sky_assign, K, [3, 64])
    rgb = rgb + (1.0 - acc) * sky_rgb
    sem = sem + (1.0 - acc) * sky_sem
    if not training:
        pass  # RGBRenderer eval-mode nan_to_num/clamp acts on the pre-sky rgb only (renderers.py:221-228)
    out = dict(rgb=rgb, accumulation=acc, depth=depth, expected_depth=exp_depth, semantics=sem,
               weights_list=weights_list, bins_list=bins_list, euclid_list=eu_list, origins=o, directions=d)
    out.update(extras)
    for i in range(len(S) - 1):
        st = (eu_list[i][:, :-1] + eu_list[i][:, 1:]) / 2
        out[f"prop_depth_{i}"] = threshold_depth(weights_list[i], st)
    return out


def loss_dict(out, batch, cfg) -> Dict[str, Tensor]:
    """ns/models/PreSight/nerfacto_nusc_ms.py:558-645 for the camera-only configs (no depth losses)."""
    L = {}
    L["rgb_loss"] = torch.nn.functional.mse_loss(batch["rgb"], out["rgb"])
    L["sky_loss"] = cfg["sky_loss_mult"] * sky_loss(out["accumulation"].view(-1, 1), batch["sky"].view(-1, 1))
    L["semantic_loss"] = cfg["semantic_loss_mult"] * semantic_loss(out["semantics"], batch["features"])
    L["interlevel_loss"] = cfg["interlevel_loss_mult"] * interlevel_loss_zaa(out["weights_list"], out["bins_list"], cfg["pulse_width"])
    L["distortion_loss"] = cfg["distortion_loss_mult"] * distortion_loss(out["bins_list"][-1], out["weights_list"][-1])
    return L


def train_step(P, cfg, scene, batch, anneal: float = 1.0):
    """Forward + losses + backward (no optimizer), as timed by the reference's rays/s metric
    (ns/engine/trainer.py:463-486 minus the optimizer).  Returns (losses, outputs, grads)."""
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    out = model_forward(Pg, cfg, scene, batch, training=True, anneal=anneal)
    L = loss_dict(out, batch, cfg)
    total = sum(L.values())
    total.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Pg.items()}
    return L, out, grads


# --------------------------------------------------------------------------------------
# f1  the training loop: schedules, Adam, N iterations (ns/engine/trainer.py:246-300, 463-505)
# --------------------------------------------------------------------------------------
def anneal_at(step: int, max_num_iters: int, slope: float = 10.0) -> float:
    """proposal weight anneal, ns/models/PreSight/nerfacto_nusc_ms.py:423-434 (set_anneal callback, BEFORE each iteration)."""
    x = float(np.clip(step / max_num_iters, 0, 1))
    return slope * x / ((slope - 1) * x + 1)


def proposal_update_sched(step: int, warmup: int, every: int = 5) -> float:
    """ns/models/PreSight/nerfacto_nusc_ms.py:300-305."""
    return float(np.clip(np.interp(step, [0, warmup], [0, every]), 1, every))


def lr_at(t: int, lr_init: float, warmup_steps: int, milestones: Sequence[int], gamma: float = 0.33, start_factor: float = 0.01) -> float:
    """learning rate after t scheduler steps: ChainedScheduler([LinearLR(start_factor 0.01, total_iters warmup_steps),
    MultiStepLR(milestones, gamma 0.33)]), ns/engine/my_schedulers.py:50-70, in closed form."""
    warm = start_factor + (1.0 - start_factor) * min(t, warmup_steps) / warmup_steps if warmup_steps else 1.0
    return lr_init * warm * gamma ** sum(1 for m in milestones if m <= t)


def adam_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, eps: float, weight_decay: float,
                b1: float = 0.9, b2: float = 0.999):
    """torch.optim.Adam (L2 weight decay added to the gradient, no amsgrad), in place; `step` = this parameter's own count of
    updates including this one.  The gradient is whatever backward left in .grad: with PreSight's default
    update_grad_scaler=False that is the 2**10-SCALED gradient, never unscaled (ns/engine/trainer.py:481-486,
    ns/engine/optimizers.py:133-140) -- so weight_decay * p is added to 1024 * dL/dp."""
    g = g + weight_decay * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def train_trajectory(P, cfg, scene, batches, max_iterations: int, loss_scale: float = 2.0 ** 10, lr: float = 1e-2, eps: float = 1e-15,
                     weight_decay: float = 1e-5, proposal_update_every: int = 5, snapshots: Sequence[int] = ()):
    """N = len(batches) training iterations as the reference's Trainer runs them for a PreSight method config with
    `max_iterations` (ns/configs/method_configs.py:130-171: anneal / proposal warm-up over max_iterations // 10, LR warm-up over
    max_iterations // 10, milestones at 1/4, 1/2, 3/4):
      before:  set_anneal(step)                                                    nerfacto_nusc_ms.py:423-434
      forward: proposal networks with gradients only when `updated`                ray_samplers.py:586, 601-606
      backward of loss_scale * sum(loss_dict)                                      trainer.py:478-481 (GradScaler 2**10, never updated)
      Adam on the scaled gradients for every parameter that received one (a parameter whose gradient is None -- proposal nets
      off schedule, sub-fields without samples -- is skipped and keeps its step count)    optimizers.py:133-140
      scheduler step                                                               trainer.py:499-505
      after:   proposal_sampler.step_cb(step)                                      ray_samplers.py:566-569
    -> dict(losses [N, 5] in loss_dict order, lr [N], anneal [N], updated [N], touched {name: [N]}, params (final), snaps {step: params})"""
    Q = {k: v.detach().clone() for k, v in P.items()}
    M = {k: torch.zeros_like(v) for k, v in Q.items()}
    V = {k: torch.zeros_like(v) for k, v in Q.items()}
    nstep = {k: 0 for k in Q}
    warm = max_iterations // 10
    miles = [max_iterations // 4, max_iterations // 2, max_iterations * 3 // 4]
    steps_since_update, sampler_step = 0, 0
    rec = dict(losses=[], lr=[], anneal=[], updated=[], touched={k: [] for k in Q}, snaps={})
    for step, batch in enumerate(batches):
        anneal = anneal_at(step, warm)
        updated = steps_since_update > proposal_update_sched(sampler_step, warm, proposal_update_every) or sampler_step < 10
        cur_lr = lr_at(step, lr, warm, miles)
        Pg = {k: v.detach().clone().requires_grad_(True) for k, v in Q.items()}
        out = model_forward(Pg, cfg, scene, batch, training=True, anneal=anneal, prop_requires_grad=updated)
        L = loss_dict(out, batch, cfg)
        (sum(L.values()) * loss_scale).backward()
        if updated:
            steps_since_update = 0
        with torch.no_grad():
            for k, pg in Pg.items():
                rec["touched"][k].append(int(pg.grad is not None))
                if pg.grad is not None:
                    nstep[k] += 1
                    adam_update(Q[k], pg.grad, M[k], V[k], nstep[k], cur_lr, eps, weight_decay)
        sampler_step = step
        steps_since_update += 1
        rec["losses"].append([float(v.detach()) for v in L.values()])
        rec["lr"].append(cur_lr)
        rec["anneal"].append(anneal)
        rec["updated"].append(int(updated))
        if step in snapshots:
            rec["snaps"][step] = {k: v.clone() for k, v in Q.items()}
    rec["params"], rec["param_steps"] = Q, nstep
    return rec


# --------------------------------------------------------------------------------------
# learnable synthetic scene (SURVEY.md 8d "PSNR vs synthetic GT after K steps"): same construction as presight_amd/synthetic.py
# --------------------------------------------------------------------------------------
def teacher_params(cfg: dict, scene: dict, seed: int = 1234, max_res: float = 128, log_density=(-2.5, 2.0), rgb_gain: float = 32.0,
                   sem_gain: float = 3.0, probe: int = 4096, init_seed: int = 77) -> Dict[str, Tensor]:
    """The teacher of the learnable scene: make_params(init_seed) with the main hash tables redrawn U(-1, 1) on the levels of
    resolution <= max_res and zero above; per sub-field the density head's output row rescaled / re-biased so that ln(density) has
    mean / std `log_density` over `probe` points uniform in the sub-field's box; colour and semantic output layers amplified,
    semantic output bias 0.5.  One generator, fixed draw order (tables in sorted key order, then the probe points of sub-field
    0, 1, ...): identical to presight_amd.synthetic.shape_teacher_ (the same teacher on both sides, up to the fp32 rounding of the
    probe statistics)."""
    P = make_params(cfg, seed=init_seed)
    g = torch.Generator().manual_seed(seed)
    m = cfg["main"]
    sc = hash_scalings(m["num_levels"], m["base_res"], m["max_res"])
    for key in sorted(P):
        if key.startswith("field.") and key.endswith("mlp_base_grid.hash_table"):
            v = P[key]
            L = sc.numel()
            tab = torch.rand(v.shape, generator=g) * 2 - 1
            tab.view(L, v.shape[0] // L, -1)[sc > max_res] = 0.0
            v.copy_(tab)
    for k in range(cfg["num_fields"]):
        pre = f"field.fields.{k}"
        box = scene["aabbs"][k]
        pts = box[0] + (box[1] - box[0]) * torch.rand(probe, 3, generator=g)
        with torch.no_grad():
            sigma, _ = main_density(P, cfg, k, pts, box)
        raw = torch.log(sigma.clamp_min(1e-30)).double()
        mu, sdev = float(raw.mean()), float(raw.std())
        gain = log_density[1] / sdev
        W, b = P[f"{pre}.mlp_base_mlp.layers.1.weight"], P[f"{pre}.mlp_base_mlp.layers.1.bias"]
        b0 = float(b[0])
        W[0] *= gain
        b[0] = log_density[0] - gain * (mu - b0)
        P[f"{pre}.rgb_head.layers.2.weight"] *= rgb_gain
        P[f"{pre}.semantic_head.layers.2.weight"] *= sem_gain
        P[f"{pre}.semantic_head.layers.2.bias"].fill_(0.5)
    return P


def teacher_targets(Pt, cfg, scene, ray_indices: Tensor, video_ids: Tensor, sky_accumulation: float = 0.5, far: float = 5.0) -> Dict[str, Tensor]:
    """per-pixel targets rendered by the teacher in eval mode, far plane `far` (beyond it the scene is empty), its samples placed by
    its own density: rgb [n,3], features [n,64] clipped to [0,1], sky [n] = accumulation < 0.5"""
    tc = dict(cfg)
    tc["far"] = far
    with torch.no_grad():
        out = model_forward(Pt, tc, scene, dict(ray_indices=ray_indices, video_ids=video_ids), training=False, prop_from_main=True)
    acc = out["accumulation"].reshape(-1)
    return dict(rgb=out["rgb"], features=out["semantics"].clamp(0.0, 1.0), sky=(acc < sky_accumulation).to(acc.dtype),
                accumulation=acc)


def eval_psnr(P, cfg, scene, ray_indices: Tensor, video_ids: Tensor, target_rgb: Tensor) -> float:
    """PSNR of the eval-mode render (no jitter, mean appearance code) against target pixels; eval clamps rgb to [0,1] before the
    sky blend (renderers.py:221-228)"""
    with torch.no_grad():
        out = model_forward(P, cfg, scene, dict(ray_indices=ray_indices, video_ids=video_ids), training=False)
    return psnr(out["rgb"], target_rgb)


def feature_colormap(feat: Tensor, dino_to_rgb: dict) -> Tensor:
    """PCA projection of 64-d features to RGB.  ns/utils/colormaps.py:212-234."""
    x = (feat - dino_to_rgb["mean"].to(feat)) @ dino_to_rgb["reduction_matrix"].to(feat)
    x = (x - dino_to_rgb["rgb_min"].to(feat)) / (dino_to_rgb["rgb_max"].to(feat) - dino_to_rgb["rgb_min"].to(feat))
    return torch.clamp(x, 0, 1)


def psnr(pred: Tensor, gt: Tensor) -> float:
    """10 log10(1/MSE), data_range 1.  ns/models/PreSight/nerfacto_nusc_ms.py:382,554."""
    return float(10.0 * torch.log10(1.0 / torch.mean((pred - gt) ** 2)))


# --------------------------------------------------------------------------------------
# a18  prior-extraction queries
# --------------------------------------------------------------------------------------
def voxel_index(points: Tensor, voxel: float, min_bound: Tensor) -> Tensor:
    """Open3D voxel_down_sample_and_trace rule (open3d is absent from the
    reference tree, parity of this row is pinned only by its documented formula):
    idx = floor((p - (min_bound - voxel/2)) / voxel) per axis, int64 [n,3].
    Call site ns/scripts/extract_priors.py:216-245 (min_bound = min-1, voxel 0.4)."""
    ref = (min_bound.double() - voxel * 0.5)
    return torch.floor((points.double() - ref) / voxel).to(torch.int64)


def prior_query(P, cfg, scene, pts: Tensor):
    """mean(sigma_prop0, sigma_prop1, sigma_main) and clipped fp16 semantics for world points.
    ns/scripts/extract_priors.py:133-138 (+ ingp_field.py:252-267)."""
    K = cfg["num_fields"]
    cent, aabbs = scene["centroids"], scene["aabbs"]
    assign = route(pts, cent)
    with torch.no_grad():
        dens = []
        for i in range(len(cfg["props"])):
            (s,) = _routed(lambda k, msk, i=i: (prop_density(P, cfg, i, k, pts[msk], aabbs[k]),), assign, K, [1])
            dens.append(s)

        def main_eval(k, msk):
            sg, emb = main_density(P, cfg, k, pts[msk], aabbs[k])
            g = cfg["main"]["geo_feat_dim"]
            sem = mlp_forward(emb[:, g:], _mlp_layers(P, f"field.fields.{k}.semantic_head"))
            return sg, sem

        sg, sem = _routed(main_eval, assign, K, [1, 64])
        dens.append(sg)
        density = torch.stack(dens, 0).mean(0)
        return density, sem.clip(0, 1).half()


def extract_frame(P, cfg, scene, camera_idx: int, scaling: float, pose_scale_factor: float, max_depth: float = 50.0,
                  min_depth: float = 0.5, depth_type: str = "depth"):
    """One iteration of the frame loop of ns/scripts/extract_priors.py:99-145 (no segmentation mask): all pixels of camera
    `camera_idx` at the rescaled resolution (Cameras.rescale_output_resolution, ns/cameras/cameras.py:953-958: intrinsics * s,
    size = trunc(size * s) in float32) -> eval depth march (get_depth_for_camera_ray_bundle, one chunk) -> world points ->
    min_depth < depth < max_depth, -3 < z < 6 -> mean density, clipped fp16 semantics, PCA colours (computed in fp16 like the
    reference, whose colormap casts its matrices to the features' dtype).
    -> dict(raw_depth [H*W], sel bool [H*W], world [n,3], dens [n], feats f16 [n,64], colors f16 [n,3])"""
    s32 = torch.tensor([scaling], dtype=torch.float32)
    H = int((torch.tensor(int(scene["H"])) * s32).to(torch.int64))
    W = int((torch.tensor(int(scene["W"])) * s32).to(torch.int64))
    rows, cols = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    ri = torch.stack([torch.full((H * W,), int(camera_idx)), rows.reshape(-1), cols.reshape(-1)], -1)
    sc = dict(scene)
    for k in ("fx", "fy", "cx", "cy"):
        sc[k] = scene[k] * s32
    with torch.no_grad():
        out = model_forward(P, cfg, sc, {"ray_indices": ri, "video_ids": torch.zeros(H * W, dtype=torch.int64)}, training=False)
        depth = out[depth_type] / pose_scale_factor
        world = (out["origins"] / pose_scale_factor + out["directions"] * depth).view(-1, 3)
        depth = depth.flatten()
        sel = (depth < max_depth) & (depth > min_depth) & (world[:, 2] > -3.0) & (world[:, 2] < 6.0)
        world = world[sel]
        res = dict(raw_depth=depth, sel=sel, world=world)
        if world.shape[0]:
            dens, feats = prior_query(P, cfg, scene, world * pose_scale_factor)
            res.update(dens=dens, feats=feats, colors=feature_colormap(feats, scene["dino_to_rgb"]))
    return res


def voxel_downsample(points: Tensor, features: Tensor, colors: Optional[Tensor], voxel: float = 0.4):
    """ns/scripts/extract_priors.py:160-191 + 216-245: Open3D voxel_down_sample_and_trace (index rule of voxel_index with
    min_bound = points.min - 1; output point = mean of the members) and the per-voxel traces: colour = fp32 mean of the members,
    feature = fp64 mean of the fp16 members -> fp16, hits = member count.  Plain python / numpy loops (small cases only).
    -> dict keyed by the integer voxel triple: (point f64 [3], feature f16 [C], colour f32 [3] | None, hits)."""
    import numpy as np

    mn = points.min(0).values - 1.0
    idx = voxel_index(points, voxel, mn).numpy()
    pts, fe = points.double().numpy(), features.numpy()
    co = colors.float().numpy() if colors is not None else None
    groups: Dict[tuple, list] = {}
    for i, key in enumerate(map(tuple, idx.tolist())):
        groups.setdefault(key, []).append(i)
    out = {}
    for key, members in groups.items():
        m = np.asarray(members)
        out[key] = (pts[m].mean(axis=0), fe[m].astype(np.float64).mean(axis=0).astype(np.float16),
                    None if co is None else co[m].mean(axis=0), len(members))
    return out
