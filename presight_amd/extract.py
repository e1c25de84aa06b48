"""Prior extraction on the HIP path (ns/scripts/extract_priors.py:112-208).

    query_priors     mean(sigma_prop0, sigma_prop1, sigma_main) + clipped fp16 semantics for world points
    dense_tile_query BASELINE config 5: the res^3 lattice of one tile, streamed in chunks (frame-/slab-shardable)
    voxel_index      Open3D's voxel_down_sample_and_trace index rule, bit exact int64
    voxelize         per-voxel mean point / colour, fp64-mean feature -> fp16, hit counts (extract_priors.py:166-191)"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from ._lib import check, lib
from .ops import _f32, _p, _stream


def voxel_index(points: Tensor, voxel: float, min_bound: Tensor) -> Tensor:
    pts = _f32(points)
    mb = (ctypes.c_double * 3)(*[float(v) for v in min_bound.double().cpu().tolist()])
    idx = torch.empty(pts.shape[0], 3, device=pts.device, dtype=torch.int64)
    check(lib().ps_voxel_index(_p(pts), pts.shape[0], float(voxel), mb, _p(idx), _stream()), "ps_voxel_index")
    return idx


def lattice_points(aabb: Tensor, res: int, start: int, count: int, device) -> Tensor:
    a = (ctypes.c_float * 6)(*[float(v) for v in aabb.reshape(-1).cpu().tolist()])
    pts = torch.empty(count, 3, device=device)
    check(lib().ps_lattice_points(a, res, start, count, _p(pts), _stream()), "ps_lattice_points")
    return pts


@torch.no_grad()
def query_priors(model, pts: Tensor) -> Tuple[Tensor, Tensor]:
    """-> (mean density [n], features fp16 [n,64]); pts are in the model's (scaled) frame."""
    dens = [p.density_fn(pts).reshape(-1) for p in model.proposal_networks]
    d_main, sem = model.field.density_and_semantics(pts)  # one pass of the main field for both (the reference makes three)
    dens.append(d_main.reshape(-1))
    if len(dens) == 3:
        out = torch.empty_like(dens[0])
        check(lib().ps_mean_density(_p(dens[0]), _p(dens[1]), _p(dens[2]), out.numel(), _p(out), _stream()), "ps_mean_density")
    else:
        out = torch.stack(dens, 0).mean(0)
    feats = sem.reshape(-1, sem.shape[-1]).clip(0.0, 1.0).to(torch.float16)
    return out, feats


@torch.no_grad()
def dense_tile_query(model, aabb: Tensor, res: int = 512, chunk: int = 1 << 22, start: int = 0, count: Optional[int] = None,
                     density_threshold: float = 1.0, voxel: float = 0.4, pose_scale_factor: float = 0.05) -> Dict[str, Tensor]:
    """Evaluate the prior fields on lattice points [start, start+count) of the res^3 lattice over `aabb` (model frame).
    Ranks shard the lattice by giving each one a contiguous [start, count) slab; there is no exchange until the final
    integer-key merge.  Returns the points above the density threshold with their features and integer voxel index."""
    dev = model.device
    total = res ** 3 if count is None else count
    keep_pts, keep_feat, keep_dens = [], [], []
    for s in range(start, start + total, chunk):
        n = min(chunk, start + total - s)
        pts = lattice_points(aabb, res, s, n, dev)
        dens, feats = query_priors(model, pts)
        m = dens > density_threshold
        keep_pts.append(pts[m] / pose_scale_factor)
        keep_feat.append(feats[m])
        keep_dens.append(dens[m])
    P = torch.cat(keep_pts) if keep_pts else torch.zeros(0, 3, device=dev)
    out = {"points": P, "features": torch.cat(keep_feat) if keep_feat else torch.zeros(0, 64, device=dev, dtype=torch.float16),
           "densities": torch.cat(keep_dens) if keep_dens else torch.zeros(0, device=dev)}
    if P.shape[0] > 0:
        out["min_bound"] = P.min(0).values - 1.0
        out["voxel_index"] = voxel_index(P, voxel, out["min_bound"])
    return out


@torch.no_grad()
def voxelize(points: Tensor, features: Tensor, colors: Optional[Tensor], voxel: float = 0.4):
    """Group points by integer voxel index: -> dict(points f32 [V,3] (mean), features f16 [V,64] (fp64 mean), colors, hits,
    index int64 [V,3]).  Output order is by sorted index (Open3D's is hash-map order: compare as sets keyed by index)."""
    mb = points.min(0).values - 1.0
    idx = voxel_index(points, voxel, mb)
    uniq, inv, hits = torch.unique(idx, dim=0, return_inverse=True, return_counts=True)
    V = uniq.shape[0]
    psum = torch.zeros(V, 3, device=points.device, dtype=torch.float64).index_add_(0, inv, points.double())
    fsum = torch.zeros(V, features.shape[1], device=points.device, dtype=torch.float64).index_add_(0, inv, features.double())
    out = {"index": uniq, "hits": hits, "points": (psum / hits[:, None]).float(), "features": (fsum / hits[:, None]).half(),
           "min_bound": mb}
    if colors is not None:
        csum = torch.zeros(V, 3, device=points.device, dtype=torch.float64).index_add_(0, inv, colors.double())
        out["colors"] = (csum / hits[:, None]).float()
    return out


@torch.no_grad()
def finalize_priors(vox: dict, origin: Tensor, hit_thr_ratio: float = 0.0) -> dict:
    """The `extracted_priors.pkl` payload of ns/scripts/extract_priors.py:186-208 from voxelize()'s output: voxels whose hit
    count exceeds the `hit_thr_ratio` quantile of all hit counts (numpy's linear-interpolation quantile), as numpy arrays
    {points f32 [V,3], features f16 [V,64], colors f32 [V,3], hits int64 [V], origin f32 [3]} -- the wire format read by
    occupancy/mmdet3d/datasets/prior_utils/city_prior.py:59-73."""
    import numpy as np

    hits = vox["hits"].cpu().numpy()
    keep = hits > np.quantile(hits, hit_thr_ratio)
    out = {"points": vox["points"].cpu().numpy()[keep].astype(np.float32),
           "features": vox["features"].cpu().numpy()[keep].astype(np.float16),
           "hits": hits[keep],
           "origin": origin.detach().cpu().numpy().astype(np.float32)}
    if "colors" in vox:
        out["colors"] = vox["colors"].cpu().numpy()[keep].astype(np.float32)
    return out


def save_priors(path: str, priors: dict) -> None:
    """pickle.dump of finalize_priors()'s dict (extract_priors.py:199-208)"""
    import pickle

    with open(path, "wb") as f:
        pickle.dump(priors, f)
