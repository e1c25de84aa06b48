"""Prior extraction on the HIP path (ns/scripts/extract_priors.py:34-245).

    extract_voxels   the reference's extraction: per camera frame depth march -> world points -> depth / height filters ->
                     mean density of the three fields + clipped fp16 semantics + PCA colours; then density threshold, voxel
                     down-sampling, hit-count filter -> the `extracted_priors.pkl` payload.  Frames shard over ranks.
    query_priors     mean(sigma_prop0, sigma_prop1, sigma_main) + clipped fp16 semantics for world points
    dense_tile_query BASELINE config 5: the res^3 lattice of one tile, streamed in chunks (slab-shardable)
    voxel_index      Open3D's voxel_down_sample_and_trace index rule, bit exact int64
    voxelize         per-voxel mean point / colour, fp64-mean feature -> fp16, hit counts (extract_priors.py:166-191) on the
                     ps_voxel_keys / ps_voxel_reduce kernels (stable key sort + one wavefront per voxel)
    merge_voxels     partial voxel sums of several ranks / slabs -> one voxel set (integer-key merge, no float exchange before it)"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import field_ops as F
from ._lib import check, lib
from .ops import _f32, _p, _stream


def _host3(v) -> "ctypes.Array":
    return (ctypes.c_double * 3)(*[float(x) for x in torch.as_tensor(v).double().reshape(-1).cpu().tolist()])


def voxel_index(points: Tensor, voxel: float, min_bound: Tensor) -> Tensor:
    pts = _f32(points)
    idx = torch.empty(pts.shape[0], 3, device=pts.device, dtype=torch.int64)
    check(lib().ps_voxel_index(_p(pts), pts.shape[0], float(voxel), _host3(min_bound), _p(idx), _stream()), "ps_voxel_index")
    return idx


def _host_aabb(aabb):
    """the six floats of an AABB as the host array ps_lattice_points takes (a device tensor costs a copy AND a host synchronisation:
    loops over chunks convert once)"""
    if isinstance(aabb, ctypes.Array):
        return aabb
    return (ctypes.c_float * 6)(*[float(v) for v in torch.as_tensor(aabb).reshape(-1).cpu().tolist()])


def lattice_points(aabb, res: int, start: int, count: int, device) -> Tensor:
    a = _host_aabb(aabb)
    pts = torch.empty(count, 3, device=device)
    check(lib().ps_lattice_points(a, res, start, count, _p(pts), _stream()), "ps_lattice_points")
    return pts


SHARE_ROUTING = os.environ.get("PRESIGHT_SHARED_ROUTING", "1") != "0"


def shared_routing(model) -> bool:
    """The three routed modules of a tile (two proposal fields, main field) are queried at the SAME points by the prior extraction
    (ns/scripts/extract_priors.py:133-138) and are built from the same centroids and sub-field boxes: the nearest-centroid routing, the
    sorted layout and the per-sub-field normalised points are then computed ONCE per chunk instead of once per module (the reference
    routes three times: ns/fields/PreSight/ingp_field_ms.py:97-126, prop_density_field_ms.py:90-102).  Checked by VALUE, once per state of the routing buffers."""
    mods = list(model.proposal_networks) + [model.field]
    # the verdict is cached per STATE of the buffers it was formed from (address + version of every centroid / box buffer): a later
    # load_state_dict / load_checkpoint on the same model object re-checks instead of reusing a stale "equal"
    from .fields import buffers_key

    key = buffers_key([m.centroids for m in mods if hasattr(m, "centroids")] + [f.aabb for m in mods for f in getattr(m, "fields", ())])
    cached = getattr(model, "_shared_routing_ok", None)
    if isinstance(cached, tuple) and cached[0] == key:
        return cached[1]
    ok = (SHARE_ROUTING and all(hasattr(m, "_ms") and len(getattr(m, "fields", ())) > 1 for m in mods)
          and F.MERGED_MS and F.merged_supported(*[model.field._ms()[n][0] for n in ("base", "sem", "rgb")]))
    if ok:
        ref = model.field._ms()
        for m in model.proposal_networks:
            mm = m._ms()
            ok = ok and torch.equal(m.centroids, model.field.centroids) and torch.equal(mm["aabbs"], ref["aabbs"]) and mm["contract"] == ref["contract"]
    model._shared_routing_ok = (key, bool(ok))
    return bool(ok)


@torch.no_grad()
def _query_raw(model, pts: Tensor, density_threshold: Optional[float] = None) -> Tuple[Tensor, Tensor]:
    """-> (mean density [n], semantics fp32 [n,64] as the field returns them).  With a threshold (and two proposal fields) the main
    field skips its semantic head for the 32-point tiles that cannot contain a kept point: those rows are uninitialised."""
    routed = None
    if density_threshold is not None and len(model.proposal_networks) == 2 and shared_routing(model):
        m = model.field._ms()
        lay = F.MsLayout(model.field.centroids, pos=pts.reshape(-1, 3))
        routed = (lay,) + tuple(lay.points(m["aabbs"], m["contract"]))
        dens = [p._ms_density(lay, points=routed[1:]).reshape(-1) for p in model.proposal_networks]
    else:
        dens = [p.density_fn(pts).reshape(-1) for p in model.proposal_networks]
    gate = None
    if density_threshold is not None and len(dens) == 2:
        thr = float(density_threshold)
        gate = (dens[0], dens[1], thr - 1e-5 * abs(thr) - 1e-30)  # a few ulp below the caller's own comparison
    # one pass of the main field for both (the reference makes three)
    d_main, sem = model.field.density_and_semantics(pts, gate, routed=routed) if routed is not None else model.field.density_and_semantics(pts, gate)
    dens.append(d_main.reshape(-1))
    if len(dens) == 3:
        out = torch.empty_like(dens[0])
        check(lib().ps_mean_density(_p(dens[0]), _p(dens[1]), _p(dens[2]), out.numel(), _p(out), _stream()), "ps_mean_density")
    else:
        out = torch.stack(dens, 0).mean(0)
    return out, sem.reshape(-1, sem.shape[-1])


def query_priors(model, pts: Tensor, density_threshold: Optional[float] = None):
    """-> (mean density [n], features fp16 [n,64]); pts are in the model's (scaled) frame.
    With `density_threshold`: -> (mean density [n], keep mask [n], features fp16 of the KEPT points [kept,64]) -- the clip to [0, 1]
    and the fp16 conversion (ns/scripts/extract_priors.py:136-138) are element-wise, so selecting the rows above the threshold
    first gives the same values while touching a tenth of the 2 GB of semantics per 8 M points."""
    out, sem = _query_raw(model, pts, density_threshold)
    if density_threshold is None:
        return out, sem.clip(0.0, 1.0).to(torch.float16)
    keep = out > density_threshold
    return out, keep, _kept_features(sem, torch.nonzero(keep).squeeze(1))


def _kept_features(sem: Tensor, idx: Tensor) -> Tensor:
    """sem[idx].clip(0, 1).to(float16) in one pass over the kept rows (ps_gather_clip_f16) instead of gather + clamp + convert"""
    sem = _f32(sem)
    out = torch.empty(idx.shape[0], sem.shape[1], device=sem.device, dtype=torch.float16)
    check(lib().ps_gather_clip_f16(_p(sem), _p(idx), idx.shape[0], sem.shape[1], 0.0, 1.0, _p(out), _stream()), "ps_gather_clip_f16")
    return out


def query_priors_indexed(model, pts: Tensor, density_threshold: float):
    """-> (mean density [n], row numbers of the points above the threshold [kept] (ascending), their features fp16 [kept,64]): the same
    values as query_priors with a threshold, with ONE nonzero() -- callers that also select points / densities reuse the row numbers
    (a boolean mask costs every tensor it indexes its own nonzero() and host sync)."""
    out, sem = _query_raw(model, pts, density_threshold)
    idx = torch.nonzero(out > density_threshold).squeeze(1)
    return out, idx, _kept_features(sem, idx)


# ------------------------------------------------------------------------------------------------ voxel down-sampling
def _grid_dims(points_max: Tensor, min_bound: Tensor, voxel: float) -> Tuple[int, int, int]:
    """voxel counts per axis that cover every index of voxel_index(points <= points_max)"""
    ext = (points_max.double().cpu() - (min_bound.double().cpu() - voxel / 2)) / voxel
    return tuple(int(v) + 2 for v in torch.floor(ext).tolist())


def _voxel_groups(points: Tensor, voxel: float, min_bound: Tensor, dims: Tuple[int, int, int]):
    """-> (keys of the voxels [V] ascending, order [n], starts [V], counts [V])"""
    n = points.shape[0]
    keys = torch.empty(n, device=points.device, dtype=torch.int64)
    check(lib().ps_voxel_keys(_p(points), n, float(voxel), _host3(min_bound), dims[1], dims[2], _p(keys), _stream()), "ps_voxel_keys")
    if dims[0] * dims[1] * dims[2] < (1 << 31):
        # a tile's voxel grid has a few million cells: 32-bit keys sort in half the radix passes of 64-bit ones (13 M kept points of a
        # 512^3 tile: 10 -> 5 passes of ~90 us); same order (stable, non-negative keys), the voxel keys are widened again below
        skeys, order = torch.sort(keys.to(torch.int32), stable=True)
        uniq, counts = torch.unique_consecutive(skeys, return_counts=True)
        uniq = uniq.to(torch.int64)
    else:
        skeys, order = torch.sort(keys, stable=True)
        uniq, counts = torch.unique_consecutive(skeys, return_counts=True)
    starts = torch.cumsum(counts, 0) - counts
    return uniq, order, starts, counts


def _key_to_index(keys: Tensor, dims: Tuple[int, int, int]) -> Tensor:
    ny, nz = dims[1], dims[2]
    iz = keys % nz
    iy = (keys // nz) % ny
    ix = keys // (nz * ny)
    return torch.stack([ix, iy, iz], -1)


@torch.no_grad()
def voxelize(points: Tensor, features: Tensor, colors: Optional[Tensor], voxel: float = 0.4, min_bound: Optional[Tensor] = None,
             points_max: Optional[Tensor] = None, want_sums: bool = False) -> Dict[str, Tensor]:
    """Group points by integer voxel index: -> dict(points f32 [V,3] (mean), features f16 [V,C] (fp64 mean), colors f32 [V,3],
    hits int64 [V], index int64 [V,3], key int64 [V], min_bound, dims).  Output order is by ascending key (Open3D's is
    hash-map order: compare as sets keyed by index).  min_bound defaults to the reference's `points.min(0) - 1`
    (extract_priors.py:236); ranks that will merge their results pass the SAME min_bound / points_max (and want_sums=True)."""
    pts = _f32(points)
    dev = pts.device
    n = pts.shape[0]
    C = features.shape[1]
    if min_bound is None:
        min_bound = pts.min(0).values - 1.0 if n else torch.zeros(3, device=dev)
    if points_max is None:
        points_max = pts.max(0).values if n else torch.zeros(3, device=dev)
    dims = _grid_dims(points_max, min_bound, voxel)
    feats = features.to(torch.float16).contiguous()
    cols = _f32(colors) if colors is not None else None
    uniq, order, starts, counts = _voxel_groups(pts, voxel, min_bound, dims)
    V = uniq.shape[0]
    o_pts = torch.empty(V, 3, device=dev)
    o_feat = torch.empty(V, C, device=dev, dtype=torch.float16)
    o_col = torch.empty(V, 3, device=dev) if cols is not None else None
    sums = torch.empty(V, 6 + C, device=dev, dtype=torch.float64) if want_sums else None
    check(lib().ps_voxel_reduce(_p(order), _p(starts), _p(counts), V, _p(pts), _p(feats), _p(cols), C, _p(o_pts), _p(o_feat), _p(o_col),
                                _p(sums), _stream()), "ps_voxel_reduce")
    out = {"key": uniq, "index": _key_to_index(uniq, dims), "hits": counts, "points": o_pts, "features": o_feat,
           "min_bound": min_bound.double().cpu(), "dims": dims}
    if o_col is not None:
        out["colors"] = o_col
    if sums is not None:
        out["sums"] = sums
    return out


@torch.no_grad()
def merge_voxels(parts: Sequence[Dict[str, Tensor]]) -> Dict[str, Tensor]:
    """Merge voxelize(..., want_sums=True) results that were computed with the same min_bound / dims (frame shards of several
    ranks, lattice slabs): voxels are matched by their integer key; hits and fp64 sums add exactly."""
    parts = [p for p in parts if p["key"].numel() > 0]
    if not parts:
        raise ValueError("merge_voxels: nothing to merge")
    dims = parts[0]["dims"]
    assert all(p["dims"] == dims and torch.equal(p["min_bound"], parts[0]["min_bound"]) for p in parts), "partials use different voxel grids"
    dev = parts[0]["key"].device
    keys = torch.cat([p["key"].to(dev) for p in parts])
    sums = torch.cat([p["sums"].to(dev) for p in parts])
    hits = torch.cat([p["hits"].to(dev) for p in parts])
    skeys, order = torch.sort(keys, stable=True)
    uniq, inv = torch.unique_consecutive(skeys, return_inverse=True)
    V = uniq.shape[0]
    tot = torch.zeros(V, sums.shape[1], device=dev, dtype=torch.float64).index_add_(0, inv, sums[order])
    th = torch.zeros(V, device=dev, dtype=torch.int64).index_add_(0, inv, hits[order])
    mean = tot / th[:, None].double()
    out = {"key": uniq, "index": _key_to_index(uniq, dims), "hits": th, "points": mean[:, :3].float(), "features": mean[:, 6:].half(),
           "min_bound": parts[0]["min_bound"], "dims": dims, "sums": tot}
    if any("colors" in p for p in parts):
        out["colors"] = mean[:, 3:6].float()
    return out


@torch.no_grad()
def finalize_priors(vox: dict, origin: Tensor, hit_thr_ratio: float = 0.0, dino_to_rgb: Optional[dict] = None) -> dict:
    """The `extracted_priors.pkl` payload of ns/scripts/extract_priors.py:186-208 from voxelize()'s output: voxels whose hit
    count exceeds the `hit_thr_ratio` quantile of all hit counts (numpy's linear-interpolation quantile), as numpy arrays
    {points f32 [V,3], features f16 [V,64], colors f32 [V,3], hits int64 [V], origin f32 [3]} -- the wire format read by
    occupancy/mmdet3d/datasets/prior_utils/city_prior.py:59-73.  `colors` is always present: the per-voxel mean of the members'
    PCA colours when the extraction produced them, otherwise apply_feature_colormap of the voxel feature (dino_to_rgb given)."""
    import numpy as np

    hits = vox["hits"].cpu().numpy()
    keep = hits > np.quantile(hits, hit_thr_ratio)
    out = {"points": vox["points"].cpu().numpy()[keep].astype(np.float32),
           "features": vox["features"].cpu().numpy()[keep].astype(np.float16),
           "hits": hits[keep],
           "origin": origin.detach().cpu().numpy().astype(np.float32)}
    if "colors" in vox:
        out["colors"] = vox["colors"].cpu().numpy()[keep].astype(np.float32)
    elif dino_to_rgb is not None:
        from .model import apply_feature_colormap

        out["colors"] = apply_feature_colormap(vox["features"].float(), dino_to_rgb).cpu().numpy()[keep].astype(np.float32)
    return out


def save_priors(path: str, priors: dict) -> None:
    """pickle.dump of finalize_priors()'s dict (extract_priors.py:199-208)"""
    import pickle

    with open(path, "wb") as f:
        pickle.dump(priors, f)


# ------------------------------------------------------------------------------------------------ the reference's extraction loop
@torch.no_grad()
def frame_hit_points(model, cameras: dict, camera_idx: int, pose_scale_factor: float, dino_to_rgb: Optional[dict],
                     camera_scaling_factor: float = 1.0, max_depth: float = 50.0, min_depth: float = 0.5, depth_type: str = "depth",
                     coords: Optional[Tensor] = None, debug: Optional[dict] = None):
    """One iteration of the reference's frame loop (extract_priors.py:99-145): rays of camera `camera_idx` (all pixels of the
    rescaled image, or the integer (row, col) `coords` that survive the segmentation mask) -> depth march
    (get_depth_for_camera_ray_bundle) -> world points `origin / scale + dir * depth / scale` -> keep min_depth < depth <
    max_depth and -3 < z < 6 -> mean density of the three fields, clipped fp16 semantics, PCA colours.
    cameras: dict(c2w [C,3,4], fx, fy, cx, cy [C], H, W) in the model's frame.  -> (world points [n,3] in metres, densities [n],
    features f16 [n,64], colours [n,3] | None)"""
    from . import ops
    from .model import apply_feature_colormap
    from .rays import RayBundle

    dev = model.device
    s = float(camera_scaling_factor)
    # Cameras.rescale_output_resolution (ns/cameras/cameras.py:953-958, called at extract_priors.py:90): intrinsics * s,
    # image size = trunc(size * s) evaluated in float32
    s32 = torch.tensor([s], dtype=torch.float32)
    H = int((torch.tensor(int(cameras["H"])) * s32).to(torch.int64))
    W = int((torch.tensor(int(cameras["W"])) * s32).to(torch.int64))
    sx = sy = s32.to(dev)
    if coords is None:
        rows, cols = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
        coords = torch.stack([rows.reshape(-1), cols.reshape(-1)], -1)
    if coords.shape[0] == 0:
        return None
    ri = torch.cat([torch.full((coords.shape[0], 1), int(camera_idx), device=dev, dtype=torch.int64), coords.to(dev).long()], -1)
    o, d, pa, dn = ops.generate_rays(ri, cameras["c2w"].to(dev), cameras["fx"].to(dev) * sx, cameras["fy"].to(dev) * sy,
                                     cameras["cx"].to(dev) * sx, cameras["cy"].to(dev) * sy)
    rb = RayBundle(o, d, pa, camera_indices=ri[:, 0:1], metadata={"directions_norm": dn})
    out = model.get_depth_for_camera_ray_bundle(rb)
    depth = out[depth_type] / pose_scale_factor
    world = (o / pose_scale_factor + d * depth).view(-1, 3)
    depth = depth.flatten()
    keep = (depth < max_depth) & (depth > min_depth) & (world[:, 2] > -3.0) & (world[:, 2] < 6.0)
    if debug is not None:
        debug.update(raw_depth=depth, sel=keep, world_all=world)
    world = world[keep]
    if world.shape[0] == 0:
        return None
    dens, feats = query_priors(model, world * pose_scale_factor)
    colors = apply_feature_colormap(feats, dino_to_rgb) if dino_to_rgb is not None else None
    return world, dens, feats, colors


@torch.no_grad()
def extract_voxels(model, cameras: dict, camera_indices: Sequence[int], pose_scale_factor: float, origin: Tensor,
                   dino_to_rgb: Optional[dict] = None, camera_scaling_factor: float = 1.0, voxel_size: float = 0.4, max_depth: float = 50.0,
                   min_depth: float = 0.5, hit_thr_ratio: float = 0.2, depth_type: str = "depth", density_threshold: float = 1.0,
                   masks: Optional[Dict[int, Tensor]] = None, group=None) -> Optional[dict]:
    """ns/scripts/extract_priors.py:34-208 on the HIP path -> the extracted_priors.pkl payload (finalize_priors), plus the
    intermediate voxel set under "_voxels".  masks[camera_idx] = bool [H,W] of valid pixels (the segmentation mask of :100-106).
    With torch.distributed initialised the frames are dealt round-robin to the ranks; every rank voxelises its own hit points
    on the COMMON grid (min over ranks of the points' minimum, one 6-float all-reduce), the partial voxel sums are gathered and
    merged by integer key on rank 0 (the other ranks return None)."""
    import torch.distributed as dist

    model.eval()
    world_size = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world_size > 1 else 0
    dev = model.device
    P, D, Fe, Co = [], [], [], []
    for i, cam in enumerate(camera_indices):
        if i % world_size != rank:
            continue
        coords = None if masks is None else torch.nonzero(masks[cam].to(dev))
        hit = frame_hit_points(model, cameras, cam, pose_scale_factor, dino_to_rgb, camera_scaling_factor, max_depth, min_depth, depth_type,
                               coords)
        if hit is None:
            continue
        P.append(hit[0]); D.append(hit[1]); Fe.append(hit[2])  # noqa: E702
        if hit[3] is not None:
            Co.append(hit[3])
    pts = torch.cat(P) if P else torch.zeros(0, 3, device=dev)
    dens = torch.cat(D) if D else torch.zeros(0, device=dev)
    feats = torch.cat(Fe) if Fe else torch.zeros(0, 64, device=dev, dtype=torch.float16)
    cols = torch.cat(Co) if Co else None
    sel = dens > density_threshold  # extract_priors.py:152
    pts, feats = pts[sel], feats[sel]
    cols = cols[sel] if cols is not None else None
    big = 3.0e38
    lo = pts.min(0).values if pts.shape[0] else torch.full((3,), big, device=dev)
    hi = pts.max(0).values if pts.shape[0] else torch.full((3,), -big, device=dev)
    if world_size > 1:
        t = torch.cat([-lo, hi])
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        lo, hi = -t[:3], t[3:]
    if float(hi[0]) < float(lo[0]):
        return None  # no point survived on any rank
    vox = voxelize(pts, feats, cols, voxel=voxel_size, min_bound=lo - 1.0, points_max=hi, want_sums=True)
    if world_size > 1:
        part = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in vox.items()}
        gathered: List[Optional[dict]] = [None] * world_size
        dist.all_gather_object(gathered, part, group=group)
        if rank != 0:
            return None
        vox = merge_voxels([g for g in gathered if g is not None and g["key"].numel() > 0])
    pri = finalize_priors(vox, origin, hit_thr_ratio, dino_to_rgb)
    pri["_voxels"] = vox
    return pri


# ------------------------------------------------------------------------------------------------ BASELINE config 5: dense lattice
@torch.no_grad()
def dense_tile_query(model, aabb: Tensor, res: int = 512, chunk: int = 1 << 22, start: int = 0, count: Optional[int] = None,
                     density_threshold: float = 1.0, voxel: float = 0.4, pose_scale_factor: float = 0.05) -> Dict[str, Tensor]:
    """Evaluate the prior fields on lattice points [start, start+count) of the res^3 lattice over `aabb` (model frame).
    Ranks shard the lattice by giving each one a contiguous [start, count) slab; there is no exchange until the final
    integer-key merge, and the voxel grid's origin is derived from the TILE's AABB (not from the slab's own points), so every
    slab computes the same integer index for the same voxel.  Returns the points above the density threshold (metres) with
    their features, densities and integer voxel index."""
    dev = model.device
    total = res ** 3 if count is None else count
    min_bound = aabb.reshape(2, 3)[0].double().cpu() / pose_scale_factor - 1.0
    points_max = aabb.reshape(2, 3)[1].double().cpu() / pose_scale_factor
    if not FUSED_SELECT:
        return _dense_tile_query_synced(model, aabb, res, chunk, start, total, density_threshold, voxel, pose_scale_factor, min_bound, points_max)
    # The selection of every chunk APPENDS to tile-wide arrays at a position kept in device memory (ps_emit_kept): no nonzero(), no host
    # synchronisation inside the loop -- the host enqueues all chunks back to back and reads the row count once at the end.  The arrays
    # hold `capacity` rows; a tile that keeps more (cursor[1] > 0) is simply run again with room for all of them.
    capacity = max(1 << 16, int(KEEP_FRACTION_GUESS * total) + 4096)
    mb = _host3(min_bound)
    aabb_h = _host_aabb(aabb)  # (once per tile: read per chunk from a device tensor it was a host synchronisation per chunk -- the host
    #                             never ran ahead of the GPU, 4 ms of launch gaps per 512^3 pass)
    while True:
        o_pts, o_dens = torch.empty(capacity, 3, device=dev), torch.empty(capacity, device=dev)
        o_feat = torch.empty(capacity, 64, device=dev, dtype=torch.float16)
        o_vox = torch.empty(capacity, 3, device=dev, dtype=torch.int64)
        cursor = torch.zeros(2, device=dev, dtype=torch.int64)
        ws = torch.empty(lib().ps_emit_kept_workspace(min(chunk, total)), device=dev, dtype=torch.uint8)
        for s in range(start, start + total, chunk):
            n = min(chunk, start + total - s)
            pts = lattice_points(aabb_h, res, s, n, dev)
            with torch.no_grad():
                dens, sem = _query_raw(model, pts, density_threshold)
            sem = _f32(sem)
            if sem.shape[1] != 64:
                raise NotImplementedError("dense_tile_query: 64 semantic channels (PRESIGHT_FUSED_SELECT=0 for other widths)")
            check(lib().ps_emit_kept(_p(dens), n, float(density_threshold), _p(pts), _p(sem), 64, float(pose_scale_factor), float(voxel), mb, s,
                                     _p(cursor), capacity, _p(ws), _p(o_pts), _p(o_dens), _p(o_feat), _p(o_vox), None, _stream()), "ps_emit_kept")
        kept, dropped = (int(v) for v in cursor.tolist())  # the ONE host synchronisation of the tile
        if dropped == 0:
            break
        capacity = int(1.1 * kept) + 4096
    return {"points": o_pts[:kept], "features": o_feat[:kept], "densities": o_dens[:kept], "voxel_index": o_vox[:kept],
            "min_bound": min_bound, "points_max": points_max}


# rows kept per lattice point the output arrays of dense_tile_query are first sized for (the reference keeps density > 1.0: a few percent
# of a trained tile; bench.py's threshold keeps 10 %); more -> one re-run with the exact size
KEEP_FRACTION_GUESS = float(os.environ.get("PRESIGHT_KEEP_FRACTION_GUESS", "0.15"))
FUSED_SELECT = os.environ.get("PRESIGHT_FUSED_SELECT", "1") != "0"


def _dense_tile_query_synced(model, aabb, res, chunk, start, total, density_threshold, voxel, pose_scale_factor, min_bound, points_max):
    """the loop of rounds 2-5: one nonzero() (host sync) per chunk, then gather / select / voxel-index launches on the kept rows"""
    dev = model.device
    keep_pts, keep_feat, keep_dens, keep_idx = [], [], [], []
    aabb = _host_aabb(aabb)
    for s in range(start, start + total, chunk):
        n = min(chunk, start + total - s)
        pts = lattice_points(aabb, res, s, n, dev)
        dens, m, feats = query_priors_indexed(model, pts, density_threshold)
        P = pts.index_select(0, m) / pose_scale_factor
        keep_pts.append(P)
        keep_feat.append(feats)
        keep_dens.append(dens.index_select(0, m))
        keep_idx.append(voxel_index(P, voxel, min_bound))
    cat = lambda xs, empty: torch.cat(xs) if xs else empty  # noqa: E731
    return {"points": cat(keep_pts, torch.zeros(0, 3, device=dev)), "features": cat(keep_feat, torch.zeros(0, 64, device=dev, dtype=torch.float16)),
            "densities": cat(keep_dens, torch.zeros(0, device=dev)), "voxel_index": cat(keep_idx, torch.zeros(0, 3, device=dev, dtype=torch.int64)),
            "min_bound": min_bound, "points_max": points_max}
