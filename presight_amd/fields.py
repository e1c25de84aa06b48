"""Fields of the PreSight model on the HIP kernels; same class names, constructor arguments, method names and
state-dict keys as the reference:

    iNGPField / iNGPFieldMS                ns/fields/PreSight/ingp_field.py:47-267, ingp_field_ms.py:46-185
    PropNetDensityField / ...FieldMS       ns/fields/PreSight/prop_density_field.py:38-156, prop_density_field_ms.py:46-105
    SkyField / SkyFieldMS                  ns/fields/PreSight/sky_field.py:40-120, sky_field_ms.py:47-117

Two call levels exist.  The plugin-surface methods (`density_fn`, `get_outputs`, `forward`, `semantic_fn`) keep the
reference's signatures.  `forward`, `semantic_fn` and the proposal `density_fn` run the fused field kernels
(field_ops); `iNGPField.density_fn` / `get_outputs`, which expose the intermediate 79-d embedding, are composed from
the operator-level kernels.  The *MS routers replace the reference's 16 masked python loops + host syncs by one
nearest-centroid kernel and (for K > 1) gather / scatter around the per-sub-field fused calls."""
from __future__ import annotations

from copy import deepcopy
from enum import Enum
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import field_ops as F
from . import ops
from .components import MLP, HashEncoding, SHEncoding, trunc_exp
from .rays import RaySamples


class FieldHeadNames(Enum):
    """ns/field_components/field_heads.py:30-42"""
    RGB = "rgb"
    SH = "sh"
    DENSITY = "density"
    NORMALS = "normals"
    PRED_NORMALS = "pred_normals"
    UNCERTAINTY = "uncertainty"
    BACKGROUND_RGB = "background_rgb"
    TRANSIENT_RGB = "transient_rgb"
    TRANSIENT_DENSITY = "transient_density"
    SEMANTICS = "semantics"
    SDF = "sdf"
    ALPHA = "alpha"
    GRADIENT = "gradient"


def get_normalized_directions(directions: Tensor) -> Tensor:
    """ns/fields/base_field.py:136-142"""
    return (directions + 1.0) / 2.0


def _same_grid(a: HashEncoding, b: HashEncoding) -> bool:
    return ((a.num_levels, a.features_per_level, a.log2_hashmap_size) == (b.num_levels, b.features_per_level, b.log2_hashmap_size)
            and torch.equal(a.scalings, b.scalings))


def _grid_cfg(enc: HashEncoding) -> F.GridCfg:
    return F.GridCfg(enc.num_levels, enc.features_per_level, enc.log2_hashmap_size)


def points_key(field) -> tuple:
    """identity of a field's point normalisation (its box buffer's storage + version, contraction on / off)"""
    return (field.aabb.data_ptr(), field.aabb._version, field.spatial_distortion is not None)


def points_of(field, ray_samples: RaySamples) -> Tuple[Tensor, Tensor]:
    """(u, sel) of `field` on the samples: the ones the sampling kernel produced for this very field on the way, else computed here"""
    pts = getattr(ray_samples, "points", None)
    if pts is not None and pts[0] == points_key(field):
        return pts[1], pts[2]
    rb = ray_samples.ray_bundle
    return field.points(origins=rb.origins, dirs=rb.directions, ebins=ray_samples.ebins)


def points_spec(field) -> tuple:
    """what a sampling kernel needs to form this field's points: (aabb, contract, key)"""
    return (field.aabb, field.spatial_distortion is not None, points_key(field))


# ------------------------------------------------------------------------------------------------------------ main
class iNGPField(nn.Module):
    def __init__(self, aabb: Tensor, num_layers: int = 2, hidden_dim: int = 64, geo_feat_dim: int = 15, num_levels: int = 16,
                 base_res: int = 16, max_res: int = 2048, log2_hashmap_size: int = 19, num_layers_color: int = 3,
                 features_per_level: int = 2, hidden_dim_color: int = 64, appearance_embedding_dim: int = 32,
                 use_semantics: bool = False, hidden_dim_semantic_head: int = 64, semantic_dim: int = 64,
                 spatial_distortion: Optional[nn.Module] = None, implementation: str = "tcnn+fp32", field_type: str = "iNGP",
                 **kwargs) -> None:
        super().__init__()
        if field_type != "iNGP":
            raise ValueError(f"Unknown `field_type`: {field_type}")
        self.register_buffer("aabb", deepcopy(aabb))
        self.geo_feat_dim = geo_feat_dim
        self.register_buffer("max_res", torch.tensor(max_res))
        self.register_buffer("num_levels", torch.tensor(num_levels))
        self.register_buffer("log2_hashmap_size", torch.tensor(log2_hashmap_size))
        self.spatial_distortion = spatial_distortion
        self.appearance_embedding_dim = appearance_embedding_dim
        self.use_semantics = use_semantics
        self.semantic_dim = semantic_dim if use_semantics else 0
        self.base_res = base_res
        self.direction_encoding = SHEncoding(levels=4, implementation=implementation)
        self.mlp_base_grid = HashEncoding(num_levels=num_levels, min_res=base_res, max_res=max_res,
                                          log2_hashmap_size=log2_hashmap_size, features_per_level=features_per_level,
                                          implementation=implementation)
        self.mlp_base_mlp = MLP(in_dim=self.mlp_base_grid.get_out_dim(), num_layers=num_layers, layer_width=hidden_dim,
                                out_dim=1 + self.geo_feat_dim + self.semantic_dim, activation=nn.ReLU(), out_activation=None)
        self.mlp_base = torch.nn.Sequential(self.mlp_base_grid, self.mlp_base_mlp)  # same aliasing as the reference
        if self.use_semantics:
            self.semantic_head = MLP(in_dim=self.semantic_dim, num_layers=3, layer_width=hidden_dim_semantic_head,
                                     out_dim=semantic_dim, activation=nn.ReLU(), out_activation=None)
        self.rgb_head = MLP(in_dim=self.direction_encoding.get_out_dim() + self.geo_feat_dim + self.appearance_embedding_dim,
                            num_layers=num_layers_color, layer_width=hidden_dim_color, out_dim=3, activation=nn.ReLU(),
                            out_activation=nn.Sigmoid())
        self._fusable = (use_semantics and semantic_dim == 64 and geo_feat_dim == 15 and num_layers == 2 and num_layers_color == 3
                         and hidden_dim_semantic_head == 64 and appearance_embedding_dim <= 16)

    # -- fused fast path -------------------------------------------------------------------------------------
    def _require_fused(self):
        if not self._fusable:
            raise NotImplementedError("presight_amd iNGPField: the fused kernels cover the PreSight layout (2-layer base MLP "
                                      "-> 1+15+64, 3-layer 64-wide semantic head, 3-layer colour head, app dim <= 16)")

    def points(self, pos=None, origins=None, dirs=None, ebins=None):
        return F.field_points(self.aabb, self.spatial_distortion is not None, pos=pos, origins=origins, dirs=dirs, ebins=ebins)

    def evaluate(self, u: Tensor, sel: Tensor, ray_dirs: Optional[Tensor], app: Optional[Tensor], S: int, want_rgb: bool = True,
                 want_sem: bool = True):
        """Fused field evaluation on prepared points: -> (density [N], rgb [N,3], semantics [N,64])."""
        self._require_fused()
        g = self.mlp_base_grid
        return F.main_field(u, sel, ray_dirs, app, S, g.hash_table, g.scalings_on(u.device), _grid_cfg(g),
                            self.mlp_base_mlp.layer_params(), self.semantic_head.layer_params(), self.rgb_head.layer_params(),
                            want_rgb=want_rgb, want_sem=want_sem)

    def render(self, ray_samples: RaySamples, appearance_embedding: Optional[Tensor], threshold: float = 0.5, want_depth: bool = True):
        """forward() + RaySamples.get_weights + the RGB / accumulation / depth / semantics renderers as one autograd node
        (field_ops.main_field_render): -> (rgb, accumulation (unclamped), threshold depth, expected depth, semantics, weights
        [R,S,1]).  want_depth=False: the node may skip the two depth renderers (they come back as None; render them from the weights
        when needed, renderers.render_all) -- a training step never reads them."""
        self._require_fused()
        rb = ray_samples.ray_bundle
        R = ray_samples.ebins.shape[0]
        u, sel = points_of(self, ray_samples)
        app = None if appearance_embedding is None else _per_ray(appearance_embedding, R)
        g = self.mlp_base_grid
        rgb, acc, depth, expd, sem, w = F.main_field_render(
            u, sel, rb.directions, app, ray_samples.ebins, g.hash_table, g.scalings_on(u.device), _grid_cfg(g),
            self.mlp_base_mlp.layer_params(), self.semantic_head.layer_params(), self.rgb_head.layer_params(), threshold, want_depth)
        return rgb, acc, depth, expd, sem, w[..., None]

    def can_render(self, ray_samples: RaySamples) -> bool:
        return self._fusable and ray_samples.num_samples <= 64

    # -- reference plugin surface ------------------------------------------------------------------------------
    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, Tensor]:
        return self.density_fn(ray_samples.frustums.get_positions())

    def density_fn(self, positions: Tensor, times=None) -> Tuple[Tensor, Tensor]:
        """-> (density [*bs,1], embedding [*bs, geo+sem]); composed from operator-level kernels because the embedding is
        an API output here (ns/fields/PreSight/ingp_field.py:168-191)."""
        flat = positions.reshape(-1, 3)
        u, sel = ops.contract(flat, self.aabb, self.spatial_distortion is not None)
        h = self.mlp_base_mlp(self.mlp_base_grid(u))
        raw, emb = torch.split(h, [1, self.geo_feat_dim + self.semantic_dim], dim=-1)
        self._density_before_activation = raw
        density = trunc_exp(raw) * sel[:, None]
        return density.view(*positions.shape[:-1], 1), emb.reshape(*positions.shape[:-1], -1)

    def get_outputs(self, directions: Tensor, density_embedding: Tensor, appearance_embedding: Optional[Tensor]):
        """ns/fields/PreSight/ingp_field.py:193-237"""
        assert density_embedding is not None
        outputs = {}
        shape = directions.shape[:-1]
        if self.use_semantics:
            density_embedding, sem_emb = torch.split(density_embedding, [self.geo_feat_dim, self.semantic_dim], dim=-1)
            outputs[FieldHeadNames.SEMANTICS] = self.semantic_head(sem_emb.reshape(-1, self.semantic_dim)).view(*shape, -1)
        d = self.direction_encoding(get_normalized_directions(directions).reshape(-1, 3))
        parts = [d, density_embedding.reshape(-1, self.geo_feat_dim)]
        if appearance_embedding is not None:
            parts.append(appearance_embedding.reshape(-1, self.appearance_embedding_dim))
        outputs[FieldHeadNames.RGB] = self.rgb_head(torch.cat(parts, dim=-1)).view(*shape, 3)
        return outputs

    def forward(self, ray_samples: RaySamples, appearance_embedding=None) -> Dict[FieldHeadNames, Tensor]:
        rb = ray_samples.ray_bundle
        R, S = ray_samples.ebins.shape[0], ray_samples.num_samples
        u, sel = points_of(self, ray_samples)
        app = None if appearance_embedding is None else _per_ray(appearance_embedding, R)
        sigma, rgb, sem = self.evaluate(u, sel, rb.directions, app, S)
        return {FieldHeadNames.DENSITY: sigma.view(R, S, 1), FieldHeadNames.RGB: rgb.view(R, S, 3),
                FieldHeadNames.SEMANTICS: sem.view(R, S, -1)}

    def semantic_fn(self, positions: Tensor) -> Tensor:
        assert self.use_semantics, "Cannot query semantics when `self.use_semantics` is set to False"
        u, sel = self.points(pos=positions.reshape(-1, 3))
        _, _, sem = self.evaluate(u, sel, None, None, 1, want_rgb=False, want_sem=True)
        return sem.view(*positions.shape[:-1], -1)


def _per_ray(appearance_embedding: Tensor, R: int) -> Tensor:
    """[R, A] appearance code of a ray from the reference's per-sample layout [R, S or 1, A] (all S copies are equal).  A pure
    view when the middle dimension is 1 -- `x[:, 0]` would cost a zero-fill + copy (select_backward) in every backward."""
    a = appearance_embedding.reshape(R, -1, appearance_embedding.shape[-1])
    return a.reshape(R, a.shape[-1]) if a.shape[1] == 1 else a[:, 0]


def buffers_key(tensors: Sequence[Tensor]) -> tuple:
    """identity AND content version of a set of buffers: load_state_dict / load_checkpoint copy into the same storage (the address stays,
    `_version` advances), .to(device) replaces the tensors -- either way a cache keyed on this is refreshed"""
    return tuple((t.data_ptr(), t._version, str(t.device)) for t in tensors)


def _stacked_aabbs(module, dev) -> Tensor:
    """[K, 2, 3] sub-field boxes of a routed module on `dev`, rebuilt whenever a sub-field's `aabb` buffer was written or moved"""
    key = (str(dev), buffers_key([f.aabb for f in module.fields]))
    if getattr(module, "_aabbs_key", None) != key:
        module._aabbs_cache = torch.stack([f.aabb.float() for f in module.fields]).to(dev).contiguous()
        module._aabbs_key = key
    return module._aabbs_cache


def _route_groups(points: Tensor, centroids: Tensor) -> List[Tuple[int, Tensor]]:
    """nearest-centroid router (ns/fields/PreSight/ingp_field_ms.py:97): [(k, indices of the points of sub-field k)]."""
    assign = ops.route(points, centroids)
    order = torch.argsort(assign, stable=True)
    counts = torch.bincount(assign, minlength=centroids.shape[0]).tolist()  # one host sync instead of K torch.any()
    groups, start = [], 0
    for k, c in enumerate(counts):
        if c > 0:
            groups.append((k, order[start:start + c]))
        start += c
    return groups


def routed_apply(points: Tensor, centroids: Tensor, fn, extras: Sequence[Tensor] = ()) -> List[Tensor]:
    """Evaluate the sub-fields on their points with ONE gather in and ONE scatter out instead of a masked gather + masked
    scatter per sub-field and output (ns/fields/PreSight/ingp_field_ms.py:97-126 does 4*K masked index ops and K host syncs):
    the points (and any per-point `extras`) are sorted by sub-field once, every sub-field works on a contiguous slice,
    the outputs are concatenated in sorted order and un-sorted with a single index_copy.
    fn(k, points_slice, *extras_slices) -> tuple of [n_k, w] tensors."""
    assign = ops.route(points, centroids)
    order = torch.argsort(assign, stable=True)
    counts = torch.bincount(assign, minlength=centroids.shape[0]).tolist()  # the one host sync of the router
    pts = points[order]
    ext = [e[order] for e in extras]
    outs: Optional[List[List[Tensor]]] = None
    start = 0
    for k, c in enumerate(counts):
        if c == 0:
            continue
        vals = fn(k, pts[start:start + c], *[e[start:start + c] for e in ext])
        if outs is None:
            outs = [[] for _ in vals]
        for o, v in zip(outs, vals):
            o.append(v.reshape(c, -1))
        start += c
    if outs is None:  # no points at all: probe the output widths with an empty slice of sub-field 0
        return [v.reshape(0, -1) for v in fn(0, pts[:0], *[e[:0] for e in ext])]
    res = []
    for o in outs:
        cat = o[0] if len(o) == 1 else torch.cat(o, dim=0)
        res.append(torch.empty_like(cat).index_copy_(0, order, cat))  # un-sort: row order[i] <- sorted row i
    return res


class iNGPFieldMS(nn.Module):
    def __init__(self, fields: List[iNGPField], centroids: Tensor) -> None:
        super().__init__()
        self.register_buffer("centroids", deepcopy(centroids))
        self.fields = nn.ModuleList(fields)

    def _routed(self, positions: Tensor, fn, widths: List[int], extras: Sequence[Tensor] = ()) -> List[Tensor]:
        """fn(field, positions_slice, *extras_slices) -> tuple of per-point tensors (widths only documents the outputs)"""
        return routed_apply(positions.reshape(-1, 3), self.centroids, lambda k, pos, *ex: fn(self.fields[k], pos, *ex), extras)

    # -- all K sub-fields in one launch per kernel (field_ops "MS" path, csrc/ms_core.hpp) ----------------------------
    def _ms(self):
        """per-sub-field tables / layers / AABBs of the fused MS kernels (the sub-fields of a tile share one configuration)"""
        f0 = self.fields[0]
        f0._require_fused()
        g0 = f0.mlp_base_grid
        if not getattr(self, "_ms_checked", False):
            for f in self.fields[1:]:
                if not _same_grid(f.mlp_base_grid, g0) or not f._fusable:
                    raise NotImplementedError("presight_amd iNGPFieldMS: the sub-fields of a tile must share one configuration")
            self._ms_checked = True
        dev = g0.hash_table.device
        aabbs = _stacked_aabbs(self, dev)
        return dict(aabbs=aabbs, contract=f0.spatial_distortion is not None, tables=[f.mlp_base_grid.hash_table for f in self.fields],
                    scalings=g0.scalings_on(dev), g=_grid_cfg(g0), base=[f.mlp_base_mlp.layer_params() for f in self.fields],
                    sem=[f.semantic_head.layer_params() for f in self.fields], rgb=[f.rgb_head.layer_params() for f in self.fields])

    def _ms_points(self, positions: Tensor, want_rgb=False, want_sem=True):
        m = self._ms()
        lay = F.MsLayout(self.centroids, pos=positions.reshape(-1, 3))
        u, sel = lay.points(m["aabbs"], m["contract"])
        return F.ms_main_field(lay, u, sel, None, None, 1, m["tables"], m["scalings"], m["g"], m["base"], m["sem"], m["rgb"],
                               want_rgb=want_rgb, want_sem=want_sem)

    def can_render(self, ray_samples: RaySamples) -> bool:
        return all(f._fusable for f in self.fields) and ray_samples.num_samples <= 64

    def render(self, ray_samples: RaySamples, appearance_embedding: Optional[Tensor], threshold: float = 0.5, want_depth: bool = True):
        """iNGPField.render for the routed tile: (rgb, accumulation, threshold depth, expected depth, semantics, weights [R,S,1]);
        want_depth=False only concerns the one-sub-field node (iNGPField.render), the routed node always renders the depths"""
        if len(self.fields) == 1:
            return self.fields[0].render(ray_samples, appearance_embedding, threshold, want_depth)
        rb = ray_samples.ray_bundle
        R = ray_samples.ebins.shape[0]
        app = None if appearance_embedding is None else _per_ray(appearance_embedding, R)
        m = self._ms()
        lay = F.MsLayout(self.centroids, origins=rb.origins, dirs=rb.directions, ebins=ray_samples.ebins)
        u, sel = lay.points(m["aabbs"], m["contract"])
        rgb, acc, depth, expd, sem, w = F.ms_main_field_render(lay, u, sel, rb.directions, app, ray_samples.ebins, m["tables"], m["scalings"],
                                                               m["g"], m["base"], m["sem"], m["rgb"], threshold)
        return rgb, acc, depth, expd, sem, w[..., None]

    def forward(self, ray_samples: RaySamples, appearance_embedding: Optional[Tensor]) -> Dict[FieldHeadNames, Tensor]:
        R, S = ray_samples.ebins.shape[0], ray_samples.num_samples
        rb = ray_samples.ray_bundle
        if len(self.fields) == 1:
            return self.fields[0](ray_samples, appearance_embedding)
        app = None if appearance_embedding is None else _per_ray(appearance_embedding, R)
        m = self._ms()
        lay = F.MsLayout(self.centroids, origins=rb.origins, dirs=rb.directions, ebins=ray_samples.ebins)
        u, sel = lay.points(m["aabbs"], m["contract"])
        sigma, rgb, sem = F.ms_main_field(lay, u, sel, rb.directions, app, S, m["tables"], m["scalings"], m["g"], m["base"], m["sem"], m["rgb"])
        return {FieldHeadNames.DENSITY: sigma.view(R, S, 1), FieldHeadNames.RGB: rgb.view(R, S, 3),
                FieldHeadNames.SEMANTICS: sem.view(R, S, -1)}

    def density_fn(self, positions: Tensor) -> Tuple[Tensor, Tensor]:
        if len(self.fields) == 1:
            return self.fields[0].density_fn(positions)
        w = self.fields[0].geo_feat_dim + self.fields[0].semantic_dim
        d, e = self._routed(positions, lambda f, pos: f.density_fn(pos), [1, w])
        return d.view(*positions.shape[:-1], 1), e.view(*positions.shape[:-1], -1)

    def density_only(self, positions: Tensor) -> Tensor:
        """density [*bs,1] through the fused kernel (heads skipped); used by get_depth and prior extraction."""
        def run(f: iNGPField, pos):
            u, sel = f.points(pos=pos)
            return (f.evaluate(u, sel, None, None, 1, want_rgb=False, want_sem=False)[0],)

        if len(self.fields) == 1:
            (d,) = run(self.fields[0], positions.reshape(-1, 3))
        else:
            d = self._ms_points(positions, want_sem=False)[0]
        return d.view(*positions.shape[:-1], 1)

    def density_and_semantics(self, positions: Tensor, gate: Optional[Tuple[Tensor, Tensor, float]] = None,
                              routed: Optional[Tuple["F.MsLayout", Tensor, Tensor]] = None) -> Tuple[Tensor, Tensor]:
        """(density [*bs,1], semantics [*bs,64]) from ONE evaluation of the field.  The reference's prior extraction calls
        density_fn and semantic_fn separately (ns/scripts/extract_priors.py:133-138) and semantic_fn re-runs density_fn
        (ingp_field.py:256): the hash encode and the base MLP run three times per point there, once here.
        gate = (density_a, density_b, threshold) (inference; one sub-field or the routed tile): the semantic head is skipped for the 32-point tiles in
        which no point reaches mean(density_a, density_b, density) >= threshold -- those rows of the semantics are uninitialised
        (field_ops.main_field_gated)."""
        def run(f: iNGPField, pos):
            u, sel = f.points(pos=pos)
            if gate is not None and not torch.is_grad_enabled():
                f._require_fused()
                gr = f.mlp_base_grid
                return F.main_field_gated(u, sel, gr.hash_table, gr.scalings_on(u.device), _grid_cfg(gr), f.mlp_base_mlp.layer_params(),
                                          f.semantic_head.layer_params(), f.rgb_head.layer_params(), gate[0], gate[1], gate[2])
            d, _, s = f.evaluate(u, sel, None, None, 1, want_rgb=False, want_sem=True)
            return d, s

        if len(self.fields) == 1:
            d, s = run(self.fields[0], positions.reshape(-1, 3))
        elif gate is not None and not torch.is_grad_enabled() and F.MERGED_MS and F.merged_supported(*[self._ms()[n][0] for n in ("base", "sem", "rgb")]):
            m = self._ms()
            if routed is not None:  # (routing + normalised points of THESE positions, shared with the proposal fields: extract.shared_routing)
                lay, u, sel = routed
            else:
                lay = F.MsLayout(self.centroids, pos=positions.reshape(-1, 3))
                u, sel = lay.points(m["aabbs"], m["contract"])
            d, s = F.ms_main_field_gated(lay, u, sel, m["tables"], m["scalings"], m["g"], m["base"], m["sem"], m["rgb"], gate[0], gate[1], gate[2])
        else:
            d, _, s = self._ms_points(positions)
        return d.view(*positions.shape[:-1], 1), s.view(*positions.shape[:-1], -1)

    def get_density(self, ray_samples: RaySamples):
        return self.density_fn(ray_samples.frustums.get_positions())

    def semantic_fn(self, positions: Tensor) -> Tensor:
        if len(self.fields) == 1:
            return self.fields[0].semantic_fn(positions)
        s = self._ms_points(positions)[2]
        return s.view(*positions.shape[:-1], -1)


# ------------------------------------------------------------------------------------------------------------ proposal
class PropNetDensityField(nn.Module):
    def __init__(self, aabb: Tensor, num_layers: int = 2, hidden_dim: int = 64, spatial_distortion: Optional[nn.Module] = None,
                 use_linear: bool = False, num_levels: int = 8, max_res: int = 1024, base_res: int = 16,
                 log2_hashmap_size: int = 18, features_per_level: int = 2, implementation: str = "tcnn+fp32",
                 field_type: str = "iNGP") -> None:
        super().__init__()
        if field_type != "iNGP":
            raise ValueError(f"Unknown `field_type`: {field_type}")
        if use_linear or num_layers != 2:
            raise NotImplementedError("presight_amd PropNetDensityField: PreSight uses the 2-layer MLP variant")
        self.register_buffer("aabb", deepcopy(aabb))
        self.spatial_distortion = spatial_distortion
        self.use_linear = use_linear
        self.register_buffer("max_res", torch.tensor(max_res))
        self.register_buffer("num_levels", torch.tensor(num_levels))
        self.register_buffer("log2_hashmap_size", torch.tensor(log2_hashmap_size))
        self.encoding = HashEncoding(num_levels=num_levels, min_res=base_res, max_res=max_res,
                                     log2_hashmap_size=log2_hashmap_size, features_per_level=features_per_level,
                                     implementation=implementation)
        network = MLP(in_dim=self.encoding.get_out_dim(), num_layers=num_layers, layer_width=hidden_dim, out_dim=1,
                      activation=nn.ReLU(), out_activation=None)
        self.mlp_base = torch.nn.Sequential(self.encoding, network)

    def points(self, pos=None, origins=None, dirs=None, ebins=None):
        return F.field_points(self.aabb, self.spatial_distortion is not None, pos=pos, origins=origins, dirs=dirs, ebins=ebins)

    def evaluate(self, u: Tensor, sel: Tensor) -> Tensor:
        e = self.encoding
        return F.prop_field(u, sel, e.hash_table, e.scalings_on(u.device), _grid_cfg(e), self.mlp_base[1].layer_params())

    def density_fn(self, positions: Tensor) -> Tensor:
        """ns/fields/PreSight/prop_density_field.py:129-153 -> [*bs,1]"""
        u, sel = self.points(pos=positions.reshape(-1, 3))
        return self.evaluate(u, sel).view(*positions.shape[:-1], 1)

    def density_of_samples(self, ray_samples: RaySamples) -> Tensor:
        u, sel = points_of(self, ray_samples)
        return self.evaluate(u, sel).view(ray_samples.ebins.shape[0], ray_samples.num_samples, 1)

    def get_density(self, ray_samples: RaySamples):
        return self.density_of_samples(ray_samples), None

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None) -> dict:
        return {}


class PropNetDensityFieldMS(nn.Module):
    def __init__(self, fields: List[PropNetDensityField], centroids: Tensor) -> None:
        super().__init__()
        self.register_buffer("centroids", deepcopy(centroids))
        self.fields = nn.ModuleList(fields)

    def _ms(self):
        e0 = self.fields[0].encoding
        if not getattr(self, "_ms_checked", False):
            for f in self.fields[1:]:
                if not _same_grid(f.encoding, e0):
                    raise NotImplementedError("presight_amd PropNetDensityFieldMS: the sub-fields of a tile must share one configuration")
            self._ms_checked = True
        dev = e0.hash_table.device
        aabbs = _stacked_aabbs(self, dev)
        return dict(aabbs=aabbs, contract=self.fields[0].spatial_distortion is not None, tables=[f.encoding.hash_table for f in self.fields],
                    scalings=e0.scalings_on(dev), g=_grid_cfg(e0), layers=[f.mlp_base[1].layer_params() for f in self.fields])

    def _ms_density(self, lay: "F.MsLayout", points: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
        """all K sub-fields in one launch per kernel (field_ops "MS" path); points = (u, sel) of this layout when the caller already has them"""
        m = self._ms()
        u, sel = points if points is not None else lay.points(m["aabbs"], m["contract"])
        return F.ms_prop_field(lay, u, sel, m["tables"], m["scalings"], m["g"], m["layers"])

    def density_fn(self, positions: Tensor) -> Tensor:
        if len(self.fields) == 1:
            return self.fields[0].density_fn(positions)
        return self._ms_density(F.MsLayout(self.centroids, pos=positions.reshape(-1, 3))).view(*positions.shape[:-1], 1)

    def density_of_samples(self, ray_samples: RaySamples) -> Tensor:
        if len(self.fields) == 1:
            return self.fields[0].density_of_samples(ray_samples)
        rb = ray_samples.ray_bundle
        lay = F.MsLayout(self.centroids, origins=rb.origins, dirs=rb.directions, ebins=ray_samples.ebins)
        return self._ms_density(lay).view(ray_samples.ebins.shape[0], ray_samples.num_samples, 1)

    def get_density(self, ray_samples: RaySamples):
        return self.density_of_samples(ray_samples), None


# ------------------------------------------------------------------------------------------------------------ sky
class SkyField(nn.Module):
    def __init__(self, direction_encoding: str = "SHEncoding", mlp_num_layers: int = 3, mlp_layer_width: int = 64,
                 appearance_embedding_dim: int = 32, use_semantics: bool = False, semantic_dim: int = 64,
                 implementation: str = "tcnn+fp32") -> None:
        super().__init__()
        self.use_semantics = use_semantics
        self.appearance_embedding_dim = appearance_embedding_dim
        self.direction_encoding = SHEncoding(levels=4, implementation=implementation)
        self.rgb_head = MLP(in_dim=self.direction_encoding.get_out_dim() + self.appearance_embedding_dim,
                            num_layers=mlp_num_layers, layer_width=mlp_layer_width, out_dim=3, activation=nn.ReLU(),
                            out_activation=nn.Sigmoid())
        if self.use_semantics:
            self.semantic_head = MLP(in_dim=self.direction_encoding.get_out_dim(), num_layers=mlp_num_layers,
                                     layer_width=mlp_layer_width, out_dim=semantic_dim, activation=nn.ReLU(), out_activation=None)

    def _fused(self, appearance_embedding: Optional[Tensor]) -> bool:
        A = 0 if appearance_embedding is None else appearance_embedding.shape[-1]
        r = self.rgb_head
        return (self.use_semantics and A == self.appearance_embedding_dim and r.layer_width == 32 and r.num_layers == 3
                and self.semantic_head.out_dim == 64 and F.sky_supported(A, 32, 3, 64))

    def get_outputs(self, directions: Tensor, appearance_embedding: Optional[Tensor]):
        """ns/fields/PreSight/sky_field.py:95-110"""
        if directions.is_cuda and directions.dim() == 2 and self._fused(appearance_embedding):
            rgb, sem = F.sky_field(directions, appearance_embedding, self.rgb_head.layer_params(), self.semantic_head.layer_params())
            return {FieldHeadNames.RGB: rgb, FieldHeadNames.SEMANTICS: sem}
        d = self.direction_encoding(get_normalized_directions(directions))
        outputs = {}
        x = torch.cat([d, appearance_embedding], dim=-1) if appearance_embedding is not None else d
        outputs[FieldHeadNames.RGB] = self.rgb_head(x)
        if self.use_semantics:
            outputs[FieldHeadNames.SEMANTICS] = self.semantic_head(d)
        return outputs

    def forward(self, ray_samples: RaySamples, appearance_embedding: Optional[Tensor]):
        R = ray_samples.ebins.shape[0]
        app = None if appearance_embedding is None else _per_ray(appearance_embedding, R)
        return self.get_outputs(ray_samples.ray_bundle.directions, app)


class SkyFieldMS(nn.Module):
    def __init__(self, fields: List[SkyField], centroids: Tensor) -> None:
        super().__init__()
        self.register_buffer("centroids", deepcopy(centroids))
        self.fields = nn.ModuleList(fields)

    def forward(self, ray_samples: RaySamples, appearance_embedding: Optional[Tensor]):
        """routed by ray ORIGIN (ns/fields/PreSight/sky_field_ms.py:97-114)"""
        rb = ray_samples.ray_bundle
        R = rb.origins.shape[0]
        app = None if appearance_embedding is None else _per_ray(appearance_embedding, R)
        if len(self.fields) == 1:
            return self.fields[0].get_outputs(rb.directions, app)
        if all(f._fused(app) for f in self.fields):  # all K sub-fields in one launch per direction, routed on the device
            lay = F.MsLayout(self.centroids, pos=rb.origins)
            rgb, sem = F.sky_field(rb.directions, app, [f.rgb_head.layer_params() for f in self.fields],
                                   [f.semantic_head.layer_params() for f in self.fields], lay=lay)
            return {FieldHeadNames.RGB: rgb, FieldHeadNames.SEMANTICS: sem}
        names: List[FieldHeadNames] = []

        def run(k, _origins, dirs, *a):
            o = self.fields[k].get_outputs(dirs, a[0] if a else None)
            if not names:
                names.extend(o.keys())
            return tuple(o[n] for n in names)

        vals = routed_apply(rb.origins, self.centroids, run, [rb.directions] + ([app] if app is not None else []))
        return dict(zip(names, vals))
