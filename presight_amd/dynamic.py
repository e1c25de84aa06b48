"""BASELINE.json configs[3] (cfg 4): static + dynamic EmerNeRF-style dual field — two hash grids + a flow MLP.

The reference has NO dynamic / flow field, so there is nothing to mirror: the model is the
build's own definition ("parity unpinned"), made from the conventions of the static stack
(ns/fields/PreSight/ingp_field.py:168-267) and run on the HIP kernels of csrc/dynamic.hip + the existing fused MLP stack:

    static branch   iNGPField (fields.py), unchanged
    dynamic branch  x4 = (u, t) -> e0 = H4(x4) -> flow = s * MLP_flow(e0) -> feat = (e0 + H4(u+f_fwd, t+dt) + H4(u+f_bwd, t-dt)) / 3
                    -> [base MLP -> sigma_d | semantic head | colour head]   (the static field's fused stack, field_ops.main_stack)
    blend           sigma = sigma_s + sigma_d, w_d = sigma_d / max(sigma, 1e-6), c = c_s + w_d (c_d - c_s)

Module / parameter names follow the static field's scheme (`dynamic_field.encoding.hash_table`,
`dynamic_field.mlp_base_mlp.layers.{i}`, `...semantic_head`, `...rgb_head`, `...flow_head`)."""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Type

import torch
from torch import Tensor, nn

from . import field_ops as F
from . import ops, prof
from ._lib import check, lib
from .components import MLP, hash_scalings
from .fields import FieldHeadNames, _per_ray, iNGPField
from .losses import _chain, _finish
from .model import NerfactoNuscMSModel, NerfactoNuscMSModelConfig
from .ops import MlpSpec, _f32, _p, _stream, direct_params, flatten_grads, grad_sink, layer_sinks, mark_touched
from .rays import RaySamples
from .renderers import render_all


class HashEncoding4D(nn.Module):
    """Multiresolution hash grid over (x, y, z, t): the torch-path grid of ns/field_components/encodings.py:265-384 with a fourth
    axis (same per-level scalings on all axes, ceil / floor corners, hash = x ^ y*2654435761 ^ z*805459861 ^ t*3674653429)."""

    def __init__(self, num_levels: int = 8, min_res: int = 16, max_res: int = 512, log2_hashmap_size: int = 19,
                 features_per_level: int = 4, hash_init_scale: float = 0.001) -> None:
        super().__init__()
        if features_per_level not in (1, 2, 4):
            raise ValueError("features_per_level must be 1, 2 or 4 for the HIP hash grid")
        self.in_dim = 4
        self.num_levels, self.features_per_level, self.log2_hashmap_size = num_levels, features_per_level, log2_hashmap_size
        self.hash_table_size = 2 ** log2_hashmap_size
        self.scalings = hash_scalings(num_levels, min_res, max_res)
        table = torch.rand(size=(self.hash_table_size * num_levels, features_per_level)) * 2 - 1
        self.hash_table = nn.Parameter(table * hash_init_scale)
        self._scalings_dev: dict = {}

    def get_out_dim(self) -> int:
        return self.num_levels * self.features_per_level

    def scalings_on(self, device) -> Tensor:
        key = str(device)
        if key not in self._scalings_dev:
            self._scalings_dev[key] = self.scalings.to(device)
        return self._scalings_dev[key]

    def cfg(self) -> F.GridCfg:
        return F.GridCfg(self.num_levels, self.features_per_level, self.log2_hashmap_size)

    def forward(self, x4: Tensor) -> Tensor:
        """x4 [*bs, 4] -> [*bs, L*F] (no gradient: the differentiable path is DynamicField's fused node)"""
        flat = _f32(x4.reshape(-1, 4))
        planes = encode4(flat, self.hash_table.detach(), self.scalings_on(flat.device), self.cfg())
        return planes.permute(1, 0, 2).reshape(*x4.shape[:-1], self.get_out_dim())


def encode4(x4: Tensor, table: Tensor, scalings: Tensor, g: F.GridCfg, e0: Optional[Tensor] = None, counts: Optional[Tensor] = None) -> Tensor:
    """feature planes [L, N, F]; with e0 (x4 [2N,4] = forward- then backward-warped positions): (e0 + H4 + H4) / 3.
    counts (int32 [L * slices], accumulated into): record counts of the binned table backward for these positions"""
    N = x4.shape[0] if e0 is None else x4.shape[0] // 2
    feat = torch.empty(g.num_levels, N, g.features_per_level, device=x4.device)
    with prof.region("grid4_encode"):
        check(lib().ps_grid4_encode(_p(x4), _p(table), _p(scalings), g.num_levels, g.features_per_level, g.log2_hashmap_size, N,
                                    N * g.features_per_level, _p(e0), _p(feat), _p(counts), _stream()), "ps_grid4_encode")
    return feat


_FLOW_SPECS: dict = {}


def _flow_spec(LF: int, hidden: int) -> MlpSpec:
    key = (LF, hidden)
    if key not in _FLOW_SPECS:
        _FLOW_SPECS[key] = MlpSpec([LF, hidden, hidden, 6])
    return _FLOW_SPECS[key]


class _DynFeatures(torch.autograd.Function):
    """(u [N,3], ray times [N // S]) -> temporally aggregated dynamic features (planes [L,N,F]); one node for the positions, the
    encode, the flow MLP + warp, the two warped encodes and the aggregation.  The three position sets live in ONE [3N,4] buffer
    (unwarped | forward-warped | backward-warped).  Backward: d(feat) -> d(warped positions) -> flow MLP -> 3 d(e0) -> one binned
    table scatter over all 3N positions (record counts from the forward encodes)."""

    @staticmethod
    def forward(ctx, u, times, S, table, scalings, g: F.GridCfg, flow_scale, dt, *wb):
        u = _f32(u, "positions")
        table = _f32(table, "hash table")
        N, dev = u.shape[0], u.device
        layers = F._layers(wb)
        hidden = layers[0][0].shape[0]
        spec = _flow_spec(g.out_dim, hidden)
        packed = spec.pack(layers, dev)
        xall = torch.empty(3 * N, 4, device=dev)
        x4, xw = xall[:N], xall[N:]
        check(lib().ps_dyn_points(_p(u), _p(_f32(times).reshape(-1)), max(int(S), 1), N, _p(x4), _stream()), "ps_dyn_points")
        counts = None
        if F._training(ctx.needs_input_grad[3]):
            counts = torch.zeros(g.num_levels * lib().ps_grid_scatter_slices(g.features_per_level, g.log2_hashmap_size), device=dev,
                                 dtype=torch.int32)
        e0 = encode4(x4, table, scalings, g, counts=counts)
        with prof.region("flow_fwd"):
            check(lib().ps_flow_fwd(_p(e0), N * g.features_per_level, g.out_dim, g.features_per_level, hidden, _p(packed), _p(x4), N,
                                    float(flow_scale), float(dt), _p(xw), _stream()), "ps_flow_fwd")
        feat = encode4(xw, table, scalings, g, e0=e0, counts=counts)
        ctx.save_for_backward(xall, e0, packed, scalings, table, counts)
        ctx.meta = (g, hidden, float(flow_scale), tuple(table.shape), [tuple(W.shape) for W, _ in layers])
        ctx.sinks = (grad_sink(table), layer_sinks(layers))
        ctx.direct = direct_params(table, *wb)
        return feat

    @staticmethod
    def backward(ctx, dagg):
        xall, e0, packed, scalings, table, counts = ctx.saved_tensors
        g, hidden, flow_scale, tshape, shapes = ctx.meta
        spec = _flow_spec(g.out_dim, hidden)
        dagg = _f32(dagg)
        N, dev = e0.shape[1], e0.device
        xw = xall[N:]
        L, Fpl, l2t = g.num_levels, g.features_per_level, g.log2_hashmap_size
        ps = N * Fpl
        dxw = torch.empty(2 * N, 3, device=dev)
        with prof.region("grid4_input_grad"):
            if os.environ.get("PRESIGHT_GRID4_GRAD_LEVELS", "1") != "0":  # level-parallel (bit-identical; 4.8 -> ~1.5 ms at cfg 4)
                wsp = torch.empty(lib().ps_grid4_input_grad_workspace(L, 2 * N) // 4, device=dev)
                check(lib().ps_grid4_input_grad_levels(_p(xw), _p(dagg), _p(table), _p(scalings), L, Fpl, l2t, 2 * N, N, ps, 1.0 / 3.0, _p(dxw),
                                                       _p(wsp), _stream()), "ps_grid4_input_grad_levels")
            else:
                check(lib().ps_grid4_input_grad(_p(xw), _p(dagg), _p(table), _p(scalings), L, Fpl, l2t, 2 * N, N, ps, 1.0 / 3.0, _p(dxw),
                                                _stream()), "ps_grid4_input_grad")
        pf, gf, npart = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        check(lib().ps_flow_sizes(g.out_dim, hidden, N, ctypes.byref(pf), ctypes.byref(gf), ctypes.byref(npart)), "ps_flow_sizes")
        assert pf.value == spec.packed and gf.value == spec.g_total, (pf.value, spec.packed, gf.value, spec.g_total)
        gpart = torch.empty(npart.value, spec.g_total, device=dev)
        de0x3 = torch.empty_like(e0)  # 3 d(e0): the aggregation's 1/3 is applied by the scatter, once for all three sets
        with prof.region("flow_bwd"):
            check(lib().ps_flow_bwd(_p(e0), ps, g.out_dim, Fpl, hidden, _p(packed), _p(dxw), _p(dagg), N, flow_scale, _p(de0x3), _p(gpart),
                                    _stream()), "ps_flow_bwd")
        sink = ctx.sinks[0]
        dtable = sink if sink is not None else torch.zeros(tshape, device=dev)
        ws = F._workspace(lib().ps_grid4_scatter_workspace(L, Fpl, l2t, 3 * N), dev)
        with prof.region("grid4_scatter"):
            check(lib().ps_grid4_scatter_binned(_p(xall), _p(de0x3), _p(dagg), _p(scalings), L, Fpl, l2t, 3 * N, N, ps, 1.0 / 3.0, _p(dtable), 1,
                                                _p(counts), _p(ws), _stream()), "ps_grid4_scatter_binned")
        grads = spec.unpack_grads(gpart, npart.value, spec.g_total, 0, shapes, ctx.sinks[1])
        mark_touched(ctx.direct)
        return (None, None, None, None if sink is not None else dtable, None, None, None, None, *flatten_grads(grads))


class _Blend(torch.autograd.Function):
    """density-weighted mixture of the static and the dynamic branch per sample (ps_blend_fwd / ps_blend_bwd)"""

    @staticmethod
    def forward(ctx, ss, rs, ms, sd, rd, md):
        ss, rs, ms, sd, rd, md = (_f32(t) for t in (ss, rs, ms, sd, rd, md))
        N, C = ss.shape[0], ms.shape[1]
        sigma, rgb, sem = torch.empty_like(ss), torch.empty_like(rs), torch.empty_like(ms)
        with prof.region("blend_fwd"):
            check(lib().ps_blend_fwd(_p(ss), _p(rs), _p(ms), _p(sd), _p(rd), _p(md), N, C, _p(sigma), _p(rgb), _p(sem), _stream()), "ps_blend_fwd")
        ctx.save_for_backward(ss, rs, ms, sd, rd, md)
        return sigma, rgb, sem

    @staticmethod
    def backward(ctx, dsigma, drgb, dsem):
        ss, rs, ms, sd, rd, md = ctx.saved_tensors
        N, C = ss.shape[0], ms.shape[1]
        dsigma = None if dsigma is None else _f32(dsigma)
        drgb = None if drgb is None else _f32(drgb)
        dsem = None if dsem is None else _f32(dsem)
        dss, drs, dms, dsd, drd, dmd = (torch.empty_like(t) for t in (ss, rs, ms, sd, rd, md))
        with prof.region("blend_bwd"):
            check(lib().ps_blend_bwd(_p(ss), _p(rs), _p(ms), _p(sd), _p(rd), _p(md), _p(dsigma), _p(drgb), _p(dsem), N, C, None,
                                     _p(dss), _p(drs), _p(dms), _p(dsd), _p(drd), _p(dmd), _stream()), "ps_blend_bwd")
        return dss, drs, dms, dsd, drd, dmd


def blend(sigma_s, rgb_s, sem_s, sigma_d, rgb_d, sem_d):
    """-> (sigma [N], rgb [N,3], sem [N,C])"""
    return _Blend.apply(sigma_s, rgb_s, sem_s, sigma_d, rgb_d, sem_d)


class _MeanLoss(torch.autograd.Function):
    """scale * mean(x) in one launch (ps_loss_finish); the gradient is the constant g * scale / n"""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.shape = x.shape
        x = _f32(x).reshape(-1)
        ctx.n, ctx.scale, ctx.dev = x.numel(), float(scale), x.device
        return _finish(x, x.numel(), scale)[0]

    @staticmethod
    def backward(ctx, g):
        ones = getattr(_MeanLoss, "_ones", None)
        if ones is None or ones.numel() < ctx.n or ones.device != ctx.dev:
            ones = torch.ones(ctx.n, device=ctx.dev)
            _MeanLoss._ones = ones
        return _chain(ones[:ctx.n], g, ctx.scale / ctx.n).view(ctx.shape), None


def dynamic_density_loss(sigma_d: Tensor, scale: float = 0.01) -> Tensor:
    """EmerNeRF's dynamic-density regulariser: scale * mean(sigma_d)"""
    return _MeanLoss.apply(sigma_d, scale)


class DynamicField(nn.Module):
    """The dynamic branch: 4-D hash grid + flow MLP + the static field's MLP stack."""

    def __init__(self, num_levels: int = 8, base_res: int = 16, max_res: int = 512, log2_hashmap_size: int = 19, features_per_level: int = 4,
                 hidden_dim: int = 64, hidden_dim_color: int = 64, flow_hidden_dim: int = 64, geo_feat_dim: int = 15, semantic_dim: int = 64,
                 appearance_embedding_dim: int = 16, flow_scale: float = 0.05, time_step: float = 1.0 / 240.0) -> None:
        super().__init__()
        if geo_feat_dim != 15 or semantic_dim != 64 or appearance_embedding_dim > 16:
            raise NotImplementedError("presight_amd DynamicField: the fused stack covers geo 15 + semantics 64, app dim <= 16")
        self.flow_scale, self.time_step = float(flow_scale), float(time_step)
        self.geo_feat_dim, self.semantic_dim, self.appearance_embedding_dim = geo_feat_dim, semantic_dim, appearance_embedding_dim
        self.encoding = HashEncoding4D(num_levels, base_res, max_res, log2_hashmap_size, features_per_level)
        nin = self.encoding.get_out_dim()
        self.flow_head = MLP(in_dim=nin, num_layers=3, layer_width=flow_hidden_dim, out_dim=6, activation=nn.ReLU(), out_activation=None)
        self.mlp_base_mlp = MLP(in_dim=nin, num_layers=2, layer_width=hidden_dim, out_dim=1 + geo_feat_dim + semantic_dim, activation=nn.ReLU(),
                                out_activation=None)
        self.semantic_head = MLP(in_dim=semantic_dim, num_layers=3, layer_width=64, out_dim=semantic_dim, activation=nn.ReLU(),
                                 out_activation=None)
        self.rgb_head = MLP(in_dim=16 + geo_feat_dim + appearance_embedding_dim, num_layers=3, layer_width=hidden_dim_color, out_dim=3,
                            activation=nn.ReLU(), out_activation=nn.Sigmoid())

    def features(self, u: Tensor, times: Tensor, S: int) -> Tensor:
        """u [N,3] (normalised positions), times [N // S] -> aggregated feature planes [L,N,F]"""
        e = self.encoding
        flat = []
        for W, b in self.flow_head.layer_params():
            flat += [W, b]
        return F._apply(_DynFeatures, u, times, S, e.hash_table, e.scalings_on(u.device), e.cfg(), self.flow_scale, self.time_step, *flat)

    def evaluate(self, u: Tensor, sel: Tensor, times: Tensor, ray_dirs: Optional[Tensor], app: Optional[Tensor], S: int,
                 want_rgb: bool = True, want_sem: bool = True):
        """u [N,3] / sel [N] of the static field (iNGPField.points), times [N // S] -> (sigma_d [N], rgb_d [N,3], sem_d [N,64])"""
        feat = self.features(u, times, S)
        return F.main_stack(feat, sel, ray_dirs, app, S, self.encoding.cfg(), self.mlp_base_mlp.layer_params(),
                            self.semantic_head.layer_params(), self.rgb_head.layer_params(), want_rgb=want_rgb, want_sem=want_sem)


class DualField(nn.Module):
    """static field + DynamicField, blended per sample.  static_field: one iNGPField, or the routed iNGPFieldMS of a K > 1 tile (the
    SG-Onenorth tiles have 16 sub-fields, ns/configs/method_configs.py:271-367): the static branch is then routed like the reference's
    multi-scene field, the dynamic branch stays ONE field over the whole tile, normalised by the union of the sub-field boxes."""

    def __init__(self, static_field, dynamic_field: DynamicField) -> None:
        super().__init__()
        self.static_field = static_field
        self.dynamic_field = dynamic_field
        self.routed = hasattr(static_field, "fields") and len(static_field.fields) > 1
        if self.routed:
            boxes = torch.stack([f.aabb.float().reshape(2, 3) for f in static_field.fields])
            self.register_buffer("dyn_aabb", torch.stack([boxes[:, 0].min(0).values, boxes[:, 1].max(0).values]), persistent=False)
            self.contract = static_field.fields[0].spatial_distortion is not None

    def forward(self, ray_samples: RaySamples, appearance_embedding: Optional[Tensor], times: Tensor) -> Dict:
        rb = ray_samples.ray_bundle
        R, S = ray_samples.ebins.shape[0], ray_samples.num_samples
        st = self.static_field
        app = None if appearance_embedding is None else _per_ray(appearance_embedding, R)
        if self.routed:
            fo = st(ray_samples, appearance_embedding)  # all K sub-fields in one launch per kernel, outputs in the caller's order
            ss, rs, ms = fo[FieldHeadNames.DENSITY].reshape(-1), fo[FieldHeadNames.RGB].reshape(-1, 3), fo[FieldHeadNames.SEMANTICS].reshape(R * S, -1)
            u, sel = F.field_points(self.dyn_aabb, self.contract, origins=rb.origins, dirs=rb.directions, ebins=ray_samples.ebins)
        else:
            u, sel = st.points(origins=rb.origins, dirs=rb.directions, ebins=ray_samples.ebins)
            ss, rs, ms = st.evaluate(u, sel, rb.directions, app, S)
        sd, rd, md = self.dynamic_field.evaluate(u, sel, times, rb.directions, app, S)
        sigma, rgb, sem = blend(ss, rs, ms, sd, rd, md)
        return {FieldHeadNames.DENSITY: sigma.view(R, S, 1), FieldHeadNames.RGB: rgb.view(R, S, 3), FieldHeadNames.SEMANTICS: sem.view(R, S, -1),
                "dynamic_density": sd.view(R, S, 1), "static_density": ss.view(R, S, 1)}

    def density_of_samples(self, ray_samples: RaySamples, times: Tensor) -> Tensor:
        """total density [R,S,1] (both branches, heads skipped)"""
        rb = ray_samples.ray_bundle
        R, S = ray_samples.ebins.shape[0], ray_samples.num_samples
        st = self.static_field
        if self.routed:
            ss = st.density_only(ray_samples.frustums.get_positions()).reshape(-1)
            u, sel = F.field_points(self.dyn_aabb, self.contract, origins=rb.origins, dirs=rb.directions, ebins=ray_samples.ebins)
        else:
            u, sel = st.points(origins=rb.origins, dirs=rb.directions, ebins=ray_samples.ebins)
            ss = st.evaluate(u, sel, None, None, 1, want_rgb=False, want_sem=False)[0]
        sd = self.dynamic_field.evaluate(u, sel, times, None, None, S, want_rgb=False, want_sem=False)[0]
        return (ss + sd).view(R, S, 1)


@dataclass
class NerfactoNuscDualModelConfig(NerfactoNuscMSModelConfig):
    """the static model's config + the dynamic branch"""
    _target: Type = field(default_factory=lambda: NerfactoNuscDualModel)
    dynamic_num_levels: int = 8
    dynamic_base_res: int = 16
    dynamic_max_res: int = 512
    dynamic_log2_hashmap_size: int = 19
    dynamic_features_per_level: int = 4
    dynamic_hidden_dim: int = 64
    dynamic_hidden_dim_color: int = 64
    flow_hidden_dim: int = 64
    flow_scale: float = 0.05
    time_step: float = 1.0 / 240.0
    dynamic_reg_mult: float = 0.01


class NerfactoNuscDualModel(NerfactoNuscMSModel):
    """NerfactoNuscMSModel with the dual field in place of the static one (one sub-field, or a routed K > 1 static branch + one
    dynamic field over the tile).  Rays carry a normalised timestamp
    (`ray_bundle.times` [R,1] in [0,1]); proposal networks, samplers, renderers, sky model and losses are the static model's."""

    config: NerfactoNuscDualModelConfig

    def populate_modules(self):
        super().populate_modules()
        c = self.config
        self.dynamic_field = DynamicField(
            num_levels=c.dynamic_num_levels, base_res=c.dynamic_base_res, max_res=c.dynamic_max_res,
            log2_hashmap_size=c.dynamic_log2_hashmap_size, features_per_level=c.dynamic_features_per_level, hidden_dim=c.dynamic_hidden_dim,
            hidden_dim_color=c.dynamic_hidden_dim_color, flow_hidden_dim=c.flow_hidden_dim, semantic_dim=c.semantic_dim,
            appearance_embedding_dim=c.appearance_embed_dim + c.video_embed_dim, flow_scale=c.flow_scale, time_step=c.time_step)
        self.dual_field = DualField(self.field.fields[0] if len(self.field.fields) == 1 else self.field, self.dynamic_field)
        self.fused_render = False  # the per-sample outputs of both branches are blended before get_weights

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        groups = super().get_param_groups()
        groups["fields"] = groups["fields"] + list(self.dynamic_field.parameters())
        return groups

    def get_outputs(self, ray_bundle, jitters=None):
        c = self.config
        if ray_bundle.times is None:
            raise ValueError("NerfactoNuscDualModel: the ray bundle needs `times` ([R,1] normalised timestamps)")
        if self.param_gate is not None:
            self.param_gate("proposal_networks")
        ray_samples, weights_list, ray_samples_list = self.proposal_sampler(ray_bundle, density_fns=self.density_fns, jitters=jitters)
        if self.param_gate is not None:
            self.param_gate("fields")
        app = self._appearance(ray_bundle)
        app3 = None if app is None else app[:, None, :]
        fo = self.dual_field(ray_samples, app3, ray_bundle.times)
        weights = ray_samples.get_weights(fo[FieldHeadNames.DENSITY])
        weights_list.append(weights)
        ray_samples_list.append(ray_samples)
        rgb_s = fo[FieldHeadNames.RGB]
        if not self.training:
            rgb_s = torch.nan_to_num(rgb_s)
        rgb, acc_raw, depth, expected_depth, semantics = render_all(weights, ray_samples, rgb_s, fo[FieldHeadNames.SEMANTICS])
        if not self.training:
            rgb = torch.clamp(rgb, min=0.0, max=1.0)
        sky_outputs = {}
        if c.use_sky_model:
            sky_outputs = self.sky_model(ray_samples, appearance_embedding=app3)
        # (sky blend, or -- training -- the entries get_loss_dict fills from its fused blend + losses launch: the static model's helper)
        outputs = self._blend_outputs(rgb, acc_raw, semantics, sky_outputs, depth, expected_depth, weights, ray_samples)
        outputs["dynamic_density"] = fo["dynamic_density"]
        if self.training:
            outputs["weights_list"] = weights_list
            outputs["ray_samples_list"] = ray_samples_list
        with torch.no_grad():
            for i in range(c.num_proposal_iterations):
                outputs[f"prop_depth_{i}"] = self.renderer_depth(weights=weights_list[i], ray_samples=ray_samples_list[i])
        return outputs

    def get_depth(self, ray_bundle, threshold=0.5):
        """nerfacto_nusc_ms.py:688-708 with the density of both branches"""
        if self.collider is not None:
            ray_bundle = self.collider(ray_bundle)
        ray_samples, weights_list, ray_samples_list = self.proposal_sampler(ray_bundle, density_fns=self.density_fns)
        weights = ray_samples.get_weights(self.dual_field.density_of_samples(ray_samples, ray_bundle.times))
        _, _, depth, expected_depth, _ = render_all(weights, ray_samples, None, None, threshold)
        outputs = {"depth": depth.detach(), "expected_depth": expected_depth}
        if self.training:
            outputs["weights_list"] = weights_list
            outputs["ray_samples_list"] = ray_samples_list
        return outputs

    def get_loss_dict(self, outputs, batch, metrics_dict=None):
        loss_dict = super().get_loss_dict(outputs, batch, metrics_dict)
        if self.training:
            loss_dict["dynamic_reg_loss"] = dynamic_density_loss(outputs["dynamic_density"], scale=self.config.dynamic_reg_mult)
        return loss_dict
