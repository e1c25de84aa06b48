"""NerfactoNuscMSModel on the HIP kernels — the caller of the hot path.

Same config field names, module attribute names and state-dict keys as
ns/models/PreSight/nerfacto_nusc_ms.py:75-760, so PreSight's prior-building configs and checkpoints
(`implementation="torch"` layout: `...mlp_base_grid.hash_table`, `...layers.{i}.weight`) load unmodified."""
from __future__ import annotations

import functools
from collections import defaultdict
from collections.abc import MutableMapping
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Literal, Optional, Tuple, Type

import numpy as np
import torch
from torch import Tensor, nn
from torch.nn import Parameter

from . import ops
from .callbacks import TrainingCallback, TrainingCallbackAttributes, TrainingCallbackLocation  # noqa: F401
from .components import Embedding, SceneContraction
from .fields import (FieldHeadNames, PropNetDensityField, PropNetDensityFieldMS, SkyField, SkyFieldMS, iNGPField,
                     iNGPFieldMS, points_spec)
from .losses import (LossDict, MSELoss, blend_losses, deferred_finish, distortion_loss, expected_depth_loss, expected_monodepth_loss, line_of_sight_loss, semantic_loss, sky_loss,
                     z_anti_aliasing_interlevel_loss)
from .rays import RayBundle, RaySamples
from .renderers import AccumulationRenderer, DepthRenderer, NearFarCollider, RGBRenderer, render_all
from .samplers import ProposalNetworkSampler, SpacedSampler

RGB, FEATURES, SKY, DEPTH, VIDEO_ID = "rgb", "features", "sky", "depth", "video_id"  # ns/data/PreSight/constants.py


@dataclass
class NerfactoNuscMSModelConfig:
    """Field-for-field copy of the reference config dataclass (nerfacto_nusc_ms.py:75-200)."""
    _target: Type = field(default_factory=lambda: NerfactoNuscMSModel)
    enable_collider: bool = True
    collider_params: Optional[Dict[str, float]] = None
    loss_coefficients: Dict[str, float] = field(default_factory=lambda: {"rgb_loss_coarse": 1.0, "rgb_loss_fine": 1.0})
    eval_num_rays_per_chunk: int = 1 << 15
    prompt: Optional[str] = None
    near_plane: float = 0.1
    far_plane: float = 1000.0
    background_color: Literal["random", "last_sample", "black", "white"] = "black"
    hidden_dim: int = 64
    hidden_dim_color: int = 64
    num_levels: int = 10
    base_res: int = 16
    max_res: int = 16384
    log2_hashmap_size: int = 20
    features_per_level: int = 4
    num_proposal_samples_per_ray: Tuple[int, ...] = (128, 64)
    num_nerf_samples_per_ray: int = 64
    proposal_update_every: int = 5
    proposal_warmup: int = 1000
    num_proposal_iterations: int = 2
    use_same_proposal_network: bool = False
    proposal_net_args_list: List[Dict] = field(default_factory=lambda: [
        {"features_per_level": 1, "log2_hashmap_size": 20, "num_levels": 8, "base_res": 16, "max_res": 1024, "use_linear": False},
        {"features_per_level": 1, "log2_hashmap_size": 20, "num_levels": 8, "base_res": 16, "max_res": 4096, "use_linear": False},
    ])
    proposal_initial_sampler: Literal["piecewise", "uniform"] = "piecewise"
    piecewise_sampler_threshold: float = 1.0
    interlevel_loss_mult: float = 1.0
    enable_z_anti_aliasing: bool = True
    pulse_width: Tuple[float, ...] = (0.03, 0.003)
    distortion_loss_mult: float = 0.002
    orientation_loss_mult: float = 0.0001
    pred_normal_loss_mult: float = 0.001
    use_proposal_weight_anneal: bool = True
    use_average_appearance_embedding: bool = True
    proposal_weights_anneal_slope: float = 10.0
    proposal_weights_anneal_max_num_iters: int = 1000
    use_single_jitter: bool = True
    disable_scene_contraction: bool = False
    use_gradient_scaling: bool = False
    implementation: Literal["tcnn", "torch", "tcnn+fp32", "hip"] = "tcnn+fp32"
    appearance_embed_dim: int = 4
    video_embed_dim: int = 12
    use_sky_model: bool = True
    use_ms_sky_model: bool = True
    num_sky_mlp_layers: int = 3
    sky_mlp_dims: int = 32
    sky_loss_mult: float = 0.001
    use_lidar_loss: bool = True
    expected_depth_loss_mult: float = 1.0
    lidar_depth_upperbound: float = 75.0
    line_of_sight_mult: float = 0.1
    line_of_sight_decay_steps: int = 5000
    line_of_sight_start_step: int = 1000
    line_of_sight_end_step: int = 30000
    line_of_sight_max_sigma: float = 5.0
    line_of_sight_min_sigma: float = 2.0
    use_semantics: bool = True
    semantic_dim: int = 64
    semantic_loss_mult: float = 0.5
    use_monodepth_loss: bool = False
    monodepth_loss_inverse: bool = False
    monodepth_depth_upperbound: float = 40.0

    def setup(self, **kwargs):
        return self._target(self, **kwargs)


class LazyOutputs(MutableMapping):
    """The model's output mapping with entries that are evaluated on first access.  The reference renders the proposal levels'
    threshold depths in every forward (`prop_depth_i`, ns/models/PreSight/nerfacto_nusc_ms.py:543-544) although only the viewer and
    the evaluation images read them; in a training step they are two compositing launches nobody looks at.  Membership, iteration
    and item access behave like a dict's: a lazy key is present from the start (at the position it was registered), reading it
    evaluates it once.  A `MutableMapping`, not a dict subclass: CPython's fast paths for dict subclasses (`dict(o)`, `{**o}`,
    `d.update(o)`) bypass overridden accessors and would hand out the placeholder; here every access goes through `__getitem__`."""

    def __init__(self, *a, **k):
        self._data: Dict = dict(*a, **k)
        self._lazy: Dict[str, Callable] = {}

    def lazy(self, key: str, fn: Callable):
        self._lazy[key] = fn
        self._data[key] = None  # (keeps the key's position and membership)

    def lazy_group(self, keys, fn: Callable):
        """several entries produced by ONE evaluation: fn() -> {key: value} for all `keys`, run when the first of them is read.
        (Stored as (keys, fn), evaluated by _force: a closure over `self` here would tie the mapping into a reference cycle with its
        own entries, and the tensors they hold would then live until the cyclic collector runs instead of until the step ends.)"""
        ent = (tuple(keys), fn)
        for k in ent[0]:
            self._lazy[k] = ent
            self._data[k] = None

    def is_lazy(self, key: str) -> bool:
        """registered and not evaluated yet"""
        return key in self._lazy

    def _force(self, key):
        fn = self._lazy.pop(key, None)
        if isinstance(fn, tuple):
            keys, f = fn
            vals = f()
            for k in keys:
                self._lazy.pop(k, None)
                self._data[k] = vals[k]
        elif fn is not None:
            self._data[key] = fn()

    def __getitem__(self, key):
        self._force(key)
        return self._data[key]

    def __setitem__(self, key, value):
        self._lazy.pop(key, None)
        self._data[key] = value

    def __delitem__(self, key):
        self._lazy.pop(key, None)
        del self._data[key]

    def __iter__(self):
        return iter(self._data)

    def __len__(self):
        return len(self._data)

    def __contains__(self, key):
        return key in self._data

    def __repr__(self):
        return "LazyOutputs({" + ", ".join(f"{k!r}: {'<lazy>' if k in self._lazy else repr(v)}" for k, v in self._data.items()) + "})"

    def copy(self):
        return dict(self)


class _PropDensityFn:
    """density_fn handed to the ProposalNetworkSampler; accepts RaySamples so positions are made in-kernel."""
    takes_ray_samples = True

    def __init__(self, net: PropNetDensityFieldMS):
        self.net = net

    def __call__(self, x):
        if isinstance(x, RaySamples):
            return self.net.density_of_samples(x)
        return self.net.density_fn(x)

    def points_spec(self):
        """(aabb, contract, key) of the ONE sub-field this network evaluates its samples with (the sampling kernels then form its
        points on the way, fields.points_of), None for a routed K > 1 module (the router forms the points)"""
        return points_spec(self.net.fields[0]) if len(self.net.fields) == 1 else None


class NerfactoNuscMSModel(nn.Module):
    config: NerfactoNuscMSModelConfig

    def __init__(self, config: NerfactoNuscMSModelConfig, scene_box=None, num_train_data: int = -1, **kwargs) -> None:
        super().__init__()
        self.config = config
        self.scene_box = scene_box
        self.render_aabb = None
        self.num_train_data = num_train_data
        self.kwargs = kwargs
        self.collider = None
        self.populate_modules()
        self.callbacks = None
        self.device_indicator_param = nn.Parameter(torch.empty(0))

    @property
    def device(self):
        return self.device_indicator_param.device

    # ------------------------------------------------------------------------------------------------ construction
    def populate_modules(self):
        c = self.config
        assert not (c.use_lidar_loss and c.use_monodepth_loss)  # nerfacto_nusc_ms.py:361-365
        contraction = None if c.disable_scene_contraction else SceneContraction(order=float("inf"))
        self.centroids = self.kwargs["centroids"]
        self.aabbs = self.kwargs["aabbs"]
        app_dim = c.appearance_embed_dim + c.video_embed_dim
        fields = [iNGPField(aabb, hidden_dim=c.hidden_dim, num_levels=c.num_levels, max_res=c.max_res, base_res=c.base_res,
                            features_per_level=c.features_per_level, log2_hashmap_size=c.log2_hashmap_size,
                            hidden_dim_color=c.hidden_dim_color, spatial_distortion=contraction, num_images=self.num_train_data,
                            use_semantics=c.use_semantics, semantic_dim=c.semantic_dim, appearance_embedding_dim=app_dim,
                            implementation=c.implementation) for aabb in self.aabbs]
        self.field = iNGPFieldMS(fields, self.centroids)
        if c.appearance_embed_dim > 0:
            self.appearance_embedding = Embedding(self.kwargs["num_train_cameras"], c.appearance_embed_dim)
        if c.video_embed_dim > 0:
            self.video_embedding = Embedding(self.kwargs["num_train_videos"], c.video_embed_dim)
        self.dino_to_rgb = self.kwargs.get("dino_to_rgb")

        self.proposal_networks = torch.nn.ModuleList()
        n_prop = c.num_proposal_iterations
        n_nets = 1 if c.use_same_proposal_network else n_prop
        for i in range(n_nets):
            args = c.proposal_net_args_list[min(i, len(c.proposal_net_args_list) - 1)]
            pf = [PropNetDensityField(aabb, spatial_distortion=contraction, **args, implementation=c.implementation)
                  for aabb in self.aabbs]
            self.proposal_networks.append(PropNetDensityFieldMS(pf, self.centroids))
        nets = [self.proposal_networks[0]] * n_prop if c.use_same_proposal_network else list(self.proposal_networks)
        self.density_fns = [_PropDensityFn(n) for n in nets]

        if not c.enable_z_anti_aliasing:
            raise NotImplementedError("presight_amd: only the z-anti-aliased interlevel loss (PreSight default) is built")
        if c.use_gradient_scaling:
            # ns/models/PreSight/nerfacto_nusc_ms.py:500-501 (scale_gradients_by_distance_squared on the field outputs): no
            # PreSight method config enables it, and ignoring it silently would train a different model
            raise NotImplementedError("presight_amd: use_gradient_scaling=True (distance-squared gradient scaling) is not built")
        self.interlevel_loss = functools.partial(z_anti_aliasing_interlevel_loss, pulse_width=c.pulse_width)

        def update_schedule(step):
            return np.clip(np.interp(step, [0, c.proposal_warmup], [0, c.proposal_update_every]), 1, c.proposal_update_every)

        if c.proposal_initial_sampler != "piecewise":
            raise NotImplementedError("presight_amd: only the piecewise initial sampler (PreSight default) is built")
        thr = c.piecewise_sampler_threshold  # the reference's exact construction (nerfacto_nusc_ms.py:311-316)
        initial_sampler = SpacedSampler(
            spacing_fn=lambda x: torch.where(x < thr, x / (2 * thr), 1 - 1 / (2 * x / thr)),
            spacing_fn_inv=lambda x: torch.where(x < 0.5, x * (2 * thr), thr / (2 - 2 * x)),
            single_jitter=c.use_single_jitter)
        self.proposal_sampler = ProposalNetworkSampler(
            num_nerf_samples_per_ray=c.num_nerf_samples_per_ray, num_proposal_samples_per_ray=c.num_proposal_samples_per_ray,
            num_proposal_network_iterations=c.num_proposal_iterations, single_jitter=c.use_single_jitter,
            update_sched=update_schedule, initial_sampler=initial_sampler)
        self.collider = NearFarCollider(near_plane=c.near_plane, far_plane=c.far_plane)

        if c.use_sky_model:
            c.background_color = "black"
            mk = lambda: SkyField(mlp_num_layers=c.num_sky_mlp_layers, mlp_layer_width=c.sky_mlp_dims,  # noqa: E731
                                  appearance_embedding_dim=app_dim, use_semantics=c.use_semantics, semantic_dim=c.semantic_dim,
                                  implementation=c.implementation)
            self.sky_model = SkyFieldMS([mk() for _ in self.aabbs], centroids=self.centroids) if c.use_ms_sky_model else mk()
            self.sky_loss = sky_loss
        self.renderer_rgb = RGBRenderer(background_color=c.background_color)
        self.renderer_accumulation = AccumulationRenderer()
        self.renderer_depth = DepthRenderer(method="threshold")
        self.renderer_expected_depth = DepthRenderer(method="expected")
        self.rgb_loss = MSELoss()
        if c.use_semantics:
            self.semantic_loss = semantic_loss
        self.step = 0
        self.fused_render = True  # training, one sub-field, <= 64 samples: field + weights + renderers as one autograd node
        # training: sky blend + rgb / sky / semantic losses (+ gradients) as one launch, depths rendered on demand (PRESIGHT_FUSED_TAIL=0:
        # the separate operators of rounds 1-5)
        self.fused_tail = __import__("os").environ.get("PRESIGHT_FUSED_TAIL", "1") != "0"
        # data-parallel trainers with a sharded optimizer leave the all-gather of the updated parameters in flight and gate
        # the first use of each optimizer group here: param_gate("proposal_networks" | "fields") (a device-side stream wait)
        self.param_gate = None

    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        groups = {"proposal_networks": list(self.proposal_networks.parameters()), "fields": list(self.field.parameters())}
        if self.config.use_sky_model:
            groups["fields"] += list(self.sky_model.parameters())
        if self.config.appearance_embed_dim > 0:
            groups["fields"] += list(self.appearance_embedding.parameters())
        if self.config.video_embed_dim > 0:
            groups["fields"] += list(self.video_embedding.parameters())
        return groups

    def anneal_for_step(self, step: int) -> float:
        """nerfacto_nusc_ms.py:423-434"""
        N = self.config.proposal_weights_anneal_max_num_iters
        x = np.clip(step / N, 0, 1)
        b = self.config.proposal_weights_anneal_slope
        return float(b * x / ((b - 1) * x + 1))

    def get_training_callbacks(self, training_callback_attributes=None) -> List[TrainingCallback]:
        """nerfacto_nusc_ms.py:417-450: the proposal-weight anneal before every iteration, the sampler's step counter after it"""
        callbacks = []
        if self.config.use_proposal_weight_anneal:
            def set_anneal(step):
                self.step = step
                self.proposal_sampler.set_anneal(self.anneal_for_step(step))

            callbacks.append(TrainingCallback(where_to_run=[TrainingCallbackLocation.BEFORE_TRAIN_ITERATION], update_every_num_iters=1,
                                              func=set_anneal))
            callbacks.append(TrainingCallback(where_to_run=[TrainingCallbackLocation.AFTER_TRAIN_ITERATION], update_every_num_iters=1,
                                              func=self.proposal_sampler.step_cb))
        return callbacks

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, ray_bundle: RayBundle, jitters: Optional[List[Tensor]] = None):
        if self.collider is not None:
            ray_bundle = self.collider(ray_bundle)
        return self.get_outputs(ray_bundle, jitters=jitters)

    def _appearance(self, ray_bundle: RayBundle) -> Optional[Tensor]:
        """per-RAY appearance embedding [R, app+video] (the reference expands it per sample; the kernels index by ray)"""
        c = self.config
        cam = ray_bundle.camera_indices.reshape(-1)
        R = cam.shape[0]
        parts = []
        if self.training:
            idx, tabs = [], []
            if c.appearance_embed_dim > 0:
                idx.append(cam)
                tabs.append(self.appearance_embedding.embedding.weight)
            if c.video_embed_dim > 0:
                idx.append(ray_bundle.metadata[VIDEO_ID].reshape(-1))
                tabs.append(self.video_embedding.embedding.weight)
            return ops.embed_cat(idx, tabs) if tabs else None  # lookups + concat (+ scatter-add backward) in one op
        elif c.use_average_appearance_embedding:
            if c.appearance_embed_dim > 0:
                parts.append(self.appearance_embedding.mean(dim=0)[None].expand(R, -1))
            if c.video_embed_dim > 0:
                parts.append(self.video_embedding.mean(dim=0)[None].expand(R, -1))
        else:
            parts.append(torch.zeros(R, c.appearance_embed_dim + c.video_embed_dim, device=cam.device))
        if not parts:
            return None
        app = torch.cat(parts, dim=-1)
        return app if app.shape[-1] > 0 else None

    def get_outputs(self, ray_bundle: RayBundle, jitters: Optional[List[Tensor]] = None):
        """nerfacto_nusc_ms.py:452-546"""
        c = self.config
        if self.param_gate is not None:
            self.param_gate("proposal_networks")
        ray_samples, weights_list, ray_samples_list = self.proposal_sampler(
            ray_bundle, density_fns=self.density_fns, jitters=jitters,
            final_points_spec=points_spec(self.field.fields[0]) if len(self.field.fields) == 1 else None)
        if self.param_gate is not None:
            self.param_gate("fields")
        app = self._appearance(ray_bundle)
        app3 = None if app is None else app[:, None, :]
        if self.training and self.fused_render and c.use_semantics and self.field.can_render(ray_samples):
            # field + get_weights + renderers in one autograd node (never materialises the per-sample output gradients).  Without a
            # depth loss nothing in a training step reads the two depth renderers: the node may skip them (depth is None then) and
            # the entries below render them from the weights when somebody asks
            rgb, acc_raw, depth, expected_depth, semantics, weights = self.field.render(
                ray_samples, app3, want_depth=not self.fused_tail or c.use_monodepth_loss or c.use_lidar_loss)
            weights_list.append(weights)
            ray_samples_list.append(ray_samples)
        else:
            field_outputs = self.field.forward(ray_samples, appearance_embedding=app3)
            weights = ray_samples.get_weights(field_outputs[FieldHeadNames.DENSITY])
            weights_list.append(weights)
            ray_samples_list.append(ray_samples)
            sem_s = field_outputs[FieldHeadNames.SEMANTICS] if c.use_semantics else None
            rgb_s = field_outputs[FieldHeadNames.RGB]
            if not self.training:
                rgb_s = torch.nan_to_num(rgb_s)
            rgb, acc_raw, depth, expected_depth, semantics = render_all(weights, ray_samples, rgb_s, sem_s)
        if not self.training:
            rgb = torch.clamp(rgb, min=0.0, max=1.0)
        sky_outputs = {}
        if c.use_sky_model:
            sky_outputs = self.sky_model(ray_samples, appearance_embedding=None if app is None else app[:, None, :])
        outputs = self._blend_outputs(rgb, acc_raw, semantics, sky_outputs, depth, expected_depth, weights, ray_samples)
        if self.training:
            outputs["weights_list"] = weights_list
            outputs["ray_samples_list"] = ray_samples_list

        def prop_depth(i):
            with torch.no_grad():
                return self.renderer_depth(weights=weights_list[i], ray_samples=ray_samples_list[i])

        for i in range(c.num_proposal_iterations):
            if self.training:  # evaluated when somebody reads it (nothing in a training step does)
                outputs.lazy(f"prop_depth_{i}", functools.partial(prop_depth, i))
            else:
                outputs[f"prop_depth_{i}"] = prop_depth(i)
        return outputs

    def _blend_outputs(self, rgb, acc_raw, semantics, sky_outputs, depth, expected_depth, weights, ray_samples) -> "LazyOutputs":
        """the rendered outputs of get_outputs from the field's composited values and the sky model's (shared with the dual model)"""
        c = self.config
        # accumulation = clamp(acc, 0, 1); rgb/semantics += (1 - accumulation) * sky   (nerfacto_nusc_ms.py:512-533)
        blend_in = (rgb, acc_raw, semantics if c.use_semantics else None, sky_outputs.get(FieldHeadNames.RGB),
                    sky_outputs.get(FieldHeadNames.SEMANTICS))
        outputs = LazyOutputs({"rgb": None, "accumulation": None, "depth": None, "expected_depth": None})
        if c.use_semantics:
            outputs["semantics"] = None
        if self.training and self.fused_tail:
            # Training: the blend is an entry group evaluated on first access (ops.sky_blend, as before) -- unless get_loss_dict gets
            # there first, which is what a training iteration does: it then runs the blend TOGETHER with the three per-ray losses and
            # their gradients in one launch (losses.blend_losses) and fills these entries from it
            keys = ("rgb", "accumulation") + (("semantics",) if c.use_semantics else ())

            outputs.lazy_group(keys, lambda: dict(zip(keys, ops.sky_blend(*blend_in))))  # (no reference to `outputs`: no cycle)
            outputs.pending_blend = blend_in  # (consumed by get_loss_dict only while the group is still unevaluated)
        else:
            rgb, accumulation, semantics = ops.sky_blend(*blend_in)
            outputs["rgb"], outputs["accumulation"] = rgb, accumulation
            if c.use_semantics:
                outputs["semantics"] = semantics
        if depth is None:  # (the fused node skipped the depth renderers)
            def depths():
                _, _, d, e, _ = render_all(weights, ray_samples, None, None)  # differentiable through the node's weights
                return {"depth": d.detach(), "expected_depth": e}

            outputs.lazy_group(("depth", "expected_depth"), depths)
        else:
            outputs["depth"], outputs["expected_depth"] = depth.detach(), expected_depth
        if c.use_semantics and not self.training and self.dino_to_rgb is not None:
            outputs["dino_rgb"] = apply_feature_colormap(outputs["semantics"], self.dino_to_rgb)
        return outputs

    def get_metrics_dict(self, outputs, batch):
        mse = torch.mean((outputs["rgb"] - batch["rgb"][..., :3]) ** 2)
        return {"psnr": 10.0 * torch.log10(1.0 / mse)}  # torchmetrics PSNR(data_range=1.0), nerfacto_nusc_ms.py:382,554

    def get_image_metrics_and_images(self, outputs: Dict[str, Tensor], batch: Dict[str, Tensor]):
        """nerfacto_nusc_ms.py:647-686: (metrics, images) of one rendered evaluation image.  PSNR as torchmetrics computes it
        (data_range = 1.0); ssim / lpips are third-party networks of the reference's logging and are not part of this path."""
        gt_rgb = self.renderer_rgb.blend_background(batch["rgb"].to(outputs["rgb"].device))
        predicted_rgb = outputs["rgb"]
        acc = apply_colormap(outputs["accumulation"])
        depth = apply_depth_colormap(outputs["depth"], accumulation=outputs["accumulation"])
        images_dict = {"img": torch.cat([gt_rgb, predicted_rgb], dim=1), "accumulation": acc, "depth": depth}
        mse = torch.mean((gt_rgb - predicted_rgb) ** 2)
        metrics_dict = {"psnr": float(10.0 * torch.log10(1.0 / mse))}
        for i in range(self.config.num_proposal_iterations):
            key = f"prop_depth_{i}"
            images_dict[key] = apply_depth_colormap(outputs[key], accumulation=outputs["accumulation"])
        return metrics_dict, images_dict

    def get_loss_dict(self, outputs, batch, metrics_dict=None):
        """nerfacto_nusc_ms.py:558-645 (camera-only); every term is scaled by its *_loss_mult INSIDE the loss launch
        (losses.py: `scale`), the values equal the reference's mult * loss"""
        with deferred_finish() as fin:  # every term's scalar value AND their sum: one launch at the end of the block (losses.py)
            loss_dict = self._loss_terms(outputs, batch)
            if loss_dict and all(torch.is_tensor(v) and v.is_cuda and v.numel() == 1 for v in loss_dict.values()):
                loss_dict.total = fin.flush(list(loss_dict.values()))
                loss_dict._total_of = fin.order_ptrs
        return loss_dict

    def _loss_terms(self, outputs, batch) -> LossDict:
        c = self.config
        loss_dict = LossDict()
        pend = getattr(outputs, "pending_blend", None)
        fused = pend is not None and RGB in batch and outputs.is_lazy("rgb")
        sem_loss = None
        if fused:
            # sky blend + MSELoss(rgb) + sky_loss + semantic_loss, values and gradients, in one launch (losses.blend_losses)
            (l_rgb, l_sky, sem_loss), (rgb, acc, sem) = blend_losses(
                *pend, rgb_target=batch[RGB][..., :3], sky_mask=batch[SKY] if (c.use_sky_model and SKY in batch) else None,
                sem_target=batch[FEATURES] if (c.use_semantics and FEATURES in batch) else None, rgb_mult=1.0, sky_mult=c.sky_loss_mult,
                sem_mult=c.semantic_loss_mult)
            outputs["rgb"], outputs["accumulation"] = rgb, acc
            if c.use_semantics:
                outputs["semantics"] = sem
            outputs.pending_blend = None
            loss_dict["rgb_loss"] = l_rgb
            if l_sky is not None:
                loss_dict["sky_loss"] = l_sky
        else:
            if RGB in batch:
                loss_dict["rgb_loss"] = self.rgb_loss(batch[RGB][..., :3], outputs["rgb"])
            if c.use_sky_model and SKY in batch:
                loss_dict["sky_loss"] = self.sky_loss(outputs["accumulation"].view(-1, 1), batch[SKY].view(-1, 1), scale=c.sky_loss_mult)
        if (c.use_monodepth_loss or c.use_lidar_loss) and DEPTH in batch:  # nerfacto_nusc_ms.py:576-629
            rs = outputs["ray_samples_list"][-1]
            scale = self._pose_scale_factor(rs)
            mono = c.use_monodepth_loss
            ub = c.monodepth_depth_upperbound if mono else c.lidar_depth_upperbound
            sky_mask = batch[SKY].view(-1, 1) if mono else None  # the reference reuses the sky branch's mask
            if mono:
                ed = expected_monodepth_loss(batch[DEPTH], outputs["expected_depth"], sky_mask, upper_bound=ub,
                                             inverse=c.monodepth_loss_inverse, pose_scale_factor=scale, scale=c.expected_depth_loss_mult)
            else:
                ed = expected_depth_loss(batch[DEPTH], outputs["expected_depth"], upper_bound=ub, pose_scale_factor=scale,
                                         scale=c.expected_depth_loss_mult)
            loss_dict["expected_depth_loss"] = ed
            loss_dict["line_of_sight_loss"] = line_of_sight_loss(
                outputs["weights_list"][-1], batch[DEPTH], rs, sigma=self.get_line_of_sight_sigma(self.step), sky_mask=sky_mask,
                upper_bound=ub, pose_scale_factor=scale, scale=self.get_line_of_sight_mult(self.step))
        if fused:
            if sem_loss is not None:
                loss_dict["semantic_loss"] = sem_loss
        elif c.use_semantics and FEATURES in batch:
            loss_dict["semantic_loss"] = self.semantic_loss(pred=outputs["semantics"], target=batch[FEATURES], clip=True, scale=c.semantic_loss_mult)
        if self.training:
            loss_dict["interlevel_loss"] = self.interlevel_loss(outputs["weights_list"], outputs["ray_samples_list"], scale=c.interlevel_loss_mult)
            loss_dict["distortion_loss"] = distortion_loss(outputs["weights_list"], outputs["ray_samples_list"], scale=c.distortion_loss_mult)
        return loss_dict

    def get_line_of_sight_sigma(self, step):
        """nerfacto_nusc_ms.py:387-396"""
        c = self.config
        frac = np.clip((step - c.line_of_sight_start_step) / (c.line_of_sight_end_step - c.line_of_sight_start_step), 0.0, 1.0)
        return c.line_of_sight_max_sigma - frac * (c.line_of_sight_max_sigma - c.line_of_sight_min_sigma)

    def get_line_of_sight_mult(self, step):
        """nerfacto_nusc_ms.py:398-403"""
        c = self.config
        if step <= c.line_of_sight_start_step:
            return 0.0
        return c.line_of_sight_mult / (2.0 ** (step // c.line_of_sight_decay_steps))

    def _pose_scale_factor(self, ray_samples) -> float:
        """metadata["pose_scale_factor"] is a per-ray copy of one dataset constant (nerfacto_nusc_ms.py:582 reads element
        [0,0,0]); it is a kernel argument here, so it is fetched from the device once and cached."""
        v = ray_samples.metadata["pose_scale_factor"]
        if not torch.is_tensor(v):
            return float(v)
        if getattr(self, "_pose_scale_cache", None) is None:
            self._pose_scale_cache = float(v.reshape(-1)[0])  # one host sync for the lifetime of the model
        return self._pose_scale_cache

    # ------------------------------------------------------------------------------------------------ depth / eval
    def get_depth(self, ray_bundle: RayBundle, threshold=0.5):
        """nerfacto_nusc_ms.py:688-708"""
        if self.collider is not None:
            ray_bundle = self.collider(ray_bundle)
        ray_samples, weights_list, ray_samples_list = self.proposal_sampler(ray_bundle, density_fns=self.density_fns)
        if len(self.field.fields) == 1:
            f = self.field.fields[0]
            u, sel = f.points(origins=ray_bundle.origins, dirs=ray_bundle.directions, ebins=ray_samples.ebins)
            density = f.evaluate(u, sel, None, None, 1, want_rgb=False, want_sem=False)[0].view(ray_samples.ebins.shape[0], -1, 1)
        else:
            density = self.field.density_only(ray_samples.frustums.get_positions())
        weights = ray_samples.get_weights(density)
        _, _, depth, expected_depth, _ = render_all(weights, ray_samples, None, None, threshold)
        outputs = {"depth": depth.detach(), "expected_depth": expected_depth}
        if self.training:
            outputs["weights_list"] = weights_list
            outputs["ray_samples_list"] = ray_samples_list
        return outputs

    def _chunked(self, camera_ray_bundle: RayBundle, fn):
        n = self.config.eval_num_rays_per_chunk
        lists = defaultdict(list)
        for i in range(0, len(camera_ray_bundle), n):
            out = fn(camera_ray_bundle.get_row_major_sliced_ray_bundle(i, i + n))
            for k, v in out.items():
                if torch.is_tensor(v):
                    lists[k].append(v)
        return {k: torch.cat(v) for k, v in lists.items()}

    @torch.no_grad()
    def get_depth_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle, threshold=0.5):
        return self._chunked(camera_ray_bundle, lambda rb: self.get_depth(ray_bundle=rb, threshold=threshold))

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle):
        return self._chunked(camera_ray_bundle, lambda rb: self.forward(ray_bundle=rb))


def apply_colormap(image: Tensor, eps: float = 1e-9) -> Tensor:
    """ns/utils/colormaps.py:46-115 for the cases the model uses: 3 channels pass through, 1 float channel goes through the
    "turbo" map (the reference looks it up in matplotlib's 256-entry table; this is the published polynomial fit of the same
    map -- a visualisation, not a compared quantity)."""
    if image.shape[-1] == 3:
        return image
    x = torch.clip(torch.nan_to_num(image[..., 0], 0.0), 0.0, 1.0)
    v4 = torch.stack([torch.ones_like(x), x, x * x, x * x * x], -1)
    v2 = torch.stack([x ** 4, x ** 5], -1)
    r4 = v4.new_tensor([0.13572138, 4.61539260, -42.66032258, 132.13108234])
    g4 = v4.new_tensor([0.09140261, 2.19418839, 4.84296658, -14.18503333])
    b4 = v4.new_tensor([0.10667330, 12.64194608, -60.58204836, 110.36276771])
    r2, g2, b2 = v2.new_tensor([-152.94239396, 59.28637943]), v2.new_tensor([4.27729857, 2.82956604]), v2.new_tensor([-89.90310912, 27.34824973])
    rgb = torch.stack([v4 @ r4 + v2 @ r2, v4 @ g4 + v2 @ g2, v4 @ b4 + v2 @ b2], -1)
    return torch.clip(rgb, 0.0, 1.0)


def apply_depth_colormap(depth: Tensor, accumulation: Optional[Tensor] = None, near_plane: Optional[float] = None,
                         far_plane: Optional[float] = None) -> Tensor:
    """ns/utils/colormaps.py:117-149"""
    near_plane = near_plane or float(torch.min(depth))
    far_plane = far_plane or float(torch.max(depth))
    depth = torch.clip((depth - near_plane) / (far_plane - near_plane + 1e-10), 0, 1)
    colored = apply_colormap(depth)
    if accumulation is not None:
        colored = colored * accumulation + (1 - accumulation)
    return colored


def apply_feature_colormap(image: Tensor, dino_to_rgb: dict) -> Tensor:
    """ns/utils/colormaps.py:212-234"""
    red = dino_to_rgb["reduction_matrix"].to(image)
    lo, hi, mean = dino_to_rgb["rgb_min"].to(image), dino_to_rgb["rgb_max"].to(image), dino_to_rgb["mean"].to(image)
    x = (image - mean) @ red
    return torch.clamp((x - lo) / (hi - lo), 0, 1)
