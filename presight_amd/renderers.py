"""Renderers and collider of the PreSight model (ns/model_components/renderers.py:58-383,
ns/model_components/scene_colliders.py:169-187) on the per-ray composite kernel."""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from . import ops
from .rays import RayBundle, RaySamples


class NearFarCollider(nn.Module):
    def __init__(self, near_plane: float, far_plane: float, **kwargs) -> None:
        super().__init__()
        self.near_plane = near_plane
        self.far_plane = far_plane

    def set_nears_and_fars(self, ray_bundle: RayBundle) -> RayBundle:
        """scene_colliders.py:182-187: nears = near_plane (training) / 0 (evaluation), fars = far_plane for every ray.  The planes are
        recorded as scalars; `ray_bundle.nears` / `.fars` [R,1] are filled when read (rays.RayBundle)"""
        near_plane = self.near_plane if self.training else 0
        ray_bundle.nears = ray_bundle.fars = None
        ray_bundle.metadata["_near_far"] = (float(near_plane), float(self.far_plane))
        return ray_bundle

    def forward(self, ray_bundle: RayBundle) -> RayBundle:
        if "_near_far" in ray_bundle.metadata:
            return ray_bundle
        return self.set_nears_and_fars(ray_bundle)


def render_all(weights: Tensor, ray_samples: RaySamples, rgb: Optional[Tensor], semantics: Optional[Tensor], threshold: float = 0.5):
    """One pass over the samples of every ray: -> (rgb [R,3], accumulation [R,1] unclamped, threshold depth [R,1],
    expected depth [R,1] (clipped to the batch-global sample range), semantics [R,C])."""
    w = weights.reshape(weights.shape[0], weights.shape[1]) if weights.dim() == 3 else weights
    return ops.composite(w, ray_samples.ebins, rgb, semantics, threshold)


class RGBRenderer(nn.Module):
    def __init__(self, background_color="black") -> None:
        super().__init__()
        if background_color != "black":
            # ns/model_components/renderers.py:199-229 blends a random colour during training for "random": rendering that as black would
            # be a silent deviation, so every non-black background is refused (PreSight sets "black", nerfacto_nusc_ms.py:86,333)
            raise NotImplementedError(f"presight_amd RGBRenderer: background_color={background_color!r} is not implemented; "
                                      "PreSight renders on a black background")
        self.background_color = background_color

    def forward(self, rgb: Tensor, weights: Tensor, ray_indices=None, num_rays=None, background_color=None) -> Tensor:
        if not self.training:
            rgb = torch.nan_to_num(rgb)
        R, S = weights.shape[0], weights.shape[1]
        eb = torch.zeros(R, S + 1, device=weights.device)
        out = ops.composite(weights[..., 0], eb, rgb, None)[0]
        if not self.training:
            out = torch.clamp(out, min=0.0, max=1.0)
        return out

    def blend_background(self, image: Tensor, background_color=None) -> Tensor:
        return image[..., :3] if image.shape[-1] >= 3 else image

    def blend_background_for_loss_computation(self, pred_image, pred_accumulation, gt_image):
        return pred_image, self.blend_background(gt_image)


class AccumulationRenderer(nn.Module):
    @classmethod
    def forward(cls, weights: Tensor, ray_indices=None, num_rays=None) -> Tensor:
        R, S = weights.shape[0], weights.shape[1]
        eb = torch.zeros(R, S + 1, device=weights.device)
        return ops.composite(weights[..., 0], eb, None, None)[1]


class DepthRenderer(nn.Module):
    def __init__(self, method: str = "threshold") -> None:
        super().__init__()
        if method not in ("threshold", "expected"):
            raise NotImplementedError(f"Method {method} not implemented")
        self.method = method

    def forward(self, weights: Tensor, ray_samples: RaySamples, ray_indices=None, num_rays=None, threshold: float = 0.5) -> Tensor:
        if self.method == "threshold":
            return ops.threshold_depth(weights[..., 0], ray_samples.ebins, threshold)
        return ops.composite(weights[..., 0], ray_samples.ebins, None, None, threshold)[3]
