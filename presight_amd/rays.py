"""Ray containers mirroring ns/cameras/rays.py (RayBundle / RaySamples / Frustums) for the parts the PreSight
path touches.  Bin edges are kept as [R, S+1] tensors (euclidean `ebins`, normalised `sbins`); the reference's
per-sample views (starts/ends/deltas/spacing_*) are derived lazily with its shapes ([R, S, 1])."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import torch
from torch import Tensor

from . import ops


class RayBundle:
    """ns/cameras/rays.py:153-249 for the fields the PreSight path touches.  `nears` / `fars` [R,1]: the collider of the PreSight model
    sets ONE near / far plane for the whole batch (scene_colliders.py:182-187) and the kernels take them as scalars
    (metadata["_near_far"]); the two tensors the reference fills are materialised when somebody reads them (a training step never
    does: three element-wise launches per step otherwise)."""

    def __init__(self, origins: Tensor, directions: Tensor, pixel_area: Tensor, camera_indices: Optional[Tensor] = None,
                 nears: Optional[Tensor] = None, fars: Optional[Tensor] = None, metadata: Optional[Dict[str, Tensor]] = None,
                 times: Optional[Tensor] = None):
        self.origins, self.directions, self.pixel_area = origins, directions, pixel_area  # [R,3] [R,3] [R,1]
        self.camera_indices = camera_indices  # [R,1]
        self._nears, self._fars = nears, fars
        self.metadata = {} if metadata is None else metadata
        self.times = times

    def _plane(self, which: int) -> Optional[Tensor]:
        nf = self.metadata.get("_near_far")
        if nf is None:
            return None
        return torch.full_like(self.origins[..., 0:1], float(nf[which]))

    @property
    def nears(self) -> Optional[Tensor]:
        if self._nears is None:
            self._nears = self._plane(0)
        return self._nears

    @nears.setter
    def nears(self, v):
        self._nears = v

    @property
    def fars(self) -> Optional[Tensor]:
        if self._fars is None:
            self._fars = self._plane(1)
        return self._fars

    @fars.setter
    def fars(self, v):
        self._fars = v

    def __len__(self) -> int:
        return self.origins.shape[0]

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        sl = slice(start_idx, end_idx)
        return RayBundle(self.origins[sl], self.directions[sl], self.pixel_area[sl],
                         None if self.camera_indices is None else self.camera_indices[sl],
                         None if self._nears is None else self._nears[sl], None if self._fars is None else self._fars[sl],
                         {k: (v[sl] if torch.is_tensor(v) else v) for k, v in self.metadata.items()}, None if self.times is None else self.times[sl])


@dataclass
class Frustums:
    origins: Tensor  # [R,1,3]
    directions: Tensor  # [R,1,3]
    starts: Tensor  # [R,S,1]
    ends: Tensor  # [R,S,1]
    pixel_area: Tensor  # [R,1,1]

    def get_positions(self) -> Tensor:
        """o + d*(start+end)/2, ns/cameras/rays.py:49-58 -> [R,S,3]."""
        eb = torch.cat([self.starts[..., 0], self.ends[:, -1:, 0]], dim=-1)
        R, S = self.starts.shape[0], self.starts.shape[1]
        return ops.sample_positions(self.origins[:, 0], self.directions[:, 0], eb).view(R, S, 3)


@dataclass
class RaySamples:
    """Samples along rays.  `ebins`/`sbins` [R,S+1] are the euclidean / normalised bin edges."""
    ray_bundle: RayBundle
    ebins: Tensor
    sbins: Tensor
    spacing_to_euclidean_fn: Optional[Callable] = None
    # (key, u [R*S,3], sel [R*S]): the normalised sample points of the field identified by `key` (its box + contraction), when the
    # kernel that formed these bins produced them on the way (fields.points_of); None: the field computes them itself
    points: Optional[tuple] = None

    @property
    def num_samples(self) -> int:
        return self.ebins.shape[1] - 1

    @property
    def frustums(self) -> Frustums:
        rb = self.ray_bundle
        return Frustums(rb.origins[:, None], rb.directions[:, None], self.ebins[:, :-1, None], self.ebins[:, 1:, None],
                        rb.pixel_area[:, None])

    @property
    def deltas(self) -> Tensor:
        return (self.ebins[:, 1:] - self.ebins[:, :-1])[..., None]

    @property
    def spacing_starts(self) -> Tensor:
        return self.sbins[:, :-1, None]

    @property
    def spacing_ends(self) -> Tensor:
        return self.sbins[:, 1:, None]

    @property
    def camera_indices(self) -> Optional[Tensor]:
        ci = self.ray_bundle.camera_indices
        return None if ci is None else ci[:, None]

    @property
    def metadata(self):
        return {k: v[:, None] for k, v in self.ray_bundle.metadata.items() if torch.is_tensor(v)}

    def get_weights(self, densities: Tensor) -> Tensor:
        """ns/cameras/rays.py:128-150: densities [R,S,1] -> weights [R,S,1]."""
        w = ops.weights_from_density(self.ebins, densities.reshape(densities.shape[0], densities.shape[1]))
        return w[..., None]
