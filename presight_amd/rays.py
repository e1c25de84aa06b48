"""Ray containers mirroring ns/cameras/rays.py (RayBundle / RaySamples / Frustums) for the parts the PreSight
path touches.  Bin edges are kept as [R, S+1] tensors (euclidean `ebins`, normalised `sbins`); the reference's
per-sample views (starts/ends/deltas/spacing_*) are derived lazily with its shapes ([R, S, 1])."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import torch
from torch import Tensor

from . import ops


@dataclass
class RayBundle:
    origins: Tensor  # [R,3]
    directions: Tensor  # [R,3]
    pixel_area: Tensor  # [R,1]
    camera_indices: Optional[Tensor] = None  # [R,1]
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)
    times: Optional[Tensor] = None

    def __len__(self) -> int:
        return self.origins.shape[0]

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        sl = slice(start_idx, end_idx)
        return RayBundle(self.origins[sl], self.directions[sl], self.pixel_area[sl],
                         None if self.camera_indices is None else self.camera_indices[sl],
                         None if self.nears is None else self.nears[sl], None if self.fars is None else self.fars[sl],
                         {k: v[sl] for k, v in self.metadata.items()}, None if self.times is None else self.times[sl])


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
