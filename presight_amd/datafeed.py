"""Device-side data feed.  The reference draws a batch by indexing an `ImageChunk` of flat per-pixel
arrays one pixel at a time in DataLoader workers and collating on the host (ns/data/PreSight/my_dataset.py:28-73,
ns/data/PreSight/my_datamanager.py:203-285); at millions of rays per second that Python path is the bottleneck.  Here the
chunk lives in HBM and a batch is two launches: draw the pixel slots, gather every field + form the ray indices."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import Tensor

from ._lib import check, lib
from .ops import _p, _stream

RGB, FEATURES, SKY, DEPTH, VIDEO_ID, RAY_INDEX = "rgb", "features", "sky", "depth", "video_id", "ray_indices"


@dataclass
class DeviceImageChunk:
    """same fields as the reference's ImageChunk, resident on the GPU (float32 / int64, contiguous)"""
    rgbs: Tensor                 # [P,3]
    pixel_indices: Tensor        # [P] flat pixel id inside its image
    image_indices: Tensor        # [P]
    video_ids: Tensor            # [P]
    widths: Tensor               # [P]
    skies: Optional[Tensor] = None      # [P] 1.0 = sky
    depths: Optional[Tensor] = None     # [P]
    features: Optional[Tensor] = None   # [P,C]

    def __post_init__(self):
        for name in ("rgbs", "skies", "depths", "features"):
            t = getattr(self, name)
            if t is not None:
                if not t.is_cuda:
                    raise RuntimeError("presight_amd DeviceImageChunk: the chunk must live on the GPU (no CPU path)")
                setattr(self, name, t.float().contiguous())
        for name in ("pixel_indices", "image_indices", "video_ids", "widths"):
            setattr(self, name, getattr(self, name).to(torch.int64).contiguous())

    def __len__(self) -> int:
        return self.rgbs.shape[0]

    def gather(self, pick: Tensor) -> Dict[str, Tensor]:
        """the collated batch of the pixel slots `pick` (int64 [R]): what iterating the reference's DataLoader yields"""
        pick = pick.to(torch.int64).contiguous()
        R, dev = pick.shape[0], self.rgbs.device
        C = 0 if self.features is None else self.features.shape[1]
        out = {RAY_INDEX: torch.empty(R, 3, device=dev, dtype=torch.int64), RGB: torch.empty(R, 3, device=dev),
               VIDEO_ID: torch.empty(R, device=dev, dtype=torch.int64)}
        if self.skies is not None:
            out[SKY] = torch.empty(R, device=dev)
        if self.depths is not None:
            out[DEPTH] = torch.empty(R, device=dev)
        if self.features is not None:
            out[FEATURES] = torch.empty(R, C, device=dev)
        check(lib().ps_gather_batch(_p(pick), R, _p(self.rgbs), _p(self.skies), _p(self.depths), _p(self.features), C,
                                    _p(self.pixel_indices), _p(self.image_indices), _p(self.video_ids), _p(self.widths),
                                    _p(out[RAY_INDEX]), _p(out[RGB]), _p(out.get(SKY)), _p(out.get(DEPTH)), _p(out.get(FEATURES)),
                                    _p(out[VIDEO_ID]), _stream()), "ps_gather_batch")
        return out

    def sample_batch(self, num_rays: int, generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
        """uniform draw with replacement over the chunk's pixels (RandomSampler of the reference's DataLoader)"""
        pick = torch.randint(0, len(self), (num_rays,), device=self.rgbs.device, generator=generator)
        return self.gather(pick)
