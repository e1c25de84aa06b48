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
        """uniform draw WITH replacement over the chunk's pixels (a convenience for tests / synthetic runs; the reference's
        loader walks a shuffled permutation of the chunk: ChunkFeed / epoch_order below)"""
        pick = torch.randint(0, len(self), (num_rays,), device=self.rgbs.device, generator=generator)
        return self.gather(pick)


def epoch_order(num_pixels: int, world: int = 1, rank: int = 0, seed: int = 0, epoch: int = 0, shuffle: bool = True) -> Tensor:
    """Pixel slots of one pass over a chunk for rank `rank`, in the order the reference's loader yields them
    (ns/data/PreSight/my_datamanager.py:203-212: DataLoader over the ImageChunk with torch's DistributedSampler —
    randperm(seed + epoch) on the host, padded by wrapping to a multiple of `world`, rank r takes every world-th slot
    starting at r).  int64 [ceil(P / world)] host tensor."""
    import math

    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(num_pixels, generator=g)
    else:
        idx = torch.arange(num_pixels)
    total = math.ceil(num_pixels / world) * world
    pad = total - num_pixels
    if pad > 0:
        idx = torch.cat([idx, idx[:pad]]) if pad <= num_pixels else torch.cat([idx, idx.repeat(math.ceil(pad / num_pixels))[:pad]])
    return idx[rank:total:world]


class ChunkFeed:
    """The training data feed of one rank: the reference keeps ONE chunk of pixels in memory, walks it
    once in shuffled order in batches of train_num_rays_per_batch // world (drop_last), and meanwhile a background executor
    loads the next chunk (my_datamanager.py:214-236, 257-285; my_dataset.py:165-205).  Here the chunk is resident in HBM and a
    batch is ONE gather launch; the next chunk is produced by `load_chunk(chunk_index)` on a background thread, uploaded on a
    side stream (pinned staging + non_blocking copies) and swapped in when the current chunk is exhausted — the training
    stream only ever waits on an event.

    load_chunk(i) -> dict(rgbs [P,3], pixel_indices, image_indices, video_ids, widths [P], skies / depths / features optional),
    host or device tensors (the fields of the reference's ImageChunk)."""

    def __init__(self, load_chunk, batch_size: int, device, world: int = 1, rank: int = 0, seed: int = 0, prefetch: bool = True):
        from concurrent.futures import ThreadPoolExecutor

        self.load_chunk, self.batch_size, self.device = load_chunk, int(batch_size), torch.device(device)
        self.world, self.rank, self.seed = world, rank, seed
        self.chunk_index = -1
        self.chunk: Optional[DeviceImageChunk] = None
        self.order: Optional[Tensor] = None
        self.pos = 0
        self.chunks_loaded = 0
        self._pool = ThreadPoolExecutor(1) if prefetch else None
        self._stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._next = None
        self._start_load(0)

    def _produce(self, i: int):
        """background thread: build chunk i and upload it on the side stream -> (DeviceImageChunk, order, ready event)"""
        raw = self.load_chunk(i)
        with torch.cuda.stream(self._stream):
            dev = {}
            for k, v in raw.items():
                if v is None:
                    dev[k] = None
                elif v.is_cuda:
                    dev[k] = v
                else:
                    dev[k] = v.pin_memory().to(self.device, non_blocking=True)
            order = epoch_order(raw["rgbs"].shape[0], self.world, self.rank, self.seed).pin_memory().to(self.device, non_blocking=True)
            chunk = DeviceImageChunk(**dev)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return chunk, order, ev

    def _start_load(self, i: int):
        self._next = self._pool.submit(self._produce, i) if self._pool is not None else None
        self._next_index = i

    def _swap(self):
        res = self._next.result() if self._next is not None else self._produce(self._next_index)
        old_chunk, old_order = self.chunk, self.order
        self.chunk, self.order, ev = res
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)  # device-side wait; the host does not block on the upload
        # The chunk tensors were allocated on the side stream but are read by gather kernels on the training stream, and the
        # host runs ahead of the GPU: without this the caching allocator would hand the OLD chunk's blocks (freed right here)
        # back to the side stream, and the next upload could overwrite them while a gather of the old chunk is still queued.
        # record_stream keeps a block from being reused until the work enqueued on `cur` so far has finished.
        for obj in (old_chunk, self.chunk):
            if obj is not None:
                for name in ("rgbs", "pixel_indices", "image_indices", "video_ids", "widths", "skies", "depths", "features"):
                    t = getattr(obj, name)
                    if t is not None and t.is_cuda:
                        t.record_stream(cur)
        for t in (old_order, self.order):
            if t is not None and t.is_cuda:
                t.record_stream(cur)
        self.chunk_index = self._next_index
        self.pos = 0
        self.chunks_loaded += 1
        self._start_load(self.chunk_index + 1)

    def next_batch(self) -> Dict[str, Tensor]:
        """the collated batch the reference's `next(iter_train_image_batch_dataloader)` yields (+ ray_indices), on the device"""
        while self.chunk is None or self.pos + self.batch_size > self.order.shape[0]:  # drop_last, then the next chunk
            self._swap()
            if self.order.shape[0] < self.batch_size:
                raise RuntimeError(f"ChunkFeed: a chunk of {self.order.shape[0]} pixels per rank cannot fill a batch of {self.batch_size}")
        pick = self.order[self.pos:self.pos + self.batch_size]
        self.pos += self.batch_size
        return self.chunk.gather(pick)

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=False, cancel_futures=True)
