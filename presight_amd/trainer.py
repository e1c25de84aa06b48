"""One training iteration of the PreSight model on the HIP path: what ns/engine/trainer.py:463-505 (`train_iteration`) does
per step -- optimizer.zero_grad, forward, get_loss_dict, backward of the loss-scaled sum (the reference's GradScaler: fixed
2**10 while no inf/nan shows up), DDP gradient averaging, unscale + Adam (lr 1e-2, eps 1e-15, weight_decay 1e-5;
ns/configs/method_configs.py:158-168) -- with the model's own training callbacks around it (trainer.py:252-267).

The reference's Trainer also owns logging, checkpoint rotation, the viewer and the datamanager; those are out of scope
(SURVEY.md 8): this class is the timed region of bench.py and the object the parity tests step."""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from . import ops, prof
from .callbacks import TrainingCallbackAttributes, TrainingCallbackLocation
from .dist import FlatGrads, global_depth_clip as _depth_hook
from .optim import HipAdam
from .rays import RayBundle


def routed_groups(model) -> list:
    """the parameter groups whose "received a gradient" is decided on the device (FlatGrads.define_groups): the sub-fields of
    every routed module of a K > 1 tile (main field, each proposal network, the routed sky model)"""
    mods = [model.field] + list(model.proposal_networks)
    sky = getattr(model, "sky_model", None)
    if sky is not None and hasattr(sky, "fields"):
        mods.append(sky)
    groups = []
    for m in mods:
        if len(m.fields) > 1:
            seen = set()
            for f in m.fields:
                ps = [p for p in f.parameters() if id(p) not in seen]
                seen.update(id(p) for p in ps)
                groups.append(ps)
    return groups


class Trainer:
    """exchange = "allreduce": bucketed all-reduce overlapped with backward, Adam over everything on every rank;
               "sharded":   bucketed reduce-scatter overlapped with backward, Adam on the owned shard, all-gather of the
                            updated parameters overlapped with the next step's ray generation / proposal sampling."""

    def __init__(self, model, scene: Dict, world: int = 1, exchange: str = "allreduce", global_depth_clip: bool = False,
                 lr: float = 1e-2, eps: float = 1e-15, weight_decay: float = 1e-5, loss_scale: float = 2.0 ** 10):
        self.model, self.scene, self.world = model, scene, world
        groups = model.get_param_groups()
        # bucket-major order, in the order backward COMPLETES the groups on the GPU (buckets are exchanged strictly in this order).
        # With the proposal networks on their side stream (ops.side_stream) their chain ends ~0.5 ms before the main
        # chain writes its last gradient (the main hash table's), so "proposal_networks" goes first and its all-reduce runs under
        # the main chain's tail; on one stream "fields" (main field, sky, embeddings) are finished before the proposal backward
        # starts and travel underneath it.  A group that receives no gradient in a step (proposal nets off-schedule) is one
        # contiguous range to skip.
        first = ("proposal_networks", "fields") if (ops.SIDE_STREAM and torch.cuda.is_available()) else ("fields", "proposal_networks")
        order = [k for k in first if k in groups] + sorted(k for k in groups if k not in first)
        seen, uniq, sizes = set(), [], []
        for k in order:
            n0 = len(uniq)
            for p in groups[k]:  # the reference registers mlp_base = Sequential(grid, mlp): the same tensors appear twice -> dedup
                if p.requires_grad and p.numel() > 0 and id(p) not in seen:
                    seen.add(id(p))
                    uniq.append(p)
            sizes.append(len(uniq) - n0)
        assert seen == {id(p) for p in model.parameters() if p.requires_grad and p.numel() > 0}
        self.group_names = order
        sharded = exchange == "sharded" and world > 1
        self.grads = FlatGrads(uniq, bucket_sizes=sizes, shard_world=world if sharded else 1)
        # K > 1: whether a sub-field got samples is decided on the device; the optimizer kernel skips the ones that did not
        # (torch.optim.Adam with grad None), and under data parallelism the decision is agreed across ranks on the device
        rg = routed_groups(model)
        if rg:
            self.grads.define_groups(rg)
        if world > 1 and not model.config.use_same_proposal_network and os.environ.get("PRESIGHT_NO_OVERLAP") != "1":
            buckets, i = [], 0
            for n in sizes:
                buckets.append(uniq[i:i + n])
                i += n
            self.grads.enable_overlap(buckets, mode="sharded" if sharded else "allreduce")
            if sharded:
                model.param_gate = lambda name: self.grads.wait_params(self.group_names.index(name))
        self.exchange = "sharded" if sharded else "allreduce"
        if global_depth_clip and world > 1:
            ops.set_depth_clip_hook(_depth_hook())
        self.loss_scale = float(loss_scale)
        # the backward pass is seeded with the loss scale; the optimizer kernel unscales (GradScaler.step: unscale_, then step)
        self.opt = HipAdam(uniq, lr=lr, eps=eps, weight_decay=weight_decay, flat_grads=self.grads, grad_scale=1.0 / self.loss_scale)
        self.callbacks = model.get_training_callbacks(TrainingCallbackAttributes(optimizers=self.opt, grad_scaler=None, pipeline=None))
        self.step_idx = 0
        self._seed: Optional[torch.Tensor] = None
        self.update_props_every_step = False

    def _run_callbacks(self, where: TrainingCallbackLocation):
        for cb in self.callbacks:
            cb.run_callback_at_location(self.step_idx, where)

    def step(self, batch: Dict[str, torch.Tensor]):
        m, s = self.model, self.scene
        m.train()
        self._run_callbacks(TrainingCallbackLocation.BEFORE_TRAIN_ITERATION)
        self.grads.zero_()
        o, d, pa, dn = ops.generate_rays(batch["ray_indices"], s["c2w"], s["fx"], s["fy"], s["cx"], s["cy"])
        vid = batch["video_ids"] if "video_ids" in batch else batch["video_id"]  # synthetic batches / the reference's collated key
        meta = {"video_id": vid.view(-1, 1), "directions_norm": dn}
        times = batch.get("times")
        if times is None and "cam_times" in s and getattr(m, "dynamic_field", None) is not None:
            times = s["cam_times"][batch["ray_indices"][:, 0]]  # ray time = normalised timestamp of its camera frame
        rb = RayBundle(o, d, pa, camera_indices=batch["ray_indices"][:, 0:1], metadata=meta, times=None if times is None else times.view(-1, 1))
        if self.update_props_every_step:
            m.proposal_sampler._steps_since_update = 1 << 30  # proposal nets receive gradients EVERY step (upper bound of the schedule)
        out = m(rb)
        loss_dict = m.get_loss_dict(out, batch)
        # one concat + one reduction instead of a chain of scalar adds; the loss scale enters as the seed of the backward pass
        loss = torch.stack(list(loss_dict.values())).sum()
        if self._seed is None:
            self._seed = torch.full((), self.loss_scale, device=loss.device)
        loss.backward(gradient=self._seed)
        ops.join_side_streams()  # the proposal networks' backward ran on their side stream: the exchange / optimizer wait for it
        with prof.region("exchange_exposed"):
            self.grads.finish_exchange()
        with prof.region("adam"):
            self.opt.step()
        self._run_callbacks(TrainingCallbackLocation.AFTER_TRAIN_ITERATION)
        self.step_idx += 1
        return loss_dict, out
