"""One training iteration of the PreSight model on the HIP path: what ns/engine/trainer.py:463-505 (`train_iteration`) does
per step -- optimizer.zero_grad, forward, get_loss_dict, backward of the loss-scaled sum, DDP gradient averaging, Adam
(lr 1e-2, eps 1e-15, weight_decay 1e-5; ns/configs/method_configs.py:158-168), scheduler step -- with the model's own training
callbacks around it (trainer.py:252-267).

Loss scale.  The reference builds GradScaler(init_scale=2**10) even in fp32 (trainer.py:70-73,132) and PreSight runs it with
`update_grad_scaler=False`: `grad_scaler.scale(loss).backward()` followed by `optimizers.optimizer_step_all()` = a plain
`optimizer.step()` on the STILL-SCALED gradients (trainer.py:481-486, optimizers.py:133-140) -- nothing ever unscales.  Adam is
invariant to the scale up to eps = 1e-15, but its L2 term is not: `weight_decay * p` is added to 1024 * dL/dp, i.e. the effective
weight decay is 1e-5 / 1024.  That is the default here (grad_scale 1.0 inside the Adam kernel).  `update_grad_scaler=True` is the
reference's other branch (optimizers.py:118-131, trainer.py:496-505): GradScaler.step per parameter group (unscale, skip the group's
step on inf / nan), GradScaler.update (halve on inf, double after 2000 clean steps), and the schedulers are stepped only when the
scale did not decrease.  Pinned by tests/golden/model_traj.npz (24 iterations of the reference's own loop).

The reference's Trainer also owns logging, checkpoint rotation, the viewer and the datamanager; those are out of scope
(SURVEY.md 8): this class is the timed region of bench.py and the object the parity tests step."""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from . import losses, ops, prof
from .callbacks import TrainingCallbackAttributes, TrainingCallbackLocation
from .dist import FlatGrads, exchanging, global_depth_clip as _depth_hook
from .optim import HipAdam, WarmupMultiStepSchedule
from .rays import RayBundle


def _some_module_in_eval_mode(model, depth: int = 3) -> bool:
    """whether `model` or any module up to `depth` levels below it is in eval mode (~100 flag reads on a K = 16 tile)"""
    if not model.training:
        return True
    if depth == 0:
        return False
    return any(_some_module_in_eval_mode(c, depth - 1) for c in model.children())


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
                 lr: float = 1e-2, eps: float = 1e-15, weight_decay: float = 1e-5, loss_scale: float = 2.0 ** 10,
                 update_grad_scaler: bool = False, max_num_iterations: Optional[int] = None, schedule: Optional[Dict] = None,
                 fused_table_adam: Optional[bool] = None, table_pieces: Optional[int] = None):
        """loss_scale = TrainerConfig.init_grad_scale, update_grad_scaler = TrainerConfig.update_grad_scaler (trainer.py:70-73).
        max_num_iterations: builds the learning-rate schedule of the PreSight method configs (method_configs.py:158-168: warm-up over
        max // 10 steps, x0.33 at max // 4, max // 2, 3 max // 4); `schedule` = explicit WarmupMultiStepSchedule keyword arguments;
        neither: constant learning rate."""
        self.model, self.scene, self.world = model, scene, world
        groups = model.get_param_groups()
        # Flat order = the order in which backward COMPLETES the gradients on the GPU; exchange buckets are consecutive runs of it and
        # go out strictly in this order, each as soon as it is complete (the reference's DDP overlaps its bucketed all-reduce with
        # backward the same way, ns/pipelines/PreSight/my_pipeline.py:121-124):
        #   proposal network n-1 .. 0   one bucket each.  Their backward depends on the interlevel loss only and runs on a side stream
        #                               (ops.side_stream): it ends long before the main chain does;
        #   main field MLPs             complete when the main MLP backward kernels and the gradient unpack are done, i.e. BEFORE the
        #                               main table backward starts;
        #   main hash tables            last.  One sub-field: the table is exchanged as level groups, one accumulate launch per group
        #                               (FlatGrads splits); routed tile: the K tables in sub-field groups, one accumulate launch per
        #                               group -- the reduce-scatter of group g runs underneath the accumulate launch of group g + 1.
        #   tail                        sky model + embeddings (the embeddings' gradient is an output of the main field's backward node):
        #                               ~0.1 MB, complete at the very end.
        # Only the last table piece and the tiny tail are complete at the end of backward: everything else is handed over while kernels
        # are still running.
        # A group that receives no gradient in a step (proposal nets off-schedule) is one contiguous range to skip.
        names: Dict[int, str] = {}
        for n, p in model.named_parameters(remove_duplicate=False):
            names.setdefault(id(p), n)  # (aliases -- mlp_base = Sequential(grid, mlp), dual_field.static_field -- keep the first name)
        is_table = lambda p: names.get(id(p), "").endswith("hash_table")  # noqa: E731
        seen, uniq, sizes, kinds = set(), [], [], []

        def add_bucket(plist, kind):
            n0 = len(uniq)
            for p in plist:  # the reference registers mlp_base = Sequential(grid, mlp): the same tensors appear twice -> dedup
                if p.requires_grad and p.numel() > 0 and id(p) not in seen:
                    seen.add(id(p))
                    uniq.append(p)
            if len(uniq) > n0:
                sizes.append(len(uniq) - n0)
                kinds.append(kind)

        # exchange = "sparse": the sharded exchange with the HASH TABLES' gradients travelling as the binned backward's records instead of
        # dense buffers (dist.FlatGrads._sparse_exchange): every network's tables form one record bucket of their own
        self._exchanging = world > 1 or exchanging()
        sparse_tables = exchange == "sparse" and self._exchanging
        sparse_pos: List[int] = []
        prop_nets = list(getattr(model, "proposal_networks", []))
        prop_ids = {id(p) for p in groups.get("proposal_networks", [])}
        if model.config.use_same_proposal_network or not prop_nets:
            if sparse_tables:
                raise ValueError("Trainer(exchange='sparse'): one record bucket per proposal network (use_same_proposal_network=False)")
            add_bucket(groups.get("proposal_networks", []), "proposal_networks")
        else:
            for i in reversed(range(len(prop_nets))):
                mine = [p for p in prop_nets[i].parameters() if id(p) in prop_ids]
                if sparse_tables:
                    n0 = len(sizes)
                    add_bucket([p for p in mine if is_table(p)], "proposal_networks")
                    sparse_pos += list(range(n0, len(sizes)))
                add_bucket(mine, "proposal_networks")
            add_bucket(groups.get("proposal_networks", []), "proposal_networks")  # (anything the loop did not reach)
        fields = list(groups.get("fields", []))
        # the main field's MLPs are complete after its backward kernels + the gradient unpack, BEFORE its table backward starts; the
        # embeddings (their gradient is an OUTPUT of the main field's backward node) and the sky model close the step: tail bucket
        add_bucket([p for p in fields if not is_table(p) and names.get(id(p), "").startswith("field.")], "fields")
        main_tables = [p for p in fields if is_table(p) and id(p) not in seen and names.get(id(p), "").startswith("field.")]
        # pieces the main hash tables' gradient is exchanged in (one accumulate launch + one collective each: the collective of piece g
        # runs under the accumulate launch of piece g + 1).  4 at full batches; a rank of a strong-scaled run (<= 16 k rays) should pass
        # 2: its accumulate launches are short and every extra piece costs launches, events and a collective on a host-bound step
        split_pieces = int(table_pieces) if table_pieces is not None else int(os.environ.get("PRESIGHT_TABLE_PIECES", "4"))
        n_table_buckets, table_split = 0, None
        if sparse_tables and main_tables:
            n0 = len(sizes)
            add_bucket(main_tables, "fields")  # all K tables: ONE record bucket (ownership = contiguous 1 / world of their slices)
            sparse_pos += list(range(n0, len(sizes)))
            n_table_buckets = 1
        elif len(main_tables) == 1:
            add_bucket(main_tables, "fields")
            table_split = len(sizes) - 1
            n_table_buckets = 1
            # pieces = whole level groups (field_ops._scatter launches the accumulate pass per group): the largest divisor of L
            L = int(model.field.fields[0].mlp_base_grid.num_levels)
            split_pieces = max(d for d in range(1, max(1, split_pieces) + 1) if L % d == 0)
        elif main_tables:
            G = max(1, min(split_pieces, len(main_tables)))
            while len(main_tables) % G:
                G -= 1
            per = len(main_tables) // G
            for gi in range(G):
                add_bucket(main_tables[gi * per:(gi + 1) * per], "fields")
            n_table_buckets = G
        add_bucket(fields, "fields")  # tail: sky model, appearance / video embeddings (small)
        for k in sorted(groups):
            if k not in ("proposal_networks", "fields"):
                add_bucket(groups[k], k)
        assert seen == {id(p) for p in model.parameters() if p.requires_grad and p.numel() > 0}
        self.group_names = kinds  # bucket -> optimizer group name
        # (a process group of one rank with PRESIGHT_EXCHANGE_WORLD_OF_ONE=1 exchanges like any other: RCCL smoke run on a one-GPU box)
        sharded = exchange in ("sharded", "sparse") and self._exchanging
        overlap = ((self._exchanging or os.environ.get("PRESIGHT_DRY_OVERLAP") == "1") and not model.config.use_same_proposal_network
                   and os.environ.get("PRESIGHT_NO_OVERLAP") != "1")
        splits = {table_split: split_pieces} if (overlap and table_split is not None and split_pieces > 1) else None
        self.grads = FlatGrads(uniq, bucket_sizes=sizes, shard_world=world if sharded else 1, splits=splits)
        # K > 1: whether a sub-field got samples is decided on the device; the optimizer kernel skips the ones that did not
        # (torch.optim.Adam with grad None), and under data parallelism the decision is agreed across ranks on the device
        rg = routed_groups(model)
        if rg:
            self.grads.define_groups(rg)
        if overlap:
            buckets, i = [], 0
            for n in sizes:
                buckets.append(uniq[i:i + n])
                i += n
            if len(main_tables) > 1 and n_table_buckets > 1:
                main_tables[0]._ps_ms_parts = n_table_buckets  # field_ops._ms_scatter: one accumulate launch per sub-field group
            self.grads.enable_overlap(buckets, mode="sharded" if sharded else "allreduce", dry=not self._exchanging,
                                      sparse=sparse_pos if sparse_tables else ())
            if sharded:
                # (a declared bucket may have become several exchange buckets: the pieces of the split table)
                gate: Dict[str, list] = {}
                for b in self.grads._buckets:
                    pi = next(j for j, (a0, a1) in enumerate(self.grads.bucket_ranges) if a0 <= b["range"][0] < a1)
                    gate.setdefault(kinds[pi], []).append(b["index"])
                model.param_gate = lambda name: [self.grads.wait_params(j) for j in gate.get(name, ())]
        self._prop_buckets = [b["index"] for b in self.grads._buckets
                              if kinds[next(j for j, (a0, a1) in enumerate(self.grads.bucket_ranges) if a0 <= b["range"][0] < a1)] == "proposal_networks"]
        self.exchange = ("sparse" if sparse_tables else "sharded") if sharded else "allreduce"
        if global_depth_clip and self._exchanging:
            ops.set_depth_clip_hook(_depth_hook())
        self.loss_scale = float(loss_scale)
        self.update_grad_scaler = bool(update_grad_scaler)
        self._growth_tracker = 0  # GradScaler defaults: growth_factor 2, backoff_factor 0.5, growth_interval 2000
        self._bucket_sizes = sizes
        # the backward pass is seeded with the loss scale; by default Adam steps on the scaled gradients like the reference
        self.opt = HipAdam(uniq, lr=lr, eps=eps, weight_decay=weight_decay, flat_grads=self.grads,
                           grad_scale=1.0 / self.loss_scale if self.update_grad_scaler else 1.0)
        if schedule is None and max_num_iterations is not None:
            schedule = dict(max_steps=max_num_iterations, warmup_steps=max_num_iterations // 10,
                            milestones=[max_num_iterations // 4, max_num_iterations // 2, max_num_iterations * 3 // 4])
        # Single-process training (world == 1: also one tile per GPU, docs/building_priors.md:7-44) exchanges no gradients, so the hash
        # tables' Adam step runs INSIDE their table backward (HipAdam.enable_fused_tables / ps_grid_scatter_binned_adam): the gradient of
        # a table is never written, read back or zeroed -- 12 of the 40 bytes a table entry moves per step, 940 M entries on a production
        # tile.  Same bits as the separate step (one shared element update).  Off: PRESIGHT_FUSED_TABLE_ADAM=0, any exchange, the
        # found-inf check of update_grad_scaler (it needs the gradients).  The dual field's static and proposal tables qualify (one
        # evaluation per step); its 4-D table (three position sets, its own scatter entry point) keeps the separate update.
        can_fuse = not self._exchanging and not overlap and not self.update_grad_scaler
        if fused_table_adam and not can_fuse:
            raise ValueError("Trainer(fused_table_adam=True) needs world == 1, no exchange overlap and update_grad_scaler=False")
        self.fused_table_adam = can_fuse and (os.environ.get("PRESIGHT_FUSED_TABLE_ADAM", "1") != "0" if fused_table_adam is None else bool(fused_table_adam))
        if self.fused_table_adam:
            # exactly ONE gradient contribution per table and step: with use_same_proposal_network (nerfacto_nusc_ms.py:263) the single
            # proposal network is evaluated once per proposal iteration, so its tables receive two -- they keep the separate update
            shared_prop = bool(model.config.use_same_proposal_network) and int(getattr(model.config, "num_proposal_iterations", 2)) > 1
            self.opt.enable_fused_tables([i for i, p in enumerate(uniq) if is_table(p) and not (shared_prop and id(p) in prop_ids)])
        self.scheduler = WarmupMultiStepSchedule(self.opt, lr_init=lr, **schedule) if schedule is not None else None
        self.callbacks = model.get_training_callbacks(TrainingCallbackAttributes(optimizers=self.opt, grad_scaler=None, pipeline=None))
        self.step_idx = 0
        # Pipelined optimizer step (opt-in, single process: world == 1 / one tile per GPU): the proposal networks' Adam runs right
        # behind backward, the FIELDS' Adam -- 26 GB of pure HBM streaming on a production tile, 4.5 ms -- on a second stream underneath
        # the next iteration's chunk gather, ray generation and proposal sampling (gather- and latency-bound kernels that read proposal
        # parameters only); the compute stream waits for it right before the main field's forward (model.param_gate, the hook the
        # sharded exchange uses for its all-gather).  Same arithmetic, same order of updates; join() before reading parameters.
        self.pipeline_adam = False
        self._pipe = None
        self._seed: Optional[torch.Tensor] = None
        self._seed_value = self.loss_scale
        self.update_props_every_step = False
        self._inconsistent: Optional[str] = None  # set when an iteration raised after a fused table update (see step())

    def clear_failure(self):
        """after restoring a consistent state (checkpoint) following an iteration that raised mid-backward with fused table updates"""
        self._inconsistent = None

    def _scaler_step(self) -> bool:
        """update_grad_scaler=True: GradScaler.step per parameter group + GradScaler.update (optimizers.py:118-131,
        trainer.py:496-498).  Like torch's GradScaler this reads the found-inf flags on the host (one sync per step).
        -> whether the scale did NOT decrease (the schedulers are stepped only then)."""
        fg = self.grads
        fg._join_side_streams()
        touched = set(fg.touched_params())
        # the reference decides per OPTIMIZER = per parameter group ("proposal_networks" / "fields", optimizers.py:118-131): an inf
        # anywhere in a group's gradients withholds the whole group's step.  The exchange buckets are finer than that (one per proposal
        # network, MLPs / table pieces / tail of the fields), so the per-bucket flags are OR-ed per group name.
        kinds = sorted(set(self.group_names))
        by_kind, flags, i = {k: [] for k in kinds}, {k: [] for k in kinds}, 0
        for n, kind in zip(self._bucket_sizes, self.group_names):
            idx = [j for j in range(i, i + n) if j in touched]
            i += n
            if not idx:
                continue  # optimizers.py:130: a group without gradients is not stepped (and reports no inf)
            a, b = fg.offsets[idx[0]], fg.offsets[idx[-1]] + self.opt.params[idx[-1]].numel()
            by_kind[kind].extend(idx)
            flags[kind].append(~torch.isfinite(fg.flat[a:b]).all())
        found = torch.stack([torch.stack(flags[k]).any() if flags[k] else torch.zeros((), dtype=torch.bool, device=fg.flat.device)
                             for k in kinds]).to(torch.int32)
        if self._exchanging and torch.distributed.is_available() and torch.distributed.is_initialized():
            # after a reduce-scatter a rank holds the reduced values of ITS shard only: a rank that neither produced nor owns the
            # inf element would not see it and would step while the others skip -> agree on the flags (GradScaler does the same across
            # its per-device found_inf tensors)
            torch.distributed.all_reduce(found, op=torch.distributed.ReduceOp.MAX)
        found = found.tolist()
        skip, any_inf = set(), False
        for k, f in zip(kinds, found):
            if f:
                any_inf = True
                skip.update(by_kind[k])
        self.opt.grad_scale = 1.0 / self.loss_scale
        self.opt.step(skip=skip)
        if any_inf:
            self.loss_scale *= 0.5
            self._growth_tracker = 0
        else:
            self._growth_tracker += 1
            if self._growth_tracker == 2000:
                self.loss_scale *= 2.0
                self._growth_tracker = 0
        return not any_inf

    def _pipe_state(self):
        if self._pipe is None:
            fg = self.grads
            fields = {i for i, off in enumerate(fg.offsets)
                      if self.group_names[next(j for j, (a0, a1) in enumerate(fg.bucket_ranges) if a0 <= off < a1)] == "fields"}
            ranges = []
            for (a0, a1), kind in zip(fg.bucket_ranges, self.group_names):
                if kind == "fields":
                    if ranges and ranges[-1][1] == a0:
                        ranges[-1] = (ranges[-1][0], a1)
                    else:
                        ranges.append((a0, a1))
            self._pipe = dict(stream=torch.cuda.Stream(device=fg.flat.device), fields=fields, others=set(range(len(fg.offsets))) - fields,
                              ranges=ranges, event=None, zeroed=False)
            prev_gate = self.model.param_gate

            def gate(name):
                if prev_gate is not None:
                    prev_gate(name)
                if name == "fields" and self._pipe["event"] is not None:
                    torch.cuda.current_stream().wait_event(self._pipe["event"])
                    self._pipe["event"] = None

            self.model.param_gate = gate
        return self._pipe

    def join(self):
        """the compute stream waits for an optimizer piece that is still running on the pipeline stream (call before reading
        parameters outside step(): evaluation, checkpoints, the end of a timed region)"""
        if self._pipe is not None and self._pipe["event"] is not None:
            torch.cuda.current_stream().wait_event(self._pipe["event"])
            self._pipe["event"] = None

    def _begin_step(self):
        """clear the gradients of the previous step; -> the pipeline state when this step's optimizer runs pipelined"""
        pipe = self._pipe_state() if (self.pipeline_adam and not self._exchanging and not self.update_grad_scaler and not self.fused_table_adam
                                      and torch.cuda.is_available()) else None
        if pipe is None:
            self.join()
        # (pipelined: the fields' gradients were cleared on the pipeline stream right behind their Adam update)
        self.grads.zero_(already_zeroed=pipe["ranges"] if (pipe is not None and pipe["zeroed"]) else None)
        if pipe is not None:
            pipe["zeroed"] = False
        return pipe

    def _optimizer_step(self, pipe) -> bool:
        """-> whether the schedulers may step (the loss scale did not decrease)"""
        if self.update_grad_scaler:
            return self._scaler_step()
        if pipe is None:
            self.opt.step()  # on the scaled gradients (optimizer_step_all, optimizers.py:133-140)
            return True
        from .dist import intersect_ranges

        self.opt.step(subset=pipe["others"])  # proposal networks: the next iteration needs them first
        try:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(pipe["stream"]):
                pipe["stream"].wait_event(ev)
                self.opt.step(subset=pipe["fields"])
                for a, b in intersect_ranges(self.grads._dirty or [], pipe["ranges"]):
                    self.grads.flat[a:b].zero_()
                done = torch.cuda.Event()
                done.record(pipe["stream"])
        except BaseException:
            # the proposal networks have taken their step, the fields have not (or only partly): one optimizer step apart
            self._inconsistent = f"iteration {self.step_idx}, between the two pieces of the pipelined optimizer step"
            raise
        pipe["event"], pipe["zeroed"] = done, self.grads._dirty is not None
        return True

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self) -> Dict:
        """what ns/engine/trainer.py:432-460 saves next to the pipeline's state_dict -- step, optimizers, schedulers, scalers -- for this
        trainer: the Adam state (moments, per-parameter step counts incl. the device-decided sub-field groups), the schedule position,
        the loss scale, and the proposal sampler's update-schedule counters (the reference keeps those on the sampler object, so a
        resumed run restarts them at 0: here they are saved, see load_state_dict)"""
        self.join()
        ps = self.model.proposal_sampler
        return {"step": self.step_idx, "optimizer": self.opt.state_dict(), "scheduler": None if self.scheduler is None else self.scheduler.state_dict(),
                "scalers": {"scale": self.loss_scale, "growth_tracker": self._growth_tracker},
                "proposal_sampler": {"steps_since_update": ps._steps_since_update, "step": ps._step, "anneal": ps._anneal}}

    def load_state_dict(self, sd: Dict, restore_sampler: bool = True):
        """restore_sampler=False reproduces the reference's resume exactly (its sampler counters restart; ns/engine/trainer.py:396-429
        loads step / pipeline / optimizers / schedulers / scalers only)"""
        self.join()
        self.step_idx = int(sd["step"])
        self.opt.load_state_dict(sd["optimizer"])
        if self.scheduler is not None and sd.get("scheduler") is not None:
            self.scheduler.load_state_dict(sd["scheduler"])
        self.loss_scale = float(sd["scalers"]["scale"])
        self._growth_tracker = int(sd["scalers"]["growth_tracker"])
        if restore_sampler and "proposal_sampler" in sd:
            ps = self.model.proposal_sampler
            ps._steps_since_update, ps._step = int(sd["proposal_sampler"]["steps_since_update"]), int(sd["proposal_sampler"]["step"])
            ps.set_anneal(float(sd["proposal_sampler"]["anneal"]))

    def _run_callbacks(self, where: TrainingCallbackLocation):
        for cb in self.callbacks:
            cb.run_callback_at_location(self.step_idx, where)

    def step(self, batch: Dict[str, torch.Tensor]):
        m, s = self.model, self.scene
        if self._inconsistent:
            raise RuntimeError("presight_amd Trainer: an earlier iteration raised AFTER part of the parameters had taken their Adam step "
                               f"({self._inconsistent}); they and the other parameters are one step apart.  Restore a checkpoint "
                               "(load_state_dict + the model's state_dict) and call clear_failure() to continue")
        # nn.Module.train() walks every sub-module (~1000 of them on a K = 16 tile: 2 ms of host time per step, and at 8192 rays per rank
        # the production tile's step is bound by the host's enqueue time, tools/dbg/host_bound.py), so the full walk only runs when the
        # model is not in training mode -- where "the model" includes the modules whose mode the step actually reads (samplers, collider,
        # fields and their sub-fields: three levels), so that a `model.field.eval()` or a callback's `.eval()` on a sub-module does not
        # silently survive into training as it would behind a check of the top-level flag alone (the reference calls pipeline.train()
        # every iteration, ns/engine/trainer.py:473)
        if _some_module_in_eval_mode(m):
            m.train()
        self._run_callbacks(TrainingCallbackLocation.BEFORE_TRAIN_ITERATION)
        pipe = self._begin_step()
        o, d, pa, dn = ops.generate_rays(batch["ray_indices"], s["c2w"], s["fx"], s["fy"], s["cx"], s["cy"])
        vid = batch["video_ids"] if "video_ids" in batch else batch["video_id"]  # synthetic batches / the reference's collated key
        meta = {"video_id": vid.view(-1, 1), "directions_norm": dn}
        times = batch.get("times")
        if times is None and "cam_times" in s and getattr(m, "dynamic_field", None) is not None:
            times = s["cam_times"][batch["ray_indices"][:, 0]]  # ray time = normalised timestamp of its camera frame
        rb = RayBundle(o, d, pa, camera_indices=batch["ray_indices"][:, 0:1], metadata=meta, times=None if times is None else times.view(-1, 1))
        if self.update_props_every_step:
            m.proposal_sampler._steps_since_update = 1 << 30  # proposal nets receive gradients EVERY step (upper bound of the schedule)
        if self._prop_buckets:
            # off-schedule step (the same decision on every rank: it is a function of the step count, ray_samplers.py:586): the
            # proposal buckets leave the launch order now, the buckets behind them are handed over during backward as usual
            ps = m.proposal_sampler
            if not (ps._steps_since_update > ps.update_sched(ps._step) or ps._step < 10):
                self.grads.skip_buckets(self._prop_buckets)
        self.opt.fused_armed = self.fused_table_adam  # table backward nodes of THIS backward pass apply their tables' Adam step
        completed = False
        try:
            # the loss scale enters as the SEED of the backward pass (grad_scaler.scale(loss).backward(), ns/engine/trainer.py:481); it is
            # registered before the forward so that the fused loss kernels store their gradients already multiplied with it and the
            # loss nodes' backward launches nothing (losses.set_seed_hint)
            if self._seed is None or float(self._seed_value) != self.loss_scale:
                self._seed = torch.full((), self.loss_scale, device=batch["ray_indices"].device)
                self._seed_value = self.loss_scale
            losses.set_seed_hint(self._seed, self.loss_scale)
            out = m(rb, jitters=list(batch["jitter"])) if "jitter" in batch else m(rb)  # stored draws: parity runs only
            loss_dict = m.get_loss_dict(out, batch)
            loss = losses.loss_sum(loss_dict)  # functools.reduce(torch.add, ...): formed by the launch that finished the terms
            loss.backward(gradient=self._seed)
            ops.join_side_streams()  # the proposal networks' backward ran on their side stream: the exchange / optimizer wait for it
            with prof.region("exchange_exposed"):
                self.grads.finish_exchange()
            with prof.region("adam"):
                scale_kept = self._optimizer_step(pipe)
            completed = True
        finally:
            losses.set_seed_hint(None)
            self.opt.fused_armed = False  # (also when the iteration raised: a later backward pass must not update anything)
            if not completed and self.fused_table_adam and any(getattr(p, "_ps_fused_done", False) for p in self.opt.params):
                # the fused update is applied DURING backward: tables that had their turn before the exception are one optimizer step
                # (and one bias-correction count) ahead of everything else.  Not recoverable in place -> refuse to go on silently
                self._inconsistent = f"iteration {self.step_idx}"
        if self.scheduler is not None and scale_kept:  # trainer.py:499-505
            self.scheduler.step()
        self._run_callbacks(TrainingCallbackLocation.AFTER_TRAIN_ITERATION)
        self.step_idx += 1
        return loss_dict, out
