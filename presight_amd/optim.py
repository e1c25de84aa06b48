"""Adam on the HIP kernel (ps_adam_step), torch.optim.Adam-compatible numerics for the reference's settings
(ns/engine/optimizers.py:73-170: one Adam per parameter group, lr 1e-2, eps 1e-15, weight_decay 1e-5)."""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import ctypes

import torch

from ._lib import check, lib
from .dist import intersect_ranges
from .ops import _p, _stream


class HipAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-15,
                 weight_decay: float = 1e-5, flat_grads=None, grad_scale: float = 1.0):
        """grad_scale: factor applied to every gradient inside the kernel before the weight decay is added.  1.0 (default) =
        the reference's default path: with update_grad_scaler=False its Trainer calls optimizer.step() on the gradients exactly
        as backward left them, loss scale included (ns/engine/trainer.py:481-486, optimizers.py:133-140).  1 / loss_scale = its
        update_grad_scaler=True branch, where GradScaler.step unscales first (optimizers.py:118-131).
        flat_grads: a presight_amd.dist.FlatGrads over exactly `params` -> parameters and both moments are moved into
        flat buffers of the same layout (every parameter's storage is re-pointed to a view of the flat buffer, values
        kept) and the whole update is ONE kernel launch over the flat range instead of one per tensor."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad and p.numel() > 0]
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("HipAdam: parameters must be contiguous fp32 CUDA tensors")
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.grad_scale = float(grad_scale)
        self.fused_armed = False  # enable_fused_tables: only a backward pass the trainer has armed applies the tables' update
        self.step_count = 0  # optimizer steps taken (informational; the bias corrections use the per-parameter counts)
        self.steps: List[int] = [0] * len(self.params)  # torch's state[p]["step"]
        self.flat = None
        if flat_grads is not None:
            if len(flat_grads.params) != len(self.params) or any(a is not b for a, b in zip(flat_grads.params, self.params)):
                raise ValueError("HipAdam: flat_grads must cover exactly the optimizer's parameters, in the same order")
            dev = self.params[0].device
            fp = torch.zeros(flat_grads.total, device=dev, dtype=torch.float32)  # padding stays 0 (grad 0, decay of 0)
            fm, fv = torch.zeros_like(fp), torch.zeros_like(fp)
            self.exp_avg, self.exp_avg_sq = [], []
            with torch.no_grad():
                for p, off in zip(self.params, flat_grads.offsets):
                    n = p.numel()
                    view = fp[off:off + n].view_as(p)
                    view.copy_(p.data)
                    p.data = view
                    self.exp_avg.append(fm[off:off + n].view_as(p))
                    self.exp_avg_sq.append(fv[off:off + n].view_as(p))
            self.flat = (fp, flat_grads.flat, fm, fv)
            self.flat_grads = flat_grads
        else:
            self.exp_avg = [torch.zeros_like(p) for p in self.params]
            self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]

    # ------------------------------------------------------------------ Adam inside the table backward (single-process training)
    def enable_fused_tables(self, indices: Sequence[int]) -> None:
        """The hash tables self.params[i], i in indices, are updated INSIDE their table backward (ps_grid_scatter_binned_adam: the
        accumulate pass applies this optimizer's element update to every slice it finishes, the gradient never reaches memory) instead
        of by step().  Valid when nobody needs the gradient itself: no exchange (one process / one tile per GPU), no found-inf check
        (update_grad_scaler=False), and one gradient contribution per table and step (a second one raises).  Same bits as step():
        both run csrc/adam_core.hpp::adam_update on the same fp32 gradient."""
        if self.flat is None:
            raise RuntimeError("HipAdam.enable_fused_tables: needs the flat buffers (flat_grads=...)")
        for i in indices:
            p = self.params[i]
            p._ps_fused_adam, p._ps_fused_index, p._ps_fused_done = self, i, False

    def disable_fused_tables(self) -> None:
        for p in self.params:
            if getattr(p, "_ps_fused_adam", None) is self:
                p._ps_fused_adam = None

    def fused_table_args(self, tables: Sequence[torch.nn.Parameter], routed: bool):
        """Called by the table backward (field_ops._scatter / _ms_scatter) right before its launch: -> the Adam arguments of
        ps_grid_scatter_binned_adam (routed: of ps_grid_scatter_binned_ms_adam) for THIS step's update of `tables`, or None when it has
        to go through step() after all.  Advances the host-side step counts and marks the tables as updated: step() skips them and
        FlatGrads.zero_ leaves their (still zero) gradient ranges alone."""
        if not self.fused_armed:
            return None  # a backward pass outside Trainer.step (gradient inspection, tests): plain gradients, step() does the update
        fg = self.flat_grads
        for t in tables:
            if getattr(t, "_ps_fused_done", False):
                raise RuntimeError("presight_amd: a hash table whose Adam update is fused into its backward received a second gradient "
                                   "contribution in one step (HipAdam.enable_fused_tables needs exactly one)")
        gids = [getattr(t, "_ps_group", None) if fg.n_groups else None for t in tables]
        if not routed and gids[0] is not None:
            return None  # a routed sub-field called on its own: whether it counts as updated is the device's decision
        idx = [t._ps_fused_index for t in tables]
        host = [i for i, g_ in zip(idx, gids) if g_ is None]
        if host and len({self.steps[i] for i in host}) != 1:
            return None
        step = self.steps[host[0]] + 1 if host else 0
        for i in host:
            self.steps[i] += 1
        for t in tables:
            t._ps_fused_done = True
        fp, fgr, fm, fv = self.flat
        args = (_p(fgr), _p(fp), _p(fm), _p(fv), self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, self.grad_scale, step)
        if not routed:
            return args
        if all(g_ is None for g_ in gids):
            return args + (None, None, None)
        key = tuple(-1 if g_ is None else g_ for g_ in gids)
        tbl = self._fused_gid_tables.get(key) if hasattr(self, "_fused_gid_tables") else None
        if tbl is None:
            if not hasattr(self, "_fused_gid_tables"):
                self._fused_gid_tables = {}
            tbl = torch.tensor(key, dtype=torch.int32, device=fp.device)
            self._fused_gid_tables[key] = tbl
        return args + (_p(tbl), _p(fg.group_flags), _p(fg.group_steps))

    @torch.no_grad()
    def step(self, skip=(), subset=None):
        """skip: indices of parameters whose update is withheld this step although they received a gradient (GradScaler.step
        found an inf / nan in their group): treated like parameters without a gradient.
        subset: only these parameter indices are updated by this call (the step is issued in pieces: presight_amd.trainer.Trainer
        runs the fields' piece on a second stream underneath the next step's proposal sampling); launches go to the current stream.
        torch.optim.Adam semantics per PARAMETER: state["step"] (the bias-correction exponent) advances only for parameters
        that received a gradient this step; the others are skipped entirely (no decay, no moment update), like torch with
        grad None after zero_grad(set_to_none=True).  Proposal networks (a gradient every ~6th step after warm-up) and
        sub-fields that saw no sample therefore keep their own, smaller step counts."""
        if subset is None or 0 in subset or not self.params:
            self.step_count += 1  # (a step issued in pieces counts once: with the piece that holds parameter 0)
        s = _stream()
        if self.flat is not None:
            fg = self.flat_grads
            fg._join_side_streams()
            idx = [i for i in fg.touched_params() if i not in skip and (subset is None or i in subset)
                   and not getattr(self.params[i], "_ps_fused_done", False)]  # (updated inside their table backward already)
            if skip and fg.n_groups:  # device-decided groups of withheld parameters: lower their flags
                for g_ in {getattr(self.params[i], "_ps_group", None) for i in skip} - {None}:
                    fg.group_flags[g_] = 0
            gid = [getattr(self.params[i], "_ps_group", None) if fg.n_groups else None for i in idx]
            for i, gr in zip(idx, gid):
                if gr is None:
                    self.steps[i] += 1  # (the step counts of device-decided groups live on the device: fg.group_steps)
            # adjacent touched parameters of the same group / with the same host step count merge into one range; in sharded
            # data-parallel mode this rank only updates the shard it owns (reduce-scatter -> Adam on the shard -> all-gather)
            runs: List[list] = []
            for i, gr in zip(idx, gid):
                a, b = fg.offsets[i], fg.offsets[i] + fg._pad(self.params[i].numel())
                key = ("g", gr) if gr is not None else ("s", self.steps[i])
                if runs and runs[-1][1] == a and runs[-1][2] == key:
                    runs[-1][1] = b
                else:
                    runs.append([a, b, key])
            owned = fg.owned_ranges()
            ranges = []
            for a, b, key in runs:
                for x, y in intersect_ranges([(a, b)], owned):
                    ranges.append((x, y - x, key))
            if ranges:
                n = len(ranges)
                starts = (ctypes.c_int64 * n)(*[r[0] for r in ranges])
                counts = (ctypes.c_int64 * n)(*[r[1] for r in ranges])
                steps = (ctypes.c_int * n)(*[(r[2][1] if r[2][0] == "s" else 0) for r in ranges])
                groups = (ctypes.c_int * n)(*[(r[2][1] if r[2][0] == "g" else -1) for r in ranges])
                check(lib().ps_adam_step_ranges(_p(self.flat[0]), _p(self.flat[1]), _p(self.flat[2]), _p(self.flat[3]), n, starts, counts,
                                                steps, groups, _p(fg.group_flags), _p(fg.group_steps), fg.n_groups, self.lr,
                                                self.betas[0], self.betas[1], self.eps, self.weight_decay, self.grad_scale, s),
                      "ps_adam_step_ranges")
            fg.gather_params(self.flat[0], [(a, b) for a, b, _ in runs])  # no-op unless the exchange is sharded
            return
        for i, (p, m, v) in enumerate(zip(self.params, self.exp_avg, self.exp_avg_sq)):
            g = p.grad
            if g is None or i in skip or (subset is not None and i not in subset):
                continue
            if not g.is_contiguous():
                g = g.contiguous()
            self.steps[i] += 1
            check(lib().ps_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                                     self.weight_decay, self.steps[i], self.grad_scale, s), "ps_adam_step")

    def zero_grad(self):
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()

    def param_steps(self) -> List[int]:
        """torch's state[p]["step"] per parameter; the counts of device-decided groups (FlatGrads.define_groups) are read back
        from the device (one synchronising copy: checkpointing / tests only)"""
        out = list(self.steps)
        fg = getattr(self, "flat_grads", None)
        if fg is not None and fg.n_groups:
            dev = fg.group_steps.tolist()
            for i, p in enumerate(self.params):
                gr = getattr(p, "_ps_group", None)
                if gr is not None:
                    out[i] = dev[gr]
        return out

    def state_dict(self):
        """Checkpoint payload.  With a sharded gradient exchange every rank's moments are valid on its OWNED shard only: they
        are all-gathered here (a collective: call it on every rank), so that any rank's state_dict is complete."""
        fg = getattr(self, "flat_grads", None)
        if fg is not None and self.flat is not None:
            fg.gather_flat(self.flat[2])
            fg.gather_flat(self.flat[3])
        return {"step": self.step_count, "steps": self.param_steps(), "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr}

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
        self.steps = list(sd.get("steps", [sd["step"]] * len(self.params)))
        fg = getattr(self, "flat_grads", None)
        if fg is not None and fg.n_groups:
            host = fg.group_steps.tolist()
            for i, p in enumerate(self.params):
                gr = getattr(p, "_ps_group", None)
                if gr is not None:
                    host[gr] = self.steps[i]
            fg.group_steps.copy_(torch.tensor(host, dtype=torch.int32))
        for a, b in zip(self.exp_avg, sd["exp_avg"]):
            a.copy_(b)
        for a, b in zip(self.exp_avg_sq, sd["exp_avg_sq"]):
            a.copy_(b)


class WarmupMultiStepSchedule:
    """Learning-rate schedule of the PreSight method configs (ns/engine/my_schedulers.py:50-70 with the arguments of
    ns/configs/method_configs.py:158-168): torch's ChainedScheduler([LinearLR(start_factor=0.01, total_iters=warmup_steps),
    MultiStepLR(milestones, gamma=0.33)]) in closed form,
        lr(t) = lr_init * (0.01 + 0.99 * min(t, warmup) / warmup) * gamma ** #{m in milestones : m <= t},
    applied to a HipAdam by step(): call once after every optimizer step, like Optimizers.scheduler_step_all."""

    def __init__(self, optimizer: HipAdam, lr_init: Optional[float] = None, max_steps: int = 1000000,
                 milestones: Sequence[int] = (500000, 750000, 900000), warmup_steps: Optional[int] = None, gamma: float = 0.33,
                 start_factor: float = 0.01):
        self.opt = optimizer
        self.lr_init = optimizer.lr if lr_init is None else lr_init
        self.milestones = sorted(int(m) for m in milestones)
        self.warmup = int(warmup_steps) if warmup_steps else 0
        self.gamma, self.start_factor, self.max_steps = gamma, start_factor, max_steps
        self.t = 0
        self.opt.lr = self.lr_at(0)

    def lr_at(self, t: int) -> float:
        warm = 1.0 if self.warmup <= 0 else self.start_factor + (1.0 - self.start_factor) * min(t, self.warmup) / self.warmup
        return self.lr_init * warm * self.gamma ** sum(1 for m in self.milestones if m <= t)

    def step(self, step: Optional[int] = None) -> float:
        self.t = self.t + 1 if step is None else step + 1
        self.opt.lr = self.lr_at(self.t)
        return self.opt.lr

    def get_last_lr(self) -> List[float]:
        return [self.opt.lr]

    def state_dict(self):
        return {"t": self.t, "lr_init": self.lr_init}

    def load_state_dict(self, sd):
        self.t = int(sd["t"])
        self.lr_init = float(sd.get("lr_init", self.lr_init))
        self.opt.lr = self.lr_at(self.t)
