"""Adam on the HIP kernel (ps_adam_step), torch.optim.Adam-compatible numerics for the reference's settings
(ns/engine/optimizers.py:73-170: one Adam per parameter group, lr 1e-2, eps 1e-15, weight_decay 1e-5)."""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch

from ._lib import check, lib
from .ops import _p, _stream


class HipAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-15,
                 weight_decay: float = 1e-5, flat_grads=None):
        """flat_grads: a presight_amd.dist.FlatGrads over exactly `params` -> parameters and both moments are moved into
        flat buffers of the same layout (every parameter's storage is re-pointed to a view of the flat buffer, values
        kept) and the whole update is ONE kernel launch over the flat range instead of one per tensor."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad and p.numel() > 0]
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("HipAdam: parameters must be contiguous fp32 CUDA tensors")
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
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

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        s = _stream()
        if self.flat is not None:
            # parameters without a gradient this step are skipped entirely (no decay, no moment update), like torch.optim
            # with grad None; the others are updated range by range (all of them -> a single launch)
            for a, b in self.flat_grads.touched_ranges():
                ptrs = [t.data_ptr() + 4 * a for t in self.flat]
                check(lib().ps_adam_step(ptrs[0], ptrs[1], ptrs[2], ptrs[3], b - a, self.lr, self.betas[0], self.betas[1], self.eps,
                                         self.weight_decay, self.step_count, s), "ps_adam_step")
            return
        for p, m, v in zip(self.params, self.exp_avg, self.exp_avg_sq):
            g = p.grad
            if g is None:
                continue
            if not g.is_contiguous():
                g = g.contiguous()
            check(lib().ps_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                                     self.weight_decay, self.step_count, s), "ps_adam_step")

    def zero_grad(self):
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr}

    def load_state_dict(self, sd):
        self.step_count = sd["step"]
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
