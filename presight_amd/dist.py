"""Data-parallel gradient exchange for one tile trained on N GPUs (one process per GPU, RCCL over xGMI through
torch.distributed's "nccl" backend; "gloo" on CPU for tests).

The reference wraps the model in DDP(find_unused_parameters=True) (ns/pipelines/PreSight/my_pipeline.py:121-124): every
parameter gradient is averaged over ranks each step.  Here all gradients live in ONE flat fp32 buffer (the parameters'
.grad tensors are views into it), so the exchange is a single large all-reduce per group with no packing copies:
 * xGMI is point-to-point, so few, large messages are what keeps all 7 links busy;
 * "unused parameter" handling is free: a sub-field that saw no sample this step just contributes zeros."""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor


def _mark_touched(p):
    p._ps_touched = True


class FlatGrads:
    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad and p.numel() > 0]
        pad = lambda n: (n + 3) // 4 * 4  # noqa: E731  keep every view 16-byte aligned (vectorised optimizer kernels)
        total = sum(pad(p.numel()) for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.offsets: List[int] = []
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            # opt in to in-place gradient accumulation by the HIP backward kernels (presight_amd.ops.grad_sink): the buffer
            # is zeroed once per step (zero_()), so "+=" from any number of uses of the parameter is the full gradient
            p._ps_direct_grad = True
            p._ps_touched = False
            p.register_post_accumulate_grad_hook(_mark_touched)  # gradients that arrive through autograd's own accumulation
            self.offsets.append(off)
            off += pad(n)
        self.total = total
        self._pad = pad
        self.flags_may_differ_across_ranks = False  # set when routing can leave a sub-field without samples on one rank only

    def zero_(self):
        """Start of a step: gradients to zero, "received a gradient this step" flags cleared (torch's zero_grad(set_to_none=True)
        + "grad is None -> the optimizer skips the parameter" semantics, without freeing the flat buffer)."""
        self.flat.zero_()
        for p in self.params:
            p._ps_touched = False

    def touched(self, group: Optional[dist.ProcessGroup] = None) -> List[bool]:
        flags = [bool(p._ps_touched) for p in self.params]
        if self.flags_may_differ_across_ranks and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            t = torch.tensor(flags, dtype=torch.int32, device=self.flat.device)  # DDP: a parameter used on ANY rank gets a gradient
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            flags = [bool(v) for v in t.tolist()]
        return flags

    def touched_ranges(self, group: Optional[dist.ProcessGroup] = None) -> List[tuple]:
        """[(first float, one-past-last float)] of the flat buffer covering exactly the parameters that received a gradient
        this step, adjacent parameters merged (all touched -> one range)."""
        out: List[list] = []
        for p, off, t in zip(self.params, self.offsets, self.touched(group)):
            if not t:
                continue
            end = off + self._pad(p.numel())
            if out and out[-1][1] == off:
                out[-1][1] = end
            else:
                out.append([off, end])
        return [tuple(r) for r in out]

    def all_reduce_mean(self, group: Optional[dist.ProcessGroup] = None, async_op: bool = False):
        """Average the flat gradient buffer over the ranks of `group` (no-op without an initialised process group)."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return None
        world = dist.get_world_size(group)
        self.flat.div_(world)
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def init_from_env(device_type: str = "cuda") -> tuple:
    """(rank, local_rank, world_size) from the torchrun environment; initialises the process group when world_size > 1."""
    import os

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = "nccl" if device_type == "cuda" else "gloo"
        if device_type == "cuda":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world
