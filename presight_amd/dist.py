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
    """autograd's own accumulation finished for p (post-accumulate-grad hook); the HIP backward nodes that write gradients
    in place call presight_amd.ops.mark_touched, which ends up here too"""
    if p._ps_touched:
        return  # autograd also runs this hook for a parameter whose backward node wrote the gradient in place and returned None
    cb = getattr(p, "_ps_on_touch", None)
    if cb is not None:
        cb(p)
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
        self._buckets: List[dict] = []
        self._group = None
        self._comm_stream = None

    def zero_(self):
        """Start of a step: gradients to zero, "received a gradient this step" flags cleared (torch's zero_grad(set_to_none=True)
        + "grad is None -> the optimizer skips the parameter" semantics, without freeing the flat buffer)."""
        self.flat.zero_()
        for p in self.params:
            p._ps_touched = False
        for b in self._buckets:
            b["seen"], b["launched"], b["work"] = 0, False, None

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

    # ------------------------------------------------------------------ overlapped, bucketed exchange
    def enable_overlap(self, buckets: List[List[torch.nn.Parameter]], group: Optional[dist.ProcessGroup] = None):
        """Exchange the gradient buffer in buckets, each as soon as it is complete, on a side stream, while the backward of the
        remaining parameters is still running (the reference's DDP overlaps its bucketed all-reduce with backward the same
        way).  A bucket = parameters that are contiguous in the flat buffer (e.g. one optimizer group); it is launched when
        every one of its parameters has received its gradient of this step; whatever has not been launched by then
        (parameters that got no gradient) goes out in finish_exchange().  A parameter must receive at most ONE gradient
        contribution per step (checked: a second one after the launch raises)."""
        index = {id(p): i for i, p in enumerate(self.params)}
        self._buckets = []
        for plist in buckets:
            ids = sorted({index[id(p)] for p in plist if id(p) in index})
            if not ids:
                continue
            if ids != list(range(ids[0], ids[-1] + 1)):
                raise ValueError("FlatGrads.enable_overlap: the parameters of a bucket must be contiguous in the flat buffer")
            a, b = self.offsets[ids[0]], self.offsets[ids[-1]] + self._pad(self.params[ids[-1]].numel())
            self._buckets.append(dict(range=(a, b), n=len(ids), seen=0, launched=False, work=None))
            for i in ids:
                self.params[i]._ps_bucket = len(self._buckets) - 1
                self.params[i]._ps_on_touch = self._on_touch
        self._group = group
        if self.flat.is_cuda:
            self._comm_stream = torch.cuda.Stream(device=self.flat.device)

    def _distributed(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self._group) > 1

    def _on_touch(self, p):
        b = self._buckets[p._ps_bucket]
        if b["launched"]:
            i = next(k for k, q in enumerate(self.params) if q is p)
            raise RuntimeError(f"FlatGrads: parameter #{i} {tuple(p.shape)} received a second gradient contribution after its bucket "
                               "had been handed to the all-reduce; build the trainer without enable_overlap for this model")
        if not p._ps_touched:
            b["seen"] += 1
            if b["seen"] == b["n"]:
                self._launch(b)

    def _launch(self, b):
        b["launched"] = True
        if not self._distributed():
            return
        world = dist.get_world_size(self._group)
        seg = self.flat[b["range"][0]:b["range"][1]]
        if self._comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record()  # everything enqueued so far on the compute stream = the complete gradients of this bucket
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                seg.div_(world)
                b["work"] = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self._group, async_op=True)
        else:
            seg.div_(world)
            b["work"] = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self._group, async_op=True)

    def finish_exchange(self):
        """after backward: launch the buckets that are still local, then make the compute stream wait for all of them"""
        if not self._buckets:
            return self.all_reduce_mean(self._group)
        for b in self._buckets:
            if not b["launched"]:
                if b["seen"] == 0 and not self.flags_may_differ_across_ranks:
                    b["launched"] = True  # no parameter of the bucket got a gradient on any rank (schedule-driven): nothing to exchange
                else:
                    self._launch(b)
        for b in self._buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
        return None

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
        backend = os.environ.get("PRESIGHT_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        if os.environ.get("PRESIGHT_SINGLE_DEVICE") == "1":
            local_rank = 0  # functional test of the multi-rank path on a one-GPU box (all ranks share GPU 0, gloo transport)
        if device_type == "cuda" and backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        elif device_type == "cuda":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world
