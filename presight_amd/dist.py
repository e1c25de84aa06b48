"""Data-parallel gradient exchange for one tile trained on N GPUs (one process per GPU, RCCL over xGMI through
torch.distributed's "nccl" backend; "gloo" on CPU for tests).

The reference wraps the model in DDP(find_unused_parameters=True) (ns/pipelines/PreSight/my_pipeline.py:121-124): every
parameter gradient is averaged over ranks each step.  Here all gradients live in ONE flat fp32 buffer (the parameters'
.grad tensors are views into it), so the exchange needs no packing copies:
 * xGMI is point-to-point, so few, large messages are what keeps all 7 links busy;
 * "unused parameter" handling is free: a sub-field that saw no sample this step just contributes zeros.

Two exchange modes (FlatGrads.enable_overlap(..., mode=)):
 * "allreduce" — every bucket is averaged with one all-reduce; every rank then runs Adam over everything (cfg 2: 134 MB);
 * "sharded"   — every bucket is averaged with a REDUCE-SCATTER (rank r receives the r-th of `world` equal shards), Adam runs
   on the owned shard only (1/world of the optimizer traffic and state: the production tile has 940 M parameters = 26 GB of
   Adam traffic per step) and the updated parameters return with an ALL-GATHER that is left in flight on the side stream:
   the next step's ray generation and proposal sampling run underneath it, `wait_params(bucket)` is called right before the
   first kernel that reads the bucket.  Same bytes on the links as a ring all-reduce.

Collectives are always ISSUED IN THE SAME ORDER on every rank (NCCL / gloo pair collectives by issue order): buckets go out
strictly in the order they were passed to enable_overlap, whether a bucket is launched from a gradient hook (complete
during backward) or from finish_exchange (a sub-field without samples on this rank never completes it)."""
from __future__ import annotations

import os

from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


class CommLog:
    """Per-rank log of every collective this module issues, in ISSUE order: sequence number, kind, bucket, bytes, stream.  NCCL /
    gloo pair collectives by issue order, so a hang is diagnosed by diffing the ranks' logs: the first line that differs (or the
    last line one rank has and the other lacks) is the mismatched collective.  PRESIGHT_COMM_LOG=<path with {rank}> turns it on
    (one flushed line per collective: what is on disk when a rank hangs is what it issued); the last 256 entries are always kept
    in memory (CommLog.tail())."""

    def __init__(self):
        import collections
        import os

        self.seq = 0
        self.ring = collections.deque(maxlen=256)
        self.file = None
        path = os.environ.get("PRESIGHT_COMM_LOG")
        if path:
            self.file = open(path.replace("{rank}", os.environ.get("RANK", "0")), "a", buffering=1)

    def issue(self, kind: str, bucket, nbytes: int, stream: str, step=None, phase: Optional[str] = None):
        """phase: "backward" = issued from a gradient hook while the backward pass is still being enqueued, "finish" = issued from
        finish_exchange after it (bucketed gradient collectives only)"""
        self.seq += 1
        line = f"{self.seq} {kind} bucket={bucket} bytes={nbytes} stream={stream} step={step}" + (f" phase={phase}" if phase else "")
        self.ring.append(line)
        if self.file is not None:
            self.file.write(line + "\n")

    def tail(self, n: int = 16):
        return list(self.ring)[-n:]


COMM_LOG = CommLog()


def _mark_touched(p):
    """autograd's own accumulation finished for p (post-accumulate-grad hook); the HIP backward nodes that write gradients
    in place call presight_amd.ops.mark_touched, which ends up here too"""
    if p._ps_touched:
        return  # autograd also runs this hook for a parameter whose backward node wrote the gradient in place and returned None
    cb = getattr(p, "_ps_on_touch", None)
    if cb is not None:
        cb(p)
    p._ps_touched = True


def _merge(ranges: Sequence[Tuple[int, int]]) -> List[Tuple[int, int]]:
    out: List[List[int]] = []
    for a, b in sorted(ranges):
        if b <= a:
            continue
        if out and out[-1][1] >= a:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return [(a, b) for a, b in out]


def intersect_ranges(xs: Sequence[Tuple[int, int]], ys: Sequence[Tuple[int, int]]) -> List[Tuple[int, int]]:
    """intersection of two sorted lists of disjoint half-open ranges"""
    out, i, j = [], 0, 0
    while i < len(xs) and j < len(ys):
        a, b = max(xs[i][0], ys[j][0]), min(xs[i][1], ys[j][1])
        if a < b:
            out.append((a, b))
        if xs[i][1] < ys[j][1]:
            i += 1
        else:
            j += 1
    return out


def exchange_in_world_of_one() -> bool:
    """PRESIGHT_EXCHANGE_WORLD_OF_ONE=1: a process group of ONE rank counts as a distributed run -- every bucket, flag and parameter
    collective is issued in it (sum over one rank / 1: the identity, bit for bit).  This is how the exchange code runs through RCCL on a
    one-GPU box (tests/test_hip_dist.py::test_rccl_group_of_one_runs_the_exchange); two ranks cannot share a GPU under RCCL."""
    return os.environ.get("PRESIGHT_EXCHANGE_WORLD_OF_ONE") == "1"


def exchanging(group: Optional["dist.ProcessGroup"] = None) -> bool:
    """a process group is up and gradients travel through it"""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or exchange_in_world_of_one())


def _zero_ranges(flat: torch.Tensor, ranges) -> None:
    """flat[a:b] = 0 for every (a, b): several small ranges on the GPU (the MLP gradients between the hash tables of a single-process
    step) in one launch (ps_zero_ranges); long ranges (a table's gradient under an exchange) keep torch's fill, one each"""
    small = [(a, b) for a, b in ranges if b > a and (b - a) <= (1 << 22)]
    for a, b in ranges:
        if b > a and (b - a) > (1 << 22):
            flat[a:b].zero_()
    if len(small) == 1 or (small and not flat.is_cuda):
        for a, b in small:
            flat[a:b].zero_()
    elif small:
        import ctypes

        from ._lib import check, lib

        for i in range(0, len(small), 16):
            part = small[i:i + 16]
            check(lib().ps_zero_ranges(flat.data_ptr(), len(part), (ctypes.c_int64 * len(part))(*[a for a, _ in part]),
                                       (ctypes.c_int64 * len(part))(*[b - a for a, b in part]),
                                       torch._C._cuda_getCurrentRawStream(flat.device.index)), "ps_zero_ranges")


class FlatGrads:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_sizes: Optional[Sequence[int]] = None, shard_world: int = 1,
                 splits: Optional[Dict[int, int]] = None):
        """bucket_sizes: number of consecutive parameters per exchange bucket (bucket-major parameter order); with
        shard_world = W every bucket's range is padded to a multiple of 4*W floats so that it splits into W equal,
        16-byte-aligned shards (mode "sharded").
        splits: {bucket index: G} -- a ONE-parameter bucket (a hash table) is exchanged as G equal consecutive pieces, each its own
        collective, handed over by the producer piece by piece (part_done): the table backward accumulates level group after level
        group, so the reduce-scatter of group g runs underneath the accumulate launch of group g + 1.  G is lowered until the pieces
        are equal and shard-aligned."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad and p.numel() > 0]
        pad = lambda n: (n + 3) // 4 * 4  # noqa: E731  keep every view 16-byte aligned (vectorised optimizer kernels)
        if bucket_sizes is None:
            bucket_sizes = [len(self.params)]
        if sum(bucket_sizes) != len(self.params):
            raise ValueError("FlatGrads: bucket_sizes must cover the parameters exactly")
        align = 4 * max(1, int(shard_world))
        self.offsets: List[int] = []
        self.bucket_ranges: List[Tuple[int, int]] = []
        self.bucket_params: List[Tuple[int, int]] = []  # (first parameter index, one past the last)
        self.bucket_parts: Dict[int, List[Tuple[int, int]]] = {}  # declared bucket -> ranges of its pieces (split buckets only)
        off, i = 0, 0
        for bi, nb in enumerate(bucket_sizes):
            start = off
            for p in self.params[i:i + nb]:
                self.offsets.append(off)
                off += pad(p.numel())
            off = (off + align - 1) // align * align
            self.bucket_ranges.append((start, off))
            self.bucket_params.append((i, i + nb))
            G = int((splits or {}).get(bi, 1))
            if G > 1:
                if nb != 1:
                    raise ValueError("FlatGrads: only a one-parameter bucket can be split into pieces")
                n = off - start
                while G > 1 and (n % G or (n // G) % align):
                    G -= 1
                if G > 1:
                    self.bucket_parts[bi] = [(start + g * (n // G), start + (g + 1) * (n // G)) for g in range(G)]
            i += nb
        self.total = off
        dev = self.params[0].device
        self.flat = torch.zeros(self.total, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            p.grad = self.flat[o:o + n].view_as(p)
            # opt in to in-place gradient accumulation by the HIP backward kernels (presight_amd.ops.grad_sink): the buffer
            # is zeroed once per step (zero_()), so "+=" from any number of uses of the parameter is the full gradient
            p._ps_direct_grad = True
            p._ps_touched = False
            p.register_post_accumulate_grad_hook(_mark_touched)  # gradients that arrive through autograd's own accumulation
        self._pad = pad
        self.shard_world = max(1, int(shard_world))
        self.flags_may_differ_across_ranks = False  # set when routing can leave a sub-field without samples on one rank only
        self._buckets: List[dict] = []
        self._next_launch = 0
        self._group = None
        self._comm_stream = None
        self.mode = "allreduce"
        self._dirty: Optional[List[Tuple[int, int]]] = None  # ranges that may be non-zero (None = unknown -> everything)
        self._param_events: Dict[int, object] = {}
        self.stats = {"collectives": 0, "bytes": 0}
        # device-decided "received a gradient this step" groups (define_groups): routed sub-fields
        self.group_flags: Optional[Tensor] = None   # int32 [n_groups], 1 = the group's sub-field received samples this step
        self.group_steps: Optional[Tensor] = None   # int32 [n_groups], torch.optim.Adam's state["step"] of the group
        self.n_groups = 0
        self.step_no = 0  # zero_() calls so far (only labels the lines of COMM_LOG)
        self._in_finish = False
        self.dry = False            # bucket bookkeeping and timeline without a process group (enable_overlap(dry=True))
        self.record_timeline = False  # record, per step, when every bucket became ready relative to the end of backward
        self._timeline: List[dict] = []

    def define_groups(self, groups: Sequence[Sequence[torch.nn.Parameter]]):
        """Parameters whose "received a gradient this step" is only known ON THE DEVICE: the sub-fields of a routed tile (the
        router never synchronises with the host, so the host cannot know which sub-fields got samples).  groups[i] = the
        parameters of one sub-field of one routed module; they receive gradients together.  The routed backward nodes raise
        group_flags[i] on the device (ps_ms_mark_groups), the optimizer kernel skips a group whose flag is down (parameters,
        moments and step count untouched, like torch.optim.Adam with grad None: the reference never calls an empty sub-field,
        ns/fields/PreSight/ingp_field_ms.py:97-126) and keeps the groups' step counts on the device.  Under data parallelism
        the flags are MAX-reduced over the ranks in finish_exchange (DDP: used on ANY rank -> averaged gradient everywhere),
        which also makes the host-side flags rank-independent (no flag all-reduce, no host sync)."""
        index = {id(p): i for i, p in enumerate(self.params)}
        n = 0
        mine = set()
        for plist in groups:
            plist = [p for p in plist if id(p) in index]
            if not plist:
                continue
            for p in plist:
                if id(p) in mine:
                    raise ValueError("FlatGrads.define_groups: a parameter belongs to two groups")
                mine.add(id(p))
                # a NEW owner takes the parameter over (a second Trainer / FlatGrads on the same model: exchange-mode comparisons,
                # rebuilding the trainer after a checkpoint load); the previous owner's flags are no longer raised for it, so the
                # previous owner is marked: its zero_() raises instead of stepping a model whose routed sub-fields it would skip
                prev = getattr(p, "_ps_group_owner", None)
                if prev is not None and prev is not self:
                    prev._groups_taken_over = True
                p._ps_group = n
                p._ps_group_owner = self
            n += 1
        from . import field_ops  # (late: field_ops does not import this module)

        field_ops._GROUP_TABLES.clear()  # mark_groups' cached group-id tables are keyed by the owner: drop the previous owner's
        self.n_groups = n
        dev = self.flat.device
        self._group_flag_bufs = [torch.zeros(max(n, 1), device=dev, dtype=torch.int32) for _ in range(2)]
        self.group_flags = self._group_flag_bufs[0]
        self.group_steps = torch.zeros(max(n, 1), device=dev, dtype=torch.int32)
        self.flags_may_differ_across_ranks = False

    # ------------------------------------------------------------------ per-step bookkeeping
    def zero_(self, already_zeroed: Optional[Sequence[Tuple[int, int]]] = None):
        """Start of a step: gradients to zero, "received a gradient this step" flags cleared (torch's zero_grad(set_to_none=True)
        + "grad is None -> the optimizer skips the parameter" semantics, without freeing the flat buffer).  Only the ranges
        that received a gradient in the previous step (as agreed across ranks: touched_ranges) are written; everything else
        is still zero (a production tile's 3.5 GiB buffer is mostly untouched when a sub-field gets no samples).
        already_zeroed: ranges somebody else has cleared (or will have cleared, in stream order, before anything writes them):
        the pipelined optimizer step clears the fields' gradients on its own stream right behind their Adam update."""
        if getattr(self, "_groups_taken_over", False):
            raise RuntimeError("presight_amd.dist.FlatGrads: another FlatGrads (a newer Trainer on the same model) has taken over this "
                               "one's device-decided parameter groups (define_groups); this one can no longer train the routed sub-fields")
        if self._dirty is None:
            self.flat.zero_()
        else:
            todo = self._dirty
            if already_zeroed:
                keep, cur = [], list(todo)
                for a, b in cur:  # subtract the sorted, disjoint ranges of already_zeroed
                    x = a
                    for c, d in already_zeroed:
                        if d <= x or c >= b:
                            continue
                        if c > x:
                            keep.append((x, c))
                        x = max(x, d)
                    if x < b:
                        keep.append((x, b))
                todo = keep
            _zero_ranges(self.flat, todo)
        self._dirty = None
        self.step_no += 1
        if self.n_groups:
            # two flag arrays, alternating: the previous step's flags may still be read by an optimizer piece running on another stream
            self.group_flags = self._group_flag_bufs[self.step_no & 1]
            self.group_flags.zero_()
        for p in self.params:
            p._ps_touched = False
            p._ps_fused_done = False  # (HipAdam.enable_fused_tables: "updated inside the table backward this step")
        for b in self._buckets:
            b["seen"], b["launched"], b["work"], b["ready"], b["ready_t"], b["phase"], b["skipped"] = 0, False, None, None, None, None, False
            b["pending"] = None
        self._next_launch = 0

    def touched(self, group: Optional[dist.ProcessGroup] = None) -> List[bool]:
        flags = [bool(p._ps_touched) for p in self.params]
        if self.flags_may_differ_across_ranks and exchanging(group):
            t = torch.tensor(flags, dtype=torch.int32, device=self.flat.device)  # DDP: a parameter used on ANY rank gets a gradient
            COMM_LOG.issue("all_reduce_max_host_flags", "-", 4 * len(flags), "current", self.step_no)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            flags = [bool(v) for v in t.tolist()]
        return flags

    def touched_params(self, group: Optional[dist.ProcessGroup] = None) -> List[int]:
        """indices of the parameters that received a gradient this step (agreed across ranks when routing may differ); also
        records their ranges as the only part of the buffer the next zero_() has to clear"""
        flags = self.touched(group)
        idx = [i for i, t in enumerate(flags) if t]
        # (a table whose Adam update ran inside its backward never had its gradient written: its range still holds zeros)
        rngs = [(self.offsets[i], self.offsets[i] + self._pad(self.params[i].numel())) for i in idx
                if not getattr(self.params[i], "_ps_fused_done", False)]
        if any(b.get("sparse") for b in self._buckets) and self._distributed():
            # a record-exchanged table's gradient exists on its owner only (the owned shard of the bucket); the rest of its range was never written
            rank, world = self._rank_world()
            own = []
            for b in self._buckets:
                a, e = b["range"]
                n = (e - a) // world
                own.append((a + rank * n, a + (rank + 1) * n) if b.get("sparse") else (a, e))
            rngs = intersect_ranges(_merge(rngs), _merge(own))
        self._dirty = _merge(rngs)
        return idx

    def touched_ranges(self, group: Optional[dist.ProcessGroup] = None) -> List[tuple]:
        """[(first float, one-past-last float)] of the flat buffer covering exactly the parameters that received a gradient
        this step, adjacent parameters merged (all touched -> one range per bucket-padding-free run)."""
        self.touched_params(group)
        return list(self._dirty)

    # ------------------------------------------------------------------ overlapped, bucketed exchange
    def enable_overlap(self, buckets: List[List[torch.nn.Parameter]], group: Optional[dist.ProcessGroup] = None,
                       mode: str = "allreduce", dry: bool = False, sparse: Sequence[int] = ()):
        """Exchange the gradient buffer in buckets, each as soon as it is complete, on a side stream, while the backward of the
        remaining parameters is still running (the reference's DDP overlaps its bucketed all-reduce with backward the same
        way).  A bucket = parameters that are contiguous in the flat buffer (e.g. one optimizer group), or one of the G pieces
        of a split one-parameter bucket (`splits` of the constructor); it becomes READY when every one of its parameters has
        received its gradient of this step / when its producer reports the piece (part_done).  Buckets are launched strictly in
        the order of `buckets` (pass them in the order backward completes them): a ready bucket waits for its predecessors, and
        whatever has not been launched by the end of backward goes out in finish_exchange() -- in the same order on every rank,
        which is what keeps the collectives of ranks whose routing left different sub-fields without samples paired correctly.
        A parameter must receive at most ONE gradient contribution per step (checked: a second one after the launch raises).
        dry: no process group is involved -- the bookkeeping, the stream events and the timeline run as they would (one GPU:
        measures when every bucket becomes ready relative to the end of backward, the input of the scaling model)."""
        if mode not in ("allreduce", "sharded"):
            raise ValueError(mode)
        if sparse and mode != "sharded":
            raise ValueError("FlatGrads: record (sparse) buckets belong to the sharded exchange (their owner updates its shard)")
        index = {id(p): i for i, p in enumerate(self.params)}
        self._buckets = []
        sparse = set(sparse)  # positions in `buckets` of the hash-table buckets exchanged as RECORDS (sparse_records / _sparse_exchange)
        for pos, plist in enumerate(buckets):
            ids = sorted({index[id(p)] for p in plist if id(p) in index})
            if not ids:
                continue
            if ids != list(range(ids[0], ids[-1] + 1)):
                raise ValueError("FlatGrads.enable_overlap: the parameters of a bucket must be contiguous in the flat buffer")
            a, b = self.offsets[ids[0]], self.offsets[ids[-1]] + self._pad(self.params[ids[-1]].numel())
            declared = [bi for bi, pr in enumerate(self.bucket_params) if pr == (ids[0], ids[-1] + 1)]
            if mode == "sharded":
                if not declared or (self.bucket_ranges[declared[0]][1] - self.bucket_ranges[declared[0]][0]) % (4 * self.shard_world):
                    raise ValueError("FlatGrads: sharded exchange needs the buckets declared at construction (bucket_sizes=, "
                                     "shard_world=) so that their ranges split into equal aligned shards")
                a, b = self.bucket_ranges[declared[0]]
            parts = self.bucket_parts.get(declared[0]) if declared else None
            if parts:
                p = self.params[ids[0]]
                p._ps_parts = len(parts)
                p._ps_part_done = self._on_part
                p._ps_part_buckets = []
                for g, rng in enumerate(parts):
                    self._buckets.append(dict(range=rng, n=1, seen=0, launched=False, work=None, index=len(self._buckets), part=(g, len(parts))))
                    p._ps_part_buckets.append(len(self._buckets) - 1)
                p._ps_bucket = p._ps_part_buckets[0]
                p._ps_on_touch = self._on_touch
                continue
            self._buckets.append(dict(range=(a, b), n=len(ids), seen=0, launched=False, work=None, index=len(self._buckets), part=None,
                                      sparse=pos in sparse, pending=None))
            if pos in sparse and parts:
                raise ValueError("FlatGrads: a record (sparse) bucket is exchanged in one piece")
            for i in ids:
                self.params[i]._ps_bucket = len(self._buckets) - 1
                self.params[i]._ps_on_touch = self._on_touch
                if pos in sparse:
                    self.params[i]._ps_sparse = (self, len(self._buckets) - 1)  # field_ops: keep the records, hand them to sparse_records
        self._group = group
        self.mode = mode
        self.dry = bool(dry)
        self._next_launch = 0
        if mode == "sharded" and self._distributed() and dist.get_world_size(group) != self.shard_world:
            raise ValueError(f"FlatGrads: built for shard_world={self.shard_world}, process group has {dist.get_world_size(group)} ranks")
        if self.flat.is_cuda:
            self._comm_stream = torch.cuda.Stream(device=self.flat.device)

    def _distributed(self) -> bool:
        return exchanging(self._group)

    def _mark_ready(self, b):
        """the bucket's gradients are complete on the streams as they stand NOW (the proposal networks' backward runs on side
        streams, ops.side_stream); the launch may happen later, from another stream's context"""
        if self._comm_stream is not None:
            from .ops import side_streams

            b["ready"] = []
            for st in [torch.cuda.current_stream(self.flat.device)] + side_streams(self.flat.device):
                ev = torch.cuda.Event(enable_timing=self.record_timeline)
                ev.record(st)
                b["ready"].append(ev)

    def _on_touch(self, p):
        parts = getattr(p, "_ps_part_buckets", None)
        if parts is not None:
            # a split parameter: its pieces are reported one by one (part_done); a whole-parameter report (a producer that does not
            # work in pieces, or the closing mark of one that does) completes whatever piece is still open
            for bi in parts:
                b = self._buckets[bi]
                if b["seen"] < b["n"]:
                    if b["launched"]:
                        raise RuntimeError("FlatGrads: a piece of a split parameter was exchanged before its gradient was complete")
                    b["seen"] = b["n"]
                    self._mark_ready(b)
            self._launch_ready()
            return
        b = self._buckets[p._ps_bucket]
        if b["launched"]:
            i = next(k for k, q in enumerate(self.params) if q is p)
            raise RuntimeError(f"FlatGrads: parameter #{i} {tuple(p.shape)} received a second gradient contribution after its bucket "
                               "had been handed to the all-reduce; build the trainer without enable_overlap for this model")
        if not p._ps_touched:
            b["seen"] += 1
            if b["seen"] == b["n"]:
                self._mark_ready(b)
                self._launch_ready()

    def _on_part(self, p, g: int):
        """piece g of the split parameter p holds its complete gradient (everything that produces it is enqueued on the current
        stream): called by the producer (field_ops._scatter) right after the launch that wrote the piece"""
        b = self._buckets[p._ps_part_buckets[g]]
        if b["launched"] or b["seen"]:
            raise RuntimeError(f"FlatGrads: piece {g} of a split parameter {tuple(p.shape)} was reported twice in one step")
        b["seen"] = 1
        self._mark_ready(b)
        self._launch_ready()

    def skip_buckets(self, indices: Sequence[int]):
        """These buckets receive NO gradient in this step on ANY rank, and every rank knows it before backward starts (the proposal
        networks on an off-schedule step: the update schedule is a function of the step count, ns/model_components/ray_samplers.py:586).
        They are taken out of the launch order at once, so that the buckets behind them do not wait for the end of backward; a
        gradient that reaches one of them after all raises (second-contribution guard).  Call after zero_()."""
        for i in indices:
            b = self._buckets[i]
            b["launched"], b["skipped"] = True, True
        self._launch_ready()

    def _launch_ready(self):
        """launch, in bucket order, every bucket whose predecessors have gone out and whose gradients are complete"""
        while self._next_launch < len(self._buckets):
            b = self._buckets[self._next_launch]
            if b.get("skipped"):
                self._next_launch += 1
                continue
            if b["seen"] < b["n"]:
                return
            self._launch(b)
            self._next_launch += 1

    def _rank_world(self):
        return dist.get_rank(self._group), dist.get_world_size(self._group)

    def _launch(self, b):
        b["launched"] = True
        b["phase"] = "finish" if self._in_finish else "backward"
        if not self._distributed():
            return
        rank, world = self._rank_world()
        a, e = b["range"]
        seg = self.flat[a:e]
        if b.get("sparse"):
            rec = b["pending"]
            if rec is None:
                raise RuntimeError(f"FlatGrads: record bucket {b['index']} is due for its exchange but this rank's table backward produced no "
                                   "records this step (every rank must run the bucket's table backward in every step it is exchanged)")

            def run():
                self._sparse_exchange(b, rec, rank, world)
                b["pending"] = None
                self.stats["in_backward"] = self.stats.get("in_backward", 0) + (b["phase"] == "backward")

            if self._comm_stream is not None:
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(self._comm_stream):
                    self._comm_stream.wait_event(ev)
                    for rev in b.get("ready") or ():
                        self._comm_stream.wait_event(rev)
                    run()
                    b["post"] = torch.cuda.Event()
                    b["post"].record()
            else:
                run()
            return

        def issue():
            # The MEAN over the ranks (DDP's semantics).  Dividing the whole bucket before the collective is a read + write of every
            # gradient byte on every rank (the production tile: 7.5 GB = 1.3 ms per step, most of what the exchange machinery cost a
            # rank before any byte moved, profiles/r05_rccl_group_of_one_*).  Sharded mode divides the OWNED shard after the
            # reduce-scatter instead (1 / world of the bytes; sum then scale -- for world a power of two the same bits as scale then
            # sum); a group of one rank divides by 1: nothing to do.
            if world > 1 and self.mode != "sharded":
                seg.div_(world)
            COMM_LOG.issue("reduce_scatter" if self.mode == "sharded" else "all_reduce", b["index"], 4 * (e - a),
                           "side" if self._comm_stream is not None else "current", self.step_no, phase=b["phase"])
            if self.mode == "sharded":
                n = (e - a) // world
                own = seg[rank * n:(rank + 1) * n]
                b["work"] = dist.reduce_scatter_tensor(own, seg, op=dist.ReduceOp.SUM, group=self._group, async_op=True)
                if world > 1:
                    b["work"].wait()  # (stream-ordered for RCCL: the division is enqueued behind the collective, the host does not block)
                    b["work"] = None  # (waited for: gloo copies a work's result into place in EVERY wait(), a second one would undo the division)
                    own.div_(world)
                    if self._comm_stream is not None:
                        b["post"] = torch.cuda.Event()
                        b["post"].record()
                self.stats["bytes"] += 4 * (e - a) * (world - 1) // world
            else:
                b["work"] = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self._group, async_op=True)
                self.stats["bytes"] += 2 * 4 * (e - a) * (world - 1) // world
            self.stats["collectives"] += 1
            self.stats["in_backward"] = self.stats.get("in_backward", 0) + (b["phase"] == "backward")

        if self._comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record()  # everything enqueued so far on the current stream (incomplete buckets go out after backward, from the compute stream)
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                for rev in b.get("ready") or ():
                    self._comm_stream.wait_event(rev)  # ... and on the streams that produced the bucket's gradients
                issue()
        else:
            issue()

    # ------------------------------------------------------------------ record (sparse) exchange of hash-table gradients
    def sparse_records(self, bucket: int, rec: dict):
        """field_ops hands over the record streams of this step's table backward for a record bucket (the binned backward's phase 1 is
        done; phase 2 -- the accumulate pass -- is replaced by the exchange).  rec: ws (the backward's workspace, uint8, not reused before
        the exchange ran), layout (ps_grid_scatter_layout), L, F, log2T, K, n_points, accumulate(run_starts, run_counts, n_runs, rec_idx,
        rec_val, plane_stride, gmax_bits, n_points_total, out_scale, item_begin, item_end).  The exchange itself runs when the bucket's turn
        in the launch order comes (mark_touched of its tables)."""
        b = self._buckets[bucket]
        if b["pending"] is not None or b["launched"]:
            raise RuntimeError(f"FlatGrads: record bucket {bucket} received a second set of records in one step")
        b["pending"] = rec

    def _a2a(self, out: Tensor, inp: Tensor, out_splits=None, in_splits=None, what: str = "all_to_all"):
        """all_to_all_single in the bucket's process group; gloo has no device transport for it: staged through the host there"""
        # (the log is compared line by line across ranks: the record segments' sizes are data-dependent and are logged as -1)
        COMM_LOG.issue(what, "-", 4 * inp.numel() if in_splits is None else -1, "side" if self._comm_stream is not None else "current", self.step_no)
        self.stats["collectives"] += 1
        if inp.is_cuda and dist.get_backend(self._group) == "gloo":
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self._group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=self._group)

    def _sparse_exchange(self, b: dict, rec: dict, rank: int, world: int):
        """SURVEY.md 8e "sparse exchange of touched rows": instead of reduce-scattering the bucket's dense gradient (the production
        tile's main tables: 3.3 GB per rank and step), every rank sends the RECORDS of its table backward -- already sorted by the 128 KiB
        table slice that owns their rows -- to the slice's owner, and the owner accumulates all ranks' runs of a slice in one int64
        accumulator.  Exact and order-independent: fixed point with a per-level scale from the MAX over the ranks.  Ownership = the
        sharded exchange's (rank r owns the r-th 1 / world of the bucket's range), so the shard-Adam and the parameter all-gather
        behind it are unchanged.  One host synchronisation (the segment lengths all_to_all_single needs on the host)."""
        lay, F, L, K = rec["layout"], rec["F"], rec["L"], rec["K"]
        n_rec_max, n_items, ls = int(lay[6]), int(lay[7]), int(lay[8])
        a, e = b["range"]
        if n_items % world or (e - a) != n_items * (1 << ls) * F:
            raise ValueError(f"FlatGrads: record bucket {b['index']}: {n_items} slices of {(1 << ls) * F} values do not split into {world} equal "
                             f"shards of its range [{a}, {e})")
        per = n_items // world
        ws = rec["ws"]
        i32 = lambda off, n: ws[off:off + 4 * n].view(torch.int32)  # noqa: E731
        gmax, cursors, upper, starts = i32(int(lay[0]), K * L), i32(int(lay[1]), n_items), i32(int(lay[2]), n_items), i32(int(lay[3]), n_items)
        rec_idx = i32(int(lay[4]), n_rec_max)
        rec_val = ws[int(lay[5]):int(lay[5]) + 4 * (F + 1) * n_rec_max].view(torch.float32).view(F + 1, n_rec_max)
        dev = ws.device
        # (1) one scale per level for everybody: MAX of the per-level |d(feature)| maxima (non-negative float bit patterns order like
        #     ints, the NaN pattern is the largest and so survives) and of the point counts (the fixed-point headroom)
        meta = torch.cat([gmax, torch.tensor([int(rec["n_points"])], dtype=torch.int32, device=dev)])
        COMM_LOG.issue("all_reduce_max_levels", b["index"], 4 * meta.numel(), "side" if self._comm_stream is not None else "current", self.step_no,
                       phase=b.get("phase"))  # (the first collective of a record bucket's exchange: carries the bucket's hand-over phase)
        dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=self._group)
        self.stats["collectives"] += 1
        gmax_all = meta[:K * L].contiguous()
        # (2) where every item's run sits inside the segment of its owner, and how many records it really holds
        seg_start = starts[::per].to(torch.int64)                                        # [world]
        end_all = (starts[-1:].to(torch.int64) + ((upper[-1:].to(torch.int64) + 3) // 4) * 4)
        seg_end = torch.cat([seg_start[1:], end_all])
        send_len = seg_end - seg_start
        rel = (starts.to(torch.int64) - seg_start.repeat_interleave(per)).to(torch.int32)
        cnt = cursors - starts
        item_meta = torch.stack([rel.view(world, per), cnt.view(world, per)], 1).contiguous()   # [world, 2, per]: destination-major
        recv_meta = torch.empty_like(item_meta)
        self._a2a(recv_meta, item_meta, what="all_to_all_item_runs")
        recv_len = torch.empty_like(send_len)
        self._a2a(recv_len, send_len, what="all_to_all_segment_lengths")
        host = torch.cat([send_len, recv_len, meta[K * L:].to(torch.int64)]).tolist()          # the ONE host synchronisation
        send_l, recv_l, n_max = host[:world], host[world:2 * world], host[-1]
        n_total = n_max * world
        total = sum(recv_l)
        # (3) the record planes: row | type words, then the F value planes and the blend weight
        recv_idx = torch.empty(max(total, 4), dtype=torch.int32, device=dev)
        recv_val = torch.empty(F + 1, max(total, 4), dtype=torch.float32, device=dev)
        used = sum(send_l)
        self._a2a(recv_idx[:total], rec_idx[:used], recv_l, send_l, what="all_to_all_records")
        for f in range(F + 1):
            self._a2a(recv_val[f, :total], rec_val[f, :used], recv_l, send_l, what="all_to_all_records")
        self.stats["bytes"] += 4 * (F + 2) * (used - send_l[rank]) + 8 * per * (world - 1)
        self.stats["sparse_record_bytes"] = self.stats.get("sparse_record_bytes", 0) + 4 * (F + 2) * (used - send_l[rank])
        # (4) the owner's accumulate pass over world runs per owned item -> its shard of the MEAN gradient
        off = torch.tensor([sum(recv_l[:s_]) for s_ in range(world)], dtype=torch.int32, device=dev)
        run_starts = (recv_meta[:, 0, :] + off[:, None]).contiguous()
        run_counts = recv_meta[:, 1, :].contiguous()
        rec["accumulate"](run_starts, run_counts, world, recv_idx, recv_val, max(total, 4), gmax_all, n_total, 1.0 / world, rank * per, (rank + 1) * per)

    def _join_side_streams(self):
        """gradients written in place by nodes that ran on the proposal networks' side stream (presight_amd.ops.side_stream) are
        invisible to the autograd engine's stream bookkeeping: the current stream waits for them here, before the buffer is read"""
        if self.flat.is_cuda:
            from .ops import join_side_streams

            join_side_streams()

    def finish_exchange(self):
        """after backward: launch the buckets that are still local (in order), then make the compute stream wait for all of them"""
        self._join_side_streams()
        if self.n_groups and exchanging(self._group):
            COMM_LOG.issue("all_reduce_max_flags", "-", 4 * self.group_flags.numel(), "current", self.step_no)
            dist.all_reduce(self.group_flags, op=dist.ReduceOp.MAX, group=self._group)  # device tensor, stream-ordered: no host sync
        if not self._buckets:
            return self.all_reduce_mean(self._group)
        end_ev = None
        if self.record_timeline and self.flat.is_cuda:
            end_ev = torch.cuda.Event(enable_timing=True)
            end_ev.record()  # the end of backward on the compute stream (the side streams have been joined)
        self._in_finish = True
        try:
            for b in self._buckets[self._next_launch:]:
                if b["seen"] == 0 and not self.flags_may_differ_across_ranks:
                    b["launched"] = True  # no parameter of the bucket got a gradient on any rank (schedule-driven): nothing to exchange
                else:
                    self._launch(b)
        finally:
            self._in_finish = False
        self._next_launch = len(self._buckets)
        for b in self._buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
            post = b.pop("post", None)
            if post is not None:  # (the owned shard's division on the communication stream)
                torch.cuda.current_stream(self.flat.device).wait_event(post)
        if end_ev is not None:
            self._timeline.append(dict(end=end_ev, buckets=[(b["index"], b["range"][1] - b["range"][0], b.get("ready"), b.get("phase"), b["seen"])
                                                           for b in self._buckets]))
        return None

    def timeline_summary(self) -> List[dict]:
        """per bucket, averaged over the recorded steps: floats, the phase it was handed over in, and `ms_before_backward_end` = how
        long before the end of backward its gradient was complete on every stream that produced it (0 for a bucket that was only
        complete at the end) -- the window its collective can hide in.  Synchronises."""
        if not self._timeline:
            return []
        torch.cuda.synchronize()
        acc: Dict[int, dict] = {}
        for rec in self._timeline:
            for idx, n, ready, phase, seen in rec["buckets"]:
                d = acc.setdefault(idx, dict(bucket=idx, floats=n, steps=0, ms=0.0, in_backward=0, exchanged=0))
                if seen == 0:
                    continue  # no gradient this step (proposal networks off schedule)
                d["exchanged"] += 1
                if ready:
                    d["ms"] += min(max(0.0, ev.elapsed_time(rec["end"])) for ev in ready)
                    d["steps"] += 1
                d["in_backward"] += int(phase == "backward")
        out = []
        for idx in sorted(acc):
            d = acc[idx]
            out.append(dict(bucket=idx, bytes=4 * d["floats"], steps_exchanged=d["exchanged"], handed_over_in_backward=d["in_backward"],
                            ms_before_backward_end=d["ms"] / d["steps"] if d["steps"] else 0.0))
        self._timeline = []
        return out

    # ------------------------------------------------------------------ sharded mode: who owns what, parameters back
    def owned_ranges(self) -> List[Tuple[int, int]]:
        """ranges of the flat buffers whose averaged gradient THIS rank holds after finish_exchange() and that its optimizer
        has to update (everything in "allreduce" mode or without a process group)"""
        if self.mode != "sharded" or not self._distributed():
            return [(0, self.total)]
        rank, world = self._rank_world()
        out = []
        for b in self._buckets:
            a, e = b["range"]
            n = (e - a) // world
            out.append((a + rank * n, a + (rank + 1) * n))
        return _merge(out)

    def gather_params(self, flat_params: Tensor, touched: Optional[Sequence[Tuple[int, int]]] = None):
        """sharded mode, after the optimizer step: all-gather the updated shards of every bucket that took part in this step's
        exchange into the (replicated) flat parameter buffer, asynchronously on the side stream; wait_params(i) orders the
        compute stream behind bucket i's gather."""
        self._param_events = {}
        if self.mode != "sharded" or not self._distributed():
            return
        rank, world = self._rank_world()
        ev = None
        if self._comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record()
        for b in self._buckets:
            a, e = b["range"]
            if touched is not None and not intersect_ranges([(a, e)], touched):
                continue  # nothing in this bucket was updated on any rank
            n = (e - a) // world
            seg = flat_params[a:e]
            COMM_LOG.issue("all_gather_params", b["index"], 4 * (e - a), "side" if self._comm_stream is not None else "current", self.step_no)
            if self._comm_stream is not None:
                with torch.cuda.stream(self._comm_stream):
                    self._comm_stream.wait_event(ev)
                    dist.all_gather_into_tensor(seg, seg[rank * n:(rank + 1) * n], group=self._group, async_op=True).wait()
                    done = torch.cuda.Event()
                    done.record()
                self._param_events[b["index"]] = done
            else:
                dist.all_gather_into_tensor(seg, seg[rank * n:(rank + 1) * n], group=self._group)
            self.stats["collectives"] += 1
            self.stats["bytes"] += 4 * (e - a) * (world - 1) // world

    def gather_flat(self, flat: Tensor):
        """sharded mode: all-gather the owned shards of `flat` (same layout as the gradient buffer: optimizer moments) so that
        every rank holds all of it; synchronous, a no-op otherwise.  A collective: every rank must call it."""
        if self.mode != "sharded" or not self._distributed():
            return
        self.wait_params()
        rank, world = self._rank_world()
        for b in self._buckets:
            a, e = b["range"]
            n = (e - a) // world
            seg = flat[a:e]
            COMM_LOG.issue("all_gather_state", b["index"], 4 * (e - a), "current", self.step_no)
            dist.all_gather_into_tensor(seg, seg[rank * n:(rank + 1) * n].clone(), group=self._group)

    def wait_params(self, bucket: Optional[int] = None):
        """compute stream waits (device side, no host sync) until the parameters of `bucket` (all buckets if None) are back"""
        keys = list(self._param_events) if bucket is None else [bucket]
        for k in keys:
            ev = self._param_events.pop(k, None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

    def all_reduce_mean(self, group: Optional[dist.ProcessGroup] = None, async_op: bool = False):
        """Average the flat gradient buffer over the ranks of `group` (no-op without an initialised process group)."""
        self._join_side_streams()
        if not exchanging(group):
            return None
        world = dist.get_world_size(group)
        self.flat.div_(world)
        COMM_LOG.issue("all_reduce_flat", "-", 4 * self.total, "current", self.step_no)
        self.stats["collectives"] += 1
        self.stats["bytes"] += 2 * 4 * self.total * (world - 1) // world
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def global_depth_clip(group: Optional[dist.ProcessGroup] = None):
    """hook for presight_amd.ops.set_depth_clip_hook: the expected-depth clip bounds become the min / max sample midpoint over
    the batches of ALL ranks (one 2-float all-reduce per render: MAX over {-min, max})"""
    def hook(minmax: Tensor):
        if not exchanging(group):
            return
        t = torch.stack([-minmax[0], minmax[1]])
        COMM_LOG.issue("all_reduce_max_depth_clip", "-", 8, "current")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        minmax[0] = -t[0]
        minmax[1] = t[1]

    return hook


def init_from_env(device_type: str = "cuda") -> tuple:
    """(rank, local_rank, world_size) from the torchrun environment; initialises the process group when world_size > 1."""
    import os

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or exchange_in_world_of_one()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("PRESIGHT_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        if backend == "gloo":
            # gloo connects its pairs lazily (inside the first collective) to the address it derives from the HOSTNAME unless
            # told which interface to use; the container hostname may not resolve (or resolve to an unroutable address), which
            # shows up as a first all-reduce that never returns.  Single-node runs always go through loopback.
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
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
