"""Samplers of the PreSight model on the HIP kernels (same names / arguments / call protocol as
ns/model_components/ray_samplers.py: SpacedSampler :53-128, PDFSampler :251-372, ProposalNetworkSampler :523-614).

The piecewise spacing of ns/models/PreSight/nerfacto_nusc_ms.py:312-317 is built into the kernels (a python callable
cannot run inside a HIP kernel): SpacedSampler keeps the reference's constructor (two spacing lambdas), recovers the threshold
from them by probing and REJECTS callables that are not that piecewise spacing."""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import ops
from .rays import RayBundle, RaySamples


class Sampler(nn.Module):
    def __init__(self, num_samples: Optional[int] = None) -> None:
        super().__init__()
        self.num_samples = num_samples

    def forward(self, *args, **kwargs):
        return self.generate_ray_samples(*args, **kwargs)


def _near_far(ray_bundle: RayBundle) -> Tuple[float, float]:
    """The collider of the PreSight model sets one near/far for the whole batch (scene_colliders.py:182-187); the
    kernels take them as scalars.  Stored as python floats by NearFarCollider to avoid a device sync."""
    nf = ray_bundle.metadata.get("_near_far")
    if nf is None:
        raise ValueError("ray bundle has no near/far: run the NearFarCollider first")
    return nf


def piecewise_threshold_of(spacing_fn: Callable, spacing_fn_inv: Optional[Callable] = None) -> float:
    """Recover `thr` from the pair of callables the reference hands to SpacedSampler
    (ns/models/PreSight/nerfacto_nusc_ms.py:311-316):  s(t) = t / (2 thr) for t < thr, 1 - thr / (2 t) otherwise, with inverse
    x < .5 ? 2 thr x : thr / (2 - 2x).  A python callable cannot run inside a HIP kernel, so the kernels implement exactly this
    family; the callables are probed on the host, and anything that is not a member of the family is REJECTED (never ignored)."""
    probe = torch.tensor([1e-4, 1e-3], dtype=torch.float64)
    s = torch.as_tensor(spacing_fn(probe), dtype=torch.float64)
    if not bool(torch.all(s > 0)):
        raise NotImplementedError("presight_amd SpacedSampler: spacing_fn is not the piecewise spacing of nerfacto_nusc_ms.py:312-317")
    thr = float((probe / (2.0 * s)).mean())
    t = torch.tensor([0.05, 0.3, 0.999, 1.0, 1.5, 4.0, 40.0, 1000.0], dtype=torch.float64) * thr
    want = torch.where(t < thr, t / (2 * thr), 1 - thr / (2 * t))
    got = torch.as_tensor(spacing_fn(t.to(torch.float32)), dtype=torch.float64)
    if not torch.allclose(got, want, rtol=1e-5, atol=1e-6):
        raise NotImplementedError(f"presight_amd SpacedSampler: spacing_fn is not t/(2 thr) | 1 - thr/(2 t) (thr probed as {thr:g}); "
                                  "only the piecewise spacing of nerfacto_nusc_ms.py:312-317 runs in the kernels")
    if spacing_fn_inv is not None:
        x = torch.tensor([0.01, 0.25, 0.499, 0.5, 0.75, 0.99], dtype=torch.float32)
        want_inv = torch.where(x < 0.5, x * (2 * thr), thr / (2 - 2 * x)).to(torch.float64)
        got_inv = torch.as_tensor(spacing_fn_inv(x), dtype=torch.float64)
        if not torch.allclose(got_inv, want_inv, rtol=1e-5, atol=1e-6):
            raise NotImplementedError("presight_amd SpacedSampler: spacing_fn_inv is not the inverse of the piecewise spacing")
    return float(round(thr, 6)) if abs(round(thr, 6) - thr) < 1e-7 * max(1.0, thr) else thr


class SpacedSampler(Sampler):
    """Piecewise spaced sampler: s(t) = t/(2 thr) for t < thr, 1 - thr/(2 t) otherwise.  Constructed like the reference
    (ns/model_components/ray_samplers.py:53-76: spacing_fn, spacing_fn_inv, num_samples, train_stratified, single_jitter) --
    the threshold is then recovered from the callables (piecewise_threshold_of) -- or directly with `piecewise_threshold`."""

    def __init__(self, spacing_fn: Optional[Callable] = None, spacing_fn_inv: Optional[Callable] = None, num_samples: Optional[int] = None,
                 train_stratified=True, single_jitter=False, piecewise_threshold: Optional[float] = None) -> None:
        super().__init__(num_samples=num_samples)
        if not single_jitter:
            raise NotImplementedError("presight_amd SpacedSampler: PreSight uses single_jitter=True")
        self.train_stratified = train_stratified
        self.single_jitter = single_jitter
        if spacing_fn is not None:
            thr = piecewise_threshold_of(spacing_fn, spacing_fn_inv)
            if piecewise_threshold is not None and abs(thr - float(piecewise_threshold)) > 1e-6 * max(1.0, thr):
                raise ValueError(f"SpacedSampler: spacing_fn has threshold {thr:g}, piecewise_threshold={piecewise_threshold:g}")
        elif spacing_fn_inv is not None:
            raise ValueError("SpacedSampler: spacing_fn_inv without spacing_fn")
        elif piecewise_threshold is None:
            raise ValueError("SpacedSampler: pass spacing_fn / spacing_fn_inv (reference signature) or piecewise_threshold")
        else:
            thr = float(piecewise_threshold)
        self.thr = thr
        self.spacing_fn, self.spacing_fn_inv = spacing_fn, spacing_fn_inv

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, num_samples: Optional[int] = None,
                             jitter: Optional[Tensor] = None, points_spec: Optional[tuple] = None) -> RaySamples:
        """points_spec = (aabb, contract, key) of the field that will be evaluated on these samples (fields.points_spec): its
        normalised points are formed by the same launch and travel with the RaySamples"""
        assert ray_bundle is not None
        num_samples = num_samples or self.num_samples
        assert num_samples is not None
        near, far = _near_far(ray_bundle)
        R, dev = ray_bundle.origins.shape[0], ray_bundle.origins.device
        if self.train_stratified and self.training:
            if jitter is None:
                jitter = torch.rand((R, 1), device=dev)
        else:
            jitter = None
        if points_spec is None:
            sb, eb = ops.spaced_bins(R, num_samples, near, far, self.thr, jitter, dev)
            return RaySamples(ray_bundle, eb, sb, spacing_to_euclidean_fn=(near, far, self.thr))
        aabb, contract, key = points_spec
        sb, eb, u, sel = ops.spaced_bins(R, num_samples, near, far, self.thr, jitter, dev, points=(ray_bundle.origins, ray_bundle.directions, aabb, contract))
        return RaySamples(ray_bundle, eb, sb, spacing_to_euclidean_fn=(near, far, self.thr), points=(key, u, sel))


class PDFSampler(Sampler):
    def __init__(self, num_samples: Optional[int] = None, train_stratified: bool = True, single_jitter: bool = False,
                 include_original: bool = True, histogram_padding: float = 0.01) -> None:
        super().__init__(num_samples=num_samples)
        if include_original or not single_jitter:
            raise NotImplementedError("presight_amd PDFSampler: PreSight uses include_original=False, single_jitter=True")
        self.train_stratified = train_stratified
        self.include_original = include_original
        self.histogram_padding = histogram_padding
        self.single_jitter = single_jitter

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, ray_samples: Optional[RaySamples] = None,
                             weights: Optional[Tensor] = None, num_samples: Optional[int] = None, eps: float = 1e-5,
                             anneal: float = 1.0, jitter: Optional[Tensor] = None) -> RaySamples:
        if ray_samples is None or ray_bundle is None:
            raise ValueError("ray_samples and ray_bundle must be provided")
        assert weights is not None, "weights must be provided"
        num_samples = num_samples or self.num_samples
        assert num_samples is not None
        assert ray_samples.spacing_to_euclidean_fn is not None, "ray_samples.spacing_to_euclidean_fn must be provided"
        near, far, thr = ray_samples.spacing_to_euclidean_fn
        R, dev = weights.shape[0], weights.device
        if self.train_stratified and self.training:
            if jitter is None:
                jitter = torch.rand((R, 1), device=dev)
        else:
            jitter = None
        w = weights.reshape(weights.shape[0], weights.shape[1]) if weights.dim() == 3 else weights
        nsb, neb = ops.pdf_resample(w, ray_samples.sbins, num_samples, jitter, anneal, near, far, thr,
                                    pad=self.histogram_padding, eps=eps)
        return RaySamples(ray_bundle, neb, nsb, spacing_to_euclidean_fn=ray_samples.spacing_to_euclidean_fn)


prop_stream = ops.side_stream  # the proposal networks' side stream (ops.side_stream: registry + join)


class JitterPool:
    """The stratified-sampling jitters of MANY steps from one torch.rand call.  The reference draws one U[0, 1) number per ray and
    sampling level in every iteration (ns/model_components/ray_samplers.py:105,322: three generator launches per step, each with
    its own fill / philox set-up); the draws are i.i.d., so a block of `steps` x `levels` x R numbers drawn at once and handed out
    slice by slice is the same distribution with one launch per `steps` iterations (<= 16 MB per block)."""

    def __init__(self, budget_bytes: int = 16 << 20):
        self.budget = budget_bytes
        self.buf: Optional[Tensor] = None
        self.next = 0
        self.rng_state = None

    @staticmethod
    def _rng_state(device):
        """(seed, philox offset) of the device's default generator -- host-side reads, no sync.  A block is only handed out while
        this is what the block's own draw left behind: torch.manual_seed(...) (or anybody else drawing from the generator) starts a
        new block, so that re-seeding reproduces a run's jitters exactly as it did with one torch.rand per step."""
        try:
            g = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
            return (g.initial_seed(), g.get_offset())
        except Exception:  # (an exotic generator without offset bookkeeping: never reuse across calls)
            return None

    def draw(self, levels: int, R: int, device) -> List[Tensor]:
        b = self.buf
        state = self._rng_state(device)
        if (b is None or b.shape[1] != levels or b.shape[2] != R or b.device != device or self.next >= b.shape[0] or state is None
                or state != self.rng_state):
            steps = max(1, min(256, self.budget // max(1, 4 * levels * R)))
            if b is None or tuple(b.shape) != (steps, levels, R) or b.device != device:
                self.buf = b = torch.empty((steps, levels, R), device=device)
            b.uniform_()  # U[0, 1) like torch.rand, IN PLACE: a fresh 16 MB block per refill cost the caching allocator a device malloc
            self.next = 0  # (+ its implicit sync: ~30 ms, once, in the middle of a timed region)
            self.rng_state = self._rng_state(device)
        i, self.next = self.next, self.next + 1
        return [b[i, l].view(R, 1) for l in range(levels)]


def arg_device(arg):
    return arg.device if torch.is_tensor(arg) else arg.ebins.device


def _tensors_of(obj):
    """the device tensors an object holds directly or inside a tuple attribute (RaySamples.points)"""
    for v in vars(obj).values():
        for t in (v if isinstance(v, tuple) else (v,)):
            if torch.is_tensor(t) and t.is_cuda:
                yield t


# weights + PDF resampling (+ next points) of a proposal level as one launch (PRESIGHT_FUSED_RESAMPLE=0: the three launches of rounds 1-5)
FUSED_RESAMPLE = os.environ.get("PRESIGHT_FUSED_RESAMPLE", "1") != "0"


class ProposalNetworkSampler(Sampler):
    def __init__(self, num_proposal_samples_per_ray: Tuple[int, ...] = (64,), num_nerf_samples_per_ray: int = 32,
                 num_proposal_network_iterations: int = 2, single_jitter: bool = False, update_sched: Callable = lambda x: 1,
                 initial_sampler: Optional[Sampler] = None) -> None:
        super().__init__()
        self.num_proposal_samples_per_ray = num_proposal_samples_per_ray
        self.num_nerf_samples_per_ray = num_nerf_samples_per_ray
        self.num_proposal_network_iterations = num_proposal_network_iterations
        self.update_sched = update_sched
        if self.num_proposal_network_iterations < 1:
            raise ValueError("num_proposal_network_iterations must be >= 1")
        if initial_sampler is None:
            raise NotImplementedError("presight_amd: pass the piecewise SpacedSampler as initial_sampler")
        self.initial_sampler = initial_sampler
        self.pdf_sampler = PDFSampler(include_original=False, single_jitter=single_jitter)
        self._anneal = 1.0
        self._steps_since_update = 0
        self._step = 0
        self._jitters = JitterPool()

    def set_anneal(self, anneal: float) -> None:
        self._anneal = anneal

    def step_cb(self, step):
        self._step = step
        self._steps_since_update += 1

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, density_fns: Optional[List[Callable]] = None,
                             jitters: Optional[List[Tensor]] = None, final_points_spec: Optional[tuple] = None) -> Tuple[RaySamples, List, List]:
        """ns/model_components/ray_samplers.py:572-614.  `density_fns[i]` receives a RaySamples when it has the
        attribute `takes_ray_samples` (fused path: positions are generated inside the field kernel), else positions.
        One launch per level (round 6): get_weights of level i, the PDF resampling of level i + 1's bins and the points of the field
        that will be evaluated on them (density_fns[i + 1].points_spec() / final_points_spec: fields.points_spec) run as ONE kernel
        (ops.weights_resample) -- same arithmetic, the PDFSampler module keeps serving direct calls."""
        assert ray_bundle is not None and density_fns is not None
        weights_list, ray_samples_list = [], []
        n = self.num_proposal_network_iterations
        weights, ray_samples = None, None
        updated = self._steps_since_update > self.update_sched(self._step) or self._step < 10
        eps = float(torch.finfo(torch.float32).eps)
        if jitters is None and self.training and getattr(self.initial_sampler, "train_stratified", True) and self.pdf_sampler.train_stratified:
            jitters = self._jitters.draw(n + 1, ray_bundle.origins.shape[0], ray_bundle.origins.device)  # one draw for many steps
        specs = [getattr(fn, "points_spec", lambda: None)() if getattr(fn, "takes_ray_samples", False) else None for fn in density_fns[:n]]
        specs.append(final_points_spec)
        pdf = self.pdf_sampler
        fuse_pdf = type(pdf) is PDFSampler and FUSED_RESAMPLE
        near, far, thr = None, None, None
        nxt = None  # (new sbins, new ebins, points | None) of the next level, formed together with this level's weights
        for i_level in range(n + 1):
            is_prop = i_level < n
            num_samples = self.num_proposal_samples_per_ray[i_level] if is_prop else self.num_nerf_samples_per_ray
            jit = None if jitters is None else jitters[i_level]
            if i_level == 0:
                ray_samples = self.initial_sampler(ray_bundle, num_samples=num_samples, jitter=jit, points_spec=specs[0])
            elif nxt is not None:
                ray_samples = RaySamples(ray_bundle, nxt[1], nxt[0], spacing_to_euclidean_fn=ray_samples.spacing_to_euclidean_fn, points=nxt[2])
            else:
                assert weights is not None
                ray_samples = self.pdf_sampler(ray_bundle, ray_samples, weights, num_samples=num_samples, eps=eps,
                                               anneal=self._anneal, jitter=jit)
            nxt = None
            if is_prop:
                fn = density_fns[i_level]
                arg = ray_samples if getattr(fn, "takes_ray_samples", False) else ray_samples.frustums.get_positions()
                n_next = self.num_proposal_samples_per_ray[i_level + 1] if i_level + 1 < n else self.num_nerf_samples_per_ray
                next_jit = None
                if fuse_pdf and pdf.train_stratified and pdf.training:
                    next_jit = jitters[i_level + 1] if jitters is not None else torch.rand((ray_samples.ebins.shape[0], 1), device=ray_samples.ebins.device)

                def weights_and_next(density):
                    """get_weights (+ the next level's bins and points from the same launch)"""
                    if not fuse_pdf:
                        return ray_samples.get_weights(density), None
                    near_, far_, thr_ = ray_samples.spacing_to_euclidean_fn
                    sp = specs[i_level + 1]
                    pts = None if sp is None else (ray_bundle.origins, ray_bundle.directions, sp[0], sp[1])
                    w, nsb, neb, u, sel = ops.weights_resample(ray_samples.ebins, density.reshape(density.shape[0], density.shape[1]), ray_samples.sbins,
                                                               n_next, next_jit, self._anneal, near_, far_, thr_, pad=pdf.histogram_padding, eps=eps,
                                                               points=pts)
                    return w[..., None], (nsb, neb, None if sp is None else (sp[2], u, sel))

                if updated and torch.is_grad_enabled():
                    # positions handed to a plain density_fn are allocated on the caller's stream and owned by nobody after this call
                    # (the caching allocator could hand their block out again while the side stream's backward still reads it):
                    # only the RaySamples path -- whose tensors the model keeps alive in its outputs until the streams are joined --
                    # goes to the side stream
                    side = prop_stream(arg_device(arg)) if getattr(fn, "takes_ray_samples", False) else None
                    if side is None:
                        density = fn(arg)
                        weights, nxt = weights_and_next(density)
                    else:
                        # The proposal network runs on a SIDE stream.  Its forward is ordered between the two waits (the sampling
                        # chain is sequential anyway); autograd runs a node's backward on the stream of its forward, so the
                        # proposal networks' backward -- which depends on the interlevel loss only, not on the main field's
                        # backward (the main weights enter that loss detached, ns/models/PreSight/nerfacto_nusc_ms.py:595-600)
                        # -- is enqueued BESIDE the main field's matrix-bound backward kernels instead of behind them: the
                        # record-writing bin kernel (68 registers) fits next to a 400-register MFMA wave on every SIMD.
                        # cfg 2: 14.77 -> 14.44 ms per step (same box, alternating runs).
                        cur = torch.cuda.current_stream()
                        side.wait_stream(cur)
                        with torch.cuda.stream(side):
                            for obj in (ray_samples, ray_samples.ray_bundle):  # allocated on the compute stream, read on the side stream
                                for tns in _tensors_of(obj):
                                    tns.record_stream(side)
                            if next_jit is not None:
                                next_jit.record_stream(side)
                            density = fn(arg)
                            weights, nxt = weights_and_next(density)
                        cur.wait_stream(side)
                        for tns in (density, weights) + (() if nxt is None else (nxt[0], nxt[1]) + (() if nxt[2] is None else nxt[2][1:])):
                            tns.record_stream(cur)
                else:
                    with torch.no_grad():
                        density = fn(arg)
                    weights, nxt = weights_and_next(density)
                weights_list.append(weights)
                ray_samples_list.append(ray_samples)
        if updated:
            self._steps_since_update = 0
        assert ray_samples is not None
        return ray_samples, weights_list, ray_samples_list
