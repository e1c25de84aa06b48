"""Python host side of the C ABI: thin wrappers (raw device pointers + sizes + current HIP stream)
and torch.autograd.Function glue for the operator-level kernels.  torch is used for device memory,
streams and autograd bookkeeping only; all arithmetic of the hot path runs in libpresight_hip.so."""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from ._lib import check, lib


def _stream() -> int:
    """raw handle of torch's current stream on the current device (every library launch goes there).  The raw-handle query is 10 x
    cheaper than building a torch.cuda.Stream object per call (~900 calls per step: 0.5 ms of host time on a routed tile)"""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


# ---- side stream of the proposal networks (samplers.ProposalNetworkSampler.generate_ray_samples) ----------------------------
# During training the proposal networks run on ONE side stream per device (PRESIGHT_PROP_STREAM=0: on the caller's stream), so
# that autograd enqueues their backward beside the main field's.  Who has to wait for it:
#   * gradients that leave a node as tensors reach their AccumulateGrad nodes through the autograd engine, which orders streams
#     itself (and syncs the caller's stream with every leaf stream at the end of backward);
#   * gradients written IN PLACE into an owner's sink (grad_sink: presight_amd.dist.FlatGrads) are invisible to the engine: the
#     owner joins the side streams before it reads the buffer (FlatGrads.finish_exchange / all_reduce_mean, Trainer.step).
_SIDE_STREAMS: dict = {}
SIDE_STREAM = __import__("os").environ.get("PRESIGHT_PROP_STREAM", "1") != "0"


def side_stream(device):
    """the side stream of `device` (None: disabled / not a GPU).  Default priority (a high-priority one measured 0.25 ms slower);
    ONE stream for all proposal levels (a stream per level measured 0.15 ms slower: the chains then fight each other for the
    compute units the main field's persistent kernels leave them)."""
    if not SIDE_STREAM or device is None or device.type != "cuda":
        return None
    key = (str(device), 0)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def side_streams(device=None) -> list:
    return [s for (dev, _), s in _SIDE_STREAMS.items() if device is None or dev == str(device)]


def join_side_streams() -> None:
    """the current stream of every device with a side stream waits (device side, no host sync) for what is enqueued there"""
    for s in _SIDE_STREAMS.values():
        torch.cuda.current_stream(s.device).wait_stream(s)


def _f32(t: Tensor, name: str = "tensor") -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"presight_amd: {name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
# hash grid (operator level)
# ------------------------------------------------------------------------------------------------
class _HashGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table, scalings, L, F, log2T):
        x = _f32(x, "positions")
        N = x.shape[0]
        out = torch.empty(N, L * F, device=x.device, dtype=torch.float32)
        check(lib().ps_hashgrid_fwd(_p(x), _p(table), _p(scalings), L, F, log2T, N, _p(out), _stream()), "ps_hashgrid_fwd")
        ctx.save_for_backward(x, scalings)
        ctx.meta = (L, F, log2T, table.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, scalings = ctx.saved_tensors
        L, F, log2T, tshape = ctx.meta
        dout = _f32(dout)
        dtable = torch.zeros(tshape, device=x.device, dtype=torch.float32)
        check(lib().ps_hashgrid_bwd(_p(x), _p(dout), _p(scalings), L, F, log2T, x.shape[0], _p(dtable), _stream()),
              "ps_hashgrid_bwd")
        return None, dtable, None, None, None, None


def hashgrid_encode(x: Tensor, table: Tensor, scalings: Tensor, L: int, F: int, log2T: int) -> Tensor:
    """x [N,3] -> [N, L*F]; differentiable w.r.t. table (positions carry no gradient: bins are detached in
    the reference, ns/model_components/ray_samplers.py:360)."""
    return _HashGrid.apply(x, table, scalings, L, F, log2T)


def hashgrid_indices(x: Tensor, scalings: Tensor, L: int, log2T: int) -> Tensor:
    x = _f32(x)
    idx = torch.empty(x.shape[0], L, 8, device=x.device, dtype=torch.int64)
    check(lib().ps_hashgrid_indices(_p(x), _p(scalings), L, log2T, x.shape[0], _p(idx), _stream()), "ps_hashgrid_indices")
    return idx


# ------------------------------------------------------------------------------------------------
# MLP (operator level)
# ------------------------------------------------------------------------------------------------
def linear_colmap(ks: int, in_dim: int) -> List[int]:
    """k-step t, lane group g supplies input column 4t+g."""
    return [(4 * t + g if 4 * t + g < in_dim else -1) for t in range(ks) for g in range(4)]


def chain_colmap(ks: int, in_dim: int) -> List[int]:
    """input comes straight from the previous layer's MFMA D registers: column 16*(t//4)+4g+t%4."""
    out = []
    for t in range(ks):
        for g in range(4):
            c = 16 * (t // 4) + 4 * g + (t % 4)
            out.append(c if c < in_dim else -1)
    return out


class MlpSpec:
    """Mirror of ps::MlpT (csrc/mlp_core.hpp): packed-parameter and packed-gradient layouts."""

    def __init__(self, dims: Sequence[int], first_colmap: Optional[List[int]] = None, ks0: Optional[int] = None):
        self.dims = list(dims)
        self.nl = len(dims) - 1
        hidden = dims[1]
        if self.nl not in (2, 3) or any(d != hidden for d in dims[1:-1]) or hidden % 16:
            raise NotImplementedError(f"presight_amd MLP: unsupported layer dims {dims} (2-3 linear layers, equal hidden "
                                      f"width that is a multiple of 16)")
        self.ks = [ks0 if ks0 is not None else (dims[0] + 3) // 4] + [hidden // 4] * (self.nl - 1)
        self.nb = [hidden // 16] * (self.nl - 1) + [(dims[-1] + 15) // 16]
        self.ib = [(k + 3) // 4 for k in self.ks]
        self.colmaps = [first_colmap if first_colmap is not None else linear_colmap(self.ks[0], dims[0])]
        self.colmaps += [chain_colmap(self.ks[i], dims[i]) for i in range(1, self.nl)]
        self.fw = [nb * 16 + nb * ib * 256 for nb, ib in zip(self.nb, self.ib)]  # k-steps stored in groups of 4 (one 16-B load/lane)
        self.wt = [ib * nb * 4 * 64 for ib, nb in zip(self.ib, self.nb)]
        self.g = [nb * ib * 256 + nb * 16 for nb, ib in zip(self.nb, self.ib)]
        self.fw_off = [sum(self.fw[:i]) for i in range(self.nl)]
        self.fw_total = sum(self.fw)
        self.wt_off = [self.fw_total + sum(self.wt[:i]) for i in range(self.nl)]
        self.packed = self.fw_total + sum(self.wt)
        self.g_off = [sum(self.g[:i]) for i in range(self.nl)]
        self.g_total = sum(self.g)
        self._dev_colmaps = {}

    def dev_colmaps(self, device) -> List[Tensor]:
        key = str(device)
        if key not in self._dev_colmaps:
            self._dev_colmaps[key] = [torch.tensor(c, dtype=torch.int32, device=device) for c in self.colmaps]
        return self._dev_colmaps[key]

    def pack_descs(self, layers: Sequence[Tuple[Tensor, Tensor]], packed: Tensor) -> list:
        """per-layer descriptors (W, b, out, in, colmap, KS, NB, fw_block ptr, wt_block ptr) for pack_layers()"""
        cms = self.dev_colmaps(packed.device)
        base = packed.data_ptr()
        return [(_f32(W), _f32(b), W.shape[0], W.shape[1], cms[i], self.ks[i], self.nb[i], base + 4 * self.fw_off[i],
                 base + 4 * self.wt_off[i]) for i, (W, b) in enumerate(layers)]

    def pack_into(self, layers: Sequence[Tuple[Tensor, Tensor]], packed: Tensor):
        """layers: [(W [out,in], b [out])] in torch layout -> packed fragment buffer (device), one launch."""
        pack_layers(self.pack_descs(layers, packed))

    def pack(self, layers, device) -> Tensor:
        packed = torch.empty(self.packed, device=device, dtype=torch.float32)
        self.pack_into(layers, packed)
        return packed

    def unpack_descs(self, gpart: Tensor, part_offset: int, shapes: Sequence[Tuple[int, int]]) -> list:
        cms = self.dev_colmaps(gpart.device)
        return [(gpart.data_ptr() + 4 * (part_offset + self.g_off[i]), o, n_in, cms[i], self.ks[i], self.nb[i])
                for i, (o, n_in) in enumerate(shapes)]

    def unpack_grads(self, gpart: Tensor, n_parts: int, part_stride: int, part_offset: int, shapes: Sequence[Tuple[int, int]],
                     sinks: Optional[Sequence[Optional[Tuple[Tensor, Tensor]]]] = None) -> List[Optional[Tuple[Tensor, Tensor]]]:
        """Sum per-workgroup partial blocks into torch-layout (dW, db) per layer (one launch)."""
        return unpack_layers(self.unpack_descs(gpart, part_offset, shapes), n_parts, part_stride, gpart.device, sinks)


def _ptr_array(vals):
    import ctypes

    return (ctypes.c_void_p * len(vals))(*[int(v) for v in vals])


def _int_array(vals):
    import ctypes

    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def pack_layers(descs: list):
    """ps_mlp_pack_layers over descriptors from MlpSpec.pack_descs (<= 8 layers per launch)."""
    for i in range(0, len(descs), 8):
        d = descs[i:i + 8]
        check(lib().ps_mlp_pack_layers(len(d), _ptr_array([_p(x[0]) for x in d]), _ptr_array([_p(x[1]) for x in d]),
                                       _int_array([x[2] for x in d]), _int_array([x[3] for x in d]),
                                       _ptr_array([_p(x[4]) for x in d]), _int_array([x[5] for x in d]),
                                       _int_array([x[6] for x in d]), _ptr_array([x[7] for x in d]), _ptr_array([x[8] for x in d]),
                                       _stream()), "ps_mlp_pack_layers")


def grad_sink(t: Tensor) -> Optional[Tensor]:
    """Destination for a parameter gradient that is accumulated IN PLACE by the backward kernels instead of being returned
    to autograd: the parameter's pre-allocated `.grad` when its owner opted in (presight_amd.dist.FlatGrads marks its
    parameters) — saves one zero-fill, one temporary and one `grad += tmp` launch per parameter and step."""
    g = getattr(t, "grad", None)
    if getattr(t, "_ps_direct_grad", False) and g is not None and g.is_contiguous() and g.dtype == torch.float32:
        return g
    return None


def direct_params(*tensors) -> list:
    """the tensors among `tensors` whose gradient the backward writes in place (see grad_sink)"""
    return [t for t in tensors if t is not None and grad_sink(t) is not None]


def mark_touched(params, groups_on_device: bool = False):
    """tell the owner of the flat gradient buffer that these parameters received a gradient this step.  Parameters of a
    device-decided group (FlatGrads.define_groups: routed sub-fields) additionally need their group flag raised on the device:
    the routed backward nodes do that themselves from the layout (groups_on_device=True, field_ops.mark_groups); any other
    node that reaches such a parameter (a sub-field called directly) raises it here."""
    raised = set()
    for t in params:
        cb = getattr(t, "_ps_on_touch", None)  # bucketed gradient exchange: FlatGrads.enable_overlap
        if cb is not None:
            cb(t)
        t._ps_touched = True
        gid = getattr(t, "_ps_group", None)
        if gid is not None and not groups_on_device and (id(t._ps_group_owner), gid) not in raised:
            raised.add((id(t._ps_group_owner), gid))
            t._ps_group_owner.group_flags[gid:gid + 1].fill_(1)


def unpack_layers(descs: list, n_parts: int, part_stride: int, device, sinks=None) -> list:
    """ps_mlp_unpack_grad_layers over descriptors from MlpSpec.unpack_descs.  sinks[i] = (dW, db) buffers to accumulate
    into (-> returns None for that layer) or None (-> fresh tensors are returned)."""
    n = len(descs)
    sinks = list(sinks) if sinks is not None else [None] * n
    need = [i for i in range(n) if sinks[i] is None]
    out: list = [None] * n
    if need:
        sizes = [descs[i][1] * descs[i][2] + descs[i][1] for i in need]
        buf = torch.zeros(sum((s + 3) // 4 * 4 for s in sizes), device=device, dtype=torch.float32)
        off = 0
        for i, sz in zip(need, sizes):
            o, n_in = descs[i][1], descs[i][2]
            out[i] = (buf[off:off + o * n_in].view(o, n_in), buf[off + o * n_in:off + o * n_in + o])
            off += (sz + 3) // 4 * 4
    dst = [sinks[i] if sinks[i] is not None else out[i] for i in range(n)]
    for i in range(0, n, 8):
        d, t = descs[i:i + 8], dst[i:i + 8]
        check(lib().ps_mlp_unpack_grad_layers(len(d), _ptr_array([x[0] for x in d]), n_parts, part_stride,
                                              _int_array([x[1] for x in d]), _int_array([x[2] for x in d]),
                                              _ptr_array([_p(x[3]) for x in d]), _int_array([x[4] for x in d]),
                                              _int_array([x[5] for x in d]), _ptr_array([_p(g[0]) for g in t]),
                                              _ptr_array([_p(g[1]) for g in t]), _stream()), "ps_mlp_unpack_grad_layers")
    return out


_SPEC_CACHE = {}


def mlp_spec(dims: Sequence[int]) -> MlpSpec:
    key = tuple(dims)
    if key not in _SPEC_CACHE:
        spec = MlpSpec(dims)
        if not lib().ps_mlp_shape_supported(dims[0], dims[1], dims[-1], len(dims) - 1):
            raise NotImplementedError(f"presight_amd MLP: no HIP kernel instantiated for layer dims {list(dims)}")
        _SPEC_CACHE[key] = spec
    return _SPEC_CACHE[key]


class _Mlp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_act, *wb):
        x = _f32(x, "MLP input")
        layers = [(wb[2 * i], wb[2 * i + 1]) for i in range(len(wb) // 2)]
        dims = [layers[0][0].shape[1]] + [W.shape[0] for W, _ in layers]
        spec = mlp_spec(dims)
        packed = spec.pack(layers, x.device)
        N = x.shape[0]
        y = torch.empty(N, dims[-1], device=x.device, dtype=torch.float32)
        check(lib().ps_mlp_fwd(_p(x), _p(packed), _p(y), N, dims[0], dims[1], dims[-1], spec.nl, out_act, _stream()), "ps_mlp_fwd")
        ctx.save_for_backward(x, packed)
        ctx.meta = (dims, out_act, [tuple(W.shape) for W, _ in layers])
        ctx.sinks = layer_sinks(layers)
        ctx.direct = direct_params(*wb)
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes

        x, packed = ctx.saved_tensors
        dims, out_act, shapes = ctx.meta
        spec = mlp_spec(dims)
        dy = _f32(dy)
        N = x.shape[0]
        pf, gf, npart = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        check(lib().ps_mlp_sizes(dims[0], dims[1], dims[-1], spec.nl, N, ctypes.byref(pf), ctypes.byref(gf), ctypes.byref(npart)),
              "ps_mlp_sizes")
        assert pf.value == spec.packed and gf.value == spec.g_total, (pf.value, spec.packed, gf.value, spec.g_total)
        gpart = torch.empty(npart.value, spec.g_total, device=x.device, dtype=torch.float32)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        check(lib().ps_mlp_bwd(_p(x), _p(dy), _p(packed), _p(dx), _p(gpart), N, dims[0], dims[1], dims[-1], spec.nl, out_act,
                               _stream()), "ps_mlp_bwd")
        grads = spec.unpack_grads(gpart, npart.value, spec.g_total, 0, shapes, ctx.sinks)
        mark_touched(ctx.direct)
        return (dx, None, *flatten_grads(grads))


def layer_sinks(layers) -> list:
    """[(dW sink, db sink) | None] per layer: direct accumulation needs both tensors of a layer to opt in"""
    out = []
    for W, b in layers:
        gw, gb = grad_sink(W), grad_sink(b)
        out.append((gw, gb) if gw is not None and gb is not None else None)
    return out


def flatten_grads(grads) -> list:
    flat = []
    for g in grads:
        flat += [None, None] if g is None else [g[0], g[1]]
    return flat


def mlp(x: Tensor, layers: Sequence[Tuple[Tensor, Tensor]], out_act: Optional[str] = None) -> Tensor:
    """y = MLP(x) with ReLU between layers and an optional sigmoid on the output."""
    flat = []
    for W, b in layers:
        flat += [W, b]
    act = {None: 0, "none": 0, "sigmoid": 1}[out_act]
    return _Mlp.apply(x, act, *flat)


# ------------------------------------------------------------------------------------------------
# point-wise operators
# ------------------------------------------------------------------------------------------------
def contract(p: Tensor, aabb: Tensor, contract_: bool = True) -> Tuple[Tensor, Tensor]:
    p = _f32(p)
    M = p.shape[0]
    u = torch.empty_like(p)
    sel = torch.empty(M, device=p.device, dtype=torch.uint8)
    check(lib().ps_contract(_p(p), _p(_f32(aabb)), M, int(contract_), _p(u), _p(sel), _stream()), "ps_contract")
    return u, sel.bool()


def sh4(dirs: Tensor) -> Tensor:
    d = _f32(dirs)
    out = torch.empty(d.shape[0], 16, device=d.device, dtype=torch.float32)
    check(lib().ps_sh4(_p(d), d.shape[0], _p(out), _stream()), "ps_sh4")
    return out


def sh_encode(x: Tensor, levels: int = 4) -> Tensor:
    """real spherical harmonics of `levels` levels evaluated on x as given -> [M, levels^2]"""
    x = _f32(x)
    out = torch.empty(x.shape[0], levels * levels, device=x.device, dtype=torch.float32)
    check(lib().ps_sh_encode(_p(x), x.shape[0], levels, _p(out), _stream()), "ps_sh_encode")
    return out


def route(p: Tensor, centroids: Tensor) -> Tensor:
    p = _f32(p)
    c = _f32(centroids)
    a = torch.empty(p.shape[0], device=p.device, dtype=torch.int32)
    check(lib().ps_route(_p(p), p.shape[0], _p(c), c.shape[0], _p(a), _stream()), "ps_route")
    return a


def sample_positions(origins: Tensor, dirs: Tensor, ebins: Tensor) -> Tensor:
    R, S = ebins.shape[0], ebins.shape[1] - 1
    pos = torch.empty(R * S, 3, device=ebins.device, dtype=torch.float32)
    check(lib().ps_sample_positions(_p(_f32(origins)), _p(_f32(dirs)), _p(_f32(ebins)), R, S, _p(pos), _stream()),
          "ps_sample_positions")
    return pos


# ------------------------------------------------------------------------------------------------
# per-ray operators
# ------------------------------------------------------------------------------------------------
def generate_rays(ray_indices: Tensor, c2w: Tensor, fx: Tensor, fy: Tensor, cx: Tensor, cy: Tensor):
    if not ray_indices.is_cuda:
        raise RuntimeError("presight_amd: ray_indices must live on the GPU")
    ri = ray_indices.to(torch.int64).contiguous()
    R = ri.shape[0]
    dev = ri.device
    o = torch.empty(R, 3, device=dev)
    d = torch.empty(R, 3, device=dev)
    pa = torch.empty(R, 1, device=dev)
    dn = torch.empty(R, 1, device=dev)
    check(lib().ps_generate_rays(_p(ri), _p(_f32(c2w)), _p(_f32(fx)), _p(_f32(fy)), _p(_f32(cx)), _p(_f32(cy)), R, _p(o), _p(d),
                                 _p(pa), _p(dn), _stream()), "ps_generate_rays")
    return o, d, pa, dn


def spaced_bins(num_rays: int, S: int, near: float, far: float, thr: float, jitter: Optional[Tensor], device, points=None):
    """-> (sbins, ebins) [R, S+1]; points = (origins, dirs, aabb, contract): -> (sbins, ebins, u [R*S,3], sel [R*S]) -- the points of
    the field that will be evaluated on these bins (field_ops.field_points) from the same launch"""
    sb = torch.empty(num_rays, S + 1, device=device)
    eb = torch.empty(num_rays, S + 1, device=device)
    j = None if jitter is None else _f32(jitter).view(-1)
    if points is None:
        check(lib().ps_spaced_bins(_p(j), num_rays, S, near, far, thr, _p(sb), _p(eb), _stream()), "ps_spaced_bins")
        return sb, eb
    origins, dirs, aabb, contract = points
    u, sel = torch.empty(num_rays * S, 3, device=device), torch.empty(num_rays * S, device=device)
    check(lib().ps_spaced_bins_points(_p(j), num_rays, S, near, far, thr, _p(sb), _p(eb), _p(_f32(origins)), _p(_f32(dirs)), _p(_f32(aabb)),
                                      int(contract), _p(u), _p(sel), _stream()), "ps_spaced_bins_points")
    return sb, eb, u, sel


class _Weights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ebins, sigma):
        ebins, sigma = _f32(ebins), _f32(sigma)
        R, S = sigma.shape
        w = torch.empty_like(sigma)
        check(lib().ps_weights_fwd(_p(ebins), _p(sigma), R, S, _p(w), _stream()), "ps_weights_fwd")
        ctx.save_for_backward(ebins, sigma)
        return w

    @staticmethod
    def backward(ctx, dw):
        ebins, sigma = ctx.saved_tensors
        R, S = sigma.shape
        ds = torch.empty_like(sigma)
        check(lib().ps_weights_bwd(_p(ebins), _p(sigma), _p(_f32(dw)), R, S, _p(ds), _stream()), "ps_weights_bwd")
        return None, ds


class _WeightsResample(torch.autograd.Function):
    """RaySamples.get_weights + PDFSampler of the next level (+ the next field's points) in one launch (ps_weights_resample); only
    the weights are differentiable (w.r.t. the densities), the new bin edges are detached like the reference's
    (ns/model_components/ray_samplers.py:360)"""

    @staticmethod
    def forward(ctx, ebins, sigma, sbins, jitter, n_new, anneal, pad, eps, near, far, thr, points):
        ebins, sigma, sbins = _f32(ebins), _f32(sigma), _f32(sbins)
        R, S = sigma.shape
        dev = sigma.device
        ctx.set_materialize_grads(False)
        w = torch.empty_like(sigma)
        nsb, neb = torch.empty(R, n_new + 1, device=dev), torch.empty(R, n_new + 1, device=dev)
        j = None if jitter is None else _f32(jitter).view(-1)
        if points is not None:
            origins, dirs, aabb, contract = points
            origins, dirs, aabb = _f32(origins), _f32(dirs), _f32(aabb)
            u, sel = torch.empty(R * n_new, 3, device=dev), torch.empty(R * n_new, device=dev)
        else:
            origins = dirs = aabb = u = sel = None
            contract = 0
        check(lib().ps_weights_resample(_p(ebins), _p(sigma), _p(sbins), _p(j), R, S, n_new, float(anneal), float(pad), float(eps), float(near),
                                        float(far), float(thr), _p(w), _p(nsb), _p(neb), _p(origins), _p(dirs), _p(aabb), int(contract), _p(u),
                                        _p(sel), _stream()), "ps_weights_resample")
        ctx.save_for_backward(ebins, sigma)
        empty = torch.empty(0, device=dev)
        outs = (w, nsb, neb, u if u is not None else empty, sel if sel is not None else empty)
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, dw, *_):
        if dw is None:
            return (None,) * 12
        ebins, sigma = ctx.saved_tensors
        R, S = sigma.shape
        ds = torch.empty_like(sigma)
        check(lib().ps_weights_bwd(_p(ebins), _p(sigma), _p(_f32(dw)), R, S, _p(ds), _stream()), "ps_weights_bwd")
        return (None, ds) + (None,) * 10


def weights_resample(ebins: Tensor, sigma: Tensor, sbins: Tensor, n_new: int, jitter: Optional[Tensor], anneal: float, near: float, far: float,
                     thr: float, pad: float = 0.01, eps: float = float(torch.finfo(torch.float32).eps), points=None):
    """-> (weights [R,S] (differentiable w.r.t. sigma), new sbins, new ebins [R, n_new+1], u | None, sel | None): weights_from_density +
    pdf_resample (+ field_ops.field_points of the field that will be evaluated on the new bins; points = (origins, dirs, aabb,
    contract)) from one launch"""
    w, nsb, neb, u, sel = _WeightsResample.apply(ebins, sigma, sbins, jitter, int(n_new), anneal, pad, eps, near, far, thr, points)
    return (w, nsb, neb, u, sel) if points is not None else (w, nsb, neb, None, None)


def weights_from_density(ebins: Tensor, sigma: Tensor) -> Tensor:
    """ebins [R,S+1] euclidean bin edges, sigma [R,S] -> weights [R,S] (differentiable w.r.t. sigma)."""
    return _Weights.apply(ebins, sigma)


def pdf_resample(weights: Tensor, sbins: Tensor, n_new: int, jitter: Optional[Tensor], anneal: float, near: float,
                 far: float, thr: float, pad: float = 0.01, eps: float = float(torch.finfo(torch.float32).eps)):
    w = _f32(weights.detach())
    sb = _f32(sbins)
    R, S = w.shape
    nsb = torch.empty(R, n_new + 1, device=w.device)
    neb = torch.empty(R, n_new + 1, device=w.device)
    j = None if jitter is None else _f32(jitter).view(-1)
    check(lib().ps_pdf_resample(_p(w), _p(sb), _p(j), R, S, n_new, float(anneal), pad, eps, near, far, thr, _p(nsb), _p(neb),
                                _stream()), "ps_pdf_resample")
    return nsb, neb


_MINMAX_INIT = {}
_MINMAX_HOOK = None


def set_depth_clip_hook(fn):
    """fn(minmax [2] device tensor = {min, max} sample midpoint of this rank's batch) runs between the composite kernel and
    the clip of the expected depth.  DepthRenderer("expected") clips to the BATCH-global sample range
    (ns/model_components/renderers.py:377-379); under data parallelism every rank sees its shard only (so does the
    reference under DDP) -- presight_amd.dist.global_depth_clip() installs a hook that takes the bounds over all ranks, which
    reproduces the single-process result.  None removes the hook."""
    global _MINMAX_HOOK
    _MINMAX_HOOK = fn


def _apply_minmax_hook(minmax):
    if _MINMAX_HOOK is not None:
        _MINMAX_HOOK(minmax)


def _minmax_init(dev):
    """{+inf, 0} on `dev`, created once (a host->device copy per call would also break hipGraph capture of a step)"""
    if dev not in _MINMAX_INIT:
        _MINMAX_INIT[dev] = torch.tensor([float("inf"), 0.0], device=dev)
    return _MINMAX_INIT[dev]


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, ebins, rgb_s, sem_s, threshold):
        weights, ebins = _f32(weights), _f32(ebins)
        R, S = weights.shape
        dev = weights.device
        rgb_s = None if rgb_s is None else _f32(rgb_s)
        sem_s = None if sem_s is None else _f32(sem_s)
        C = 0 if sem_s is None else sem_s.shape[-1]
        rgb = torch.empty(R, 3, device=dev) if rgb_s is not None else None
        sem = torch.empty(R, C, device=dev) if sem_s is not None else None
        acc = torch.empty(R, 1, device=dev)
        depth = torch.empty(R, 1, device=dev)
        expd = torch.empty(R, 1, device=dev)
        minmax = _minmax_init(dev).clone()
        check(lib().ps_composite_fwd(_p(weights), _p(ebins), _p(rgb_s), _p(sem_s), R, S, C, threshold, _p(rgb), _p(acc), _p(depth),
                                     _p(expd), _p(sem), _p(minmax), _stream()), "ps_composite_fwd")
        _apply_minmax_hook(minmax)
        raw = torch.empty_like(expd)  # 1 where the batch-global clip left the value alone (its derivative)
        check(lib().ps_clip(_p(expd), R, _p(minmax), _p(raw), _stream()), "ps_clip")
        ctx.save_for_backward(weights, ebins, rgb_s, sem_s, raw, expd)
        ctx.C = C
        ctx.mark_non_differentiable(depth)
        outs = (rgb if rgb is not None else torch.empty(0, device=dev), acc, depth, expd,
                sem if sem is not None else torch.empty(0, device=dev))
        return outs

    @staticmethod
    def backward(ctx, d_rgb, d_acc, _d_depth, d_exp, d_sem):
        weights, ebins, rgb_s, sem_s, raw, expd = ctx.saved_tensors
        R, S = weights.shape
        dev = weights.device
        d_rgb = _f32(d_rgb) if rgb_s is not None and d_rgb is not None else None
        d_sem = _f32(d_sem) if sem_s is not None and d_sem is not None else None
        d_acc = _f32(d_acc) if d_acc is not None else None
        if d_exp is not None:
            d_exp = _f32(d_exp * raw)  # gradient of clip
        dw = torch.empty_like(weights)
        d_rgb_s = torch.empty_like(rgb_s) if (rgb_s is not None and ctx.needs_input_grad[2]) else None
        d_sem_s = torch.empty_like(sem_s) if (sem_s is not None and ctx.needs_input_grad[3]) else None
        check(lib().ps_composite_bwd(_p(weights), _p(ebins), _p(rgb_s) if d_rgb is not None else None,
                                     _p(sem_s) if d_sem is not None else None, _p(d_rgb), _p(d_acc), _p(d_sem), _p(d_exp), R, S,
                                     ctx.C, _p(dw), _p(d_rgb_s) if d_rgb is not None else None,
                                     _p(d_sem_s) if d_sem is not None else None, None, None, _stream()), "ps_composite_bwd")
        if d_rgb is None and d_rgb_s is not None:
            d_rgb_s.zero_()
        if d_sem is None and d_sem_s is not None:
            d_sem_s.zero_()
        return dw, None, d_rgb_s, d_sem_s, None


def threshold_depth(weights: Tensor, ebins: Tensor, threshold: float = 0.5) -> Tensor:
    """Depth of the first sample whose cumulative weight reaches `threshold` ([R,1], no gradient);
    ns/model_components/renderers.py:352-362."""
    w, eb = _f32(weights.detach()), _f32(ebins)
    R, S = w.shape
    depth = torch.empty(R, 1, device=w.device)
    check(lib().ps_composite_fwd(_p(w), _p(eb), None, None, R, S, 0, threshold, None, None, _p(depth), None, None, None, _stream()),
          "ps_composite_fwd")
    return depth


def composite(weights: Tensor, ebins: Tensor, rgb_s: Optional[Tensor], sem_s: Optional[Tensor], threshold: float = 0.5):
    """-> (rgb [R,3], acc [R,1] (unclamped), threshold depth [R,1], expected depth [R,1] (batch-clipped), sem [R,C])."""
    return _Composite.apply(weights, ebins, rgb_s, sem_s, threshold)


# ------------------------------------------------------------------------------------------------
# per-ray tail: embedding lookup, sky blending
# ------------------------------------------------------------------------------------------------
class _EmbedCat(torch.autograd.Function):
    """cat([table_k[idx_k] for k], dim=-1) in one launch per table; backward scatter-adds the rows (in place into the
    parameter's flat .grad when the owner opted in, see grad_sink)."""

    @staticmethod
    def forward(ctx, n_tab, *args):
        idxs, tables = args[:n_tab], args[n_tab:]
        R = idxs[0].shape[0]
        dev = tables[0].device
        widths = [t.shape[1] for t in tables]
        out = torch.empty(R, sum(widths), device=dev)
        # two tables (the model's case): one launch, indices read in place through their stride (camera index = ray_indices[:, 0])
        pair = n_tab == 2 and all(i.dtype == torch.int64 and i.is_cuda and i.dim() >= 1 for i in idxs)
        if pair:
            idxs = [i.reshape(-1) for i in idxs]
            pair = all(i.numel() == R and (R <= 1 or i.stride(0) >= 1) for i in idxs)
        if pair:
            tabs = [_f32(t) for t in tables]
            strides = [1 if R <= 1 else i.stride(0) for i in idxs]
            check(lib().ps_embedding_pair_fwd(_p(idxs[0]), strides[0], _p(tabs[0]), widths[0], _p(idxs[1]), strides[1], _p(tabs[1]), widths[1], R,
                                              _p(out), _stream()), "ps_embedding_pair_fwd")
            ctx.strides = strides
        else:
            idxs = [i.reshape(-1).to(torch.int64).contiguous() for i in idxs]
            col = 0
            for i, t in zip(idxs, tables):
                check(lib().ps_embedding_fwd(_p(i), _p(_f32(t)), R, t.shape[1], out.shape[1], col, _p(out), _stream()), "ps_embedding_fwd")
                col += t.shape[1]
            ctx.strides = None
        ctx.save_for_backward(*idxs)
        ctx.shapes = [tuple(t.shape) for t in tables]
        ctx.sinks = [grad_sink(t) for t in tables]
        ctx.direct = direct_params(*tables)
        return out

    @staticmethod
    def backward(ctx, dout):
        idxs = ctx.saved_tensors
        dout = _f32(dout)
        R = dout.shape[0]
        grads, col = [], 0
        dsts = [sink if sink is not None else torch.zeros(rows, D, device=dout.device) for (rows, D), sink in zip(ctx.shapes, ctx.sinks)]
        if ctx.strides is not None:
            (r0, D0), (r1, D1) = ctx.shapes
            check(lib().ps_embedding_pair_bwd(_p(idxs[0]), ctx.strides[0], r0, D0, _p(dsts[0]), _p(idxs[1]), ctx.strides[1], r1, D1, _p(dsts[1]),
                                              _p(dout), R, _stream()), "ps_embedding_pair_bwd")
        else:
            for i, (rows, D), dst in zip(idxs, ctx.shapes, dsts):
                check(lib().ps_embedding_bwd(_p(i), _p(dout), R, D, rows, dout.shape[1], col, _p(dst), _stream()), "ps_embedding_bwd")
                col += D
        grads = [None if sink is not None else dst for sink, dst in zip(ctx.sinks, dsts)]
        mark_touched(ctx.direct)
        return (None, *([None] * len(idxs)), *grads)


def embed_cat(indices: Sequence[Tensor], tables: Sequence[Tensor]) -> Tensor:
    """[R, sum(D_k)] = concatenation of nn.Embedding lookups (ns/field_components/embedding.py:27-55)"""
    return _EmbedCat.apply(len(tables), *indices, *tables)


class _SkyBlend(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_f, acc_raw, sem_f, sky_rgb, sky_sem):
        rgb_f, acc_raw = _f32(rgb_f), _f32(acc_raw)
        R = rgb_f.shape[0]
        dev = rgb_f.device
        sem_f = None if sem_f is None else _f32(sem_f)
        sky_rgb = None if sky_rgb is None else _f32(sky_rgb)
        sky_sem = None if (sky_sem is None or sem_f is None) else _f32(sky_sem)
        C = 0 if sem_f is None else sem_f.shape[1]
        rgb, acc = torch.empty(R, 3, device=dev), torch.empty(R, 1, device=dev)
        sem = torch.empty(R, C, device=dev) if sem_f is not None else None
        check(lib().ps_sky_blend_fwd(_p(rgb_f), _p(acc_raw), _p(sem_f), _p(sky_rgb), _p(sky_sem), R, C, _p(rgb), _p(acc), _p(sem),
                                     _stream()), "ps_sky_blend_fwd")
        ctx.save_for_backward(acc_raw, sky_rgb, sky_sem)
        ctx.C = C
        return rgb, acc, (sem if sem is not None else torch.empty(0, device=dev))

    @staticmethod
    def backward(ctx, d_rgb, d_acc, d_sem):
        acc_raw, sky_rgb, sky_sem = ctx.saved_tensors
        R = acc_raw.shape[0]
        dev = acc_raw.device
        d_rgb = None if d_rgb is None else _f32(d_rgb)
        d_acc = None if d_acc is None else _f32(d_acc)
        d_sem = None if (d_sem is None or ctx.C == 0) else _f32(d_sem)
        d_acc_raw = torch.empty_like(acc_raw)
        d_sky_rgb = torch.empty_like(sky_rgb) if sky_rgb is not None else None
        d_sky_sem = torch.empty_like(sky_sem) if sky_sem is not None else None
        check(lib().ps_sky_blend_bwd(_p(acc_raw), _p(sky_rgb), _p(sky_sem), _p(d_rgb), _p(d_acc), _p(d_sem), R, ctx.C, _p(d_acc_raw),
                                     _p(d_sky_rgb), _p(d_sky_sem), _stream()), "ps_sky_blend_bwd")
        return d_rgb, d_acc_raw, d_sem, d_sky_rgb, d_sky_sem


def sky_blend(rgb_f: Tensor, acc_raw: Tensor, sem_f: Optional[Tensor], sky_rgb: Optional[Tensor], sky_sem: Optional[Tensor]):
    """-> (rgb, accumulation = clamp(acc_raw, 0, 1), semantics | None): rgb_f + (1-acc) sky_rgb, sem_f + (1-acc) sky_sem
    (ns/models/PreSight/nerfacto_nusc_ms.py:512-533); without sky tensors only the clamp is applied."""
    rgb, acc, sem = _SkyBlend.apply(rgb_f, acc_raw, sem_f, sky_rgb, sky_sem)
    return rgb, acc, (sem if sem_f is not None else None)
