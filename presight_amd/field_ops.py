"""Field-level fused operators (autograd glue over ps_field_points / ps_grid_encode / ps_*_field_fwd|bwd /
ps_grid_scatter).  One call evaluates a whole PreSight field for a batch of points:

    prop_field : ns/fields/PreSight/prop_density_field.py:129-153
    main_field : ns/fields/PreSight/ingp_field.py:168-237 (density_fn + get_outputs)

Positions carry no gradient (the reference detaches the sampled bins, ray_samplers.py:360)."""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import prof
from ._lib import check, lib
from .ops import (MlpSpec, _f32, _p, _stream, chain_colmap, direct_params, flatten_grads, grad_sink, layer_sinks, linear_colmap,
                  mark_touched, pack_layers, unpack_layers)


@dataclass(frozen=True)
class GridCfg:
    num_levels: int
    features_per_level: int
    log2_hashmap_size: int

    @property
    def out_dim(self) -> int:
        return self.num_levels * self.features_per_level


def field_points(aabb: Tensor, contract: bool, pos: Optional[Tensor] = None, origins: Optional[Tensor] = None,
                 dirs: Optional[Tensor] = None, ebins: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """-> (u [N,3] in [0,1] (masked), sel [N] float 0/1).  Either world positions or a ray batch + bin edges."""
    aabb = _f32(aabb)
    if pos is not None:
        pos = _f32(pos).view(-1, 3)
        N, S, dev = pos.shape[0], 0, pos.device
    else:
        origins, dirs, ebins = _f32(origins), _f32(dirs), _f32(ebins)
        S = ebins.shape[1] - 1
        N, dev = ebins.shape[0] * S, ebins.device
    u = torch.empty(N, 3, device=dev)
    sel = torch.empty(N, device=dev)
    check(lib().ps_field_points(_p(pos), _p(origins), _p(dirs), _p(ebins), S, _p(aabb), int(contract), N, _p(u), _p(sel),
                                _stream()), "ps_field_points")
    return u, sel


def _encode(u: Tensor, table: Tensor, scalings: Tensor, g: GridCfg, count: bool = False):
    """-> (feature planes [L,N,F], slice_counts | None).  count=True (a table backward will follow): the kernel also counts
    the records of the binned backward per (level, table slice), see ps_grid_encode."""
    N = u.shape[0]
    feat = torch.empty(g.num_levels, N, g.features_per_level, device=u.device)
    counts = None
    if count and _binned(N, g.num_levels):
        counts = torch.empty(g.num_levels * lib().ps_grid_scatter_slices(g.features_per_level, g.log2_hashmap_size), device=u.device,
                             dtype=torch.int32)
    with prof.region(f"grid_encode_L{g.num_levels}F{g.features_per_level}"):
        check(lib().ps_grid_encode(_p(u), _p(table), _p(scalings), g.num_levels, g.features_per_level, g.log2_hashmap_size, N,
                                   N * g.features_per_level, _p(feat), _p(counts), _stream()), "ps_grid_encode")
    return feat, counts


_CALL = threading.local()


def _apply(fn, *args):
    """fn.apply(*args) with the CALLER's grad mode on record.  Inside a Function.forward grad mode is always off, and
    ctx.needs_input_grad ignores torch.no_grad(): without this an inference call on trainable parameters would still count the
    table backward's records during the hash encode and keep 1.4 KB of activations per point for a backward that never comes."""
    prev = getattr(_CALL, "grad", True)
    _CALL.grad = torch.is_grad_enabled()
    try:
        return fn.apply(*args)
    finally:
        _CALL.grad = prev


def _training(needs_grad) -> bool:
    """needs_input_grad of a node's parameters AND grad mode was on where the node was applied"""
    return bool(needs_grad) and getattr(_CALL, "grad", True)


_WORKSPACES = {}
SCATTER_IMPL = "binned"  # "binned" (records + int64 LDS accumulation) or "owner" (LDS slice-owner scan)
# the coarsest level of a one-table backward as a dense int64 histogram instead of records (csrc/encode.hip level0_hist_kernel; phase bit 3 of
# the binned entry points): bit-identical, measured neutral on cfg 2 (EXPERIMENTS.md A.7) -> off unless PRESIGHT_DENSE_LEVEL0=1
DENSE_LEVEL0 = __import__("os").environ.get("PRESIGHT_DENSE_LEVEL0", "0") == "1"
# training forward of the main field keeps its hidden activations (1.6 KB/point) for the backward; PRESIGHT_KEEP_ACTIVATIONS=0
# (or this flag) switches to recomputing them there (7 GB less memory at cfg 2, same results)
KEEP_ACTIVATIONS = __import__("os").environ.get("PRESIGHT_KEEP_ACTIVATIONS", "1") != "0"


def _workspace(nbytes: int, device, owner=None) -> Tensor:
    """Scratch for the binned scatter, grown on demand and reused across steps (stream-ordered reuse is safe: every
    consumer of the previous contents was enqueued on the same stream before the next producer).  owner: a table whose gradient is
    exchanged as RECORDS (dist.FlatGrads record buckets) keeps a workspace of its own -- its records wait there for the bucket's turn
    in the exchange order, possibly past the next table backward on the same stream."""
    key = (str(device), _stream()) if owner is None else (str(device), "records", id(owner))  # per stream: the proposal networks may run beside the main field (samplers.prop_stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.05) + 4096, device=device, dtype=torch.uint8)
        _WORKSPACES[key] = ws
    return ws


def _binned(N: int, L: int = 16) -> bool:
    """the binned backward addresses its record streams with 32-bit offsets: same bound as ps_grid_scatter_binned's own guard
    (N * L * 8 records + stream padding < 2^32); beyond it the slice-owner scatter takes over"""
    return SCATTER_IMPL == "binned" and N * L * 8 + 4096 + 4 * L * 256 < (1 << 32)


def _scatter_ws(g: GridCfg, N: int, device, table: Optional[Tensor] = None) -> Optional[Tensor]:
    """workspace of the binned table backward; its first L words receive the per-level max |d(feature)| straight from the
    field backward kernel (level_absmax argument), which saves the scatter its own pass over d(features)"""
    if not _binned(N, g.num_levels):
        return None
    return _workspace(lib().ps_grid_scatter_workspace(g.num_levels, g.features_per_level, g.log2_hashmap_size, N), device,
                      owner=table if getattr(table, "_ps_sparse", None) is not None else None)


def _hand_over_records(tables: Sequence[Tensor], ws: Tensor, g: GridCfg, N: int, K: int, dtable: Optional[Tensor], dtables: Optional[Tensor]):
    """record (sparse) exchange: phase 1 of the binned backward has left this step's records in `ws`; the owner of the flat gradient
    buffer exchanges them when the bucket's turn comes and runs the accumulate pass over all ranks' runs (dist.FlatGrads._sparse_exchange)"""
    fg, bucket = tables[0]._ps_sparse
    L, F, l2t = g.num_levels, g.features_per_level, g.log2_hashmap_size
    lay = (ctypes.c_int64 * 9)()
    check(lib().ps_grid_scatter_layout(L, F, l2t, N, K, lay), "ps_grid_scatter_layout")

    def accumulate(run_starts, run_counts, n_runs, rec_idx, rec_val, plane_stride, gmax_bits, n_points_total, out_scale, item_begin, item_end):
        check(lib().ps_grid_accumulate_runs(_p(run_starts), _p(run_counts), n_runs, _p(rec_idx), _p(rec_val), plane_stride, _p(gmax_bits), L, F, l2t, K,
                                            int(n_points_total), _p(dtable), _p(dtables), float(out_scale), item_begin, item_end, _stream()),
              "ps_grid_accumulate_runs")

    fg.sparse_records(bucket, dict(ws=ws, layout=list(lay), L=L, F=F, log2T=l2t, K=K, n_points=N, accumulate=accumulate, keep=(dtable, dtables)))


def _sink_is_zero(param: Optional[Tensor]) -> bool:
    """True when `param`'s in-place gradient sink is known to hold zeros: its owner (presight_amd.dist.FlatGrads) cleared it
    at the start of the step (zero_ resets `_ps_touched`) and nothing has been written to it since"""
    return param is not None and getattr(param, "_ps_touched", True) is False


def _refuse_second_contribution(tables: Sequence[Optional[Tensor]]) -> None:
    for t in tables:
        if t is not None and getattr(t, "_ps_fused_done", False):
            raise RuntimeError("presight_amd: a hash table whose Adam update is fused into its backward received a second gradient "
                               "contribution in one step (HipAdam.enable_fused_tables needs exactly one; PRESIGHT_FUSED_TABLE_ADAM=0)")


def _fused_adam(tables: Sequence[Optional[Tensor]], routed: bool):
    """the Adam arguments for a table backward that also applies the optimizer step (HipAdam.enable_fused_tables), or None: every
    table must be owned by the same optimizer and its gradient sink must still hold the zeros the step started with"""
    opt = getattr(tables[0], "_ps_fused_adam", None) if tables and tables[0] is not None else None
    if opt is None or any(t is None or getattr(t, "_ps_fused_adam", None) is not opt or not _sink_is_zero(t) for t in tables):
        return None
    return opt.fused_table_args(list(tables), routed)


def _scatter(u: Tensor, dfeat: Tensor, scalings: Tensor, g: GridCfg, table_shape, sink: Optional[Tensor] = None,
             counts: Optional[Tensor] = None, ws_with_absmax: Optional[Tensor] = None, sink_owner: Optional[Tensor] = None) -> Optional[Tensor]:
    """table gradient; with `sink` (the parameter's pre-allocated .grad, ops.grad_sink) it is ADDED there and None is returned.
    counts: slice record counts from the forward encode of the same points (_encode(count=True))."""
    N = u.shape[0]
    acc = int(sink is not None)
    if acc and _binned(N, g.num_levels) and _sink_is_zero(sink_owner):
        acc = 2  # first contribution of the step into the zeroed flat gradient buffer: written, not read-modify-written
    dtable = sink if sink is not None else torch.empty(table_shape, device=u.device, dtype=torch.float32)
    L, F, l2t = g.num_levels, g.features_per_level, g.log2_hashmap_size
    pieces = getattr(sink_owner, "_ps_parts", 1) if (sink is not None and sink_owner is not None) else 1
    _refuse_second_contribution([sink_owner])
    fused = _fused_adam([sink_owner], routed=False) if (acc == 2 and pieces == 1) else None
    sparse = sink is not None and getattr(sink_owner, "_ps_sparse", None) is not None and _binned(N, L) and sink_owner._ps_sparse[0]._distributed()
    with prof.region(f"grid_scatter_L{L}F{F}"):
        if sparse:
            # the gradient is exchanged as RECORDS: write them (phase 1) into the table's own workspace and hand them over; the accumulate
            # pass runs on the slices' owner, over all ranks' records (dist.FlatGrads._sparse_exchange)
            ws = ws_with_absmax if ws_with_absmax is not None else _scatter_ws(g, N, u.device, sink_owner)
            check(lib().ps_grid_scatter_binned_part(_p(u), _p(dfeat), _p(scalings), L, F, l2t, N, N * F, _p(dtable), acc, _p(counts),
                                                    int(ws_with_absmax is not None), _p(ws), 1 | 4, 0, 0, _stream()), "ps_grid_scatter_binned_part")
            _hand_over_records([sink_owner], ws, g, N, 1, dtable, None)  # (phase bit 2: every level as records -- they are what travels)
        elif fused is not None:
            # single-process training: the accumulate pass applies the table's Adam step itself, the gradient is never written
            ws = ws_with_absmax if ws_with_absmax is not None else _scatter_ws(g, N, u.device)
            for phase, reg in _scatter_phases(L, F):
                with (prof.region(reg) if reg else contextlib.nullcontext()):
                    check(lib().ps_grid_scatter_binned_adam(_p(u), _p(dfeat), _p(scalings), L, F, l2t, N, N * F, _p(dtable), _p(counts),
                                                            int(ws_with_absmax is not None), _p(ws), phase | (8 if DENSE_LEVEL0 else 0), 0, -1, *fused, _stream()),
                          "ps_grid_scatter_binned_adam")
        elif _binned(N, L) and pieces > 1:
            # the gradient is exchanged in `pieces` level groups (presight_amd.dist.FlatGrads splits): one accumulate launch per group,
            # every group handed to the exchange as soon as its launch is enqueued -- its reduce-scatter runs under the next launches
            ws = ws_with_absmax if ws_with_absmax is not None else _scatter_ws(g, N, u.device)
            items = lib().ps_grid_scatter_items(L, F, l2t, 1)
            per_level = items // L
            if L % pieces:
                raise RuntimeError(f"presight_amd: a table of {L} levels cannot be exchanged in {pieces} equal level groups")
            args = (_p(u), _p(dfeat), _p(scalings), L, F, l2t, N, N * F, _p(dtable), acc, _p(counts), int(ws_with_absmax is not None), _p(ws))
            d0 = 8 if DENSE_LEVEL0 else 0
            check(lib().ps_grid_scatter_binned_part(*args, 1 | d0, 0, 0, _stream()), "ps_grid_scatter_binned_part")
            for gi in range(pieces):
                l0, l1 = L * gi // pieces, L * (gi + 1) // pieces
                check(lib().ps_grid_scatter_binned_part(*args, 2 | d0, l0 * per_level, l1 * per_level, _stream()), "ps_grid_scatter_binned_part")
                sink_owner._ps_part_done(sink_owner, gi)
        elif _binned(N, L):
            ws = ws_with_absmax if ws_with_absmax is not None else _scatter_ws(g, N, u.device)
            if DENSE_LEVEL0:
                check(lib().ps_grid_scatter_binned_part(_p(u), _p(dfeat), _p(scalings), L, F, l2t, N, N * F, _p(dtable), acc, _p(counts),
                                                        int(ws_with_absmax is not None), _p(ws), 3 | 8, 0, -1, _stream()), "ps_grid_scatter_binned_part")
            else:
                check(lib().ps_grid_scatter_binned(_p(u), _p(dfeat), _p(scalings), L, F, l2t, N, N * F, _p(dtable), acc, _p(counts),
                                                   int(ws_with_absmax is not None), _p(ws), _stream()), "ps_grid_scatter_binned")
        else:
            check(lib().ps_grid_scatter(_p(u), _p(dfeat), _p(scalings), L, F, l2t, N, N * F, _p(dtable), acc, _stream()),
                  "ps_grid_scatter")
    return None if sink is not None else dtable


def _scatter_phases(L: int, F: int):
    """the binned table backward as one call -- or, while bench.py times its two long kernels, as two calls of the same entry point
    (phase 1: prefix + record writer `bin_kernel`, phase 2: `accumulate_kernel`), each inside its own HIP-event region: same kernels,
    same order, same stream"""
    names = (f"bin_kernel_L{L}F{F}", f"accumulate_kernel_L{L}F{F}")
    if prof.enabled(names[0]) and prof.enabled(names[1]):
        return ((1, names[0]), (2, names[1]))
    return ((3, None),)


def _layers(flat: Sequence[Tensor]) -> List[Tuple[Tensor, Tensor]]:
    return [(flat[2 * i], flat[2 * i + 1]) for i in range(len(flat) // 2)]


# ------------------------------------------------------------------------------------------------ proposal field
_PROP_SPECS = {}


def _prop_spec(LF: int, hidden: int) -> MlpSpec:
    key = (LF, hidden)
    if key not in _PROP_SPECS:
        _PROP_SPECS[key] = MlpSpec([LF, hidden, 1])
    return _PROP_SPECS[key]


class _PropField(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, sel, table, scalings, g: GridCfg, *wb):
        layers = _layers(wb)
        hidden = layers[0][0].shape[0]
        spec = _prop_spec(g.out_dim, hidden)
        N = u.shape[0]
        table = _f32(table, "hash table")
        feat, counts = _encode(u, table, scalings, g, count=_training(ctx.needs_input_grad[2]))
        packed = spec.pack(layers, u.device)
        sigma = torch.empty(N, device=u.device)
        with prof.region("prop_field_fwd"):
            check(lib().ps_prop_field_fwd(_p(feat), N * g.features_per_level, g.out_dim, g.features_per_level, hidden, _p(sel),
                                          _p(packed), N, _p(sigma), _stream()), "ps_prop_field_fwd")
        ctx.save_for_backward(u, sel, scalings, feat, packed, counts)
        ctx.meta = (g, hidden, tuple(table.shape), [tuple(W.shape) for W, _ in layers])
        ctx.sinks = (grad_sink(table), layer_sinks(layers))
        ctx.table_ref = table
        ctx.direct = direct_params(table, *wb)
        return sigma

    @staticmethod
    def backward(ctx, dsigma):
        u, sel, scalings, feat, packed, counts = ctx.saved_tensors
        g, hidden, tshape, shapes = ctx.meta
        spec = _prop_spec(g.out_dim, hidden)
        N = u.shape[0]
        pf, gf, npart = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        check(lib().ps_prop_field_sizes(g.out_dim, hidden, N, ctypes.byref(pf), ctypes.byref(gf), ctypes.byref(npart)),
              "ps_prop_field_sizes")
        assert pf.value == spec.packed and gf.value == spec.g_total
        gpart = torch.empty(npart.value, spec.g_total, device=u.device)
        dfeat = torch.empty_like(feat)
        ws = _scatter_ws(g, N, u.device, ctx.table_ref)
        with prof.region("prop_field_bwd"):
            check(lib().ps_prop_field_bwd(_p(feat), N * g.features_per_level, g.out_dim, g.features_per_level, hidden, _p(sel),
                                          _p(packed), _p(_f32(dsigma)), N, _p(dfeat), _p(gpart), _p(ws), _stream()), "ps_prop_field_bwd")
        dtable = _scatter(u, dfeat, scalings, g, tshape, ctx.sinks[0], counts, ws, sink_owner=ctx.table_ref)
        grads = spec.unpack_grads(gpart, npart.value, spec.g_total, 0, shapes, ctx.sinks[1])
        mark_touched(ctx.direct)
        return (None, None, dtable, None, None, *flatten_grads(grads))


def prop_field(u: Tensor, sel: Tensor, table: Tensor, scalings: Tensor, g: GridCfg,
               layers: Sequence[Tuple[Tensor, Tensor]]) -> Tensor:
    """density [N] = trunc_exp(MLP(hash(u))) * sel."""
    flat = []
    for W, b in layers:
        flat += [W, b]
    return _apply(_PropField, u, sel, table, scalings, g, *flat)


# ------------------------------------------------------------------------------------------------ main field
GEO_DIM = 15
SEM_DIM = 64
BASE_OUT = 1 + GEO_DIM + SEM_DIM


def colour_colmap(app_dim: int) -> List[int]:
    """First layer of the colour head: k-steps 0-3 = SH16 (linear), 4-7 = base-output block 0 straight from the MFMA
    D registers (neuron 0 = sigma_raw has no column), 8-11 = appearance embedding (linear)."""
    cm = []
    for t in range(12):
        for g in range(4):
            if t < 4:
                cm.append(4 * t + g)
            elif t < 8:
                n = 4 * g + (t - 4)
                cm.append(-1 if n == 0 else 15 + n)
            else:
                c = 4 * (t - 8) + g
                cm.append(31 + c if c < app_dim else -1)
    return cm


class MainSpec:
    def __init__(self, LF: int, hidden: int, hidden_color: int, app_dim: int):
        self.key = (LF, hidden, hidden_color)
        self.app_dim = app_dim
        self.base = MlpSpec([LF, hidden, BASE_OUT])
        self.sem = MlpSpec([SEM_DIM, 64, 64, SEM_DIM], first_colmap=chain_colmap(16, SEM_DIM), ks0=16)
        self.rgb = MlpSpec([16 + GEO_DIM + app_dim, hidden_color, hidden_color, 3], first_colmap=colour_colmap(app_dim), ks0=12)
        self.p_off = [0, self.base.packed, self.base.packed + self.sem.packed]
        self.packed = self.p_off[2] + self.rgb.packed
        self.g_off = [0, self.base.g_total, self.base.g_total + self.sem.g_total]
        self.g_total = self.g_off[2] + self.rgb.g_total

    def pack(self, base, sem, rgb, device) -> Tensor:
        packed = torch.empty(self.packed, device=device)
        descs = []
        for spec, layers, off in ((self.base, base, self.p_off[0]), (self.sem, sem, self.p_off[1]), (self.rgb, rgb, self.p_off[2])):
            descs += spec.pack_descs(layers, packed[off: off + spec.packed])
        pack_layers(descs)  # all 8 layers of the three stacks in one launch
        return packed


_MAIN_SPECS = {}


def _main_spec(LF, hidden, hidden_color, app_dim) -> MainSpec:
    key = (LF, hidden, hidden_color, app_dim)
    if key not in _MAIN_SPECS:
        _MAIN_SPECS[key] = MainSpec(LF, hidden, hidden_color, app_dim)
    return _MAIN_SPECS[key]


def _stack_forward(ctx, keep_acts, feat, N, sel, dirs, app, S, g: GridCfg, want_rgb, want_sem, n_base, n_sem, wb):
    """the fused MLP stack of a main field (base MLP -> density, semantic head, colour head) on given feature planes [L,N,F];
    stores what _stack_backward needs on ctx and returns (tensors to save), (sigma, rgb, sem)"""
    layers = _layers(wb)
    base, sem_l, rgb_l = layers[:n_base], layers[n_base:n_base + n_sem], layers[n_base + n_sem:]
    hidden, hidden_color = base[0][0].shape[0], rgb_l[0][0].shape[0]
    A = rgb_l[0][0].shape[1] - 16 - GEO_DIM  # appearance columns the colour head was built with
    if want_rgb and (0 if app is None else app.shape[1]) != A:
        raise ValueError(f"colour head expects SH16 + geo15 + app{A} inputs, got an appearance embedding of width "
                         f"{0 if app is None else app.shape[1]}")
    spec = _main_spec(g.out_dim, hidden, hidden_color, A)
    dev = feat.device
    packed = spec.pack(base, sem_l, rgb_l, dev)
    sigma = torch.empty(N, device=dev)
    rgb = torch.empty(N, 3, device=dev) if want_rgb else None
    sem = torch.empty(N, SEM_DIM, device=dev) if want_sem else None
    dirs = _f32(dirs) if dirs is not None else torch.zeros(1, 3, device=dev)
    app_c = _f32(app) if (app is not None and want_rgb) else None
    # training forward of the full field: keep the hidden activations for the backward (no recompute there)
    acts = None
    if KEEP_ACTIVATIONS and keep_acts and want_rgb and want_sem and N > 0:
        acts = torch.empty((N + 15) // 16 * 16, lib().ps_main_field_act_width(g.out_dim, hidden, hidden_color), device=dev)
    with prof.region("main_field_fwd"):
        check(lib().ps_main_field_fwd(_p(feat), N * g.features_per_level, g.out_dim, g.features_per_level, hidden, hidden_color,
                                      _p(sel), _p(dirs), _p(app_c), max(S, 1), A, _p(packed), N, _p(sigma), _p(rgb), _p(sem),
                                      _p(acts), _stream()), "ps_main_field_fwd")
    ctx.stack_meta = (g, hidden, hidden_color, A, S, [tuple(W.shape) for W, _ in layers], n_base, n_sem, want_rgb, want_sem)
    ctx.layer_sinks = layer_sinks(layers)
    return (dirs, app_c, packed, acts), (sigma, rgb, sem)


def _stack_backward(ctx, sel, dirs, app, feat, packed, acts, d_sigma, d_rgb, d_sem, weights):
    """fused MLP backward + weight-gradient reduction.  weights != None: d_rgb / d_sem are per-RAY gradients (see
    ps_main_field_bwd).  -> (dapp, d(feature planes), [dW0, db0, ...] with None for in-place gradients)"""
    g, hidden, hidden_color, A, S, shapes, n_base, n_sem, want_rgb, want_sem = ctx.stack_meta
    if d_rgb is None or d_sem is None:
        acts = None  # a head without gradient: the recompute kernel skips it
    spec = _main_spec(g.out_dim, hidden, hidden_color, A)
    N = sel.shape[0]
    dev = sel.device
    pf, gf, npart = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
    offs = (ctypes.c_int64 * 6)()
    check(lib().ps_main_field_sizes(g.out_dim, hidden, hidden_color, N, ctypes.byref(pf), ctypes.byref(gf), ctypes.byref(npart),
                                    offs), "ps_main_field_sizes")
    assert pf.value == spec.packed and gf.value == spec.g_total, (pf.value, spec.packed, gf.value, spec.g_total)
    assert list(offs) == spec.p_off + spec.g_off, (list(offs), spec.p_off, spec.g_off)
    gpart = torch.empty(npart.value, spec.g_total, device=dev)
    dfeat = torch.empty_like(feat)
    dapp = torch.zeros_like(app) if app is not None else None
    # (the per-level |d(feature)| maxima are NOT tracked in this kernel, unlike the proposal backward: at its register
    #  pressure the 8 extra live values cost more (+0.2 ms) than the separate 0.13 ms absmax pass)
    dzb = _dzb_scratch(N, dev) if acts is not None else None
    with prof.region("main_field_bwd"):
        for stages in _bwd_stages(dzb):
            check(lib().ps_main_field_bwd(_p(feat), N * g.features_per_level, g.out_dim, g.features_per_level, hidden, hidden_color,
                                          _p(sel), _p(dirs), _p(app), max(S, 1), A, _p(packed), _p(d_sigma), _p(d_rgb), _p(d_sem),
                                          _p(weights), N, _p(dfeat), _p(dapp), _p(gpart), _p(acts), _p(dzb), stages, _stream()), "ps_main_field_bwd")
    descs = []
    for sp, off, sh in ((spec.base, spec.g_off[0], shapes[:n_base]), (spec.sem, spec.g_off[1], shapes[n_base:n_base + n_sem]),
                        (spec.rgb, spec.g_off[2], shapes[n_base + n_sem:])):
        descs += sp.unpack_descs(gpart, off, sh)
    grads = unpack_layers(descs, npart.value, spec.g_total, dev, ctx.layer_sinks)  # all layers in one launch
    flat = flatten_grads(grads)
    assert len(flat) == 2 * len(shapes)
    return dapp, dfeat, flat


def _main_forward(ctx, table_needs_grad, u, sel, dirs, app, S, table, scalings, g: GridCfg, want_rgb, want_sem, n_base, n_sem, wb):
    """shared forward of the main-field nodes: encode (+ record counts for the table backward) -> fused MLP kernel.
    Stores everything the shared backward needs on ctx (tensors are returned for save_for_backward)."""
    table = _f32(table, "hash table")
    feat, counts = _encode(u, table, scalings, g, count=table_needs_grad)
    (dirs, app_c, packed, acts), outs = _stack_forward(ctx, table_needs_grad, feat, u.shape[0], sel, dirs, app, S, g, want_rgb, want_sem,
                                                       n_base, n_sem, wb)
    ctx.meta = (g, tuple(table.shape), want_rgb, want_sem)
    ctx.table_sink = grad_sink(table)
    ctx.table_ref = table
    ctx.direct = direct_params(table, *wb)
    return (u, sel, dirs, app_c, scalings, feat, packed, counts, acts), outs


def _dzb_scratch(n_points: int, dev):
    """workspace of the three-kernel main backward (ps_main_field_bwd dzb_scratch); PRESIGHT_MAIN_BWD_SPLIT=0 -> the fused kernel"""
    if os.environ.get("PRESIGHT_MAIN_BWD_SPLIT", "1") == "0":
        return None
    return torch.empty((n_points + 15) // 16 * 16, 80, device=dev)


def _bwd_stages(dzb):
    """stage masks of ps_main_field_bwd: one call normally; while bench.py's per-kernel timing is on, the three kernels of the
    split backward are launched by three calls, each inside its own HIP-event region (same kernels, same order, same stream)"""
    if dzb is None or not prof.enabled("main_bwd_sem_kernel"):
        yield 7
        return
    for mask, name in ((1, "main_bwd_sem_kernel"), (2, "main_bwd_rgb_kernel"), (4, "main_bwd_base_kernel")):
        with prof.region(name):
            yield mask


def _main_backward(ctx, saved, d_sigma, d_rgb, d_sem, weights):
    """shared backward: fused MLP backward -> table scatter -> weight-gradient reduction.
    -> (dapp, dtable | None, [dW0, db0, ...] with None for in-place gradients)"""
    u, sel, dirs, app, scalings, feat, packed, counts, acts = saved
    g, tshape, _, _ = ctx.meta
    dapp, dfeat, flat = _stack_backward(ctx, sel, dirs, app, feat, packed, acts, d_sigma, d_rgb, d_sem, weights)
    # the MLP gradients are complete (unpacked): their bucket may leave while the table backward runs; the table follows (in pieces
    # when its bucket is split, _scatter reports them one by one)
    mark_touched([t for t in ctx.direct if t is not ctx.table_ref])
    dtable = _scatter(u, dfeat, scalings, g, tshape, ctx.table_sink, counts, sink_owner=ctx.table_ref)
    mark_touched([t for t in ctx.direct if t is ctx.table_ref])
    return dapp, dtable, flat


class _MainStack(torch.autograd.Function):
    """The fused MLP stack of a main field on feature planes that come from somewhere else (the dynamic branch of the dual
    field, presight_amd/dynamic.py): the gradient w.r.t. the planes is returned instead of being scattered into a table."""

    @staticmethod
    def forward(ctx, feat, sel, dirs, app, S, g: GridCfg, want_rgb, want_sem, n_base, n_sem, *wb):
        feat = _f32(feat)
        N = sel.shape[0]
        (dirs, app_c, packed, acts), (sigma, rgb, sem) = _stack_forward(ctx, _training(feat.requires_grad), feat, N, sel, dirs, app, S, g, want_rgb,
                                                                        want_sem, n_base, n_sem, wb)
        ctx.save_for_backward(sel, dirs, app_c, feat, packed, acts)
        ctx.want = (want_rgb, want_sem)
        ctx.direct = direct_params(*wb)
        empty = torch.empty(0, device=feat.device)
        return sigma, (rgb if want_rgb else empty), (sem if want_sem else empty)

    @staticmethod
    def backward(ctx, dsigma, drgb, dsem):
        sel, dirs, app, feat, packed, acts = ctx.saved_tensors
        want_rgb, want_sem = ctx.want
        d_sigma = _f32(dsigma) if dsigma is not None else None
        d_rgb = _f32(drgb) if (want_rgb and drgb is not None) else None
        d_sem = _f32(dsem) if (want_sem and dsem is not None) else None
        dapp, dfeat, flat = _stack_backward(ctx, sel, dirs, app, feat, packed, acts, d_sigma, d_rgb, d_sem, None)
        mark_touched(ctx.direct)
        return (dfeat, None, None, dapp, None, None, None, None, None, None, *flat)


def main_stack(feat: Tensor, sel: Tensor, dirs: Optional[Tensor], app: Optional[Tensor], S: int, g: GridCfg, base, sem, rgb,
               want_rgb: bool = True, want_sem: bool = True):
    """-> (density [N], rgb [N,3], semantics [N,64]) of the main-field MLP stack on feature planes feat [L,N,F]"""
    flat = []
    for W, b in list(base) + list(sem) + list(rgb):
        flat += [W, b]
    return _apply(_MainStack, feat, sel, dirs, app, S, g, want_rgb, want_sem, len(base), len(sem), *flat)


class _MainField(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, sel, dirs, app, S, table, scalings, g: GridCfg, want_rgb, want_sem, n_base, n_sem, *wb):
        saved, (sigma, rgb, sem) = _main_forward(ctx, _training(ctx.needs_input_grad[5]), u, sel, dirs, app, S, table, scalings, g, want_rgb,
                                                 want_sem, n_base, n_sem, wb)
        ctx.save_for_backward(*saved)
        empty = torch.empty(0, device=u.device)
        return sigma, (rgb if want_rgb else empty), (sem if want_sem else empty)

    @staticmethod
    def backward(ctx, dsigma, drgb, dsem):
        want_rgb, want_sem = ctx.meta[-2:]
        d_sigma = _f32(dsigma) if dsigma is not None else None
        d_rgb = _f32(drgb) if (want_rgb and drgb is not None) else None
        d_sem = _f32(dsem) if (want_sem and dsem is not None) else None
        dapp, dtable, flat = _main_backward(ctx, ctx.saved_tensors, d_sigma, d_rgb, d_sem, None)
        return (None, None, None, dapp, None, dtable, None, None, None, None, None, None, *flat)


class _MainFieldRender(torch.autograd.Function):
    """main field + RaySamples.get_weights + the renderers as ONE autograd node (training, one sub-field, S <= 64).
    Versus the three separate nodes this never materialises the per-sample gradients d(rgb_s) [N,3] / d(sem_s) [N,64]
    (1.1 GB written by the composite backward and read again by the field backward at cfg 2): the composite backward only
    produces d(weights), the field backward forms w[n] * d(out)[ray] from the per-ray gradients in registers."""

    @staticmethod
    def forward(ctx, u, sel, dirs, app, S, ebins, threshold, table, scalings, g: GridCfg, n_base, n_sem, *wb):
        from . import ops

        saved, (sigma, rgb_s, sem_s) = _main_forward(ctx, _training(ctx.needs_input_grad[7]), u, sel, dirs, app, S, table, scalings, g, True, True,
                                                     n_base, n_sem, wb)
        R = ebins.shape[0]
        dev = u.device
        ebins = _f32(ebins)
        w = torch.empty(R, S, device=dev)
        check(lib().ps_weights_fwd(_p(ebins), _p(sigma), R, S, _p(w), _stream()), "ps_weights_fwd")
        rgb, sem = torch.empty(R, 3, device=dev), torch.empty(R, SEM_DIM, device=dev)
        acc, depth, expd = torch.empty(R, 1, device=dev), torch.empty(R, 1, device=dev), torch.empty(R, 1, device=dev)
        minmax = ops._minmax_init(dev).clone()
        check(lib().ps_composite_fwd(_p(w), _p(ebins), _p(rgb_s), _p(sem_s), R, S, SEM_DIM, threshold, _p(rgb), _p(acc), _p(depth),
                                     _p(expd), _p(sem), _p(minmax), _stream()), "ps_composite_fwd")
        ops._apply_minmax_hook(minmax)
        raw = torch.empty_like(expd)  # 1 where the batch-global clip left the value alone (its derivative)
        check(lib().ps_clip(_p(expd), R, _p(minmax), _p(raw), _stream()), "ps_clip")
        ctx.save_for_backward(*saved, ebins, sigma, w, rgb_s, sem_s, raw, expd)
        ctx.n_field_saved = len(saved)
        ctx.mark_non_differentiable(depth)
        return rgb, acc, depth, expd, sem, w

    @staticmethod
    def backward(ctx, d_rgb, d_acc, _d_depth, d_exp, d_sem, d_w_ext):
        t = ctx.saved_tensors
        saved, (ebins, sigma, w, rgb_s, sem_s, raw, expd) = t[:ctx.n_field_saved], t[ctx.n_field_saved:]
        R, S = w.shape
        d_rgb = _f32(d_rgb) if d_rgb is not None else None
        d_sem = _f32(d_sem) if d_sem is not None else None
        d_acc = _f32(d_acc) if d_acc is not None else None
        if d_exp is not None:
            d_exp = _f32(d_exp * raw)  # gradient of the batch-global clip
        dw = torch.empty_like(w)
        ext = _f32(d_w_ext).contiguous() if d_w_ext is not None else None  # losses that act on the weights directly (distortion, line of sight)
        check(lib().ps_composite_bwd(_p(w), _p(ebins), _p(rgb_s) if d_rgb is not None else None,
                                     _p(sem_s) if d_sem is not None else None, _p(d_rgb), _p(d_acc), _p(d_sem), _p(d_exp), R, S,
                                     SEM_DIM, _p(dw), None, None, _p(ext), None, _stream()), "ps_composite_bwd")
        dsig = torch.empty_like(sigma)
        check(lib().ps_weights_bwd(_p(ebins), _p(sigma), _p(dw), R, S, _p(dsig), _stream()), "ps_weights_bwd")
        dapp, dtable, flat = _main_backward(ctx, saved, dsig, d_rgb, d_sem, w)  # a missing head gradient skips that head
        return (None, None, None, dapp, None, None, None, dtable, None, None, None, None, *flat)


# ------------------------------------------------------------------------------------------------ factored semantic path
# The training render node of ONE sub-field with two algebraic rewrites of the semantic branch (csrc/field.hip MainCfg FACT,
# csrc/factored.hip): base layer 1 rows 16..79 merged with semantic layer 0, semantic output layer applied per ray after
# compositing, the direction / appearance columns of the colour head's first layer evaluated once per ray, and the rendering
# weights + the compositing of the semantic branch done inside the field kernel (no per-sample [N, 64] array exists).
# PRESIGHT_FACTORED=0 selects the unfactored node (same results to fp32 re-association).
FACTORED = os.environ.get("PRESIGHT_FACTORED", "1") != "0"


class MainSpecF:
    def __init__(self, LF: int, hidden: int, hidden_color: int, app_dim: int):
        self.base = MlpSpec([LF, hidden, 16])
        self.sem = MlpSpec([hidden, 64, SEM_DIM], first_colmap=chain_colmap(hidden // 4, hidden), ks0=hidden // 4)
        # the kernel's first colour layer reads base-output block 0 (sigma_raw | geo15) only: k-steps 4..7 of colour_colmap
        self.rgb = MlpSpec([16 + GEO_DIM + app_dim, hidden_color, hidden_color, 3], first_colmap=colour_colmap(app_dim)[16:32], ks0=4)
        self.p_off = [0, self.base.packed, self.base.packed + self.sem.packed]
        self.packed = self.p_off[2] + self.rgb.packed
        self.g_off = [0, self.base.g_total, self.base.g_total + self.sem.g_total]
        self.g_total = self.g_off[2] + self.rgb.g_total
        pf, gf, npart, aw, dw = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        offs = (ctypes.c_int64 * 6)()
        check(lib().ps_main_field_f_sizes(LF, hidden, hidden_color, 1, ctypes.byref(pf), ctypes.byref(gf), ctypes.byref(npart), offs,
                                          ctypes.byref(aw), ctypes.byref(dw)), "ps_main_field_f_sizes")
        assert pf.value == self.packed and gf.value == self.g_total and list(offs) == self.p_off + self.g_off, (pf.value, self.packed, list(offs))
        self.act_width, self.dzb_width = aw.value, dw.value


_MAIN_SPECS_F: dict = {}


def _main_spec_f(LF, hidden, hidden_color, app_dim) -> MainSpecF:
    key = (LF, hidden, hidden_color, app_dim)
    if key not in _MAIN_SPECS_F:
        _MAIN_SPECS_F[key] = MainSpecF(LF, hidden, hidden_color, app_dim)
    return _MAIN_SPECS_F[key]


def factored_supported(base, sem, rgb, S: int = 32) -> bool:
    """2-layer base MLP ending in 1 + 15 + 64 outputs, 3-layer 64-wide semantic head (the PreSight layout), and a sample count per
    ray that is a multiple of the kernels' 32-point tile (a ray is then a whole number of tiles of one wavefront)"""
    return (FACTORED and len(base) == 2 and len(sem) == 3 and len(rgb) == 3 and base[1][0].shape[0] == BASE_OUT
            and sem[0][0].shape == (64, SEM_DIM) and sem[2][0].shape == (SEM_DIM, 64) and base[0][0].shape[0] % 16 == 0
            and S > 0 and S % 32 == 0)


class _MainFieldRenderF(torch.autograd.Function):
    """_MainFieldRender on the factored semantic path (one sub-field, training, S <= 64)"""

    @staticmethod
    def forward(ctx, u, sel, dirs, app, S, ebins, threshold, want_depth, table, scalings, g: GridCfg, *wb):
        from . import ops

        ctx.set_materialize_grads(False)  # an output nobody differentiates arrives as None, not as a freshly filled zero tensor
        (Wb0, bb0), (Wb1, bb1), (Ws0, bs0), (Ws1, bs1), (Ws2, bs2), r0, r1, r2 = _layers(wb)
        hidden, hidden_color = Wb0.shape[0], r0[0].shape[0]
        A = r0[0].shape[1] - 16 - GEO_DIM
        if (0 if app is None else app.shape[1]) != A:
            raise ValueError(f"colour head expects SH16 + geo15 + app{A} inputs, got an appearance embedding of width "
                             f"{0 if app is None else app.shape[1]}")
        spec = _main_spec_f(g.out_dim, hidden, hidden_color, A)
        N, dev = u.shape[0], u.device
        R = ebins.shape[0]
        table = _f32(table, "hash table")
        train = _training(ctx.needs_input_grad[8])
        feat, counts = _encode(u, table, scalings, g, count=train)
        # merged first semantic layer: W' = W_sem0 W_base1[16:], b' = W_sem0 b_base1[16:] + b_sem0
        Wb1, bb1, Ws0, bs0 = _f32(Wb1), _f32(bb1), _f32(Ws0), _f32(bs0)
        Wm, bm = torch.empty(64, hidden, device=dev), torch.empty(64, device=dev)
        with prof.region("merge_linear_fwd"):
            check(lib().ps_merge_linear_fwd(_p(Ws0), _p(bs0), Wb1.data_ptr() + 4 * 16 * hidden, bb1.data_ptr() + 4 * 16, 64, SEM_DIM, hidden,
                                            _p(Wm), _p(bm), _stream()), "ps_merge_linear_fwd")
        packed = torch.empty(spec.packed, device=dev)
        descs = spec.base.pack_descs([(Wb0, bb0), (Wb1[:16], bb1[:16])], packed[: spec.base.packed])
        descs += spec.sem.pack_descs([(Wm, bm), (Ws1, bs1)], packed[spec.p_off[1]: spec.p_off[1] + spec.sem.packed])
        descs += spec.rgb.pack_descs([r0, r1, r2], packed[spec.p_off[2]:])
        pack_layers(descs)
        sigma, rgb_s = torch.empty(N, device=dev), torch.empty(N, 3, device=dev)
        w, hid = torch.empty(R, S, device=dev), torch.empty(R, SEM_DIM, device=dev)
        ebins = _f32(ebins)
        acts = torch.empty((N + 15) // 16 * 16, spec.act_width, device=dev)
        dirs = _f32(dirs)
        app_c = _f32(app) if app is not None else None
        Wr0 = _f32(r0[0])
        ray_colour = torch.empty(R, hidden_color, device=dev)
        with prof.region("ray_colour_fwd"):
            check(lib().ps_ray_colour_fwd(_p(dirs), _p(app_c), _p(Wr0), R, A, hidden_color, _p(ray_colour), _stream()), "ps_ray_colour_fwd")
        with prof.region("main_field_fwd"):
            check(lib().ps_main_field_f_fwd(_p(feat), N * g.features_per_level, g.out_dim, g.features_per_level, hidden, hidden_color, _p(sel),
                                            _p(ray_colour), _p(ebins), S, _p(packed), N, _p(sigma), _p(rgb_s), _p(w), _p(hid), _p(acts),
                                            _stream()), "ps_main_field_f_fwd")
        rgb, acc, sem = torch.empty(R, 3, device=dev), torch.empty(R, 1, device=dev), torch.empty(R, SEM_DIM, device=dev)
        Ws2, bs2 = _f32(Ws2), _f32(bs2)
        if want_depth:
            depth, expd = torch.empty(R, 1, device=dev), torch.empty(R, 1, device=dev)
            minmax = ops._minmax_init(dev).clone()
            check(lib().ps_composite_fwd(_p(w), _p(ebins), _p(rgb_s), None, R, S, SEM_DIM, threshold, _p(rgb), _p(acc), _p(depth), _p(expd),
                                         None, _p(minmax), _stream()), "ps_composite_fwd")
            with prof.region("sem_out_fwd"):
                check(lib().ps_sem_out_fwd(_p(hid), _p(acc), _p(Ws2), _p(bs2), R, SEM_DIM, _p(sem), _stream()), "ps_sem_out_fwd")
            ops._apply_minmax_hook(minmax)
            raw = torch.empty_like(expd)  # 1 where the batch-global clip left the value alone (its derivative)
            check(lib().ps_clip(_p(expd), R, _p(minmax), _p(raw), _stream()), "ps_clip")
        else:
            # the training step's call: nobody reads the depths in a step (the model hands them out as lazily evaluated entries that
            # go through ops.composite on the returned weights), so compositing is colour + accumulation and shares its launch with the
            # per-ray output layer of the semantic head -- no batch-wide min / max, no clip, no depth rows
            depth = expd = raw = torch.empty(0, device=dev)
            with prof.region("sem_out_fwd"):
                check(lib().ps_ray_out_fwd(_p(w), _p(rgb_s), _p(hid), _p(Ws2), _p(bs2), R, S, SEM_DIM, _p(rgb), _p(acc), _p(sem), _stream()),
                      "ps_ray_out_fwd")
        ctx.save_for_backward(u, sel, dirs, app_c, scalings, feat, packed, counts, acts, ebins, sigma, w, rgb_s, raw, expd, hid, acc, Wm, bm)
        ctx.meta = (g, hidden, hidden_color, A, S, tuple(table.shape), spec, bool(want_depth))
        ctx.params = wb
        ctx.table_sink = grad_sink(table)
        ctx.table_ref = table
        ctx.direct = direct_params(table, *wb)
        ctx.mark_non_differentiable(depth)
        if not want_depth:
            ctx.mark_non_differentiable(expd)
        return rgb, acc, depth, expd, sem, w

    @staticmethod
    def backward(ctx, d_rgb, d_acc, _d_depth, d_exp, d_sem, d_w_ext):
        (u, sel, dirs, app, scalings, feat, packed, counts, acts, ebins, sigma, w, rgb_s, raw, expd, hid, acc, Wm, bm) = ctx.saved_tensors
        g, hidden, hidden_color, A, S, tshape, spec, want_depth = ctx.meta
        if not want_depth:
            d_exp = None
        wb = ctx.params
        (Wb0, bb0), (Wb1, bb1), (Ws0, bs0), (Ws1, bs1), (Ws2, bs2), r0, r1, r2 = _layers(wb)
        R, dev, N = w.shape[0], w.device, u.shape[0]
        # gradient destinations: the parameters' in-place sinks where the owner opted in, fresh zero tensors (returned) otherwise
        dst, ret = [], []
        for t in wb:
            gs = grad_sink(t)
            if gs is not None:
                dst.append(gs)
                ret.append(None)
            else:
                z = torch.zeros_like(t, dtype=torch.float32)
                dst.append(z)
                ret.append(z)
        (dWb0, dbb0, dWb1, dbb1, dWs0, dbs0, dWs1, dbs1, dWs2, dbs2, dWr0, dbr0, dWr1, dbr1, dWr2, dbr2) = dst
        d_rgb = _f32(d_rgb) if d_rgb is not None else torch.zeros(R, 3, device=dev)
        d_sem = _f32(d_sem) if d_sem is not None else torch.zeros(R, SEM_DIM, device=dev)
        # output layer of the semantic head, per ray: v = W_out^T d(sem), d(acc) += <d(sem), b_out>, dW_out / db_out
        v, cray = torch.empty(R, SEM_DIM, device=dev), torch.empty(R, 1, device=dev)
        with prof.region("sem_out_bwd"):
            check(lib().ps_sem_out_bwd(_p(d_sem), _p(hid), _p(acc), _p(_f32(Ws2)), _p(_f32(bs2)), R, SEM_DIM, _p(v), _p(cray), _p(dWs2), _p(dbs2),
                                       _stream()), "ps_sem_out_bwd")
        fused_tail = d_exp is None  # (always in a training step) d(weights) is formed inside the weights' backward launch
        if not fused_tail:
            d_acc = cray if d_acc is None else _f32(d_acc) + cray
            d_exp = _f32(d_exp * raw)  # gradient of the batch-global clip
        pf, gf, npart = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        check(lib().ps_main_field_f_sizes(g.out_dim, hidden, hidden_color, N, ctypes.byref(pf), ctypes.byref(gf), ctypes.byref(npart), None,
                                          None, None), "ps_main_field_f_sizes")
        gpart = torch.empty(npart.value, spec.g_total, device=dev)
        dfeat = torch.empty_like(feat)
        dapp = torch.empty_like(app) if app is not None else None
        dzb = torch.empty((N + 15) // 16 * 16, spec.dzb_width, device=dev)
        dray = torch.empty(N // 16, hidden_color, device=dev)
        dw_sem = torch.empty(R, S, device=dev)

        def field_bwd(stages, dsig):
            check(lib().ps_main_field_f_bwd(_p(feat), N * g.features_per_level, g.out_dim, g.features_per_level, hidden, hidden_color,
                                            _p(sel), S, _p(packed), _p(dsig), _p(d_rgb), _p(v), _p(w), N, _p(dfeat), _p(dray), _p(dw_sem),
                                            _p(gpart), _p(acts), _p(dzb), stages, _stream()), "ps_main_field_f_bwd")

        # semantic kernel first: it needs the weights only and yields the semantic branch's part of d(weights)
        timed = prof.enabled("main_bwd_sem_kernel")
        with prof.region("main_field_bwd"):
            with prof.region("main_bwd_sem_kernel") if timed else contextlib.nullcontext():
                field_bwd(1, None)
        # (+ the semantic branch's part and the losses that act on the weights directly -- distortion, line of sight -- added in the kernel)
        ext = _f32(d_w_ext).contiguous() if d_w_ext is not None else None
        dsig = torch.empty_like(sigma)
        if fused_tail:
            check(lib().ps_ray_dsigma_bwd(_p(ebins), _p(sigma), _p(rgb_s), _p(d_rgb), _p(None if d_acc is None else _f32(d_acc)), _p(cray),
                                          _p(dw_sem), _p(ext), R, S, _p(dsig), _stream()), "ps_ray_dsigma_bwd")
        else:
            dw = torch.empty_like(w)
            check(lib().ps_composite_bwd(_p(w), _p(ebins), _p(rgb_s), None, _p(d_rgb), _p(d_acc), None, _p(d_exp), R, S, SEM_DIM, _p(dw), None,
                                         None, _p(dw_sem), _p(ext), _stream()), "ps_composite_bwd")
            check(lib().ps_weights_bwd(_p(ebins), _p(sigma), _p(dw), R, S, _p(dsig), _stream()), "ps_weights_bwd")
        with prof.region("main_field_bwd", extend=True):
            if timed:
                for mask, name in ((2, "main_bwd_rgb_kernel"), (4, "main_bwd_base_kernel")):
                    with prof.region(name):
                        field_bwd(mask, dsig)
            else:
                field_bwd(6, dsig)
        # direction / appearance columns of the colour head's first layer and d(appearance), per ray
        with prof.region("ray_colour_bwd"):
            check(lib().ps_ray_colour_bwd(_p(dray), _p(dirs), _p(app), _p(_f32(r0[0])), R, S, A, hidden_color, _p(dWr0), _p(dapp), _stream()),
                  "ps_ray_colour_bwd")
        # weight gradients: partial blocks -> torch layout.  Base layer 1: rows 0..15 (sigma_raw | geo15) directly, rows 16..79
        # through the merged layer (chain rule of W' = W_sem0 W_base1[16:], b' = W_sem0 b_base1[16:] + b_sem0)
        dWmb = torch.zeros(Wm.numel() + bm.numel(), device=dev)  # (one fill for both)
        dWm, dbm = dWmb[:Wm.numel()].view_as(Wm), dWmb[Wm.numel():]
        descs = spec.base.unpack_descs(gpart, spec.g_off[0], [tuple(Wb0.shape), (16, hidden)])
        descs += spec.sem.unpack_descs(gpart, spec.g_off[1], [(64, hidden), (64, 64)])
        descs += spec.rgb.unpack_descs(gpart, spec.g_off[2], [tuple(r0[0].shape), tuple(r1[0].shape), tuple(r2[0].shape)])
        sinks = [(dWb0, dbb0), (dWb1[:16], dbb1[:16]), (dWm, dbm), (dWs1, dbs1), (dWr0, dbr0), (dWr1, dbr1), (dWr2, dbr2)]
        unpack_layers(descs, npart.value, spec.g_total, dev, sinks)
        Wb1c, bb1c = _f32(Wb1), _f32(bb1)
        with prof.region("merge_linear_bwd"):
            check(lib().ps_merge_linear_bwd(_p(dWm), _p(dbm), _p(_f32(Ws0)), Wb1c.data_ptr() + 4 * 16 * hidden, bb1c.data_ptr() + 4 * 16, 64, SEM_DIM,
                                            hidden, _p(dWs0), _p(dbs0), dWb1.data_ptr() + 4 * 16 * hidden, dbb1.data_ptr() + 4 * 16, _stream()),
                  "ps_merge_linear_bwd")
        # the MLP gradients are complete: their bucket may leave while the table backward runs; then the table (in pieces when split)
        mark_touched([t for t in ctx.direct if t is not ctx.table_ref])
        dtable = _scatter(u, dfeat, scalings, g, tshape, ctx.table_sink, counts, sink_owner=ctx.table_ref)
        mark_touched([t for t in ctx.direct if t is ctx.table_ref])
        return (None, None, None, dapp, None, None, None, None, dtable, None, None, *ret)


def main_field_render(u: Tensor, sel: Tensor, dirs: Tensor, app: Optional[Tensor], ebins: Tensor, table: Tensor, scalings: Tensor,
                      g: GridCfg, base, sem, rgb, threshold: float = 0.5, want_depth: bool = True):
    """-> (rgb [R,3], accumulation [R,1] (unclamped), threshold depth [R,1], expected depth [R,1], semantics [R,64],
    weights [R,S]) of a ray batch whose S = ebins.shape[1]-1 <= 64 samples per ray are the points u (point n = ray n // S).
    want_depth=False (factored training node only; elsewhere ignored): the two depths come back as None -- render them from the
    returned weights with ops.composite when somebody asks (differentiable through the weights)."""
    S = ebins.shape[1] - 1
    if S > 64:
        raise NotImplementedError("main_field_render: at most 64 samples per ray (use main_field + ops.composite)")
    flat = []
    for W, b in list(base) + list(sem) + list(rgb):
        flat += [W, b]
    if factored_supported(base, sem, rgb, S) and torch.is_grad_enabled():
        rgb_, acc, depth, expd, sem_, w = _apply(_MainFieldRenderF, u, sel, dirs, app, S, ebins, float(threshold), bool(want_depth), table,
                                                 scalings, g, *flat)
        return (rgb_, acc, depth, expd, sem_, w) if want_depth else (rgb_, acc, None, None, sem_, w)
    return _apply(_MainFieldRender, u, sel, dirs, app, S, ebins, float(threshold), table, scalings, g, len(base), len(sem), *flat)


class MainSpecGated:
    """packed layout of the gated inference forward (csrc/field.hip MainCfg MERGE_): base ending in 16 outputs, three-layer semantic
    head whose first layer is the merged one and reads the base hidden layer, the colour head as in MainSpec (not evaluated)"""

    def __init__(self, LF: int, hidden: int, hidden_color: int, app_dim: int):
        self.base = MlpSpec([LF, hidden, 16])
        self.sem = MlpSpec([hidden, 64, 64, SEM_DIM], first_colmap=chain_colmap(hidden // 4, hidden), ks0=hidden // 4)
        self.rgb = MlpSpec([16 + GEO_DIM + app_dim, hidden_color, hidden_color, 3], first_colmap=colour_colmap(app_dim), ks0=12)
        self.p_off = [0, self.base.packed, self.base.packed + self.sem.packed]
        self.packed = self.p_off[2] + self.rgb.packed
        pf, offs = ctypes.c_int64(), (ctypes.c_int64 * 3)()
        check(lib().ps_main_field_gated_sizes(LF, hidden, hidden_color, ctypes.byref(pf), offs), "ps_main_field_gated_sizes")
        assert pf.value == self.packed and list(offs) == self.p_off, (pf.value, self.packed, list(offs), self.p_off)


_MAIN_SPECS_G: dict = {}
GATE_STATS: Optional[Tensor] = None  # int64 device tensor [2] while a caller (bench.py) counts executed / visited tiles of the gated query


def main_field_gated(u: Tensor, sel: Tensor, table: Tensor, scalings: Tensor, g: GridCfg, base, sem, rgb, gate_a: Tensor, gate_b: Tensor,
                     threshold: float):
    """Inference query of the prior extraction (no autograd): -> (density [N], semantics [N,64]) where the semantic head is only
    evaluated for the 32-point tiles in which some point has mean(gate_a, gate_b, density) >= threshold; the other rows of the
    semantics are UNINITIALISED (ns/scripts/extract_priors.py:133-150 drops those points).  gate_a / gate_b: the proposal fields'
    densities of the same points.  The kernel runs the network with the base output layer's rows 16..79 merged into the semantic
    head's first layer (rewrite 1): densities bit-identical to main_field, semantics equal to fp32 rounding."""
    (Wb0, bb0), (Wb1, bb1) = base
    (Ws0, bs0), (Ws1, bs1), (Ws2, bs2) = sem
    hidden, hidden_color = Wb0.shape[0], rgb[0][0].shape[0]
    A = rgb[0][0].shape[1] - 16 - GEO_DIM
    key = (g.out_dim, hidden, hidden_color, A)
    if key not in _MAIN_SPECS_G:
        _MAIN_SPECS_G[key] = MainSpecGated(*key)
    spec = _MAIN_SPECS_G[key]
    N, dev = u.shape[0], u.device
    with torch.no_grad():
        feat, _ = _encode(u, _f32(table, "hash table"), scalings, g, count=False)
        Wb1, bb1, Ws0, bs0 = _f32(Wb1), _f32(bb1), _f32(Ws0), _f32(bs0)
        Wm, bm = torch.empty(64, hidden, device=dev), torch.empty(64, device=dev)
        check(lib().ps_merge_linear_fwd(_p(Ws0), _p(bs0), Wb1.data_ptr() + 4 * 16 * hidden, bb1.data_ptr() + 4 * 16, 64, SEM_DIM, hidden,
                                        _p(Wm), _p(bm), _stream()), "ps_merge_linear_fwd")
        packed = torch.empty(spec.packed, device=dev)
        descs = spec.base.pack_descs([(Wb0, bb0), (Wb1[:16], bb1[:16])], packed[: spec.base.packed])
        descs += spec.sem.pack_descs([(Wm, bm), (Ws1, bs1), (Ws2, bs2)], packed[spec.p_off[1]: spec.p_off[1] + spec.sem.packed])
        descs += spec.rgb.pack_descs(list(rgb), packed[spec.p_off[2]:])
        pack_layers(descs)
        sigma, semantics = torch.empty(N, device=dev), torch.empty(N, SEM_DIM, device=dev)
        with prof.region("main_field_fwd"):
            check(lib().ps_main_field_fwd_gated(_p(feat), N * g.features_per_level, g.out_dim, g.features_per_level, hidden, hidden_color,
                                                _p(sel), _p(packed), N, _p(_f32(gate_a).reshape(-1)), _p(_f32(gate_b).reshape(-1)),
                                                float(threshold), _p(sigma), _p(semantics), _p(GATE_STATS), _stream()), "ps_main_field_fwd_gated")
    return sigma, semantics


def ms_main_field_gated(lay: "MsLayout", u: Tensor, sel: Tensor, tables: Sequence[Tensor], scalings: Tensor, g: GridCfg, base, sem, rgb,
                        gate_a: Tensor, gate_b: Tensor, threshold: float):
    """main_field_gated for the K routed sub-fields of a tile (no autograd): -> (density [N], semantics [N,64]) in the caller's point
    order; base / sem / rgb: per sub-field layer lists; gate_a / gate_b [N] in the caller's order."""
    K = lay.K
    layers = [list(base[k]) + list(sem[k]) + list(rgb[k]) for k in range(K)]
    hidden, hidden_color = base[0][0][0].shape[0], rgb[0][0][0].shape[0]
    A = rgb[0][0][0].shape[1] - 16 - GEO_DIM
    spec = _main_spec_m(g.out_dim, hidden, hidden_color, A)
    dev = u.device
    with torch.no_grad():
        feat, _ = _ms_encode(lay, u, [_f32(t, "hash table") for t in tables], scalings, g, count=False)
        mm = _MsMerged.get(layers)
        mm.merge()
        st = _MsStacks.get("main_m", [spec.base, spec.sem, spec.rgb], (spec.p_off, spec.g_off), spec.packed, spec.g_total, mm.klayers,
                           lib().ps_main_field_parts_ms(1 << 40, K))
        st.pack()
        sigma, semantics = torch.empty(lay.N, device=dev), torch.empty(lay.N, SEM_DIM, device=dev)
        with prof.region("main_field_fwd"):
            check(lib().ps_main_field_fwd_gated_ms(_p(feat), lay.n_slots * g.features_per_level, g.out_dim, g.features_per_level, hidden,
                                                   hidden_color, _p(sel), _p(st.packed), lay.n_slots, _p(_f32(gate_a).reshape(-1)),
                                                   _p(_f32(gate_b).reshape(-1)), float(threshold), _p(sigma), _p(semantics), _p(GATE_STATS),
                                                   _p(lay.perm), lay.field_start, K, _stream()), "ps_main_field_fwd_gated_ms")
    return sigma, semantics


def main_field(u: Tensor, sel: Tensor, dirs: Optional[Tensor], app: Optional[Tensor], S: int, table: Tensor, scalings: Tensor,
               g: GridCfg, base: Sequence[Tuple[Tensor, Tensor]], sem: Sequence[Tuple[Tensor, Tensor]],
               rgb: Sequence[Tuple[Tensor, Tensor]], want_rgb: bool = True, want_sem: bool = True):
    """-> (density [N], rgb [N,3], semantics [N,64]).  Point n belongs to ray n // S; dirs [R,3], app [R,A]."""
    flat = []
    for W, b in list(base) + list(sem) + list(rgb):
        flat += [W, b]
    return _apply(_MainField, u, sel, dirs, app, S, table, scalings, g, want_rgb, want_sem, len(base), len(sem), *flat)


# ================================================================================================ multi-sub-field ("MS") path
# All K sub-fields of a tile in ONE launch per kernel (csrc/ms_core.hpp): route + stable sort into the padded chunk layout on
# the device (no host sync), per-sub-field tables / AABBs / MLP weights selected inside the kernels.  Reference semantics:
# ns/fields/PreSight/ingp_field_ms.py:97-126, prop_density_field_ms.py:90-102.
class MsLayout:
    """Sorted, chunk-padded layout of one routed call (device arrays only)."""

    def __init__(self, centroids: Tensor, pos: Optional[Tensor] = None, origins: Optional[Tensor] = None, dirs: Optional[Tensor] = None,
                 ebins: Optional[Tensor] = None):
        c = _f32(centroids)
        self.K = int(c.shape[0])
        if pos is not None:
            pos = _f32(pos).view(-1, 3)
            self.N, self.S = pos.shape[0], 0
        else:
            origins, dirs, ebins = _f32(origins), _f32(dirs), _f32(ebins)
            self.S = ebins.shape[1] - 1
            self.N = ebins.shape[0] * self.S
        self.src = (pos, origins, dirs, ebins)
        dev = c.device
        out = (ctypes.c_int64 * 5)()
        check(lib().ps_ms_layout(self.N, self.K, out), "ps_ms_layout")
        self.plan = torch.empty(out[0], device=dev, dtype=torch.int32)
        self.field_start = self.plan.data_ptr() + 4 * out[1]
        self.chunk_field = self.plan.data_ptr() + 4 * out[2]
        self.n_slots, self.chunks = int(out[3]), int(out[4])
        self.perm = torch.empty(self.n_slots, device=dev, dtype=torch.int32)
        check(lib().ps_ms_route(_p(pos), _p(origins), _p(dirs), _p(ebins), self.S, self.N, _p(c), self.K, _p(self.plan), _p(self.perm),
                                _stream()), "ps_ms_route")

    def points(self, aabbs: Tensor, contract: bool) -> Tuple[Tensor, Tensor]:
        """-> (u [slots,3], sel [slots]) with every slot normalised by ITS sub-field's AABB"""
        pos, origins, dirs, ebins = self.src
        dev = self.perm.device
        u = torch.empty(self.n_slots, 3, device=dev)
        sel = torch.empty(self.n_slots, device=dev)
        check(lib().ps_ms_field_points(_p(pos), _p(origins), _p(dirs), _p(ebins), self.S, _p(_f32(aabbs)), int(contract), self.N, self.K,
                                       _p(self.plan), _p(self.perm), _p(u), _p(sel), _stream()), "ps_ms_field_points")
        return u, sel


_GROUP_TABLES: dict = {}


def mark_groups(lay: "MsLayout", reps: Sequence[Tensor]):
    """raise, ON THE DEVICE, the "received a gradient this step" flag of every sub-field of the routed layout that got points
    (presight_amd.dist.FlatGrads.define_groups / ps_ms_mark_groups); reps[k] = any parameter of sub-field k.  The optimizer
    kernel skips the sub-fields whose flag stays down, as torch.optim.Adam skips the parameters the reference's sub-field loop
    never touched (ns/fields/PreSight/ingp_field_ms.py:97-126)."""
    owner = getattr(reps[0], "_ps_group_owner", None)
    if owner is None:
        return
    gids = tuple(-1 if getattr(p, "_ps_group", None) is None else p._ps_group for p in reps)
    key = (str(reps[0].device), id(owner), gids)
    tbl = _GROUP_TABLES.get(key)
    if tbl is None:
        if len(_GROUP_TABLES) > 64:
            _GROUP_TABLES.clear()
        tbl = torch.tensor(gids, dtype=torch.int32, device=reps[0].device)
        _GROUP_TABLES[key] = tbl
    check(lib().ps_ms_mark_groups(lay.field_start, lay.K, _p(tbl), _p(owner.group_flags), _stream()), "ps_ms_mark_groups")


class _LayerDesc(ctypes.Structure):
    _fields_ = [("W", ctypes.c_void_p), ("b", ctypes.c_void_p), ("colmap", ctypes.c_void_p), ("dst0", ctypes.c_void_p),
                ("dst1", ctypes.c_void_p), ("out_dim", ctypes.c_int), ("in_dim", ctypes.c_int), ("KS", ctypes.c_int), ("NB", ctypes.c_int)]


def _to_device_bytes(arr, device) -> Tensor:
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


_PTR_CACHE: dict = {}


def _ptr_table(tensors: Sequence[Tensor]) -> Tensor:
    """device array of the tensors' addresses (int64), cached by address tuple (the parameters and their flat-buffer gradients
    sit at fixed addresses for the life of a trainer)"""
    key = (str(tensors[0].device), tuple(t.data_ptr() for t in tensors))
    t = _PTR_CACHE.get(key)
    if t is None:
        if len(_PTR_CACHE) > 256:
            _PTR_CACHE.clear()
        t = torch.tensor(list(key[1]), dtype=torch.int64, device=tensors[0].device)
        _PTR_CACHE[key] = t
    return t


class _MsStacks:
    """K copies of one fused MLP stack (e.g. [base | semantic head | colour head] of the main field): persistent packed-weight
    and partial-gradient buffers + the device descriptor tables of ps_mlp_pack_table / ps_mlp_unpack_table_ms."""

    _cache: dict = {}

    @classmethod
    def get(cls, kind, specs, offsets, packed_per_field, g_per_field, layers_per_field: List[List[Tuple[Tensor, Tensor]]], max_parts):
        key = (kind, tuple(t.data_ptr() for f in layers_per_field for W, b in f for t in (W, b)), max_parts)
        obj = cls._cache.get(key)
        if obj is None:
            if len(cls._cache) > 32:
                cls._cache.clear()
            obj = cls(specs, offsets, packed_per_field, g_per_field, layers_per_field, max_parts)
            cls._cache[key] = obj
        return obj

    def __init__(self, specs, offsets, packed_per_field, g_per_field, layers_per_field, max_parts):
        assert ctypes.sizeof(_LayerDesc) == lib().ps_mlp_layer_desc_bytes()
        self.K = len(layers_per_field)
        dev = layers_per_field[0][0][0].device
        self.specs, self.offsets = specs, offsets  # offsets = (p_off list, g_off list) of the stacks inside one field's block
        self.packed_per_field, self.g_per_field = packed_per_field, g_per_field
        self.packed = torch.empty(self.K * packed_per_field, device=dev)
        self.gpart = torch.empty(max_parts, g_per_field, device=dev)
        self.n_layers = sum(sp.nl for sp in specs)
        descs = []
        for k, layers in enumerate(layers_per_field):
            i = 0
            for sp, off in zip(specs, offsets[0]):
                blk = self.packed[k * packed_per_field + off: k * packed_per_field + off + sp.packed]
                descs += sp.pack_descs(layers[i:i + sp.nl], blk)
                i += sp.nl
        self._keep = descs  # the descriptors hold contiguous views of the parameters
        arr = (_LayerDesc * len(descs))(*[_LayerDesc(_p(d[0]), _p(d[1]), _p(d[4]), d[7], d[8], d[2], d[3], d[5], d[6]) for d in descs])
        self.pack_table = _to_device_bytes(arr, dev)
        self.pack_elems = max(d[6] * 16 + 2 * d[6] * ((d[5] + 3) // 4) * 256 for d in descs)
        self.shapes = [(d[2], d[3]) for d in descs[: self.n_layers]]
        self._unpack_cache: dict = {}

    def pack(self):
        check(lib().ps_mlp_pack_table(_p(self.pack_table), self.K * self.n_layers, self.pack_elems, _stream()), "ps_mlp_pack_table")

    def unpack(self, dsts: List[Tuple[Tensor, Tensor]], field_start: int, B: int, parts_per_block: int):
        """dsts: (dW, db) per (field, layer), accumulated into"""
        key = tuple(t.data_ptr() for d in dsts for t in d)
        ent = self._unpack_cache.get(key)
        if ent is None:
            if len(self._unpack_cache) > 4:
                self._unpack_cache.clear()
            recs = []
            for k in range(self.K):
                i = 0
                for sp, off in zip(self.specs, self.offsets[1]):
                    for d in sp.unpack_descs(self.gpart, off, self.shapes[i:i + sp.nl]):
                        dW, db = dsts[k * self.n_layers + i]
                        recs.append(_LayerDesc(d[0], None, _p(d[3]), _p(dW), _p(db), d[1], d[2], d[4], d[5]))
                        i += 1
            arr = (_LayerDesc * len(recs))(*recs)
            elems = max(r.NB * ((r.KS + 3) // 4) * 256 + r.NB * 16 for r in recs)
            ent = (_to_device_bytes(arr, self.gpart.device), elems)
            self._unpack_cache[key] = ent
        check(lib().ps_mlp_unpack_table_ms(_p(ent[0]), self.K * self.n_layers, self.n_layers, field_start, self.K, B, parts_per_block,
                                           self.g_per_field, ent[1], _stream()), "ps_mlp_unpack_table_ms")


def _ms_encode(lay: MsLayout, u: Tensor, tables: Sequence[Tensor], scalings: Tensor, g: GridCfg, count: bool):
    feat = torch.empty(g.num_levels, lay.n_slots, g.features_per_level, device=u.device)
    counts = None
    if count:
        counts = torch.empty(lay.K * g.num_levels * lib().ps_grid_scatter_slices(g.features_per_level, g.log2_hashmap_size),
                             device=u.device, dtype=torch.int32)
    with prof.region(f"grid_encode_L{g.num_levels}F{g.features_per_level}"):
        check(lib().ps_grid_encode_ms(_p(u), _p(_ptr_table(tables)), _p(scalings), g.num_levels, g.features_per_level,
                                      g.log2_hashmap_size, lay.n_slots, lay.n_slots * g.features_per_level, _p(feat), _p(counts), lay.K,
                                      lay.chunk_field, _stream()), "ps_grid_encode_ms")
    return feat, counts


def _ms_scatter_ws(lay: MsLayout, g: GridCfg, device, tables: Optional[Sequence[Tensor]] = None) -> Tensor:
    own = tables[0] if (tables and getattr(tables[0], "_ps_sparse", None) is not None) else None  # (record exchange: a workspace of the tables' own)
    return _workspace(lib().ps_grid_scatter_workspace_ms(g.num_levels, g.features_per_level, g.log2_hashmap_size, lay.n_slots, lay.K),
                      device, owner=own)


def _ms_scatter(lay: MsLayout, u, dfeat, scalings, g: GridCfg, tables: Sequence[Tensor], counts, ws, absmax_ready: bool, mark: bool = False):
    """table gradients of all sub-fields; -> list of returned gradients (None where the gradient went into the parameter's
    pre-allocated .grad in place).  mark: report the tables to the owner of the flat gradient buffer (mark_touched) from here --
    in sub-field GROUPS when the owner exchanges them group by group (tables[0]._ps_ms_parts, set by the trainer): one accumulate
    launch per group, every group's tables handed over as soon as its launch is enqueued."""
    sinks = [grad_sink(t) for t in tables]
    fresh = [None if s is not None else torch.zeros_like(t) for s, t in zip(sinks, tables)]
    dst = [s if s is not None else f for s, f in zip(sinks, fresh)]
    zero_dst = all(s is None or _sink_is_zero(t) for s, t in zip(sinks, tables))  # fresh zeros / untouched, cleared sinks
    L, F, l2t = g.num_levels, g.features_per_level, g.log2_hashmap_size
    K = lay.K
    groups = getattr(tables[0], "_ps_ms_parts", 1) if mark else 1
    _refuse_second_contribution(tables)
    fused = _fused_adam(list(tables), routed=True) if (zero_dst and groups == 1 and all(s is not None for s in sinks)) else None
    sp = getattr(tables[0], "_ps_sparse", None)
    if sp is not None and all(s is not None for s in sinks) and sp[0]._distributed():
        # record (sparse) exchange of the K tables: write the records (phase 1), hand them over; the slices' owners accumulate
        with prof.region(f"grid_scatter_L{L}F{F}"):
            ptrs = _ptr_table(dst)
            check(lib().ps_grid_scatter_binned_ms_part(_p(u), _p(dfeat), _p(scalings), L, F, l2t, lay.n_slots, lay.n_slots * F, _p(ptrs), K,
                                                       lay.chunk_field, _p(counts), int(absmax_ready), _p(ws), int(zero_dst), 1, 0, 0, _stream()),
                  "ps_grid_scatter_binned_ms_part")
            _hand_over_records(list(tables), ws, g, lay.n_slots, K, None, ptrs)
        if mark:
            mark_touched(direct_params(*tables), groups_on_device=True)
        return fresh
    with prof.region(f"grid_scatter_L{L}F{F}"):
        if fused is not None:
            check(lib().ps_grid_scatter_binned_ms_adam(_p(u), _p(dfeat), _p(scalings), L, F, l2t, lay.n_slots, lay.n_slots * F,
                                                       _p(_ptr_table(dst)), K, lay.chunk_field, _p(counts), int(absmax_ready), _p(ws), 3, 0, -1,
                                                       *fused, _stream()), "ps_grid_scatter_binned_ms_adam")
            if mark:
                mark_touched(direct_params(*tables), groups_on_device=True)
            return fresh
        if groups > 1 and K % groups == 0:
            per_field = lib().ps_grid_scatter_items(L, F, l2t, 1)
            args = (_p(u), _p(dfeat), _p(scalings), L, F, l2t, lay.n_slots, lay.n_slots * F, _p(_ptr_table(dst)), K, lay.chunk_field, _p(counts),
                    int(absmax_ready), _p(ws), int(zero_dst))
            check(lib().ps_grid_scatter_binned_ms_part(*args, 1, 0, 0, _stream()), "ps_grid_scatter_binned_ms_part")
            for gi in range(groups):
                k0, k1 = K * gi // groups, K * (gi + 1) // groups
                check(lib().ps_grid_scatter_binned_ms_part(*args, 2, k0 * per_field, k1 * per_field, _stream()), "ps_grid_scatter_binned_ms_part")
                mark_touched(direct_params(*tables[k0:k1]), groups_on_device=True)
            return fresh
        check(lib().ps_grid_scatter_binned_ms(_p(u), _p(dfeat), _p(scalings), L, F, l2t, lay.n_slots, lay.n_slots * F, _p(_ptr_table(dst)),
                                              K, lay.chunk_field, _p(counts), int(absmax_ready), _p(ws), int(zero_dst), _stream()),
              "ps_grid_scatter_binned_ms")
    if mark:
        mark_touched(direct_params(*tables), groups_on_device=True)
    return fresh


def _ms_layer_dsts(layers_flat: Sequence[Tensor]):
    """(dW, db) destinations per layer: the parameters' in-place gradient sinks, or fresh zero tensors that are returned"""
    dsts, returned = [], []
    for i in range(0, len(layers_flat), 2):
        W, b = layers_flat[i], layers_flat[i + 1]
        gw, gb = grad_sink(W), grad_sink(b)
        if gw is not None and gb is not None:
            dsts.append((gw, gb))
            returned += [None, None]
        else:
            dW, db = torch.zeros_like(W), torch.zeros_like(b)
            dsts.append((dW, db))
            returned += [dW, db]
    return dsts, returned


class _PropFieldMS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lay: MsLayout, u, sel, scalings, g: GridCfg, *params):
        K = lay.K
        tables = [_f32(t, "hash table") for t in params[:K]]
        wb = params[K:]
        per = len(wb) // K  # tensors per sub-field (W, b per layer)
        layers = [_layers(wb[k * per:(k + 1) * per]) for k in range(K)]
        hidden = layers[0][0][0].shape[0]
        spec = _prop_spec(g.out_dim, hidden)
        dev = u.device
        train = _training(any(ctx.needs_input_grad[5:5 + K]))
        feat, counts = _ms_encode(lay, u, tables, scalings, g, count=train)
        st = _MsStacks.get("prop", [spec], ([0], [0]), spec.packed, spec.g_total, layers, lib().ps_prop_field_parts_ms(1 << 40, K))
        st.pack()
        sigma = torch.empty(lay.N, device=dev)
        with prof.region("prop_field_fwd"):
            check(lib().ps_prop_field_fwd_ms(_p(feat), lay.n_slots * g.features_per_level, g.out_dim, g.features_per_level, hidden, _p(sel),
                                             _p(st.packed), lay.n_slots, _p(sigma), _p(lay.perm), lay.field_start, K, _stream()),
                  "ps_prop_field_fwd_ms")
        ctx.save_for_backward(u, sel, scalings, feat, counts)
        ctx.meta = (lay, g, hidden, st, tables, wb)
        ctx.direct = direct_params(*params)
        return sigma

    @staticmethod
    def backward(ctx, dsigma):
        u, sel, scalings, feat, counts = ctx.saved_tensors
        lay, g, hidden, st, tables, wb = ctx.meta
        K = lay.K
        dfeat = torch.empty_like(feat)
        ws = _ms_scatter_ws(lay, g, u.device, tables)
        nparts = lib().ps_prop_field_parts_ms(lay.n_slots, K)
        with prof.region("prop_field_bwd"):
            check(lib().ps_prop_field_bwd_ms(_p(feat), lay.n_slots * g.features_per_level, g.out_dim, g.features_per_level, hidden, _p(sel),
                                             _p(st.packed), _p(_f32(dsigma)), lay.n_slots, _p(dfeat), _p(st.gpart), _p(ws), _p(lay.perm),
                                             lay.field_start, K, _stream()), "ps_prop_field_bwd_ms")
        mark_groups(lay, tables)  # (before the table backward: with the tables' Adam step fused into it, it reads the flags)
        dtables = _ms_scatter(lay, u, dfeat, scalings, g, tables, counts, ws, absmax_ready=True)
        dsts, returned = _ms_layer_dsts(wb)
        st.unpack(dsts, lay.field_start, nparts // 4, 4)
        mark_touched(ctx.direct, groups_on_device=True)
        return (None, None, None, None, None, *dtables, *returned)


def ms_prop_field(lay: MsLayout, u: Tensor, sel: Tensor, tables: Sequence[Tensor], scalings: Tensor, g: GridCfg,
                  layers_per_field: Sequence[Sequence[Tuple[Tensor, Tensor]]]) -> Tensor:
    """density [N] (caller's point order) of K proposal sub-fields in one launch per kernel"""
    flat = []
    for layers in layers_per_field:
        for W, b in layers:
            flat += [W, b]
    return _apply(_PropFieldMS, lay, u, sel, scalings, g, *tables, *flat)


# ---- routed tiles on the MERGED network (DESIGN.md 4.5 rewrite 1 per sub-field; csrc/field.hip MainCfg MERGE_): base output rows 16..79
# folded into the semantic head's first layer, W' = W_sem0 W_base1[16:], b' = W_sem0 b_base1[16:] + b_sem0.  Training (three-kernel
# backward on kept activations) and the gated inference query of the prior extraction.  PRESIGHT_MERGED_MS=0: the unmerged kernels.
MERGED_MS = os.environ.get("PRESIGHT_MERGED_MS", "1") != "0"


class MainSpecM:
    def __init__(self, LF: int, hidden: int, hidden_color: int, app_dim: int):
        self.base = MlpSpec([LF, hidden, 16])
        self.sem = MlpSpec([hidden, 64, 64, SEM_DIM], first_colmap=chain_colmap(hidden // 4, hidden), ks0=hidden // 4)
        self.rgb = MlpSpec([16 + GEO_DIM + app_dim, hidden_color, hidden_color, 3], first_colmap=colour_colmap(app_dim), ks0=12)
        self.p_off = [0, self.base.packed, self.base.packed + self.sem.packed]
        self.packed = self.p_off[2] + self.rgb.packed
        self.g_off = [0, self.base.g_total, self.base.g_total + self.sem.g_total]
        self.g_total = self.g_off[2] + self.rgb.g_total
        pf, gf, aw, dw = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int(), ctypes.c_int()
        offs = (ctypes.c_int64 * 6)()
        check(lib().ps_main_field_m_sizes(LF, hidden, hidden_color, ctypes.byref(pf), ctypes.byref(gf), offs, ctypes.byref(aw), ctypes.byref(dw)),
              "ps_main_field_m_sizes")
        assert pf.value == self.packed and gf.value == self.g_total and list(offs) == self.p_off + self.g_off, (pf.value, self.packed, list(offs))
        self.act_width, self.dzb_width = aw.value, dw.value


_MAIN_SPECS_M: dict = {}


def _main_spec_m(LF, hidden, hidden_color, app_dim) -> MainSpecM:
    key = (LF, hidden, hidden_color, app_dim)
    if key not in _MAIN_SPECS_M:
        _MAIN_SPECS_M[key] = MainSpecM(*key)
    return _MAIN_SPECS_M[key]


def merged_supported(base, sem, rgb) -> bool:
    """the PreSight layout: 2-layer base MLP ending in 1 + 15 + 64 outputs, 3-layer 64-wide semantic head, 3-layer colour head"""
    return (len(base) == 2 and len(sem) == 3 and len(rgb) == 3 and base[1][0].shape[0] == BASE_OUT and sem[0][0].shape == (64, SEM_DIM)
            and sem[2][0].shape[0] == SEM_DIM and base[0][0].shape[0] % 16 == 0
            and all(t.dtype == torch.float32 and t.is_contiguous() for W, b in list(base) + list(sem) for t in (W, b)))


class _MsMerged:
    """K merged first semantic layers of a routed tile: persistent W' / b' (and d(W') / d(b')) buffers at fixed addresses, the device
    pointer table of ps_merge_linear_fwd_batch, and the kernel-side layer lists [base0, base1[:16], merged, sem1, sem2, rgb0..2]."""

    _cache: dict = {}

    @classmethod
    def get(cls, layers: List[List[Tuple[Tensor, Tensor]]]):
        key = tuple(t.data_ptr() for f in layers for W, b in f for t in (W, b))
        obj = cls._cache.get(key)
        if obj is None:
            if len(cls._cache) > 8:
                cls._cache.clear()
            obj = cls(layers)
            cls._cache[key] = obj
        return obj

    def __init__(self, layers):
        self.K = len(layers)
        hidden = layers[0][0][0].shape[0]
        dev = layers[0][0][0].device
        self.hidden = hidden
        self.Wm, self.bm = torch.empty(self.K, 64, hidden, device=dev), torch.empty(self.K, 64, device=dev)
        self.dWm, self.dbm = torch.zeros_like(self.Wm), torch.zeros_like(self.bm)
        rows, self.klayers = [], []
        for k, f in enumerate(layers):
            (Wb0, bb0), (Wb1, bb1), (Ws0, bs0), (Ws1, bs1), (Ws2, bs2) = f[:5]
            rows += [Ws0.data_ptr(), bs0.data_ptr(), Wb1.data_ptr() + 4 * 16 * hidden, bb1.data_ptr() + 4 * 16]
            self.klayers.append([(Wb0, bb0), (Wb1[:16], bb1[:16]), (self.Wm[k], self.bm[k]), (Ws1, bs1), (Ws2, bs2)] + list(f[5:]))
        self.fwd_ptrs = torch.tensor(rows, dtype=torch.int64, device=dev)
        self._bwd_cache: dict = {}

    def merge(self):
        check(lib().ps_merge_linear_fwd_batch(_p(self.fwd_ptrs), self.K, 64, SEM_DIM, self.hidden, _p(self.Wm), _p(self.bm), _stream()),
              "ps_merge_linear_fwd_batch")

    def kernel_dsts(self, dsts: List[Tuple[Tensor, Tensor]], per: int):
        """dsts: (dW, db) of every ORIGINAL layer (field-major) -> destinations of the kernels' layers (base layer 1: its first 16
        rows; merged layer: this object's zeroed d(W') / d(b')) + the pointer table of ps_merge_linear_bwd_batch"""
        self.dWm.zero_()
        self.dbm.zero_()
        out = []
        for k in range(self.K):
            d = dsts[k * per:(k + 1) * per]
            out += [d[0], (d[1][0][:16], d[1][1][:16]), (self.dWm[k], self.dbm[k])] + list(d[3:])
        return out

    def merge_backward(self, layers, dsts, per: int):
        key = tuple(t.data_ptr() for d in dsts for t in d)
        tab = self._bwd_cache.get(key)
        if tab is None:
            if len(self._bwd_cache) > 4:
                self._bwd_cache.clear()
            rows = []
            h = self.hidden
            for k, f in enumerate(layers):
                (Wb1, bb1), (Ws0, _) = f[1], f[2]
                (dWb1, dbb1), (dWs0, dbs0) = dsts[k * per + 1], dsts[k * per + 2]
                rows += [Ws0.data_ptr(), Wb1.data_ptr() + 4 * 16 * h, bb1.data_ptr() + 4 * 16, dWs0.data_ptr(), dbs0.data_ptr(),
                         dWb1.data_ptr() + 4 * 16 * h, dbb1.data_ptr() + 4 * 16]
            tab = torch.tensor(rows, dtype=torch.int64, device=self.Wm.device)
            self._bwd_cache[key] = tab
        check(lib().ps_merge_linear_bwd_batch(_p(tab), self.K, _p(self.dWm), _p(self.dbm), 64, SEM_DIM, self.hidden, _stream()),
              "ps_merge_linear_bwd_batch")


def _ms_main_forward(ctx, train, lay: MsLayout, u, sel, dirs, app, S, scalings, g: GridCfg, want_rgb, want_sem, n_base, n_sem, params):
    K = lay.K
    tables = [_f32(t, "hash table") for t in params[:K]]
    wb = params[K:]
    per = len(wb) // K
    layers = [_layers(wb[k * per:(k + 1) * per]) for k in range(K)]
    base, rgb_l = layers[0][:n_base], layers[0][n_base + n_sem:]
    hidden, hidden_color = base[0][0].shape[0], rgb_l[0][0].shape[0]
    A = rgb_l[0][0].shape[1] - 16 - GEO_DIM
    if want_rgb and (0 if app is None else app.shape[1]) != A:
        raise ValueError(f"colour head expects SH16 + geo15 + app{A} inputs, got an appearance embedding of width "
                         f"{0 if app is None else app.shape[1]}")
    # the merged network (rewrite 1): full evaluations -- training with kept activations, or inference
    keep = KEEP_ACTIVATIONS and train and want_rgb and want_sem and lay.N > 0
    merged = (MERGED_MS and want_sem and (keep or not train)
              and merged_supported(layers[0][:n_base], layers[0][n_base:n_base + n_sem], layers[0][n_base + n_sem:])
              and (not keep or os.environ.get("PRESIGHT_MAIN_BWD_SPLIT", "1") != "0"))
    dev = u.device
    feat, counts = _ms_encode(lay, u, tables, scalings, g, count=train)
    mm = None
    if merged:
        spec = _main_spec_m(g.out_dim, hidden, hidden_color, A)
        mm = _MsMerged.get(layers)
        mm.merge()
        st = _MsStacks.get("main_m", [spec.base, spec.sem, spec.rgb], (spec.p_off, spec.g_off), spec.packed, spec.g_total, mm.klayers,
                           lib().ps_main_field_parts_ms(1 << 40, K))
    else:
        spec = _main_spec(g.out_dim, hidden, hidden_color, A)
        st = _MsStacks.get("main", [spec.base, spec.sem, spec.rgb], (spec.p_off, spec.g_off), spec.packed, spec.g_total, layers,
                           lib().ps_main_field_parts_ms(1 << 40, K))
    st.pack()
    N = lay.N
    sigma = torch.empty(N, device=dev)
    rgb = torch.empty(N, 3, device=dev) if want_rgb else None
    sem = torch.empty(N, SEM_DIM, device=dev) if want_sem else None
    dirs = _f32(dirs) if dirs is not None else torch.zeros(1, 3, device=dev)
    app_c = _f32(app) if (app is not None and want_rgb) else None
    acts = None
    if keep:
        acts = torch.empty(lay.n_slots, spec.act_width if merged else lib().ps_main_field_act_width(g.out_dim, hidden, hidden_color), device=dev)
    fwd = lib().ps_main_field_m_fwd_ms if merged else lib().ps_main_field_fwd_ms
    with prof.region("main_field_fwd"):
        check(fwd(_p(feat), lay.n_slots * g.features_per_level, g.out_dim, g.features_per_level, hidden, hidden_color,
                  _p(sel), _p(dirs), _p(app_c), max(S, 1), A, _p(st.packed), lay.n_slots, _p(sigma), _p(rgb), _p(sem),
                  _p(acts), _p(lay.perm), lay.field_start, K, _stream()), "ps_main_field_fwd_ms")
    ctx.meta = (lay, g, hidden, hidden_color, A, S, st, tables, wb, want_rgb, want_sem)
    ctx.merged = (mm, spec, layers) if merged else None
    ctx.direct = direct_params(*params)
    return (u, sel, dirs, app_c, scalings, feat, counts, acts), (sigma, rgb, sem)


def _ms_main_backward(ctx, saved, d_sigma, d_rgb, d_sem, weights):
    u, sel, dirs, app, scalings, feat, counts, acts = saved
    lay, g, hidden, hidden_color, A, S, st, tables, wb, want_rgb, want_sem = ctx.meta
    K = lay.K
    if d_rgb is None or d_sem is None:
        acts = None
    dfeat = torch.empty_like(feat)
    nparts = lib().ps_main_field_parts_ms(lay.n_slots, K)
    merged = getattr(ctx, "merged", None)
    if merged is not None and acts is None:
        raise RuntimeError("presight_amd: the merged routed network needs both head gradients in its backward")
    if merged is not None:
        dzb = torch.empty(lay.n_slots, merged[1].dzb_width, device=u.device)
    else:
        dzb = _dzb_scratch(lay.n_slots, u.device) if acts is not None else None
    # three-kernel backward: d(appearance) per point, summed over the samples of a ray below (no atomics)
    per_point = app is not None and dzb is not None and S > 0 and lay.N == app.shape[0] * S
    dapp_pt = torch.empty(lay.N, A, device=u.device) if per_point else None
    dapp = None if (app is None or per_point) else torch.zeros_like(app)
    bwd = lib().ps_main_field_m_bwd_ms if merged is not None else lib().ps_main_field_bwd_ms
    with prof.region("main_field_bwd"):
        for stages in _bwd_stages(dzb):
            check(bwd(_p(feat), lay.n_slots * g.features_per_level, g.out_dim, g.features_per_level, hidden, hidden_color,
                      _p(sel), _p(dirs), _p(app), max(S, 1), A, _p(st.packed), _p(d_sigma), _p(d_rgb), _p(d_sem),
                      _p(weights), lay.n_slots, _p(dfeat), _p(dapp), _p(st.gpart), _p(acts), _p(dzb), _p(dapp_pt), _p(lay.perm),
                      lay.field_start, K, stages, _stream()), "ps_main_field_bwd_ms")
    if per_point:
        dapp = dapp_pt.view(app.shape[0], S, A).sum(1)
    dsts, returned = _ms_layer_dsts(wb)
    if merged is not None:
        # kernel layers -> [base0, base1 rows 0..15, merged layer, sem1, sem2, colour head]; the merged layer's gradient is carried
        # back to semantic layer 0 and base layer 1 rows 16..79 of every sub-field by one launch (chain rule of W' = W_sem0 W_base1[16:])
        mm, _, layers = merged
        per = len(wb) // K // 2
        st.unpack(mm.kernel_dsts(dsts, per), lay.field_start, nparts, 1)
        mm.merge_backward(layers, dsts, per)
    else:
        st.unpack(dsts, lay.field_start, nparts, 1)
    # the MLP gradients are complete: their bucket may leave while the table backward runs; the tables follow (group by group)
    mark_touched(direct_params(*wb), groups_on_device=True)
    ws = _ms_scatter_ws(lay, g, u.device, tables)
    mark_groups(lay, tables)  # (before the table backward: with the tables' Adam step fused into it, it reads the flags)
    dtables = _ms_scatter(lay, u, dfeat, scalings, g, tables, counts, ws, absmax_ready=False, mark=True)
    return dapp, dtables, returned


class _MainFieldMS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lay: MsLayout, u, sel, dirs, app, S, scalings, g: GridCfg, want_rgb, want_sem, n_base, n_sem, *params):
        train = _training(any(ctx.needs_input_grad[12:12 + lay.K]))
        saved, (sigma, rgb, sem) = _ms_main_forward(ctx, train, lay, u, sel, dirs, app, S, scalings, g, want_rgb, want_sem, n_base, n_sem,
                                                    params)
        ctx.save_for_backward(*saved)
        empty = torch.empty(0, device=u.device)
        return sigma, (rgb if want_rgb else empty), (sem if want_sem else empty)

    @staticmethod
    def backward(ctx, dsigma, drgb, dsem):
        want_rgb, want_sem = ctx.meta[-2:]
        d_sigma = _f32(dsigma) if dsigma is not None else None
        d_rgb = _f32(drgb) if (want_rgb and drgb is not None) else None
        d_sem = _f32(dsem) if (want_sem and dsem is not None) else None
        dapp, dtables, returned = _ms_main_backward(ctx, ctx.saved_tensors, d_sigma, d_rgb, d_sem, None)
        return (None, None, None, None, dapp, None, None, None, None, None, None, None, *dtables, *returned)


class _MainFieldRenderMS(torch.autograd.Function):
    """_MainFieldRender for K sub-fields: the field kernels write densities / colours / semantics straight into the caller's
    (ray-major) order through the layout's permutation, so get_weights and the renderers work on them unchanged."""

    @staticmethod
    def forward(ctx, lay: MsLayout, u, sel, dirs, app, S, ebins, threshold, scalings, g: GridCfg, n_base, n_sem, *params):
        from . import ops

        train = _training(any(ctx.needs_input_grad[12:12 + lay.K]))
        saved, (sigma, rgb_s, sem_s) = _ms_main_forward(ctx, train, lay, u, sel, dirs, app, S, scalings, g, True, True, n_base, n_sem, params)
        R = ebins.shape[0]
        dev = u.device
        ebins = _f32(ebins)
        w = torch.empty(R, S, device=dev)
        check(lib().ps_weights_fwd(_p(ebins), _p(sigma), R, S, _p(w), _stream()), "ps_weights_fwd")
        rgb, sem = torch.empty(R, 3, device=dev), torch.empty(R, SEM_DIM, device=dev)
        acc, depth, expd = torch.empty(R, 1, device=dev), torch.empty(R, 1, device=dev), torch.empty(R, 1, device=dev)
        minmax = ops._minmax_init(dev).clone()
        check(lib().ps_composite_fwd(_p(w), _p(ebins), _p(rgb_s), _p(sem_s), R, S, SEM_DIM, threshold, _p(rgb), _p(acc), _p(depth),
                                     _p(expd), _p(sem), _p(minmax), _stream()), "ps_composite_fwd")
        ops._apply_minmax_hook(minmax)
        raw = torch.empty_like(expd)  # 1 where the batch-global clip left the value alone (its derivative)
        check(lib().ps_clip(_p(expd), R, _p(minmax), _p(raw), _stream()), "ps_clip")
        ctx.save_for_backward(*saved, ebins, sigma, w, rgb_s, sem_s, raw, expd)
        ctx.n_field_saved = len(saved)
        ctx.mark_non_differentiable(depth)
        return rgb, acc, depth, expd, sem, w

    @staticmethod
    def backward(ctx, d_rgb, d_acc, _d_depth, d_exp, d_sem, d_w_ext):
        t = ctx.saved_tensors
        saved, (ebins, sigma, w, rgb_s, sem_s, raw, expd) = t[:ctx.n_field_saved], t[ctx.n_field_saved:]
        R, S = w.shape
        d_rgb = _f32(d_rgb) if d_rgb is not None else None
        d_sem = _f32(d_sem) if d_sem is not None else None
        d_acc = _f32(d_acc) if d_acc is not None else None
        if d_exp is not None:
            d_exp = _f32(d_exp * raw)  # gradient of the batch-global clip
        dw = torch.empty_like(w)
        ext = _f32(d_w_ext).contiguous() if d_w_ext is not None else None  # losses that act on the weights directly (distortion, line of sight)
        check(lib().ps_composite_bwd(_p(w), _p(ebins), _p(rgb_s) if d_rgb is not None else None,
                                     _p(sem_s) if d_sem is not None else None, _p(d_rgb), _p(d_acc), _p(d_sem), _p(d_exp), R, S,
                                     SEM_DIM, _p(dw), None, None, _p(ext), None, _stream()), "ps_composite_bwd")
        dsig = torch.empty_like(sigma)
        check(lib().ps_weights_bwd(_p(ebins), _p(sigma), _p(dw), R, S, _p(dsig), _stream()), "ps_weights_bwd")
        dapp, dtables, returned = _ms_main_backward(ctx, saved, dsig, d_rgb, d_sem, w)
        return (None, None, None, None, dapp, None, None, None, None, None, None, None, *dtables, *returned)


def _ms_params(tables, base, sem, rgb):
    flat = []
    for k in range(len(tables)):
        for W, b in list(base[k]) + list(sem[k]) + list(rgb[k]):
            flat += [W, b]
    return flat


def ms_main_field(lay: MsLayout, u, sel, dirs, app, S, tables, scalings, g: GridCfg, base, sem, rgb, want_rgb=True, want_sem=True):
    """-> (density [N], rgb [N,3], semantics [N,64]) in the caller's point order; base/sem/rgb: per sub-field layer lists.
    Point n belongs to ray n // S (dirs [R,3], app [R,A])."""
    return _apply(_MainFieldMS, lay, u, sel, dirs, app, S, scalings, g, want_rgb, want_sem, len(base[0]), len(sem[0]), *tables,
                              *_ms_params(tables, base, sem, rgb))


def ms_main_field_render(lay: MsLayout, u, sel, dirs, app, ebins, tables, scalings, g: GridCfg, base, sem, rgb, threshold: float = 0.5):
    """main_field_render for K sub-fields"""
    S = ebins.shape[1] - 1
    if S > 64:
        raise NotImplementedError("ms_main_field_render: at most 64 samples per ray")
    return _apply(_MainFieldRenderMS, lay, u, sel, dirs, app, S, ebins, float(threshold), scalings, g, len(base[0]), len(sem[0]), *tables,
                                    *_ms_params(tables, base, sem, rgb))


# ================================================================================================ fused sky field
class SkySpec:
    def __init__(self, A: int):
        self.A = A
        self.rgb = MlpSpec([16 + A, 32, 32, 3])
        self.sem = MlpSpec([16, 32, 32, SEM_DIM])
        pf, gf, npart = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        offs = (ctypes.c_int64 * 4)()
        check(lib().ps_sky_field_sizes(A, 1, 1, 0, ctypes.byref(pf), ctypes.byref(gf), ctypes.byref(npart), offs), "ps_sky_field_sizes")
        self.p_off, self.g_off = [int(offs[0]), int(offs[1])], [int(offs[2]), int(offs[3])]
        self.packed, self.g_total = int(pf.value), int(gf.value)
        assert self.packed == self.rgb.packed + self.sem.packed and self.g_total == self.rgb.g_total + self.sem.g_total


_SKY_SPECS: dict = {}


def _sky_spec(A: int) -> SkySpec:
    if A not in _SKY_SPECS:
        _SKY_SPECS[A] = SkySpec(A)
    return _SKY_SPECS[A]


def sky_supported(app_dim: int, width: int, num_layers: int, semantic_dim: int) -> bool:
    return bool(lib().ps_sky_field_supported(app_dim, width, num_layers, semantic_dim))


class _SkyField(torch.autograd.Function):
    """colour + semantic head of the sky model for a batch of rays: one kernel per direction; `lay` (an MsLayout over the ray
    origins) selects the routed K-sub-field variant.  params = per sub-field [rgb W0 b0 W1 b1 W2 b2 | sem W0 ... b2]."""

    @staticmethod
    def forward(ctx, lay: Optional[MsLayout], dirs, app, *params):
        K = 1 if lay is None else lay.K
        dirs = _f32(dirs)
        R, dev = dirs.shape[0], dirs.device
        app_c = _f32(app) if app is not None else None
        A = 0 if app_c is None else app_c.shape[1]
        spec = _sky_spec(A)
        per = len(params) // K
        layers = [_layers(params[k * per:(k + 1) * per]) for k in range(K)]
        max_parts = ctypes.c_int()
        pf, gf = ctypes.c_int64(), ctypes.c_int64()
        check(lib().ps_sky_field_sizes(A, 1 << 40, K, int(lay is not None), ctypes.byref(pf), ctypes.byref(gf), ctypes.byref(max_parts), None),
              "ps_sky_field_sizes")
        st = _MsStacks.get("sky", [spec.rgb, spec.sem], (spec.p_off, spec.g_off), spec.packed, spec.g_total, layers, max_parts.value)
        st.pack()
        rgb, sem = torch.empty(R, 3, device=dev), torch.empty(R, SEM_DIM, device=dev)
        N = R if lay is None else lay.n_slots
        with prof.region("sky_field_fwd"):
            check(lib().ps_sky_field_fwd(_p(dirs), _p(app_c), A, _p(st.packed), N, _p(rgb), _p(sem), _p(lay.perm) if lay else None,
                                         lay.field_start if lay else None, K, _stream()), "ps_sky_field_fwd")
        ctx.save_for_backward(dirs, app_c)
        ctx.meta = (lay, A, st, params, N)
        ctx.direct = direct_params(*params)
        return rgb, sem

    @staticmethod
    def backward(ctx, drgb, dsem):
        dirs, app = ctx.saved_tensors
        lay, A, st, params, N = ctx.meta
        K = 1 if lay is None else lay.K
        drgb = _f32(drgb) if drgb is not None else None
        dsem = _f32(dsem) if dsem is not None else None
        dapp = torch.empty_like(app) if app is not None else None
        npart = ctypes.c_int()
        pf, gf = ctypes.c_int64(), ctypes.c_int64()
        check(lib().ps_sky_field_sizes(A, N, K, int(lay is not None), ctypes.byref(pf), ctypes.byref(gf), ctypes.byref(npart), None),
              "ps_sky_field_sizes")
        with prof.region("sky_field_bwd"):
            check(lib().ps_sky_field_bwd(_p(dirs), _p(app), A, _p(st.packed), _p(drgb), _p(dsem), N, _p(dapp), _p(st.gpart),
                                         _p(lay.perm) if lay else None, lay.field_start if lay else None, K, _stream()), "ps_sky_field_bwd")
        dsts, returned = _ms_layer_dsts(params)
        st.unpack(dsts, lay.field_start if lay else _single_field_start(dirs.device, N), npart.value, 1)
        mark_touched(ctx.direct, groups_on_device=lay is not None)
        if lay is not None:
            per = len(params) // K
            mark_groups(lay, [params[k * per] for k in range(K)])
        return (None, None, dapp, *returned)


_SINGLE_FS: dict = {}


def _single_field_start(device, N: int) -> int:
    """field_start of a one-group layout ({0, chunks}) for the table-driven gradient unpack of single-field launches"""
    chunks = max(1, (N + lib().ps_ms_chunk() - 1) // lib().ps_ms_chunk())
    key = (str(device), chunks)
    t = _SINGLE_FS.get(key)
    if t is None:
        if len(_SINGLE_FS) > 64:
            _SINGLE_FS.clear()
        t = torch.tensor([0, chunks], dtype=torch.int32, device=device)
        _SINGLE_FS[key] = t
    return t.data_ptr()


def sky_field(dirs: Tensor, app: Optional[Tensor], rgb_layers, sem_layers, lay: Optional[MsLayout] = None):
    """-> (rgb [R,3], semantics [R,64]).  Single field: rgb_layers / sem_layers are layer lists; routed (lay given): one list
    per sub-field."""
    flat = []
    if lay is None:
        rgb_layers, sem_layers = [rgb_layers], [sem_layers]
    for k in range(len(rgb_layers)):
        for W, b in list(rgb_layers[k]) + list(sem_layers[k]):
            flat += [W, b]
    return _SkyField.apply(lay, dirs, app, *flat)
