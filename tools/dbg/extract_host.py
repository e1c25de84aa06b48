"""Host time of the pieces of the gated routed main-field query during a 512^3 extraction pass (which call makes the GPU wait 0.7 ms
before main_fwd_kernel in the slow mode?).   python tools/dbg/extract_host.py"""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from presight_amd import field_ops as F  # noqa: E402
from presight_amd._lib import lib  # noqa: E402
from presight_amd.extract import dense_tile_query  # noqa: E402

acc = defaultdict(lambda: [0.0, 0])


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            r = acc[name]
            r[0] += time.perf_counter() - t0
            r[1] += 1
    return w


dev = torch.device("cuda", 0)
model, scene = bench.build_model(dev, seed=42, config="cfg3")
model.eval()
boxes = scene["aabbs"].reshape(-1, 2, 3)
aabb = torch.stack([boxes[:, 0].min(0).values, boxes[:, 1].max(0).values]).to(dev)
probe = dense_tile_query(model, aabb, res=64, density_threshold=-1.0)
thr = float(torch.quantile(probe["densities"][:: max(1, probe["densities"].numel() // 100000)], 0.9))  # bench.py's threshold: densest 10 %
del probe
F._ms_encode = timed("_ms_encode", F._ms_encode)
F._MsMerged.merge = timed("merge", F._MsMerged.merge)
F._MsStacks.pack = timed("pack", F._MsStacks.pack)
F._MsStacks.get = staticmethod(timed("stacks.get", F._MsStacks.get)) if isinstance(F._MsStacks.__dict__.get("get"), staticmethod) else classmethod(timed("stacks.get", F._MsStacks.get.__func__))
real_empty = torch.empty


def empty(*a, **k):
    t0 = time.perf_counter()
    out = real_empty(*a, **k)
    if out.numel() * out.element_size() >= 1 << 28:
        r = acc[f"torch.empty >= 256 MiB"]
        r[0] += time.perf_counter() - t0
        r[1] += 1
    return out


torch.empty = empty
L = lib()
fn = L.ps_main_field_fwd_gated_ms
L.ps_main_field_fwd_gated_ms = timed("ps_main_field_fwd_gated_ms (launch call)", fn)
for i in range(6):
    if i == 2:
        acc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        out = dense_tile_query(model, aabb, res=512, chunk=1 << 23, density_threshold=thr)
    torch.cuda.synchronize()
    print(f"pass {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms, kept {out['points'].shape[0]}", flush=True)
    del out
for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {t / 4 * 1e3:8.3f} ms per pass  {c / 4:6.1f} calls per pass  {t / max(c, 1) * 1e6:8.1f} us per call  {k}")
