"""Host time of every autograd node of a training step (forward and backward, which runs on the autograd engine's own thread where
cProfile does not look): wraps forward / backward of every torch.autograd.Function subclass defined in presight_amd.
    python tools/dbg/host_nodes.py cfg3 8192"""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import presight_amd  # noqa: E402,F401

cfg, rays = sys.argv[1], int(sys.argv[2])
acc = defaultdict(lambda: [0.0, 0])


def wrap(cls, name):
    fn = getattr(cls, name).__func__ if hasattr(getattr(cls, name), "__func__") else getattr(cls, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            r = acc[f"{cls.__name__}.{name}"]
            r[0] += time.perf_counter() - t0
            r[1] += 1

    setattr(cls, name, staticmethod(timed))


def all_subclasses(c):
    for s in c.__subclasses__():
        yield s
        yield from all_subclasses(s)


dev = torch.device("cuda", 0)
model, scene = bench.build_model(dev, seed=42, config=cfg)
tr = bench.Trainer(model, scene, 1)
for c in set(all_subclasses(torch.autograd.Function)):  # (after the model exists: presight_amd's modules are imported by then)
    if c.__module__.startswith("presight_amd"):
        wrap(c, "forward")
        wrap(c, "backward")
batches = bench.make_batches(scene, dev, 4, 0, rays=rays)
for i in range(8):
    tr.step(batches[i % 4])
torch.cuda.synchronize()
acc.clear()
n = 20
t0 = time.perf_counter()
for i in range(n):
    tr.step(batches[i % 4])
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{cfg} {rays} rays: enqueue {t_enq / n * 1e3:.2f} ms per step, wall {t_all / n * 1e3:.2f} ms per step")
tot = 0.0
for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    tot += t
    print(f"  {t / n * 1e3:7.3f} ms/step  {c / n:5.1f} calls/step  {t / max(c, 1) * 1e6:7.1f} us/call  {k}")
print(f"  {tot / n * 1e3:7.3f} ms/step inside presight_amd autograd nodes (nested nodes counted twice)")
