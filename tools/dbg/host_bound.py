"""Is a step host-bound?  enqueue time (host returns from Trainer.step) vs wall time per step, for a bench configuration / ray count.
    python tools/dbg/host_bound.py cfg3 8192"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

cfg, rays = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0)
model, scene = bench.build_model(dev, seed=42, config=cfg)
tr = bench.Trainer(model, scene, 1)
batches = bench.make_batches(scene, dev, 4, 0, rays=rays)
for rep in range(3):
    for i in range(5):
        tr.step(batches[i % 4])
    torch.cuda.synchronize()
    enq, t0 = [], time.perf_counter()
    for i in range(20):
        a = time.perf_counter()
        tr.step(batches[i % 4])
        enq.append(time.perf_counter() - a)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{cfg} {rays} rays: enqueue {t_enq / 20 * 1e3:.2f} ms/step (min {min(enq) * 1e3:.2f}, max {max(enq) * 1e3:.2f}), wall {t_all / 20 * 1e3:.2f} ms/step", flush=True)
