"""cProfile of the host side of Trainer.step (which Python code the enqueue time of a step goes into).
    python tools/dbg/host_profile.py cfg3 8192 [top]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

cfg, rays = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
dev = torch.device("cuda", 0)
model, scene = bench.build_model(dev, seed=42, config=cfg)
tr = bench.Trainer(model, scene, 1)
batches = bench.make_batches(scene, dev, 4, 0, rays=rays)
for i in range(8):
    tr.step(batches[i % 4])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    tr.step(batches[i % 4])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(top)
st.sort_stats("tottime").print_stats(30)
