"""Which torch (aten) operators still run in one training step, and how often?  (what is left to fuse)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
model, scene = bench.build_model(dev, seed=42)
tr = bench.Trainer(model, scene, 1)
batches = bench.make_batches(scene, dev, 2, 0)
for i in range(3):
    tr.step(batches[i % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    tr.step(batches[0])
torch.cuda.synchronize()
rows = [(e.count, e.key, str(e.input_shapes)[:90]) for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
rows.sort(reverse=True)
for c, k, sh in rows[:70]:
    print(f"{c:4d} {k:38s} {sh}")
