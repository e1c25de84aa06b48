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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    tr.step(batches[0])
torch.cuda.synchronize()
rows = [(e.count, e.key, str(e.input_shapes)[:90]) for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
rows.sort(reverse=True)
for c, k, sh in rows[:70]:
    print(f"{c:4d} {k:38s} {sh}")

# where do the GPU-launching element-wise ops come from?
print()
launching = ("aten::mul", "aten::add", "aten::add_", "aten::sum", "aten::mean", "aten::fill_", "aten::zero_", "aten::copy_", "aten::clone",
             "aten::rand", "aten::uniform_", "aten::div", "aten::sub", "aten::neg", "aten::where", "aten::clamp", "aten::sqrt", "aten::pow")
import collections
sites = collections.Counter()
for e in prof.events():
    if e.name in launching and e.stack:
        fr = [f for f in e.stack if "/root/repo" in f or "presight_amd" in f or "bench.py" in f]
        sites[(e.name, fr[0] if fr else e.stack[0])] += 1
for (name, site), c in sites.most_common(60):
    print(f"{c:3d} {name:16s} {site[:130]}")
