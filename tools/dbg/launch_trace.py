"""Which Python line every aten-level launch of one training step comes from (the library's own kernels are named in the rocprofv3
timeline; the anonymous `vectorized_elementwise_kernel`s, fills and copies are not).
    python tools/dbg/launch_trace.py [cfg2] [65536]
Prints, per aten operator that launched device work, the call count per step and the innermost presight_amd / bench frames."""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = 4
dev = torch.device("cuda", 0)
model, scene = bench.build_model(dev, seed=42, config=cfg)
tr = bench.Trainer(model, scene, 1)
batches = bench.make_batches(scene, dev, 4, 0, rays=rays)
for i in range(6):
    tr.step(batches[i % 4])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(steps):
        tr.step(batches[i % 4])
    torch.cuda.synchronize()

rows = Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    kernels = [k for k in getattr(ev, "kernels", [])]
    if not kernels:
        continue
    # only the innermost operator owns the launch: skip events whose child also has kernels
    if any(getattr(c, "kernels", []) for c in ev.cpu_children):
        continue
    frames = [f for f in (ev.stack or []) if ("presight_amd/" in f or "bench.py" in f or "tools/" in f)]
    where = " <- ".join(f.split("presight_amd/")[-1].split("(")[0].strip() + ":" + f.split("(")[-1].rstrip(")") if "(" in f else f
                        for f in frames[:3]) or "(autograd engine)"
    kn = ",".join(sorted({k.name.split("<")[0].split("(")[0][-40:] for k in kernels}))
    rows[(ev.name, kn, where)] += len(kernels)
print(f"{cfg} {rays} rays: aten-level launches per step (operator, kernels, where)")
tot = 0
for (name, kn, where), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    tot += n
    print(f"{n / steps:6.2f}  {name:34s} {kn:42s} {where}")
print(f"total {tot / steps:.1f} launches per step carried by aten operators")
