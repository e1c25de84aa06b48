"""400 training steps of cfg 2 on the reference's proposal-update schedule: losses, peak memory, finiteness (gpurun -- python tools/soak.py)"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
dev = torch.device("cuda:0")
model, scene = bench.build_model(dev, 42, config="cfg2")
tr = bench.Trainer(model, scene, 1)
tr.update_props_every_step = False  # the reference schedule
batches = bench.make_batches(scene, dev, 8, 0, rays=65536)
t0 = time.time()
for i in range(400):
    ld, out = tr.step(batches[i % 8])
    if i % 50 == 0 or i == 399:
        torch.cuda.synchronize()
        print(i, round(float(sum(v.detach() for v in ld.values())), 5), {k: round(float(v), 5) for k, v in ld.items()}, "mem GB", round(torch.cuda.max_memory_allocated() / 2**30, 2), flush=True)
assert all(torch.isfinite(p).all() for p in model.parameters())
print("finite, %.1f s" % (time.time() - t0))
