"""Smoke/timing of the production (BASELINE cfg 3) shape on one GPU: 16 sub-fields, L=10 F=4 T=2^20 main tables,
8192 rays per rank (what one of 8 DP ranks processes), full training step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from presight_amd.model import NerfactoNuscMSModel, NerfactoNuscMSModelConfig

dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
torch.manual_seed(0)
conf = NerfactoNuscMSModelConfig(near_plane=0.005, far_plane=50.0, piecewise_sampler_threshold=5.0, implementation="hip", use_lidar_loss=False)
scene = bench.make_scene(1440, 6)
C = scene["c2w"].shape[0]
sel = torch.linspace(0, C - 1, K + 2)[1:-1].long()
cent = scene["c2w"][sel, :, 3].clone()
ext = 15.0 * 0.05
aabbs = torch.stack([torch.stack([c - 3 * ext, c + 3 * ext]) for c in cent])
scene["centroids"], scene["aabbs"] = cent, aabbs
t0 = time.time()
model = NerfactoNuscMSModel(conf, num_train_cameras=1440, num_train_videos=6, dino_to_rgb=None, centroids=cent, aabbs=aabbs).to(dev)
print("params", sum(p.numel() for p in set(model.parameters())) / 1e6, "M  build", time.time() - t0, "s")
scene = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
tr = bench.Trainer(model, scene, 1)
batches = bench.make_batches(scene, dev, 2, 0, rays=rays)
for i in range(3):
    ld, out = tr.step(batches[i % 2])
torch.cuda.synchronize()
t0 = time.time()
n = 5
for i in range(n):
    ld, out = tr.step(batches[i % 2])
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(f"cfg3 shape K={K} rays={rays}: {dt*1e3:.1f} ms/step -> {rays/dt:.0f} rays/s per GPU; loss {float(sum(ld.values())):.4f}; "
      f"mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
