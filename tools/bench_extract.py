"""Prior-extraction throughput (BASELINE cfg 5): dense res^3 lattice over one tile through the three fields
(mean density of 2 proposal nets + main field, 64-d semantics clipped to fp16, density threshold, integer voxel index)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from presight_amd import extract

res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
model, scene = bench.build_model(dev, seed=42)
model.eval()
aabb = scene["aabbs"][0]
extract.dense_tile_query(model, aabb, res=64, density_threshold=1e9)  # warm-up
torch.cuda.synchronize()
t0 = time.time()
out = extract.dense_tile_query(model, aabb, res=res, chunk=1 << 23, density_threshold=1e9)  # nothing kept: pure query rate
torch.cuda.synchronize()
dt = time.time() - t0
n = res ** 3
print(f"lattice {res}^3 = {n/1e6:.1f} M points: {dt:.2f} s -> {n/dt/1e6:.1f} M points/s on one GPU "
      f"(algorithmic 1792 B + 42 240 FLOP per point, SURVEY 8d: {n*1792/dt/1e12:.2f} TB/s, {n*42240/dt/1e12:.1f} TFLOP/s)")
