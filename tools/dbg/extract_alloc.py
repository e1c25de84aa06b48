"""Does the 512^3 extraction pass of the production tile go back to the driver for memory (hipMalloc / hipFree: synchronous, ~0.7 ms each)
in steady state?  Prints per pass: wall time, device allocations / frees of the caching allocator, reserved memory.
    python tools/dbg/extract_alloc.py [cfg3]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from presight_amd.extract import dense_tile_query  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
dev = torch.device("cuda", 0)
model, scene = bench.build_model(dev, seed=42, config=cfg)
model.eval()
boxes = scene["aabbs"].reshape(-1, 2, 3)
aabb = torch.stack([boxes[:, 0].min(0).values, boxes[:, 1].max(0).values]).to(dev)
for i in range(8):
    s0 = torch.cuda.memory_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        out = dense_tile_query(model, aabb, res=512)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1 = torch.cuda.memory_stats()
    print(f"pass {i}: {dt * 1e3:7.2f} ms  device allocs +{s1['num_device_alloc'] - s0['num_device_alloc']} frees +{s1['num_device_free'] - s0['num_device_free']} "
          f"retries +{s1['num_alloc_retries'] - s0['num_alloc_retries']}  reserved {s1['reserved_bytes.all.current'] / 2**30:.2f} GiB  kept {out['points'].shape[0]}", flush=True)
    del out
