"""Step time of BASELINE cfg 2 with the gradient exchange LIVE through RCCL in a process group of one rank (one-GPU box: the only way
RCCL can run this code here) against the same step without an exchange: what the exchange machinery itself costs a rank -- bucket
hand-over on the side stream, RCCL launches, the separate (unfused) table update, the sharded mode's parameter all-gather -- before any
byte crosses a link.
    RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 python tools/rccl_self_exchange.py [cfg2] [rays] [steps] [variant,...]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from presight_amd.dist import init_from_env  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
only = sys.argv[4].split(",") if len(sys.argv) > 4 else None  # e.g. rccl_group_of_one_sharded (for a kernel trace of one variant)
os.environ["PRESIGHT_EXCHANGE_WORLD_OF_ONE"] = "1"
init_from_env("cuda")
import torch.distributed as dist  # noqa: E402

dev = torch.device("cuda", 0)
out = {"config": cfg, "rays": rays, "steps": steps, "backend": dist.get_backend()}
pieces = {"table_pieces": 2} if rays <= 16384 else {}  # (what bench.py gives a strong-scaled rank)
for name, flag, kw in (("no_exchange_fused_tables", "0", {}), ("no_exchange_separate_table_update", "0", {"fused_table_adam": False}),
                       ("rccl_group_of_one_allreduce", "1", {"exchange": "allreduce", **pieces}),
                       ("rccl_group_of_one_sharded", "1", {"exchange": "sharded", **pieces})):
    if only is not None and name not in only:
        continue
    os.environ["PRESIGHT_EXCHANGE_WORLD_OF_ONE"] = flag
    model, scene = bench.build_model(dev, seed=42, config=cfg)
    tr = bench.Trainer(model, scene, 1, **kw)
    batches = bench.make_batches(scene, dev, 4, 0, rays=rays)
    for i in range(5):
        tr.step(batches[i % 4])
    torch.cuda.synchronize()
    tr.grads.stats = {"collectives": 0, "bytes": 0}
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(batches[i % 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out[name] = {"ms_per_step": round(dt * 1e3, 3), "collectives_per_step": tr.grads.stats["collectives"] / steps,
                 "handed_over_in_backward_per_step": tr.grads.stats.get("in_backward", 0) / steps, "buckets": len(tr.grads._buckets)}
    del tr, model
    torch.cuda.empty_cache()
dist.destroy_process_group()
print(json.dumps(out))
