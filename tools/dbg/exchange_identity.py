"""Which part of the exchange path changes the result of a step?  One rank, process group of one (RCCL): parameters after 2 steps of
  v0 plain trainer (nothing armed, separate table update)      v1 bucket bookkeeping only (PRESIGHT_DRY_OVERLAP)
  v2 one all-reduce of the whole buffer after backward (PRESIGHT_NO_OVERLAP)      v3 bucketed all-reduce during backward      v4 sharded
against v0.    RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python tools/dbg/exchange_identity.py [K]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from presight_amd.dist import init_from_env  # noqa: E402
from test_hip_dist import _tiny_model  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
os.environ["PRESIGHT_EXCHANGE_WORLD_OF_ONE"] = "1"
init_from_env("cuda")
dev = torch.device("cuda", 0)


def run(env, exchange="allreduce", steps=2):
    for k in ("PRESIGHT_EXCHANGE_WORLD_OF_ONE", "PRESIGHT_DRY_OVERLAP", "PRESIGHT_NO_OVERLAP"):
        os.environ[k] = env.get(k, "0")
    model, scene = _tiny_model(dev, K=K)
    tr = bench.Trainer(model, scene, 1, exchange=exchange, fused_table_adam=False if not env.get("PRESIGHT_EXCHANGE_WORLD_OF_ONE") == "1" else None)
    g = torch.Generator(device=dev).manual_seed(5)
    batches = bench.make_batches(scene, dev, steps, 0, rays=512)
    grads = []
    for i in range(steps):
        jit = [torch.rand(512, 1, device=dev, generator=g) for _ in range(3)]
        tr.step(dict(batches[i], jitter=jit))
        torch.cuda.synchronize()
        grads.append(tr.grads.flat.clone())
    return {n: p.detach().clone() for n, p in model.named_parameters()}, grads, tr


base, g0, tr0 = run({})
for name, env, ex in (("v1 dry buckets", {"PRESIGHT_DRY_OVERLAP": "1"}, "allreduce"),
                      ("v2 one all-reduce after backward", {"PRESIGHT_EXCHANGE_WORLD_OF_ONE": "1", "PRESIGHT_NO_OVERLAP": "1"}, "allreduce"),
                      ("v3 bucketed all-reduce", {"PRESIGHT_EXCHANGE_WORLD_OF_ONE": "1"}, "allreduce"),
                      ("v4 bucketed reduce-scatter", {"PRESIGHT_EXCHANGE_WORLD_OF_ONE": "1"}, "sharded")):
    p, g, tr = run(env, ex)
    bad = [(n, float((p[n] - base[n]).abs().max())) for n in base if not torch.equal(p[n], base[n])]
    gd = [float((a - b).abs().max()) if a.shape == b.shape else -1.0 for a, b in zip(g, g0)]
    print(f"{name}: buckets {len(tr.grads._buckets)} dry {tr.grads.dry} mode {tr.grads.mode}; {len(bad)} of {len(base)} parameters differ; "
          f"flat gradient after step 0/1 differs by {gd}")
    for n, d in bad[:12]:
        print(f"      {d:.3e}  {n}")
