"""Soak: a 1000-iteration run of the reference loop at cfg-2 size (65 536 rays/step, the reference's proposal update schedule, LR
schedule with max_num_iterations = 1000, Adam on the 2**10-scaled gradients) on the LEARNABLE teacher-rendered scene
(presight_amd/synthetic.py): prints losses / held-out PSNR / peak memory every 100 iterations and asserts that the PSNR rises and
every parameter stays finite.      gpurun -- python tools/soak.py [iterations]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

import bench  # noqa: E402
from presight_amd.datafeed import ChunkFeed  # noqa: E402
from presight_amd.synthetic import TEACHER_FAR, TeacherScene, eval_psnr, shape_teacher_  # noqa: E402
from presight_amd.trainer import Trainer  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
tmodel, scene = bench.build_model(dev, seed=7, config="cfg2", far_plane=TEACHER_FAR)
shape_teacher_(tmodel)
teacher = TeacherScene(tmodel, scene)
chunk = teacher.chunk(0, pixels=1 << 22)
g = torch.Generator(device=dev).manual_seed(99)
tri = torch.stack([torch.randint(0, scene["c2w"].shape[0], (65536,), device=dev, generator=g), torch.randint(0, 900, (65536,), device=dev, generator=g),
                   torch.randint(0, 1600, (65536,), device=dev, generator=g)], -1)
tvid = torch.clamp(tri[:, 0] // scene["frames_per_video"], max=5)
test = teacher.targets(tri, tvid)
del teacher, tmodel
model, scene = bench.build_model(dev, seed=42, config="cfg2", proposal_weights_anneal_max_num_iters=K // 10, proposal_warmup=K // 10)
tr = Trainer(model, scene, 1, max_num_iterations=K)
feed = ChunkFeed(lambda i: chunk, batch_size=65536, device=dev, world=1, rank=0)
psnr = [eval_psnr(model, scene, tri, tvid, test["rgb"])]
print("iteration 0: held-out PSNR vs teacher %.2f dB" % psnr[0], flush=True)
t0 = time.time()
for i in range(K):
    ld, out = tr.step(feed.next_batch())
    if (i + 1) % 100 == 0 or i + 1 == K:
        psnr.append(eval_psnr(model, scene, tri, tvid, test["rgb"]))
        print(i + 1, "PSNR %.2f dB" % psnr[-1], "lr %.2e" % tr.opt.lr, {k: round(float(v.detach()), 5) for k, v in ld.items()},
              "mem GB", round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), flush=True)
feed.close()
assert all(torch.isfinite(p).all() for p in model.parameters())
assert psnr[-1] > psnr[0] + 6.0 and all(b >= a - 0.5 for a, b in zip(psnr, psnr[1:])), psnr
print("finite, PSNR %.2f -> %.2f dB, %.1f s" % (psnr[0], psnr[-1], time.time() - t0))
