"""Long-horizon behaviour of the REFERENCE's training loop on the learnable scene, by the pinned CPU oracle (and, with a GPU, by the HIP
trainer on the very same batches and jitters): held-out eval PSNR (no jitter, MEAN appearance code: the reference's eval branch) at
regular marks of an N-iteration run of Trainer.train_iteration's loop -- Adam(lr 1e-2, eps 1e-15, weight_decay 1e-5) on the 2**10-scaled
gradients, the method configs' LR schedule.  Does the eval PSNR sag while the training loss keeps falling (tools/soak.py at cfg-2 size
does), and is that the loop's own behaviour or the HIP step's?          python tools/long_run_oracle.py [iterations] [marks] [hip]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_hip_model, learnable_scene_setup  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402

torch.set_num_threads(min(16, torch.get_num_threads()))  # (torch's CPU kernels peak at ~16 threads on the GPU boxes' 128-core hosts)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n_marks = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mode = sys.argv[3] if len(sys.argv) > 3 else "oracle"  # oracle | hip (the HIP trainer only: the oracle curve comes from a CPU run) | both
hip = mode in ("hip", "both")
t0 = time.time()
CAMS = int(os.environ.get("LONG_RUN_CAMERAS", "24"))  # (more cameras than rays per batch: most appearance codes see no ray in a step, as at full size)
cfg, scene, Pt, batches, test, P0 = learnable_scene_setup(rays=192, steps=N, test_rays=768, num_cameras=CAMS)
print(f"{cfg['num_cameras']} cameras", flush=True)
print(f"{N} batches of 192 teacher-rendered rays in {time.time() - t0:.0f} s", flush=True)
marks = sorted({round(N * (i + 1) / n_marks) - 1 for i in range(n_marks)})
print("iterations      ", [0] + [m + 1 for m in marks])
oracle = None
if mode in ("oracle", "both"):
    t0 = time.time()
    r = O.train_trajectory(P0, cfg, scene, batches, N, snapshots=marks)
    ev = lambda P: O.eval_psnr(P, cfg, scene, test["ray_indices"], test["video_ids"], test["rgb"])  # noqa: E731
    oracle = [ev(P0)] + [ev(r["snaps"][s]) for s in marks]
    print(f"oracle: {N} iterations in {time.time() - t0:.0f} s")
    print("oracle eval PSNR", [round(x, 2) for x in oracle])
    code = lambda P: float((P["appearance_embedding.embedding.weight"] - P["appearance_embedding.embedding.weight"].mean(0)).norm(dim=1).mean())  # noqa: E731
    print("oracle: mean distance of the appearance codes from their mean", [round(code(P0), 3)] + [round(code(r["snaps"][s]), 3) for s in marks])
    print("oracle total loss at the marks", [round(float(sum(r["losses"][s])), 5) for s in marks])
if hip:
    from presight_amd.synthetic import eval_psnr
    from presight_amd.trainer import Trainer

    dev = torch.device("cuda:0")
    sdev = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
    model = build_hip_model(cfg, scene, P0, dev, proposal_weights_anneal_max_num_iters=N // 10, proposal_warmup=N // 10)
    tr = Trainer(model, sdev, max_num_iterations=N)
    tri, tvid, trgb = test["ray_indices"].to(dev), test["video_ids"].to(dev), test["rgb"].to(dev)
    got = [eval_psnr(model, sdev, tri, tvid, trgb)]
    for s in range(N):
        tr.step({k: v.to(dev) for k, v in batches[s].items() if k != "accumulation"})
        if s in marks:
            got.append(eval_psnr(model, sdev, tri, tvid, trgb))
    print("HIP eval PSNR   ", [round(x, 2) for x in got])
    w = model.appearance_embedding.embedding.weight.detach()
    print("HIP: mean distance of the appearance codes from their mean at the end", round(float((w - w.mean(0)).norm(dim=1).mean()), 3))
    if oracle is not None:
        print("HIP - oracle    ", [round(a - b, 2) for a, b in zip(got, oracle)])
