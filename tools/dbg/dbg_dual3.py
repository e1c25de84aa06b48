import sys, os
R_ = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import torch
import test_hip_dual as T
from conftest import to_double
from oracle import dual_oracle as D
dev = torch.device("cuda:0")
for seed in (3, 4, 5, 6, 7, 8):
  for rays in (96, 160):
    model, cfg, scene, P, batch, bundle, _ = T._dual_setup(dev, K=3, rays=rays, seed=seed)
    out = model(bundle(), jitters=[j.to(dev) for j in batch["jitter"]])
    gt = {k: batch[k].to(dev) for k in ("rgb", "features", "sky")}
    sum(model.get_loss_dict(out, gt).values()).backward()
    L_ref, out_ref, g_ref = D.dual_train_step(P, cfg, scene, batch)
    named = dict(model.named_parameters())
    worst = 0; wn = ""
    for n in g_ref:
        sc = float(g_ref[n].abs().max())
        if sc == 0: continue
        e = float((named[n].grad.detach().cpu() - g_ref[n]).abs().max()) / sc
        if e > worst: worst, wn = e, n
    print("seed", seed, "rays", rays, "worst", f"{worst:.1e}", wn)
