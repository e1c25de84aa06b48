import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch
import test_hip_dual as T
from conftest import to_double
from oracle import dual_oracle as D
dev = torch.device("cuda:0")
import oracle.nerf_oracle as O
orig = D.dual_config
for K, cams in ((1, 12), (1, 48), (3, 48)):
    def cfgf(tiny=True, levels=2, feats=2, _c=cams):
        c = orig(tiny, levels, feats); c["num_cameras"] = _c; return c
    D.dual_config = cfgf
    model, cfg, scene, P, batch, bundle, _ = T._dual_setup(dev, K=K, rays=160)
    out = model(bundle(), jitters=[j.to(dev) for j in batch["jitter"]])
    gt = {k: batch[k].to(dev) for k in ("rgb", "features", "sky")}
    losses = model.get_loss_dict(out, gt)
    sum(losses.values()).backward()
    L_ref, out_ref, g_ref = D.dual_train_step(P, cfg, scene, batch)
    _, _, g64 = D.dual_train_step(to_double(P), cfg, to_double(scene), to_double(batch))
    named = dict(model.named_parameters())
    print("K", K, "cams", cams, "times", sorted(set(batch["times"].tolist()))[:4])
    for n in g_ref:
        if n.startswith("dynamic_field") or "hash_table" in n:
            got = named[n].grad.detach().cpu(); ref = g_ref[n]
            sc = float(ref.abs().max())
            if sc == 0: continue
            e = float((got-ref).abs().max())/sc; e64 = float((ref.double()-g64[n]).abs().max())/float(g64[n].abs().max())
            l2 = float((got-ref).norm()/ref.norm())
            print(f"   {n:50s} max {e:.1e} l2 {l2:.1e} oracle-noise {e64:.1e}")
