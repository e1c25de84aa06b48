import sys, os
R_ = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import torch
import test_hip_dual as T
from oracle import dual_oracle as D
import oracle.nerf_oracle as O
dev = torch.device("cuda:0")
orig = D.dual_config
def cfgf(tiny=True, levels=2, feats=2):
    c = orig(tiny, levels, feats); c["num_cameras"] = 48; return c
D.dual_config = cfgf
model, cfg, scene, P, batch, bundle, _ = T._dual_setup(dev, K=1, rays=160)
# the model's actual main-field sample points
out = D.dual_model_forward(P, cfg, scene, batch, training=True)
eu = out["euclid_list"][-1]; o, d = out["origins"], out["directions"]
mid = (eu[:, :-1] + eu[:, 1:]) / 2
pos = (o[:, None, :] + d[:, None, :] * mid[:, :, None]).reshape(-1, 3)
u, sel = O.normalize_contract(pos, D.dynamic_aabb(scene))
Rn, S = mid.shape
times = batch["times"]
tt = times[:, None].expand(Rn, S).reshape(-1)
print("points", u.shape[0], "masked", int((sel == 0).sum()), "distinct times", sorted(set(times.tolist())))
for label, tvec in (("actual times", tt), ("times rounded to {0,1}", tt.round())):
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items() if k.startswith("dynamic_field")}
    feat_ref, parts = D.dynamic_features(Pg, cfg, u, tvec, return_parts=True)
    g = torch.Generator().manual_seed(5)
    wgt = torch.randn(feat_ref.shape, generator=g)
    (feat_ref * wgt).sum().backward()
    df = model.dynamic_field
    model.zero_grad(set_to_none=True)
    tr = tvec.view(Rn, S)[:, 0].contiguous()
    feat = df.features(u.to(dev), tr.to(dev), S)
    rows = T._planes_to_rows(feat)
    dfeat = (rows.cpu() - feat_ref.detach()).abs()
    print(label, "feature max diff", float(dfeat.max()), "rows > 1e-5:", int((dfeat.max(1).values > 1e-5).sum()))
    bad = dfeat.max(1).values > 1e-5
    if bad.any():
        i = int(torch.nonzero(bad)[0]); print("   first bad row", i, "u", u[i].tolist(), "t", float(tvec[i]), "sel", float(sel[i]), "xf", parts["xf"][i].tolist(), "xb", parts["xb"][i].tolist())
    (rows * wgt.to(dev)).sum().backward()
    for n, p in df.named_parameters():
        ref = Pg["dynamic_field." + n].grad
        if ref is None or float(ref.abs().max()) == 0: continue
        print(f"   {n:40s} max {float((p.grad.cpu()-ref).abs().max())/float(ref.abs().max()):.1e}")
