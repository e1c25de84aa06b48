"""which run of the fused-render test disagrees on the sky gradients (debug)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from presight_amd import ops, field_ops
from presight_amd.model import NerfactoNuscMSModel, NerfactoNuscMSModelConfig
from presight_amd.rays import RayBundle

dev = torch.device("cuda:0")
scene = bench.make_scene(60, 6)
samples = (128, 64, 64)
conf = NerfactoNuscMSModelConfig(near_plane=0.005, far_plane=50.0, piecewise_sampler_threshold=5.0, num_levels=2,
                                 features_per_level=2, log2_hashmap_size=12, base_res=16, max_res=128, hidden_dim=32,
                                 hidden_dim_color=32, implementation="hip", use_lidar_loss=False,
                                 num_proposal_samples_per_ray=samples[:2], num_nerf_samples_per_ray=samples[2])
torch.manual_seed(11)
model = NerfactoNuscMSModel(conf, num_train_cameras=60, num_train_videos=6, dino_to_rgb=None, centroids=scene["centroids"],
                            aabbs=scene["aabbs"]).to(dev)
with torch.no_grad():
    model.field.fields[0].mlp_base_mlp.layers[-1].bias[0] = 2.0
scene = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
R = 300
batch = bench.make_batches(scene, dev, 1, 0, rays=R)[0]
g = torch.Generator().manual_seed(2)
jit = [torch.rand(R, 1, generator=g).to(dev) for _ in range(3)]
res = []
for fused in sys.argv[1].split(","):
    model.fused_render = fused != "u"
    field_ops.FACTORED = fused == "f"
    model.zero_grad(set_to_none=True)
    model.train()
    o, d, pa, dn = ops.generate_rays(batch["ray_indices"], scene["c2w"], scene["fx"], scene["fy"], scene["cx"], scene["cy"])
    rb = RayBundle(o, d, pa, camera_indices=batch["ray_indices"][:, 0:1], metadata={"video_id": batch["video_ids"][:, None], "directions_norm": dn})
    out = model(rb, jitters=jit)
    ld = model.get_loss_dict(out, batch)
    hooks = {}
    for k in ("rgb", "accumulation", "semantics"):
        out[k].retain_grad()
    (sum(ld.values()) + out["expected_depth"].mean() * 0.1).backward()
    res.append((fused, {k: out[k].grad.clone() for k in ("rgb", "accumulation", "semantics")}, {k: out[k].detach().clone() for k in ("rgb", "accumulation", "semantics")},
                {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None and "sky" in k}))
for i in range(1, len(res)):
    print(res[i][0], "vs", res[0][0])
    for k in res[0][1]:
        print("  d", k, float((res[i][1][k] - res[0][1][k]).abs().max()), float(res[0][1][k].abs().max()), " out", float((res[i][2][k] - res[0][2][k]).abs().max()))
    for k in list(res[0][3])[:3]:
        print("  ", k, float((res[i][3][k] - res[0][3][k]).abs().max()), float(res[0][3][k].abs().max()))
