import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import grad_error_stats
from oracle import nerf_oracle as O
from presight_amd import ops
from presight_amd.model import NerfactoNuscMSModel, NerfactoNuscMSModelConfig
from presight_amd.rays import RayBundle
dev = torch.device("cuda:0")
cfg = O.tiny_config()
for p in [cfg["main"]] + cfg["props"]:
    p["log2_hashmap_size"] = 10
scene = O.make_scene(cfg)
import itertools
for bias, rays in itertools.product((-2.0,), (64, 96, 128, 192, 256, 512)):
    P = O.make_params(cfg, seed=3, table_scale=0.3)
    P["field.fields.0.mlp_base_mlp.layers.1.bias"][0] = bias
    for i in range(2):
        P[f"proposal_networks.{i}.fields.0.mlp_base.1.layers.1.bias"][0] = -2.0
    batch = O.make_batch(cfg, scene, rays, step=0)
    m = cfg["main"]
    conf = NerfactoNuscMSModelConfig(near_plane=cfg["near"], far_plane=cfg["far"], piecewise_sampler_threshold=cfg["thr"], hidden_dim=m["hidden_dim"],
        hidden_dim_color=m["hidden_dim_color"], num_levels=m["num_levels"], base_res=m["base_res"], max_res=m["max_res"],
        log2_hashmap_size=m["log2_hashmap_size"], features_per_level=m["features_per_level"], use_lidar_loss=False,
        proposal_net_args_list=[dict(features_per_level=p["features_per_level"], log2_hashmap_size=p["log2_hashmap_size"], num_levels=p["num_levels"], base_res=p["base_res"], max_res=p["max_res"], hidden_dim=p["hidden_dim"], use_linear=False) for p in cfg["props"]], implementation="hip")
    L_ref, out_ref, g_ref = O.train_step(P, cfg, scene, batch)
    for fused in (True,):
        model = NerfactoNuscMSModel(conf, num_train_cameras=cfg["num_cameras"], num_train_videos=cfg["num_videos"], dino_to_rgb=None, centroids=scene["centroids"], aabbs=scene["aabbs"])
        sd = dict(model.state_dict())
        for k, v in P.items():
            for name in (k, k.replace("mlp_base_grid.", "mlp_base.0.").replace("mlp_base_mlp.", "mlp_base.1."), k.replace("encoding.hash_table", "mlp_base.0.hash_table")):
                if name in sd: sd[name] = v
        model.load_state_dict(sd); model.to(dev).train(); model.fused_render = fused
        ri = batch["ray_indices"].to(dev)
        o, d, pa, dn = ops.generate_rays(ri, *(scene[k].to(dev) for k in ("c2w", "fx", "fy", "cx", "cy")))
        rb = RayBundle(o, d, pa, camera_indices=ri[:, 0:1], metadata={"video_id": batch["video_ids"].to(dev)[:, None]})
        out = model(rb, jitters=[j.to(dev) for j in batch["jitter"]])
        gt = {k: batch[k].to(dev) for k in ("rgb", "features", "sky")}
        sum(model.get_loss_dict(out, gt).values()).backward()
        errs, names, nz = grad_error_stats({n: p.grad for n, p in model.named_parameters()}, g_ref)
        print("rays", rays, "bias", bias, "fused", fused, "acc max", float(out_ref["accumulation"].max()), [(n.split("fields.0.")[-1], f"{float(e):.1e}") for n, e in zip(names[-6:], errs[-6:])])
