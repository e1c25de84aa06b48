"""bench.py — training rays/sec of the PreSight NeRF prior-builder hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full training iteration of NerfactoNuscMSModel on one synthetic nuScenes-shaped ray batch
(BASELINE.json configs[1]: one Boston-Seaport sub-tile, 16-level hash grid (F=2, T=2^19) + 64-wide MLPs, proposal nets
L=8 F=1 T=2^20, 65 536 rays, 128/64/64 samples): ray generation -> proposal sampling (2 proposal fields, 2 PDF
resamplings) -> main field -> compositing -> sky -> 5 losses -> full backward (proposal nets updated every step) ->
gradient exchange (N > 1) -> Adam.  Inputs are resident in HBM before the timed region.  Weak scaling: every rank
trains on its own 65 536-ray batch and the gradients are averaged with one RCCL all-reduce per step.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, measured live with HIP
events on the launch stream) and `cpu_baseline` (the CPU oracle timed on the host cores, rank 0, N == 1 only)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RAYS = 65536
# algorithmic work per training ray at cfg 2: MLP flops fwd = 2*(64*26752 + 192*576)
MAIN_MAC_PER_SAMPLE = 26752
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense fp32 matrix peak
# HBM bytes per launch of main_bwd_kernel from the PMC passes of this very command (profiles/r01_pmc_summary_v6.txt, separate
# --pmc passes): FETCH_SIZE 3 717 794 KB, doubled as MI355X_MICROARCH.md prescribes for gfx950 streaming 16-byte reads, +
# WRITE_SIZE 585 184 KB.  Algorithmic: kept activations in 7.0 GB + features in 0.54 GB + d(features) out 0.54 GB.
MAIN_BWD_HBM_BYTES_PMC = (2 * 3717793.5 + 585184.0) * 1024


def cfg2():
    return dict(near=0.005, far=50.0, thr=5.0, num_cameras=1440, num_videos=6)


def build_model(dev, seed):
    from presight_amd.model import NerfactoNuscMSModel, NerfactoNuscMSModelConfig

    torch.manual_seed(seed)
    c = cfg2()
    conf = NerfactoNuscMSModelConfig(
        near_plane=c["near"], far_plane=c["far"], piecewise_sampler_threshold=c["thr"],
        # iNGPField constructor defaults = BASELINE cfg 2 (ns/fields/PreSight/ingp_field.py:74-84)
        num_levels=16, features_per_level=2, log2_hashmap_size=19, base_res=16, max_res=2048, hidden_dim=64, hidden_dim_color=64,
        implementation="hip", use_lidar_loss=False, proposal_weights_anneal_max_num_iters=10000, proposal_warmup=10000)
    scene = make_scene(c["num_cameras"], c["num_videos"])
    model = NerfactoNuscMSModel(conf, num_train_cameras=c["num_cameras"], num_train_videos=c["num_videos"], dino_to_rgb=None,
                                centroids=scene["centroids"], aabbs=scene["aabbs"])
    model.to(dev)
    return model, {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}


def make_scene(num_cameras, num_videos, seed=7):
    """Synthetic nuScenes-shaped rig: 6 pinhole cameras 1600x900 moving along a polyline at 0.5 m/frame,
    poses scaled by 0.05 and mean-centred; one sub-field whose AABB is the trajectory quantile box +-15 m."""
    import math

    gen = torch.Generator().manual_seed(seed)
    n_frames = num_cameras // 6
    scale = 0.05
    t = torch.arange(n_frames, dtype=torch.float32) * 0.5
    heading = 0.3 * torch.sin(t / 40.0)
    pos = torch.stack([torch.cumsum(0.5 * torch.cos(heading), 0), torch.cumsum(0.5 * torch.sin(heading), 0), torch.full_like(t, 1.5)], -1)
    yaws = torch.tensor([0.0, 55.0, -55.0, 180.0, 110.0, -110.0]) * math.pi / 180
    c2w = torch.zeros(n_frames, 6, 3, 4)
    for j in range(6):
        a = heading + yaws[j]
        fwd = torch.stack([torch.cos(a), torch.sin(a), torch.zeros_like(a)], -1)
        up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
        right = torch.linalg.cross(fwd, up)
        c2w[:, j, :, 0], c2w[:, j, :, 1], c2w[:, j, :, 2], c2w[:, j, :, 3] = right, up, -fwd, pos
    c2w = c2w.reshape(-1, 3, 4)
    c2w[:, :, 3] = (c2w[:, :, 3] - c2w[:, :, 3].mean(0)) * scale
    C = c2w.shape[0]
    fx = torch.full((C,), 1266.0) + torch.rand(C, generator=gen)
    ext = 15.0 * scale
    lo = c2w[:, :, 3].quantile(0.02, dim=0) - ext
    hi = c2w[:, :, 3].quantile(0.98, dim=0) + ext
    return dict(c2w=c2w, fx=fx, fy=fx.clone(), cx=torch.full((C,), 800.0), cy=torch.full((C,), 450.0),
                centroids=c2w[C // 2: C // 2 + 1, :, 3].clone(), aabbs=torch.stack([lo, hi])[None], H=900, W=1600,
                frames_per_video=max(1, C // num_videos))


def make_batches(scene, dev, n_batches, rank, rays=RAYS):
    """uniform-random ray indices + random targets, manual_seed(1234 + step) (+rank: each DP rank draws its own rays)"""
    out = []
    C = scene["c2w"].shape[0]
    for step in range(n_batches):
        g = torch.Generator(device="cpu").manual_seed(1234 + step + 1000 * rank)
        idx = torch.stack([torch.randint(0, C, (rays,), generator=g), torch.randint(0, scene["H"], (rays,), generator=g),
                           torch.randint(0, scene["W"], (rays,), generator=g)], -1)
        out.append(dict(ray_indices=idx.to(dev), video_ids=torch.clamp(idx[:, 0] // scene["frames_per_video"], max=5).to(dev),
                        rgb=torch.rand(rays, 3, generator=g).to(dev), features=torch.rand(rays, 64, generator=g).to(dev),
                        sky=(torch.rand(rays, generator=g) < 0.15).float().to(dev)))
    return out


class Trainer:
    """The timed region: what ns/engine/trainer.py:463-505 does per iteration (zero_grad, forward, loss, backward with
    the reference's fixed loss scale of 2**10, DDP gradient averaging, Adam lr 1e-2 eps 1e-15 wd 1e-5)."""

    def __init__(self, model, scene, world):
        from presight_amd.dist import FlatGrads

        self.model, self.scene, self.world = model, scene, world
        groups = model.get_param_groups()  # group-major order: a group that receives no gradient in a step (proposal nets
        params = [p for k in sorted(groups, reverse=True) for p in groups[k]]  # off-schedule) is one contiguous range to skip
        params = [p for p in params if p.requires_grad and p.numel() > 0]
        assert {id(p) for p in params} == {id(p) for p in model.parameters() if p.requires_grad and p.numel() > 0}
        # the reference registers mlp_base = Sequential(grid, mlp): the same tensors appear twice in parameters() -> dedup
        seen, uniq = set(), []
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.grads = FlatGrads(uniq)
        # K > 1: a sub-field may get samples on one rank only; "received a gradient" must then be agreed across ranks (DDP)
        self.grads.flags_may_differ_across_ranks = world > 1 and len(model.field.fields) > 1
        if world > 1 and not model.config.use_same_proposal_network and os.environ.get("PRESIGHT_NO_OVERLAP") != "1":
            # one bucket per optimizer group: the "fields" bucket (main table + MLPs + sky + embeddings, 2/3 of the bytes) is
            # complete before the proposal networks' backward starts and is exchanged underneath it
            uid = {id(p) for p in uniq}
            self.grads.enable_overlap([[p for p in groups[k] if id(p) in uid] for k in sorted(groups, reverse=True)])
        from presight_amd.optim import HipAdam

        self.opt = HipAdam(uniq, lr=1e-2, eps=1e-15, weight_decay=1e-5, flat_grads=self.grads)
        self.step_idx = 0
        self.loss_scale = 2.0 ** 10
        self.update_props_every_step = True

    def step(self, batch):
        from presight_amd import ops
        from presight_amd.rays import RayBundle

        m, s = self.model, self.scene
        m.train()
        m.before_train_iteration(self.step_idx)
        self.grads.zero_()
        o, d, pa, dn = ops.generate_rays(batch["ray_indices"], s["c2w"], s["fx"], s["fy"], s["cx"], s["cy"])
        rb = RayBundle(o, d, pa, camera_indices=batch["ray_indices"][:, 0:1],
                       metadata={"video_id": batch["video_ids"][:, None], "directions_norm": dn})
        if self.update_props_every_step:
            m.proposal_sampler._steps_since_update = 1 << 30  # proposal nets receive gradients EVERY step (upper bound of the schedule)
        out = m(rb)
        loss_dict = m.get_loss_dict(out, batch)
        loss = sum(loss_dict.values())
        (loss * self.loss_scale).backward()
        self.grads.finish_exchange()
        self.opt.step()
        m.after_train_iteration(self.step_idx)
        self.step_idx += 1
        return loss_dict, out


def cpu_baseline(seconds_budget=25.0):
    """The CPU oracle (validated restatement of the reference's torch path) on a bounded sample of the same workload:
    same field/table sizes, fewer rays.  kind = "port".  torch's CPU ops scale badly past a few dozen threads on this
    many-core host, so the thread count is the best of a short probe (reported as `cores`)."""
    from oracle import nerf_oracle as O

    cfg = O.default_config()
    scene = O.make_scene(cfg)
    P = O.make_params(cfg, seed=42)
    rays = 2048
    max_threads = torch.get_num_threads()
    O.train_step(P, cfg, scene, O.make_batch(cfg, scene, 256, step=0))  # page in the 64 + 2*32 MiB tables
    best_t, best_dt = None, float("inf")
    for th in (8, 16, 32):
        if th > max_threads:
            continue
        torch.set_num_threads(th)
        t0 = time.time()
        O.train_step(P, cfg, scene, O.make_batch(cfg, scene, 512, step=0))
        dt = time.time() - t0
        if dt < best_dt:
            best_t, best_dt = th, dt
    torch.set_num_threads(best_t or max_threads)
    t0 = time.time()
    O.train_step(P, cfg, scene, O.make_batch(cfg, scene, rays, step=1))
    first = time.time() - t0
    n = max(1, min(6, int(seconds_budget / max(first, 1e-3)) - 1))
    t0 = time.time()
    for i in range(n):
        O.train_step(P, cfg, scene, O.make_batch(cfg, scene, rays, step=2 + i))
    dt = (time.time() - t0) / n
    used = torch.get_num_threads()
    torch.set_num_threads(max_threads)
    global _PSNR_VS_ORACLE
    _PSNR_VS_ORACLE = psnr_vs_oracle(O, cfg, scene)
    return dict(value=rays / dt, unit="rays/s", cores=used, kind="port",
                sample=f"{n} full training steps (fwd + 5 losses + bwd, no optimizer) of {rays} rays, cfg-2 tables, torch-CPU oracle, "
                       f"{used} threads (best of 8/16/32; host has {os.cpu_count()} logical CPUs)")


_PSNR_VS_ORACLE = None


def psnr_vs_oracle(O, cfg, scene, rays=1024):
    """"PSNR vs ref" half of BASELINE.json's metric: the HIP model and the CPU oracle render the same rays
    (eval mode: no jitter, mean appearance code) from the same cfg-2 parameters; PSNR = 10 log10(1 / MSE) of the RGB
    difference, plus the largest semantics / expected-depth deviations.  Tables are drawn 300x wider than the init so that
    the rendering is not a constant."""
    from presight_amd import ops
    from presight_amd.model import NerfactoNuscMSModel, NerfactoNuscMSModelConfig
    from presight_amd.rays import RayBundle

    dev = torch.device("cuda", torch.cuda.current_device())
    P = O.make_params(cfg, seed=5, table_scale=0.3)
    P["field.fields.0.mlp_base_mlp.layers.1.bias"][0] = -1.0
    batch = O.make_batch(cfg, scene, rays, step=99)
    with torch.no_grad():
        ref = O.model_forward(P, cfg, scene, batch, training=False)
    m = cfg["main"]
    conf = NerfactoNuscMSModelConfig(
        near_plane=cfg["near"], far_plane=cfg["far"], piecewise_sampler_threshold=cfg["thr"], hidden_dim=m["hidden_dim"],
        hidden_dim_color=m["hidden_dim_color"], num_levels=m["num_levels"], base_res=m["base_res"], max_res=m["max_res"],
        log2_hashmap_size=m["log2_hashmap_size"], features_per_level=m["features_per_level"], use_lidar_loss=False,
        proposal_net_args_list=[dict(features_per_level=p["features_per_level"], log2_hashmap_size=p["log2_hashmap_size"],
                                     num_levels=p["num_levels"], base_res=p["base_res"], max_res=p["max_res"],
                                     hidden_dim=p["hidden_dim"], use_linear=False) for p in cfg["props"]],
        implementation="hip")
    model = NerfactoNuscMSModel(conf, num_train_cameras=cfg["num_cameras"], num_train_videos=cfg["num_videos"], dino_to_rgb=None,
                                centroids=scene["centroids"], aabbs=scene["aabbs"])
    sd = dict(model.state_dict())
    for k, v in P.items():
        sd[k] = v
        for alias in (k.replace("mlp_base_grid.", "mlp_base.0.").replace("mlp_base_mlp.", "mlp_base.1."),
                      k.replace("encoding.hash_table", "mlp_base.0.hash_table")):
            if alias in sd:
                sd[alias] = v
    model.load_state_dict(sd)
    model.to(dev).eval()
    ri = batch["ray_indices"].to(dev)
    o, d, pa, dn = ops.generate_rays(ri, *(scene[k].to(dev) for k in ("c2w", "fx", "fy", "cx", "cy")))
    with torch.no_grad():
        out = model(RayBundle(o, d, pa, camera_indices=ri[:, 0:1], metadata={"video_id": batch["video_ids"].to(dev)[:, None]}))
    mse = float(((out["rgb"].cpu() - ref["rgb"]) ** 2).mean())
    return {"psnr_db": 10.0 * __import__("math").log10(1.0 / max(mse, 1e-30)), "rays": rays,
            "rgb_std": float(ref["rgb"].std()),
            "max_abs_semantics": float((out["semantics"].cpu() - ref["semantics"]).abs().max()),
            "max_rel_expected_depth": float(((out["expected_depth"].cpu() - ref["expected_depth"]).abs()
                                             / ref["expected_depth"].abs().clamp_min(1e-6)).max())}


def main():
    if os.environ.get("PRESIGHT_HANG_DUMP"):  # debugging aid: dump every thread's stack after N seconds and exit
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["PRESIGHT_HANG_DUMP"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from presight_amd import prof
    from presight_amd.dist import init_from_env

    rank, local_rank, world = init_from_env("cuda")
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path for the product)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    model, scene = build_model(dev, seed=42)  # same init on every rank (DDP broadcast equivalent)
    trainer = Trainer(model, scene, world)
    batches = make_batches(scene, dev, 4, rank)

    for i in range(args.warmup):
        trainer.step(batches[i % len(batches)])
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    prof.enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss_dict, out = trainer.step(batches[i % len(batches)])
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern = prof.summary()
    prof.enable(False)
    # secondary figure (NOT `value`): the reference's own steady-state proposal-update schedule after warm-up
    # (ray_samplers.py:586 + nerfacto_nusc_ms.py:300-305: gradients reach the proposal nets every 6th step)
    trainer.update_props_every_step = False
    trainer.step_idx = 50000
    model.proposal_sampler._steps_since_update = 0
    n_sched = 12
    for i in range(6):
        trainer.step(batches[i % len(batches)])
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(n_sched):
        trainer.step(batches[i % len(batches)])
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt_sched = time.perf_counter() - t1
    replica_diff = None
    if world > 1:
        tmax = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
        # data-parallel consistency: every rank applied the same averaged gradients, so the replicas must still be identical
        mine = trainer.opt.flat[0]
        ref = mine.clone()
        torch.distributed.broadcast(ref, src=0)
        d = (mine - ref).abs().max().reshape(1)
        torch.distributed.all_reduce(d, op=torch.distributed.ReduceOp.MAX)
        replica_diff = float(d.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * RAYS * args.steps / dt
        N = RAYS * 64
        # dominant kernel: fused main-field backward.  Algorithmic flops per launch = dX + dW passes of the three MLPs
        # (2 x forward MACs x 2 flop) for N = 65536*64 samples; the in-kernel forward recompute is NOT counted.
        n_launch, t_bwd = kern.get("main_field_bwd", (0, float("nan")))
        flops = 2 * 2 * MAIN_MAC_PER_SAMPLE * N
        achieved = flops / (t_bwd * 1e-3) / 1e12
        psnr = float(model.get_metrics_dict(out, batches[(args.steps - 1) % len(batches)])["psnr"].detach())
        line = {
            "metric": "training rays/sec (whole node)", "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE cfg 2: one sub-tile, 16-level hash grid (F=2,T=2^19) + 64-wide MLPs, 2 proposal nets "
                                   "(L=8,F=1,T=2^20), 65536 rays/GPU/step, 128/64/64 samples, fwd+5 losses+bwd+Adam",
                       "rays_per_gpu": RAYS, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": "main_bwd_kernel (fused main-field backward)", "achieved": achieved,
                         "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": MAIN_BWD_HBM_BYTES_PMC,
                         "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/r01_pmc_summary_v6.txt)",
                         "avg_launch_ms": t_bwd, "launches": n_launch},
            "kernels_ms": {k: round(v[1], 4) for k, v in sorted(kern.items())},
            "value_reference_schedule": world * RAYS * n_sched / dt_sched,
            "replicas_max_abs_diff": replica_diff,
            "psnr_vs_random_targets": psnr,
            "loss": float(sum(v.detach() for v in loss_dict.values())),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["speedup_vs_cpu"] = value / line["cpu_baseline"]["value"]
            line["psnr_vs_oracle"] = _PSNR_VS_ORACLE
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
