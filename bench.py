"""bench.py — training rays/sec of the PreSight NeRF prior-builder hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|extract] [--scaling weak|strong]

With --gpus N > 1 and no torchrun environment the script starts its own N worker processes (one per GPU, RCCL) through
`python -m torch.distributed.run` BEFORE touching the GPU, relays rank 0's JSON line and exits with the workers' code; under
an external `torchrun` (RANK / WORLD_SIZE set) it is a worker.  A mismatch between --gpus and WORLD_SIZE is an error.

One "step" = one full training iteration of NerfactoNuscMSModel on one synthetic nuScenes-shaped ray batch: ray generation ->
proposal sampling (2 proposal fields, 2 PDF resamplings) -> main field -> compositing -> sky -> 5 losses -> full backward
(proposal nets updated every step) -> gradient exchange (N > 1) -> Adam.  Inputs are resident in HBM before the timed region.

  --config cfg2 (default; BASELINE.json configs[1], the configuration the metric is quoted on at N = 1): one Boston-Seaport
      sub-tile, 16-level hash grid (F=2, T=2^19) + 64-wide MLPs, proposal nets L=8 F=1 T=2^20, 128/64/64 samples.
  --config cfg3 (BASELINE.json configs[2]): the production tile — K = 16 sub-fields, main tables L=10 F=4 T=2^20 (940 M
      parameters), nearest-centroid routing; default scaling "strong" (65 536 rays per step over all ranks = 8192 per GPU at
      N = 8, as ns/data/PreSight/my_datamanager.py:203-212 splits them), sharded gradient exchange.
  --config extract (BASELINE.json configs[4]): prior extraction of one tile, 512^3 lattice (see presight_amd/extract.py).
  --scaling weak: every rank trains on its own 65 536-ray batch (default for cfg2); strong: 65 536 // N rays per rank.

The LAST stdout line (rank 0) is the compact line of record (< 4 KB, `compact_line`): the contract fields plus `roofline` (dominant
kernel, measured live with HIP events on the launch stream), `end_to_end` (fraction of the binding end-to-end ceiling of SURVEY.md
8d), `cpu_baseline` (the CPU oracle timed on the host cores, rank 0, N == 1 only), the headline figures of the secondary shapes and
`value_unfused_tables` (the step an N > 1 data-parallel rank runs).  The full record -- `roofline_kernels` (the MFMA-bound and
HBM-bound kernels of the step, same measurement), `kernels_ms`, the exchange dry-run tables, the full `secondary` block, the bucket
timeline of an N > 1 run -- is written to `bench_detail.json` (PRESIGHT_BENCH_DETAIL), whose path the line names."""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: without this RCCL's peer mapping fails (hipIpcGetMemHandle: invalid argument).  Already
# exported on the GPU boxes; set here too so that a worker started by an external torchrun from a bare environment still has it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

RAYS = 65536
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
XGMI_LINK_GBS = 153.0          # MI355X_MICROARCH.md: xGMI, point-to-point, ~153 GB/s per link and direction, 7 links per GPU

CONFIGS = {
    # iNGPField constructor defaults = BASELINE cfg 2 (ns/fields/PreSight/ingp_field.py:74-84); proposal nets nerfacto_nusc_ms.py:114-121
    "cfg2": dict(K=1, model=dict(num_levels=16, features_per_level=2, log2_hashmap_size=19, base_res=16, max_res=2048, hidden_dim=64,
                                 hidden_dim_color=64),
                 scaling="weak", exchange="allreduce",
                 workload="BASELINE cfg 2: one sub-tile, 16-level hash grid (F=2,T=2^19) + 64-wide MLPs, 2 proposal nets (L=8,F=1,T=2^20), "
                          "128/64/64 samples, fwd+5 losses+bwd+Adam"),
    # production tile: model defaults L=10,F=4,T=2^20,max 16384 (nerfacto_nusc_ms.py:88-101), num_aabbs=16 (method_configs.py:141)
    "cfg3": dict(K=16, model=dict(num_levels=10, features_per_level=4, log2_hashmap_size=20, base_res=16, max_res=16384, hidden_dim=64,
                                  hidden_dim_color=64),
                 scaling="strong", exchange="sharded",
                 workload="BASELINE cfg 3: production tile, K=16 sub-fields (L=10,F=4,T=2^20 main tables, 940 M parameters), routed, "
                          "128/64/64 samples, fwd+5 losses+bwd+Adam"),
    # BASELINE cfg 4: static + dynamic dual field.  The reference has no dynamic field: the model is the one
    # defined by oracle/dual_oracle.py -- static branch = the cfg-2 field, dynamic branch = 4-D hash grid (L=8, F=4, T=2^19,
    # resolutions 16..512 on x, y, z, t) + 64-wide flow MLP + the same MLP stack, temporal aggregation over 3 warped positions
    "cfg4": dict(K=1, model=dict(num_levels=16, features_per_level=2, log2_hashmap_size=19, base_res=16, max_res=2048, hidden_dim=64,
                                 hidden_dim_color=64),
                 dynamic=dict(dynamic_num_levels=8, dynamic_features_per_level=4, dynamic_log2_hashmap_size=19, dynamic_base_res=16,
                              dynamic_max_res=512, dynamic_hidden_dim=64, dynamic_hidden_dim_color=64, flow_hidden_dim=64),
                 scaling="weak", exchange="allreduce",
                 workload="BASELINE cfg 4: static (cfg-2 field) + dynamic (4-D hash grid L=8,F=4,T=2^19 + flow MLP 32-64-64-6 + MLP stack, "
                          "3-position temporal aggregation) dual field, density-weighted blend, 2 static proposal nets, 128/64/64 samples, "
                          "fwd+6 losses+bwd+Adam"),
    # the same dual field on the SG-Onenorth tile shape (ns/configs/method_configs.py:271-367: num_aabbs = 16, production grids): the
    # static branch is the routed K = 16 production field of cfg 3, the dynamic branch one field over the union of the sub-field boxes
    "cfg4prod": dict(K=16, model=dict(num_levels=10, features_per_level=4, log2_hashmap_size=20, base_res=16, max_res=16384, hidden_dim=64,
                                      hidden_dim_color=64),
                     dynamic=dict(dynamic_num_levels=8, dynamic_features_per_level=4, dynamic_log2_hashmap_size=19, dynamic_base_res=16,
                                  dynamic_max_res=512, dynamic_hidden_dim=64, dynamic_hidden_dim_color=64, flow_hidden_dim=64),
                     scaling="strong", exchange="sharded",
                     workload="BASELINE cfg 4 on the SG-Onenorth tile shape: static branch = K=16 routed production field (L=10,F=4,T=2^20), dynamic "
                              "branch = 4-D hash grid L=8,F=4,T=2^19 + flow MLP + MLP stack over the tile, 2 routed proposal nets, 128/64/64 samples, "
                              "fwd+6 losses+bwd+Adam"),
}


# --------------------------------------------------------------------------------------------------------- line of record
LINE_BUDGET_BYTES = 4096  # the driver keeps a bounded tail of stdout: the LAST stdout line must stay far below it (tests/test_host_logic.py)
DETAIL_FILE = os.environ.get("PRESIGHT_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))


def _r(x, digits=4):
    """floats of the compact line are rounded to `digits` SIGNIFICANT digits (the detail file keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def _pick(d, keys):
    return None if not isinstance(d, dict) else {k: d[k] for k in keys if k in d}


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 3] + "..."


def compact_line(full: dict, detail_path=None) -> dict:
    """The line of record: the contract fields + `roofline` + `cpu_baseline` + the handful of headline figures, nothing that grows with
    the number of buckets / kernels / secondary shapes.  Everything else (`roofline_kernels`, `kernels_ms`, the exchange dry-run tables,
    the full `secondary` block, the bucket timeline of an N > 1 run) lives in the detail file the line names."""
    out = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    cfg = dict(full.get("config") or {})
    cfg["workload"] = _clip(cfg.get("workload"), 330)
    out["config"] = {k: cfg[k] for k in ("workload", "rays_per_gpu", "points_per_gpu", "rays_per_step_global", "parallelism", "exchange")
                     if cfg.get(k) is not None}
    roof = full.get("roofline")
    if roof is not None:
        r = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_algorithmic", "avg_launch_ms", "launches",
                         "bytes_per_launch", "flops_per_launch", "traffic_source", "semantic_head_tiles_executed", "traffic_over_algorithmic_io",
                         "kept_activation_bytes", "traffic_TBps"))
        r["kernel"] = _clip(r.get("kernel"), 120)
        r["traffic_source"] = _clip(r.get("traffic_source"), 100)
        if isinstance(roof.get("gather"), dict):
            r["gather_frac_of_l2_resident_ceiling"] = roof["gather"].get("frac_of_l2_resident_ceiling")
        out["roofline"] = r
    rm = full.get("roofline_mfma")
    if isinstance(rm, dict):  # the matrix-bound kernel next to the dominant one (the dominant kernel may be an HBM-bound one)
        out["roofline_mfma"] = _pick(rm, ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_algorithmic", "avg_launch_ms", "traffic",
                                          "traffic_over_algorithmic_io", "kept_activation_bytes"))
    if isinstance(full.get("end_to_end"), dict):
        out["end_to_end"] = _pick(full["end_to_end"], ("binding", "frac_of_binding", "frac_of_mfma", "frac_of_hbm", "frac_of_hbm_incl_optimizer"))
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind"))
        c["sample"] = _clip(cb.get("sample"), 220)
        if isinstance(cb.get("host"), dict):
            c["cpu_model"], c["physical_cores"] = cb["host"].get("cpu_model"), cb["host"].get("physical_cores")
        out["cpu_baseline"] = c
    for k in ("speedup_vs_cpu", "value_reference_schedule", "value_unfused_tables", "ms_per_step_unfused_tables", "replicas_max_abs_diff",
              "psnr_vs_random_targets", "loss", "kept_points_rank0", "voxels_rank0", "launches_per_step", "half_batch_pipeline"):
        if full.get(k) is not None:
            out[k] = full[k]
    if isinstance(full.get("optimizer"), dict):
        out["fused_table_adam"] = full["optimizer"].get("fused_table_adam")
    if isinstance(full.get("psnr_vs_oracle"), dict):
        out["psnr_vs_oracle"] = _pick(full["psnr_vs_oracle"], ("psnr_db", "max_abs_semantics", "max_rel_expected_depth"))
    pk = full.get("psnr_after_k_steps")
    if isinstance(pk, dict):
        out["psnr_after_k_steps"] = _pick(pk, ("K", "psnr_final_db", "psnr_gain_db", "monotone", "error"))
        if "psnr_db" in pk:
            out["psnr_after_k_steps"]["psnr_first_db"] = pk["psnr_db"][0]
    sec = full.get("secondary")
    if isinstance(sec, dict):
        out["secondary"] = {}
        for name, e in sec.items():
            if not isinstance(e, dict):
                continue
            c = _pick(e, ("ms_per_step", "value", "unit", "frac_of_binding", "binding"))
            if "error" in e:
                c["error"] = _clip(e["error"], 120)
            if isinstance(e.get("exchange_overlap_dry_run"), dict) and "ms_per_step_with_split_launches" in e["exchange_overlap_dry_run"]:
                c["ms_per_step_unfused_tables"] = e["exchange_overlap_dry_run"]["ms_per_step_with_split_launches"]
            for k, v in e.items():  # (PREDICTED, never measured: one-GPU measurements + link assumptions, predicted_strong_scaling)
                if k.startswith("predicted_N") and isinstance(v, dict) and isinstance(v.get("all_links_ASSUMED"), dict):
                    c[k] = {"step_ms_all_links_ASSUMED": v["all_links_ASSUMED"]["predicted_step_ms"],
                            "speedup_all_links_ASSUMED": v["all_links_ASSUMED"]["predicted_speedup_vs_one_gpu"],
                            "speedup_ring_one_link": v["ring_one_link"]["predicted_speedup_vs_one_gpu"], "measured": False}
                    sp = v.get("sparse_records")
                    if isinstance(sp, dict) and isinstance(sp.get("all_links_ASSUMED"), dict):  # (the same prediction with the tables travelling as records)
                        c[k]["sparse_speedup_all_links_ASSUMED"] = sp["all_links_ASSUMED"]["predicted_speedup_vs_one_gpu"]
                        c[k]["sparse_speedup_ring_one_link"] = sp["ring_one_link"]["predicted_speedup_vs_one_gpu"]
            out["secondary"][name] = c
    other = full.get("other_scaling")
    if isinstance(other, dict):
        out["other_scaling"] = _pick(other, ("scaling", "rays_per_gpu", "value", "ms_per_step"))
    comm = full.get("comm")
    if isinstance(comm, dict):
        c = _pick(comm, ("backend", "ranks", "collectives_per_step", "gradient_buckets_issued_during_backward_per_step",
                         "record_bytes_on_link_per_rank_per_step", "dense_table_gradient_bytes",
                         "bytes_on_link_per_rank_per_step", "exchange_exposed_ms"))
        if isinstance(comm.get("model"), dict):
            c["predicted_ms_all_links_ASSUMED"] = comm["model"].get("predicted_ms_all_links")
            c["predicted_ms_ring_one_link"] = comm["model"].get("predicted_ms_ring_one_link")
        sched = comm.get("schedule_by_construction")
        if isinstance(sched, dict):
            c["exposed_ms_by_construction"] = {k: v.get("exposed_ms") for k, v in sched.items() if isinstance(v, dict)}
        out["comm"] = c
    if detail_path:
        out["detail"] = detail_path
    out = _r(out)
    for k in ("value", "ms_per_step"):  # the figures of record keep their precision
        out[k] = full.get(k)
    return out


def emit(full: dict):
    """write the full record to the detail file and print the compact line of record as the LAST stdout line"""
    path = None
    try:
        os.makedirs(os.path.dirname(DETAIL_FILE) or ".", exist_ok=True)
        with open(DETAIL_FILE, "w") as f:
            json.dump(full, f)
        path = os.path.relpath(DETAIL_FILE, ROOT) if DETAIL_FILE.startswith(ROOT) else DETAIL_FILE
    except OSError as e:  # a read-only tree must not take the line down
        print(f"bench.py: detail file not written ({e})", file=sys.stderr)
    line = compact_line(full, path)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_BUDGET_BYTES:  # never happens by construction (test_bench_line_is_compact); degrade instead of losing the record
        for k in ("secondary", "comm", "other_scaling", "psnr_after_k_steps", "psnr_vs_oracle"):
            line.pop(k, None)
        text = json.dumps(line, separators=(",", ":"))
    sys.stdout.flush()
    print(text)
    sys.stdout.flush()


# --------------------------------------------------------------------------------------------------------- launcher
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg4", "cfg4prod", "extract"], default="cfg2")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None)
    ap.add_argument("--exchange", choices=["allreduce", "sharded", "sparse"], default=None,
                    help="sparse: the sharded exchange with the hash tables' gradients travelling as records of touched rows (SURVEY.md 8e)")
    ap.add_argument("--rays", type=int, default=RAYS, help="rays per step: per GPU (weak) / over all GPUs (strong)")
    ap.add_argument("--global-depth-clip", action="store_true", help="expected-depth clip bounds over ALL ranks' batches")
    ap.add_argument("--fixed-batches", action="store_true", help="recycle 4 pre-made batches instead of the device chunk feed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short cfg3 / cfg4 / extraction measurements of the default line")
    ap.add_argument("--parallelism", choices=["dp", "tiles"], default="dp",
                    help="dp: data-parallel training of ONE tile over the ranks (gradient exchange every step); tiles: one independent tile "
                         "per rank, no exchange at all -- how the reference builds a city (docs/building_priors.md:7-44: one ns-train per tile)")
    ap.add_argument("--extract-model", choices=["cfg2", "cfg3"], default="cfg3",
                    help="--config extract: the fields that are queried -- cfg3 = the production tile (K = 16 routed sub-fields, L10 F4 T2^20), cfg2 = one sub-field")
    ap.add_argument("--psnr-steps", type=int, default=300, help="iterations of the learnable-scene training run behind `psnr_after_k_steps` (0: skip)")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """--gpus N > 1 outside torchrun: start N ranks (one per GPU) and relay their output.  Nothing here touches the GPU
    (torch.cuda.device_count() does not initialise it), so the children are ordinary fresh processes."""
    import torch

    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and os.environ.get("PRESIGHT_SINGLE_DEVICE") != "1":
        print(f"bench.py: --gpus {args.gpus} but this node has {n_dev} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# --------------------------------------------------------------------------------------------------------- workload
def build_model(dev, seed, config="cfg2", **overrides):
    import torch

    from presight_amd.model import NerfactoNuscMSModel, NerfactoNuscMSModelConfig

    torch.manual_seed(seed)
    c = CONFIGS[config]
    common = dict(near_plane=0.005, far_plane=50.0, piecewise_sampler_threshold=5.0, implementation="hip", use_lidar_loss=False,
                  proposal_weights_anneal_max_num_iters=10000, proposal_warmup=10000, **c["model"])
    common.update(overrides)
    scene = make_scene(1440, 6, K=c["K"])
    if "dynamic" in c:
        from presight_amd.dynamic import NerfactoNuscDualModel, NerfactoNuscDualModelConfig

        conf = NerfactoNuscDualModelConfig(time_step=1.0 / (1440 // 6 - 1), **common, **c["dynamic"])
        cls = NerfactoNuscDualModel
    else:
        conf, cls = NerfactoNuscMSModelConfig(**common), NerfactoNuscMSModel
    model = cls(conf, num_train_cameras=1440, num_train_videos=6, dino_to_rgb=None, centroids=scene["centroids"], aabbs=scene["aabbs"])
    model.to(dev)
    return model, {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}


def make_scene(num_cameras, num_videos, seed=7, K=1):
    """Synthetic nuScenes-shaped rig: 6 pinhole cameras 1600x900 moving along a polyline at 0.5 m/frame,
    poses scaled by 0.05 and mean-centred.  K = 1: one sub-field whose AABB is the trajectory quantile box +-15 m;
    K > 1: K centroids on the polyline, one AABB (+-45 m) around each (the reference clusters the poses with k-means,
    ns/data/dataparsers/mynuscenes_ms_dataparser.py:231-269)."""
    import math

    import torch

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
    if K == 1:
        lo = c2w[:, :, 3].quantile(0.02, dim=0) - ext
        hi = c2w[:, :, 3].quantile(0.98, dim=0) + ext
        centroids, aabbs = c2w[C // 2: C // 2 + 1, :, 3].clone(), torch.stack([lo, hi])[None]
    else:
        sel = torch.linspace(0, C - 1, K + 2)[1:-1].long()
        centroids = c2w[sel, :, 3].clone()
        aabbs = torch.stack([torch.stack([c - 3 * ext, c + 3 * ext]) for c in centroids])
    # normalised timestamp of every camera (frame-major, 6 cameras per frame): what the dual field of cfg 4 reads as ray time
    cam_times = (torch.arange(C) // 6).float() / float(max(1, n_frames - 1))
    return dict(c2w=c2w, fx=fx, fy=fx.clone(), cx=torch.full((C,), 800.0), cy=torch.full((C,), 450.0), centroids=centroids, aabbs=aabbs,
                H=900, W=1600, frames_per_video=max(1, C // num_videos), cam_times=cam_times)


def make_batches(scene, dev, n_batches, rank, rays=RAYS):
    """uniform-random ray indices + random targets, manual_seed(1234 + step) (+1000*rank: every DP rank draws its own rays,
    as the reference seeds its workers with seed + rank, ns/scripts/train.py:99)"""
    import torch

    out = []
    C = scene["c2w"].shape[0]
    for step in range(n_batches):
        g = torch.Generator(device="cpu").manual_seed(1234 + step + 1000 * rank)
        idx = torch.stack([torch.randint(0, C, (rays,), generator=g), torch.randint(0, scene["H"], (rays,), generator=g),
                           torch.randint(0, scene["W"], (rays,), generator=g)], -1)
        out.append(dict(ray_indices=idx.to(dev), video_ids=torch.clamp(idx[:, 0] // scene["frames_per_video"], max=5).to(dev),
                        times=scene["cam_times"].cpu()[idx[:, 0]].to(dev),
                        rgb=torch.rand(rays, 3, generator=g).to(dev), features=torch.rand(rays, 64, generator=g).to(dev),
                        sky=(torch.rand(rays, generator=g) < 0.15).float().to(dev)))
    return out


def synthetic_chunk(scene, dev, chunk_index, pixels=1 << 22):
    """One chunk of the training set as the reference keeps it in memory (ns/data/PreSight/my_dataset.py:28-50: flat per-pixel
    arrays of the pixels that survived the masks, `chunk_ratio` of every loaded image): uniformly drawn (image, pixel) slots with
    random targets, generated on the device (the background loader thread of presight_amd.datafeed.ChunkFeed calls this)."""
    import torch

    g = torch.Generator(device=dev).manual_seed(4321 + chunk_index)
    C, H, W = scene["c2w"].shape[0], scene["H"], scene["W"]
    img = torch.randint(0, C, (pixels,), device=dev, generator=g)
    return dict(rgbs=torch.rand(pixels, 3, device=dev, generator=g), pixel_indices=torch.randint(0, H * W, (pixels,), device=dev, generator=g),
                image_indices=img, video_ids=torch.clamp(img // scene["frames_per_video"], max=5),
                widths=torch.full((pixels,), W, device=dev, dtype=torch.int64), skies=(torch.rand(pixels, device=dev, generator=g) < 0.15).float(),
                depths=None, features=torch.rand(pixels, 64, device=dev, generator=g))


def Trainer(model, scene, world, exchange="allreduce", global_depth_clip=False, **kw):
    """presight_amd.trainer.Trainer (the timed region) with the bench's setting: proposal networks receive gradients EVERY step
    (the upper bound of the reference's update schedule; its steady state is reported as `value_reference_schedule`)"""
    from presight_amd.trainer import Trainer as _Trainer

    pipelined = world == 1 and os.environ.get("PRESIGHT_PIPELINE_ADAM", "0") == "1"
    if pipelined:
        kw.setdefault("fused_table_adam", False)  # (the pipelined optimizer step is the alternative to the fused table update)
    t = _Trainer(model, scene, world, exchange=exchange, global_depth_clip=global_depth_clip, **kw)
    t.update_props_every_step = True
    # PRESIGHT_PIPELINE_ADAM=1 (single process): the fields' Adam on a second stream underneath the next step's proposal sampling
    # (presight_amd/trainer.py).  Measured and left OFF: on the production tile the sampling front is HBM-bound itself (512 MB of
    # proposal tables per net) and shares the memory system with the 26 GB Adam stream -- cfg 3 26.3 -> 27.2 ms at 65 536 rays,
    # 10.5 -> 10.2 ms at 8192; cfg 2 (0.15 ms of Adam) unchanged
    t.pipeline_adam = pipelined
    return t


# --------------------------------------------------------------------------------------------------------- CPU baseline
def host_cpu_info():
    """(physical cores, model name) from /proc/cpuinfo"""
    cores, model = set(), "unknown"
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return (len(cores) or os.cpu_count() or 1), model


def cpu_baseline():
    """SURVEY.md 8d protocol: the CPU oracle (validated restatement of the reference's torch path, kind = "port") runs the
    SAME step as the GPU side — forward, 5 losses, backward AND torch.optim.Adam (lr 1e-2, eps 1e-15, wd 1e-5) — on a bounded
    sample of the workload (cfg-2 tables and networks, 2048-ray batches instead of 65 536): 3 warm-up + 10 timed steps,
    median.  torch's CPU kernels stop scaling long before this host's core count, so the thread count is probed (one
    512-ray step each at 8/16/32/64/128/#physical threads) and the best one is used and reported."""
    import statistics

    import torch

    from oracle import nerf_oracle as O

    cfg = O.default_config()
    scene = O.make_scene(cfg)
    P = O.make_params(cfg, seed=42)
    for v in P.values():
        v.requires_grad_(True)
    opt = torch.optim.Adam(list(P.values()), lr=1e-2, eps=1e-15, weight_decay=1e-5)
    rays = 2048
    phys, model_name = host_cpu_info()
    max_threads = torch.get_num_threads()

    def one_step(n_rays, step):
        opt.zero_grad(set_to_none=True)
        L, _, g = O.train_step(P, cfg, scene, O.make_batch(cfg, scene, n_rays, step=step))
        for k, v in P.items():
            v.grad = g.get(k)
        opt.step()

    one_step(256, 0)  # page in the 64 + 2*32 MiB tables and the optimizer state
    best_t, best_dt = None, float("inf")
    probe = sorted({t for t in (8, 16, 32, 64, 128, phys) if t <= max(phys, max_threads)})
    probe_log = {}
    for th in probe:
        torch.set_num_threads(th)
        t0 = time.time()
        one_step(512, 0)
        dt = time.time() - t0
        probe_log[th] = round(512 / dt, 1)
        if dt < best_dt:
            best_t, best_dt = th, dt
    torch.set_num_threads(best_t or max_threads)
    for i in range(3):
        one_step(rays, 1 + i)
    times = []
    for i in range(10):
        t0 = time.time()
        one_step(rays, 4 + i)
        times.append(time.time() - t0)
        if sum(times) > 40.0 and len(times) >= 5:
            break  # bounded: a slow host must not stretch the default bench run past a few minutes
    dt = statistics.median(times)
    used = torch.get_num_threads()
    torch.set_num_threads(max_threads)
    global _PSNR_VS_ORACLE
    for v in P.values():
        v.requires_grad_(False)
    _PSNR_VS_ORACLE = psnr_vs_oracle(O, cfg, scene)
    return dict(value=rays / dt, unit="rays/s", cores=used, kind="port",
                sample=f"median of {len(times)} timed steps after 3 warm-up (fwd + 5 losses + bwd + torch.optim.Adam) of {rays} rays, "
                       f"cfg-2 tables, torch-CPU oracle, {used} threads",
                host=dict(cpu_model=model_name, physical_cores=phys, logical_cpus=os.cpu_count(), thread_probe_rays_per_s=probe_log))


_PSNR_VS_ORACLE = None
_HANG_FILE = None


def psnr_vs_oracle(O, cfg, scene, rays=1024):
    """"PSNR vs ref" half of BASELINE.json's metric: the HIP model and the CPU oracle render the same rays
    (eval mode: no jitter, mean appearance code) from the same cfg-2 parameters; PSNR = 10 log10(1 / MSE) of the RGB
    difference, plus the largest semantics / expected-depth deviations.  Tables are drawn 300x wider than the init so that
    the rendering is not a constant."""
    import math

    import torch

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
    return {"psnr_db": 10.0 * math.log10(1.0 / max(mse, 1e-30)), "rays": rays, "rgb_std": float(ref["rgb"].std()),
            "max_abs_semantics": float((out["semantics"].cpu() - ref["semantics"]).abs().max()),
            "max_rel_expected_depth": float(((out["expected_depth"].cpu() - ref["expected_depth"]).abs()
                                             / ref["expected_depth"].abs().clamp_min(1e-6)).max())}


# --------------------------------------------------------------------------------------------------------- training quality
def psnr_after_k_steps(dev, config="cfg2", K=300, rays=RAYS, marks=6):
    """"PSNR vs synthetic GT after K steps"  -- the quality half of BASELINE.json's metric as a TRAINING figure.  A
    complete miniature run of the reference's loop (presight_amd.trainer.Trainer with max_num_iterations = K: anneal and proposal
    warm-up over K // 10, LR warm-up over K // 10, x0.33 at K/4, K/2, 3K/4; the reference's proposal update schedule; Adam on the
    2**10-scaled gradients) on a LEARNABLE scene: per-pixel targets rendered by a fixed teacher parameter set of the same model
    (presight_amd/synthetic.py), 4 M training pixels served by the device chunk feed, 65 536 held-out rays.  PSNR = 10 log10(1/MSE)
    of the eval render (ns/models/PreSight/nerfacto_nusc_ms.py:548-556) at `marks` + 1 evenly spaced checkpoints."""
    import gc

    import torch

    from presight_amd.datafeed import ChunkFeed
    from presight_amd.synthetic import TEACHER_FAR, TeacherScene, eval_psnr, shape_teacher_
    from presight_amd.trainer import Trainer as _Trainer

    t0 = time.perf_counter()
    tmodel, scene = build_model(dev, seed=7, config=config, far_plane=TEACHER_FAR)
    shape_teacher_(tmodel)
    teacher = TeacherScene(tmodel, scene)
    chunk = teacher.chunk(0, pixels=1 << 22)
    g = torch.Generator(device=dev).manual_seed(99)
    C, H, W = scene["c2w"].shape[0], scene["H"], scene["W"]
    tri = torch.stack([torch.randint(0, C, (RAYS,), device=dev, generator=g), torch.randint(0, H, (RAYS,), device=dev, generator=g),
                       torch.randint(0, W, (RAYS,), device=dev, generator=g)], -1)
    tvid = torch.clamp(tri[:, 0] // scene["frames_per_video"], max=5)
    test = teacher.targets(tri, tvid)
    acc_q = [float(x) for x in torch.quantile(test["accumulation"], torch.tensor([0.05, 0.5, 0.95], device=dev))]
    t_teacher = time.perf_counter() - t0
    del teacher, tmodel
    gc.collect()
    torch.cuda.empty_cache()
    model, scene = build_model(dev, seed=42, config=config, proposal_weights_anneal_max_num_iters=K // 10, proposal_warmup=K // 10)
    tr = _Trainer(model, scene, 1, max_num_iterations=K)  # the reference's proposal update schedule and LR schedule
    feed = ChunkFeed(lambda i: chunk, batch_size=rays, device=dev, world=1, rank=0)
    at = sorted({round(K * i / marks) for i in range(marks + 1)})
    psnr, rgb_loss = [], []
    t1 = time.perf_counter()
    for step in range(K + 1):
        if step in at:
            psnr.append(eval_psnr(model, scene, tri, tvid, test["rgb"]))
        if step == K:
            break
        ld, _ = tr.step(feed.next_batch())
        if step in (0, K - 1):
            rgb_loss.append(float(ld["rgb_loss"].detach()))
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t1
    feed.close()
    res = {"workload": f"{K}-iteration run of the reference loop (max_num_iterations={K}) on the teacher-rendered scene, {rays} rays/step, "
                       f"{config} model, 4 M training pixels, 65 536 held-out rays", "K": K, "at_steps": at,
           "psnr_db": [round(x, 3) for x in psnr], "psnr_final_db": psnr[-1], "psnr_gain_db": psnr[-1] - psnr[0],
           "monotone": all(b >= a - 0.25 for a, b in zip(psnr, psnr[1:])), "rgb_loss_first_last": rgb_loss,
           "teacher": {"accumulation_q05_q50_q95": [round(x, 3) for x in acc_q], "sky_fraction": float(test["sky"].mean()),
                       "rgb_std": float(test["rgb"].std())},
           "seconds": {"teacher_render": round(t_teacher, 2), "train_and_eval": round(t_train, 2)}}
    del tr, model, feed, chunk, test
    gc.collect()
    torch.cuda.empty_cache()
    return res


# --------------------------------------------------------------------------------------------------------- roofline
def pmc_traffic(kernel_key: str):
    """HBM bytes per launch of a kernel from the committed PMC passes (profiles/traffic.json, written by
    tools/pmc_traffic.py from separate --pmc FETCH_SIZE / WRITE_SIZE passes of this very command, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for 16-byte streaming reads on gfx950).  The entry carries the hash of the kernel
    sources it was measured on: after the kernels change the figure is stale and None is reported instead."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(path))
        ent = rec["kernels"][kernel_key]
    except (OSError, KeyError, ValueError):
        return None, "no PMC record"
    if rec.get("src_sha16") != kernel_sources_sha():
        return None, f"stale: {os.path.relpath(path, ROOT)} was measured on other kernel sources"
    return float(ent["hbm_bytes_per_launch"]), ent.get("source", "profiles/traffic.json")


def kernel_sources_sha() -> str:
    h = hashlib.sha256()
    d = os.path.join(ROOT, "presight_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def roofline_entries(kern, cfg, rays, n_params=0, live=(), fused_tables=None):
    """per-kernel roofline rows from the HIP-event regions of presight_amd.prof (mean ms per launch over the timed steps).
    Algorithmic work: MLP flops = 2 x MACs (backward = 2 x forward: dX + dW); hash bytes = one
    F*4-byte row per corner, 8 corners per (point, level), gathered once forward, read + written once backward.
    fused_tables = (main table entries, mean entries of one proposal net's tables) when the tables' Adam step runs inside their table
    backward (Trainer.fused_table_adam): those launches also stream p, m, v in and out (24 bytes per entry) and the optimizer kernel
    covers the remaining parameters only."""
    m = cfg["model"]
    L, F = m["num_levels"], m["features_per_level"]
    n_main, n_p0, n_p1 = rays * 64, rays * 128, rays * 64
    ft_main, ft_prop = fused_tables if fused_tables else (0, 0)
    mac_main = (L * F) * 64 + 64 * 80 + 3 * 64 * 64 + (47 * 64 + 64 * 64 + 64 * 3)  # base + semantic head + colour head
    mac_prop = 8 * 64 + 64
    rows = []

    def add(name, region, bound, work, unit, executed=None, extra_regions=()):
        """`frac` is what the hardware DID: matrix-core flops actually issued (padding included) / duration / peak for the MFMA rows --
        never above 1 by construction -- and algorithmic bytes / duration / peak for the HBM rows.  `frac_algorithmic` prices the
        flops the reference's unfactored network defines against the same duration: it exceeds `frac` where the
        factored kernels do the same job with fewer matrix operations, and can exceed 1.  extra_regions: small launches that carry
        work the kernel used to do (per-ray layers of the factored path); their time is added to the row's duration."""
        if region not in kern:
            return
        n, ms = kern[region]
        if not ms > 0:
            return
        extra = {r: kern[r][1] * kern[r][0] / n for r in extra_regions if r in kern and kern[r][1] > 0}
        ms_row = ms + sum(extra.values())
        peak = FP32_MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS
        div = 1e12 if bound == "mfma" else 1e9
        done = work if executed is None else executed
        row = dict(kernel=name, bound=bound, achieved=done / (ms_row * 1e-3) / div, peak=peak, unit=unit, frac=done / (ms_row * 1e-3) / div / peak,
                   avg_launch_ms=ms, launches=n, timed_region=region in live)
        if bound == "mfma":
            row.update(executed=done, algorithmic=work, frac_algorithmic=work / (ms_row * 1e-3) / div / peak)
        else:
            row.update(algorithmic=work)
        if extra:
            row.update(duration_ms_incl_moved_work=ms_row, moved_work_ms={k: round(v, 4) for k, v in extra.items()})
        rows.append(row)

    mac_base, mac_sem, mac_rgb = (L * F) * 64 + 64 * 80, 3 * 64 * 64, 47 * 64 + 64 * 64 + 64 * 3
    # The factored render node (presight_amd/field_ops.py _MainFieldRenderF) runs the SAME function with fewer
    # matrix operations per sample: base layer 1 ends in 16 outputs, the semantic stack is two 64x64 layers (first one merged,
    # output layer per ray), the colour head's first layer sees the 16 base outputs only (direction / appearance term per ray).
    # `executed` counts the MACs the MFMA units perform, padding included (outputs padded to 16 rows).
    from presight_amd import field_ops
    fact = field_ops.FACTORED and "dynamic" not in cfg and cfg["K"] == 1
    LFp = (L * F + 3) // 4 * 4
    if fact:
        ex_base, ex_sem, ex_rgb = LFp * 64 + 64 * 16, 2 * 64 * 64, 16 * 64 + 64 * 64 + 64 * 16
    else:  # the unfactored kernels: inputs padded to whole k-steps (47 -> 48), outputs to whole 16-row blocks (3 -> 16)
        ex_base, ex_sem, ex_rgb = LFp * 64 + 64 * 80, 3 * 64 * 64, 48 * 64 + 64 * 64 + 64 * 16
    ex = lambda macs, k: 2 * k * macs * n_main  # noqa: E731
    add("main_fwd_kernel", "main_field_fwd", "mfma", 2 * mac_main * n_main, "TFLOP/s", ex(ex_base + ex_sem + ex_rgb, 1),
        extra_regions=("sem_out_fwd", "ray_colour_fwd", "merge_linear_fwd") if fact else ())
    n_adam = n_params - (ft_main + 2 * ft_prop)
    if n_params and (not fused_tables or n_adam > 20e6):  # dense Adam: p, m, v read + written, g read = 28 bytes per parameter
        add("adam_ranges_kernel" + (" (all parameters but the hash tables)" if fused_tables else ""), "adam", "hbm", 28.0 * n_adam, "GB/s")
    # the main backward is three kernels (semantic head, colour head, base MLP), timed one by one
    # (the merged layer's own rows of base layer 1 are priced with the semantic head in the factored split, as the kernels run them)
    add("main_bwd_sem_kernel", "main_bwd_sem_kernel", "mfma", 2 * 2 * mac_sem * n_main, "TFLOP/s", ex(ex_sem, 2),
        extra_regions=("sem_out_bwd", "merge_linear_bwd") if fact else ())
    add("main_bwd_rgb_kernel", "main_bwd_rgb_kernel", "mfma", 2 * 2 * mac_rgb * n_main, "TFLOP/s", ex(ex_rgb, 2),
        extra_regions=("ray_colour_bwd",) if fact else ())
    add("main_bwd_base_kernel", "main_bwd_base_kernel", "mfma", 2 * 2 * mac_base * n_main, "TFLOP/s", ex(ex_base, 2))
    add("main backward (the three kernels above, summed)", "main_field_bwd", "mfma", 2 * 2 * mac_main * n_main, "TFLOP/s",
        ex(ex_base + ex_sem + ex_rgb, 2), extra_regions=("sem_out_bwd", "merge_linear_bwd", "ray_colour_bwd") if fact else ())
    # proposal MLP 8 -> 64 -> 1: the 64 x 8 layer runs on the matrix cores, the scalar head on the vector ALU (512 of 576 MACs executed as MFMA)
    add("prop_bwd_kernel (both fields)", "prop_field_bwd", "mfma", 2 * 2 * mac_prop * (n_p0 + n_p1) / 2, "TFLOP/s", 2 * 2 * 512 * (n_p0 + n_p1) / 2)
    add("prop_fwd_kernel (both fields)", "prop_field_fwd", "mfma", 2 * mac_prop * (n_p0 + n_p1) / 2, "TFLOP/s", 2 * 512 * (n_p0 + n_p1) / 2)
    add(f"grid_encode main (L{L} F{F})", f"grid_encode_L{L}F{F}", "hbm", n_main * L * 8 * F * 4, "GB/s")
    add("grid_encode proposal (L8 F1, mean of both)", "grid_encode_L8F1", "hbm", (n_p0 + n_p1) / 2 * 8 * 8 * 4, "GB/s")
    tag = " + Adam of the table" if fused_tables else ""
    add(f"table backward main (absmax+bin+accumulate{tag}, L{L} F{F})", f"grid_scatter_L{L}F{F}", "hbm",
        2 * n_main * L * 8 * F * 4 + 24.0 * ft_main, "GB/s")
    # its two long kernels one by one (single-process training: field_ops._scatter_phases): the record writer reads one d(feature) plane
    # entry per (point, level) and writes 4 x-pair records of (2 + F) words each; the accumulate pass reads those records and streams
    # the table's p, m, v in and out (24 bytes per entry) -- the gradient itself never reaches memory.  Algorithmic bytes; the measured
    # HBM bytes (PMC) are `roofline.traffic` when this kernel is the dominant one.
    rec = n_main * L * 4 * (2 + F) * 4
    add(f"bin_kernel<{F}> (main table record writer)", f"bin_kernel_L{L}F{F}", "hbm", n_main * L * F * 4 + rec, "GB/s")
    add(f"accumulate_kernel<{F}> (main table{tag})", f"accumulate_kernel_L{L}F{F}", "hbm", rec + 24.0 * ft_main, "GB/s")
    add(f"table backward proposal (bin+accumulate{tag}, L8 F1, mean of both)", "grid_scatter_L8F1", "hbm",
        2 * (n_p0 + n_p1) / 2 * 8 * 8 * 4 + 24.0 * ft_prop, "GB/s")
    return rows


def binding_ceiling(cfg, rays, n_params, fused_tables=True):
    """-> dict: the end-to-end ceilings of one training step in rays/s.  SURVEY.md 8d prices MLP flops against the fp32 matrix peak
    and hash-table bytes against HBM, per RAY; a step also streams the optimizer state once whatever the ray count -- p, m, v read and
    written (24 bytes per parameter; + the gradient's write, read and zero fill = 36 when the tables' update is not fused into their
    backward).  On the production tile (940 M parameters) that is 22.6 GB per step, more than the per-ray hash traffic of 65 536 rays:
    the HBM ceiling INCLUDING the optimizer stream is the binding one there, and the line says so instead of quoting the MFMA ceiling."""
    flop_ray, byte_ray = end_to_end_ceilings(cfg)
    opt_bytes = (24.0 if fused_tables else 36.0) * n_params
    ceil_mfma = FP32_MFMA_PEAK_TFLOPS * 1e12 / flop_ray
    ceil_hbm = HBM_PEAK_GBS * 1e9 / byte_ray
    ceil_hbm_opt = HBM_PEAK_GBS * 1e9 / (byte_ray + opt_bytes / max(rays, 1))
    binding = "mfma" if ceil_mfma <= ceil_hbm_opt else "hbm (hash traffic + optimizer stream)"
    return {"flop_per_ray": flop_ray, "hash_bytes_per_ray": byte_ray, "optimizer_bytes_per_step": opt_bytes, "ceiling_mfma_rays_per_s": ceil_mfma,
            "ceiling_hbm_rays_per_s": ceil_hbm, "ceiling_hbm_incl_optimizer_rays_per_s": ceil_hbm_opt, "binding": binding,
            "ceiling_binding_rays_per_s": min(ceil_mfma, ceil_hbm_opt)}


def main_fwd_io(cfg, rays, traffic, launch_ms):
    """algorithmic I/O and kept-activation bytes of one main-forward launch (64 samples per ray), next to the measured HBM traffic"""
    from presight_amd import field_ops

    m = cfg["model"]
    n = rays * 64
    lf = m["num_levels"] * m["features_per_level"]
    fact = field_ops.FACTORED and "dynamic" not in cfg and cfg["K"] == 1
    io = n * (lf * 4 + 4) + (n * (4 + 12 + 4) + rays * 64 * 4 if fact else n * (4 + 12 + 256))
    try:
        act_w = field_ops._main_spec_f(lf, m["hidden_dim"], m["hidden_dim_color"], 16).act_width if fact else None
    except Exception:
        act_w = None
    kept = None if act_w is None else n * act_w * 4
    out = {"io_bytes_algorithmic": io, "kept_activation_bytes": kept}
    if traffic:
        out["traffic_over_algorithmic_io"] = traffic / io
        if launch_ms:
            out["traffic_TBps"] = traffic / (launch_ms * 1e-3) / 1e12
    return out


def end_to_end_ceilings(cfg):
    """(flop_per_ray, hash_bytes_per_ray) of one training ray, SURVEY.md 8d: MLP flops (fwd + 2x bwd) and hash-table bytes (gather
    fwd, read + write bwd); proposal nets counted every step"""
    m = cfg["model"]
    L, F = m["num_levels"], m["features_per_level"]
    mac_main = (L * F) * 64 + 64 * 80 + 3 * 64 * 64 + (47 * 64 + 64 * 64 + 64 * 3)
    flop_ray = 3 * 2 * (64 * mac_main + 192 * 576)
    byte_ray = 3 * (64 * L * 8 * F * 4 + 192 * 8 * 8 * 4)
    if "dynamic" in cfg:  # cfg 4: the dynamic branch's MLP stack + flow MLP, 3 position sets x 16 corners of its 4-D grid
        dy = cfg["dynamic"]
        Ld, Fd = dy["dynamic_num_levels"], dy["dynamic_features_per_level"]
        mac_dyn = (Ld * Fd) * 64 + 64 * 80 + 3 * 64 * 64 + (47 * 64 + 64 * 64 + 64 * 3) + (Ld * Fd * 64 + 64 * 64 + 64 * 6)
        flop_ray += 3 * 2 * 64 * mac_dyn
        byte_ray += 3 * 64 * (3 * Ld * 16 * Fd * 4)
    return flop_ray, byte_ray


def secondary_training_lines(config, shapes, dev):
    """Short single-GPU measurements of another BASELINE configuration for the `secondary` block of the default line (so that the
    driver's own run carries them): per (rays, steps, warmup) in `shapes` -> ms/step, rays/s, the three longest kernel regions,
    fraction of the binding ceiling.  One model build for all shapes."""
    import gc

    import torch

    from presight_amd import prof

    cfg = CONFIGS[config]
    model, scene = build_model(dev, seed=42, config=config)
    trainer = Trainer(model, scene, 1, exchange="allreduce")
    n_params = sum(p.numel() for p in trainer.grads.params)
    lines = {}
    for rays, steps, warmup in shapes:
        batches = make_batches(scene, dev, 2, 0, rays=rays)
        for i in range(warmup):
            trainer.step(batches[i % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            trainer.step(batches[i % 2])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        from presight_amd import ops as _ops

        side_was, _ops.SIDE_STREAM = _ops.SIDE_STREAM, False  # per-region times with the chip to themselves
        pipe_was, trainer.pipeline_adam = trainer.pipeline_adam, False
        prof.enable(True)
        for i in range(2):
            trainer.step(batches[i % 2])
        kern = prof.summary()
        prof.enable(False)
        _ops.SIDE_STREAM = side_was
        trainer.pipeline_adam = pipe_was
        # ("main_field_bwd" is the sum of the three stage regions when those are timed)
        per_step = {k: n * ms / 2 for k, (n, ms) in kern.items() if k != "main_field_bwd" or "main_bwd_sem_kernel" not in kern}
        top = sorted(per_step.items(), key=lambda kv: -kv[1])[:3]
        bc = binding_ceiling(cfg, rays, n_params, bool(getattr(trainer, "fused_table_adam", False)))
        lines[rays] = {"workload": cfg["workload"], "rays_per_step": rays, "steps": steps, "warmup": warmup, "ms_per_step": dt * 1e3,
                       "value": rays / dt, "unit": "rays/s", "parameters": n_params,
                       "top3_regions_ms_per_step": {k: round(v, 3) for k, v in top}, "binding": bc["binding"],
                       "frac_of_binding": rays / dt / bc["ceiling_binding_rays_per_s"], "frac_of_mfma": rays / dt / bc["ceiling_mfma_rays_per_s"],
                       "frac_of_hbm_incl_optimizer": rays / dt / bc["ceiling_hbm_incl_optimizer_rays_per_s"]}
        del batches
    if config == "cfg3" and os.environ.get("PRESIGHT_NO_DRY_OVERLAP") != "1":
        # strong scaling of the production tile: the exchange schedule by construction from a dry run of the bucketed, sharded exchange
        # at the per-rank shapes (65 536 rays = N 1, 8192 rays = one of 8 ranks)
        for rays, _, _ in shapes:
            try:
                dry = dry_overlap_timeline(model, scene, rays, dev, steps=3, exchange="sharded")
                n = max(1, 65536 // rays)
                dry["predicted"] = {f"N{n}": exchange_schedule(dry["buckets"], n, "sharded")} if n > 1 else {
                    f"N{m}_upper_bound_full_batch_compute": exchange_schedule(dry["buckets"], m, "sharded") for m in (2, 4)}
                lines[rays]["exchange_overlap_dry_run"] = dry
                if n > 1 and 65536 in lines:
                    tables = sum(p.numel() for nm, p in model.named_parameters() if nm.endswith("hash_table"))
                    lines[rays][f"predicted_N{n}_strong"] = predicted_strong_scaling(lines[65536]["ms_per_step"], dry, n, 4.0 * tables,
                                                                                      record_bytes=record_bytes_per_rank(cfg, rays))
            except Exception as e:
                lines[rays]["exchange_overlap_dry_run"] = {"error": f"{type(e).__name__}: {e}"}
    del trainer, model, scene
    gc.collect()
    torch.cuda.empty_cache()
    return lines


def extract_workload(model_cfg):
    """per lattice point of the dense prior query (SURVEY.md 8d cfg 5: main field once + 2 proposal nets, no_grad), for the roofline rows:
    algorithmic flops / hash bytes as SURVEY.md prices them; EXECUTED MACs of the kernels that run (merged network: base ending in 16
    outputs for every point, the three 64 x 64 layers of the semantic head only for the 32-point tiles the gate lets through; the
    proposal nets' 64 x 8 layer on the matrix cores) and the 64-byte lines their gathers touch (one line per x-pair, 4 per (point, level):
    whatever the row width)."""
    m = CONFIGS[model_cfg]["model"]
    L, F = m["num_levels"], m["features_per_level"]
    LF = L * F
    alg_flop = 2 * (LF * 64 + 64 * 80 + 3 * 64 * 64) + 2 * 2 * 576
    alg_byte = L * 8 * F * 4 + 2 * 8 * 8 * 4 + 64 * 2 + 4
    ex_mac_always = (LF + 3) // 4 * 4 * 64 + 64 * 16 + 2 * 512
    ex_mac_gated = 3 * 64 * 64
    lines_byte = (L + 2 * 8) * 4 * 64 + 64 * 2 + 4
    return dict(alg_flop=alg_flop, alg_byte=alg_byte, ex_mac_always=ex_mac_always, ex_mac_gated=ex_mac_gated, lines_byte=lines_byte)


GATHER_LINES_L2_RESIDENT = 267e9  # measured: distinct 64-B lines/s of random gathers into an L2-resident 4 MiB table (profiles/r02_microbench_random_gather.txt)
GATHER_LINES_HBM_RESIDENT = 54e9  # measured: the same into a 2 GiB table (every line from HBM = 3.5 TB/s of 64-B lines)


def extract_roofline(model_cfg, points_per_s_per_gpu, gated_frac):
    """roofline object of an extraction line, on the work the kernels EXECUTE.  The pass is neither HBM-stream- nor MFMA-bound: its
    compulsory HBM traffic is the tables once + the kept outputs (a few hundred MB per 134 M points), its time goes into gathers that
    the caches serve.  The contract row is therefore the matrix-core one (executed flops / time / peak, small and honest); the gather
    side is reported as 64-byte lines per second next to the two MEASURED gather ceilings of this part (an L2-resident table and an
    HBM-resident one): lattice points are spatially coherent, so the rate lies between them."""
    w = extract_workload(model_cfg)
    ex_flop = 2 * (w["ex_mac_always"] + gated_frac * w["ex_mac_gated"])
    t = 1.0 / points_per_s_per_gpu
    lines = w["lines_byte"] // 64
    return {"bound": "mfma", "kernel": "whole pass (3 field queries per lattice point; executed matrix-core work)",
            "achieved": ex_flop / t / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ex_flop / t / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            "traffic": None, "semantic_head_tiles_executed": gated_frac, "executed_flop_per_point": ex_flop,
            "frac_algorithmic": w["alg_flop"] / t / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            "gather": {"lines_per_point": lines, "lines_per_s": lines / t, "measured_ceiling_l2_resident_lines_per_s": GATHER_LINES_L2_RESIDENT,
                       "measured_ceiling_hbm_resident_lines_per_s": GATHER_LINES_HBM_RESIDENT,
                       "frac_of_l2_resident_ceiling": lines / t / GATHER_LINES_L2_RESIDENT,
                       "frac_of_hbm_resident_ceiling": lines / t / GATHER_LINES_HBM_RESIDENT},
            "frac_algorithmic_hbm": w["alg_byte"] / t / 1e9 / HBM_PEAK_GBS}


# lattice points per query chunk of the extraction benchmark (8 M: 2 GB of fp32 semantics per chunk in flight)
EXTRACT_CHUNK = int(os.environ.get("PRESIGHT_EXTRACT_CHUNK", str(1 << 23)))


def tile_aabb(scene):
    """the box the dense lattice spans: the sub-field's AABB (K = 1) / the union of the K sub-field boxes (routed tile)"""
    import torch

    b = scene["aabbs"].reshape(-1, 2, 3)
    return torch.stack([b[:, 0].min(0).values, b[:, 1].max(0).values])


def secondary_extract_line(dev, res=512, passes=2, model_cfg="cfg2"):
    """BASELINE cfg 5 on one GPU, as `--config extract` measures it (see extract_main), for the `secondary` block"""
    import gc

    import torch

    from presight_amd import field_ops as F
    from presight_amd.extract import dense_tile_query, voxelize

    model, scene = build_model(dev, seed=42, config=model_cfg)
    model.eval()
    aabb = tile_aabb(scene)
    probe = dense_tile_query(model, aabb, res=64, density_threshold=-1.0)
    thr = float(torch.quantile(probe["densities"][:: max(1, probe["densities"].numel() // 100000)], 0.9))
    del probe

    def one_pass():
        out = dense_tile_query(model, aabb, res=res, chunk=EXTRACT_CHUNK, start=0, count=res ** 3, density_threshold=thr)
        return out, voxelize(out["points"], out["features"], None, voxel=0.4, min_bound=out["min_bound"], points_max=out["points_max"])

    one_pass()
    F.GATE_STATS = torch.zeros(2, device=dev, dtype=torch.int64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        out, vox = one_pass()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / passes
    st = F.GATE_STATS.tolist()
    F.GATE_STATS = None
    gated = st[0] / max(st[1], 1)
    roof = extract_roofline(model_cfg, res ** 3 / dt, gated)
    line = {"workload": f"BASELINE cfg 5: dense {res}^3 lattice of one tile ({CONFIGS[model_cfg]['workload'].split(':')[1].split(',')[0].strip()}; "
                        f"K = {CONFIGS[model_cfg]['K']}), 3 field queries + fp16 features + threshold + bit-exact voxel index + voxel down-sampling",
            "ms_per_step": dt * 1e3, "value": res ** 3 / dt, "unit": "points/s", "passes": passes,
            "kept_points": int(out["points"].shape[0]), "voxels": int(vox["key"].shape[0]),
            "frac_of_binding": roof["frac"], "frac_of_binding_note": "executed matrix-core flops / time / fp32 MFMA peak; gather rates in roofline.gather",
            "roofline": roof}
    del model, scene, out, vox
    gc.collect()
    torch.cuda.empty_cache()
    return line


def exchange_model(bytes_on_link_per_rank: float, world: int, step_ms_single: float = None) -> dict:
    """The link-time bounds of DESIGN.md section 6 for this run: a rank sends (and receives) `bytes_on_link_per_rank` per step --
    (N-1)/N of every reduce-scattered / all-gathered byte, twice that for an all-reduce.  xGMI is point-to-point, ~153 GB/s per link and
    direction, 7 links per GPU.  ASSUMPTION (never measured here: no multi-GPU node was available to the builder): RCCL's direct
    (all-to-all) algorithms keep all N-1 peer links busy at once -> t = bytes / ((N-1) * link rate); a ring moves everything over ONE
    link per direction -> t = bytes / link rate.  Which one RCCL picks for these message sizes is not known; both are printed next to
    the measured exposed time, which is what is left after the overlap with backward / the next step's sampling."""
    if world <= 1:
        return None
    direct = bytes_on_link_per_rank / ((world - 1) * XGMI_LINK_GBS * 1e9) * 1e3
    ring = bytes_on_link_per_rank / (XGMI_LINK_GBS * 1e9) * 1e3
    return {"link_GBps_per_direction": XGMI_LINK_GBS, "peer_links_used_ASSUMED": world - 1, "predicted_ms_all_links": direct,
            "predicted_ms_ring_one_link": ring,
            "assumption": "all N-1 peer links concurrently (direct algorithm) vs one link (ring): RCCL's choice is not known, no multi-GPU run exists"}


def exchange_schedule(timeline, world: int, mode: str, compute_scale: float = 1.0) -> dict:
    """Exposed vs hidden exchange time BY CONSTRUCTION, from a measured bucket timeline (FlatGrads.timeline_summary: for every
    exchange bucket its bytes and how long before the end of backward its gradient was complete on this GPU).  The communication
    stream runs the buckets in order: start_b = max(ready_b, end_{b-1}), end_b = start_b + t_b with t_b = the bucket's bytes on the
    links / link rate (exchange_model's two bounds); what is left after the end of backward is EXPOSED, the rest is hidden under
    kernels that are still running.  compute_scale: the per-rank backward of a strong-scaled run is shorter than the measured one."""
    if not timeline:
        return None
    out = {}
    for label, links in (("all_links_ASSUMED", world - 1), ("ring_one_link", 1)):
        t_end, rows, total = None, [], 0.0
        for b in timeline:
            if b["steps_exchanged"] == 0:
                continue
            on_link = b["bytes"] * (world - 1) / world * (2.0 if mode == "allreduce" else 1.0)
            t_b = on_link / (links * XGMI_LINK_GBS * 1e9) * 1e3
            ready = -b["ms_before_backward_end"] * compute_scale
            start = ready if t_end is None else max(ready, t_end)
            t_end = start + t_b
            total += t_b
            rows.append({"bucket": b["bucket"], "MB": round(b["bytes"] / 1e6, 2), "ready_ms_before_backward_end": round(-ready, 3),
                         "link_ms": round(t_b, 3), "exposed_ms": round(max(0.0, t_end) - max(0.0, start), 3) if t_end > 0 else 0.0,
                         "hidden_ms": round(t_b - (max(0.0, t_end) - max(0.0, start)), 3) if t_end > 0 else round(t_b, 3)})
        out[label] = {"link_ms_total": round(total, 3), "exposed_ms": round(max(0.0, t_end or 0.0), 3),
                      "hidden_ms": round(total - max(0.0, t_end or 0.0), 3), "buckets": rows}
    return out


def dry_overlap_timeline(model, scene, rays, dev, steps=5, exchange="allreduce"):
    """bucket timeline of the overlapped exchange measured on ONE GPU: a second trainer on the same model with the bucketed exchange
    armed but no process group (PRESIGHT_DRY_OVERLAP: split accumulate launches, hand-over bookkeeping and stream events run as they
    would with N ranks; no collective is issued)"""
    import torch

    old = os.environ.get("PRESIGHT_DRY_OVERLAP")
    os.environ["PRESIGHT_DRY_OVERLAP"] = "1"
    try:
        tr = Trainer(model, scene, 1, exchange=exchange)
    finally:
        if old is None:
            os.environ.pop("PRESIGHT_DRY_OVERLAP", None)
        else:
            os.environ["PRESIGHT_DRY_OVERLAP"] = old
    from presight_amd import prof

    batches = make_batches(scene, dev, 2, 0, rays=rays)
    for i in range(3):
        tr.step(batches[i % 2])
    tr.grads.record_timeline = True
    prof.enable(True, only=("adam",))  # (the DENSE optimizer pass of this un-fused, un-sharded trainer: a sharded rank runs 1 / N of it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(batches[i % 2])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    adam = prof.summary().get("adam", (0, None))[1]
    prof.enable(False)
    tl = tr.grads.timeline_summary()
    return {"ms_per_step_with_split_launches": ms, "dense_adam_ms": adam, "buckets": tl,
            "handed_over_in_backward": sum(1 for t in tl if t["handed_over_in_backward"] == t["steps_exchanged"] > 0), "n_buckets": len(tl)}


def record_bytes_per_rank(cfg: dict, rays: int) -> float:
    """bytes of the binned backward's record streams one rank writes per step (exchange = sparse ships them instead of the dense
    table gradient): every sample emits 4 x-pair records of 4 * (F + 2) bytes per level, on the main grid and on both proposal grids"""
    m = cfg["model"]
    total = rays * 64 * m["num_levels"] * 4 * 4 * (m["features_per_level"] + 2)   # main field: 64 samples per ray
    for S in (128, 64):                                                              # proposal nets (L = 8, F = 1) on 128 / 64 samples
        total += rays * S * 8 * 4 * 4 * (1 + 2)
    return float(total)


def predicted_strong_scaling(one_gpu_ms: float, dry: dict, n: int, param_bytes: float, record_bytes: float = None) -> dict:
    """What a strong-scaled N-rank run of the production tile would take, assembled from what ONE GPU can measure -- no multi-GPU run
    exists, every figure below is a measurement of one rank's work or a stated assumption about the links:
      compute        the per-rank step at 65 536 / N rays with the gradients written (un-fused tables), split accumulate launches and
                     hand-over bookkeeping, dense Adam (dry run of the bucketed exchange, this process)
      - (N-1)/N of the dense optimizer pass (a sharded rank updates its 1 / N shard)
      + machinery    exchange code live through RCCL in a process group of one rank minus the same step without it
                     (tools/rccl_self_exchange.py, committed under profiles/: host calls, events, the owned shard's division)
      + exposed      link time of the reduce-scatter buckets left after the end of backward (exchange_schedule: bucket hand-over times
                     of the dry run, xGMI bounds of exchange_model; both the all-links ASSUMPTION and the one-link ring bound)
      the all-gather of the updated parameters is left in flight under the next step's sampling front (assumed hidden when it fits)"""
    out = {"n_gpus": n, "one_gpu_ms_full_batch_fused": one_gpu_ms, "rank_compute_ms_unfused_dense_adam": dry.get("ms_per_step_with_split_launches"),
           "dense_adam_ms": dry.get("dense_adam_ms")}
    mach, src = None, None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r06_rccl_group_of_one_cfg3_8192rays.json")))
        mach = rec["rccl_group_of_one_sharded"]["ms_per_step"] - rec["no_exchange_separate_table_update"]["ms_per_step"]
        src = "profiles/r06_rccl_group_of_one_cfg3_8192rays.json (sharded exchange live in a group of one - separate table update, same box)"
    except (OSError, KeyError, ValueError):
        pass
    out["machinery_ms"], out["machinery_source"] = mach, src
    sched = exchange_schedule(dry.get("buckets"), n, "sharded")
    if not sched or out["rank_compute_ms_unfused_dense_adam"] is None:
        return out
    adam = dry.get("dense_adam_ms") or 0.0
    base = out["rank_compute_ms_unfused_dense_adam"] - adam * (n - 1) / n + max(mach or 0.0, 0.0)
    gather_ms = param_bytes * (n - 1) / n / ((n - 1) * XGMI_LINK_GBS * 1e9) * 1e3
    out["param_allgather_link_ms_all_links_ASSUMED"] = gather_ms
    for label in ("all_links_ASSUMED", "ring_one_link"):
        step = base + sched[label]["exposed_ms"]
        out[label] = {"exposed_reduce_scatter_ms": sched[label]["exposed_ms"], "hidden_ms": sched[label]["hidden_ms"], "predicted_step_ms": step,
                      "predicted_speedup_vs_one_gpu": one_gpu_ms / step}
    if record_bytes:
        # exchange = sparse: the hash tables' buckets (everything above 64 MB on this tile) travel as RECORDS.  Same hand-over times as
        # the dense pieces (conservative: the records are complete one accumulate pass earlier), bytes scaled to the record streams; the
        # owner's accumulate pass over the received runs replaces the rank's own (same record count, 1 / N of the slices to flush).
        try:
            tl = dry.get("buckets") or []
            dense_tables = sum(b_["bytes"] for b_ in tl if b_["bytes"] >= 64e6 and b_["steps_exchanged"])
            scaled = [dict(b_, bytes=b_["bytes"] * record_bytes / dense_tables) if (b_["bytes"] >= 64e6 and dense_tables) else b_ for b_ in tl]
            ssched = exchange_schedule(scaled, n, "sharded")
            out["sparse_records"] = {"record_bytes_per_rank": record_bytes, "dense_table_gradient_bytes": dense_tables}
            for label in ("all_links_ASSUMED", "ring_one_link"):
                step = base + ssched[label]["exposed_ms"]
                out["sparse_records"][label] = {"exposed_ms": ssched[label]["exposed_ms"], "predicted_step_ms": step,
                                                "predicted_speedup_vs_one_gpu": one_gpu_ms / step}
        except Exception as e:  # (a prediction must never take the line down)
            out["sparse_records"] = {"error": f"{type(e).__name__}: {e}"}
    out["note"] = ("a prediction from one-GPU measurements and the stated link assumptions; no scaling curve has been measured in any round "
                   "(SCALE_rNN.json of the driver is the only source of measured N > 1 numbers)")
    return out


# --------------------------------------------------------------------------------------------------------- extraction bench
def extract_main(args) -> int:
    """python bench.py --config extract [--gpus N]: BASELINE configs[4], prior extraction of one tile as a dense 512^3 lattice
    query (cfg-2 fields): lattice points -> 2 proposal fields + ONE pass of the main field (density + 64-d semantics) -> mean
    density, clipped fp16 features, density threshold, bit-exact integer voxel index, voxel down-sampling of the kept points.
    The lattice is split into one contiguous slab per rank; there is no exchange inside the timed region.  One "step" = one
    pass over the rank's slab.  Prints ONE JSON line."""
    import torch

    from presight_amd.dist import init_from_env
    from presight_amd.extract import dense_tile_query, voxelize

    rank, local_rank, world = init_from_env("cuda")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    res = int(os.environ.get("PRESIGHT_EXTRACT_RES", "512"))
    mcfg = args.extract_model
    model, scene = build_model(dev, seed=42, config=mcfg)
    model.eval()
    aabb = tile_aabb(scene)
    total = res ** 3
    per = (total + world - 1) // world
    start, count = rank * per, max(0, min(per, total - rank * per))

    # the reference keeps points with mean density > 1.0 (extract_priors.py:152), which on a TRAINED tile is the few per cent of
    # the lattice near surfaces; the synthetic (random-init) fields have no surfaces, so the bench keeps the densest 10 % instead
    # (threshold = 90th percentile of a 64^3 probe), which gives the voxel down-sampling a realistic amount of work
    probe = dense_tile_query(model, aabb, res=64, density_threshold=-1.0)  # also the warm-up: kernels, allocator
    thr = float(torch.quantile(probe["densities"][:: max(1, probe["densities"].numel() // 100000)], 0.9))
    del probe

    def one_pass():
        out = dense_tile_query(model, aabb, res=res, chunk=EXTRACT_CHUNK, start=start, count=count, density_threshold=thr)
        vox = voxelize(out["points"], out["features"], None, voxel=0.4, min_bound=out["min_bound"], points_max=out["points_max"], want_sums=world > 1)
        return out, vox

    for _ in range(max(0, args.warmup - 1)):
        one_pass()
    steps = max(1, min(args.steps, 5))
    from presight_amd import field_ops as F

    F.GATE_STATS = torch.zeros(2, device=dev, dtype=torch.int64)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out, vox = one_pass()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    kept, nvox = int(out["points"].shape[0]), int(vox["key"].shape[0])
    gst = F.GATE_STATS.tolist()
    F.GATE_STATS = None
    if rank == 0:
        value = total * steps / dt
        per_gpu = value / world
        line = {"metric": "prior-extraction lattice points/sec (whole node)", "value": value, "unit": "points/s", "n_gpus": world, "steps": steps,
                "warmup": args.warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"BASELINE cfg 5: dense {res}^3 lattice of one tile ({mcfg} fields, K = {CONFIGS[mcfg]['K']} sub-fields): 3 field "
                                       "queries + fp16 features + density threshold + bit-exact voxel index + voxel down-sampling of the kept points",
                           "points_per_gpu": count, "parallelism": f"slabs{world}"},
                "roofline": extract_roofline(mcfg, per_gpu, gst[0] / max(gst[1], 1)),
                "density_threshold": thr, "kept_points_rank0": kept, "voxels_rank0": nvox}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_query_baseline(mcfg)
            line["speedup_vs_cpu"] = value / line["cpu_baseline"]["value"]
        emit(line)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return 0


def cpu_query_baseline(model_cfg="cfg2") -> dict:
    """the CPU oracle's prior query (restatement of extract_priors.py:133-138, kind = "port") on a bounded sample of lattice points;
    model_cfg cfg3: the production tile (K = 16 routed sub-fields, L10 F4 T2^20 main tables)"""
    import statistics

    import torch

    from oracle import nerf_oracle as O

    cfg = O.default_config() if model_cfg == "cfg2" else O.prod_shaped_config(16, log2_hashmap_size=20)
    if model_cfg != "cfg2":
        cfg["num_cameras"], cfg["num_videos"] = 1440, 6
    scene = O.make_scene(cfg)
    P = O.make_params(cfg, seed=42)
    n = 1 << 18
    g = torch.Generator().manual_seed(0)
    lo, hi = scene["aabbs"][:, 0].min(0).values, scene["aabbs"][:, 1].max(0).values
    pts = lo + (hi - lo) * torch.rand(n, 3, generator=g)
    phys, model_name = host_cpu_info()
    maxt = torch.get_num_threads()
    best, best_dt, probe = None, float("inf"), {}
    with torch.no_grad():
        O.prior_query(P, cfg, scene, pts[:4096])
        for th in sorted({t for t in (8, 16, 32, 64, 128, phys) if t <= max(phys, maxt)}):
            torch.set_num_threads(th)
            t0 = time.time()
            O.prior_query(P, cfg, scene, pts[:65536])
            d = time.time() - t0
            probe[th] = round(65536 / d, 1)
            if d < best_dt:
                best, best_dt = th, d
        torch.set_num_threads(best or maxt)
        times = []
        for _ in range(5):
            t0 = time.time()
            O.prior_query(P, cfg, scene, pts)
            times.append(time.time() - t0)
            if sum(times) > 30:
                break
    used = torch.get_num_threads()
    torch.set_num_threads(maxt)
    return dict(value=n / statistics.median(times), unit="points/s", cores=used, kind="port",
                sample=f"median of {len(times)} passes over {n} lattice points (2 proposal fields + main field density and semantics, {model_cfg} tables), "
                       f"torch-CPU oracle, {used} threads",
                host=dict(cpu_model=model_name, physical_cores=phys, logical_cpus=os.cpu_count(), thread_probe_points_per_s=probe))


# --------------------------------------------------------------------------------------------------------- main
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.config == "extract":
        sys.exit(extract_main(args))

    import torch

    if os.environ.get("PRESIGHT_HANG_DUMP"):  # debugging aid: dump every thread's stack after N seconds and exit
        import faulthandler

        path = os.environ.get("PRESIGHT_HANG_DUMP_FILE")
        global _HANG_FILE  # keep the file object alive
        _HANG_FILE = open(path.replace("{rank}", os.environ.get("RANK", "0")), "w") if path else sys.stderr
        faulthandler.dump_traceback_later(float(os.environ["PRESIGHT_HANG_DUMP"]), exit=True, file=_HANG_FILE)

    from presight_amd import prof
    from presight_amd.dist import init_from_env

    rank, local_rank, world = init_from_env("cuda")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path for the product)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg = CONFIGS[args.config]
    tiles = args.parallelism == "tiles"
    scaling = "weak" if tiles else (args.scaling or cfg["scaling"])
    exchange = args.exchange or cfg["exchange"]
    rays = args.rays if scaling == "weak" else args.rays // world  # ns/data/PreSight/my_datamanager.py:203-212: R // world
    # dp: same init on every rank (DDP broadcast equivalent).  tiles: every rank owns a DIFFERENT tile -- its own parameters, its own
    # rays, no collective in the step (the process group only brackets the timed region)
    model, scene = build_model(dev, seed=42 + (rank if tiles else 0), config=args.config)
    trainer = Trainer(model, scene, 1 if tiles else world, exchange=exchange, global_depth_clip=args.global_depth_clip and not tiles,
                      table_pieces=2 if (world > 1 and not tiles and rays <= 16384) else None)
    # data: the device-resident chunk feed (one gather launch per batch, next chunk prefetched on a side stream; the reference's
    # loader semantics: shuffled pass over the chunk, rank r takes every world-th slot) or 4 recycled pre-made batches
    feed, batches = None, None
    if args.fixed_batches:
        batches = make_batches(scene, dev, 4, rank, rays=rays)
    else:
        from presight_amd.datafeed import ChunkFeed

        feed = ChunkFeed(lambda i: synthetic_chunk(scene, dev, i), batch_size=rays, device=dev, world=1 if tiles else world, rank=0 if tiles else rank)
    last_batch = [None]

    def run(n):
        for i in range(n):
            last_batch[0] = feed.next_batch() if feed is not None else batches[i % len(batches)]
            out = trainer.step(last_batch[0])
        return out

    def timed(n):
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run(n)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=dev)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, out

    run(args.warmup)
    trainer.grads.stats = {"collectives": 0, "bytes": 0}
    # the longest kernels are timed live inside the timed steps (one HIP-event pair each); every other region -- two dozen
    # event pairs per step and the three-call split of the backward cost the timeline 1 % -- in a separate pass behind it
    mcfg_ = cfg["model"]
    table_regions = (f"bin_kernel_L{mcfg_['num_levels']}F{mcfg_['features_per_level']}", f"accumulate_kernel_L{mcfg_['num_levels']}F{mcfg_['features_per_level']}")
    live = (("main_field_fwd",) if trainer.pipeline_adam else ("main_field_fwd", "adam")) + (table_regions if (getattr(trainer, "fused_table_adam", False) and os.environ.get("PRESIGHT_BENCH_LIVE_TABLES", "1") != "0") else ())
    prof.enable(True, only=live)
    dt, (loss_dict, out) = timed(args.steps)
    psnr = float(model.get_metrics_dict(out, last_batch[0])["psnr"].detach())
    kern = prof.summary()
    # (the per-kernel pass runs WITHOUT the proposal networks' side stream: a kernel's roofline row is its time with the chip to
    # itself; in the timed steps the proposal backward overlaps the main field's)
    from presight_amd import ops as _ops

    comm = dict(trainer.grads.stats)
    side_was, _ops.SIDE_STREAM = _ops.SIDE_STREAM, False
    pipe_was, trainer.pipeline_adam = trainer.pipeline_adam, False
    live_adam = kern.pop("adam", None) if pipe_was else None  # (pipelined: the live region only brackets the proposal networks' piece)
    prof.enable(True)
    run(min(args.steps, 8))
    kern = {**prof.summary(), **kern}  # live figures win
    prof.enable(False)
    _ops.SIDE_STREAM = side_was
    trainer.pipeline_adam = pipe_was
    timeline = None
    if world > 1 and not tiles:  # when every bucket became ready relative to the end of backward (normal schedule: proposal side stream on)
        trainer.grads.record_timeline = True
        run(4)
        timeline = trainer.grads.timeline_summary()
        trainer.grads.record_timeline = False
    # secondary figure (NOT `value`): the reference's own steady-state proposal-update schedule after warm-up
    # (ray_samplers.py:586 + nerfacto_nusc_ms.py:300-305: gradients reach the proposal nets every 6th step)
    trainer.update_props_every_step = False
    trainer.step_idx = 50000
    model.proposal_sampler._steps_since_update = 0
    n_sched = 12
    run(6)
    dt_sched, _ = timed(n_sched)
    # secondary figure: the other scaling mode on the same ranks (strong: args.rays over all ranks; weak: args.rays per rank)
    other = None
    if world > 1 and not tiles:
        trainer.update_props_every_step = True
        o_rays = args.rays // world if scaling == "weak" else args.rays
        keep = (feed, batches)
        feed, batches = None, make_batches(scene, dev, 2, rank, rays=o_rays)
        run(3)
        dt_o, _ = timed(8)
        feed, batches = keep
        other = dict(scaling="strong" if scaling == "weak" else "weak", rays_per_gpu=o_rays, value=world * o_rays * 8 / dt_o,
                     ms_per_step=dt_o / 8 * 1e3)
    replica_diff = None
    if world > 1 and not tiles:
        # data-parallel consistency: every rank applied the same averaged gradients, so the replicas must still be identical
        trainer.grads.wait_params()
        mine = trainer.opt.flat[0]
        ref = mine.clone()
        torch.distributed.broadcast(ref, src=0)
        d = (mine - ref).abs().max().reshape(1)
        torch.distributed.all_reduce(d, op=torch.distributed.ReduceOp.MAX)
        replica_diff = float(d.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * rays * args.steps / dt
        fused_tables = None
        if getattr(trainer, "fused_table_adam", False):
            main_t = sum(p.numel() for f_ in model.field.fields for n_, p in f_.named_parameters() if n_.endswith("hash_table"))
            prop_t = sum(p.numel() for net in model.proposal_networks for n_, p in net.named_parameters() if n_.endswith("hash_table"))
            fused_tables = (main_t, prop_t / max(1, len(model.proposal_networks)))
        rows = roofline_entries(kern, cfg, rays, n_params=sum(p.numel() for p in trainer.grads.params), live=live, fused_tables=fused_tables)
        # the dominant kernel: the single KERNEL with the longest average launch INSIDE the timed steps (what rocprofv3's kernel table of
        # the same command ranks first) -- the main field's forward and the two long kernels of the main table backward are timed live
        # (`timed_region`), the other rows come from the per-kernel pass; `roofline_mfma` keeps the matrix-bound forward next to it
        single = [r for r in rows if "summed" not in r["kernel"] and "mean of both" not in r["kernel"] and "absmax+bin" not in r["kernel"]]
        dom = max(single, key=lambda r: r["avg_launch_ms"]) if single else None
        dom_mfma = next((r for r in rows if r["kernel"] == "main_fwd_kernel"), None)
        traffic, traffic_src = (pmc_traffic(dom["kernel"].split(" ")[0].split("<")[0]) if (dom is not None and args.config == "cfg2" and rays == RAYS)
                                else (None, "not collected for this shape"))
        traffic_mfma = pmc_traffic("main_fwd_kernel")[0] if (dom_mfma is not None and args.config == "cfg2" and rays == RAYS) else None
        # end-to-end ceilings per training ray: MLP flops (fwd + 2x bwd) against the fp32 matrix peak, hash bytes
        # (gather fwd, read + write bwd) against HBM; the binding (lower) ceiling is the fp32 MFMA one
        bc = binding_ceiling(cfg, rays, sum(p.numel() for p in trainer.grads.params), bool(getattr(trainer, "fused_table_adam", False)))
        flop_ray, byte_ray, ceil_mfma, ceil_hbm = bc["flop_per_ray"], bc["hash_bytes_per_ray"], bc["ceiling_mfma_rays_per_s"], bc["ceiling_hbm_rays_per_s"]
        per_gpu = value / world
        line = {
            "metric": "training rays/sec (whole node)", "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "data_feed": "4 recycled batches" if feed is None else f"device chunk feed ({feed.chunks_loaded} chunk(s) of {1 << 22} pixels loaded)",
            "config": {"workload": cfg["workload"], "rays_per_gpu": rays, "rays_per_step_global": rays * world,
                       "parallelism": f"tiles{world} (one independent tile per GPU, no exchange)" if tiles else f"dp{world}",
                       "exchange": trainer.exchange if (world > 1 and not tiles) else None},
            "roofline": None if dom is None else {
                "bound": dom["bound"], "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": dom["peak"],
                "unit": dom["unit"], "frac": dom["frac"], "traffic": traffic, "traffic_unit": "bytes/launch (rocprofv3 PMC)",
                "traffic_source": traffic_src, "avg_launch_ms": dom["avg_launch_ms"], "launches": dom["launches"],
                # `achieved` / `frac`: the matrix-core flops the kernel actually ISSUES (padding included) / its HIP-event duration (+ the
                # small per-ray launches that took over part of its work) / peak.  `frac_algorithmic` prices the flops of the reference's
                # unfactored network (26 752 MAC per sample) against the same duration
                **({"executed_flops": dom.get("executed"), "algorithmic_flops": dom["algorithmic"], "frac_algorithmic": dom.get("frac_algorithmic")}
                   if dom["bound"] == "mfma" else {"algorithmic_bytes": dom["algorithmic"], "timed_inside_the_timed_steps": dom.get("timed_region")}),
                "duration_ms_incl_moved_work": dom.get("duration_ms_incl_moved_work"), "moved_work_ms": dom.get("moved_work_ms"),
                # what the traffic is: the kernel's algorithmic I/O (feature planes in; density, colour, weights per sample and the composited
                # semantic activations per ray out) against the hidden activations it KEEPS for the three backward kernels (register order,
                # fp32: the price of an exact-fp32 backward without recompute) -- the MFMA-bound forward also streams this
                **(main_fwd_io(cfg, rays, traffic, dom.get("avg_launch_ms")) if dom["kernel"] == "main_fwd_kernel" else {})},
            "roofline_mfma": None if dom_mfma is None else {
                "bound": "mfma", "kernel": dom_mfma["kernel"], "achieved": dom_mfma["achieved"], "peak": dom_mfma["peak"], "unit": dom_mfma["unit"],
                "frac": dom_mfma["frac"], "frac_algorithmic": dom_mfma.get("frac_algorithmic"), "avg_launch_ms": dom_mfma["avg_launch_ms"],
                "traffic": traffic_mfma, **main_fwd_io(cfg, rays, traffic_mfma, dom_mfma.get("avg_launch_ms"))},
            "roofline_kernels": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in rows],
            "end_to_end": {**bc, "frac_of_binding": per_gpu / bc["ceiling_binding_rays_per_s"], "frac_of_mfma": per_gpu / ceil_mfma,
                           "frac_of_hbm": per_gpu / ceil_hbm, "frac_of_hbm_incl_optimizer": per_gpu / bc["ceiling_hbm_incl_optimizer_rays_per_s"]},
            "kernels_ms": {k: round(v[1], 4) for k, v in sorted(kern.items())},
            "kernels_ms_note": "per-kernel pass without the proposal side stream (each kernel alone on the chip); the timed steps overlap "
                               "the proposal networks' backward with the main field's" if side_was else "single stream",
            "optimizer": {"adam": "every step, inside the timed region",
                          "hash_tables": ("updated INSIDE their table backward (ps_grid_scatter_binned_adam: the accumulate pass applies the "
                                          "optimizer kernel's own element update to every slice, bit-equal to the separate step; the "
                                          "`adam` region below covers the remaining parameters only)") if getattr(trainer, "fused_table_adam", False)
                          else "updated by adam_ranges_kernel like every other parameter (a gradient exchange needs the gradients)",
                          "fused_table_adam": bool(getattr(trainer, "fused_table_adam", False))},
            "value_reference_schedule": world * rays * n_sched / dt_sched,
            "other_scaling": other,
            "comm": None if (world == 1 or tiles) else {"backend": torch.distributed.get_backend(), "ranks": torch.distributed.get_world_size(),
                                             "collectives_per_step": comm["collectives"] / args.steps,
                                             "gradient_buckets_issued_during_backward_per_step": comm.get("in_backward", 0) / args.steps,
                                             "bytes_on_link_per_rank_per_step": comm["bytes"] / args.steps,
                                             # exchange = sparse: the part of it that is table-gradient RECORDS (the dense table gradient
                                             # the sharded mode would reduce-scatter instead: dense_table_gradient_bytes * (N - 1) / N)
                                             "record_bytes_on_link_per_rank_per_step": comm.get("sparse_record_bytes", 0) / args.steps,
                                             "dense_table_gradient_bytes": 4.0 * sum(p.numel() for n_, p in model.named_parameters() if n_.endswith("hash_table")),
                                             "bucket_timeline": timeline,
                                             "schedule_by_construction": exchange_schedule(timeline, world, trainer.exchange),
                                             "exchange_exposed_ms": kern.get("exchange_exposed", (0, None))[1],
                                             "model": exchange_model(comm["bytes"] / args.steps, world)},
            "replicas_max_abs_diff": replica_diff,
            "psnr_vs_random_targets": psnr,
            "loss": float(sum(v.detach() for v in loss_dict.values())),
        }
        if world == 1 and rays == RAYS and os.environ.get("PRESIGHT_NO_DRY_OVERLAP") != "1" and "dynamic" not in cfg:
            # the overlapped exchange, measured as far as ONE GPU can: bucket hand-over times from a dry run of the bucketed exchange
            # (split accumulate launches, no collectives) -> exposed / hidden link time per bucket for N = 2, 4, 8 by construction
            try:
                if feed is not None:
                    feed.close()
                    feed = None
                dry = dry_overlap_timeline(model, scene, rays, dev, exchange=exchange)
                dry["predicted"] = {f"N{n}": exchange_schedule(dry["buckets"], n, exchange) for n in (2, 4, 8)}
                dry["note"] = ("weak scaling (every rank runs this step): per-bucket link time from the xGMI bounds of exchange_model, hidden "
                               "under the kernels still running after the bucket's hand-over; no multi-GPU run exists")
                line["exchange_overlap_dry_run"] = dry
                # the step an N > 1 data-parallel rank runs: table gradients written + separate Adam (the fused table update needs no exchange)
                line["ms_per_step_unfused_tables"] = dry["ms_per_step_with_split_launches"]
                line["value_unfused_tables"] = rays / (dry["ms_per_step_with_split_launches"] * 1e-3)
            except Exception as e:
                line["exchange_overlap_dry_run"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.config == "cfg2" and not args.no_secondary and rays == RAYS:
            # the other single-GPU BASELINE shapes, a few steps each (cfg 2 is the line itself; their own full runs: --config ...)
            if feed is not None:
                feed.close()
            last_batch[0] = None
            del trainer, model, feed, batches, out, loss_dict
            import gc

            gc.collect()
            torch.cuda.empty_cache()
            sec = {}

            def run_sec(keys, fn):
                t_sec = time.perf_counter()
                try:
                    res = fn()
                except Exception as e:  # a secondary figure must never take the headline line down
                    res = {k: {"error": f"{type(e).__name__}: {e}"} for k in keys}
                for k in keys:
                    sec[k] = res[k]
                sec[keys[0]]["wall_s_incl_model_build"] = round(time.perf_counter() - t_sec, 1)

            def cfg3():
                r = secondary_training_lines("cfg3", [(65536, 4, 2), (8192, 12, 4)], dev)
                return {"cfg3_65536": r[65536], "cfg3_8192": r[8192]}

            run_sec(["cfg3_65536", "cfg3_8192"], cfg3)
            run_sec(["cfg4_65536"], lambda: {"cfg4_65536": secondary_training_lines("cfg4", [(65536, 4, 2)], dev)[65536]})
            run_sec(["cfg4prod_65536"], lambda: {"cfg4prod_65536": secondary_training_lines("cfg4prod", [(65536, 3, 2)], dev)[65536]})
            run_sec(["extract_512"], lambda: {"extract_512": secondary_extract_line(dev)})
            run_sec(["extract_512_prod"], lambda: {"extract_512_prod": secondary_extract_line(dev, model_cfg="cfg3")})
            line["secondary"] = sec
        if world == 1 and args.config == "cfg2" and args.psnr_steps > 0:
            try:
                line["psnr_after_k_steps"] = psnr_after_k_steps(dev, "cfg2", K=args.psnr_steps)
            except Exception as e:  # never takes the headline down
                line["psnr_after_k_steps"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline and args.config == "cfg2":
            line["cpu_baseline"] = cpu_baseline()
            line["speedup_vs_cpu"] = value / line["cpu_baseline"]["value"]
            line["psnr_vs_oracle"] = _PSNR_VS_ORACLE
        if world > 1 and not tiles:
            mdl = line["comm"]["model"]
            print(f"bench.py: gradient exchange per step and rank: {line['comm']['bytes_on_link_per_rank_per_step'] / 1e6:.1f} MB on the links; "
                  f"predicted {mdl['predicted_ms_all_links']:.2f} ms over all {world - 1} peer links ({mdl['predicted_ms_ring_one_link']:.2f} ms as a "
                  f"one-link ring) vs {line['comm']['exchange_exposed_ms']} ms measured EXPOSED (not hidden under backward / sampling); "
                  f"step {ms:.2f} ms", file=sys.stderr)
        emit(line)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
