"""A LEARNABLE synthetic scene for training-quality checks ("PSNR vs synthetic GT after K steps").

There is no nuScenes on the GPU box and random per-pixel targets carry nothing to learn (their rgb loss sits at Var U[0,1]).
A scene that CAN be learnt: a fixed "teacher" parameter set of the same model family, rendered in eval mode through the
same kernels -- per-pixel targets (rgb, 64-d features, sky mask from the accumulation) that are a deterministic function of
the ray, exactly what a camera log is to the reference's data manager (ns/data/PreSight/my_dataset.py:28-73: flat per-pixel
arrays of rgb / sky / features).  The teacher's samples are placed by its OWN density (its main field stands in for the
proposal networks), so its renders are converged volume renderings of its field.

Used by bench.py (`psnr_after_k_steps`), tools/soak.py and tests/test_hip_trainer.py; the tests' CPU checker builds the same scene
with its own code (draw order and constants below are the contract) for the parity check at fixture size."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
from torch import Tensor

from . import ops
from .rays import RayBundle

TEACHER_FAR = 5.0             # the teacher's far plane: beyond it the scene is empty (rays that accumulate little inside are "sky")
TEACHER_MAX_RES = 128         # hash levels up to this resolution carry the scene; finer levels are zero (a smooth, learnable field)
TEACHER_LOG_DENSITY = (-2.5, 2.0)   # mean / std of ln(density) over a sub-field's box: accumulations spread over ~0.5 .. 1, ~5 % sky
TEACHER_RGB_GAIN, TEACHER_SEM_GAIN = 32.0, 3.0
SKY_ACCUMULATION = 0.5        # a ray whose teacher accumulation stays below this is labelled sky


@torch.no_grad()
def shape_teacher_(model, seed: int = 1234, max_res: float = TEACHER_MAX_RES, log_density=TEACHER_LOG_DENSITY,
                   rgb_gain: float = TEACHER_RGB_GAIN, sem_gain: float = TEACHER_SEM_GAIN, probe: int = 4096) -> None:
    """Rewrite the parameters of `model` (a NerfactoNuscMSModel on the GPU, any K) into the teacher, in place:
      * main hash tables redrawn U(-1, 1) on the levels with resolution <= max_res, zero above;
      * per sub-field, the density head's output row is rescaled and re-biased so that ln(density) has the mean / std
        `log_density` over `probe` points drawn uniformly in the sub-field's box (whatever the grid / MLP shape);
      * colour and semantic output layers amplified (a default-initialised head renders a constant grey), semantic output bias
        0.5 (features spread around the middle of the [0, 1] range the loss clips its targets to).
    All random numbers come from one CPU generator in a fixed order (tables in sorted key order, then the probe points of
    sub-field 0, 1, ...), so that the tests' CPU checker can make the same teacher."""
    sd = model.state_dict()  # tensors that alias the parameters
    g = torch.Generator().manual_seed(seed)
    dev = next(model.parameters()).device
    for key in sorted(sd):
        if key.startswith("field.") and key.endswith("mlp_base_grid.hash_table"):
            k = int(key.split(".")[2])
            sc = model.field.fields[k].mlp_base_grid.scalings.cpu()
            v = sd[key]
            L = sc.numel()
            tab = torch.rand(v.shape, generator=g) * 2 - 1
            tab.view(L, v.shape[0] // L, -1)[sc > max_res] = 0.0
            v.copy_(tab.to(v))
    for k, f in enumerate(model.field.fields):
        W, b = sd[f"field.fields.{k}.mlp_base_mlp.layers.1.weight"], sd[f"field.fields.{k}.mlp_base_mlp.layers.1.bias"]
        box = f.aabb.detach().cpu()
        pts = box[0] + (box[1] - box[0]) * torch.rand(probe, 3, generator=g)
        raw = torch.log(f.density_fn(pts.to(dev))[0].reshape(-1).clamp_min(1e-30)).double()
        mu, sdev = float(raw.mean()), float(raw.std())
        gain = log_density[1] / sdev
        b0 = float(b[0])
        W[0] *= gain
        b[0] = log_density[0] - gain * (mu - b0)
        sd[f"field.fields.{k}.rgb_head.layers.2.weight"].mul_(rgb_gain)
        last = max(int(key.split(".")[5]) for key in sd if key.startswith(f"field.fields.{k}.semantic_head.layers."))
        sd[f"field.fields.{k}.semantic_head.layers.{last}.weight"].mul_(sem_gain)
        sd[f"field.fields.{k}.semantic_head.layers.{last}.bias"].fill_(0.5)


class TeacherScene:
    """teacher: a NerfactoNuscMSModel (any K) whose parameters define the scene; rendered in eval mode, samples guided by the
    teacher's own density."""

    def __init__(self, teacher, scene: Dict):
        self.model, self.scene = teacher, scene
        teacher.eval()
        field = teacher.field
        fn = lambda pos: field.density_fn(pos)[0]  # noqa: E731
        n = len(teacher.density_fns)
        teacher.density_fns = [fn] * n

    @torch.no_grad()
    def targets(self, ray_indices: Tensor, video_ids: Optional[Tensor] = None, chunk: int = 1 << 16) -> Dict[str, Tensor]:
        """per-pixel targets of rays (camera, row, col) [n, 3]: rgb [n, 3], features [n, 64] (clipped to [0, 1] like the
        reference's DINO PCA features), sky [n] (1.0 = sky), accumulation [n]"""
        s, m = self.scene, self.model
        outs = dict(rgb=[], features=[], sky=[], accumulation=[])
        for a in range(0, ray_indices.shape[0], chunk):
            ri = ray_indices[a:a + chunk]
            o, d, pa, dn = ops.generate_rays(ri, s["c2w"], s["fx"], s["fy"], s["cx"], s["cy"])
            vid = torch.zeros(ri.shape[0], dtype=torch.int64, device=ri.device) if video_ids is None else video_ids[a:a + chunk]
            out = m(RayBundle(o, d, pa, camera_indices=ri[:, 0:1], metadata={"video_id": vid.view(-1, 1), "directions_norm": dn}))
            acc = out["accumulation"].reshape(-1)
            outs["rgb"].append(out["rgb"])
            outs["features"].append(out["semantics"].clamp(0.0, 1.0))
            outs["accumulation"].append(acc)
            outs["sky"].append((acc < SKY_ACCUMULATION).float())
        return {k: torch.cat(v) for k, v in outs.items()}

    def chunk(self, chunk_index: int, pixels: int = 1 << 22, seed: int = 4321) -> Dict[str, Tensor]:
        """one chunk of the training set in the layout presight_amd.datafeed.ChunkFeed serves (the reference's ImageChunk arrays):
        uniformly drawn (image, pixel) slots with the teacher's targets"""
        s = self.scene
        dev = s["c2w"].device
        g = torch.Generator(device=dev).manual_seed(seed + chunk_index)
        C, H, W = s["c2w"].shape[0], s["H"], s["W"]
        img = torch.randint(0, C, (pixels,), device=dev, generator=g)
        pix = torch.randint(0, H * W, (pixels,), device=dev, generator=g)
        ri = torch.stack([img, pix // W, pix % W], -1)
        vid = torch.clamp(img // s["frames_per_video"], max=self.model.kwargs["num_train_videos"] - 1)
        t = self.targets(ri, vid)
        return dict(rgbs=t["rgb"], pixel_indices=pix, image_indices=img, video_ids=vid, widths=torch.full((pixels,), W, device=dev, dtype=torch.int64),
                    skies=t["sky"], depths=None, features=t["features"])


@torch.no_grad()
def eval_psnr(model, scene: Dict, ray_indices: Tensor, video_ids: Tensor, target_rgb: Tensor, chunk: int = 1 << 16) -> float:
    """PSNR = 10 log10(1 / MSE) (torchmetrics PSNR with data_range 1, ns/models/PreSight/nerfacto_nusc_ms.py:382,548-556) of the
    model's EVAL render (no jitter, mean appearance code: get_outputs_for_camera_ray_bundle's mode) against target pixels"""
    was = model.training
    model.eval()
    se, n = 0.0, 0
    for a in range(0, ray_indices.shape[0], chunk):
        ri = ray_indices[a:a + chunk]
        o, d, pa, dn = ops.generate_rays(ri, scene["c2w"], scene["fx"], scene["fy"], scene["cx"], scene["cy"])
        out = model(RayBundle(o, d, pa, camera_indices=ri[:, 0:1], metadata={"video_id": video_ids[a:a + chunk].view(-1, 1), "directions_norm": dn}))
        se += float(((out["rgb"] - target_rgb[a:a + chunk]) ** 2).sum())
        n += ri.shape[0] * 3
    model.train(was)
    return 10.0 * math.log10(1.0 / max(se / n, 1e-30))
