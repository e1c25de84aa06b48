"""Reference-format checkpoint I/O for the training loop of presight_amd.trainer.Trainer.

The reference's Trainer writes `<checkpoint_dir>/step-%09d.ckpt` = torch.save of

    {"step": int,
     "pipeline":   pipeline.state_dict()          -- the model's tensors under `_model.` (under DDP: `_model.module.`),
     "optimizers": {group: torch.optim.Adam.state_dict()}   for the groups "proposal_networks" and "fields",
     "schedulers": {group: ChainedScheduler.state_dict()}   (WarmupMultiStepScheduler = LinearLR + MultiStepLR),
     "scalers":    GradScaler.state_dict()}

(ns/engine/trainer.py:432-460), reads it back with `_load_checkpoint` (trainer.py:396-429: `_start_step = step + 1`, pipeline,
optimizers, schedulers when `load_scheduler`, scalers), and the extraction script restores a model through `eval_setup` ->
`eval_load_checkpoint` -> `Pipeline.load_pipeline` (ns/utils/eval_utils.py:38-110, ns/pipelines/base_pipeline.py:426-437: a leading
`module.` is stripped, `model.update_to_step(step)`, strict `load_state_dict`).  This module writes and reads exactly that layout:
`tests/golden/checkpoint.npz` is such a checkpoint produced by the reference's own objects.

Scope notes.  (i) Only checkpoints of the `implementation="torch"` grid layout are table-compatible (fields.py warns about
tiny-cuda-nn's).  (ii) The reference keeps the proposal sampler's update-schedule counters on the sampler object and does not save
them: a resumed reference run restarts them.  `save_checkpoint` adds them under the extra top-level key "presight_amd" (ignored by the
reference's loader, which indexes the five keys above); `load_checkpoint(restore_sampler=True)` uses them when present."""
from __future__ import annotations

import collections
import os
from typing import Dict, List, Optional

import torch

GROUPS = ("proposal_networks", "fields")  # ns/models/PreSight/nerfacto_nusc_ms.py get_param_groups / ns/configs/method_configs.py optimizers


def checkpoint_path(checkpoint_dir: str, step: int) -> str:
    return os.path.join(str(checkpoint_dir), f"step-{int(step):09d}.ckpt")


def latest_step(checkpoint_dir: str) -> int:
    """the newest step-*.ckpt of a directory, by the reference's own rule (trainer.py:404-405, eval_utils.py:55)"""
    steps = sorted(int(x[x.find("-") + 1: x.find(".")]) for x in os.listdir(str(checkpoint_dir)) if x.startswith("step-") and x.endswith(".ckpt"))
    if not steps:
        raise FileNotFoundError(f"no step-*.ckpt in {checkpoint_dir}")
    return steps[-1]


# --------------------------------------------------------------------------------------------------------- pipeline <-> model
def pipeline_state(model, ddp: bool = False) -> "collections.OrderedDict[str, torch.Tensor]":
    """pipeline.state_dict() of a VanillaPipeline whose `_model` is `model` (under DDP the wrapper adds `module.`); host copies"""
    prefix = "_model.module." if ddp else "_model."
    return collections.OrderedDict((prefix + k, v.detach().cpu().clone()) for k, v in model.state_dict().items())


def model_state(pipeline: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """the model's state_dict out of a checkpoint's "pipeline" entry: load_pipeline's `module.` strip, then the `_model.` child prefix
    (and DDP's inner `module.`); entries of other pipeline children (a data manager's camera optimizer) are not the model's"""
    out = {}
    for k, v in pipeline.items():
        if k.startswith("module."):
            k = k[len("module."):]
        if not k.startswith("_model."):
            continue
        k = k[len("_model."):]
        if k.startswith("module."):
            k = k[len("module."):]
        out[k] = v
    return out


def load_pipeline(model, pipeline: Dict[str, torch.Tensor], step: int) -> None:
    """Pipeline.load_pipeline (base_pipeline.py:426-437) for the model: update_to_step + STRICT load_state_dict"""
    if hasattr(model, "update_to_step"):
        model.update_to_step(step)
    model.load_state_dict(model_state(pipeline), strict=True)


def model_kwargs_from_pipeline(pipeline: Dict[str, torch.Tensor]) -> Dict:
    """the constructor arguments the reference's pipeline takes from its DATA MANAGER (ns/pipelines/PreSight/my_pipeline.py:104-117:
    centroids and per-sub-field AABBs from the dataparser's k-means, the numbers of training cameras / videos), recovered from a
    checkpoint instead: they are all there as buffers and embedding shapes.  With them a model can be rebuilt from its config and a
    checkpoint alone -- what prior extraction needs when the dataset the tile was trained on is not at hand."""
    sd = model_state(pipeline)
    K = int(sd["field.centroids"].shape[0])
    out = {"centroids": sd["field.centroids"].clone(), "aabbs": torch.stack([sd[f"field.fields.{k}.aabb"] for k in range(K)]).clone(),
           "dino_to_rgb": None}
    if "appearance_embedding.embedding.weight" in sd:
        out["num_train_cameras"] = int(sd["appearance_embedding.embedding.weight"].shape[0])
    if "video_embedding.embedding.weight" in sd:
        out["num_train_videos"] = int(sd["video_embedding.embedding.weight"].shape[0])
    return out


def build_model_from_checkpoint(model_config, path_or_ckpt, device=None, load_step: Optional[int] = None, **overrides):
    """`model_config.setup(...)` with the data-manager arguments taken from the checkpoint (model_kwargs_from_pipeline; `overrides`, e.g.
    dino_to_rgb, win), then load_checkpoint on the evaluation path.  -> (model in eval mode, step)"""
    ckpt = path_or_ckpt if isinstance(path_or_ckpt, dict) else read_checkpoint(path_or_ckpt, load_step)
    kw = model_kwargs_from_pipeline(ckpt["pipeline"])
    kw.update(overrides)
    kw.setdefault("num_train_cameras", 1)
    kw.setdefault("num_train_videos", 1)
    model = model_config.setup(scene_box=None, num_train_data=-1, **kw)
    if device is not None:
        model = model.to(device)
    step = load_checkpoint(ckpt, model)
    model.eval()
    return model, step


# --------------------------------------------------------------------------------------------------------- optimizer / scheduler layouts
def _adam_state_dict(opt, params: List[torch.nn.Parameter], index: Dict[int, int], steps: List[int], initial_lr: Optional[float]) -> Dict:
    """torch.optim.Adam.state_dict() of one parameter group: per-parameter state only for parameters that were ever stepped (torch
    creates it lazily on the first step with a gradient)"""
    state = {}
    for j, p in enumerate(params):
        i = index[id(p)]
        if steps[i] > 0:
            state[j] = {"step": torch.tensor(float(steps[i])), "exp_avg": opt.exp_avg[i].detach().cpu().clone(),
                        "exp_avg_sq": opt.exp_avg_sq[i].detach().cpu().clone()}
    group = {"lr": opt.lr, "betas": tuple(opt.betas), "eps": opt.eps, "weight_decay": opt.weight_decay, "amsgrad": False, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None}
    if initial_lr is not None:
        group["initial_lr"] = initial_lr  # (added by torch's LRScheduler constructor)
    group["params"] = list(range(len(params)))
    return {"state": state, "param_groups": [group]}


def _chained_scheduler_state(sched) -> Dict:
    """ChainedScheduler([LinearLR(start_factor, total_iters=warmup), MultiStepLR(milestones, gamma)]).state_dict() after `t` steps
    (ns/engine/my_schedulers.py:50-70).  The chain applies LinearLR's step first: its `_last_lr` still carries the previous step's
    milestone factor."""
    t, init = int(sched.t), float(sched.lr_init)
    lr = sched.lr_at(t)
    warm = lambda s: 1.0 if sched.warmup <= 0 else sched.start_factor + (1.0 - sched.start_factor) * min(s, sched.warmup) / sched.warmup  # noqa: E731
    n_prev = sum(1 for m in sched.milestones if m <= max(t - 1, 0))
    common = {"base_lrs": [init], "last_epoch": t, "_step_count": t + 1, "_get_lr_called_within_step": False}
    linear = {"start_factor": sched.start_factor, "end_factor": 1.0, "total_iters": sched.warmup, **common,
              "_last_lr": [init * warm(t) * sched.gamma ** n_prev]}
    multi = {"milestones": collections.Counter(sched.milestones), "gamma": sched.gamma, **common, "_last_lr": [lr]}
    return {"_schedulers": [linear, multi], "_last_lr": [lr]}


def _scheduler_position(sd: Dict) -> int:
    if "_schedulers" in sd:
        return int(sd["_schedulers"][0]["last_epoch"])
    if "last_epoch" in sd:
        return int(sd["last_epoch"])
    return int(sd["t"])  # (Trainer.state_dict's own layout)


# --------------------------------------------------------------------------------------------------------- trainer <-> checkpoint
def reference_checkpoint(trainer, step: Optional[int] = None, ddp: bool = False) -> Dict:
    """what Trainer.save_checkpoint(step) of the reference would torch.save for this trainer (called after iteration `step`;
    default: the last completed one)"""
    trainer.join()
    model, opt = trainer.model, trainer.opt
    step = trainer.step_idx - 1 if step is None else int(step)
    if step < 0:
        raise ValueError(f"reference_checkpoint: step {step} < 0 (no iteration has completed; the reference names files step-%09d.ckpt)")
    index = {id(p): i for i, p in enumerate(opt.params)}
    steps = opt.param_steps()
    if getattr(opt, "flat", None) is not None:  # sharded exchange: every rank's moments are valid on its shard only -> gather (collective)
        opt.flat_grads.gather_flat(opt.flat[2])
        opt.flat_grads.gather_flat(opt.flat[3])
    groups = model.get_param_groups()
    sched = trainer.scheduler
    ckpt = {"step": step, "pipeline": pipeline_state(model, ddp=ddp), "optimizers": {}, "schedulers": {}}
    for name, params in groups.items():
        seen, uniq = set(), []
        for p in params:
            if id(p) not in seen and id(p) in index:
                seen.add(id(p))
                uniq.append(p)
        ckpt["optimizers"][name] = _adam_state_dict(opt, uniq, index, steps, None if sched is None else sched.lr_init)
        if sched is not None:
            ckpt["schedulers"][name] = _chained_scheduler_state(sched)
    # GradScaler.state_dict() (the reference builds the scaler even in fp32, trainer.py:132)
    ckpt["scalers"] = {"scale": float(trainer.loss_scale), "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000,
                       "_growth_tracker": int(trainer._growth_tracker)}
    ps = model.proposal_sampler
    ckpt["presight_amd"] = {"proposal_sampler": {"steps_since_update": int(ps._steps_since_update), "step": int(ps._step), "anneal": float(ps._anneal)},
                            "optimizer_step_count": int(opt.step_count)}
    return ckpt


def _rank() -> int:
    return torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0


def save_checkpoint(trainer, checkpoint_dir: str, step: Optional[int] = None, save_only_latest_checkpoint: bool = True, ddp: bool = False,
                    write: Optional[bool] = None) -> str:
    """Trainer.save_checkpoint (ns/engine/trainer.py:432-460): step-%09d.ckpt in `checkpoint_dir`, older files removed when
    `save_only_latest_checkpoint` (the reference's default).  Under data parallelism call it on EVERY rank: with the sharded exchange
    the moment gather inside `reference_checkpoint` is a collective.  Only ONE rank touches the file system -- `write` (default: rank 0
    of the default process group, as the reference's `@check_main_thread` saves on the main process only); the other ranks return
    the path without building host copies.  The file is written under a temporary name and moved into place (`os.replace`) BEFORE the
    older checkpoints are removed, so a crash mid-save never leaves a truncated latest checkpoint next to nothing."""
    if step is not None and int(step) < 0:
        raise ValueError(f"save_checkpoint: step {step} < 0")
    if trainer.step_idx < 1 and step is None:
        raise ValueError("save_checkpoint: no iteration has completed yet (the reference saves after an iteration: step >= 0); pass step=")
    write = (_rank() == 0) if write is None else bool(write)
    if not write:
        # the collective part only (sharded exchange: every rank's moments are valid on its shard) -- no host copies, no file
        trainer.join()
        opt = trainer.opt
        if getattr(opt, "flat", None) is not None:
            opt.flat_grads.gather_flat(opt.flat[2])
            opt.flat_grads.gather_flat(opt.flat[3])
        return checkpoint_path(checkpoint_dir, trainer.step_idx - 1 if step is None else int(step))
    ckpt = reference_checkpoint(trainer, step, ddp=ddp)
    os.makedirs(str(checkpoint_dir), exist_ok=True)
    path = checkpoint_path(checkpoint_dir, ckpt["step"])
    tmp = f"{path}.tmp.{os.getpid()}"
    try:
        torch.save(ckpt, tmp)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)
    if save_only_latest_checkpoint:
        for f in os.listdir(str(checkpoint_dir)):
            full = os.path.join(str(checkpoint_dir), f)
            if full != path and os.path.isfile(full) and f.endswith(".ckpt"):
                os.unlink(full)
    return path


def read_checkpoint(path_or_dir: str, load_step: Optional[int] = None) -> Dict:
    """torch.load of a checkpoint file, or of step `load_step` / the newest step of a checkpoint directory"""
    path = str(path_or_dir)
    if os.path.isdir(path):
        path = checkpoint_path(path, latest_step(path) if load_step is None else load_step)
    if not os.path.exists(path):
        raise FileNotFoundError(f"Checkpoint {path} does not exist")
    return torch.load(path, map_location="cpu", weights_only=False)


def load_checkpoint(path_or_dir, model, trainer=None, load_step: Optional[int] = None, load_scheduler: bool = True,
                    restore_sampler: bool = True) -> int:
    """Trainer._load_checkpoint (ns/engine/trainer.py:396-429) / eval_load_checkpoint (ns/utils/eval_utils.py:38-65).
    trainer=None: the evaluation path -- only the model is restored (what `eval_setup` does before `extract_voxels`).  With a trainer
    (built on `model` BEFORE the call): Adam moments and per-parameter step counts, learning-rate schedule position, loss scale; the
    next iteration is `step + 1`.  `path_or_dir` may also be an already loaded checkpoint dict.  -> the checkpoint's step."""
    ckpt = path_or_dir if isinstance(path_or_dir, dict) else read_checkpoint(path_or_dir, load_step)
    step = int(ckpt["step"])
    if step < 0:
        raise ValueError(f"load_checkpoint: step {step} < 0 (a checkpoint is written after a completed iteration)")
    with torch.no_grad():
        load_pipeline(model, ckpt["pipeline"], step)
    if trainer is None:
        return step
    if trainer.model is not model:
        raise ValueError("load_checkpoint: `trainer` must have been built on `model`")
    trainer.join()
    opt = trainer.opt
    index = {id(p): i for i, p in enumerate(opt.params)}
    groups = model.get_param_groups()
    missing = [g for g in groups if g not in ckpt["optimizers"]]
    if missing:
        raise KeyError(f"checkpoint has no optimizer state for the parameter group(s) {missing}")
    steps = [0] * len(opt.params)
    dev = opt.params[0].device
    exp_avg = [torch.zeros_like(p) for p in opt.params]
    exp_avg_sq = [torch.zeros_like(p) for p in opt.params]
    lr, lr_group, first = None, None, None
    for name, params in groups.items():
        osd = ckpt["optimizers"][name]
        seen, uniq = set(), []
        for p in params:
            if id(p) not in seen and id(p) in index:
                seen.add(id(p))
                uniq.append(p)
        pg = osd["param_groups"][0]
        if len(pg["params"]) != len(uniq):
            raise ValueError(f"optimizer group {name!r}: the checkpoint has {len(pg['params'])} parameters, the model {len(uniq)}")
        state = {int(k): v for k, v in osd["state"].items()}
        for j, p in enumerate(uniq):
            st = state.get(pg["params"][j], state.get(j))
            if st is None:
                continue  # never stepped (a sub-field without samples so far, a sky sub-field no ray reached)
            i = index[id(p)]
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer group {name!r}, parameter {j}: moment shape {tuple(st['exp_avg'].shape)} != {tuple(p.shape)}")
            steps[i] = int(float(st["step"]))
            exp_avg[i] = st["exp_avg"].to(dev, torch.float32)
            exp_avg_sq[i] = st["exp_avg_sq"].to(dev, torch.float32)
        this = (float(pg["lr"]), None if pg.get("initial_lr") is None else float(pg["initial_lr"]))
        if lr is None:
            lr, lr_group, first = this[0], name, this
        elif this != first:
            # HipAdam steps every group with ONE learning rate and ONE schedule (the PreSight method configs give "proposal_networks" and
            # "fields" identical optimizers, method_configs.py:158-168): a checkpoint whose groups disagree cannot be resumed faithfully
            raise ValueError(f"load_checkpoint: optimizer groups {lr_group!r} and {name!r} disagree on (lr, initial_lr): {first} vs {this}; "
                             "this trainer steps all groups with one learning rate")
    extra = ckpt.get("presight_amd", {})
    opt.load_state_dict({"step": int(extra.get("optimizer_step_count", step + 1)), "steps": steps, "exp_avg": exp_avg, "exp_avg_sq": exp_avg_sq})
    if lr is not None:
        opt.lr = lr
    if trainer.scheduler is not None and load_scheduler and ckpt.get("schedulers"):
        positions = {n: _scheduler_position(sd) for n, sd in ckpt["schedulers"].items()}
        if len(set(positions.values())) != 1:
            raise ValueError(f"load_checkpoint: the schedulers of the optimizer groups are at different positions {positions}; "
                             "this trainer steps one schedule for all groups")
        trainer.scheduler.load_state_dict({"t": next(iter(positions.values()))})
    sc = ckpt.get("scalers") or {}
    if "scale" in sc:  # ({} when the reference ran without CUDA: torch's GradScaler is disabled there)
        trainer.loss_scale = float(sc["scale"])
        trainer._growth_tracker = int(sc.get("_growth_tracker", 0))
    trainer.step_idx = step + 1  # trainer.py:411: _start_step = loaded_state["step"] + 1
    ps = extra.get("proposal_sampler")
    if restore_sampler and ps is not None:
        s = model.proposal_sampler
        s._steps_since_update, s._step = int(ps["steps_since_update"]), int(ps["step"])
        s.set_anneal(float(ps["anneal"]))
    return step
