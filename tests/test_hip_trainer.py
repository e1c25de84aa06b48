"""presight_amd.trainer.Trainer against the REFERENCE's own training loop: tests/golden/model_traj.npz is 24 iterations of the
reference's NerfactoNuscMSModel (K = 3) under its own callbacks, Optimizers (torch.optim.Adam per parameter group on the
2**10-scaled gradients -- PreSight's update_grad_scaler=False, ns/engine/trainer.py:481-486) and WarmupMultiStepScheduler."""
import numpy as np
import pytest
import torch

from conftest import (build_hip_model, checkpoint_from_fixture, load_golden, model_traj_setup, t, to_double, traj_param_error,
                      traj_param_max_diff)

pytestmark = pytest.mark.gpu


def _dev_batch(b, dev):
    return {k: v.to(dev) for k, v in b.items()}


def _scene_dev(scene, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}


def test_trainer_follows_the_reference_trajectory(gold_model_traj):
    from oracle import nerf_oracle as O
    from presight_amd.trainer import Trainer

    G = gold_model_traj
    dev = torch.device("cuda:0")
    cfg, scene, P, batches = model_traj_setup(G)
    M, N = int(G["max_iterations"]), int(G["n_steps"])
    model = build_hip_model(cfg, scene, P, dev, proposal_weights_anneal_max_num_iters=M // 10, proposal_warmup=M // 10)
    tr = Trainer(model, _scene_dev(scene, dev), loss_scale=float(G["loss_scale"]), max_num_iterations=M)
    assert tr.opt.grad_scale == 1.0 and not tr.update_grad_scaler  # the reference default: Adam sees the scaled gradients
    names = {id(p): n for n, p in model.named_parameters()}
    pidx = {}
    for i, p in enumerate(tr.opt.params):  # (aliases: the first registered name that is a reference key)
        for n, q in model.named_parameters(remove_duplicate=False):
            if q is p and n in P:
                pidx[n] = i
    assert set(pidx) == set(P)
    losses, lrs, steps_before, snaps = [], [], tr.opt.param_steps(), {}
    touched = {k: [] for k in P}
    for s in range(N):
        lrs.append(tr.opt.lr)
        ld, out = tr.step(_dev_batch(batches[s], dev))
        assert abs(model.proposal_sampler._anneal - float(G["anneal"][s])) < 1e-12
        losses.append([float(ld[str(n)].detach()) for n in G["loss_names"]])
        steps_now = tr.opt.param_steps()
        for k, i in pidx.items():
            touched[k].append(steps_now[i] - steps_before[i])
        steps_before = steps_now
        if s == 11:
            snaps[11] = {k: model.state_dict()[k].detach().cpu().clone() for k in P}
    np.testing.assert_allclose(lrs, G["lr"], rtol=1e-12)
    # which parameters were stepped in which iteration: proposal networks off schedule and sub-fields without samples are skipped
    for i, k in enumerate(G["keys"]):
        assert touched[str(k)] == G["touched"][:, i].tolist(), (k, touched[str(k)], G["touched"][:, i].tolist())
    # bounds: the pinned oracle's own fp32-vs-fp64 distance on the same run
    r32 = O.train_trajectory(P, cfg, scene, batches, M, loss_scale=float(G["loss_scale"]), snapshots=(11,))
    r64 = O.train_trajectory(to_double(P), cfg, to_double(scene), to_double(batches), M, loss_scale=float(G["loss_scale"]), snapshots=(11,))
    l32, l64, ref, got = np.array(r32["losses"]), np.array(r64["losses"]), G["losses"], np.array(losses)
    bound = 2e-4 * np.abs(ref) + 4 * np.abs(l32 - l64) + 1e-8
    assert (np.abs(got - ref) <= bound).all(), (np.argwhere(np.abs(got - ref) > bound), np.abs(got - ref).max())
    final = {k: model.state_dict()[k].detach().cpu() for k in P}
    worst = 0.0
    for tag, mine, p32, p64 in (("S11", snaps[11], r32["snaps"][11], r64["snaps"][11]), ("S23", final, r32["params"], r64["params"])):
        want = {k: t(G[f"{tag}_{k}"]) for k in P}
        err, noise = traj_param_error(mine, want, P), traj_param_error(p32, p64, P)
        bad = {k: (f"{e:.1e}", f"oracle noise {noise[k]:.1e}") for k, e in err.items() if e > max(5e-5, 4 * noise[k])}
        assert not bad, (tag, bad)
        assert traj_param_max_diff(mine, want) <= 4 * float(G["lr"].max())  # the trimmed entries: a few Adam sign flips at most
        worst = max(worst, max(err.values()))
    print(f"24 reference iterations: max loss deviation {np.abs(got - ref).max():.1e}, worst parameter distance / movement {worst:.1e}")


def test_learnable_scene_psnr_matches_the_oracle_run():
    """"PSNR vs ref" as a TRAINING figure: the teacher-rendered scene at fixture size, the same 40-iteration
    miniature run (max_iterations = 40) on the HIP trainer and on the oracle -- same targets, same stored jitters.  The teacher's
    targets rendered by presight_amd.synthetic.TeacherScene equal the oracle's; the held-out eval PSNR rises on both sides and
    lands within 0.1 dB of the oracle's."""
    from conftest import learnable_scene_setup
    from oracle import nerf_oracle as O
    from presight_amd.synthetic import TEACHER_FAR, TeacherScene, eval_psnr
    from presight_amd.trainer import Trainer

    dev = torch.device("cuda:0")
    cfg, scene, Pt, batches, test, P0 = learnable_scene_setup()
    sdev = _scene_dev(scene, dev)
    tcfg = dict(cfg)
    tcfg["far"] = TEACHER_FAR
    teacher = TeacherScene(build_hip_model(tcfg, scene, Pt, dev), sdev)
    tgt = teacher.targets(test["ray_indices"].to(dev), test["video_ids"].to(dev))
    for k, tol in (("rgb", 1e-4), ("features", 1e-4), ("accumulation", 1e-4)):  # one PDF-resampled bin edge moved by an ulp shifts a ray by a few 1e-5
        assert float((tgt[k].cpu().reshape(test[k].shape) - test[k]).abs().max()) < tol, k
    assert torch.equal(tgt["sky"].cpu(), test["sky"])
    K = len(batches)
    model = build_hip_model(cfg, scene, P0, dev, proposal_weights_anneal_max_num_iters=K // 10, proposal_warmup=K // 10)
    tr = Trainer(model, sdev, max_num_iterations=K)
    tri, tvid, trgb = test["ray_indices"].to(dev), test["video_ids"].to(dev), test["rgb"].to(dev)
    marks = (9, 19, 29, K - 1)
    got = [eval_psnr(model, sdev, tri, tvid, trgb)]
    for s in range(K):
        tr.step(_dev_batch({k: v for k, v in batches[s].items() if k != "accumulation"}, dev))
        if s in marks:
            got.append(eval_psnr(model, sdev, tri, tvid, trgb))
    r = O.train_trajectory(P0, cfg, scene, batches, K, snapshots=marks)
    want = [O.eval_psnr(P0, cfg, scene, test["ray_indices"], test["video_ids"], test["rgb"])]
    want += [O.eval_psnr(r["snaps"][s], cfg, scene, test["ray_indices"], test["video_ids"], test["rgb"]) for s in marks]
    print("PSNR vs teacher after 0/10/20/30/40 iterations  HIP:", [round(x, 3) for x in got], " oracle:", [round(x, 3) for x in want])
    assert got[-1] > got[0] + 6.0
    assert all(abs(a - b) < 0.1 for a, b in zip(got, want)), (got, want)


def test_update_grad_scaler_branch_skips_the_group_with_an_inf(gold_model_traj):
    """The reference's other branch (update_grad_scaler=True; ns/engine/optimizers.py:118-131, trainer.py:496-505): GradScaler.step
    per parameter group -- a group whose gradients hold an inf / nan is not stepped --, the scale halves, the schedulers wait."""
    from presight_amd.trainer import Trainer

    G = gold_model_traj
    dev = torch.device("cuda:0")
    cfg, scene, P, batches = model_traj_setup(G)
    model = build_hip_model(cfg, scene, P, dev, proposal_weights_anneal_max_num_iters=6, proposal_warmup=6)
    tr = Trainer(model, _scene_dev(scene, dev), loss_scale=1024.0, update_grad_scaler=True, max_num_iterations=60)
    tr.step(_dev_batch(batches[0], dev))
    assert tr.loss_scale == 1024.0 and tr.opt.grad_scale == 1.0 / 1024
    lr1 = tr.opt.lr
    before = {k: v.detach().clone() for k, v in model.state_dict().items() if k in P}
    bad = _dev_batch(batches[1], dev)
    bad["rgb"] = bad["rgb"].clone()
    bad["rgb"][0, 0] = float("inf")  # -> rgb loss inf -> non-finite gradients in the "fields" group only
    tr.step(bad)
    after = {k: v.detach() for k, v in model.state_dict().items() if k in P}
    moved = {k: not torch.equal(before[k], after[k]) for k in P}
    assert not any(v for k, v in moved.items() if not k.startswith("proposal_networks.")), [k for k, v in moved.items() if v]
    assert any(v for k, v in moved.items() if k.startswith("proposal_networks."))  # its own GradScaler.step found no inf
    assert tr.loss_scale == 512.0 and tr.opt.lr == lr1  # scale decreased -> scheduler_step_all is not called
    tr.step(_dev_batch(batches[2], dev))
    assert tr.opt.lr > lr1 and all(torch.isfinite(v).all() for v in model.state_dict().values())


def test_checkpoint_resume_continues_the_run(gold_model_traj):
    """Trainer.state_dict / load_state_dict (what ns/engine/trainer.py:432-460 saves: step, optimizers, schedulers, scalers): a run that is
    saved after 12 iterations (an off-schedule proposal step, a sub-field that has been skipped: uneven Adam step counts) and resumed in a
    FRESH model + trainer continues like the uninterrupted one -- same learning rates, same pattern of stepped parameters, losses within
    the run-to-run noise of the step, final parameters within the trimmed 2-norm bound of the trajectory test."""
    import copy

    from presight_amd.trainer import Trainer

    G = gold_model_traj
    dev = torch.device("cuda:0")
    cfg, scene, P, batches = model_traj_setup(G)
    M = int(G["max_iterations"])

    def fresh():
        model = build_hip_model(cfg, scene, P, dev, proposal_weights_anneal_max_num_iters=M // 10, proposal_warmup=M // 10)
        return model, Trainer(model, _scene_dev(scene, dev), loss_scale=float(G["loss_scale"]), max_num_iterations=M)

    model_a, tr_a = fresh()
    for s in range(12):
        tr_a.step(_dev_batch(batches[s], dev))
    ckpt = {"pipeline": copy.deepcopy(model_a.state_dict()), "trainer": copy.deepcopy(tr_a.state_dict())}  # (what torch.save would hold)
    model_b, tr_b = fresh()
    model_b.load_state_dict(ckpt["pipeline"])
    tr_b.load_state_dict(ckpt["trainer"])
    assert tr_b.step_idx == 12 and tr_b.opt.lr == tr_a.opt.lr and tr_b.opt.param_steps() == tr_a.opt.param_steps()
    assert len(set(tr_b.opt.param_steps())) > 1
    for k in P:
        assert torch.equal(model_a.state_dict()[k], model_b.state_dict()[k]), k
    la, lb = [], []
    for s in range(12, 24):
        a, _ = tr_a.step(_dev_batch(batches[s], dev))
        b, _ = tr_b.step(_dev_batch(batches[s], dev))
        assert tr_a.opt.lr == tr_b.opt.lr and abs(tr_a.opt.lr - float(G["lr"][s + 1] if s + 1 < 24 else tr_a.opt.lr)) < 1e-12
        la.append([float(v.detach()) for v in a.values()])
        lb.append([float(v.detach()) for v in b.values()])
    assert tr_a.opt.param_steps() == tr_b.opt.param_steps()
    np.testing.assert_allclose(np.array(lb), np.array(la), rtol=2e-4, atol=1e-7)
    fa = {k: model_a.state_dict()[k].detach().cpu() for k in P}
    fb = {k: model_b.state_dict()[k].detach().cpu() for k in P}
    err = traj_param_error(fb, fa, P)
    assert max(err.values()) < 1e-4 and traj_param_max_diff(fb, fa) <= 4 * float(G["lr"].max()), sorted(err.items(), key=lambda kv: -kv[1])[:3]


def test_fused_table_adam_equals_the_separate_optimizer_step(gold_model_traj):
    """Single-process training applies the hash tables' Adam step inside their table backward (ps_grid_scatter_binned_adam /
    ps_grid_scatter_binned_ms_adam: the gradient is never written) -- the default of Trainer(world=1).  It must be the SAME update as
    backward -> optimizer.step() (ns/engine/trainer.py:470-486): the first iteration from equal parameters leaves bit-equal tables and
    moments (the table gradient is integer-accumulated = deterministic, and both paths run one shared element update); over the 24
    reference iterations (K = 3 routed tile, off-schedule proposal steps, a sub-field without samples: device-decided skips) the step
    counts agree exactly and the parameters stay within the run-to-run noise of the loop (the MLP gradients use float atomics)."""
    from presight_amd.trainer import Trainer

    G = gold_model_traj
    dev = torch.device("cuda:0")
    cfg, scene, P, batches = model_traj_setup(G)
    M, N = int(G["max_iterations"]), int(G["n_steps"])
    runs = []
    for fused in (True, False):
        model = build_hip_model(cfg, scene, P, dev, proposal_weights_anneal_max_num_iters=M // 10, proposal_warmup=M // 10)
        tr = Trainer(model, _scene_dev(scene, dev), loss_scale=float(G["loss_scale"]), max_num_iterations=M, fused_table_adam=fused)
        assert tr.fused_table_adam == fused
        runs.append((model, tr))
    (ma, ta), (mb, tb) = runs
    tables = [i for i, p in enumerate(ta.opt.params) if getattr(p, "_ps_fused_adam", None) is ta.opt]
    assert len(tables) == 9  # 3 sub-fields x (main table + 2 proposal tables)
    for s in range(N):
        ta.step(_dev_batch(batches[s], dev))
        tb.step(_dev_batch(batches[s], dev))
        if s == 0:
            for i in tables:
                assert torch.equal(ta.opt.params[i], tb.opt.params[i]), i
                assert torch.equal(ta.opt.exp_avg[i], tb.opt.exp_avg[i]) and torch.equal(ta.opt.exp_avg_sq[i], tb.opt.exp_avg_sq[i]), i
                assert float(ta.opt.params[i].grad.abs().max()) == 0.0  # the fused path never writes the gradient
        assert ta.opt.param_steps() == tb.opt.param_steps(), s
    fa = {k: ma.state_dict()[k].detach().cpu() for k in P}
    fb = {k: mb.state_dict()[k].detach().cpu() for k in P}
    err = traj_param_error(fa, fb, P)
    assert max(err.values()) < 1e-4 and traj_param_max_diff(fa, fb) <= 4 * float(G["lr"].max()), sorted(err.items(), key=lambda kv: -kv[1])[:3]


def test_resume_from_a_reference_checkpoint_continues_the_reference_run(gold_model_traj, tmp_path):
    """tests/golden/checkpoint.npz = the file the REFERENCE's Trainer.save_checkpoint objects wrote after iteration 11 of the trajectory
    run (ns/engine/trainer.py:432-460: `_model.`-prefixed pipeline state, one torch Adam state_dict and one ChainedScheduler state_dict
    per parameter group).  Written to `step-000000011.ckpt`, loaded through presight_amd.checkpoint.load_checkpoint into a ZERO-
    initialised HIP model + a fresh trainer (trainer.py:396-429), the run continues like the reference's own iterations 12..23:
    learning rates, which parameter is stepped when (uneven per-parameter Adam step counts: off-schedule proposal steps, sub-fields
    without samples), losses and final parameters inside the trajectory test's computed bounds."""
    from oracle import nerf_oracle as O
    from presight_amd import checkpoint as C
    from presight_amd.trainer import Trainer

    G = gold_model_traj
    ckpt, meta = checkpoint_from_fixture(load_golden("checkpoint"))
    dev = torch.device("cuda:0")
    cfg, scene, P, batches = model_traj_setup(G)
    M, N = int(G["max_iterations"]), int(G["n_steps"])
    model = build_hip_model(cfg, scene, {k: torch.zeros_like(v) for k, v in P.items()}, dev, proposal_weights_anneal_max_num_iters=M // 10,
                            proposal_warmup=M // 10)
    tr = Trainer(model, _scene_dev(scene, dev), loss_scale=float(G["loss_scale"]), max_num_iterations=M)
    torch.save(ckpt, C.checkpoint_path(tmp_path, 11))
    torch.save({**ckpt, "step": 5}, C.checkpoint_path(tmp_path, 5))  # (an older file in the directory: the newest step is taken)
    assert C.load_checkpoint(str(tmp_path), model, tr) == 11
    assert tr.step_idx == 12 and abs(tr.opt.lr - float(G["lr"][12])) < 1e-12
    for k in P:
        assert torch.equal(model.state_dict()[k].cpu(), t(G["S11_" + k])), k
    # torch Adam's per-parameter step counts of the reference run = how often each parameter had a gradient in iterations 0..11
    keys = [str(k) for k in G["keys"]]
    pidx = {}
    for i, p in enumerate(tr.opt.params):
        for n, q in model.named_parameters(remove_duplicate=False):
            if q is p and n in P:
                pidx[n] = i
    steps = tr.opt.param_steps()
    for j, k in enumerate(keys):
        assert steps[pidx[k]] == int(G["touched"][:12, j].sum()), k
    assert len(set(steps)) > 2
    # the sampler's update-schedule counters are not part of the reference's checkpoint (its resumed runs restart them); to continue
    # the UNINTERRUPTED run of model_traj.npz they are taken from the fixture's side record
    sm = meta["sampler_not_in_checkpoint"]
    ps = model.proposal_sampler
    ps._steps_since_update, ps._step = sm["steps_since_update"], sm["step"]
    ps.set_anneal(sm["anneal"])
    losses, before = [], steps
    touched = {k: [] for k in P}
    for s in range(12, N):
        assert abs(tr.opt.lr - float(G["lr"][s])) < 1e-12
        ld, _ = tr.step(_dev_batch(batches[s], dev))
        assert abs(ps._anneal - float(G["anneal"][s])) < 1e-12
        losses.append([float(ld[str(n)].detach()) for n in G["loss_names"]])
        now = tr.opt.param_steps()
        for k, i in pidx.items():
            touched[k].append(now[i] - before[i])
        before = now
    for j, k in enumerate(keys):
        assert touched[k] == G["touched"][12:, j].tolist(), k
    r32 = O.train_trajectory(P, cfg, scene, batches, M, loss_scale=float(G["loss_scale"]))
    r64 = O.train_trajectory(to_double(P), cfg, to_double(scene), to_double(batches), M, loss_scale=float(G["loss_scale"]))
    l32, l64, ref, got = np.array(r32["losses"])[12:], np.array(r64["losses"])[12:], G["losses"][12:], np.array(losses)
    bound = 2e-4 * np.abs(ref) + 4 * np.abs(l32 - l64) + 1e-8
    assert (np.abs(got - ref) <= bound).all(), (np.argwhere(np.abs(got - ref) > bound), np.abs(got - ref).max())
    final = {k: model.state_dict()[k].detach().cpu() for k in P}
    want = {k: t(G[f"S23_{k}"]) for k in P}
    err, noise = traj_param_error(final, want, P), traj_param_error(r32["params"], r64["params"], P)
    bad = {k: (f"{e:.1e}", f"oracle noise {noise[k]:.1e}") for k, e in err.items() if e > max(5e-5, 4 * noise[k])}
    assert not bad, bad
    assert traj_param_max_diff(final, want) <= 4 * float(G["lr"].max())


def test_written_checkpoint_has_the_reference_layout_and_resumes(gold_model_traj, tmp_path):
    """presight_amd.checkpoint.save_checkpoint after the same 12 iterations: `step-000000011.ckpt` holds the reference checkpoint's
    structure (tests/golden/checkpoint.npz) -- the five top-level keys, the same pipeline keys / shapes / dtypes in the same order, per
    optimizer group the same parameters WITH state and the same step counts, the same hyper-parameters, the same scheduler state, a
    GradScaler state -- with values inside the trajectory bounds; older files are removed (save_only_latest_checkpoint); a fresh model
    + trainer resumed from the FILE continues like the uninterrupted run."""
    from presight_amd import checkpoint as C
    from presight_amd.trainer import Trainer

    G = gold_model_traj
    ref, meta = checkpoint_from_fixture(load_golden("checkpoint"))
    dev = torch.device("cuda:0")
    cfg, scene, P, batches = model_traj_setup(G)
    M = int(G["max_iterations"])

    def fresh(params):
        model = build_hip_model(cfg, scene, params, dev, proposal_weights_anneal_max_num_iters=M // 10, proposal_warmup=M // 10)
        return model, Trainer(model, _scene_dev(scene, dev), loss_scale=float(G["loss_scale"]), max_num_iterations=M)

    model_a, tr_a = fresh(P)
    for s in range(6):
        tr_a.step(_dev_batch(batches[s], dev))
    C.save_checkpoint(tr_a, tmp_path)
    for s in range(6, 12):
        tr_a.step(_dev_batch(batches[s], dev))
    path = C.save_checkpoint(tr_a, tmp_path)
    assert sorted(f.name for f in tmp_path.iterdir()) == ["step-000000011.ckpt"] and path.endswith("step-000000011.ckpt")
    mine = torch.load(path, map_location="cpu", weights_only=False)
    assert set(mine) - {"presight_amd"} == set(ref) == {"step", "pipeline", "optimizers", "schedulers", "scalers"} and mine["step"] == ref["step"] == 11
    assert list(mine["pipeline"]) == list(ref["pipeline"])
    for k, v in mine["pipeline"].items():
        assert v.shape == ref["pipeline"][k].shape and v.dtype == ref["pipeline"][k].dtype, k
    init = {k: P[k[len("_model."):]] for k in ref["pipeline"] if k[len("_model."):] in P}
    err = traj_param_error({k: mine["pipeline"][k] for k in init}, {k: ref["pipeline"][k] for k in init}, init)
    assert max(err.values()) < 2e-3, sorted(err.items(), key=lambda kv: -kv[1])[:3]
    assert set(mine["optimizers"]) == set(ref["optimizers"]) == {"proposal_networks", "fields"}
    for g in ref["optimizers"]:
        a, b = mine["optimizers"][g], ref["optimizers"][g]
        assert sorted(a["state"]) == sorted(b["state"]), g  # the same parameters have been stepped at least once
        ga, gb = a["param_groups"][0], b["param_groups"][0]
        assert set(ga) <= set(gb) and set(gb) - set(ga) <= {"decoupled_weight_decay"} and ga["params"] == gb["params"]
        for k in ("betas", "eps", "weight_decay", "amsgrad", "maximize", "initial_lr"):
            assert ga[k] == gb[k], (g, k)
        assert abs(ga["lr"] - gb["lr"]) < 1e-12
        worst = 0.0
        for j in a["state"]:
            assert float(a["state"][j]["step"]) == float(b["state"][j]["step"]), (g, j)
            for mom in ("exp_avg", "exp_avg_sq"):
                x, y = a["state"][j][mom].double(), b["state"][j][mom].double()
                assert x.shape == y.shape
                worst = max(worst, float((x - y).norm() / y.norm().clamp_min(1e-30)))
        assert worst < 2e-2, (g, worst)  # (moments of 12 fp32 iterations: ReLU flips move single networks by 1e-3)
    assert set(mine["schedulers"]) == set(ref["schedulers"])
    sa, sb = mine["schedulers"]["fields"], ref["schedulers"]["fields"]
    assert sa["_schedulers"][0]["last_epoch"] == sb["_schedulers"][0]["last_epoch"] == 12
    assert sa["_schedulers"][1]["milestones"] == sb["_schedulers"][1]["milestones"] and abs(sa["_last_lr"][0] - sb["_last_lr"][0]) < 1e-12
    assert mine["scalers"]["scale"] == float(G["loss_scale"]) and set(mine["scalers"]) == {"scale", "growth_factor", "backoff_factor", "growth_interval",
                                                                                          "_growth_tracker"}
    # resume from the file (our own extra record restores the sampler counters) == the uninterrupted run
    model_b, tr_b = fresh({k: torch.zeros_like(v) for k, v in P.items()})
    assert C.load_checkpoint(path, model_b, tr_b) == 11
    assert tr_b.step_idx == tr_a.step_idx == 12 and tr_b.opt.param_steps() == tr_a.opt.param_steps() and tr_b.opt.lr == tr_a.opt.lr
    la, lb = [], []
    for s in range(12, 24):
        a, _ = tr_a.step(_dev_batch(batches[s], dev))
        b, _ = tr_b.step(_dev_batch(batches[s], dev))
        la.append([float(v.detach()) for v in a.values()])
        lb.append([float(v.detach()) for v in b.values()])
    assert tr_a.opt.param_steps() == tr_b.opt.param_steps() and tr_a.opt.lr == tr_b.opt.lr
    np.testing.assert_allclose(np.array(lb), np.array(la), rtol=2e-4, atol=1e-7)
    fa = {k: model_a.state_dict()[k].detach().cpu() for k in P}
    fb = {k: model_b.state_dict()[k].detach().cpu() for k in P}
    err = traj_param_error(fb, fa, P)
    assert max(err.values()) < 1e-4 and traj_param_max_diff(fb, fa) <= 4 * float(G["lr"].max()), sorted(err.items(), key=lambda kv: -kv[1])[:3]
