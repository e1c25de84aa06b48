"""Pin the CPU oracle (oracle/nerf_oracle.py) against fixtures produced by the reference itself
(tests/golden/make_golden.py) and against SURVEY.md Appendix A known answers.  CPU only."""
import numpy as np
import torch

from conftest import load_golden, model_fixture_setup, t
from oracle import nerf_oracle as O

F32 = dict(rtol=1e-5, atol=1e-6)


def close(a, b, **kw):
    kw = {**F32, **kw}
    a = a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(a)
    b = t(b) if isinstance(b, np.ndarray) else b
    torch.testing.assert_close(a.to(b.dtype).reshape(b.shape), b, **kw)


# ------------------------------------------------------------------------------ hash grid
def test_hash_scalings_match_reference(gold_hashgrid):
    G = gold_hashgrid
    for tag in ["kat", "cfg2small", "prodsmall", "prop0small", "prop1small", "cfg2_full", "prod_full", "prop0_full", "prop1_full"]:
        L, base, mx, _, _ = G[tag + "_meta"]
        sc = O.hash_scalings(int(L), int(base), int(mx))
        assert torch.equal(sc, t(G[tag + "_scalings"])), tag
    # values quoted in SURVEY.md 8a row a7
    assert O.hash_scalings(16, 16, 2048).tolist() == [16, 22, 30, 42, 58, 80, 111, 153, 212, 294, 406, 561, 776, 1072, 1482, 2047]
    assert O.hash_scalings(10, 16, 16384).tolist() == [16, 34, 74, 161, 348, 752, 1625, 3511, 7584, 16384]


def test_hash_encode_known_answers(gold_hashgrid):
    G = gold_hashgrid
    out, idx = O.hash_encode(t(G["kat_x"]), t(G["kat_table"]), t(G["kat_scalings"]), 5, return_indices=True)
    close(out[0], np.array([0.032, 0.033, 0.064, 0.065], np.float32))
    close(out[1], np.array([0.031488001, 0.032488003, 0.101247981, 0.102247983], np.float32))
    assert idx[0].tolist() == [[16] * 8, [32] * 8]
    assert idx[1, 0, 0] == 3 and idx[1, 1, 0] == 58 and idx[1, 0, 6] == 10 and idx[1, 1, 6] == 33
    assert idx[1, 0, 1] == 20 and idx[1, 1, 1] == 43


def test_hash_encode_matches_reference(gold_hashgrid):
    G = gold_hashgrid
    for tag in ["kat", "cfg2small", "prodsmall", "prop0small", "prop1small"]:
        l2t = int(G[tag + "_meta"][3])
        table = t(G[tag + "_table"]).clone().requires_grad_(True)
        out, idx = O.hash_encode(t(G[tag + "_x"]), table, t(G[tag + "_scalings"]), l2t, return_indices=True)
        assert torch.equal(idx, t(G[tag + "_idx"])), tag  # integer work: bit exact
        assert torch.equal(out.detach(), t(G[tag + "_out"])), tag  # same ATen ops in the same order -> bit exact on CPU
        (g,) = torch.autograd.grad((out * t(G[tag + "_cot"])).sum(), table)
        close(g, G[tag + "_grad_table"])


def test_hash_index_full_size(gold_hashgrid):
    G = gold_hashgrid
    for tag in ["cfg2_full", "prod_full", "prop0_full", "prop1_full"]:
        l2t = int(G[tag + "_meta"][3])
        sc = t(G[tag + "_scalings"])
        dummy = torch.zeros((1 << l2t) * sc.numel(), 1)
        _, idx = O.hash_encode(t(G[tag + "_x"]), dummy, sc, l2t, return_indices=True)
        assert torch.equal(idx, t(G[tag + "_idx"])), tag


# ------------------------------------------------------------------------------ small ops
def test_contraction(gold_ops):
    G = gold_ops
    u, sel = O.normalize_contract(t(G["p"]), t(G["aabb"]))
    assert torch.equal(sel, t(G["sel"]))
    assert torch.equal(u, t(G["u"]))
    # Appendix A.2 KAT
    q = torch.tensor([[2.0, -4.0, 1.0]])
    mag = q.abs().amax(-1, keepdim=True)
    close((2 - 1 / mag) * (q / mag), np.array([[0.875, -1.75, 0.4375]], np.float32))
    close(t(G["contract_kat"]), np.array([[0.875, -1.75, 0.4375]], np.float32))


def test_sh4(gold_ops):
    G = gold_ops
    sh = O.sh4_of_direction(t(G["d"]))
    assert torch.equal(sh, t(G["sh"]))
    kat = [0.28209481, 0.24430126, 0.48860252, 0.24430126, 0.27313712, 0.54627424, 0.63078308, 0.54627424, 0, 0.14751090,
           0.72265285, 0.91409159, 0.74635267, 0.91409159, 0, -0.14751090]
    close(sh[0], np.array(kat, np.float32), atol=1e-7)


def test_trunc_exp(gold_ops):
    G = gold_ops
    x = t(G["te_x"]).clone().requires_grad_(True)
    y = O.trunc_exp(x)
    assert torch.equal(y.detach(), t(G["te_y"]))
    (g,) = torch.autograd.grad((y * t(G["te_cot"])).sum(), x)
    assert torch.equal(g, t(G["te_gx"]))


def test_mlp_all_shapes(gold_ops):
    G = gold_ops
    for tag in ["base", "sem", "rgb", "prop", "skyrgb", "skysem", "base_prod", "tiny"]:
        n = len([k for k in G if k.startswith(f"mlp_{tag}_W")])
        layers = [(t(G[f"mlp_{tag}_W{i}"]).clone().requires_grad_(True), t(G[f"mlp_{tag}_b{i}"]).clone().requires_grad_(True))
                  for i in range(n)]
        x = t(G[f"mlp_{tag}_x"]).clone().requires_grad_(True)
        y = O.mlp_forward(x, layers, out_act="sigmoid" if int(G[f"mlp_{tag}_sigmoid"]) else None)
        close(y, G[f"mlp_{tag}_y"])
        gr = torch.autograd.grad((y * t(G[f"mlp_{tag}_cot"])).sum(), [x] + [p for wb in layers for p in wb])
        close(gr[0], G[f"mlp_{tag}_gx"])
        for i in range(n):
            close(gr[1 + 2 * i], G[f"mlp_{tag}_gW{i}"], rtol=1e-4, atol=1e-5)
            close(gr[2 + 2 * i], G[f"mlp_{tag}_gb{i}"], rtol=1e-4, atol=1e-5)


def test_router(gold_ops):
    G = gold_ops
    assert torch.equal(O.route(t(G["route_pts"]), t(G["route_centroids"])), t(G["route_assign"]))


# ------------------------------------------------------------------------------ rays, samplers, renderers
def test_ray_generation(gold_sampling):
    G = gold_sampling
    o, d, pa, dn = O.generate_rays(t(G["ray_indices"]), t(G["c2w"]), t(G["fx"]), t(G["fy"]), t(G["cx"]), t(G["cy"]))
    close(o, G["origins"])
    close(d, G["directions"], atol=1e-7)
    close(pa, G["pixel_area"], rtol=1e-4, atol=1e-12)
    close(dn, G["directions_norm"])


def _euclid(G, mode, bins):
    R = bins.shape[0]
    nears = torch.full((R, 1), 0.005 if mode == "train" else 0.0)
    fars = torch.full((R, 1), 50.0)
    return O.s_to_euclid(bins, nears, fars, 5.0)


def test_spaced_sampler(gold_sampling):
    G = gold_sampling
    for mode in ("train", "eval"):
        jit = t(G[f"sp_{mode}_jitter"]) if mode == "train" else None
        bins = O.spaced_bins(G["ray_indices"].shape[0], 128, jit)
        close(bins[:, :-1], G[f"sp_{mode}_sstarts"], atol=1e-7)
        close(bins[:, 1:], G[f"sp_{mode}_sends"], atol=1e-7)
        eu = _euclid(G, mode, bins)
        close(eu[:, :-1], G[f"sp_{mode}_starts"])
        close(eu[:, 1:], G[f"sp_{mode}_ends"])
        mid = (eu[:, :-1] + eu[:, 1:]) / 2
        pos = t(G["origins"])[:, None] + t(G["directions"])[:, None] * mid[..., None]
        close(pos, G[f"sp_{mode}_positions"])
    # Appendix A.5
    bins = O.spaced_bins(1, 8, None)
    eu = O.s_to_euclid(bins, torch.zeros(1, 1), torch.full((1, 1), 50.0), 5.0)
    close(eu[:, :-1], G["kat_sp_starts"])
    close(eu[0, :-1], np.array([0, 1.1875, 2.375, 3.5625, 4.75, 6.15384626, 8.69565105, 14.81481552], np.float32))
    close(eu[0, -1], np.array(49.99999237, np.float32))
    # Appendix A.6
    w = torch.zeros(1, 8)
    w[0, 2] = 1.0
    nb = O.pdf_resample(w, bins, 4, None)
    eu2 = O.s_to_euclid(nb, torch.zeros(1, 1), torch.full((1, 1), 50.0), 5.0)
    close(eu2[:, :-1], G["kat_pdf_starts"])
    close(eu2[:, 1:], G["kat_pdf_ends"])
    close(eu2[0, :-1], np.array([2.47846532, 2.73242569, 2.98638606, 3.24034643], np.float32))


def test_weights_scan(gold_sampling):
    G = gold_sampling
    for mode in ("train", "eval"):
        deltas = t(G[f"sp_{mode}_ends"]) - t(G[f"sp_{mode}_starts"])
        sigma = t(G[f"w_{mode}_sigma"]).clone().requires_grad_(True)
        w = O.weights_from_density(deltas, sigma)
        close(w, G[f"w_{mode}_weights"])
        (g,) = torch.autograd.grad((w * t(G[f"w_{mode}_cot"])).sum(), sigma)
        close(g, G[f"w_{mode}_gsigma"])
    # Appendix A.4
    w = O.weights_from_density(torch.tensor([[0.1, 0.2, 0.3, 0.4]]), torch.tensor([[1.0, 2.0, 0.0, 5.0]]))
    close(w[0], np.array([0.09516257, 0.29830676, 0, 0.52444565], np.float32))
    steps = torch.tensor([[0.05, 0.2, 0.45, 0.8]])
    close(O.expected_depth(w, steps), np.array([[0.52725583]], np.float32))
    close(O.threshold_depth(w, steps), np.array([[0.8]], np.float32))


def test_pdf_sampler(gold_sampling):
    G = gold_sampling
    for mode in ("train", "eval"):
        bins = torch.cat([t(G[f"sp_{mode}_sstarts"]), t(G[f"sp_{mode}_sends"])[:, -1:]], -1)
        w = torch.pow(t(G[f"w_{mode}_weights"]), float(G[f"pdf_{mode}_anneal"]))
        jit = t(G[f"pdf_{mode}_jitter"]) if mode == "train" else None
        nb = O.pdf_resample(w, bins, 64, jit)
        close(nb[:, :-1], G[f"pdf_{mode}_sstarts"], atol=1e-7)
        close(nb[:, 1:], G[f"pdf_{mode}_sends"], atol=1e-7)
        eu = _euclid(G, mode, nb)
        close(eu[:, :-1], G[f"pdf_{mode}_starts"])
        close(eu[:, 1:], G[f"pdf_{mode}_ends"])


def test_renderers(gold_sampling):
    G = gold_sampling
    for mode in ("train", "eval"):
        st, en = t(G[f"pdf_{mode}_starts"]), t(G[f"pdf_{mode}_ends"])
        w = O.weights_from_density(en - st, t(G[f"r_{mode}_sigma"]))
        close(w, G[f"r_{mode}_w"])
        steps = (st + en) / 2
        close((w[..., None] * t(G[f"r_{mode}_rgb_in"])).sum(1), G[f"r_{mode}_rgb"])
        close(w.sum(-1, keepdim=True), G[f"r_{mode}_acc"])
        assert torch.equal(O.threshold_depth(w, steps), t(G[f"r_{mode}_depth"]))
        close(O.expected_depth(w, steps), G[f"r_{mode}_expdepth"])
        close((w[..., None] * t(G[f"r_{mode}_sem_in"])).sum(1), G[f"r_{mode}_sem"])


# ------------------------------------------------------------------------------ losses
def test_losses(gold_losses):
    G = gold_losses
    raw = [t(G[f"w{i}_raw"]).clone().requires_grad_(True) for i in range(3)]
    wl = [w / w.sum(-1, keepdim=True) * 0.9 for w in raw]
    bl = [t(G[f"bins{i}"]) for i in range(3)]
    il = O.interlevel_loss_zaa(wl, bl, (0.03, 0.003))
    close(il, G["interlevel"])
    g = torch.autograd.grad(il, raw[:2], retain_graph=True)
    close(g[0], G["g_interlevel_w0"], rtol=1e-4)
    close(g[1], G["g_interlevel_w1"], rtol=1e-4)
    dl = O.distortion_loss(bl[2], wl[2])
    close(dl, G["distortion"])
    (g2,) = torch.autograd.grad(dl, raw[2])
    close(g2, G["g_distortion_w2"], rtol=1e-4)
    acc = t(G["acc"]).clone().requires_grad_(True)
    sl = O.sky_loss(acc, t(G["sky_mask"]))
    close(sl, G["sky_loss"])
    close(torch.autograd.grad(sl, acc)[0], G["g_sky"])
    pred = t(G["sem_pred"]).clone().requires_grad_(True)
    sm = O.semantic_loss(pred, t(G["sem_tgt"]))
    close(sm, G["sem_loss"])
    close(torch.autograd.grad(sm, pred)[0], G["g_sem"])


def test_depth_losses():
    """lidar / monodepth supervision vs the reference (tests/golden/make_golden.py::gold_depth_losses)"""
    G = load_golden("depth_losses")
    w = t(G["w"])[..., 0].clone().requires_grad_(True)
    steps, depth, sky = t(G["steps"])[..., 0], t(G["depth"])[:, 0], t(G["sky"])[:, 0]
    for tag, use_sky in (("lidar", False), ("mono", True)):
        los = O.line_of_sight_loss(w, depth, steps, float(G[f"sigma_{tag}"]), sky if use_sky else None, float(G[f"ub_{tag}"]))
        close(los, G[f"los_{tag}"])
        close(torch.autograd.grad(los, w)[0], G[f"g_los_{tag}"][..., 0], rtol=1e-4)
    pred = t(G["pred"])[:, 0].clone().requires_grad_(True)
    for tag, kw in (("lidar", dict(upper_bound=75.0)), ("mono", dict(upper_bound=40.0, sky_mask=sky)),
                    ("mono_inv", dict(upper_bound=40.0, sky_mask=sky, inverse=True))):
        ed = O.expected_depth_loss(depth, pred, **kw)
        close(ed, G[f"ed_{tag}"])
        close(torch.autograd.grad(ed, pred)[0], G[f"g_ed_{tag}"][:, 0], rtol=1e-4)
    assert O.line_of_sight_mult(1000) == 0.0 and O.line_of_sight_mult(1001) == 0.1 and O.line_of_sight_mult(12000) == 0.025
    assert O.line_of_sight_sigma(0) == 5.0 and O.line_of_sight_sigma(30000) == 2.0 and abs(O.line_of_sight_sigma(15500) - 3.5) < 1e-12


# ------------------------------------------------------------------------------ fields + whole model
def test_fields(gold_model):
    G = gold_model
    cfg, scene, P, _ = model_fixture_setup(G)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    pos, dirs, app = t(G["F_pos"]), t(G["F_dirs"]), t(G["F_app"])
    dens, emb = O.main_density(Pg, cfg, 1, pos, scene["aabbs"][1])
    rgb, sem = O.main_heads(Pg, cfg, 1, dirs, emb, app)
    close(dens, G["F_density"], rtol=1e-4)
    close(emb, G["F_embedding"], rtol=1e-4, atol=1e-5)
    close(rgb, G["F_rgb"], rtol=1e-4)
    close(sem, G["F_sem"], rtol=1e-4, atol=1e-5)
    close(sem, G["F_semantic_fn"], rtol=1e-4, atol=1e-5)
    s = (dens * t(G["F_cot_density"])[:, 0]).sum() + (rgb * t(G["F_cot_rgb"])).sum() + (sem * t(G["F_cot_sem"])).sum()
    s.backward()
    for k in G:
        if k.startswith("Fg_"):
            name = "field.fields.1." + k[3:]
            close(Pg[name].grad, G[k], rtol=2e-4, atol=2e-5)
    pd = O.prop_density(Pg, cfg, 0, 1, pos, scene["aabbs"][1])
    close(pd, G["Pp_density"], rtol=1e-4)
    for p in Pg.values():
        p.grad = None
    (pd * t(G["Pp_cot"])[:, 0]).sum().backward()
    for k in G:
        if k.startswith("Ppg_"):
            name = "proposal_networks.0.fields.1." + k[4:]
            close(Pg[name].grad, G[k], rtol=2e-4, atol=2e-5)
    srgb, ssem = O.sky_outputs(P, 1, dirs, app)
    close(srgb, G["S_rgb"])
    close(ssem, G["S_sem"], atol=1e-5)


def test_whole_model_training_step(gold_model):
    G = gold_model
    cfg, scene, P, batch = model_fixture_setup(G)
    L, out, grads = O.train_step(P, cfg, scene, batch, anneal=float(G["T_anneal"]))
    for i in range(3):
        close(out["bins_list"][i], G[f"T_sbins_{i}"], atol=2e-6)
        close(out["weights_list"][i], G[f"T_weights_{i}"], rtol=1e-4, atol=1e-6)
    for k in ["rgb", "accumulation", "expected_depth", "semantics"]:
        close(out[k], G["T_" + k], rtol=1e-4, atol=1e-5)
    for k in ["depth", "prop_depth_0", "prop_depth_1"]:
        close(out[k], G["T_" + k], rtol=1e-5, atol=1e-6)
    for k, v in L.items():
        close(v, G["TL_" + k], rtol=1e-4, atol=1e-7)
    n_checked = 0
    for k in G:
        if k.startswith("TG_"):
            ref = t(G[k])
            tol = 2e-4 * float(ref.abs().max()) + 1e-7
            close(grads[k[3:]], ref, rtol=2e-3, atol=tol)
            n_checked += 1
    assert n_checked == len(P)


def test_whole_model_training_step_k8_production_shape(gold_model_k8):
    """the oracle against the REFERENCE's own training step with K = 8 routed sub-fields at the production shape (L10 F4 up to
    resolution 16384, 64-wide MLPs): pins the restatement of the router + sub-field loop where the HIP multi-sub-field path is
    then tested against it at K = 16"""
    from conftest import model_k8_setup

    G = gold_model_k8
    cfg, scene, P, batch = model_k8_setup(G)
    L, out, grads = O.train_step(P, cfg, scene, batch, anneal=float(G["T_anneal"]))
    for i in range(3):
        close(out["weights_list"][i], G[f"T_weights_{i}"], rtol=1e-4, atol=1e-6)
    for k in ["rgb", "accumulation", "expected_depth", "semantics"]:
        close(out[k], G["T_" + k], rtol=1e-4, atol=1e-5)
    for k in ["depth", "prop_depth_0", "prop_depth_1"]:
        close(out[k], G["T_" + k], rtol=1e-5, atol=1e-6)
    for k, v in L.items():
        close(v, G["TL_" + k], rtol=2e-4, atol=1e-7)
    n_checked = 0
    for k in G:
        if k.startswith("TG_"):
            ref = t(G[k])
            close(grads[k[3:]], ref, rtol=2e-3, atol=3e-4 * float(ref.abs().max()) + 1e-7)
            n_checked += 1
    assert n_checked == int(G["n_grads"]) and n_checked > 100
    # sub-fields the reference never called (grad None there) have exactly zero gradients here
    for name, g in grads.items():
        if "TG_" + name not in G:
            assert float(g.abs().max()) == 0.0, name


def test_training_trajectory_matches_the_reference_loop(gold_model_traj):
    """24 iterations of the REFERENCE's own training loop (its model, callbacks, Optimizers: torch.optim.Adam per parameter group
    on the 2**10-scaled gradients, WarmupMultiStepScheduler; tests/golden/make_golden.py::gold_model_traj) against the oracle's
    restatement of that loop: learning rate, anneal, the proposal update schedule (11 warm-up steps, then every 6th), which
    parameters received a gradient in which step (off-schedule proposal networks, sub-fields without samples), the five losses
    of every step and the parameters after 12 and 24 steps.  Bounds are computed: the oracle's own fp32-vs-fp64 distance."""
    from conftest import model_traj_setup, to_double, traj_param_error, traj_param_max_diff

    G = gold_model_traj
    cfg, scene, P, batches = model_traj_setup(G)
    M = int(G["max_iterations"])
    r32 = O.train_trajectory(P, cfg, scene, batches, M, loss_scale=float(G["loss_scale"]), snapshots=(11,))
    r64 = O.train_trajectory(to_double(P), cfg, to_double(scene), to_double(batches), M, loss_scale=float(G["loss_scale"]), snapshots=(11,))
    assert r32["updated"] == G["updated"].tolist() and 0 in r32["updated"][11:] and 1 in r32["updated"][11:]
    np.testing.assert_allclose(r32["lr"], G["lr"], rtol=1e-12)
    np.testing.assert_allclose(r32["anneal"], G["anneal"], rtol=1e-12)
    for i, k in enumerate(G["keys"]):
        assert r32["touched"][str(k)] == G["touched"][:, i].tolist(), k
    assert [str(n) for n in G["loss_names"]] == ["rgb_loss", "sky_loss", "semantic_loss", "interlevel_loss", "distortion_loss"]
    l32, l64, ref = np.array(r32["losses"]), np.array(r64["losses"]), G["losses"]
    bound = 1e-5 * np.abs(ref) + 4 * np.abs(l32 - l64) + 1e-9
    assert (np.abs(l32 - ref) <= bound).all(), np.argwhere(np.abs(l32 - ref) > bound)
    for tag, p32, p64 in (("S11", r32["snaps"][11], r64["snaps"][11]), ("S23", r32["params"], r64["params"])):
        want = {k: t(G[f"{tag}_{k}"]) for k in P}
        err, noise = traj_param_error(p32, want, P), traj_param_error(p32, p64, P)
        bad = {k: (e, noise[k]) for k, e in err.items() if e > max(2e-5, 4 * noise[k])}
        assert not bad, (tag, bad)
        assert traj_param_max_diff(p32, want) <= 4 * float(G["lr"].max())
    # the fixture separates the two loss-scale semantics by five orders of magnitude: Adam on UNSCALED gradients (weight decay
    # 1024 x stronger relative to them) ends somewhere else
    r1 = O.train_trajectory(P, cfg, scene, batches, M, loss_scale=1.0)
    err1 = traj_param_error(r1["params"], {k: t(G[f"S23_{k}"]) for k in P}, P)
    assert max(err1.values()) > 0.5


def test_learnable_scene_psnr_rises_on_the_oracle():
    """SURVEY.md 8d "PSNR vs synthetic GT after K steps": a complete miniature run of the reference loop (max_iterations = 40:
    anneal, proposal schedule, LR warm-up + milestones) on the teacher-rendered scene.  Held-out eval PSNR
    (ns/models/PreSight/nerfacto_nusc_ms.py:548-556: 10 log10(1 / MSE)) must rise by several dB -- with the random targets of
    the throughput benchmark nothing is learnable and it cannot."""
    from conftest import learnable_scene_setup

    cfg, scene, Pt, batches, test, P0 = learnable_scene_setup()
    acc = test["accumulation"]
    assert 0.2 < float(acc.min()) and float((acc < 0.9).float().mean()) > 0.2 and float(test["rgb"].std()) > 0.1  # a scene, not a constant
    K = len(batches)
    r = O.train_trajectory(P0, cfg, scene, batches, K, snapshots=(9, 19, 29, K - 1))
    psnr = [O.eval_psnr(P0, cfg, scene, test["ray_indices"], test["video_ids"], test["rgb"])]
    psnr += [O.eval_psnr(r["snaps"][s], cfg, scene, test["ray_indices"], test["video_ids"], test["rgb"]) for s in (9, 19, 29, K - 1)]
    print("oracle PSNR vs teacher after 0/10/20/30/40 iterations:", [round(x, 2) for x in psnr])
    assert psnr[-1] > psnr[0] + 6.0 and all(b > max(psnr[:i + 1]) - 1.5 for i, b in enumerate(psnr[1:]))
    rgb_loss = [l[0] for l in r["losses"]]
    assert sum(rgb_loss[-5:]) < 0.4 * sum(rgb_loss[:5])


def test_whole_model_eval_and_extraction(gold_model):
    G = gold_model
    cfg, scene, P, batch = model_fixture_setup(G)
    with torch.no_grad():
        out = O.model_forward(P, cfg, scene, batch, training=False, anneal=float(G["T_anneal"]))  # sampler keeps its anneal in eval
    for k in ["rgb", "accumulation", "expected_depth", "semantics"]:
        close(out[k], G["E_" + k], rtol=1e-4, atol=1e-5)
    close(out["depth"], G["E_depth"])
    close(out["depth"], G["E_get_depth"])
    close(out["expected_depth"], G["E_get_expected_depth"], rtol=1e-4)
    close(O.feature_colormap(out["semantics"], scene["dino_to_rgb"]), G["E_dino_rgb"], rtol=1e-4, atol=1e-5)
    dens, feats = O.prior_query(P, cfg, scene, t(G["X_pts"]))
    close(dens, G["X_density_mean"], rtol=1e-4)
    assert feats.dtype == torch.float16
    d = (feats.float() - t(G["X_feats"]).float()).abs().max()
    assert d <= 2 ** -10, d  # at most one fp16 ulp in [0,1]
    close(O.feature_colormap(t(G["X_feats"]), scene["dino_to_rgb"]).float(), G["X_colors"].astype(np.float32), rtol=2e-3, atol=2e-3)


def test_voxel_index_rule():
    pts = torch.tensor([[0.0, 0.0, 0.0], [0.39, 0.0, -0.01], [0.41, 0.8, 1.2], [-0.2, -0.2, -0.2]])
    mn = pts.min(0).values - 1.0
    idx = O.voxel_index(pts, 0.4, mn)
    ref = np.floor((pts.double().numpy() - (mn.double().numpy() - 0.2)) / 0.4).astype(np.int64)
    assert np.array_equal(idx.numpy(), ref)
    assert (idx >= 0).all()


def test_losses_on_real_sampler_bins():
    """interlevel / distortion losses on bins produced by the reference's own sampler chain (tests/golden/losses_real.npz)"""
    G = load_golden("losses_real")
    wl = [t(G[f"w{i}"]).clone().requires_grad_(True) for i in range(3)]
    bl = [t(G[f"sbins{i}"]) for i in range(3)]
    il = O.interlevel_loss_zaa(wl, bl, (0.03, 0.003))
    close(il, G["interlevel"], rtol=1e-4)
    g = torch.autograd.grad(il, wl[:2], retain_graph=True)
    for got, key in zip(g, ("g_interlevel_w0", "g_interlevel_w1")):
        ref = t(G[key])
        assert float((got - ref).abs().max()) <= 2e-4 * float(ref.abs().max()), key
    dl = O.distortion_loss(bl[2], wl[2])
    close(dl, G["distortion"], rtol=1e-4)
    close(torch.autograd.grad(dl, wl[2])[0], G["g_distortion_w2"], rtol=1e-4, atol=1e-8)


def test_extraction_frame_loop():
    """the body of the reference's frame loop (ns/scripts/extract_priors.py:108-145) run on the fixture model
    (tests/golden/extract.npz): raw depths, the depth / height selection, world points, densities, fp16 features, colours"""
    from conftest import model_fixture_setup

    G = load_golden("extract")
    cfg, scene, P, _ = model_fixture_setup(load_golden("model"))
    for k in range(cfg["num_fields"]):  # the extraction fixture's density heads (make_golden.gold_extract)
        P[f"field.fields.{k}.mlp_base_mlp.layers.1.bias"][0] = 1.0
        for i in range(2):
            P[f"proposal_networks.{i}.fields.{k}.mlp_base.1.layers.1.bias"][0] = 1.0
    scene["dino_to_rgb"] = O.make_scene(cfg)["dino_to_rgb"]
    n_pts = 0
    for depth_type in ("depth", "expected_depth"):
        for cam in G["frames"].tolist():
            tag = f"{depth_type}_{cam}"
            r = O.extract_frame(P, cfg, scene, cam, float(G["scaling"]), float(G["pose_scale_factor"]), float(G["max_depth"]),
                                float(G["min_depth"]), depth_type)
            assert r["raw_depth"].shape[0] == int(G["H"]) * int(G["W"])
            close(r["raw_depth"], G["raw_depth_" + tag], rtol=2e-5, atol=1e-5)
            assert torch.equal(r["sel"], t(G["sel_" + tag]))
            if int(G["sel_" + tag].sum()) == 0:
                continue
            close(r["world"], G["world_" + tag], rtol=2e-5, atol=2e-5)
            close(r["dens"], G["dens_" + tag], rtol=1e-4)
            assert float((r["feats"].float() - t(G["feats_" + tag]).float()).abs().max()) <= 2 ** -10
            assert float((r["colors"].float() - t(G["colors_" + tag]).float()).abs().max()) <= 4e-3
            n_pts += r["world"].shape[0]
    assert n_pts > 500


def test_voxel_downsample_rule():
    """per-voxel traces of extract_priors.py:166-191 on a hand-made case: two points share a voxel, two sit alone (the point
    that defines min_bound lies on a voxel boundary in exact arithmetic, so it is kept away from the others)"""
    pts = torch.tensor([[0.3, 0.3, 0.3], [0.35, 0.32, 0.28], [1.5, 0.3, 0.3], [-0.43, -0.43, -0.43]])
    feats = torch.tensor([[0.25, 1.0], [0.75, 0.0], [0.5, 0.5], [0.125, 0.125]], dtype=torch.float16)
    cols = torch.tensor([[0.0, 0.2, 0.4], [1.0, 0.4, 0.0], [0.3, 0.3, 0.3], [0.9, 0.9, 0.9]])
    vox = O.voxel_downsample(pts, feats, cols, voxel=0.4)
    assert len(vox) == 3
    mn = pts.min(0).values - 1.0
    k01 = tuple(O.voxel_index(pts[:1], 0.4, mn)[0].tolist())
    assert k01 == (4, 4, 4) and tuple(O.voxel_index(pts[1:2], 0.4, mn)[0].tolist()) == k01
    p, f, c, h = vox[k01]
    assert h == 2 and np.allclose(p, [0.325, 0.31, 0.29]) and np.array_equal(f, np.array([0.5, 0.5], dtype=np.float16))
    assert np.allclose(c, [0.5, 0.3, 0.2])
    assert vox[(7, 4, 4)][3] == 1
