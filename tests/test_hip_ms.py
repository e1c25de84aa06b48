"""GPU parity of the multi-sub-field path (all K sub-fields of a tile in one launch per kernel, csrc/ms_core.hpp) against
(a) the reference router's definition — cdist().argmin() + per-sub-field masks, ns/fields/PreSight/ingp_field_ms.py:97-126 —
through the CPU oracle, at the production configuration's shape (K = 16, L = 10, F = 4, max_res 16384) with small tables, and
(b) the per-sub-field kernels it replaces (same arithmetic per point: outputs bit-identical)."""
import numpy as np
import pytest
import torch

from conftest import assert_grads_within_oracle_noise, grad_error_stats, to_double
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _scene(K, seed=0, spread=0.6):
    g = torch.Generator().manual_seed(seed)
    cent = (torch.rand(K, 3, generator=g) - 0.5) * 2 * spread
    half = 0.25 + 0.2 * torch.rand(K, 1, generator=g)
    aabbs = torch.stack([cent - half, cent + half], 1)
    return cent, aabbs


def test_route_is_a_stable_sort_into_padded_chunks(dev):
    """perm lists, group after group, the points of every sub-field in increasing order; groups start on chunk boundaries;
    padding is -1; chunk_field / field_start describe exactly that — for point sets and for ray samples, ragged sizes"""
    from presight_amd import field_ops as F
    from presight_amd._lib import lib

    CH = lib().ps_ms_chunk()
    for K, N, seed in ((3, 1000, 0), (16, 70001, 1), (16, 5, 2), (7, 4096 * 3, 3), (64, 30000, 4), (2, 0, 5)):
        cent, _ = _scene(K, seed)
        g = torch.Generator().manual_seed(seed)
        pos = (torch.rand(N, 3, generator=g) - 0.5) * 2
        lay = F.MsLayout(cent.to(dev), pos=pos.to(dev))
        assign = O.route(pos, cent)
        perm = lay.perm.cpu()
        plan = lay.plan.cpu()
        off_fs = (lay.field_start - lay.plan.data_ptr()) // 4
        off_cf = (lay.chunk_field - lay.plan.data_ptr()) // 4
        fs = plan[off_fs:off_fs + K + 1].tolist()
        cf = plan[off_cf:off_cf + lay.chunks].tolist()
        assert fs[0] == 0 and len(perm) == lay.n_slots == lay.chunks * CH
        for k in range(K):
            idx = torch.nonzero(assign == k).flatten()
            n_chunks = (len(idx) + CH - 1) // CH
            assert fs[k + 1] - fs[k] == n_chunks, (K, N, k)
            seg = perm[fs[k] * CH: fs[k + 1] * CH]
            assert torch.equal(seg[:len(idx)].long(), idx), (K, N, k)  # stable: increasing point index
            assert bool((seg[len(idx):] == -1).all())
            assert cf[fs[k]:fs[k + 1]] == [k] * n_chunks
        assert all(c == -1 for c in cf[fs[K]:]) and bool((perm[fs[K] * CH:] == -1).all())
    # ray samples: positions are generated inside the router
    R, S = 300, 48
    cent, _ = _scene(5, 9)
    g = torch.Generator().manual_seed(9)
    o, d = (torch.rand(R, 3, generator=g) - 0.5), torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    eb = torch.sort(torch.rand(R, S + 1, generator=g) * 1.5, dim=1).values
    lay = F.MsLayout(cent.to(dev), origins=o.to(dev), dirs=d.to(dev), ebins=eb.to(dev))
    pos = o[:, None, :] + d[:, None, :] * ((eb[:, :-1] + eb[:, 1:]) / 2)[..., None]
    assign = O.route(pos.reshape(-1, 3), cent)
    perm = lay.perm.cpu()
    used = perm[perm >= 0].long()
    assert torch.equal(torch.sort(used).values, torch.arange(R * S))
    assert bool((assign[used][1:] >= assign[used][:-1]).all())  # grouped by sub-field


def _prod_cfg(K, log2T=12):
    cfg = O.default_config()
    cfg["num_fields"] = K
    cfg["main"] = dict(num_levels=10, features_per_level=4, log2_hashmap_size=log2T, base_res=16, max_res=16384, hidden_dim=64,
                       hidden_dim_color=64, geo_feat_dim=15, semantic_dim=64)
    cfg["props"] = [dict(num_levels=8, features_per_level=1, log2_hashmap_size=log2T, base_res=16, max_res=1024, hidden_dim=64),
                    dict(num_levels=8, features_per_level=1, log2_hashmap_size=log2T, base_res=16, max_res=4096, hidden_dim=64)]
    cfg["num_cameras"], cfg["num_videos"] = 24, 2
    return cfg


def _build_model(cfg, scene, P, dev):
    from presight_amd.model import NerfactoNuscMSModel, NerfactoNuscMSModelConfig

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
        assert k in sd, k
        sd[k] = v
        for alias in (k.replace("mlp_base_grid.", "mlp_base.0.").replace("mlp_base_mlp.", "mlp_base.1."),
                      k.replace("encoding.hash_table", "mlp_base.0.hash_table")):
            if alias in sd:
                sd[alias] = v
    model.load_state_dict(sd)
    return model.to(dev)


def _bundle(scene, batch, dev):
    from presight_amd import ops
    from presight_amd.rays import RayBundle

    ri = batch["ray_indices"].to(dev)
    o, d, pa, dn = ops.generate_rays(ri, *(scene[k].to(dev) for k in ("c2w", "fx", "fy", "cx", "cy")))
    return RayBundle(o, d, pa, camera_indices=ri[:, 0:1], metadata={"video_id": batch["video_ids"].to(dev)[:, None], "directions_norm": dn})


def _scaled_err(got, ref):
    scale = float(ref.abs().max())
    if scale == 0:
        return float(got.abs().max())
    return float(((got - ref).abs() / scale).max())


def test_production_shape_k16_training_step_matches_oracle(dev):
    """K = 16 sub-fields, L10 F4 main grids up to resolution 16384, proposal grids L8 F1: one full training step (forward,
    5 losses, backward) through the single-launch multi-sub-field kernels vs the CPU oracle — outputs, losses and the
    gradient of EVERY parameter of every sub-field (sub-fields that received no sample: exactly zero)."""
    K = 16
    cfg = _prod_cfg(K)
    scene = O.make_scene(cfg)
    assert scene["centroids"].shape[0] == K
    P = O.make_params(cfg, seed=11, table_scale=0.3)
    for k in range(K):
        P[f"field.fields.{k}.mlp_base_mlp.layers.1.bias"][0] = -2.0
        for i in range(2):
            P[f"proposal_networks.{i}.fields.{k}.mlp_base.1.layers.1.bias"][0] = -2.0
    batch = O.make_batch(cfg, scene, 192, step=0)
    model = _build_model(cfg, scene, P, dev)
    model.train()
    out = model(_bundle(scene, batch, dev), jitters=[j.to(dev) for j in batch["jitter"]])
    gt = {k: batch[k].to(dev) for k in ("rgb", "features", "sky")}
    losses = model.get_loss_dict(out, gt)
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    out_ref = O.model_forward(Pg, cfg, scene, batch, training=True)
    L_ref = O.loss_dict(out_ref, batch, cfg)
    for k in ("rgb", "semantics", "accumulation", "expected_depth"):
        torch.testing.assert_close(out[k].detach().cpu().reshape(out_ref[k].shape), out_ref[k].detach(), rtol=2e-4, atol=2e-5, msg=lambda m: f"{k}: {m}")
    # fp64 run of the same oracle: its distance from the fp32 run is the rounding noise of the COMPUTATION (not of the kernels) and
    # sets every bound below that is not a plain fp32 tolerance
    P64 = {k: v.detach().double().requires_grad_(True) for k, v in P.items()}
    out64 = O.model_forward(P64, cfg, to_double(scene), to_double(batch), training=True)
    L64 = O.loss_dict(out64, to_double(batch), cfg)
    for k in ("rgb_loss", "semantic_loss", "distortion_loss", "sky_loss", "interlevel_loss"):
        noise = abs(float(L_ref[k]) - float(L64[k]))
        torch.testing.assert_close(losses[k].detach().cpu(), L_ref[k].detach(), rtol=5e-4, atol=1e-8 + 4 * noise, msg=f"{k} (oracle fp32-fp64 {noise:.1e})")
    sum(losses.values()).backward()
    g_ref = torch.autograd.grad(sum(L_ref.values()), list(Pg.values()), allow_unused=True)
    g_ref = {n: (g if g is not None else torch.zeros_like(p.detach())) for (n, p), g in zip(Pg.items(), g_ref)}
    g64 = torch.autograd.grad(sum(L64.values()), list(P64.values()), allow_unused=True)
    g64 = {n: (g if g is not None else torch.zeros_like(p.detach())) for (n, p), g in zip(P64.items(), g64)}
    # per-tensor bound = max(5e-5, 4 x |oracle fp32 - oracle fp64|): a few gradients of sub-fields that see a handful of
    # near-saturated rays are ill-conditioned in fp32 on the reference side too (4e-3..7e-3 at this seed), the bulk is ~1e-5
    errs, names, bounds = assert_grads_within_oracle_noise({n: p.grad for n, p in model.named_parameters()}, g_ref, g64, what="K=16 step",
                                                           cap=1.2e-2)  # (2 x the worst error observed on any tensor of a network whose computed bound exceeds it: 5.2e-3, r06)
    n_zero = len(g_ref) - len(errs)
    q = lambda f: errs[min(len(errs) - 1, int(f * len(errs)))]  # noqa: E731
    print(f"K=16 step: {len(errs)} parameter gradients compared, {n_zero} exactly zero on both sides (sub-fields without samples); "
          f"scaled error median {q(0.5):.1e}, 90% {q(0.9):.1e}, max {errs[-1]:.1e} ({names[-1]}, bound {bounds[-1]:.1e})")
    assert len(errs) > 100 and q(0.5) < 1e-4


def test_k8_training_step_matches_reference_fixture(dev, gold_model_k8):
    """tests/golden/model_k8.npz = the REFERENCE's own training step with K = 8 routed sub-fields at the production shape (L10 F4 up
    to resolution 16384, L8 F1 proposal grids, 64-wide MLPs): outputs, the five losses and every gradient the reference produced
    (parameters of sub-fields it never called carry None there and exactly zero here).  Gradient bounds per network = max(5e-5,
    4 x the fp32 reference's distance from the fp64 oracle run)."""
    from conftest import assert_threshold_depth, model_k8_setup, t

    G = gold_model_k8
    cfg, scene, P, batch = model_k8_setup(G)
    model = _build_model(cfg, scene, P, dev)
    model.train()
    model.proposal_sampler.set_anneal(float(G["T_anneal"]))
    out = model(_bundle(scene, batch, dev), jitters=[j.to(dev) for j in batch["jitter"]])
    for i in range(3):
        torch.testing.assert_close(out["weights_list"][i][..., 0].detach().cpu(), t(G[f"T_weights_{i}"]), rtol=2e-4, atol=2e-6)
    for k in ("rgb", "accumulation", "expected_depth", "semantics"):
        torch.testing.assert_close(out[k].detach().cpu(), t(G["T_" + k]), rtol=2e-4, atol=2e-5, msg=k)
    for k, lvl in (("depth", 2), ("prop_depth_0", 0), ("prop_depth_1", 1)):
        assert_threshold_depth(out[k], G["T_" + k], G[f"T_weights_{lvl}"], out["ray_samples_list"][lvl].ebins, what=k)
    gt = {k: batch[k].to(dev) for k in ("rgb", "features", "sky")}
    losses = model.get_loss_dict(out, gt)
    anneal = float(G["T_anneal"])
    P64 = {k: v.detach().double().requires_grad_(True) for k, v in P.items()}
    L64 = O.loss_dict(O.model_forward(P64, cfg, to_double(scene), to_double(batch), training=True, anneal=anneal), to_double(batch), cfg)
    for k, v in losses.items():
        noise = abs(float(G["TL_" + k]) - float(L64[k]))
        torch.testing.assert_close(v.detach().cpu().reshape(()), t(G["TL_" + k]).reshape(()), rtol=5e-4, atol=1e-8 + 4 * noise,
                                   msg=f"{k}: {float(v):.8g} vs reference {float(G['TL_' + k]):.8g}, oracle fp64 {float(L64[k]):.8g}")
    sum(losses.values()).backward()
    g64 = torch.autograd.grad(sum(L64.values()), list(P64.values()), allow_unused=True)
    g64 = {n: (g if g is not None else torch.zeros_like(p.detach())) for (n, p), g in zip(P64.items(), g64)}
    g_ref = {n: (t(G["TG_" + n]) if "TG_" + n in G else torch.zeros_like(v)) for n, v in P.items()}
    # cap 5e-2 instead of the default 1e-2, on purpose and printed by the helper: at the production shape (levels up to resolution
    # 16384, 64 rays) the REFERENCE's own fp32 gradients of these tables sit 0.2 - 0.7 of their largest entry away from the fp64
    # gradient in the max-norm (single hash rows behind flipped ReLU units); the per-tensor 2-norm guard -- as close to the exact
    # gradient as the reference's own fp32 run -- is the tight check for those networks
    errs, names, bounds = assert_grads_within_oracle_noise({n: p.grad for n, p in model.named_parameters()}, g_ref, g64, what="K=8 reference step",
                                                           cap=5e-2)
    print(f"K=8 step vs the reference fixture: {len(errs)} gradients, median {errs[len(errs) // 2]:.1e}, max {errs[-1]:.1e} ({names[-1]}, bound {bounds[-1]:.1e})")
    assert len(errs) == int(G["n_grads"]) and errs[len(errs) // 2] < 5e-5


@pytest.mark.parametrize("backward,merged", [("three kernels", False), ("fused", False), ("three kernels", True)])
def test_ms_fields_equal_the_per_field_kernels(dev, backward, merged, monkeypatch):
    """Same arithmetic per point in both paths: densities / colours / semantics are bit-identical; table gradients too (int64
    fixed-point accumulation with per-(sub-field, level) scales); MLP weight gradients agree to summation order.  Both forms of
    the main backward (one kernel per MLP stack = the training path, and the single fused kernel).
    merged: the routed kernels run the MERGED network (base output rows 16..79 folded into the semantic head's first layer per
    sub-field, the product path) against the unmerged per-field kernels: densities and colours stay bit-identical (their arithmetic
    is untouched), semantics and every gradient agree to fp32 re-association."""
    monkeypatch.setenv("PRESIGHT_MAIN_BWD_SPLIT", "1" if backward == "three kernels" else "0")
    from presight_amd import field_ops as F

    monkeypatch.setattr(F, "MERGED_MS", merged)
    from presight_amd.fields import iNGPField, PropNetDensityField, iNGPFieldMS, PropNetDensityFieldMS, routed_apply
    from presight_amd.components import SceneContraction

    torch.manual_seed(0)
    K = 5
    cent, aabbs = _scene(K, 3)
    contraction = SceneContraction(order=float("inf"))
    mains = [iNGPField(aabbs[k], num_levels=10, features_per_level=4, log2_hashmap_size=11, max_res=16384, use_semantics=True,
                       appearance_embedding_dim=16, spatial_distortion=contraction, implementation="hip") for k in range(K)]
    props = [PropNetDensityField(aabbs[k], num_levels=8, features_per_level=1, log2_hashmap_size=11, max_res=1024,
                                 spatial_distortion=contraction, implementation="hip") for k in range(K)]
    for f in mains + props:
        for n, p in f.named_parameters():
            if "hash_table" in n:
                p.data.mul_(300.0)
    ms_main, ms_prop = iNGPFieldMS(mains, cent).to(dev), PropNetDensityFieldMS(props, cent).to(dev)
    R, S = 257, 24
    g = torch.Generator().manual_seed(1)
    o = ((torch.rand(R, 3, generator=g) - 0.5) * 0.8).to(dev)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
    eb = torch.sort(torch.rand(R, S + 1, generator=g) * 1.2, dim=1).values.to(dev)
    app = torch.randn(R, 16, generator=g).to(dev).requires_grad_(True)
    pos = (o[:, None, :] + d[:, None, :] * ((eb[:, :-1] + eb[:, 1:]) / 2)[..., None]).reshape(-1, 3)
    # per-field path (round-1 router: sort, slice per sub-field, un-sort) on the same modules
    per_point = lambda t: t[:, None, :].expand(R, S, t.shape[-1]).reshape(R * S, t.shape[-1])  # noqa: E731

    def run_main(k, p, dd, aa):
        u, sel = mains[k].points(pos=p)
        return mains[k].evaluate(u, sel, dd, aa, 1)

    def grads(mods):
        out = {n: (p.grad.clone() if p.grad is not None else None) for m in mods for n, p in m.named_parameters()}
        for m in mods:
            m.zero_grad(set_to_none=True)
        return out

    ws, wc, wm = torch.randn(R * S, generator=g).to(dev), torch.randn(R * S, 3, generator=g).to(dev), torch.randn(R * S, 64, generator=g).to(dev)
    s0, c0, m0 = routed_apply(pos, cent.to(dev), run_main, [per_point(d), per_point(app)])
    ((s0.view(-1) * ws).sum() + (c0 * wc).sum() + (m0 * wm).sum()).backward()
    g_ref, dapp_ref = grads([ms_main]), app.grad.clone()
    app.grad = None
    lay = F.MsLayout(cent.to(dev), origins=o, dirs=d, ebins=eb)
    mm = ms_main._ms()
    u, sel = lay.points(mm["aabbs"], mm["contract"])
    s1, c1, m1 = F.ms_main_field(lay, u, sel, d, app, S, mm["tables"], mm["scalings"], mm["g"], mm["base"], mm["sem"], mm["rgb"])
    assert torch.equal(s1, s0.view(-1)) and torch.equal(c1, c0)
    if merged:
        assert _scaled_err(m1, m0) < 2e-6 and not torch.equal(m1, m0)  # (the merged path really ran)
    else:
        assert torch.equal(m1, m0)
    ((s1 * ws).sum() + (c1 * wc).sum() + (m1 * wm).sum()).backward()
    g_ms = grads([ms_main])
    torch.testing.assert_close(app.grad, dapp_ref, rtol=1e-4, atol=1e-5)
    for n, a in g_ref.items():
        b = g_ms[n]
        if a is None:
            assert b is None or float(b.abs().max()) == 0, n
        elif "hash_table" in n and not merged:
            assert torch.equal(a, b), n
        else:
            assert _scaled_err(b, a) < 1e-5, (n, _scaled_err(b, a))
    # proposal fields
    p0 = routed_apply(pos, cent.to(dev), lambda k, p: (props[k].density_fn(p),))[0].view(-1)
    (p0 * ws).sum().backward()
    gp_ref = grads([ms_prop])
    from presight_amd.rays import RayBundle, RaySamples

    rs = RaySamples(RayBundle(o, d, torch.ones(R, 1, device=dev)), eb, eb, spacing_to_euclidean_fn=(0.0, 1.0, 1.0))
    p1 = ms_prop.density_of_samples(rs).view(-1)
    assert torch.equal(p1, p0)
    (p1 * ws).sum().backward()
    gp_ms = grads([ms_prop])
    for n, a in gp_ref.items():
        b = gp_ms[n]
        if a is None:
            assert b is None or float(b.abs().max()) == 0, n
        elif "hash_table" in n:
            assert torch.equal(a, b), n
        else:
            assert _scaled_err(b, a) < 1e-5, (n, _scaled_err(b, a))
    # no-grad queries (prior extraction): density / semantics by position
    with torch.no_grad():
        dd, ss = ms_main.density_and_semantics(pos)
        assert torch.equal(dd.view(-1), s0.view(-1).detach())
        assert (_scaled_err(ss, m0.detach()) < 2e-6) if merged else torch.equal(ss, m0.detach())
        assert torch.equal(ms_prop.density_fn(pos).view(-1), p0.detach())


def test_full_size_k16_step_properties(dev):
    """BASELINE cfg 3 shape at full size on one rank (K = 16, T = 2^20 tables, 8192 rays): the routed step runs without a host
    synchronisation inside the fields, every sample lands in exactly one sub-field (densities finite, weights sum <= 1),
    table gradients are bit-reproducible run to run, and parameters of sub-fields without samples receive no gradient."""
    import bench

    model, scene = bench.build_model(dev, seed=1, config="cfg3")
    tr = bench.Trainer(model, scene, 1)
    batch = bench.make_batches(scene, dev, 1, 0, rays=8192)[0]
    flats = []
    for rep in range(2):
        tr.grads.zero_()
        torch.manual_seed(5)
        loss_dict, out = None, None
        m = model
        m.train()
        from presight_amd import ops
        from presight_amd.rays import RayBundle

        o, d, pa, dn = ops.generate_rays(batch["ray_indices"], scene["c2w"], scene["fx"], scene["fy"], scene["cx"], scene["cy"])
        rb = RayBundle(o, d, pa, camera_indices=batch["ray_indices"][:, 0:1], metadata={"video_id": batch["video_ids"][:, None], "directions_norm": dn})
        out = m(rb)
        ld = m.get_loss_dict(out, batch)
        sum(ld.values()).backward()
        flats.append(tr.grads.flat.clone())
        w = out["weights_list"][-1]
        assert bool(torch.isfinite(out["rgb"]).all()) and float(w.sum(1).max()) <= 1.0 + 1e-4
    names = {id(p): n for n, p in model.named_parameters()}
    tables = [(tr.grads.offsets[i], p.numel()) for i, p in enumerate(tr.grads.params) if "hash_table" in names[id(p)]]
    assert len(tables) == 48
    for off, n in tables:
        assert torch.equal(flats[0][off:off + n], flats[1][off:off + n])
    touched = tr.grads.touched()
    for p, t_ in zip(tr.grads.params, touched):
        if not t_:
            assert float(p.grad.abs().max()) == 0.0
    # Sub-fields without samples: the router decides on the device which of the K sub-fields received points; their group flag
    # stays down and the optimizer kernel leaves parameters, both moments and the step count BIT-identical (the reference never
    # calls an empty sub-field, ingp_field_ms.py:97-126, so torch.optim.Adam sees grad None).  Rays of ONE camera only reach a
    # few of the 16 sub-fields.
    one_cam = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    one_cam["ray_indices"][:, 0] = 0
    one_cam["video_ids"][:] = 0
    for _ in range(2):
        tr.step(batch)  # every group that can be reached gets moments first
    before = [(p.detach().clone(), m.clone(), v.clone()) for p, m, v in zip(tr.opt.params, tr.opt.exp_avg, tr.opt.exp_avg_sq)]
    steps_before = tr.opt.param_steps()
    tr.step(one_cam)
    flags = tr.grads.group_flags.tolist()
    steps_after = tr.opt.param_steps()
    assert tr.grads.n_groups == 16 * 4 and 0 < sum(flags) < len(flags)
    n_skipped = 0
    for i, p in enumerate(tr.opt.params):
        gid = getattr(p, "_ps_group", None)
        if gid is not None and flags[gid] == 0:
            n_skipped += 1
            assert steps_after[i] == steps_before[i]
            assert torch.equal(p.detach(), before[i][0]) and torch.equal(tr.opt.exp_avg[i], before[i][1]) and torch.equal(tr.opt.exp_avg_sq[i], before[i][2])
        elif gid is not None:
            assert steps_after[i] == steps_before[i] + 1 and not torch.equal(p.detach(), before[i][0])
    assert n_skipped > 0


def test_fused_sky_field_equals_operator_level_path(dev):
    """sky colour / semantic heads: the fused per-ray kernel (single field and routed K = 5) against SH encoding + concat + the
    operator-level MLP kernels (golden-tested in test_hip_ops.py), forward and every gradient incl. d(appearance)"""
    from presight_amd.fields import FieldHeadNames, SkyField, SkyFieldMS
    from presight_amd.rays import RayBundle, RaySamples

    torch.manual_seed(3)
    R = 3000
    g = torch.Generator().manual_seed(4)
    o = ((torch.rand(R, 3, generator=g) - 0.5) * 1.5).to(dev)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
    wr, wsm = torch.randn(R, 3, generator=g).to(dev), torch.randn(R, 64, generator=g).to(dev)
    for K in (1, 5):
        cent, _ = _scene(K, 7)
        fields = [SkyField(mlp_num_layers=3, mlp_layer_width=32, appearance_embedding_dim=16, use_semantics=True, semantic_dim=64,
                           implementation="hip") for _ in range(K)]
        sky = SkyFieldMS(fields, cent).to(dev)
        app = torch.randn(R, 16, generator=g).to(dev).requires_grad_(True)
        rs = RaySamples(RayBundle(o, d, torch.ones(R, 1, device=dev)), torch.zeros(R, 2, device=dev), torch.zeros(R, 2, device=dev))

        def run(fused):
            saved = SkyField._fused
            if not fused:
                SkyField._fused = lambda self, a: False
            try:
                out = sky(rs, app[:, None, :])
            finally:
                SkyField._fused = saved
            ((out[FieldHeadNames.RGB] * wr).sum() + (out[FieldHeadNames.SEMANTICS] * wsm).sum()).backward()
            grads = {n: p.grad.clone() for n, p in sky.named_parameters()}
            da = app.grad.clone()
            sky.zero_grad(set_to_none=True)
            app.grad = None
            return out, grads, da

        out_f, g_f, da_f = run(True)
        out_o, g_o, da_o = run(False)
        torch.testing.assert_close(out_f[FieldHeadNames.RGB], out_o[FieldHeadNames.RGB], rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(out_f[FieldHeadNames.SEMANTICS], out_o[FieldHeadNames.SEMANTICS], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(da_f, da_o, rtol=1e-4, atol=1e-6)
        assert set(g_f) == set(g_o)
        for n in g_o:
            assert _scaled_err(g_f[n], g_o[n]) < 2e-5, (K, n, _scaled_err(g_f[n], g_o[n]))
