"""Whole-model GPU parity: NerfactoNuscMSModel on the HIP kernels vs the fixture produced by running the reference's
own NerfactoNuscMSModel (3 sub-fields, full training step with all five losses, eval render, depth march and the
prior-extraction queries)."""
import numpy as np
import pytest
import torch

from conftest import assert_threshold_depth, grad_error_stats, model_fixture_setup, t

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().cpu()
    b = t(b) if isinstance(b, np.ndarray) else b.detach().cpu()
    torch.testing.assert_close(a.to(b.dtype).reshape(b.shape), b, rtol=rtol, atol=atol)


def build(G, dev):
    from presight_amd import ops
    from presight_amd.model import NerfactoNuscMSModel, NerfactoNuscMSModelConfig
    from presight_amd.rays import RayBundle

    cfg, scene, P, batch = model_fixture_setup(G)
    m = cfg["main"]
    conf = NerfactoNuscMSModelConfig(
        near_plane=cfg["near"], far_plane=cfg["far"], piecewise_sampler_threshold=cfg["thr"], hidden_dim=m["hidden_dim"],
        hidden_dim_color=m["hidden_dim_color"], num_levels=m["num_levels"], base_res=m["base_res"], max_res=m["max_res"],
        log2_hashmap_size=m["log2_hashmap_size"], features_per_level=m["features_per_level"],
        proposal_net_args_list=[dict(features_per_level=p["features_per_level"], log2_hashmap_size=p["log2_hashmap_size"],
                                     num_levels=p["num_levels"], base_res=p["base_res"], max_res=p["max_res"],
                                     hidden_dim=p["hidden_dim"], use_linear=False) for p in cfg["props"]],
        implementation="hip", use_lidar_loss=False, distortion_loss_mult=cfg["distortion_loss_mult"],
        sky_mlp_dims=cfg["sky"]["width"], num_sky_mlp_layers=cfg["sky"]["num_layers"])
    model = NerfactoNuscMSModel(conf, num_train_cameras=cfg["num_cameras"], num_train_videos=cfg["num_videos"],
                                dino_to_rgb=scene["dino_to_rgb"], centroids=scene["centroids"], aabbs=scene["aabbs"])
    sd = model.state_dict()
    missing = [k for k in P if k not in sd]
    assert not missing, missing  # the reference's (torch-implementation) checkpoint keys all exist
    full = dict(sd)
    for k, v in P.items():
        full[k] = v
        alias = k.replace("mlp_base_grid.", "mlp_base.0.").replace("mlp_base_mlp.", "mlp_base.1.").replace("encoding.hash_table", "mlp_base.0.hash_table")
        if alias in full:
            full[alias] = v
    model.load_state_dict(full)
    model.to(dev)

    def bundle():
        ri = batch["ray_indices"].to(dev)
        o, d, pa, dn = ops.generate_rays(ri, scene["c2w"].to(dev), scene["fx"].to(dev), scene["fy"].to(dev), scene["cx"].to(dev),
                                         scene["cy"].to(dev))
        return RayBundle(o, d, pa, camera_indices=ri[:, 0:1], metadata={"video_id": batch["video_ids"].to(dev)[:, None],
                                                                         "directions_norm": dn})

    return model, cfg, scene, P, batch, bundle


def test_training_step_matches_reference(gold_model):
    G = gold_model
    dev = torch.device("cuda:0")
    model, cfg, scene, P, batch, bundle = build(G, dev)
    model.train()
    model.proposal_sampler.set_anneal(float(G["T_anneal"]))
    out = model(bundle(), jitters=[j.to(dev) for j in batch["jitter"]])
    for i in range(3):
        close(out["ray_samples_list"][i].sbins, G[f"T_sbins_{i}"], atol=4e-6)
        close(out["weights_list"][i][..., 0], G[f"T_weights_{i}"], rtol=2e-4, atol=2e-6)
    for k in ["rgb", "accumulation", "expected_depth", "semantics"]:
        close(out[k], G["T_" + k], rtol=2e-4, atol=2e-5)
    for k, lvl in (("depth", 2), ("prop_depth_0", 0), ("prop_depth_1", 1)):
        assert_threshold_depth(out[k], G["T_" + k], G[f"T_weights_{lvl}"], out["ray_samples_list"][lvl].ebins, what=k)
    gt = {"rgb": batch["rgb"].to(dev), "features": batch["features"].to(dev), "sky": batch["sky"].to(dev)}
    ld = model.get_loss_dict(out, gt)
    # The interlevel loss divides fp32 cumsums by very narrow PDF-resampled bins, which amplifies summation-order differences
    # (tests/test_hip_losses.py).  Its tolerance is COMPUTED: the distance between the fp32 and the fp64 run of the pinned oracle
    # on the fixture's inputs = the rounding noise any fp32 evaluation of this loss carries, the reference's own included.
    from conftest import to_double
    from oracle import nerf_oracle as O

    with torch.no_grad():
        anneal = float(G["T_anneal"])
        l32 = O.loss_dict(O.model_forward(P, cfg, scene, batch, training=True, anneal=anneal), batch, cfg)
        l64 = O.loss_dict(O.model_forward(to_double(P), cfg, to_double(scene), to_double(batch), training=True, anneal=anneal), to_double(batch), cfg)
    for k, v in ld.items():
        noise = abs(float(l32[k]) - float(l64[k]))
        assert abs(float(l32[k]) - float(G["TL_" + k])) <= 5e-4 * abs(float(G["TL_" + k])) + 4 * noise + 1e-7, k  # the oracle restates the fixture
        close(v, G["TL_" + k], rtol=5e-4, atol=1e-7 + 4 * noise)
    sum(ld.values()).backward()
    g_ref = {k[3:]: t(G[k]) for k in G if k.startswith("TG_")}
    assert len(g_ref) == len(P)
    errs, names, n_zero = grad_error_stats({n: p.grad for n, p in model.named_parameters()}, g_ref)
    q = lambda f: float(errs[min(len(errs) - 1, int(f * len(errs)))])  # noqa: E731
    print(f"training step vs reference: {len(errs)} parameter gradients (+{n_zero} exactly zero); max |err| / max |ref|: "
          f"median {q(0.5):.1e}, 90% {q(0.9):.1e}, max {float(errs[-1]):.1e} ({names[-1]})")
    # achieved on MI355X (round 2): median 1.0e-6, 90 % 9e-6, worst tensor 1.2e-5; the bounds leave ~5x headroom over that
    assert q(0.5) < 1e-5 and q(0.9) < 5e-5 and float(errs[-1]) < 1e-4, (names[-3:], errs[-3:])
    psnr = float(model.get_metrics_dict(out, gt)["psnr"])
    ref_psnr = float(10 * torch.log10(1.0 / torch.mean((t(G["T_rgb"]) - batch["rgb"]) ** 2)))
    assert abs(psnr - ref_psnr) < 1e-3


def test_eval_render_depth_and_extraction_queries(gold_model):
    G = gold_model
    dev = torch.device("cuda:0")
    model, cfg, scene, P, batch, bundle = build(G, dev)
    model.eval()
    model.proposal_sampler.set_anneal(float(G["T_anneal"]))
    with torch.no_grad():
        out = model(bundle())
        for k in ["rgb", "accumulation", "expected_depth", "semantics", "dino_rgb"]:
            close(out[k], G["E_" + k], rtol=2e-4, atol=3e-5)
        # the fixture does not hold the eval weights: the oracle (bit-faithful to the reference here, test_oracle_golden.py) supplies them
        from oracle import nerf_oracle as O

        ref = O.model_forward(P, cfg, scene, batch, training=False, anneal=float(G["T_anneal"]))
        assert_threshold_depth(out["depth"], G["E_depth"], ref["weights_list"][-1], ref["euclid_list"][-1], what="eval depth")
        dd = model.get_depth_for_camera_ray_bundle(bundle())
        assert_threshold_depth(dd["depth"], G["E_get_depth"], ref["weights_list"][-1], ref["euclid_list"][-1], what="get_depth")
        close(dd["expected_depth"], G["E_get_expected_depth"], rtol=2e-4, atol=3e-5)
        # prior-extraction field queries (ns/scripts/extract_priors.py:133-138)
        pts = t(G["X_pts"]).to(dev)
        dens = [p.density_fn(pts).squeeze(-1) for p in model.proposal_networks]
        dens.append(model.field.density_fn(pts)[0].squeeze(-1))
        close(torch.stack(dens, 0).mean(0), G["X_density_mean"], rtol=1e-4, atol=1e-6)  # north_star: prior outputs within 1e-4 rel
        dens[-1] = model.field.density_only(pts).squeeze(-1)  # fused variant of the same query
        close(torch.stack(dens, 0).mean(0), G["X_density_mean"], rtol=1e-4, atol=1e-6)
        feats = model.field.semantic_fn(pts).clip(0.0, 1.0).to(torch.float16)
        assert (feats.float().cpu() - t(G["X_feats"]).float()).abs().max() <= 2 ** -10


def test_voxel_index_and_dense_lattice_query(gold_model):
    """BASELINE config 5 path: lattice points, fused field queries, bit-exact voxel index, voxel grouping."""
    from oracle import nerf_oracle as O
    from presight_amd import extract

    G = gold_model
    dev = torch.device("cuda:0")
    model, cfg, scene, P, batch, bundle = build(G, dev)
    model.eval()
    aabb = scene["aabbs"][1]
    res = 24
    pts = extract.lattice_points(aabb, res, 0, res ** 3, dev)
    ax = [aabb[0][k] + (aabb[1][k] - aabb[0][k]) * ((torch.arange(res, dtype=torch.float32) + 0.5) / res) for k in range(3)]
    ref_pts = torch.stack(torch.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)  # z fastest
    torch.testing.assert_close(pts.cpu(), ref_pts, rtol=0, atol=1e-6)
    dens, feats = extract.query_priors(model, pts)
    d_ref, f_ref = O.prior_query(P, cfg, scene, pts.cpu())
    close(dens, d_ref, rtol=1e-4, atol=1e-6)
    assert (feats.float().cpu() - f_ref.float()).abs().max() <= 2 ** -10
    # integer voxel index: bit exact against the oracle's fp64 rule, including negative coordinates
    world = (pts / 0.05).cpu()
    mn = world.min(0).values - 1.0
    idx = extract.voxel_index(world.to(dev), 0.4, mn)
    assert torch.equal(idx.cpu(), O.voxel_index(world, 0.4, mn))
    # dense tile query in two slabs == one pass (sharding the lattice needs no exchange)
    full = extract.dense_tile_query(model, aabb, res=res, chunk=5000, density_threshold=-1.0)
    half = res ** 3 // 2
    a = extract.dense_tile_query(model, aabb, res=res, chunk=5000, start=0, count=half, density_threshold=-1.0)
    b = extract.dense_tile_query(model, aabb, res=res, chunk=5000, start=half, count=res ** 3 - half, density_threshold=-1.0)
    assert torch.equal(torch.cat([a["points"], b["points"]]), full["points"])
    assert torch.equal(torch.cat([a["densities"], b["densities"]]), full["densities"])
    # voxel grouping: hits sum to n, per-voxel mean of points lies inside its voxel
    vox = extract.voxelize(full["points"], full["features"], None, voxel=0.4)
    assert int(vox["hits"].sum()) == full["points"].shape[0]
    lo = (vox["min_bound"].to(dev) - 0.2) + vox["index"].double() * 0.4
    assert bool(((vox["points"].double() >= lo - 1e-4) & (vox["points"].double() <= lo + 0.4 + 1e-4)).all())
    # wire format of extracted_priors.pkl (extract_priors.py:186-208): hit-count quantile filter, dtypes, keys, pickle round trip
    import pickle
    import tempfile

    pri = extract.finalize_priors(vox, origin=torch.tensor([1.0, 2.0, 3.0]), hit_thr_ratio=0.25)
    hits = vox["hits"].cpu().numpy()
    keep = hits > np.quantile(hits, 0.25)
    assert set(pri) == {"points", "features", "hits", "origin"} and pri["points"].dtype == np.float32
    assert pri["features"].dtype == np.float16 and pri["features"].shape == (int(keep.sum()), 64)
    assert np.array_equal(pri["hits"], hits[keep]) and pri["origin"].dtype == np.float32
    with tempfile.TemporaryDirectory() as td:
        extract.save_priors(td + "/extracted_priors.pkl", pri)
        back = pickle.load(open(td + "/extracted_priors.pkl", "rb"))
    assert all(np.array_equal(back[k], pri[k]) for k in pri)


def test_flat_gradients_and_fused_adam_match_autograd_path(gold_model):
    """Trainer plumbing (dist.FlatGrads + optim.HipAdam(flat_grads=...)): the backward kernels accumulate parameter gradients
    in place in one flat buffer and one Adam launch updates every parameter that received a gradient.  Must equal the
    plain autograd path (gradients returned as tensors) + torch.optim.Adam with the reference's settings, and must leave
    parameters without a gradient (sub-fields that saw no sample, proposal nets off-schedule) untouched like torch does."""
    from presight_amd.dist import FlatGrads
    from presight_amd.optim import HipAdam

    G = gold_model
    dev = torch.device("cuda:0")
    jit = None

    def run(model, batch, bundle):
        model.train()
        model.proposal_sampler.set_anneal(float(G["T_anneal"]))
        out = model(bundle(), jitters=[j.to(dev) for j in batch["jitter"]])
        gt = {"rgb": batch["rgb"].to(dev), "features": batch["features"].to(dev), "sky": batch["sky"].to(dev)}
        (sum(model.get_loss_dict(out, gt).values()) * 1024.0).backward()

    # A: plain autograd accumulation
    mA, _, _, P, batch, bundle = build(G, dev)
    run(mA, batch, bundle)
    gA = {k: (None if p.grad is None else p.grad.clone()) for k, p in mA.named_parameters()}
    # B: flat buffer, in-place accumulation
    mB, _, _, _, _, bundleB = build(G, dev)
    uniq, seen = [], set()
    for p in mB.parameters():
        if p.requires_grad and p.numel() > 0 and id(p) not in seen:
            seen.add(id(p))
            uniq.append(p)
    fg = FlatGrads(uniq)
    opt = HipAdam(uniq, lr=1e-2, eps=1e-15, weight_decay=1e-5, flat_grads=fg)
    before = {k: p.detach().clone() for k, p in mB.named_parameters()}
    fg.zero_()
    run(mB, batch, bundleB)
    fg.finish_exchange()  # (single process: only joins the proposal networks' side stream, as every owner of in-place gradients must)
    touched = dict(zip([id(p) for p in uniq], fg.touched()))
    n_untouched = 0
    for k, p in mB.named_parameters():
        if p.numel() == 0:
            continue
        assert p.grad.data_ptr() >= fg.flat.data_ptr() and p.grad.data_ptr() < fg.flat.data_ptr() + 4 * fg.total
        if gA[k] is None:
            assert not touched[id(p)] and float(p.grad.abs().max()) == 0.0, k
            n_untouched += 1
        else:
            assert touched[id(p)], k
            scale = float(gA[k].abs().max())
            torch.testing.assert_close(p.grad / max(scale, 1e-30), gA[k] / max(scale, 1e-30), rtol=1e-5, atol=2e-6, msg=lambda m: f"{k}: {m}")
    # optimizer: torch.optim.Adam on CPU copies of (param, grad) of the autograd path
    ref_p = {k: torch.nn.Parameter(before[k].cpu().clone()) for k in before if before[k].numel() > 0 and gA[k] is not None}
    for k, p in ref_p.items():
        p.grad = gA[k].cpu()
    torch.optim.Adam(list(ref_p.values()), lr=1e-2, eps=1e-15, weight_decay=1e-5).step()
    opt.step()
    for k, p in mB.named_parameters():
        if p.numel() == 0:
            continue
        if k in ref_p:
            # first Adam step moves every entry by ~lr*sign(g): compare the update, entries with |g| ~ 0 are ill-conditioned
            big = gA[k].abs().cpu() > 1e-6 * float(gA[k].abs().max())
            torch.testing.assert_close(p.detach().cpu()[big], ref_p[k].detach()[big], rtol=1e-4, atol=2e-5, msg=lambda m: f"{k}: {m}")
        else:
            assert torch.equal(p.detach(), before[k]), f"{k} has no gradient and must not move"
    assert len(fg.touched_ranges()) >= 1


@pytest.mark.parametrize("samples", [(128, 64, 64), (96, 40, 40)])
def test_fused_render_node_matches_separate_nodes(samples):
    """One sub-field: field + get_weights + renderers as one autograd node (field_ops.main_field_render, per-ray output
    gradients expanded inside the field backward) against the three separate nodes, outputs and every parameter gradient.
    40 samples per ray: 16-point blocks straddle rays (per-point d(appearance) atomics) and the composite kernels run with
    idle lanes."""
    import bench
    from presight_amd import ops
    from presight_amd.model import NerfactoNuscMSModel, NerfactoNuscMSModelConfig
    from presight_amd.rays import RayBundle

    dev = torch.device("cuda:0")
    scene = bench.make_scene(60, 6)
    conf = NerfactoNuscMSModelConfig(near_plane=0.005, far_plane=50.0, piecewise_sampler_threshold=5.0, num_levels=2,
                                     features_per_level=2, log2_hashmap_size=12, base_res=16, max_res=128, hidden_dim=32,
                                     hidden_dim_color=32, implementation="hip", use_lidar_loss=False,
                                     num_proposal_samples_per_ray=samples[:2], num_nerf_samples_per_ray=samples[2])
    torch.manual_seed(11)
    model = NerfactoNuscMSModel(conf, num_train_cameras=60, num_train_videos=6, dino_to_rgb=None, centroids=scene["centroids"],
                                aabbs=scene["aabbs"]).to(dev)
    with torch.no_grad():  # lift the densities so that the weights are not vanishingly small
        model.field.fields[0].mlp_base_mlp.layers[-1].bias[0] = 2.0
    scene = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
    R = 300
    batch = bench.make_batches(scene, dev, 1, 0, rays=R)[0]
    g = torch.Generator().manual_seed(2)
    jit = [torch.rand(R, 1, generator=g).to(dev) for _ in range(3)]
    from presight_amd import field_ops

    results, aux = {}, {}
    for fused in (False, True, "factored"):
        model.fused_render = bool(fused)
        field_ops.FACTORED = fused == "factored"
        model.zero_grad(set_to_none=True)
        model.train()
        o, d, pa, dn = ops.generate_rays(batch["ray_indices"], scene["c2w"], scene["fx"], scene["fy"], scene["cx"], scene["cy"])
        rb = RayBundle(o, d, pa, camera_indices=batch["ray_indices"][:, 0:1], metadata={"video_id": batch["video_ids"][:, None],
                                                                                       "directions_norm": dn})
        out = model(rb, jitters=jit)
        ld = model.get_loss_dict(out, batch)
        (sum(ld.values()) + out["expected_depth"].mean() * 0.1).backward()
        aux[fused] = (out["weights_list"][-1][..., 0].detach().clone(), out["ray_samples_list"][-1].ebins.detach().clone())
        results[fused] = ({k: out[k].detach().clone() for k in ("rgb", "accumulation", "depth", "expected_depth", "semantics")},
                          {k: v.detach().clone() for k, v in ld.items()},
                          {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    field_ops.FACTORED = True
    (oa, la, ga), (ob, lb, gb), (oc, lc, gc) = results[False], results[True], results["factored"]
    for k in oa:
        torch.testing.assert_close(ob[k], oa[k], rtol=0, atol=0, msg=lambda m: f"{k}: {m}")  # same forward kernels
    for k in la:
        torch.testing.assert_close(lb[k], la[k], rtol=0, atol=0)
    assert set(ga) == set(gb) == set(gc)
    for k in ga:
        s = float(ga[k].abs().max()) + 1e-30
        torch.testing.assert_close(gb[k] / s, ga[k] / s, rtol=1e-4, atol=1e-5, msg=lambda m: f"{k}: {m}")
    assert float(ga["field.fields.0.mlp_base_grid.hash_table"].abs().max()) > 0
    # The FACTORED node (the training default: base layer 1 rows 16..79 merged with semantic layer 0, semantic output layer applied per
    # ray after compositing, direction / appearance columns of the colour head's first layer evaluated per ray, rendering weights and
    # the semantic branch's compositing inside the field kernel) computes the same function with re-associated fp32 sums: every
    # output agrees to fp32 rounding (the weights' optical-depth prefix sums run in another order: a few ulp of the weights), the
    # threshold depth -- index work -- is the same sample except where the cumulative weight sits within rounding of the threshold
    from conftest import assert_threshold_depth

    assert_threshold_depth(oc["depth"], oa["depth"], *aux[False], what="factored depth")
    torch.testing.assert_close(aux["factored"][0], aux[False][0], rtol=2e-6, atol=1e-7)  # the rendering weights themselves
    for k in ("accumulation", "expected_depth"):
        torch.testing.assert_close(oc[k], oa[k], rtol=2e-6, atol=2e-6, msg=lambda m: f"factored {k}: {m}")
    torch.testing.assert_close(oc["rgb"], oa["rgb"], rtol=0, atol=8 * 2.0 ** -24, msg=lambda m: f"factored rgb: {m}")
    torch.testing.assert_close(oc["semantics"], oa["semantics"], rtol=2e-5, atol=2e-6)
    for k in la:
        torch.testing.assert_close(lc[k], la[k], rtol=2e-5, atol=1e-9, msg=lambda m: f"factored {k}: {m}")
    # (the rays of this scene saturate: 1 - accumulation is 0 or one ulp, so the sky branch's gradients are that ulp times the
    # output gradients -- 1e-8 of the other gradients and decided by the last bit of the accumulation; they are compared wherever
    # the accumulations agree bit for bit, i.e. between the unfactored nodes above)
    g_max = max(float(v.abs().max()) for v in ga.values())
    bad = {}
    for k in ga:
        s = float(ga[k].abs().max()) + 1e-30
        if k.startswith("sky_model.") and s < 1e-6 * g_max:
            continue
        err = float(((gc[k] - ga[k]).abs() / s - 1e-4 * ga[k].abs() / s).max())
        if not err <= 2e-5:
            bad[k] = f"{err:.2e}"
    assert not bad, f"factored gradients: {bad}"


def test_proposal_side_stream_changes_nothing_but_the_schedule():
    """During training the proposal networks run on a side stream (presight_amd.ops.side_stream) so that their backward overlaps
    the main field's; with plain autograd gradients (AccumulateGrad: the engine orders the streams) and with a flat in-place
    gradient buffer (the owner joins the side stream) the losses and every gradient are what the single-stream run gives."""
    import bench
    from presight_amd import ops
    from presight_amd.dist import FlatGrads
    from presight_amd.trainer import Trainer

    dev = torch.device("cuda:0")

    def one(side: bool, flat: bool):
        ops.SIDE_STREAM = side
        try:
            model, scene = bench.build_model(dev, seed=5, config="cfg2")
            batch = bench.make_batches(scene, dev, 1, 0, rays=2048)[0]
            torch.manual_seed(3)
            if flat:
                tr = Trainer(model, scene, 1, exchange="allreduce", fused_table_adam=False)  # (this test reads the table gradients)
                ld, _ = tr.step(batch)
                torch.cuda.synchronize()
                # (the trainer's bucket order, hence the layout of its flat buffers, follows the stream setting: compare by name;
                # the gradients -- views into the flat buffer -- stay in place until the next step clears them)
                return ({k: float(v) for k, v in ld.items()},
                        torch.cat([p.grad.detach().reshape(-1) for _, p in sorted(model.named_parameters()) if p.grad is not None]).clone())
            o, d, pa, dn = ops.generate_rays(batch["ray_indices"], scene["c2w"], scene["fx"], scene["fy"], scene["cx"], scene["cy"])
            from presight_amd.rays import RayBundle

            rb = RayBundle(o, d, pa, camera_indices=batch["ray_indices"][:, 0:1], metadata={"video_id": batch["video_ids"][:, None], "directions_norm": dn})
            model.train()
            model.proposal_sampler._steps_since_update = 1 << 30
            out = model(rb)
            ld = model.get_loss_dict(out, batch)
            sum(ld.values()).backward()
            g = torch.cat([p.grad.reshape(-1) for _, p in sorted(model.named_parameters()) if p.grad is not None])  # caller's stream, no sync
            return {k: float(v) for k, v in ld.items()}, g
        finally:
            ops.SIDE_STREAM = True

    for flat in (False, True):
        (la, ga), (lb, gb) = one(False, flat), one(True, flat)
        assert la == lb, (la, lb)
        assert ga.shape == gb.shape
        s = float(ga.abs().max())
        # hash-table gradients are bit-reproducible (integer accumulation), MLP / embedding gradients use float atomics
        torch.testing.assert_close(gb / s, ga / s, rtol=1e-4, atol=1e-6)
        assert float(ga.abs().max()) > 0


@pytest.mark.gpu
def test_dense_level0_histogram_in_a_training_step_is_bit_identical():
    """PRESIGHT_DENSE_LEVEL0 (field_ops.DENSE_LEVEL0: the coarsest level of every one-table backward summed into a dense int64
    histogram instead of written as records -- off by default, EXPERIMENTS.md A.7): a cfg-2 training step gives the same hash tables
    bit for bit, with the tables' Adam step inside their backward (the default of a single-process trainer) and with written
    gradients + the optimizer kernel."""
    import bench
    from presight_amd import field_ops
    from presight_amd.trainer import Trainer

    dev = torch.device("cuda:0")

    def tables_after_one_step(dense0: bool, fused: bool):
        field_ops.DENSE_LEVEL0 = dense0
        try:
            model, scene = bench.build_model(dev, seed=11, config="cfg2")
            batch = bench.make_batches(scene, dev, 1, 0, rays=4096)[0]
            torch.manual_seed(4)
            tr = Trainer(model, scene, 1, exchange="allreduce", fused_table_adam=fused)
            tr.step(batch)  # (ONE step: the table gradient is integer-accumulated = deterministic; the MLP gradients behind a second one use float atomics)
            torch.cuda.synchronize()
            return {k: v.detach().clone() for k, v in model.state_dict().items() if k.endswith("hash_table")}
        finally:
            field_ops.DENSE_LEVEL0 = False

    for fused in (True, False):
        a, b = tables_after_one_step(False, fused), tables_after_one_step(True, fused)
        assert len(a) >= 3 and a.keys() == b.keys()  # (main + two proposal tables, also under their mlp_base alias keys)
        for k in a:
            assert torch.equal(a[k], b[k]), (fused, k)
        assert any(bool((v != 0).any()) for v in a.values())
