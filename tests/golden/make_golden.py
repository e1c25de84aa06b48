"""Generate golden fixtures by running the REFERENCE's own pure-torch path (container only).

    python tests/golden/make_golden.py

Imports /root/reference/nerfstudio-0.3.3 (with stub modules for packages that are missing from
this image, see _ref_import.py), pushes deterministic parameters into the reference's modules,
runs them on seeded inputs and stores inputs + outputs (+ autograd gradients) as small .npz files
in this directory.  The fixtures are data only; no reference source travels.

Inputs/parameters come from oracle.nerf_oracle.make_params / make_scene / make_batch (our own
deterministic generators), the *expected outputs* come from the reference.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _ref_import as R  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402

ns = R.ref_modules()
torch.manual_seed(0)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


class PatchedRand:
    """Replace torch.rand by a queue of pre-drawn tensors (ray_samplers.py:105,322)."""

    def __init__(self, draws):
        self.draws = list(draws)
        self.orig = torch.rand

    def __enter__(self):
        def fake(*size, **kw):
            d = self.draws.pop(0)
            shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
            assert tuple(d.shape) == shape, (d.shape, shape)
            return d.clone()

        torch.rand = fake
        return self

    def __exit__(self, *a):
        torch.rand = self.orig


# ---------------------------------------------------------------------------- hash grid
def gold_hashgrid():
    g = torch.Generator().manual_seed(11)
    cases = {}
    # (tag, L, base, max, log2T, F, N)
    for tag, L, base, mx, l2t, F, N in [
        ("kat", 2, 16, 64, 5, 2, 0),
        ("cfg2small", 16, 16, 2048, 10, 2, 300),
        ("prodsmall", 10, 16, 16384, 10, 4, 300),
        ("prop0small", 8, 16, 1024, 11, 1, 300),
        ("prop1small", 8, 16, 4096, 11, 1, 300),
    ]:
        enc = ns.encodings.HashEncoding(num_levels=L, min_res=base, max_res=mx, log2_hashmap_size=l2t,
                                        features_per_level=F, implementation="torch")
        T = 1 << l2t
        if tag == "kat":
            table = (torch.arange(T * L * F, dtype=torch.float32).view(T * L, F)) * 1e-3
            x = torch.tensor([[0.5, 0.25, 0.75], [0.3, 0.7, 0.1], [0.0, 0.0, 0.0], [1.0, 1.0, 1.0]])
        else:
            table = (torch.rand(T * L, F, generator=g) * 2 - 1) * 1e-1
            x = torch.rand(N, 3, generator=g)
            # exact lattice hits, zeros (what masked-out samples become) and ones
            x[:8] = torch.tensor([0.0, 0.0, 0.0])
            x[8:12] = torch.tensor([1.0, 1.0, 1.0])
            x[12:20] = torch.round(x[12:20] * 16) / 16
        enc.hash_table.data = table.clone()
        out = enc(x)
        cot = torch.rand(out.shape, generator=g) - 0.5
        (gt,) = torch.autograd.grad((out * cot).sum(), enc.hash_table)
        # indices via the reference's own hash_fn
        scaled = x[:, None, :] * enc.scalings.view(-1, 1)
        c = torch.ceil(scaled).type(torch.int32)
        f = torch.floor(scaled).type(torch.int32)
        pick = {"c": c, "f": f}
        idx = torch.stack([enc.hash_fn(torch.stack([pick[k[0]][..., 0], pick[k[1]][..., 1], pick[k[2]][..., 2]], -1))
                           for k in O._CORNERS], -1)
        cases.update({f"{tag}_x": x, f"{tag}_table": table, f"{tag}_scalings": enc.scalings, f"{tag}_out": out,
                      f"{tag}_idx": idx, f"{tag}_cot": cot, f"{tag}_grad_table": gt,
                      f"{tag}_meta": np.array([L, base, mx, l2t, F])})
    # full-size index-only checks (table contents irrelevant): cfg2 + production + prop nets
    for tag, L, base, mx, l2t in [("cfg2", 16, 16, 2048, 19), ("prod", 10, 16, 16384, 20), ("prop0", 8, 16, 1024, 20),
                                   ("prop1", 8, 16, 4096, 20)]:
        enc = ns.encodings.HashEncoding(num_levels=L, min_res=base, max_res=mx, log2_hashmap_size=4,
                                        features_per_level=1, implementation="torch")
        enc.hash_table_size = 1 << l2t
        enc.hash_offset = torch.arange(L) * enc.hash_table_size
        x = torch.rand(128, 3, generator=g)
        scaled = x[:, None, :] * enc.scalings.view(-1, 1)
        c = torch.ceil(scaled).type(torch.int32)
        f = torch.floor(scaled).type(torch.int32)
        pick = {"c": c, "f": f}
        idx = torch.stack([enc.hash_fn(torch.stack([pick[k[0]][..., 0], pick[k[1]][..., 1], pick[k[2]][..., 2]], -1))
                           for k in O._CORNERS], -1)
        cases.update({f"{tag}_full_x": x, f"{tag}_full_scalings": enc.scalings, f"{tag}_full_idx": idx,
                      f"{tag}_full_meta": np.array([L, base, mx, l2t, 1])})
    save("hashgrid", **cases)


# ---------------------------------------------------------------------------- small ops
def gold_ops():
    g = torch.Generator().manual_seed(12)
    # contraction + normalisation (a6)
    aabb = torch.tensor([[-1.0, -2.0, -0.5], [3.0, 1.0, 0.7]])
    p = (torch.rand(400, 3, generator=g) - 0.5) * 30
    p[:50] = aabb[0] + (aabb[1] - aabb[0]) * torch.rand(50, 3, generator=g)  # inside
    p[50] = aabb[0]
    p[51] = aabb[1]
    q = ns.ingp.get_normalized_position(p, aabb)
    q = ns.spatial.SceneContraction(order=float("inf"))(q)
    q = (q + 2.0) / 4.0
    sel = ((q > 0.0) & (q < 1.0)).all(dim=-1)
    u = q * sel[..., None]
    kat = ns.spatial.SceneContraction(order=float("inf"))(torch.tensor([[2.0, -4.0, 1.0]]))
    # SH (a10)
    d = torch.nn.functional.normalize(torch.randn(300, 3, generator=g), dim=-1)
    d[0] = torch.tensor([0.0, 0.0, 1.0])
    sh = ns.encodings.SHEncoding(levels=4, implementation="torch")(ns.ingp.get_normalized_directions(d))
    # trunc_exp (a9)
    xr = (torch.rand(200, generator=g) - 0.5) * 60
    xr.requires_grad_(True)
    y = ns.activations.trunc_exp(xr)
    cot = torch.rand(200, generator=g)
    (gx,) = torch.autograd.grad((y * cot).sum(), xr)
    # MLPs (a8) - every shape of SURVEY 8a row a8
    mlps = {}
    for tag, dims, act in [("base", [32, 64, 80], None), ("sem", [64, 64, 64, 64], None), ("rgb", [47, 64, 64, 3], "sig"),
                           ("prop", [8, 64, 1], None), ("skyrgb", [32, 32, 32, 3], "sig"), ("skysem", [16, 32, 32, 64], None),
                           ("base_prod", [40, 64, 80], None), ("tiny", [4, 32, 80], None)]:
        m = ns.mlp.MLP(in_dim=dims[0], num_layers=len(dims) - 1, layer_width=dims[1], out_dim=dims[-1],
                       activation=torch.nn.ReLU(), out_activation=torch.nn.Sigmoid() if act else None,
                       implementation="torch")
        x = torch.randn(200, dims[0], generator=g, requires_grad=True)
        yy = m(x)
        ct = torch.randn(yy.shape, generator=g)
        grads = torch.autograd.grad((yy * ct).sum(), [x] + list(m.parameters()))
        mlps[f"mlp_{tag}_x"] = x
        mlps[f"mlp_{tag}_y"] = yy
        mlps[f"mlp_{tag}_cot"] = ct
        mlps[f"mlp_{tag}_gx"] = grads[0]
        for i, layer in enumerate(m.layers):
            mlps[f"mlp_{tag}_W{i}"] = layer.weight
            mlps[f"mlp_{tag}_b{i}"] = layer.bias
            mlps[f"mlp_{tag}_gW{i}"] = grads[1 + 2 * i]
            mlps[f"mlp_{tag}_gb{i}"] = grads[2 + 2 * i]
        mlps[f"mlp_{tag}_sigmoid"] = np.array(1 if act else 0)
    # router (a5)
    cent = torch.randn(16, 3, generator=g)
    pts = torch.randn(500, 3, generator=g) * 2
    assign = torch.cdist(pts, cent).argmin(dim=1)
    save("ops", aabb=aabb, p=p, u=u, sel=sel, contract_kat=kat, d=d, sh=sh, te_x=xr, te_y=y, te_cot=cot, te_gx=gx,
         route_pts=pts, route_centroids=cent, route_assign=assign, **mlps)


# ---------------------------------------------------------------------------- rays / samplers / renderers / losses
def _cameras(scene):
    C = scene["c2w"].shape[0]
    return ns.cameras.Cameras(camera_to_worlds=scene["c2w"], fx=scene["fx"], fy=scene["fy"], cx=scene["cx"], cy=scene["cy"],
                              width=scene["W"], height=scene["H"])


def _ray_bundle(scene, ray_indices):
    cams = _cameras(scene)
    coords = cams.get_image_coords()[ray_indices[:, 1], ray_indices[:, 2]]
    return cams.generate_rays(camera_indices=ray_indices[:, 0:1], coords=coords)


def gold_sampling():
    cfg = O.tiny_config()
    scene = O.make_scene(cfg)
    batch = O.make_batch(cfg, scene, 32, step=3)
    rb = _ray_bundle(scene, batch["ray_indices"])
    arrs = dict(ray_indices=batch["ray_indices"], c2w=scene["c2w"], fx=scene["fx"], fy=scene["fy"], cx=scene["cx"],
                cy=scene["cy"], origins=rb.origins, directions=rb.directions, pixel_area=rb.pixel_area,
                directions_norm=rb.metadata["directions_norm"])
    thr = 5.0
    sampler = ns.samplers.SpacedSampler(
        spacing_fn=lambda x: torch.where(x < thr, x / (2 * thr), 1 - 1 / (2 * x / thr)),
        spacing_fn_inv=lambda x: torch.where(x < 0.5, x * (2 * thr), thr / (2 - 2 * x)),
        single_jitter=True)
    g = torch.Generator().manual_seed(13)
    for mode in ("train", "eval"):
        sampler.train(mode == "train")
        col = ns.colliders.NearFarCollider(near_plane=0.005, far_plane=50.0)
        col.train(mode == "train")
        rbc = col(_ray_bundle(scene, batch["ray_indices"]))
        jit = torch.rand(32, 1, generator=g)
        with PatchedRand([jit]):
            rs = sampler(rbc, num_samples=128)
        arrs[f"sp_{mode}_jitter"] = jit
        arrs[f"sp_{mode}_starts"] = rs.frustums.starts[..., 0]
        arrs[f"sp_{mode}_ends"] = rs.frustums.ends[..., 0]
        arrs[f"sp_{mode}_sstarts"] = rs.spacing_starts[..., 0]
        arrs[f"sp_{mode}_sends"] = rs.spacing_ends[..., 0]
        arrs[f"sp_{mode}_positions"] = rs.frustums.get_positions()
        # weights from a random density, then pdf resample
        sigma = torch.rand(32, 128, 1, generator=g) * 3
        sigma[:4] = 0.0  # zero-density rays exercise the eps padding path
        sigma[4, 10] = 1e4  # saturating sample
        sigma.requires_grad_(True)
        w = rs.get_weights(sigma)
        cot = torch.rand(w.shape, generator=g)
        (gs,) = torch.autograd.grad((w * cot).sum(), sigma)
        arrs[f"w_{mode}_sigma"] = sigma[..., 0]
        arrs[f"w_{mode}_weights"] = w[..., 0]
        arrs[f"w_{mode}_cot"] = cot[..., 0]
        arrs[f"w_{mode}_gsigma"] = gs[..., 0]
        pdf = ns.samplers.PDFSampler(include_original=False, single_jitter=True)
        pdf.train(mode == "train")
        jit2 = torch.rand(32, 1, generator=g)
        anneal = 0.37
        with PatchedRand([jit2]):
            rs2 = pdf(rbc, rs, torch.pow(w.detach(), anneal), num_samples=64, eps=torch.finfo(torch.float32).eps)
        arrs[f"pdf_{mode}_jitter"] = jit2
        arrs[f"pdf_{mode}_anneal"] = np.array(anneal)
        arrs[f"pdf_{mode}_sstarts"] = rs2.spacing_starts[..., 0]
        arrs[f"pdf_{mode}_sends"] = rs2.spacing_ends[..., 0]
        arrs[f"pdf_{mode}_starts"] = rs2.frustums.starts[..., 0]
        arrs[f"pdf_{mode}_ends"] = rs2.frustums.ends[..., 0]
        # renderers on level-2 samples
        sigma2 = torch.rand(32, 64, 1, generator=g) * 40
        sigma2[:3] = 0.0
        w2 = rs2.get_weights(sigma2)
        rgb = torch.rand(32, 64, 3, generator=g)
        sem = torch.rand(32, 64, 64, generator=g)
        rr = ns.renderers.RGBRenderer(background_color="black")
        rr.train(mode == "train")
        arrs[f"r_{mode}_sigma"] = sigma2[..., 0]
        arrs[f"r_{mode}_w"] = w2[..., 0]
        arrs[f"r_{mode}_rgb_in"] = rgb
        arrs[f"r_{mode}_sem_in"] = sem
        arrs[f"r_{mode}_rgb"] = rr(rgb=rgb, weights=w2)
        arrs[f"r_{mode}_acc"] = ns.renderers.AccumulationRenderer()(weights=w2)
        arrs[f"r_{mode}_depth"] = ns.renderers.DepthRenderer(method="threshold")(weights=w2, ray_samples=rs2)
        arrs[f"r_{mode}_expdepth"] = ns.renderers.DepthRenderer(method="expected")(weights=w2, ray_samples=rs2)
        arrs[f"r_{mode}_sem"] = torch.sum(sem * w2, dim=-2)
    # appendix A known answers (eval, near 0, far 50, 8 samples)
    sampler.eval()
    rb1 = rb[:1]
    rb1.nears = torch.zeros(1, 1)
    rb1.fars = torch.full((1, 1), 50.0)
    rs = sampler(rb1, num_samples=8)
    arrs["kat_sp_starts"] = rs.frustums.starts[..., 0]
    arrs["kat_sp_ends"] = rs.frustums.ends[..., 0]
    w1 = torch.zeros(1, 8, 1)
    w1[0, 2, 0] = 1.0
    pdf = ns.samplers.PDFSampler(include_original=False, single_jitter=True)
    pdf.eval()
    rs2 = pdf(rb1, rs, w1, num_samples=4, eps=torch.finfo(torch.float32).eps)
    arrs["kat_pdf_starts"] = rs2.frustums.starts[..., 0]
    arrs["kat_pdf_ends"] = rs2.frustums.ends[..., 0]
    save("sampling", **arrs)


def gold_losses():
    g = torch.Generator().manual_seed(14)
    R_ = 40

    def rand_bins(S):
        b = torch.sort(torch.rand(R_, S + 1, generator=g), dim=-1).values
        b[:, 0], b[:, -1] = 0.0, 1.0
        return b

    def fake_samples(b):
        return ns.rays.RaySamples(frustums=None, spacing_starts=b[:, :-1, None], spacing_ends=b[:, 1:, None])

    bl = [rand_bins(128), rand_bins(64), rand_bins(64)]
    wl = [torch.rand(R_, S, generator=g).requires_grad_(True) for S in (128, 64, 64)]
    wl_n = [w / w.sum(-1, keepdim=True) * 0.9 for w in wl]
    il = ns.ps_losses.z_anti_anliasing_interlevel_loss([w[..., None] for w in wl_n], [fake_samples(b) for b in bl],
                                                       pulse_width=(0.03, 0.003))
    g_il = torch.autograd.grad(il, wl[:2], retain_graph=True)
    dl = ns.losses.distortion_loss([w[..., None] for w in wl_n], [fake_samples(b) for b in bl])
    (g_dl,) = torch.autograd.grad(dl, wl[2])
    acc = torch.rand(R_, 1, generator=g).requires_grad_(True)
    acc.data[0] = 0.0
    acc.data[1] = 1.0
    skym = (torch.rand(R_, 1, generator=g) < 0.3).float()
    sl = ns.ps_losses.sky_loss(acc, skym)
    (g_sl,) = torch.autograd.grad(sl, acc)
    pred = torch.rand(R_, 64, generator=g).requires_grad_(True)
    tgt = torch.rand(R_, 64, generator=g) * 1.4 - 0.2
    sm = ns.ps_losses.semantic_loss(pred, tgt, clip=True)
    (g_sm,) = torch.autograd.grad(sm, pred)
    save("losses", bins0=bl[0], bins1=bl[1], bins2=bl[2], w0=wl_n[0], w1=wl_n[1], w2=wl_n[2], w0_raw=wl[0], w1_raw=wl[1],
         w2_raw=wl[2], interlevel=il, g_interlevel_w0=g_il[0], g_interlevel_w1=g_il[1], distortion=dl, g_distortion_w2=g_dl,
         acc=acc, sky_mask=skym, sky_loss=sl, g_sky=g_sl, sem_pred=pred, sem_tgt=tgt, sem_loss=sm, g_sem=g_sm)


def gold_depth_losses():
    """lidar / monodepth supervision (ns/model_components/PreSight/losses.py:28-103), exactly as get_loss_dict calls them
    (nerfacto_nusc_ms.py:576-629): depth [R,1] in metres, steps and predicted depth divided by pose_scale_factor."""
    g = torch.Generator().manual_seed(15)
    R_, S = 48, 64
    scale = 0.05
    edges = torch.sort(torch.rand(R_, S + 1, generator=g) * 60.0 * scale, dim=-1).values
    steps = ((edges[:, :-1] + edges[:, 1:]) / 2)[..., None] / scale  # [R,S,1] metres
    w = (torch.rand(R_, S, 1, generator=g) * 0.1).requires_grad_(True)
    depth = torch.rand(R_, 1, generator=g) * 90.0  # some < 1, some > upper bound
    depth[0], depth[1], depth[2] = 0.5, 76.0, 39.9
    sky = (torch.rand(R_, 1, generator=g) < 0.25).float()
    pred = (torch.rand(R_, 1, generator=g) * 80.0).requires_grad_(True)
    out = dict(edges=edges, steps=steps, w=w, depth=depth, sky=sky, pred=pred, pose_scale_factor=np.float32(scale))
    for tag, sigma, ub, use_sky in (("lidar", 5.0, 75.0, False), ("mono", 2.6, 40.0, True)):
        los = ns.ps_losses.line_of_sight_loss(weights=w, termination_depth=depth, steps=steps, sigma=sigma,
                                              sky_mask=sky if use_sky else None, upper_bound=ub)
        (g_los,) = torch.autograd.grad(los, w)
        out.update({f"los_{tag}": los, f"g_los_{tag}": g_los, f"sigma_{tag}": np.float32(sigma), f"ub_{tag}": np.float32(ub)})
    ed = ns.ps_losses.expected_depth_loss(termination_depth=depth, predicted_depth=pred, upper_bound=75.0)
    (g_ed,) = torch.autograd.grad(ed, pred)
    out.update(ed_lidar=ed, g_ed_lidar=g_ed)
    for tag, inv in (("mono", False), ("mono_inv", True)):
        em = ns.ps_losses.expected_monodepth_loss(termination_depth=depth, predicted_depth=pred, sky_mask=sky, upper_bound=40.0,
                                                  inverse=inv)
        (g_em,) = torch.autograd.grad(em, pred)
        out.update({f"ed_{tag}": em, f"g_ed_{tag}": g_em})
    save("depth_losses", **out)


# ---------------------------------------------------------------------------- fields + whole model
def _build_ref_model(cfg, scene, P, **conf_overrides):
    mod = __import__("importlib").import_module("nerfstudio.models.PreSight.nerfacto_nusc_ms")
    m = cfg["main"]
    conf = mod.NerfactoNuscMSModelConfig(
        near_plane=cfg["near"], far_plane=cfg["far"], piecewise_sampler_threshold=cfg["thr"],
        hidden_dim=m["hidden_dim"], hidden_dim_color=m["hidden_dim_color"], num_levels=m["num_levels"],
        base_res=m["base_res"], max_res=m["max_res"], log2_hashmap_size=m["log2_hashmap_size"],
        features_per_level=m["features_per_level"],
        proposal_net_args_list=[dict(features_per_level=p["features_per_level"], log2_hashmap_size=p["log2_hashmap_size"],
                                     num_levels=p["num_levels"], base_res=p["base_res"], max_res=p["max_res"],
                                     hidden_dim=p["hidden_dim"], use_linear=False) for p in cfg["props"]],
        implementation="torch", use_lidar_loss=False, distortion_loss_mult=cfg["distortion_loss_mult"],
        sky_mlp_dims=cfg["sky"]["width"], num_sky_mlp_layers=cfg["sky"]["num_layers"], **conf_overrides,
    )
    conf.enable_collider = False
    conf.collider_params = None
    model = mod.NerfactoNuscMSModel(conf, scene_box=None, num_train_data=-1, num_train_cameras=cfg["num_cameras"],
                                    num_train_videos=cfg["num_videos"], dino_to_rgb=scene["dino_to_rgb"], centroids=scene["centroids"],
                                    aabbs=scene["aabbs"])
    sd = model.state_dict()
    missing = [k for k in P if k not in sd]
    assert not missing, missing
    for k, v in P.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
    model.load_state_dict({**sd, **P})
    return model, mod


def gold_model():
    cfg = O.tiny_config()
    cfg["num_fields"] = 3
    for p in [cfg["main"]] + cfg["props"]:
        p["log2_hashmap_size"] = 9
    scene = O.make_scene(cfg)
    P = O.make_params(cfg, seed=5, table_scale=0.3)
    # density heads biased so that accumulations spread over (0,1) instead of saturating at 1 - 2^-23 (a saturated
    # ray sits on the bound of sky_loss' clip and makes (1-acc) a pure rounding artefact)
    for k in range(cfg["num_fields"]):
        P[f"field.fields.{k}.mlp_base_mlp.layers.1.bias"][0] = -2.5
        for i in range(2):
            P[f"proposal_networks.{i}.fields.{k}.mlp_base.1.layers.1.bias"][0] = -2.0
    model, mod = _build_ref_model(cfg, scene, P)
    R_ = 64
    batch = O.make_batch(cfg, scene, R_, step=1)
    rb = _ray_bundle(scene, batch["ray_indices"])
    rb.metadata["video_id"] = batch["video_ids"][:, None]
    arrs = {"P_" + k: v for k, v in P.items()}
    arrs.update({"B_" + k: v for k, v in batch.items()})
    arrs.update(centroids=scene["centroids"], aabbs=scene["aabbs"])
    # --- per-field fixtures (sub-field 1)
    f1 = model.field.fields[1]
    g = torch.Generator().manual_seed(15)
    lo, hi = scene["aabbs"][1][0], scene["aabbs"][1][1]
    pos = lo + (hi - lo) * (torch.rand(200, 3, generator=g) * 1.6 - 0.3)
    dirs = torch.nn.functional.normalize(torch.randn(200, 3, generator=g), dim=-1)
    app = torch.randn(200, 16, generator=g)
    dens, emb = f1.density_fn(pos)
    fo = f1.get_outputs(dirs, density_embedding=emb, appearance_embedding=app)
    cots = [torch.rand(dens.shape, generator=g), torch.rand(200, 3, generator=g) - 0.5, torch.rand(200, 64, generator=g) - 0.5]
    scalar = (dens * cots[0]).sum() + (fo[ns.field_heads.FieldHeadNames.RGB] * cots[1]).sum() + \
        (fo[ns.field_heads.FieldHeadNames.SEMANTICS] * cots[2]).sum()
    names = [n for n, _ in f1.named_parameters()]
    gr = torch.autograd.grad(scalar, list(f1.parameters()))
    arrs.update(F_pos=pos, F_dirs=dirs, F_app=app, F_density=dens, F_embedding=emb,
                F_rgb=fo[ns.field_heads.FieldHeadNames.RGB], F_sem=fo[ns.field_heads.FieldHeadNames.SEMANTICS],
                F_cot_density=cots[0], F_cot_rgb=cots[1], F_cot_sem=cots[2], F_semantic_fn=f1.semantic_fn(pos))
    for n, gg in zip(names, gr):
        arrs["Fg_" + n] = gg
    p1 = model.proposal_networks[0].fields[1]
    pd = p1.density_fn(pos)
    cot = torch.rand(pd.shape, generator=g)
    gr = torch.autograd.grad((pd * cot).sum(), list(p1.parameters()))
    arrs.update(Pp_density=pd, Pp_cot=cot)
    for (n, _), gg in zip(p1.named_parameters(), gr):
        arrs["Ppg_" + n] = gg
    sk = model.sky_model.fields[1]
    so = sk.get_outputs(dirs, app)
    arrs.update(S_rgb=so[ns.field_heads.FieldHeadNames.RGB], S_sem=so[ns.field_heads.FieldHeadNames.SEMANTICS])
    # --- whole model, training mode
    model.train()
    model.proposal_sampler._anneal = 0.6
    # Conditioning of the fixture: the sky BCE takes -log(accumulation) of every non-sky ray; a ray whose accumulation
    # is ~1e-7 (1-exp(-x) cancellation noise) would make the expected gradients depend on last-ulp rounding.  Such rays
    # (nothing along the ray) are labelled sky, as they would be in real data.
    with torch.no_grad(), PatchedRand([batch["jitter"][0], batch["jitter"][1], batch["jitter"][2]]):
        acc0 = model(_with_meta(_ray_bundle(scene, batch["ray_indices"]), batch))["accumulation"][:, 0]
    batch["sky"] = torch.where(acc0 < 1e-3, torch.ones_like(batch["sky"]), batch["sky"])
    arrs["B_sky"] = batch["sky"]
    model.proposal_sampler._steps_since_update = 0
    with PatchedRand([batch["jitter"][0], batch["jitter"][1], batch["jitter"][2]]):
        out = model(rb)
    gt = {"rgb": batch["rgb"], "features": batch["features"], "sky": batch["sky"]}
    ld = model.get_loss_dict(out, gt)
    total = sum(ld.values())
    model.zero_grad()
    total.backward()
    for k in ["rgb", "accumulation", "depth", "expected_depth", "semantics", "prop_depth_0", "prop_depth_1"]:
        arrs["T_" + k] = out[k]
    for i in range(3):
        arrs[f"T_weights_{i}"] = out["weights_list"][i][..., 0]
        arrs[f"T_sbins_{i}"] = torch.cat([out["ray_samples_list"][i].spacing_starts[..., 0],
                                          out["ray_samples_list"][i].spacing_ends[..., -1:, 0]], -1)
    for k, v in ld.items():
        arrs["TL_" + k] = v
    for n, p in model.named_parameters():
        if n in P:
            arrs["TG_" + n] = p.grad if p.grad is not None else torch.zeros_like(p)
    arrs["T_anneal"] = np.array(0.6)
    # --- whole model, eval mode
    model.eval()
    with torch.no_grad():
        oute = model(_with_meta(_ray_bundle(scene, batch["ray_indices"]), batch))
        for k in ["rgb", "accumulation", "depth", "expected_depth", "semantics", "dino_rgb"]:
            arrs["E_" + k] = oute[k]
        dd = model.get_depth(_with_meta(_ray_bundle(scene, batch["ray_indices"]), batch))
        arrs["E_get_depth"] = dd["depth"]
        arrs["E_get_expected_depth"] = dd["expected_depth"]
        # extraction queries (extract_priors.py:133-138)
        pts = scene["c2w"][:, :, 3][torch.randint(0, scene["c2w"].shape[0], (150,), generator=g)] + \
            torch.randn(150, 3, generator=g) * 0.3
        dl = [p.density_fn(pts).squeeze(-1) for p in model.proposal_networks]
        dl.append(model.field.density_fn(pts)[0].squeeze(-1))
        arrs["X_pts"] = pts
        arrs["X_density_mean"] = torch.stack(dl, 0).mean(0)
        arrs["X_feats"] = model.field.semantic_fn(pts).clip(0.0, 1.0).to(torch.float16)
        cm = __import__("importlib").import_module("nerfstudio.utils.colormaps")
        arrs["X_colors"] = cm.apply_feature_colormap(arrs["X_feats"], scene["dino_to_rgb"])
    save("model", **arrs)


def gold_model_k8():
    """One training step (forward, 5 losses, backward: every parameter gradient) of the REFERENCE's NerfactoNuscMSModel with K = 8
    routed sub-fields at the production shape -- the multi-sub-field kernels' reference-generated fixture (model.npz is K = 3,
    2-level)."""
    cfg = O.prod_shaped_config(8)
    scene = O.make_scene(cfg)
    P = O.make_params(cfg, seed=21, table_scale=0.3)
    for k in range(cfg["num_fields"]):
        P[f"field.fields.{k}.mlp_base_mlp.layers.1.bias"][0] = -2.5
        for i in range(2):
            P[f"proposal_networks.{i}.fields.{k}.mlp_base.1.layers.1.bias"][0] = -2.0
    model, mod = _build_ref_model(cfg, scene, P)
    batch = O.make_batch(cfg, scene, 160, step=2)
    model.train()
    model.proposal_sampler._anneal = 0.8
    with torch.no_grad(), PatchedRand([batch["jitter"][0], batch["jitter"][1], batch["jitter"][2]]):
        acc0 = model(_with_meta(_ray_bundle(scene, batch["ray_indices"]), batch))["accumulation"][:, 0]
    batch["sky"] = torch.where(acc0 < 1e-3, torch.ones_like(batch["sky"]), batch["sky"])  # see gold_model
    model.proposal_sampler._steps_since_update = 0
    with PatchedRand([batch["jitter"][0], batch["jitter"][1], batch["jitter"][2]]):
        out = model(_with_meta(_ray_bundle(scene, batch["ray_indices"]), batch))
    gt = {"rgb": batch["rgb"], "features": batch["features"], "sky": batch["sky"]}
    ld = model.get_loss_dict(out, gt)
    model.zero_grad()
    sum(ld.values()).backward()
    arrs = {"B_" + k: v for k, v in batch.items()}
    arrs.update(centroids=scene["centroids"], aabbs=scene["aabbs"], seed=np.array(21), T_anneal=np.array(0.8))
    for k in ["rgb", "accumulation", "depth", "expected_depth", "semantics", "prop_depth_0", "prop_depth_1"]:
        arrs["T_" + k] = out[k]
    for i in range(3):
        arrs[f"T_weights_{i}"] = out["weights_list"][i][..., 0]
    for k, v in ld.items():
        arrs["TL_" + k] = v
    # parameters are NOT stored (O.make_params(O.prod_shaped_config(8), seed=21, table_scale=0.3) + the two bias edits above regenerate them);
    # gradients as fp32, sub-fields that received no sample (grad None) are left out
    n = 0
    for name, p in model.named_parameters():
        if name in P and p.grad is not None and float(p.grad.abs().max()) > 0:
            arrs["TG_" + name] = p.grad
            n += 1
    arrs["n_grads"] = np.array(n)
    save("model_k8", **arrs)


def gold_losses_real():
    """interlevel / distortion losses on bins that come out of the reference's REAL sampler chain (piecewise spaced sampler ->
    PDF -> PDF on a synthetic ray batch), not sorted uniforms: the reference-generated vector the fused loss kernels are held to"""
    cfg = O.tiny_config()
    scene = O.make_scene(cfg)
    R_ = 48
    batch = O.make_batch(cfg, scene, R_, step=21)
    thr = 5.0
    sampler = ns.samplers.SpacedSampler(
        spacing_fn=lambda x: torch.where(x < thr, x / (2 * thr), 1 - 1 / (2 * x / thr)),
        spacing_fn_inv=lambda x: torch.where(x < 0.5, x * (2 * thr), thr / (2 - 2 * x)), single_jitter=True)
    sampler.train()
    col = ns.colliders.NearFarCollider(near_plane=0.005, far_plane=50.0)
    col.train()
    rbc = col(_ray_bundle(scene, batch["ray_indices"]))
    g = torch.Generator().manual_seed(22)

    def bumpy_sigma(S):
        # a few surfaces per ray: smooth background + narrow peaks, like a trained field
        base = torch.rand(R_, S, 1, generator=g) * 0.05
        centre = torch.randint(4, S - 4, (R_, 1, 1), generator=g).float()
        idx = torch.arange(S).view(1, S, 1).float()
        return base + 30.0 * torch.exp(-0.5 * ((idx - centre) / 2.5) ** 2)

    with PatchedRand([torch.rand(R_, 1, generator=g)]):
        rs0 = sampler(rbc, num_samples=128)
    w0 = rs0.get_weights(bumpy_sigma(128))
    pdf = ns.samplers.PDFSampler(include_original=False, single_jitter=True)
    pdf.train()
    with PatchedRand([torch.rand(R_, 1, generator=g)]):
        rs1 = pdf(rbc, rs0, w0, num_samples=64, eps=torch.finfo(torch.float32).eps)
    w1 = rs1.get_weights(bumpy_sigma(64))
    with PatchedRand([torch.rand(R_, 1, generator=g)]):
        rs2 = pdf(rbc, rs1, w1, num_samples=64, eps=torch.finfo(torch.float32).eps)
    w2 = rs2.get_weights(bumpy_sigma(64))
    wl = [w.detach().clone().requires_grad_(True) for w in (w0, w1, w2)]
    rsl = [rs0, rs1, rs2]
    il = ns.ps_losses.z_anti_anliasing_interlevel_loss(wl, rsl, pulse_width=(0.03, 0.003))
    g_il = torch.autograd.grad(il, wl[:2], retain_graph=True)
    dl = ns.losses.distortion_loss(wl, rsl)
    (g_dl,) = torch.autograd.grad(dl, wl[2])
    arrs = {}
    for i, rs in enumerate(rsl):
        arrs[f"sbins{i}"] = torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[:, -1:, 0]], -1)
        arrs[f"w{i}"] = wl[i][..., 0]
    save("losses_real", interlevel=il, g_interlevel_w0=g_il[0][..., 0], g_interlevel_w1=g_il[1][..., 0], distortion=dl,
         g_distortion_w2=g_dl[..., 0], **arrs)


def gold_extract():
    """the body of the reference's frame loop (ns/scripts/extract_priors.py:108-145) on the fixture model of gold_model():
    camera rays at 1/75 resolution -> get_depth_for_camera_ray_bundle -> world points -> depth / height filters -> mean density
    of the three fields, clipped fp16 semantics, PCA colours"""
    cfg = O.tiny_config()
    cfg["num_fields"] = 3
    for p in [cfg["main"]] + cfg["props"]:
        p["log2_hashmap_size"] = 9
    scene = O.make_scene(cfg)
    P = O.make_params(cfg, seed=5, table_scale=0.3)
    for k in range(cfg["num_fields"]):
        P[f"field.fields.{k}.mlp_base_mlp.layers.1.bias"][0] = 1.0  # dense enough for surfaces inside the 0.5..50 m window
        for i in range(2):
            P[f"proposal_networks.{i}.fields.{k}.mlp_base.1.layers.1.bias"][0] = 1.0
    model, mod = _build_ref_model(cfg, scene, P)
    model.eval()
    cm = __import__("importlib").import_module("nerfstudio.utils.colormaps")
    pose_scale_factor, max_depth, min_depth = 0.05, 8.6, 5.0  # a window that cuts part of every frame (script defaults: 50 / 0.5)
    cams = _cameras(scene)
    cams.rescale_output_resolution(1.0 / 75.0)
    coords = cams.get_image_coords()
    arrs = dict(scaling=np.array(1.0 / 75.0), H=np.array(int(cams.height[0])), W=np.array(int(cams.width[0])),
                pose_scale_factor=np.array(pose_scale_factor), max_depth=np.array(max_depth), min_depth=np.array(min_depth))
    frames = [0, 5, 11]
    arrs["frames"] = np.array(frames)
    with torch.no_grad():
        for depth_type in ("depth", "expected_depth"):
            for cam in frames:
                crb = cams.generate_rays(camera_indices=cam, coords=coords, aabb_box=None)
                outputs = model.get_depth_for_camera_ray_bundle(crb)
                depth = outputs[depth_type] / pose_scale_factor
                world = (crb.origins / pose_scale_factor + crb.directions * depth).view(-1, 3)
                depth = depth.flatten()
                sel = (depth < max_depth) & (depth > min_depth) & (world[:, 2] > -3.0) & (world[:, 2] < 6.0)
                world = world[sel]
                tag = f"{depth_type}_{cam}"
                arrs[f"raw_depth_{tag}"] = depth
                arrs[f"sel_{tag}"] = sel
                if len(world) == 0:  # extract_priors.py:124-126
                    continue
                dl = [p.density_fn(world * pose_scale_factor).squeeze(-1) for p in model.proposal_networks]
                dl.append(model.field.density_fn(world * pose_scale_factor)[0].squeeze(-1))
                dens = torch.stack(dl, dim=0).mean(dim=0)
                feats = model.field.semantic_fn(world * pose_scale_factor).clip(0.0, 1.0).to(torch.float16)
                colors = cm.apply_feature_colormap(feats, scene["dino_to_rgb"])
                arrs[f"world_{tag}"] = world
                arrs[f"dens_{tag}"] = dens
                arrs[f"feats_{tag}"] = feats
                arrs[f"colors_{tag}"] = colors
    save("extract", **arrs)


def gold_datafeed():
    """what the reference's loader yields for a chunk: ImageChunk.__getitem__ per pixel slot (ns/data/PreSight/my_dataset.py:52-73)
    collated by torch's DataLoader, in the order of a DistributedSampler over the chunk (my_datamanager.py:203-212)"""
    import importlib

    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler

    ds = importlib.import_module("nerfstudio.data.PreSight.my_dataset")
    g = torch.Generator().manual_seed(31)
    P_ = 1000
    widths = torch.tensor([1600, 800, 320])[torch.randint(0, 3, (P_,), generator=g)]
    heights = widths * 9 // 16
    chunk = ds.ImageChunk(
        rgbs=torch.rand(P_, 3, generator=g), segs=torch.randint(0, 19, (P_,), generator=g).to(torch.uint8),
        skies=(torch.rand(P_, generator=g) < 0.2).float(), depths=torch.rand(P_, generator=g) * 60, features=torch.rand(P_, 64, generator=g),
        pixel_indices=(torch.rand(P_, generator=g) * (widths * heights)).long(), image_indices=torch.randint(0, 240, (P_,), generator=g),
        video_ids=torch.randint(0, 6, (P_,), generator=g), widths=widths)
    arrs = dict(rgbs=chunk.rgbs, skies=chunk.skies, depths=chunk.depths, features=chunk.features, pixel_indices=chunk.pixel_indices,
                image_indices=chunk.image_indices, video_ids=chunk.video_ids, widths=chunk.widths)
    for world, rank in ((1, 0), (3, 1)):
        sampler = DistributedSampler(chunk, world, rank)
        loader = DataLoader(chunk, batch_size=96, sampler=sampler, num_workers=0, drop_last=True)
        arrs[f"order_w{world}r{rank}"] = torch.tensor(list(iter(sampler)))
        batches = list(loader)
        arrs[f"n_batches_w{world}r{rank}"] = np.array(len(batches))
        for name in ("rgb", "sky", "depth", "features", "video_id", "ray_index"):
            arrs[f"b_{name}_w{world}r{rank}"] = torch.stack([b[name] for b in batches])
    save("datafeed", **arrs)


def traj_setup():
    """(cfg, scene, params, schedule constants) of the training-trajectory fixture -- shared with tests/conftest.py through the
    values stored in the fixture itself.  Three sub-fields whose centroids are placed so that routing is NOT uniform: c0 / c1
    split the cameras (sky model: routed by ray origin), c2 sits 12 units above the rig, so only the far samples of rays pointing
    upwards reach it -- a batch drawn from the lower image half leaves sub-field 2 of every routed module without samples."""
    cfg = O.tiny_config()
    cfg["num_fields"] = 3
    cfg["num_cameras"], cfg["num_videos"] = 24, 2
    for p in [cfg["main"]] + cfg["props"]:
        p["log2_hashmap_size"] = 9
    scene = O.make_scene(cfg)
    scene["centroids"] = torch.tensor([[-0.5, 0.0, 0.0], [0.5, 0.0, 0.0], [0.0, 0.0, 12.0]])
    ext = 2.25
    scene["aabbs"] = torch.stack([torch.stack([c - ext, c + ext]) for c in scene["centroids"]])
    P = O.make_params(cfg, seed=11, table_scale=0.3)
    for k in range(cfg["num_fields"]):
        P[f"field.fields.{k}.mlp_base_mlp.layers.1.bias"][0] = -2.5
        for i in range(2):
            P[f"proposal_networks.{i}.fields.{k}.mlp_base.1.layers.1.bias"][0] = -2.0
    return cfg, scene, P


def gold_model_traj(stop_after=None, on_stop=None):
    """(stop_after / on_stop: gold_checkpoint re-runs the first iterations of this very loop and takes the reference's objects over
    at the point where its Trainer would write a checkpoint; nothing is saved then.)
    N training iterations of the REFERENCE: its NerfactoNuscMSModel (K = 3), its own training callbacks (anneal + proposal
    update schedule, nerfacto_nusc_ms.py:417-450), its Optimizers object (one torch.optim.Adam(lr 1e-2, eps 1e-15, wd 1e-5) and one
    WarmupMultiStepScheduler per parameter group, method_configs.py:158-168 with max_iterations = 60) driven exactly as
    Trainer.train_iteration does for PreSight's default update_grad_scaler=False (ns/engine/trainer.py:463-505):
        zero_grad_all -> forward -> loss = sum(loss_dict) -> grad_scaler.scale(loss).backward() -> optimizer_step_all()
        (Adam on the STILL-SCALED gradients, weight decay added to them) -> scheduler_step_all
    torch's GradScaler disables itself on a CUDA-less host, so the fixed 2**10 scale (init_grad_scale, never updated) is applied
    as a factor on the loss -- the same arithmetic grad_scaler.scale() performs on the GPU."""
    eng_opt = __import__("importlib").import_module("nerfstudio.engine.optimizers")
    eng_sch = __import__("importlib").import_module("nerfstudio.engine.my_schedulers")
    eng_cb = __import__("importlib").import_module("nerfstudio.engine.callbacks")
    cfg, scene, P = traj_setup()
    MAX_IT, N_STEPS, R_, SCALE = 60, 24, 64, 2.0 ** 10
    LOWER_HALF = (3, 12, 13, 18)   # batches drawn from image rows >= 450 (rays level or pointing down): sub-field 2 gets nothing
    FRAMES_01 = (5, 13, 20)        # batches from the cameras of frames 0-1 only: sky sub-field 1 gets no ray
    model, mod = _build_ref_model(cfg, scene, P, proposal_weights_anneal_max_num_iters=MAX_IT // 10, proposal_warmup=MAX_IT // 10)
    groups = model.get_param_groups()
    mk = lambda: {"optimizer": eng_opt.AdamOptimizerConfig(lr=1e-2, eps=1e-15, weight_decay=1e-5),  # noqa: E731
                  "scheduler": eng_sch.WarmupMultiStepSchedulerConfig(max_steps=MAX_IT, milestones=[MAX_IT // 4, MAX_IT // 2, MAX_IT * 3 // 4],
                                                                      warmup_steps=MAX_IT // 10)}
    optimizers = eng_opt.Optimizers({"proposal_networks": mk(), "fields": mk()}, groups)
    callbacks = model.get_training_callbacks(eng_cb.TrainingCallbackAttributes(optimizers=optimizers, grad_scaler=None, pipeline=None))
    name_of = {id(p): n for n, p in model.named_parameters()}
    keys = [k for k in P]
    batches = []
    for step in range(N_STEPS):
        b = O.make_batch(cfg, scene, R_, step=100 + step)
        if step in LOWER_HALF:
            b["ray_indices"][:, 1] = 450 + b["ray_indices"][:, 1] % 450
        if step in FRAMES_01:
            b["ray_indices"][:, 0] = b["ray_indices"][:, 0] % 12
            b["video_ids"] = torch.clamp(b["ray_indices"][:, 0] // scene["frames_per_video"], max=cfg["num_videos"] - 1)
        b["features"] = b["features"].to(torch.float16).float()  # stored as fp16: the run sees exactly the stored values
        batches.append(b)
    model.train()
    # conditioning (see gold_model): rays that hit nothing at the initial parameters are labelled sky
    for b in batches:
        with torch.no_grad(), PatchedRand([b["jitter"][0], b["jitter"][1], b["jitter"][2]]):
            acc0 = model(_with_meta(_ray_bundle(scene, b["ray_indices"]), b))["accumulation"][:, 0]
        b["sky"] = torch.where(acc0 < 1e-3, torch.ones_like(b["sky"]), b["sky"])
    model.proposal_sampler._steps_since_update = 0
    model.proposal_sampler._step = 0
    losses, lrs, updated, anneals, touched, snaps = [], [], [], [], [], {}
    loss_names = None
    for step in range(N_STEPS):
        b = batches[step]
        for cb in callbacks:
            cb.run_callback_at_location(step, location=eng_cb.TrainingCallbackLocation.BEFORE_TRAIN_ITERATION)
        lrs.append(optimizers.optimizers["fields"].param_groups[0]["lr"])
        assert lrs[-1] == optimizers.optimizers["proposal_networks"].param_groups[0]["lr"]
        anneals.append(float(model.proposal_sampler._anneal))
        optimizers.zero_grad_all()
        with PatchedRand([b["jitter"][0], b["jitter"][1], b["jitter"][2]]):
            out = model(_with_meta(_ray_bundle(scene, b["ray_indices"]), b))
        ld = model.get_loss_dict(out, {"rgb": b["rgb"], "features": b["features"], "sky": b["sky"]})
        import functools
        loss = functools.reduce(torch.add, ld.values())
        (loss * SCALE).backward()
        optimizers.optimizer_step_all()
        optimizers.scheduler_step_all(step)
        for cb in callbacks:
            cb.run_callback_at_location(step, location=eng_cb.TrainingCallbackLocation.AFTER_TRAIN_ITERATION)
        loss_names = list(ld.keys())
        losses.append([float(v) for v in ld.values()])
        named = dict(model.named_parameters())
        touched.append([int(named[k].grad is not None) for k in keys])
        updated.append(int(any(named[k].grad is not None for k in keys if k.startswith("proposal_networks."))))
        if step in (11, N_STEPS - 1):
            snaps[step] = {k: v.detach().clone() for k, v in model.state_dict().items() if k in P}
        if stop_after is not None and step == stop_after:
            return on_stop(model=model, optimizers=optimizers, cfg=cfg, scene=scene, P=P, step=step, batches=batches, name_of=name_of)
    touched = np.array(touched, dtype=np.int8)
    updated = np.array(updated, dtype=np.int8)
    # what this fixture is for: off-schedule proposal steps and sub-fields without samples must actually occur
    prop_keys = [i for i, k in enumerate(keys) if k.startswith("proposal_networks.")]
    assert updated[:11].all() and updated.tolist()[11:] == [0, 0, 0, 0, 0, 1] * 2 + [0], updated  # ray_samplers.py:586 with warm-up 6
    f2 = [i for i, k in enumerate(keys) if ".fields.2." in k and not k.startswith("sky_model")]
    for s in range(N_STEPS):
        if s in LOWER_HALF:
            assert not touched[s, f2].any(), s
    assert touched[[s for s in range(N_STEPS) if s not in LOWER_HALF and updated[s]]][:, f2].all()
    s1 = [i for i, k in enumerate(keys) if k.startswith("sky_model.fields.1.")]
    s2 = [i for i, k in enumerate(keys) if k.startswith("sky_model.fields.2.")]
    assert not touched[list(FRAMES_01)][:, s1].any() and touched[[s for s in range(N_STEPS) if s not in FRAMES_01]][:, s1].all()
    assert not touched[:, s2].any()
    print("updated:", updated.tolist())
    print("lr:", [f"{x:.5f}" for x in lrs])
    print("total loss:", [f"{sum(l):.4f}" for l in losses])
    arrs = dict(centroids=scene["centroids"], aabbs=scene["aabbs"], seed=np.array(11), max_iterations=np.array(MAX_IT),
                n_steps=np.array(N_STEPS), loss_scale=np.array(SCALE), loss_names=np.array(loss_names), losses=np.array(losses, dtype=np.float64),
                lr=np.array(lrs, dtype=np.float64), anneal=np.array(anneals, dtype=np.float64), updated=updated, touched=touched,
                keys=np.array(keys))
    for k in ("ray_indices", "video_ids", "rgb", "features", "sky", "jitter"):
        v = torch.stack([b[k] for b in batches])
        arrs["B_" + k] = v.to(torch.float16) if k == "features" else v   # (targets: fp16-representable values, half the bytes)
    for step, sd in snaps.items():
        for k, v in sd.items():
            arrs[f"S{step}_" + k] = v
    save("model_traj", **arrs)


def _jsonable(x):
    """optimizer / scheduler state without tensors -> plain JSON (Counter and tuple become dict / list; integer keys become strings)"""
    import collections

    if isinstance(x, collections.Counter):
        return {"__counter__": {str(k): int(v) for k, v in x.items()}}
    if isinstance(x, dict):
        return {str(k): _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, torch.Tensor):
        return x.item() if x.numel() == 1 else x.tolist()
    return x


def gold_checkpoint():
    """What the reference's Trainer.save_checkpoint (ns/engine/trainer.py:432-460) writes after iteration 11 of the trajectory run of
    gold_model_traj -- {"step", "pipeline", "optimizers", "schedulers", "scalers"} with the pipeline's `_model.`-prefixed keys (a module
    with the model as its `_model` child, like VanillaPipeline; the data manager's camera optimizer is "off" for PreSight and
    contributes no key: checked below), one torch.optim.Adam state_dict and one ChainedScheduler state_dict per parameter group --
    flattened into arrays + one JSON string, and the body of the extraction frame loop (ns/scripts/extract_priors.py:108-145) run on the
    model AS RESTORED FROM THAT CHECKPOINT through the reference's own load path (Pipeline.load_pipeline,
    ns/pipelines/base_pipeline.py:426-437).  model_traj.npz holds how the run continues (iterations 12..23)."""
    import importlib
    import json

    from torch.cuda.amp import GradScaler

    def on_stop(model, optimizers, cfg, scene, P, step, batches, name_of):
        class _DataManager(torch.nn.Module):
            def __init__(self):
                super().__init__()
                co = importlib.import_module("nerfstudio.cameras.camera_optimizers")
                self.train_camera_optimizer = co.CameraOptimizerConfig(mode="off").setup(num_cameras=cfg["num_cameras"], device="cpu")

        class _Pipeline(torch.nn.Module):  # (VanillaPipeline's module tree: datamanager + _model; `model` is a property)
            def __init__(self):
                super().__init__()
                self.datamanager = _DataManager()
                self._model = model

        pipe = _Pipeline()
        ckpt = {"step": step, "pipeline": pipe.state_dict(),
                "optimizers": {k: v.state_dict() for k, v in optimizers.optimizers.items()},
                "schedulers": {k: v.state_dict() for k, v in optimizers.schedulers.items()},
                "scalers": GradScaler(init_scale=2.0 ** 10).state_dict()}
        assert all(k.startswith("_model.") for k in ckpt["pipeline"])
        arrs, meta = {}, {"step": step, "pipeline_keys": list(ckpt["pipeline"].keys()), "optimizers": {}, "schedulers": _jsonable(ckpt["schedulers"]),
                          "scalers": _jsonable(ckpt["scalers"]),
                          "scalers_note": "torch's GradScaler is disabled on a CUDA-less host: its state_dict is {}; on a GPU the reference writes "
                                          "{scale, growth_factor, backoff_factor, growth_interval, _growth_tracker}"}
        ps = model.proposal_sampler  # NOT part of the reference's checkpoint (its resumed runs restart these); stored so that a test can
        #                              continue the uninterrupted trajectory of model_traj.npz from this checkpoint
        meta["sampler_not_in_checkpoint"] = {"steps_since_update": int(ps._steps_since_update), "step": int(ps._step), "anneal": float(ps._anneal)}
        for k, v in ckpt["pipeline"].items():
            arrs["P::" + k] = v
        groups = model.get_param_groups()
        for gname, osd in ckpt["optimizers"].items():
            names = [name_of[id(p)] for p in groups[gname]]
            assert osd["param_groups"][0]["params"] == list(range(len(names)))
            meta["optimizers"][gname] = {"param_groups": _jsonable(osd["param_groups"]), "param_names": names,
                                         "state_indices": sorted(int(i) for i in osd["state"])}
            for i, st in osd["state"].items():
                assert set(st) == {"step", "exp_avg", "exp_avg_sq"}, st.keys()
                arrs[f"O::{gname}::{i}::step"] = st["step"]
                arrs[f"O::{gname}::{i}::exp_avg"] = st["exp_avg"]
                arrs[f"O::{gname}::{i}::exp_avg_sq"] = st["exp_avg_sq"]
        # the reference's load path into a FRESH model (eval_setup -> eval_load_checkpoint -> load_pipeline): strict key match
        cfg2, scene2, P2 = traj_setup()
        fresh, _ = _build_ref_model(cfg2, scene2, {k: torch.zeros_like(v) for k, v in P2.items()},
                                    proposal_weights_anneal_max_num_iters=6, proposal_warmup=6)
        pipe2 = _Pipeline.__new__(_Pipeline)
        torch.nn.Module.__init__(pipe2)
        pipe2.datamanager, pipe2._model = _DataManager(), fresh
        state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in ckpt["pipeline"].items()}
        fresh.update_to_step(step)
        pipe2.load_state_dict(state)
        fresh.eval()
        cm = importlib.import_module("nerfstudio.utils.colormaps")
        psf = 1.0  # (after 12 iterations the fixture model is still nearly empty: depths sit near the far plane, and in the script's metre
        #            units at the real pose scale 0.05 its fixed -3 < z < 6 height window would keep two dozen points)
        cams = _cameras(scene)
        cams.rescale_output_resolution(1.0 / 75.0)
        coords = cams.get_image_coords()
        frames = [0, 7]
        arrs.update(scaling=np.array(1.0 / 75.0), H=np.array(int(cams.height[0])), W=np.array(int(cams.width[0])), pose_scale_factor=np.array(psf),
                    frames=np.array(frames))
        with torch.no_grad():
            d_all = torch.cat([fresh.get_depth_for_camera_ray_bundle(cams.generate_rays(camera_indices=c, coords=coords, aabb_box=None))["depth"].flatten()
                               for c in frames]) / psf
            lo, hi = float(torch.quantile(d_all, 0.2)), float(torch.quantile(d_all, 0.85))
            arrs.update(min_depth=np.array(lo), max_depth=np.array(hi))
            n_sel = 0
            for depth_type in ("depth", "expected_depth"):
                for cam in frames:
                    crb = cams.generate_rays(camera_indices=cam, coords=coords, aabb_box=None)
                    outputs = fresh.get_depth_for_camera_ray_bundle(crb)
                    depth = outputs[depth_type] / psf
                    world = (crb.origins / psf + crb.directions * depth).view(-1, 3)
                    depth = depth.flatten()
                    sel = (depth < hi) & (depth > lo) & (world[:, 2] > -3.0) & (world[:, 2] < 6.0)
                    world = world[sel]
                    tag = f"{depth_type}_{cam}"
                    arrs[f"raw_depth_{tag}"] = depth
                    arrs[f"sel_{tag}"] = sel
                    if len(world) == 0:
                        continue
                    n_sel += len(world)
                    dl = [p.density_fn(world * psf).squeeze(-1) for p in fresh.proposal_networks]
                    dl.append(fresh.field.density_fn(world * psf)[0].squeeze(-1))
                    arrs[f"world_{tag}"] = world
                    arrs[f"dens_{tag}"] = torch.stack(dl, dim=0).mean(dim=0)
                    feats = fresh.field.semantic_fn(world * psf).clip(0.0, 1.0).to(torch.float16)
                    arrs[f"feats_{tag}"] = feats
                    arrs[f"colors_{tag}"] = cm.apply_feature_colormap(feats, scene["dino_to_rgb"])
        print(f"checkpoint at step {step}: {len(ckpt['pipeline'])} pipeline keys, depth window {lo:.2f}..{hi:.2f}, {n_sel} hit points")
        assert n_sel > 200
        arrs["meta_json"] = np.array(json.dumps(meta))
        save("checkpoint", **arrs)

    gold_model_traj(stop_after=11, on_stop=on_stop)


def _with_meta(rb, batch):
    rb.metadata["video_id"] = batch["video_ids"][:, None]
    return rb


if __name__ == "__main__":
    which = sys.argv[1:] or ["hashgrid", "ops", "sampling", "losses", "losses_real", "depth_losses", "model", "model_k8", "model_traj", "extract", "datafeed", "checkpoint"]
    for w in which:
        globals()["gold_" + w]()
