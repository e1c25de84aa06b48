import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def t(a):
    """numpy -> torch (keeps dtype)."""
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="session")
def gold_hashgrid():
    return load_golden("hashgrid")


@pytest.fixture(scope="session")
def gold_ops():
    return load_golden("ops")


@pytest.fixture(scope="session")
def gold_sampling():
    return load_golden("sampling")


@pytest.fixture(scope="session")
def gold_losses():
    return load_golden("losses")


@pytest.fixture(scope="session")
def gold_model():
    return load_golden("model")


def model_fixture_setup(G):
    """Rebuild (cfg, scene, params, batch) of tests/golden/model.npz from the fixture itself."""
    from oracle import nerf_oracle as O

    cfg = O.tiny_config()
    cfg["num_fields"] = 3
    for p in [cfg["main"]] + cfg["props"]:
        p["log2_hashmap_size"] = 9
    scene = O.make_scene(cfg)
    scene["centroids"] = t(G["centroids"])
    scene["aabbs"] = t(G["aabbs"])
    P = {k[2:]: t(v) for k, v in G.items() if k.startswith("P_")}
    batch = {k[2:]: t(v) for k, v in G.items() if k.startswith("B_")}
    return cfg, scene, P, batch



@pytest.fixture(scope="session")
def gold_model_k8():
    return load_golden("model_k8")


def model_k8_setup(G):
    """(cfg, scene, params, batch) of tests/golden/model_k8.npz: K = 8 routed sub-fields at the production shape; the parameters are
    regenerated from the fixture's seed exactly as tests/golden/make_golden.py::gold_model_k8 made them"""
    from oracle import nerf_oracle as O

    cfg = O.prod_shaped_config(8)
    scene = O.make_scene(cfg)
    assert torch.equal(scene["centroids"], t(G["centroids"])) and torch.equal(scene["aabbs"], t(G["aabbs"]))
    P = O.make_params(cfg, seed=int(G["seed"]), table_scale=0.3)
    for k in range(cfg["num_fields"]):
        P[f"field.fields.{k}.mlp_base_mlp.layers.1.bias"][0] = -2.5
        for i in range(2):
            P[f"proposal_networks.{i}.fields.{k}.mlp_base.1.layers.1.bias"][0] = -2.0
    batch = {k[2:]: t(v) for k, v in G.items() if k.startswith("B_")}
    return cfg, scene, P, batch


@pytest.fixture(scope="session")
def gold_model_traj():
    return load_golden("model_traj")


def model_traj_setup(G):
    """(cfg, scene, params, batches) of tests/golden/model_traj.npz -- the reference's own 24-iteration training run (K = 3, the
    reference's Optimizers / schedulers / callbacks); parameters are regenerated from the fixture's seed exactly as
    tests/golden/make_golden.py::traj_setup made them"""
    from oracle import nerf_oracle as O

    cfg = O.tiny_config()
    cfg["num_fields"] = 3
    cfg["num_cameras"], cfg["num_videos"] = 24, 2
    for p in [cfg["main"]] + cfg["props"]:
        p["log2_hashmap_size"] = 9
    scene = O.make_scene(cfg)
    scene["centroids"], scene["aabbs"] = t(G["centroids"]), t(G["aabbs"])
    P = O.make_params(cfg, seed=int(G["seed"]), table_scale=0.3)
    for k in range(cfg["num_fields"]):
        P[f"field.fields.{k}.mlp_base_mlp.layers.1.bias"][0] = -2.5
        for i in range(2):
            P[f"proposal_networks.{i}.fields.{k}.mlp_base.1.layers.1.bias"][0] = -2.0
    assert list(P) == [str(k) for k in G["keys"]]
    n = int(G["n_steps"])
    batches = [{k[2:]: (t(G[k][s]).float() if G[k].dtype == np.float16 else t(G[k][s])) for k in G if k.startswith("B_")} for s in range(n)]
    return cfg, scene, P, batches


def traj_param_error(got, ref, init, trim: float = 0.01):
    """distance of two parameter sets after the same training run, per tensor, RELATIVE TO THE DISTANCE THE RUN MOVED THE TENSOR:
    ||got - ref||_2 / ||ref - init||_2 over all entries but the `trim` fraction (at least one entry) with the largest |got - ref|.
    Why trimmed: Adam with eps = 1e-15 normalises every gradient entry, so an entry whose gradient is pure rounding noise in some
    step (a hash row reached only by samples of ~zero weight) still moves by +-lr in that step, with a sign any two fp32 evaluations
    may disagree on: isolated entries differ by 2 lr (observed on MI355X: 1 of 1024 entries by 2.0e-4 = 2 x the first step's lr)
    while the run as a whole is the same run.  The trimmed entries are bounded separately (traj_param_max_diff)."""
    out = {}
    for k, r in ref.items():
        g = got[k].detach().cpu().double().flatten()
        r = r.double().flatten()
        d = (g - r).abs()
        n_trim = max(1, int(trim * d.numel())) if d.numel() > 8 else 0
        if n_trim:
            d = torch.sort(d).values[:-n_trim]
        moved = float((r - init[k].double().flatten()).norm())
        out[k] = float(d.norm()) / max(moved, 1e-30) if moved > 0 else float(d.max())
    return out


def traj_param_max_diff(got, ref):
    """largest entry-wise |got - ref| over all tensors (the entries traj_param_error trims): a few Adam sign flips = a few lr"""
    return max(float((got[k].detach().cpu().double() - r.double()).abs().max()) for k, r in ref.items())


def grad_error_stats(named_grads, ref_grads):
    """Per-tensor max |got - ref| / max|ref| of a set of parameter gradients against the oracle's, as a sorted tensor plus the
    name of the worst one; tensors whose reference gradient is exactly zero must be exactly zero (asserted) and are counted."""
    errs, names, n_zero = [], [], 0
    for name, ref in ref_grads.items():
        got = named_grads.get(name)
        got = torch.zeros_like(ref) if got is None else got.detach().cpu()
        scale = float(ref.abs().max())
        if scale == 0:
            assert float(got.abs().max()) == 0, f"{name}: the reference gradient is exactly zero"
            n_zero += 1
            continue
        errs.append(float(((got - ref).abs() / scale).max()))
        names.append(name)
    e = torch.tensor(errs)
    order = torch.argsort(e)
    return e[order], [names[i] for i in order.tolist()], n_zero


def to_double(x):
    """float tensors of a nested dict / list -> float64 (an fp64 run of the CPU oracle: its rounding noise is negligible)"""
    if torch.is_tensor(x):
        return x.double() if x.is_floating_point() else x
    if isinstance(x, dict):
        return {k: to_double(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(to_double(v) for v in x)
    return x


def assert_grads_within_oracle_noise(named_grads, g32, g64, floor: float = 5e-5, factor: float = 4.0, what: str = "gradients",
                                     cap: float = 1e-2, q90: float = 1e-3):
    """Parameter gradients against the fp32 oracle with a bound that is computed, not guessed: the fp32 oracle's own distance from
    its fp64 run.  ReLU networks amplify one-ulp differences into flipped units / moved samples, so single networks of an fp32 run
    can sit 1e-3 away from the exact gradient while the rest agrees to 1e-6; a flat tolerance is either blind or flaky.  A flipped
    unit perturbs every tensor of its network, so the noise is taken per NETWORK (all tensors of one sub-field of one module:
    `...fields.K.*`): bound(tensor) = min(cap, max(floor, factor * max over its network of max|g32 - g64| / max|g64|)); errors are
    max|got - g32| / max|g32|.  Three guards keep an ill-conditioned oracle network from opening a window a wrong gradient could
    pass through: (1) the computed bound is CAPPED; (2) the 90th percentile of all per-tensor errors must stay below `q90`;
    (3) every tensor of a network whose bound was inflated beyond 10 x floor must be AS CLOSE TO THE EXACT (fp64) GRADIENT AS THE
    REFERENCE'S OWN fp32 RUN IS, in the 2-norm: ||got - g64|| / ||g64|| <= max(10 x floor, factor x the network's largest
    ||g32 - g64|| / ||g64||) -- a flipped unit moves single hash rows by percents (max-norm) but not the tensor as a whole, a
    wrong gradient moves the whole tensor.  Inflated networks are printed.  -> (sorted errors, names, bounds in the same order)"""
    import re

    def network(name):
        m = re.match(r"(.*?fields\.\d+)\.", name)
        return m.group(1) if m else name.rsplit(".", 1)[0]

    noise = {}
    for name, ref in g32.items():
        if float(ref.abs().max()) > 0:
            n = float((ref.double() - g64[name]).abs().max()) / max(float(g64[name].abs().max()), 1e-300)
            noise[network(name)] = max(noise.get(network(name), 0.0), n)
    rows = []
    for name, ref in g32.items():
        got = named_grads.get(name)
        got = torch.zeros_like(ref) if got is None else got.detach().cpu().float()
        scale = float(ref.abs().max())
        if scale == 0:
            assert float(got.abs().max()) == 0, f"{name}: the reference gradient is exactly zero"
            continue
        err = float((got - ref).abs().max()) / scale
        rows.append((err, name, min(cap, max(floor, factor * noise[network(name)]))))
    rows.sort()
    capped = sorted({network(n) for _, n, b in rows if b >= cap})
    if capped:  # (the cap is a ceiling, not a tolerance: say so whenever it is what bounds a network)
        print(f"{what}: the bound of {capped} is the CAP {cap:.0e} (oracle noise x {factor:g} = {[f'{factor * noise[n]:.1e}' for n in capped]})")
    inflated = sorted({network(n) for _, n, b in rows if b > 10 * floor})
    if inflated:
        print(f"{what}: oracle fp32-vs-fp64 noise inflates the bound of {[(n, f'{factor * noise[n]:.1e}') for n in inflated]}")
        rel2 = lambda a, b_: float((a.double() - b_.double()).norm()) / max(float(b_.double().norm()), 1e-300)  # noqa: E731
        noise2 = {}
        for _, n, _b in rows:
            if network(n) in inflated:
                noise2[network(n)] = max(noise2.get(network(n), 0.0), rel2(g32[n], g64[n]))
        bad2 = []
        for _, n, _b in rows:
            if network(n) in inflated:
                e2, b2 = rel2(named_grads[n].detach().cpu(), g64[n]), max(10 * floor, factor * noise2[network(n)])
                if e2 > b2:
                    bad2.append((n, f"2-norm distance from the fp64 gradient {e2:.1e}", f"bound {b2:.1e} (the oracle's own fp32 run: {noise2[network(n)]:.1e})"))
        assert not bad2, f"{what}: {bad2}"
    bad = [(n, f"{e:.1e}", f"bound {b:.1e}") for e, n, b in rows if e > b]
    assert not bad, f"{what}: {bad}"
    if rows:
        e90 = rows[min(len(rows) - 1, int(0.9 * len(rows)))][0]
        assert e90 < q90, f"{what}: 90th percentile of the per-tensor errors {e90:.1e} >= {q90:.1e}"
    return [r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows]


def assert_threshold_depth(depth, depth_ref, weights_ref, ebins_ref, threshold: float = 0.5, what: str = "threshold depth"):
    """DepthRenderer("threshold") is INDEX work (ns/model_components/renderers.py:352-362): the depth is the mid-point of the
    first sample whose inclusive cumulative weight reaches `threshold`.  It must equal the reference's, except on rays whose
    reference cumulative weight sits within fp32 summation rounding of the threshold at the decisive sample -- and there the
    index may move by ONE sample only.  weights_ref [R,S] / ebins_ref [R,S+1]: the reference's (or the oracle's) weights and
    euclidean bin edges of the same rays."""
    depth, depth_ref = torch.as_tensor(depth).detach().cpu().reshape(-1), torch.as_tensor(depth_ref).detach().cpu().reshape(-1)
    w = torch.as_tensor(weights_ref).detach().cpu().float()
    w = w.reshape(w.shape[0], -1)
    eb = torch.as_tensor(ebins_ref).detach().cpu().float()
    R, S = w.shape
    mids = (eb[:, :-1] + eb[:, 1:]) / 2
    cs = torch.cumsum(w, -1)
    ref_idx = torch.clamp(torch.searchsorted(cs, torch.full((R, 1), threshold), side="left"), 0, S - 1)[:, 0]
    mism = (depth - depth_ref).abs() > 1e-5 * depth_ref.abs().clamp_min(1.0)
    n_bad = int(mism.sum())
    if n_bad == 0:
        return 0
    rows = torch.nonzero(mism).flatten()
    got_idx = (mids[rows] - depth[rows, None]).abs().argmin(-1)
    assert bool(((mids[rows, got_idx] - depth[rows]).abs() <= 1e-4 * depth[rows].abs().clamp_min(1.0)).all()), f"{what}: not a sample mid-point"
    step = (got_idx - ref_idx[rows]).abs()
    assert bool((step == 1).all()), f"{what}: index moved by {step.tolist()} samples"
    lo = torch.minimum(got_idx, ref_idx[rows])  # the sample whose cumulative weight decides between the two indices
    margin = (cs[rows, lo] - threshold).abs()
    tol = 16 * S * torch.finfo(torch.float32).eps  # fp32 summation of S weights <= 1
    assert bool((margin <= tol).all()), f"{what}: {n_bad} rays differ, cumulative weight {margin.max():.2e} away from the threshold (> {tol:.1e})"
    return n_bad


def build_hip_model(cfg, scene, P, dev, **conf_overrides):
    """presight_amd's NerfactoNuscMSModel for an oracle-style (cfg, scene, params) triple, parameters loaded through the
    reference's state-dict keys (and their mlp_base aliases)"""
    from presight_amd.model import NerfactoNuscMSModel, NerfactoNuscMSModelConfig

    m = cfg["main"]
    conf = NerfactoNuscMSModelConfig(
        near_plane=cfg["near"], far_plane=cfg["far"], piecewise_sampler_threshold=cfg["thr"], hidden_dim=m["hidden_dim"],
        hidden_dim_color=m["hidden_dim_color"], num_levels=m["num_levels"], base_res=m["base_res"], max_res=m["max_res"],
        log2_hashmap_size=m["log2_hashmap_size"], features_per_level=m["features_per_level"],
        proposal_net_args_list=[dict(features_per_level=p["features_per_level"], log2_hashmap_size=p["log2_hashmap_size"],
                                     num_levels=p["num_levels"], base_res=p["base_res"], max_res=p["max_res"],
                                     hidden_dim=p["hidden_dim"], use_linear=False) for p in cfg["props"]],
        implementation="hip", use_lidar_loss=False, distortion_loss_mult=cfg["distortion_loss_mult"],
        sky_mlp_dims=cfg["sky"]["width"], num_sky_mlp_layers=cfg["sky"]["num_layers"], **conf_overrides)
    model = NerfactoNuscMSModel(conf, num_train_cameras=cfg["num_cameras"], num_train_videos=cfg["num_videos"],
                                dino_to_rgb=scene.get("dino_to_rgb"), centroids=scene["centroids"], aabbs=scene["aabbs"])
    sd = model.state_dict()
    missing = [k for k in P if k not in sd]
    assert not missing, missing
    full = dict(sd)
    for k, v in P.items():
        full[k] = v
        alias = k.replace("mlp_base_grid.", "mlp_base.0.").replace("mlp_base_mlp.", "mlp_base.1.").replace("encoding.hash_table", "mlp_base.0.hash_table")
        if alias in full:
            full[alias] = v
    model.load_state_dict(full)
    return model.to(dev)


def learnable_scene_setup(rays: int = 192, steps: int = 40, test_rays: int = 384, num_cameras: int = 24):
    """The learnable synthetic scene at fixture size, built by the ORACLE (oracle/nerf_oracle.py::teacher_params / teacher_targets;
    presight_amd/synthetic.py is the same construction on the HIP side): tiny K = 1 model, a teacher parameter set, `steps` training
    batches whose targets are the teacher's renders (+ stored jitters), a held-out test batch, and the student's initial parameters.
    -> (cfg, scene, teacher params, batches, test batch, student params)"""
    from oracle import nerf_oracle as O

    cfg = O.tiny_config()
    cfg["num_cameras"] = num_cameras
    for p in [cfg["main"]] + cfg["props"]:
        p["log2_hashmap_size"] = 12
    scene = O.make_scene(cfg)
    Pt = O.teacher_params(cfg, scene, max_res=64)

    def batch(step, n):
        b = O.make_batch(cfg, scene, n, step=500 + step)
        tgt = O.teacher_targets(Pt, cfg, scene, b["ray_indices"], b["video_ids"])
        b.update(rgb=tgt["rgb"], features=tgt["features"], sky=tgt["sky"], accumulation=tgt["accumulation"])
        return b

    return cfg, scene, Pt, [batch(s, rays) for s in range(steps)], batch(9999, test_rays), O.make_params(cfg, seed=3)


def checkpoint_from_fixture(G):
    """tests/golden/checkpoint.npz -> the dict the REFERENCE's Trainer.save_checkpoint wrote (ns/engine/trainer.py:432-460): "step",
    "pipeline" (`_model.`-prefixed tensors), "optimizers" (one torch.optim.Adam.state_dict per parameter group), "schedulers" (one
    ChainedScheduler.state_dict per group), "scalers"; plus the fixture's meta record (parameter names per group, sampler counters)"""
    import collections
    import json

    meta = json.loads(str(G["meta_json"]))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).reshape(a.shape)  # noqa: E731  (0-d buffers / step counts stay 0-d)

    def unjson(x):
        if isinstance(x, dict):
            if set(x) == {"__counter__"}:
                return collections.Counter({int(k): int(v) for k, v in x["__counter__"].items()})
            return {k: unjson(v) for k, v in x.items()}
        if isinstance(x, list):
            return [unjson(v) for v in x]
        return x

    ckpt = {"step": int(meta["step"]), "pipeline": collections.OrderedDict((k, t(G["P::" + k])) for k in meta["pipeline_keys"]),
            "optimizers": {}, "schedulers": unjson(meta["schedulers"]), "scalers": unjson(meta["scalers"])}
    for g, rec in meta["optimizers"].items():
        pgs = unjson(rec["param_groups"])
        for pg in pgs:
            pg["betas"] = tuple(pg["betas"])
        ckpt["optimizers"][g] = {"state": {i: {k: t(G[f"O::{g}::{i}::{k}"]) for k in ("step", "exp_avg", "exp_avg_sq")} for i in rec["state_indices"]},
                                 "param_groups": pgs}
    return ckpt, meta
