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


def assert_grads_within_oracle_noise(named_grads, g32, g64, floor: float = 5e-5, factor: float = 4.0, what: str = "gradients"):
    """Parameter gradients against the fp32 oracle with a bound that is computed, not guessed: the fp32 oracle's own distance from
    its fp64 run.  ReLU networks amplify one-ulp differences into flipped units / moved samples, so single networks of an fp32 run
    can sit 1e-3 away from the exact gradient while the rest agrees to 1e-6; a flat tolerance is either blind or flaky.  A flipped
    unit perturbs every tensor of its network, so the noise is taken per NETWORK (all tensors of one sub-field of one module:
    `...fields.K.*`): bound(tensor) = max(floor, factor * max over its network of max|g32 - g64| / max|g64|); errors are
    max|got - g32| / max|g32|.  -> (sorted errors, names, bounds in the same order)"""
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
        rows.append((err, name, max(floor, factor * noise[network(name)])))
    rows.sort()
    bad = [(n, f"{e:.1e}", f"bound {b:.1e}") for e, n, b in rows if e > b]
    assert not bad, f"{what}: {bad}"
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
