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
