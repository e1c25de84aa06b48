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
