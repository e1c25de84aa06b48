"""The C restatement of the integer/index arithmetic (oracle/hashgrid_ref.c) against the reference-generated fixture
and against the torch-CPU oracle.  CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, t
from oracle import nerf_oracle as O


@pytest.fixture(scope="module")
def cref():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle_ref.so"))
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("tag", ["kat", "cfg2small", "prodsmall", "prop0small", "prop1small", "cfg2_full", "prod_full"])
def test_c_hash_indices_bit_exact(cref, gold_hashgrid, tag):
    G = gold_hashgrid
    L, _, _, l2t, _ = [int(v) for v in G[tag + "_meta"]]
    x = np.ascontiguousarray(G[tag + "_x"], np.float32)
    sc = np.ascontiguousarray(G[tag + "_scalings"], np.float32)
    idx = np.zeros((x.shape[0], L, 8), np.int64)
    cref.ref_hash_indices(_ptr(x), _ptr(sc), L, l2t, ctypes.c_int64(x.shape[0]), _ptr(idx))
    assert np.array_equal(idx, G[tag + "_idx"])


@pytest.mark.parametrize("tag", ["kat", "cfg2small", "prodsmall", "prop0small"])
def test_c_encode_and_scatter(cref, gold_hashgrid, tag):
    G = gold_hashgrid
    L, _, _, l2t, F = [int(v) for v in G[tag + "_meta"]]
    x = np.ascontiguousarray(G[tag + "_x"], np.float32)
    sc = np.ascontiguousarray(G[tag + "_scalings"], np.float32)
    table = np.ascontiguousarray(G[tag + "_table"], np.float32)
    out = np.zeros((x.shape[0], L * F), np.float32)
    cref.ref_hash_encode(_ptr(x), _ptr(table), _ptr(sc), L, F, l2t, ctypes.c_int64(x.shape[0]), _ptr(out))
    assert np.array_equal(out, G[tag + "_out"])  # same roundings in the same order as the reference: bit exact
    cot = np.ascontiguousarray(G[tag + "_cot"], np.float32)
    dt = np.zeros_like(table)
    cref.ref_hash_scatter(_ptr(x), _ptr(cot), _ptr(sc), L, F, l2t, ctypes.c_int64(x.shape[0]), _ptr(dt))
    np.testing.assert_allclose(dt, G[tag + "_grad_table"], rtol=1e-4, atol=1e-6)


def test_c_voxel_index_matches_oracle(cref):
    g = torch.Generator().manual_seed(1)
    pts = (torch.rand(5000, 3, generator=g) - 0.5) * 200
    mn = pts.min(0).values - 1.0
    ref = O.voxel_index(pts, 0.4, mn).numpy()
    p = np.ascontiguousarray(pts.numpy(), np.float32)
    mb = np.ascontiguousarray(mn.double().numpy())
    idx = np.zeros((5000, 3), np.int64)
    cref.ref_voxel_index(_ptr(p), ctypes.c_int64(5000), ctypes.c_double(0.4), _ptr(mb), _ptr(idx))
    assert np.array_equal(idx, ref)
