"""CPU checks of oracle/dual_oracle.py (BASELINE cfg 4; "parity unpinned": the reference has no dynamic field).  What can be
pinned is pinned here: the 4-D hash grid restricted to t = 0 IS the reference-pinned 3-D grid, an independent corner-by-corner
evaluation agrees with the vectorised one (indices bit exact), the position gradient passes a float64 finite-difference check,
and a dynamic branch with zero density reproduces the pinned static model bit for bit."""
import itertools

import numpy as np
import pytest
import torch

from oracle import dual_oracle as D
from oracle import nerf_oracle as O


def _table(L, T, F, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(L * T, F, generator=g) * 2 - 1


def test_hash4_at_time_zero_is_the_pinned_3d_grid():
    L, log2T, F = 4, 9, 2
    sc = O.hash_scalings(L, 16, 128)
    tab = _table(L, 1 << log2T, F)
    g = torch.Generator().manual_seed(1)
    u = torch.rand(500, 3, generator=g)
    e3, i3 = O.hash_encode(u, tab, sc, log2T, return_indices=True)
    e4, i4 = D.hash_encode4(torch.cat([u, torch.zeros(500, 1)], -1), tab, sc, log2T, return_indices=True)
    assert torch.equal(i4[..., :8], i3) and torch.equal(i4[..., 8:], i3)  # t = 0: ceil == floor == 0, the time term hashes to 0
    assert torch.equal(e4, e3)


def test_hash4_against_an_independent_corner_loop():
    """16 corners enumerated one by one in python ints / floats: index = (x ^ y*P1 ^ z*P2 ^ t*P3) mod T + l*T, weight =
    product over the axes of (offset if the corner is the ceil one else 1 - offset)"""
    L, log2T, F = 3, 6, 2
    T = 1 << log2T
    sc = O.hash_scalings(L, 4, 16)
    tab = _table(L, T, F, seed=3).double()
    g = torch.Generator().manual_seed(2)
    x = torch.rand(40, 4, generator=g).double() * 1.3 - 0.15  # also outside [0,1]: warped positions are not clamped
    x[0] = torch.tensor([0.5, 0.25, 0.75, 0.5])  # exact integers on every level: ceil == floor
    e, idx = D.hash_encode4(x, tab, sc.double(), log2T, return_indices=True)
    primes = (1, 2654435761, 805459861, 3674653429)
    for n in range(x.shape[0]):
        for l in range(L):
            s = [float(x[n, a]) * float(sc[l]) for a in range(4)]
            fl, ce = [int(np.floor(v)) for v in s], [int(np.ceil(v)) for v in s]
            off = [s[a] - fl[a] for a in range(4)]
            want = np.zeros(F)
            seen = set()
            for corner in itertools.product((0, 1), repeat=4):  # 1 = ceil
                coord = [ce[a] if corner[a] else fl[a] for a in range(4)]
                h = 0
                for a in range(4):
                    h ^= (coord[a] * primes[a]) & 0xFFFFFFFFFFFFFFFF
                row = h % T + l * T
                seen.add(row)
                w = np.prod([off[a] if corner[a] else 1.0 - off[a] for a in range(4)])
                want += w * tab[row].numpy()
            assert seen == set(int(v) for v in idx[n, l].tolist())
            np.testing.assert_allclose(e[n, l * F:(l + 1) * F].numpy(), want, rtol=1e-12, atol=1e-14)


def test_hash4_position_gradient_finite_differences():
    L, log2T, F = 3, 8, 2
    sc = O.hash_scalings(L, 4, 32).double()
    tab = _table(L, 1 << log2T, F, seed=5).double()
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(12, 4, generator=g).double() * 0.9 + 0.05)
    # keep away from cell faces of every level (the encode is piecewise multilinear: the derivative jumps there)
    frac = (x[:, None, :] * sc.view(1, L, 1)) % 1.0
    x = x[((frac > 0.05) & (frac < 0.95)).all(-1).all(-1)]
    assert x.shape[0] >= 3
    x.requires_grad_(True)
    assert torch.autograd.gradcheck(lambda v: D.hash_encode4(v, tab, sc, log2T), (x,), eps=1e-7, atol=1e-6, rtol=1e-5)
    tab.requires_grad_(True)
    assert torch.autograd.gradcheck(lambda tb: D.hash_encode4(x.detach(), tb, sc, log2T), (tab,), eps=1e-6, atol=1e-7)


def test_zero_dynamic_density_reproduces_the_static_model_bitwise():
    cfg = D.dual_config(tiny=True)
    for p in [cfg["main"]] + cfg["props"]:
        p["log2_hashmap_size"] = 10
    scene = O.make_scene(cfg)
    P = D.make_dual_params(cfg, seed=3, table_scale=0.3)
    P["dynamic_field.mlp_base_mlp.layers.1.bias"][0] = -1e30  # exp -> exactly 0
    batch = O.make_batch(cfg, scene, 64, step=0)
    with torch.no_grad():
        dual = D.dual_model_forward(P, cfg, scene, batch, training=True)
        stat = O.model_forward(P, cfg, scene, batch, training=True)
    assert float(dual["dynamic_density"].abs().max()) == 0.0
    for k in ("rgb", "semantics", "accumulation", "expected_depth", "depth"):
        assert torch.equal(dual[k], stat[k]), k
    # gradients of the static parameters are the static model's as well
    Ld, _, gd = D.dual_train_step(P, cfg, scene, batch)
    Ls, _, gs = O.train_step({k: v for k, v in P.items() if not k.startswith("dynamic_field")}, cfg, scene, batch)
    assert float(Ld["dynamic_reg_loss"]) == 0.0
    for k, v in gs.items():
        assert torch.equal(gd[k], v), k


def test_dual_step_moves_every_dynamic_parameter():
    cfg = D.dual_config(tiny=True, levels=2, feats=2)
    for p in [cfg["main"]] + cfg["props"]:
        p["log2_hashmap_size"] = 10
    scene = O.make_scene(cfg)
    P = D.make_dual_params(cfg, seed=4, table_scale=0.3)
    batch = O.make_batch(cfg, scene, 96, step=1)
    L, out, g = D.dual_train_step(P, cfg, scene, batch)
    assert set(L) == {"rgb_loss", "sky_loss", "semantic_loss", "interlevel_loss", "distortion_loss", "dynamic_reg_loss"}
    assert all(bool(torch.isfinite(v)) for v in L.values())
    for k, v in g.items():
        if k.startswith("dynamic_field"):
            assert float(v.abs().max()) > 0, k  # incl. the flow head: it only gets gradients through the warped encodes
    t = D.ray_times(scene, batch["ray_indices"])
    assert float(t.min()) >= 0 and float(t.max()) <= 1
