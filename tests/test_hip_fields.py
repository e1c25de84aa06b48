"""GPU parity tests of the fused field kernels (points -> encode -> MFMA MLP stack -> outputs, and the full
backward incl. the LDS slice-owner table scatter) against the reference-generated fixtures and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import model_fixture_setup, t
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def F():
    from presight_amd import field_ops

    return field_ops


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(a)
    b = t(b) if isinstance(b, np.ndarray) else b.detach().cpu()
    torch.testing.assert_close(a.to(b.dtype).reshape(b.shape), b, rtol=rtol, atol=atol)


def close_scaled(a, b, rtol=2e-4, atol=2e-5):
    b = t(b) if isinstance(b, np.ndarray) else b.detach().cpu()
    s = float(b.abs().max()) + 1e-20
    close(a.detach().cpu() / s, b / s, rtol=rtol, atol=atol)


def _layers(P, prefix, dev, grad=True):
    out, i = [], 0
    while f"{prefix}.layers.{i}.weight" in P:
        out.append((P[f"{prefix}.layers.{i}.weight"].to(dev).requires_grad_(grad), P[f"{prefix}.layers.{i}.bias"].to(dev).requires_grad_(grad)))
        i += 1
    return out


def test_main_field_golden(F, dev, gold_model):
    """Sub-field 1 of the reference model fixture: density/embedding-derived outputs and every parameter gradient."""
    G = gold_model
    cfg, scene, P, _ = model_fixture_setup(G)
    m = cfg["main"]
    g = F.GridCfg(m["num_levels"], m["features_per_level"], m["log2_hashmap_size"])
    sc = O.hash_scalings(m["num_levels"], m["base_res"], m["max_res"]).to(dev)
    pre = "field.fields.1"
    table = P[f"{pre}.mlp_base_grid.hash_table"].to(dev).requires_grad_(True)
    base, sem, rgb = _layers(P, f"{pre}.mlp_base_mlp", dev), _layers(P, f"{pre}.semantic_head", dev), _layers(P, f"{pre}.rgb_head", dev)
    pos, dirs, app = t(G["F_pos"]).to(dev), t(G["F_dirs"]).to(dev), t(G["F_app"]).to(dev).requires_grad_(True)
    u, sel = F.field_points(scene["aabbs"][1].to(dev), True, pos=pos)
    uo, so = O.normalize_contract(t(G["F_pos"]), scene["aabbs"][1])
    assert torch.equal(sel.cpu() > 0, so)
    close(u, uo, rtol=1e-6, atol=1e-7)
    sigma, c, s = F.main_field(u, sel, dirs, app, 1, table, sc, g, base, sem, rgb)
    close(sigma, G["F_density"][:, 0], rtol=1e-4, atol=1e-6)
    close(c, G["F_rgb"])
    close(s, G["F_sem"])
    scalar = (sigma * t(G["F_cot_density"])[:, 0].to(dev)).sum() + (c * t(G["F_cot_rgb"]).to(dev)).sum() + (s * t(G["F_cot_sem"]).to(dev)).sum()
    params = [table] + [p for wb in base + sem + rgb for p in wb]
    names = ["mlp_base_grid.hash_table"] + [f"mlp_base_mlp.layers.{i}.{k}" for i in range(2) for k in ("weight", "bias")] + \
        [f"semantic_head.layers.{i}.{k}" for i in range(3) for k in ("weight", "bias")] + \
        [f"rgb_head.layers.{i}.{k}" for i in range(3) for k in ("weight", "bias")]
    grads = torch.autograd.grad(scalar, params + [app])
    for n, gr in zip(names, grads):
        close_scaled(gr, G["Fg_" + n])
    # appearance gradient vs the oracle (the reference fixture does not store it)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    appc = t(G["F_app"]).clone().requires_grad_(True)
    dens, emb = O.main_density(Pg, cfg, 1, t(G["F_pos"]), scene["aabbs"][1])
    rgb_o, sem_o = O.main_heads(Pg, cfg, 1, t(G["F_dirs"]), emb, appc)
    so_ = (dens * t(G["F_cot_density"])[:, 0]).sum() + (rgb_o * t(G["F_cot_rgb"])).sum() + (sem_o * t(G["F_cot_sem"])).sum()
    (ga,) = torch.autograd.grad(so_, appc)
    close_scaled(grads[-1], ga)


def test_prop_field_golden(F, dev, gold_model):
    G = gold_model
    cfg, scene, P, _ = model_fixture_setup(G)
    pc = cfg["props"][0]
    g = F.GridCfg(pc["num_levels"], pc["features_per_level"], pc["log2_hashmap_size"])
    sc = O.hash_scalings(pc["num_levels"], pc["base_res"], pc["max_res"]).to(dev)
    pre = "proposal_networks.0.fields.1"
    table = P[f"{pre}.encoding.hash_table"].to(dev).requires_grad_(True)
    layers = _layers(P, f"{pre}.mlp_base.1", dev)
    u, sel = F.field_points(scene["aabbs"][1].to(dev), True, pos=t(G["F_pos"]).to(dev))
    sigma = F.prop_field(u, sel, table, sc, g, layers)
    close(sigma, G["Pp_density"][:, 0], rtol=1e-4, atol=1e-6)
    grads = torch.autograd.grad((sigma * t(G["Pp_cot"])[:, 0].to(dev)).sum(), [table] + [p for wb in layers for p in wb])
    names = ["encoding.hash_table"] + [f"mlp_base.1.layers.{i}.{k}" for i in range(2) for k in ("weight", "bias")]
    for n, gr in zip(names, grads):
        close_scaled(gr, G["Ppg_" + n])


@pytest.mark.parametrize("backward", ["three kernels", "fused"])
@pytest.mark.parametrize("which", ["cfg2", "prod"])
def test_main_field_vs_oracle_ray_batch(F, dev, which, backward, monkeypatch):
    """cfg-2 / production shaped field on a ray batch (positions generated in-kernel from rays + bins); the backward as three
    kernels (one MLP stack each, the training path) and as the single fused kernel"""
    monkeypatch.setenv("PRESIGHT_MAIN_BWD_SPLIT", "1" if backward == "three kernels" else "0")
    # Seeded: the appearance codes / cotangents must not depend on what ran before.  (A generator state exists -- found by a
    # full-suite run -- for which ONE hidden neuron's pre-activation is within rounding of zero at one point: the ReLU mask then
    # differs between the CPU and the GPU arithmetic and that neuron's row of the weight gradient is off by the point's
    # contribution, identically for both backward paths.  That is a property of the comparison, not of a kernel.)
    torch.manual_seed(77)
    cfg = O.default_config()
    if which == "prod":
        cfg["main"].update(num_levels=10, features_per_level=4, log2_hashmap_size=14, max_res=16384)
    else:
        cfg["main"]["log2_hashmap_size"] = 15
    for p in cfg["props"]:
        p["log2_hashmap_size"] = 12
    cfg["num_cameras"] = 60
    P = O.make_params(cfg, seed=9, table_scale=0.2)
    P["field.fields.0.mlp_base_mlp.layers.1.bias"][0] = 1.0
    scene = O.make_scene(cfg)
    m = cfg["main"]
    R, S = 50, 64
    batch = O.make_batch(cfg, scene, R, step=2)
    o, d, _, _ = O.generate_rays(batch["ray_indices"], scene["c2w"], scene["fx"], scene["fy"], scene["cx"], scene["cy"])
    bins = O.spaced_bins(R, S, batch["jitter"][0])
    eb = O.s_to_euclid(bins, torch.full((R, 1), 0.005), torch.full((R, 1), 50.0), 5.0)
    mid = (eb[:, :-1] + eb[:, 1:]) / 2
    pos = (o[:, None] + d[:, None] * mid[..., None]).reshape(-1, 3)
    app = torch.randn(R, 16)
    cots = [torch.rand(R * S), torch.rand(R * S, 3) - 0.5, torch.rand(R * S, 64) - 0.5]
    # oracle
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    appo = app.clone().requires_grad_(True)
    app_s = appo[:, None, :].expand(R, S, 16).reshape(R * S, 16)
    dir_s = d[:, None, :].expand(R, S, 3).reshape(-1, 3)
    dens, emb = O.main_density(Pg, cfg, 0, pos, scene["aabbs"][0])
    rgb_o, sem_o = O.main_heads(Pg, cfg, 0, dir_s, emb, app_s)
    ((dens * cots[0]).sum() + (rgb_o * cots[1]).sum() + (sem_o * cots[2]).sum()).backward()
    # HIP
    g = F.GridCfg(m["num_levels"], m["features_per_level"], m["log2_hashmap_size"])
    sc = O.hash_scalings(m["num_levels"], m["base_res"], m["max_res"]).to(dev)
    pre = "field.fields.0"
    table = P[f"{pre}.mlp_base_grid.hash_table"].to(dev).requires_grad_(True)
    base, sem, rgb = _layers(P, f"{pre}.mlp_base_mlp", dev), _layers(P, f"{pre}.semantic_head", dev), _layers(P, f"{pre}.rgb_head", dev)
    appd = app.to(dev).requires_grad_(True)
    u, sel = F.field_points(scene["aabbs"][0].to(dev), True, origins=o.to(dev), dirs=d.to(dev), ebins=eb.to(dev))
    sigma, c, s = F.main_field(u, sel, d.to(dev), appd, S, table, sc, g, base, sem, rgb)
    close(sigma, dens, rtol=2e-4, atol=1e-6)
    close(c, rgb_o)
    close(s, sem_o)
    scalar = (sigma * cots[0].to(dev)).sum() + (c * cots[1].to(dev)).sum() + (s * cots[2].to(dev)).sum()
    params = [table] + [p for wb in base + sem + rgb for p in wb] + [appd]
    names = [f"{pre}.mlp_base_grid.hash_table"] + [f"{pre}.mlp_base_mlp.layers.{i}.{k}" for i in range(2) for k in ("weight", "bias")] + \
        [f"{pre}.semantic_head.layers.{i}.{k}" for i in range(3) for k in ("weight", "bias")] + \
        [f"{pre}.rgb_head.layers.{i}.{k}" for i in range(3) for k in ("weight", "bias")]
    grads = torch.autograd.grad(scalar, params)
    for n, gr in zip(names, grads[:-1]):
        close_scaled(gr, Pg[n].grad, rtol=5e-4, atol=5e-5)
    close_scaled(grads[-1], appo.grad, rtol=5e-4, atol=5e-5)
    # density-only query (extraction path): heads skipped
    with torch.no_grad():
        s2, _, sem2 = F.main_field(u, sel, None, None, 1, table, sc, g, base, sem, rgb, want_rgb=False, want_sem=True)
    close(s2, dens, rtol=2e-4, atol=1e-6)
    close(sem2, sem_o)


@pytest.mark.parametrize("which", ["cfg2", "prod"])
def test_factored_render_node_vs_oracle(F, dev, which):
    """The training render node of one sub-field (field + get_weights + renderers, field_ops.main_field_render) on its FACTORED path
    (merged linear layers, per-ray semantic output layer and colour term, weights and the semantic branch's
    compositing inside the field kernel) directly against the oracle's unfactored arithmetic: every rendered output and the gradient
    of every parameter, of the appearance codes and of externally supplied d(weights)."""
    torch.manual_seed(78)
    cfg = O.default_config()
    if which == "prod":
        cfg["main"].update(num_levels=10, features_per_level=4, log2_hashmap_size=14, max_res=16384)
    else:
        cfg["main"]["log2_hashmap_size"] = 15
    cfg["num_cameras"] = 60
    P = O.make_params(cfg, seed=10, table_scale=0.2)
    P["field.fields.0.mlp_base_mlp.layers.1.bias"][0] = -1.0  # unsaturated rays: every output carries gradient
    scene = O.make_scene(cfg)
    m = cfg["main"]
    R, S = 70, 64
    batch = O.make_batch(cfg, scene, R, step=2)
    o, d, _, _ = O.generate_rays(batch["ray_indices"], scene["c2w"], scene["fx"], scene["fy"], scene["cx"], scene["cy"])
    bins = O.spaced_bins(R, S, batch["jitter"][0])
    eb = O.s_to_euclid(bins, torch.full((R, 1), 0.005), torch.full((R, 1), 50.0), 5.0)
    mid = (eb[:, :-1] + eb[:, 1:]) / 2
    pos = (o[:, None] + d[:, None] * mid[..., None]).reshape(-1, 3)
    app = torch.randn(R, 16)
    cots = dict(rgb=torch.rand(R, 3) - 0.5, acc=torch.rand(R, 1) - 0.5, exp=torch.rand(R, 1) - 0.5, sem=torch.rand(R, 64) - 0.5,
                w=(torch.rand(R, S) - 0.5) * 0.1)

    def scalar(rgb, acc, expd, sem, w):
        return ((rgb * cots["rgb"].to(rgb.device)).sum() + (acc * cots["acc"].to(rgb.device)).sum() + (expd * cots["exp"].to(rgb.device)).sum()
                + (sem * cots["sem"].to(rgb.device)).sum() + (w * cots["w"].to(rgb.device)).sum())

    # oracle (per-sample heads, then get_weights and the renderers)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    appo = app.clone().requires_grad_(True)
    app_s = appo[:, None, :].expand(R, S, 16).reshape(R * S, 16)
    dir_s = d[:, None, :].expand(R, S, 3).reshape(-1, 3)
    dens, emb = O.main_density(Pg, cfg, 0, pos, scene["aabbs"][0])
    rgb_s, sem_s = O.main_heads(Pg, cfg, 0, dir_s, emb, app_s)
    w_o = O.weights_from_density(eb[:, 1:] - eb[:, :-1], dens.reshape(R, S))
    rgb_o = (w_o[..., None] * rgb_s.reshape(R, S, 3)).sum(1)
    sem_o = (w_o[..., None] * sem_s.reshape(R, S, 64)).sum(1)
    acc_o = w_o.sum(-1, keepdim=True)
    exp_o = O.expected_depth(w_o, mid)
    dep_o = O.threshold_depth(w_o, mid)
    scalar(rgb_o, acc_o, exp_o, sem_o, w_o).backward()
    # HIP, factored node
    assert F.FACTORED
    g = F.GridCfg(m["num_levels"], m["features_per_level"], m["log2_hashmap_size"])
    sc = O.hash_scalings(m["num_levels"], m["base_res"], m["max_res"]).to(dev)
    pre = "field.fields.0"
    table = P[f"{pre}.mlp_base_grid.hash_table"].to(dev).requires_grad_(True)
    base, sem, rgb = _layers(P, f"{pre}.mlp_base_mlp", dev), _layers(P, f"{pre}.semantic_head", dev), _layers(P, f"{pre}.rgb_head", dev)
    assert F.factored_supported(base, sem, rgb, S)
    appd = app.to(dev).requires_grad_(True)
    u, sel = F.field_points(scene["aabbs"][0].to(dev), True, origins=o.to(dev), dirs=d.to(dev), ebins=eb.to(dev))
    rgb_h, acc_h, dep_h, exp_h, sem_h, w_h = F.main_field_render(u, sel, d.to(dev), appd, eb.to(dev), table, sc, g, base, sem, rgb)
    close(w_h, w_o, rtol=2e-4, atol=1e-7)
    close(rgb_h, rgb_o)
    close(acc_h, acc_o)
    close(exp_h, exp_o)
    close(sem_h, sem_o)
    from conftest import assert_threshold_depth
    assert_threshold_depth(dep_h, dep_o, w_o.detach(), eb, what="factored threshold depth")
    params = [table] + [p for wb in base + sem + rgb for p in wb] + [appd]
    names = [f"{pre}.mlp_base_grid.hash_table"] + [f"{pre}.mlp_base_mlp.layers.{i}.{k}" for i in range(2) for k in ("weight", "bias")] + \
        [f"{pre}.semantic_head.layers.{i}.{k}" for i in range(3) for k in ("weight", "bias")] + \
        [f"{pre}.rgb_head.layers.{i}.{k}" for i in range(3) for k in ("weight", "bias")]
    grads = torch.autograd.grad(scalar(rgb_h, acc_h, exp_h, sem_h, w_h), params)
    for n, gr in zip(names, grads[:-1]):
        close_scaled(gr, Pg[n].grad, rtol=5e-4, atol=5e-5)
    close_scaled(grads[-1], appo.grad, rtol=5e-4, atol=5e-5)


def test_prop_field_vs_oracle_full_size_table(F, dev):
    cfg = O.default_config()
    P = {}
    gen = torch.Generator().manual_seed(21)
    pc = cfg["props"][1]
    T = 1 << pc["log2_hashmap_size"]
    table = (torch.rand(T * pc["num_levels"], 1, generator=gen) * 2 - 1) * 0.3
    layers = [O._linear_init(gen, 64, 8), O._linear_init(gen, 1, 64)]
    aabb = torch.tensor([[-1.0, -1.0, -0.2], [1.0, 1.0, 0.4]])
    N = 20011
    pos = (torch.rand(N, 3, generator=gen) - 0.5) * 6
    P["proposal_networks.1.fields.0.encoding.hash_table"] = table.clone().requires_grad_(True)
    for i, (W, b) in enumerate(layers):
        P[f"proposal_networks.1.fields.0.mlp_base.1.layers.{i}.weight"] = W.clone().requires_grad_(True)
        P[f"proposal_networks.1.fields.0.mlp_base.1.layers.{i}.bias"] = b.clone().requires_grad_(True)
    ref = O.prop_density(P, cfg, 1, 0, pos, aabb)
    cot = torch.rand(N, generator=gen)
    (ref * cot).sum().backward()
    g = F.GridCfg(pc["num_levels"], 1, pc["log2_hashmap_size"])
    sc = O.hash_scalings(pc["num_levels"], pc["base_res"], pc["max_res"]).to(dev)
    td = table.to(dev).requires_grad_(True)
    ld = [(W.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)) for W, b in layers]
    u, sel = F.field_points(aabb.to(dev), True, pos=pos.to(dev))
    sigma = F.prop_field(u, sel, td, sc, g, ld)
    close(sigma, ref, rtol=2e-4, atol=1e-6)
    grads = torch.autograd.grad((sigma * cot.to(dev)).sum(), [td] + [p for wb in ld for p in wb])
    close_scaled(grads[0], P["proposal_networks.1.fields.0.encoding.hash_table"].grad, rtol=5e-4, atol=5e-5)
    for i in range(2):
        close_scaled(grads[1 + 2 * i], P[f"proposal_networks.1.fields.0.mlp_base.1.layers.{i}.weight"].grad, rtol=5e-4, atol=5e-5)
        close_scaled(grads[2 + 2 * i], P[f"proposal_networks.1.fields.0.mlp_base.1.layers.{i}.bias"].grad, rtol=5e-4, atol=5e-5)
    # checksum-of-checksums: the table gradient of sum(features) is the total interpolation weight = N*L exactly
    feat_ones = torch.ones(pc["num_levels"], N, 1, device=dev)
    from presight_amd.field_ops import _scatter
    dt = _scatter(u, feat_ones, sc, g, tuple(td.shape))
    torch.testing.assert_close(dt.sum().cpu(), torch.tensor(float(N * pc["num_levels"])), rtol=1e-4, atol=0)


@pytest.mark.parametrize("L,nf,l2t,mx,N", [(16, 2, 19, 2048, 100003), (8, 1, 20, 4096, 70001), (10, 4, 14, 16384, 5000),
                                           (2, 2, 9, 64, 700), (4, 1, 12, 128, 1)])
def test_table_gradient_binned_vs_owner_vs_oracle(F, dev, L, nf, l2t, mx, N):
    """Both table-backward implementations against the C-free oracle scatter (autograd of O.hash_encode), plus the
    binned path's bit-reproducibility (integer accumulation)."""
    from presight_amd import field_ops

    gen = torch.Generator().manual_seed(L * 7 + nf)
    g = F.GridCfg(L, nf, l2t)
    sc = O.hash_scalings(L, 16, mx)
    u = torch.rand(N, 3, generator=gen)
    u[: min(N, 3)] = 0.0
    # ray-coherent cluster: many points in the same coarse cells (contended rows)
    if N > 100:
        u[3:60] = u[3:4] + torch.rand(57, 3, generator=gen) * 1e-3
    dfeat = (torch.randn(L, N, nf, generator=gen) * torch.logspace(-6, 2, L).view(L, 1, 1))  # wide dynamic range
    table = torch.zeros((1 << l2t) * L, nf, requires_grad=True)
    enc = O.hash_encode(u, table, sc, l2t)  # [N, L*nf]
    cot = dfeat.permute(1, 0, 2).reshape(N, L * nf)
    (ref,) = torch.autograd.grad((enc * cot).sum(), table)
    outs = {}
    for impl in ("binned", "owner"):
        field_ops.SCATTER_IMPL = impl
        outs[impl] = field_ops._scatter(u.to(dev), dfeat.to(dev).contiguous(), sc.to(dev), g, tuple(table.shape)).cpu()
    field_ops.SCATTER_IMPL = "binned"
    again = field_ops._scatter(u.to(dev), dfeat.to(dev).contiguous(), sc.to(dev), g, tuple(table.shape)).cpu()
    assert torch.equal(again, outs["binned"])  # integer accumulation: bit-reproducible
    T = 1 << l2t
    for impl, got in outs.items():
        for l in range(L):  # per level: the fixture spans 8 orders of magnitude across levels
            a, b = got[l * T:(l + 1) * T], ref[l * T:(l + 1) * T]
            s = float(b.abs().max()) + 1e-30
            torch.testing.assert_close(a / s, b / s, rtol=2e-4, atol=2e-6, msg=lambda m: f"{impl} level {l}: {m}")


def test_table_gradient_binned_split_pairs_and_accumulate(F, dev):
    """Points whose x cell straddles a slice boundary of the binned scatter (floor and ceil corner hash into different
    slices -> split records, more than the staging area of a bin workgroup holds -> overflow path), and accumulate=1
    (gradient ADDED into a pre-filled buffer, the mode the trainer's flat gradient buffer uses)."""
    from presight_amd import field_ops
    from presight_amd._lib import check, lib

    L, nf, l2t, mx, N = 10, 4, 14, 16384, 3000
    gen = torch.Generator().manual_seed(5)
    g = F.GridCfg(L, nf, l2t)
    sc = O.hash_scalings(L, 16, mx)
    u = torch.rand(N, 3, generator=gen)
    # finest level: scaled x in (4095, 4096) -> fx = 4095, cx = 4096: the xor reaches bit 12 = above every slice size
    u[:2048, 0] = (4095.0 + 0.05 + 0.9 * torch.rand(2048, generator=gen)) / float(sc[-1])
    dfeat = torch.randn(L, N, nf, generator=gen)
    table = torch.zeros((1 << l2t) * L, nf, requires_grad=True)
    enc = O.hash_encode(u, table, sc, l2t)
    (ref,) = torch.autograd.grad((enc * dfeat.permute(1, 0, 2).reshape(N, L * nf)).sum(), table)
    got = field_ops._scatter(u.to(dev), dfeat.to(dev).contiguous(), sc.to(dev), g, tuple(table.shape)).cpu()
    s = float(ref.abs().max())
    torch.testing.assert_close(got / s, ref / s, rtol=2e-4, atol=2e-6)
    # record counts taken in the forward encode (upper bounds, incl. the split pairs) instead of the backward's own pass
    td = torch.zeros(table.shape, device=dev)
    feat, counts = field_ops._encode(u.to(dev), td, sc.to(dev), g, count=True)
    assert counts is not None and int(counts.sum()) >= 4 * N * L
    got_c = field_ops._scatter(u.to(dev), dfeat.to(dev).contiguous(), sc.to(dev), g, tuple(table.shape), counts=counts).cpu()
    assert torch.equal(got_c, got)
    feat0, none = field_ops._encode(u.to(dev), td, sc.to(dev), g)
    assert none is None and torch.equal(feat0, feat)
    base = torch.randn(table.shape, generator=gen).to(dev)
    sink = base.clone()
    assert field_ops._scatter(u.to(dev), dfeat.to(dev).contiguous(), sc.to(dev), g, tuple(table.shape), sink=sink) is None
    torch.testing.assert_close(sink.cpu(), base.cpu() + got, rtol=0, atol=1e-6 * s)
    # few points per table (production shape: 16 sub-fields): the accumulate kernel flushes only the rows it touched
    few = 40
    ref_few = field_ops._scatter(u[:few].to(dev), dfeat[:, :few].to(dev).contiguous(), sc.to(dev), g, tuple(table.shape)).cpu()
    sink2 = base.clone()
    field_ops._scatter(u[:few].to(dev), dfeat[:, :few].to(dev).contiguous(), sc.to(dev), g, tuple(table.shape), sink=sink2)
    torch.testing.assert_close(sink2.cpu(), base.cpu() + ref_few, rtol=0, atol=1e-6 * float(ref_few.abs().max()))
    assert int((sink2 != base).sum()) <= few * L * 8 * nf  # untouched rows were left alone


def test_field_level_empty_and_ragged_inputs(F, dev):
    """N = 0 (a sub-field that saw no sample) and N not a multiple of any tile size (16-point MFMA blocks, 64-lane waves,
    128/512-point encode / bin chunks): forward finite, backward runs, gradients match the oracle on the ragged case."""
    g = F.GridCfg(2, 1, 9)
    sc = O.hash_scalings(2, 16, 64).to(dev)
    gen = torch.Generator().manual_seed(1)
    table = (torch.rand((1 << 9) * 2, 1, generator=gen) * 0.2 - 0.1).to(dev).requires_grad_(True)
    mk = lambda o, i: (torch.randn(o, i, generator=gen) * 0.3).to(dev).requires_grad_(True)  # noqa: E731
    vb = lambda o: (torch.randn(o, generator=gen) * 0.1).to(dev).requires_grad_(True)  # noqa: E731
    layers = [(mk(32, 2), vb(32)), (mk(1, 32), vb(1))]
    for N in (0, 1, 531):
        u = torch.rand(N, 3, generator=gen).to(dev)
        sel = torch.ones(N, device=dev)
        sigma = F.prop_field(u, sel, table, sc, g, layers)
        assert sigma.shape == (N,) and bool(torch.isfinite(sigma).all())
        if N == 0:
            continue
        grads = torch.autograd.grad(sigma.sum(), [table] + [t for wb in layers for t in wb])
        # oracle: hash encode -> Linear/ReLU/Linear -> trunc_exp
        tc = table.detach().cpu().requires_grad_(True)
        lc = [(W.detach().cpu().requires_grad_(True), b.detach().cpu().requires_grad_(True)) for W, b in layers]
        h = O.hash_encode(u.cpu(), tc, sc.cpu(), 9)
        ref = torch.exp(O.mlp_forward(h, lc)[:, 0])
        torch.testing.assert_close(sigma.detach().cpu(), ref.detach(), rtol=2e-4, atol=1e-6)
        gref = torch.autograd.grad(ref.sum(), [tc] + [t for wb in lc for t in wb])
        for a, b in zip(grads, gref):
            s = float(b.abs().max()) + 1e-30
            torch.testing.assert_close(a.cpu() / s, b / s, rtol=1e-3, atol=2e-5)


def test_table_gradient_non_finite_input_gives_nan_not_garbage(F, dev):
    """A NaN / inf in d(features) must surface as NaN in that level's table gradient (torch's index_add would propagate it);
    the fixed-point scatter used to convert it to int64 (undefined).  Other levels stay finite."""
    from presight_amd._lib import lib

    g = F.GridCfg(4, 2, 12)
    gen = torch.Generator().manual_seed(0)
    N = 5000
    u = torch.rand(N, 3, generator=gen).to(dev)
    sc = torch.tensor([16.0, 32.0, 64.0, 128.0], device=dev)
    for bad in (float("nan"), float("inf")):
        dfeat = torch.randn(4, N, 2, generator=gen).to(dev)
        dfeat[2, 123, 1] = bad
        dt = F._scatter(u, dfeat, sc, g, (4 << 12, 2))
        per_level = dt.view(4, 1 << 12, 2)
        assert bool(torch.isnan(per_level[2]).any())
        assert bool(torch.isfinite(per_level[[0, 1, 3]]).all())


@pytest.mark.parametrize("L,nf,l2t,mx,N,ready", [(16, 2, 19, 2048, 150001, False), (8, 1, 20, 4096, 120001, True), (8, 1, 20, 4096, 9001, False),
                                                 (2, 2, 13, 64, 30000, False)])
def test_table_gradient_level0_histogram_equals_records(F, dev, L, nf, l2t, mx, N, ready):
    """The coarsest level summed per cell corner into the dense int64 histogram (csrc/encode.hip level0_hist_kernel: what a
    single-table backward does) against the same backward with EVERY level written as records (phase bit 2, what the sparse exchange
    asks for): bit-equal -- same fixed-point terms, integer sums.  With ray-coherent runs (consecutive points in one cell: the
    register merge), points with zero gradient, points on exact cell boundaries, points OUTSIDE the unit cube (no cell of the
    histogram: they travel as records in the level-0 streams) and -- `ready` -- the per-level absmax handed in by the caller (the
    proposal fields' backward publishes it), plus accumulate = 1 and the forward's record counts.  Against the oracle as well."""
    from presight_amd import field_ops
    from presight_amd._lib import check, lib

    gen = torch.Generator().manual_seed(L * 13 + nf + N)
    g = F.GridCfg(L, nf, l2t)
    sc = O.hash_scalings(L, 16, mx)
    u = torch.rand(N, 3, generator=gen)
    run = torch.rand(N // 64 + 1, 3, generator=gen).repeat_interleave(64, 0)[:N]  # 64 consecutive points march through a few cells
    u[: N // 2] = (run + torch.arange(N).view(-1, 1).remainder(64) * torch.tensor([[0.004, 0.001, 0.0005]]))[: N // 2].clamp(0, 1)
    u[5] = torch.tensor([0.0, 0.0, 0.0])
    u[6] = torch.tensor([1.0, 1.0, 1.0])
    u[7] = torch.tensor([0.5, 0.25, 1.0])                 # exact cell boundaries at level 0 (scale 16)
    u[100:140] = torch.rand(40, 3, generator=gen) * 1.4          # beyond 1 on some axes: no cell of the histogram, record path of its kernel
    dfeat = (torch.randn(L, N, nf, generator=gen) * torch.logspace(-3, 1, L).view(L, 1, 1))
    dfeat[:, 200:400] = 0.0                                # no gradient: nothing is emitted
    table = torch.zeros((1 << l2t) * L, nf, requires_grad=True)
    enc = O.hash_encode(u, table, sc, l2t)
    (ref,) = torch.autograd.grad((enc * dfeat.permute(1, 0, 2).reshape(N, L * nf)).sum(), table)
    ud, dd, scd = u.to(dev), dfeat.to(dev).contiguous(), sc.to(dev)
    nbytes = lib().ps_grid_scatter_workspace(L, nf, l2t, N)
    s = torch.cuda.current_stream().cuda_stream

    def run_scatter(phases, accumulate=0, counts=None, base=None):
        ws = torch.zeros(nbytes + 4096, dtype=torch.uint8, device=dev)
        if ready:  # per-level max |d(feature)| bits in the first words of the workspace, as ps_prop_field_bwd leaves them
            ws[: 4 * L].view(torch.int32).copy_(dd.abs().amax(dim=(1, 2)).view(torch.int32))
        out = torch.full(table.shape, float("nan"), device=dev) if base is None else base.clone()
        for ph in phases:
            check(lib().ps_grid_scatter_binned_part(ud.data_ptr(), dd.data_ptr(), scd.data_ptr(), L, nf, l2t, N, N * nf, out.data_ptr(), accumulate,
                                                    None if counts is None else counts.data_ptr(), int(ready), ws.data_ptr(), ph, 0, -1, s), "scatter")
        # records in the level-0 streams: stream ends (cursors) - stream starts of the level's slices (workspace layout: csrc/encode.hip)
        n_sl = lib().ps_grid_scatter_slices(nf, l2t)
        words = ws[4096: 4096 + 12 * L * n_sl].view(torch.int32)
        level0_records = int((words[:n_sl] - words[2 * L * n_sl: 2 * L * n_sl + n_sl]).sum())
        return out, level0_records

    def run(*a):
        return run_scatter(*a)[0]

    hist, n_hist = run_scatter([3 | 8])        # level 0 through the histogram (phase bit 3)
    recs, n_recs = run_scatter([1 | 4, 2 | 4])  # every level as records
    assert n_recs > 3 * (N - 250) and n_hist <= 8 * 40, (n_hist, n_recs)  # the histogram path ran: only the points outside the cube left records
    assert torch.equal(hist, recs)
    assert torch.equal(run([1 | 8, 2 | 8]), hist)  # the two phases as separate calls (how a gradient exchanged in pieces runs)
    T = 1 << l2t
    for l in range(L):
        a, b = hist[l * T:(l + 1) * T].cpu(), ref[l * T:(l + 1) * T]
        sl = float(b.abs().max()) + 1e-30
        torch.testing.assert_close(a / sl, b / sl, rtol=2e-4, atol=2e-6, msg=lambda m: f"level {l}: {m}")
    # record counts from the forward encode + accumulate = 1 into a pre-filled buffer
    _, counts = field_ops._encode(ud, torch.zeros(table.shape, device=dev), scd, g, count=True)
    base = torch.randn(table.shape, generator=gen).to(dev)
    assert torch.equal(run([3 | 8], 1, counts, base), run([1 | 4, 2 | 4], 1, counts, base))
    assert torch.equal(run([3], 1, counts, base), run([3 | 8], 1, counts, base))  # (and the default: records, no bit)


# (which accumulate kernel a shape runs -- csrc/encode.hip scatter_binned_impl: full 128-KiB slices with < 128 k records per item on
#  average take accumulate_adam_kernel (every load of an item up front, slices without records skip the accumulators): the first
#  three; many records per item, (2, 2, 13, ...), or slices smaller than 128 KiB, (2, 2, 9, ...), stay on accumulate_kernel)
@pytest.mark.parametrize("L,nf,l2t,mx,N", [(16, 2, 19, 2048, 60001), (8, 1, 20, 4096, 30001), (10, 4, 14, 16384, 5000), (2, 2, 13, 64, 80000),
                                           (2, 2, 9, 64, 700)])
def test_table_backward_with_the_adam_step_inside_equals_scatter_then_adam(F, dev, L, nf, l2t, mx, N):
    """ps_grid_scatter_binned_adam (the table backward that applies the optimizer's element update to every slice it finishes: what
    a single-process trainer runs) against the two-call sequence it replaces -- ps_grid_scatter_binned into a zeroed gradient, then
    ps_adam_step_ranges over the table: parameters and both moments bit-equal, the gradient buffer untouched; every entry is updated
    (weight decay / moment decay reach slices without records), and a level whose d(features) hold a NaN turns its entries NaN in
    both paths (torch's index_add of a NaN followed by Adam)."""
    import ctypes

    from presight_amd._lib import check, lib

    gen = torch.Generator().manual_seed(L * 11 + nf)
    sc = O.hash_scalings(L, 16, mx).to(dev)
    u = torch.rand(N, 3, generator=gen)
    u[: N // 2] = u[:1] + torch.rand(N // 2, 3, generator=gen) * 0.05  # half of the points in one corner: most slices stay empty
    u = u.clamp(0, 1).to(dev)
    dfeat = (torch.randn(L, N, nf, generator=gen) * torch.logspace(-4, 1, L).view(L, 1, 1)).to(dev).contiguous()
    dfeat[L - 1, 7, 0] = float("nan")
    n = (1 << l2t) * L * nf
    p0 = ((torch.rand(n, generator=gen) - 0.5) * 2e-4).to(dev)
    m0 = (torch.randn(n, generator=gen) * 1e-3).to(dev)
    v0 = (torch.rand(n, generator=gen) * 1e-6).to(dev)
    ws = torch.empty(lib().ps_grid_scatter_workspace(L, nf, l2t, N) + 4096, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    lr, b1, b2, eps, wd, gs, step = 3e-3, 0.9, 0.999, 1e-15, 1e-5, 1.0, 7
    # the two-call sequence
    pa, ma, va, ga = p0.clone(), m0.clone(), v0.clone(), torch.zeros(n, device=dev)
    check(lib().ps_grid_scatter_binned(u.data_ptr(), dfeat.data_ptr(), sc.data_ptr(), L, nf, l2t, N, N * nf, ga.data_ptr(), 2, None, 0,
                                       ws.data_ptr(), s), "scatter")
    starts, counts = (ctypes.c_int64 * 1)(0), (ctypes.c_int64 * 1)(n)
    steps, groups = (ctypes.c_int * 1)(step), (ctypes.c_int * 1)(-1)
    check(lib().ps_adam_step_ranges(pa.data_ptr(), ga.data_ptr(), ma.data_ptr(), va.data_ptr(), 1, starts, counts, steps, groups, None, None, 0,
                                    lr, b1, b2, eps, wd, gs, s), "adam")
    # the fused call
    pb, mb, vb, gb = p0.clone(), m0.clone(), v0.clone(), torch.zeros(n, device=dev)
    check(lib().ps_grid_scatter_binned_adam(u.data_ptr(), dfeat.data_ptr(), sc.data_ptr(), L, nf, l2t, N, N * nf, gb.data_ptr(), None, 0,
                                            ws.data_ptr(), 3, 0, -1, gb.data_ptr(), pb.data_ptr(), mb.data_ptr(), vb.data_ptr(),
                                            lr, b1, b2, eps, wd, gs, step, s), "scatter+adam")
    torch.cuda.synchronize()
    assert float(gb.abs().max()) == 0.0  # the gradient is never written
    T = (1 << l2t) * nf
    assert bool(torch.isnan(pa[(L - 1) * T:]).all()) and bool(torch.isfinite(pa[:(L - 1) * T]).all())
    for a, b in ((pa, pb), (ma, mb), (va, vb)):
        assert torch.equal(torch.isnan(a), torch.isnan(b))
        assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    assert not torch.equal(pa[:T], p0[:T]) and bool((ma[:(L - 1) * T] != m0[:(L - 1) * T]).all())  # every entry moved (decay), records or not
