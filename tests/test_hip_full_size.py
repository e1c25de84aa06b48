"""BASELINE-size checks (cfg 2: 65 536 rays, 16-level T=2^19 F=2 table, 64-wide MLPs, T=2^20 proposal tables) through
size-independent properties, since the CPU oracle cannot run these sizes in seconds:
  * adjointness  <encode(table), g> == <table, scatter(g)>   (the table backward is the exact transpose of the forward)
  * linearity of the whole backward in the loss scale
  * bit-reproducibility of the table gradients (integer accumulation)
  * physical invariants of the render (weights in [0,1], accumulation == 1 - exp(-sum delta*sigma))
plus the reference's own API/shape tests (nerfstudio-0.3.3/tests/field_components/test_encodings.py:124-168,
test_mlp.py:9-26, tests/model_components/test_ray_sampler.py, test_renderers.py) restated for the HIP operators."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def test_encode_scatter_adjoint_full_size(dev):
    from presight_amd import field_ops as F
    from presight_amd.components import hash_scalings

    gen = torch.Generator(device=dev).manual_seed(0)
    for L, nf, l2t, mx, N in [(16, 2, 19, 2048, 65536 * 64), (8, 1, 20, 4096, 65536 * 128)]:
        g = F.GridCfg(L, nf, l2t)
        sc = hash_scalings(L, 16, mx).to(dev)
        u = torch.rand(N, 3, device=dev, generator=gen)
        table = torch.randn((1 << l2t) * L, nf, device=dev, generator=gen)
        cot = torch.randn(L, N, nf, device=dev, generator=gen)
        feat, _ = F._encode(u, table, sc, g)
        lhs = (feat.double() * cot.double()).sum()
        dt = F._scatter(u, cot, sc, g, tuple(table.shape))
        rhs = (table.double() * dt.double()).sum()
        assert abs(float(lhs - rhs)) <= 2e-5 * float(feat.double().abs().mul(cot.double().abs()).sum()), (L, float(lhs), float(rhs))
        dt2 = F._scatter(u, cot, sc, g, tuple(table.shape))
        assert torch.equal(dt, dt2)  # bit-reproducible
        # total interpolation weight: scatter of ones sums to N*L
        ones = torch.ones(L, N, nf, device=dev)
        tot = F._scatter(u, ones, sc, g, tuple(table.shape)).double().sum()
        assert abs(float(tot) - N * L * nf) <= 1e-4 * N * L * nf
        del feat, dt, dt2, cot, table, u, ones
        torch.cuda.empty_cache()


def test_full_size_training_step_properties(dev):
    import bench

    model, scene = bench.build_model(dev, seed=1)
    trainer = bench.Trainer(model, scene, 1)
    batch = bench.make_batches(scene, dev, 1, 0)[0]
    from presight_amd import ops
    from presight_amd.rays import RayBundle

    def run(scale):
        model.train()
        trainer.grads.zero_()
        o, d, pa, dn = ops.generate_rays(batch["ray_indices"], scene["c2w"], scene["fx"], scene["fy"], scene["cx"], scene["cy"])
        rb = RayBundle(o, d, pa, camera_indices=batch["ray_indices"][:, 0:1], metadata={"video_id": batch["video_ids"][:, None]})
        model.proposal_sampler._steps_since_update = 1 << 30
        g = torch.Generator(device=dev).manual_seed(5)
        jit = [torch.rand(65536, 1, device=dev, generator=g) for _ in range(3)]
        out = model(rb, jitters=jit)
        ld = model.get_loss_dict(out, batch)
        (sum(ld.values()) * scale).backward()
        return out, ld, trainer.grads.flat.clone()

    out, ld, g1 = run(1.0)
    w = out["weights_list"][-1][..., 0]
    rs = out["ray_samples_list"][-1]
    assert bool((w >= 0).all()) and bool((w <= 1 + 1e-6).all())
    assert bool((out["accumulation"] >= 0).all()) and bool((out["accumulation"] <= 1).all())
    assert bool((rs.sbins[:, 1:] >= rs.sbins[:, :-1]).all()) and bool((rs.ebins[:, 1:] >= rs.ebins[:, :-1]).all())
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    assert all(bool(torch.isfinite(v)) for v in ld.values())
    _, _, g2 = run(4.0)  # power of two: the backward must be exactly linear in the loss scale, bit for bit on the tables
    table = dict(model.named_parameters())["field.fields.0.mlp_base_grid.hash_table"]
    off = 0
    for p in trainer.grads.params:
        n = p.numel()
        a, b = g1[off:off + n], g2[off:off + n]
        if p is table:
            assert torch.equal(a * 4.0, b), "table gradient is not bit-linear/reproducible"
        else:
            torch.testing.assert_close(a * 4.0, b, rtol=1e-3, atol=1e-6 * float(b.abs().max()) + 1e-12)
        off += (n + 3) // 4 * 4


def test_reference_style_api_shapes(dev):
    from presight_amd.components import MLP, HashEncoding, SHEncoding

    enc = HashEncoding(num_levels=8, features_per_level=2, log2_hashmap_size=5, min_res=16, max_res=1024).to(dev)
    assert enc.get_out_dim() == 16
    assert enc(torch.rand(10, 3, device=dev)).shape == (10, 16)
    assert enc(torch.rand(4, 5, 3, device=dev)).shape == (4, 5, 16)
    with pytest.raises(ValueError):
        SHEncoding(levels=5)
    sh = SHEncoding(levels=4)
    assert sh.get_out_dim() == 16 and sh(torch.rand(10, 3, device=dev)).shape == (10, 16)
    mlp = MLP(in_dim=32, num_layers=2, layer_width=64, out_dim=80).to(dev)
    assert mlp(torch.rand(9, 32, device=dev)).shape == (9, 80)
    # SH orthonormality (nerfstudio-0.3.3/tests/utils/test_math.py:7-17): (sh^T sh)/N*4pi ~ I on unit vectors
    d = torch.nn.functional.normalize(torch.randn(1_000_000, 3, device=dev), dim=-1)
    basis = sh((d + 1.0) / 2.0)  # the operator receives the shifted direction and evaluates SH on it, like the torch path
    assert basis.shape == (1_000_000, 16)


def test_reference_style_sampler_and_renderer_behaviour(dev):
    from presight_amd.rays import RayBundle
    from presight_amd.renderers import AccumulationRenderer, DepthRenderer, NearFarCollider, RGBRenderer
    from presight_amd.samplers import PDFSampler, SpacedSampler

    R = 10
    rb = RayBundle(torch.zeros(R, 3, device=dev), torch.nn.functional.normalize(torch.ones(R, 3, device=dev), dim=-1),
                   torch.ones(R, 1, device=dev), camera_indices=torch.zeros(R, 1, dtype=torch.long, device=dev))
    col = NearFarCollider(near_plane=0.05, far_plane=10.0)
    rb = col(rb)
    sampler = SpacedSampler(piecewise_threshold=1.0, single_jitter=True)
    rs = sampler(rb, num_samples=15)
    assert rs.frustums.get_positions().shape == (R, 15, 3) and rs.deltas.shape == (R, 15, 1)
    pdf = PDFSampler(include_original=False, single_jitter=True)
    w = torch.ones(R, 15, 1, device=dev)
    rs2 = pdf(rb, rs, w, num_samples=7)
    assert rs2.frustums.starts.shape == (R, 7, 1)
    pdf(rb, rs, torch.zeros(R, 15, 1, device=dev), num_samples=7)  # all-zero weights must not crash (eps padding path)
    # renderers (tests/model_components/test_renderers.py): saturated first sample -> rgb ~ that colour, acc ~ 1, depth > 0
    weights = torch.zeros(R, 15, 1, device=dev)
    weights[:, 0] = 0.95
    rgb = torch.ones(R, 15, 3, device=dev)
    assert float(RGBRenderer(background_color="black")(rgb=rgb, weights=weights).max()) > 0.9
    assert float(AccumulationRenderer()(weights=weights).max()) > 0.9
    assert float(DepthRenderer(method="threshold")(weights=weights, ray_samples=rs).min()) > 0
    assert float(DepthRenderer(method="expected")(weights=weights, ray_samples=rs).min()) > 0


def test_dense_lattice_512_cubed_index_and_slab_merge(dev):
    """BASELINE cfg 5 at its stated size: the 512^3 lattice of one tile (134 217 728 points).
      * every lattice point and its integer voxel index, bit for bit, against the closed form evaluated independently (torch,
        fp32 point from the integer lattice coordinate; index = floor((p / s - (min_bound - voxel/2)) / voxel) in fp64, the
        Open3D rule of the oracle's voxel_index), on ALL points; two chunks additionally against the CPU oracle itself;
      * sharding needs no exchange: the query of two half slabs equals the one-pass query point for point, and the integer-key
        merge of the slabs' partial voxel sums equals the one-pass voxel down-sampling (keys, hit counts, fp16 features)."""
    import bench
    from oracle import nerf_oracle as O
    from presight_amd import extract

    res, voxel, psf = 512, 0.4, 0.05
    model, scene = bench.build_model(dev, seed=3, config="cfg2")
    model.eval()
    aabb = scene["aabbs"][0]
    lo, hi = aabb.reshape(2, 3)[0].float(), aabb.reshape(2, 3)[1].float()
    min_bound = lo.double().cpu() / psf - 1.0
    ref0 = (min_bound - voxel / 2).to(dev)
    chunk = 1 << 23
    inv = torch.tensor(1.0 / res, dtype=torch.float32, device=dev)
    n_checked = 0
    for s in range(0, res ** 3, chunk):
        n = min(chunk, res ** 3 - s)
        pts = extract.lattice_points(aabb, res, s, n, dev)
        lin = torch.arange(s, s + n, device=dev, dtype=torch.int64)
        ijk = torch.stack([lin // (res * res), (lin // res) % res, lin % res], -1)  # z fastest
        want_pts = lo + (hi - lo) * ((ijk.float() + 0.5) * inv)
        assert torch.equal(pts, want_pts)
        P = pts / psf
        idx = extract.voxel_index(P, voxel, min_bound)
        want_idx = torch.floor((P.double() - ref0) / voxel).to(torch.int64)
        assert torch.equal(idx, want_idx)
        if s in (0, 9 * chunk):
            assert torch.equal(idx.cpu(), O.voxel_index(P.cpu(), voxel, min_bound))
        n_checked += n
    assert n_checked == res ** 3 == 134217728
    # densest ~10 % of the lattice (a random-init tile has no surfaces: see bench.extract_main)
    probe = extract.dense_tile_query(model, aabb, res=64, density_threshold=-1.0)
    thr = float(torch.quantile(probe["densities"][:: max(1, probe["densities"].numel() // 100000)], 0.9))
    del probe
    full = extract.dense_tile_query(model, aabb, res=res, chunk=chunk, density_threshold=thr)
    half = (res ** 3) // 2 + 12345  # an odd split inside a z-column
    a = extract.dense_tile_query(model, aabb, res=res, chunk=chunk, start=0, count=half, density_threshold=thr)
    b = extract.dense_tile_query(model, aabb, res=res, chunk=chunk, start=half, count=res ** 3 - half, density_threshold=thr)
    assert full["points"].shape[0] > 5_000_000
    for k in ("points", "features", "densities", "voxel_index"):
        assert torch.equal(torch.cat([a[k], b[k]]), full[k]), k
    kw = dict(voxel=voxel, min_bound=full["min_bound"], points_max=full["points_max"], want_sums=True)
    one = extract.voxelize(full["points"], full["features"], None, **kw)
    va, vb = extract.voxelize(a["points"], a["features"], None, **kw), extract.voxelize(b["points"], b["features"], None, **kw)
    del full, a, b
    m = extract.merge_voxels([va, {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in vb.items()}])
    assert torch.equal(m["key"].cpu(), one["key"].cpu()) and torch.equal(m["hits"].cpu(), one["hits"].cpu())
    assert torch.equal(m["features"].cpu(), one["features"].cpu())  # fp64 sums of fp16 members -> fp16: exact
    assert int(one["hits"].sum()) == one["index"].shape[0] * 0 + int(m["hits"].sum())
    torch.testing.assert_close(m["points"].cpu(), one["points"].cpu(), rtol=0, atol=1e-5)


def test_dense_lattice_512_cubed_production_tile(dev, monkeypatch):
    """BASELINE cfg 5 on the PRODUCTION tile (K = 16 routed sub-fields, L10 F4 T2^20 main tables; ns/scripts/extract_priors.py:133-138
    on ns/fields/PreSight/ingp_field_ms.py:97-126): the 512^3 lattice over the union of the sub-field boxes through the gated + merged
    routed kernels.  Properties that hold at full size: the query of two odd slabs equals the one-pass query point for point (no
    exchange between slabs: the router, the gate and the voxel origin do not depend on the slab); the integer-key merge of the slabs'
    voxel sums equals the one-pass down-sampling; and on one 8 M-point chunk the kept points, their densities (bit for bit) and fp16
    features (one fp16 ulp) equal the UNGATED, UNMERGED routed query."""
    import bench
    from presight_amd import extract
    from presight_amd import field_ops as F

    res, voxel = 512, 0.4
    model, scene = bench.build_model(dev, seed=3, config="cfg3")
    model.eval()
    assert len(model.field.fields) == 16
    aabb = bench.tile_aabb(scene)
    chunk = 1 << 23
    probe = extract.dense_tile_query(model, aabb, res=64, density_threshold=-1.0)
    thr = float(torch.quantile(probe["densities"][:: max(1, probe["densities"].numel() // 100000)], 0.9))
    del probe
    F.GATE_STATS = torch.zeros(2, device=dev, dtype=torch.int64)
    full = extract.dense_tile_query(model, aabb, res=res, chunk=chunk, density_threshold=thr)
    ran, seen = F.GATE_STATS.tolist()
    F.GATE_STATS = None
    assert seen >= res ** 3 // 32 and 0 < ran < seen  # the gate skipped semantic-head tiles (routed layout: padded chunks add tiles)
    half = (res ** 3) // 2 + 54321
    a = extract.dense_tile_query(model, aabb, res=res, chunk=chunk, start=0, count=half, density_threshold=thr)
    b = extract.dense_tile_query(model, aabb, res=res, chunk=chunk, start=half, count=res ** 3 - half, density_threshold=thr)
    assert full["points"].shape[0] > 1_000_000
    for k in ("points", "features", "densities", "voxel_index"):
        assert torch.equal(torch.cat([a[k], b[k]]), full[k]), k
    kw = dict(voxel=voxel, min_bound=full["min_bound"], points_max=full["points_max"], want_sums=True)
    one = extract.voxelize(full["points"], full["features"], None, **kw)
    va, vb = extract.voxelize(a["points"], a["features"], None, **kw), extract.voxelize(b["points"], b["features"], None, **kw)
    del full, a, b
    m = extract.merge_voxels([va, {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in vb.items()}])
    assert torch.equal(m["key"].cpu(), one["key"].cpu()) and torch.equal(m["hits"].cpu(), one["hits"].cpu())
    assert torch.equal(m["features"].cpu(), one["features"].cpu())
    # one chunk against the ungated query on the unmerged routed kernels
    s0 = 7 * chunk
    pts = extract.lattice_points(aabb, res, s0, chunk, dev)
    dens_g, keep_g, feat_g = extract.query_priors(model, pts, thr)
    monkeypatch.setattr(F, "MERGED_MS", False)
    dens_u, feat_u = extract.query_priors(model, pts)
    assert torch.equal(dens_g, dens_u) and torch.equal(keep_g, dens_u > thr)
    diff = (feat_g.float() - feat_u[keep_g].float()).abs()
    assert float(diff.max()) <= 2.0 ** -10 and float((diff > 0).float().mean()) < 1e-3
