"""GPU parity tests of the operator-level HIP kernels (through the C ABI) against the CPU oracle and the
golden fixtures generated from the reference.  Integer/index work must be bit exact; fp32 work is held to
rtol 1e-4..1e-5 (the kernels use fused multiply-adds and wavefront-parallel scans, so individual roundings
differ from ATen's sequential CPU loops)."""
import numpy as np
import pytest
import torch

from conftest import t
from conftest import assert_threshold_depth
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from presight_amd import ops as _ops

    return _ops


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(a)
    b = t(b) if isinstance(b, np.ndarray) else b.detach().cpu()
    torch.testing.assert_close(a.to(b.dtype).reshape(b.shape), b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------------------ hash grid
@pytest.mark.parametrize("tag", ["kat", "cfg2small", "prodsmall", "prop0small", "prop1small"])
def test_hashgrid_golden(ops, dev, gold_hashgrid, tag):
    G = gold_hashgrid
    L, _, _, l2t, F = [int(v) for v in G[tag + "_meta"]]
    x, table, sc = t(G[tag + "_x"]).to(dev), t(G[tag + "_table"]).to(dev).requires_grad_(True), t(G[tag + "_scalings"]).to(dev)
    idx = ops.hashgrid_indices(x, sc, L, l2t)
    assert torch.equal(idx.cpu(), t(G[tag + "_idx"]))  # bit exact
    out = ops.hashgrid_encode(x, table, sc, L, F, l2t)
    close(out, G[tag + "_out"], rtol=1e-5, atol=1e-7)
    (g,) = torch.autograd.grad((out * t(G[tag + "_cot"]).to(dev)).sum(), table)
    close(g, G[tag + "_grad_table"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("tag", ["cfg2_full", "prod_full", "prop0_full", "prop1_full"])
def test_hashgrid_indices_full_size(ops, dev, gold_hashgrid, tag):
    G = gold_hashgrid
    L, _, _, l2t, _ = [int(v) for v in G[tag + "_meta"]]
    idx = ops.hashgrid_indices(t(G[tag + "_x"]).to(dev), t(G[tag + "_scalings"]).to(dev), L, l2t)
    assert torch.equal(idx.cpu(), t(G[tag + "_idx"]))


@pytest.mark.parametrize("L,F,l2t,mx,N", [(16, 2, 19, 2048, 70001), (10, 4, 20, 16384, 33333), (8, 1, 20, 4096, 50000), (2, 2, 15, 64, 1)])
def test_hashgrid_vs_oracle_full_tables(ops, dev, L, F, l2t, mx, N):
    g = torch.Generator().manual_seed(L * 100 + F)
    sc = O.hash_scalings(L, 16, mx)
    table = (torch.rand((1 << l2t) * L, F, generator=g) * 2 - 1) * 1e-1
    x = torch.rand(N, 3, generator=g)
    x[: min(N, 5)] = 0.0
    ref, ridx = O.hash_encode(x, table, sc, l2t, return_indices=True)
    td = table.to(dev).requires_grad_(True)
    out = ops.hashgrid_encode(x.to(dev), td, sc.to(dev), L, F, l2t)
    assert torch.equal(ops.hashgrid_indices(x.to(dev), sc.to(dev), L, l2t).cpu(), ridx)
    close(out, ref, rtol=1e-5, atol=1e-7)
    cot = torch.rand(ref.shape, generator=g) - 0.5
    tr = table.clone().requires_grad_(True)
    (gref,) = torch.autograd.grad((O.hash_encode(x, tr, sc, l2t) * cot).sum(), tr)
    (gd,) = torch.autograd.grad((out * cot.to(dev)).sum(), td)
    close(gd, gref, rtol=1e-4, atol=1e-6)
    # size-independent property: linearity in the table
    out2 = ops.hashgrid_encode(x.to(dev), td.detach() * 3.0, sc.to(dev), L, F, l2t)
    close(out2, out.detach() * 3.0, rtol=1e-5, atol=1e-7)


def test_hashgrid_empty(ops, dev):
    sc = O.hash_scalings(4, 16, 128).to(dev)
    table = torch.zeros(4 << 10, 2, device=dev)
    out = ops.hashgrid_encode(torch.zeros(0, 3, device=dev), table, sc, 4, 2, 10)
    assert out.shape == (0, 8)


# ------------------------------------------------------------------------------ MLP
@pytest.mark.parametrize("tag", ["base", "sem", "rgb", "prop", "skyrgb", "skysem", "base_prod", "tiny"])
def test_mlp_golden(ops, dev, gold_ops, tag):
    G = gold_ops
    n = len([k for k in G if k.startswith(f"mlp_{tag}_W")])
    layers = [(t(G[f"mlp_{tag}_W{i}"]).to(dev).requires_grad_(True), t(G[f"mlp_{tag}_b{i}"]).to(dev).requires_grad_(True))
              for i in range(n)]
    x = t(G[f"mlp_{tag}_x"]).to(dev).requires_grad_(True)
    y = ops.mlp(x, layers, out_act="sigmoid" if int(G[f"mlp_{tag}_sigmoid"]) else None)
    close(y, G[f"mlp_{tag}_y"], rtol=1e-4, atol=1e-5)
    gr = torch.autograd.grad((y * t(G[f"mlp_{tag}_cot"]).to(dev)).sum(), [x] + [p for wb in layers for p in wb])
    close(gr[0], G[f"mlp_{tag}_gx"], rtol=1e-4, atol=1e-5)
    for i in range(n):
        close(gr[1 + 2 * i], G[f"mlp_{tag}_gW{i}"], rtol=1e-4, atol=2e-5)
        close(gr[2 + 2 * i], G[f"mlp_{tag}_gb{i}"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("dims,act,N", [([32, 64, 80], None, 100003), ([47, 64, 64, 3], "sigmoid", 65537), ([64, 64, 64, 64], None, 40000),
                                        ([8, 64, 1], None, 1), ([8, 64, 1], None, 17), ([16, 32, 32, 64], None, 4097),
                                        ([32, 32, 32, 3], "sigmoid", 23), ([16, 32, 32, 64], None, 23), ([32, 32, 32, 3], "sigmoid", 300)])
def test_mlp_vs_oracle_ragged_sizes(ops, dev, dims, act, N):
    g = torch.Generator().manual_seed(sum(dims) + N)
    layers = [((torch.rand(dims[i + 1], dims[i], generator=g) - 0.5) * (2.0 / dims[i] ** 0.5), torch.rand(dims[i + 1], generator=g) - 0.5)
              for i in range(len(dims) - 1)]
    x = torch.randn(N, dims[0], generator=g)
    # A hidden pre-activation within rounding distance of 0 may take the other ReLU branch on the GPU (different
    # summation order) and legitimately changes that row's gradients; nudge such rows away from the kink.
    h = x
    for W, b in layers[:-1]:
        z = torch.nn.functional.linear(h, W, b)
        x[(z.abs() < 1e-5).any(dim=1)] *= 1.01
        h = torch.relu(torch.nn.functional.linear(h, W, b))
    h = x
    for W, b in layers[:-1]:
        z = torch.nn.functional.linear(h, W, b)
        assert not bool((z.abs() < 1e-6).any())
        h = torch.relu(z)
    # asymmetric data: catches transposed fragments (cdna guide 5.4 rule 16)
    xr = x.clone().requires_grad_(True)
    lr = [(W.clone().requires_grad_(True), b.clone().requires_grad_(True)) for W, b in layers]
    yr = O.mlp_forward(xr, lr, out_act=act)
    cot = torch.randn(yr.shape, generator=g)
    gref = torch.autograd.grad((yr * cot).sum(), [xr] + [p for wb in lr for p in wb])
    xd = x.to(dev).requires_grad_(True)
    ld = [(W.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)) for W, b in layers]
    yd = ops.mlp(xd, ld, out_act=act)
    close(yd, yr, rtol=1e-4, atol=1e-5)
    gd = torch.autograd.grad((yd * cot.to(dev)).sum(), [xd] + [p for wb in ld for p in wb])
    close(gd[0], gref[0], rtol=1e-4, atol=1e-5)
    for a, b in zip(gd[1:], gref[1:]):
        scale = float(b.abs().max()) + 1e-12
        close(a / scale, b / scale, rtol=2e-4, atol=2e-5)


def test_mlp_unsupported_shape_raises(ops, dev):
    with pytest.raises(NotImplementedError):
        ops.mlp(torch.zeros(4, 7, device=dev), [(torch.zeros(24, 7, device=dev), torch.zeros(24, device=dev)),
                                                 (torch.zeros(3, 24, device=dev), torch.zeros(3, device=dev))])


def test_cpu_tensor_rejected(ops):
    with pytest.raises(RuntimeError):
        ops.sh4(torch.zeros(4, 3))


# ------------------------------------------------------------------------------ point-wise
def test_contract_sh_route(ops, dev, gold_ops):
    G = gold_ops
    u, sel = ops.contract(t(G["p"]).to(dev), t(G["aabb"]).to(dev))
    assert torch.equal(sel.cpu(), t(G["sel"]))
    close(u, G["u"], rtol=1e-6, atol=1e-7)
    close(ops.sh4(t(G["d"]).to(dev)), G["sh"], rtol=1e-6, atol=1e-7)
    a = ops.route(t(G["route_pts"]).to(dev), t(G["route_centroids"]).to(dev))
    assert torch.equal(a.cpu().long(), t(G["route_assign"]))


# ------------------------------------------------------------------------------ rays, samplers, renderers
def test_rays_and_spaced_sampler(ops, dev, gold_sampling):
    G = gold_sampling
    o, d, pa, dn = ops.generate_rays(t(G["ray_indices"]).to(dev), t(G["c2w"]).to(dev), t(G["fx"]).to(dev), t(G["fy"]).to(dev),
                                     t(G["cx"]).to(dev), t(G["cy"]).to(dev))
    close(o, G["origins"])
    close(d, G["directions"], atol=2e-7)
    close(pa, G["pixel_area"], rtol=2e-3, atol=1e-12)
    close(dn, G["directions_norm"])
    R = G["ray_indices"].shape[0]
    for mode in ("train", "eval"):
        jit = t(G[f"sp_{mode}_jitter"]).to(dev) if mode == "train" else None
        sb, eb = ops.spaced_bins(R, 128, 0.005 if mode == "train" else 0.0, 50.0, 5.0, jit, dev)
        close(sb[:, :-1], G[f"sp_{mode}_sstarts"], atol=2e-7)
        close(sb[:, 1:], G[f"sp_{mode}_sends"], atol=2e-7)
        close(eb[:, :-1], G[f"sp_{mode}_starts"], rtol=1e-5, atol=1e-6)
        close(eb[:, 1:], G[f"sp_{mode}_ends"], rtol=1e-5, atol=1e-5)
        pos = ops.sample_positions(t(G["origins"]).to(dev), t(G["directions"]).to(dev), eb)
        close(pos.view(R, 128, 3), G[f"sp_{mode}_positions"], rtol=1e-5, atol=1e-5)


def test_weights_pdf_composite(ops, dev, gold_sampling):
    G = gold_sampling
    for mode in ("train", "eval"):
        near = 0.005 if mode == "train" else 0.0
        eb = torch.cat([t(G[f"sp_{mode}_starts"]), t(G[f"sp_{mode}_ends"])[:, -1:]], -1).to(dev)
        sb = torch.cat([t(G[f"sp_{mode}_sstarts"]), t(G[f"sp_{mode}_sends"])[:, -1:]], -1).to(dev)
        sigma = t(G[f"w_{mode}_sigma"]).to(dev).requires_grad_(True)
        w = ops.weights_from_density(eb, sigma)
        close(w, G[f"w_{mode}_weights"], rtol=1e-4, atol=1e-7)
        (g,) = torch.autograd.grad((w * t(G[f"w_{mode}_cot"]).to(dev)).sum(), sigma)
        close(g, G[f"w_{mode}_gsigma"], rtol=1e-4, atol=1e-6)
        jit = t(G[f"pdf_{mode}_jitter"]).to(dev) if mode == "train" else None
        nsb, neb = ops.pdf_resample(t(G[f"w_{mode}_weights"]).to(dev), sb, 64, jit, float(G[f"pdf_{mode}_anneal"]), near, 50.0, 5.0)
        close(nsb[:, :-1], G[f"pdf_{mode}_sstarts"], rtol=1e-5, atol=2e-6)
        close(nsb[:, 1:], G[f"pdf_{mode}_sends"], rtol=1e-5, atol=2e-6)
        close(neb[:, :-1], G[f"pdf_{mode}_starts"], rtol=1e-4, atol=1e-5)
        close(neb[:, 1:], G[f"pdf_{mode}_ends"], rtol=1e-4, atol=1e-4)
        # renderers on the reference's own level-2 samples
        eb2 = torch.cat([t(G[f"pdf_{mode}_starts"]), t(G[f"pdf_{mode}_ends"])[:, -1:]], -1).to(dev)
        w2 = ops.weights_from_density(eb2, t(G[f"r_{mode}_sigma"]).to(dev))
        close(w2, G[f"r_{mode}_w"], rtol=1e-4, atol=1e-7)
        rgb, acc, depth, expd, sem = ops.composite(t(G[f"r_{mode}_w"]).to(dev), eb2, t(G[f"r_{mode}_rgb_in"]).to(dev),
                                                   t(G[f"r_{mode}_sem_in"]).to(dev))
        close(rgb, G[f"r_{mode}_rgb"], rtol=1e-5, atol=1e-6)
        close(acc, G[f"r_{mode}_acc"], rtol=1e-5, atol=1e-6)
        close(expd, G[f"r_{mode}_expdepth"], rtol=1e-5, atol=1e-5)
        close(sem, G[f"r_{mode}_sem"], rtol=1e-5, atol=1e-5)
        assert_threshold_depth(depth, t(G[f"r_{mode}_depth"]), G[f"r_{mode}_w"], eb2, what=f"threshold depth ({mode})")


def test_composite_backward_vs_oracle(ops, dev):
    g = torch.Generator().manual_seed(3)
    R, S, C = 37, 64, 64
    eb = torch.sort(torch.rand(R, S + 1, generator=g) * 10 + 0.1, dim=-1).values
    w = torch.rand(R, S, generator=g) / S
    rgb_s = torch.rand(R, S, 3, generator=g)
    sem_s = torch.rand(R, S, C, generator=g)
    cots = [torch.randn(R, 3, generator=g), torch.randn(R, 1, generator=g), torch.randn(R, 1, generator=g), torch.randn(R, C, generator=g)]

    def run(wt, rs, ss, on_gpu):
        if on_gpu:
            rgb, acc, _, expd, sem = ops.composite(wt, eb.to(dev), rs, ss)
            cs = [c.to(dev) for c in cots]
        else:
            steps = (eb[:, :-1] + eb[:, 1:]) / 2
            rgb = (wt[..., None] * rs).sum(1)
            acc = wt.sum(-1, keepdim=True)
            expd = O.expected_depth(wt, steps)
            sem = (wt[..., None] * ss).sum(1)
            cs = cots
        return (rgb * cs[0]).sum() + (acc * cs[1]).sum() + (expd * cs[2]).sum() + (sem * cs[3]).sum()

    a = [v.clone().requires_grad_(True) for v in (w, rgb_s, sem_s)]
    gr = torch.autograd.grad(run(*a, False), a)
    b = [v.to(dev).requires_grad_(True) for v in (w, rgb_s, sem_s)]
    gd = torch.autograd.grad(run(*b, True), b)
    for x, y in zip(gd, gr):
        close(x, y, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("S,C", [(64, 64), (40, 64), (64, 8), (48, 6), (130, 64)])
def test_composite_weights_only_backward(ops, dev, S, C):
    """d(weights) alone (the fused render's path: per-sample colour / semantic gradients are formed inside the field backward):
    16-byte row loads when C % 4 == 0, dword loads otherwise, the general kernel beyond 64 samples"""
    g = torch.Generator().manual_seed(11)
    R = 53
    eb = torch.sort(torch.rand(R, S + 1, generator=g) * 10 + 0.1, dim=-1).values
    w = torch.rand(R, S, generator=g) / S
    rgb_s, sem_s = torch.rand(R, S, 3, generator=g), torch.randn(R, S, C, generator=g)
    cots = [torch.randn(R, 3, generator=g), torch.randn(R, 1, generator=g), torch.randn(R, 1, generator=g), torch.randn(R, C, generator=g)]
    wr = w.clone().requires_grad_(True)
    steps = (eb[:, :-1] + eb[:, 1:]) / 2
    ref_out = [(wr[..., None] * rgb_s).sum(1), wr.sum(-1, keepdim=True), O.expected_depth(wr, steps), (wr[..., None] * sem_s).sum(1)]
    (gr,) = torch.autograd.grad(sum((o * c).sum() for o, c in zip(ref_out, cots)), wr)
    wd = w.to(dev).requires_grad_(True)
    rgb, acc, _, expd, sem = ops.composite(wd, eb.to(dev), rgb_s.to(dev), sem_s.to(dev))
    for got, want in zip((rgb, acc, expd, sem), ref_out):
        close(got, want.detach(), rtol=1e-5, atol=1e-5)
    (gd,) = torch.autograd.grad(sum((o * c.to(dev)).sum() for o, c in zip((rgb, acc, expd, sem), cots)), wd)
    close(gd, gr, rtol=1e-4, atol=1e-5)


def test_weights_large_batch_properties(ops, dev):
    """BASELINE-size batch: size-independent properties instead of a CPU comparison."""
    g = torch.Generator(device=dev).manual_seed(5)
    R, S = 65536, 128
    sb, eb = ops.spaced_bins(R, S, 0.005, 50.0, 5.0, torch.rand(R, 1, device=dev, generator=g), dev)
    assert bool((sb[:, 1:] >= sb[:, :-1]).all()) and bool((eb[:, 1:] >= eb[:, :-1]).all())  # sortedness
    sigma = torch.rand(R, S, device=dev, generator=g) * 5
    w = ops.weights_from_density(eb, sigma)
    acc = w.sum(-1)
    assert bool((w >= 0).all()) and bool((acc <= 1.0 + 1e-5).all())
    # transmittance identity: sum w = 1 - exp(-sum delta*sigma)
    tot = ((eb[:, 1:] - eb[:, :-1]) * sigma).sum(-1)
    torch.testing.assert_close(acc, 1 - torch.exp(-tot), rtol=1e-4, atol=1e-5)
    nsb, neb = ops.pdf_resample(w, sb, 64, torch.rand(R, 1, device=dev, generator=g), 1.0, 0.005, 50.0, 5.0)
    assert bool((nsb[:, 1:] >= nsb[:, :-1]).all()) and bool((nsb >= 0).all()) and bool((nsb <= 1).all())
    # idempotence: resampling a uniform histogram with centred u reproduces centred uniform bins
    ones = torch.ones(8, 64, device=dev)
    sb_u = torch.linspace(0, 1, 65, device=dev).expand(8, 65).contiguous()
    r_u, _ = ops.pdf_resample(ones, sb_u, 64, None, 1.0, 0.0, 50.0, 5.0)
    expect = torch.linspace(0.0, 1.0 - 1.0 / 65, 65, device=dev) + 1.0 / 130
    torch.testing.assert_close(r_u, expect.expand(8, 65), rtol=1e-5, atol=1e-6)


def test_adam_matches_torch_optim(dev):
    """ps_adam_step against torch.optim.Adam (CPU, fp32) with the reference's hyper-parameters, 5 steps, ragged sizes."""
    from presight_amd.optim import HipAdam

    g = torch.Generator().manual_seed(8)
    shapes = [(1000, 2), (7,), (33, 5), (1,)]
    ref_p = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    hip_p = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=1e-2, eps=1e-15, weight_decay=1e-5)
    hip = HipAdam(hip_p, lr=1e-2, eps=1e-15, weight_decay=1e-5)
    for step in range(5):
        for a, b in zip(ref_p, hip_p):
            gr = torch.randn(a.shape, generator=g) * (10.0 ** (step - 2))
            a.grad = gr.clone()
            b.grad = gr.to(dev)
        ref.step()
        hip.step()
    for a, b in zip(ref_p, hip_p):
        close(b, a.detach(), rtol=2e-5, atol=1e-6)


def test_device_data_feed_matches_per_pixel_getitem(dev):
    """DeviceImageChunk.gather == collating ImageChunk.__getitem__ (ns/data/PreSight/my_dataset.py:52-73) pixel by pixel"""
    from presight_amd.datafeed import DeviceImageChunk

    g = torch.Generator().manual_seed(4)
    P, W, H = 5000, 1600, 900
    chunk = dict(rgbs=torch.rand(P, 3, generator=g), skies=(torch.rand(P, generator=g) < 0.2).float(), depths=torch.rand(P, generator=g) * 80,
                 features=torch.rand(P, 64, generator=g), pixel_indices=torch.randint(0, W * H, (P,), generator=g),
                 image_indices=torch.randint(0, 1440, (P,), generator=g), video_ids=torch.randint(0, 6, (P,), generator=g),
                 widths=torch.full((P,), W))
    dc = DeviceImageChunk(**{k: v.to(dev) for k, v in chunk.items()})
    pick = torch.randint(0, P, (777,), generator=g)
    b = dc.gather(pick.to(dev))
    ri = torch.stack([chunk["image_indices"][pick], chunk["pixel_indices"][pick] // W, chunk["pixel_indices"][pick] % W], -1)
    assert torch.equal(b["ray_indices"].cpu(), ri)
    assert torch.equal(b["rgb"].cpu(), chunk["rgbs"][pick]) and torch.equal(b["features"].cpu(), chunk["features"][pick])
    assert torch.equal(b["sky"].cpu(), chunk["skies"][pick]) and torch.equal(b["depth"].cpu(), chunk["depths"][pick])
    assert torch.equal(b["video_id"].cpu(), chunk["video_ids"][pick])
    s = dc.sample_batch(1000)
    assert s["ray_indices"].shape == (1000, 3) and int(s["ray_indices"][:, 2].max()) < W and int(s["ray_indices"][:, 1].max()) < H
