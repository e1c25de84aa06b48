"""GPU parity at the MODULE level of the reference's operator seam: HashEncoding / SHEncoding / MLP objects
with the reference's constructor arguments, fed the reference-generated fixtures (the op-level kernels are tested in
test_hip_ops.py); the trunc_exp clamp branch (a9) through the fused fields; the per-ray loss kernels on bins that came out of
the reference's real sampler chain."""
import numpy as np
import pytest
import torch
from torch import nn

from conftest import load_golden, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().cpu()
    b = t(b) if isinstance(b, np.ndarray) else b.detach().cpu()
    torch.testing.assert_close(a.to(b.dtype).reshape(b.shape), b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag", ["kat", "cfg2small", "prodsmall", "prop0small", "prop1small"])
def test_hash_encoding_module(dev, gold_hashgrid, tag):
    """HashEncoding(num_levels, min_res, max_res, log2_hashmap_size, features_per_level): its own float32 `scalings` equal the
    reference's (encodings.py:281-284, e.g. 2047 not 2048), forward and table gradient match"""
    from presight_amd.components import HashEncoding

    G = gold_hashgrid
    L, mn, mx, l2t, F = [int(v) for v in G[tag + "_meta"]]
    enc = HashEncoding(num_levels=L, min_res=mn, max_res=mx, log2_hashmap_size=l2t, features_per_level=F, implementation="hip")
    assert torch.equal(enc.scalings, t(G[tag + "_scalings"]))
    assert enc.hash_table.shape == G[tag + "_table"].shape and enc.get_out_dim() == L * F
    enc.load_state_dict({"hash_table": t(G[tag + "_table"])})
    enc.to(dev)
    out = enc(t(G[tag + "_x"]).to(dev))
    close(out, G[tag + "_out"], rtol=1e-5, atol=1e-7)
    (out * t(G[tag + "_cot"]).to(dev)).sum().backward()
    close(enc.hash_table.grad, G[tag + "_grad_table"], rtol=1e-4, atol=1e-6)


def test_sh_encoding_module(dev, gold_ops):
    """SHEncoding(levels=4) on the shifted direction (d+1)/2, exactly as the fields call it (base_field.py:136-142)"""
    from presight_amd.components import SHEncoding
    from presight_amd.fields import get_normalized_directions

    G = gold_ops
    enc = SHEncoding(levels=4, implementation="hip")
    d = t(G["d"]).to(dev)
    close(enc(get_normalized_directions(d)), G["sh"], rtol=1e-6, atol=1e-7)
    assert enc.get_out_dim() == 16
    close(SHEncoding(levels=2)(get_normalized_directions(d)), G["sh"][:, :4], rtol=1e-6, atol=1e-7)
    with pytest.raises(ValueError):
        SHEncoding(levels=5)


@pytest.mark.parametrize("tag", ["base", "sem", "rgb", "prop", "skyrgb", "skysem", "base_prod", "tiny"])
def test_mlp_module(dev, gold_ops, tag):
    """MLP(in_dim, num_layers, layer_width, out_dim, out_activation) with the reference's layers.{i}.weight / .bias keys"""
    from presight_amd.components import MLP

    G = gold_ops
    n = len([k for k in G if k.startswith(f"mlp_{tag}_W")])
    Ws = [t(G[f"mlp_{tag}_W{i}"]) for i in range(n)]
    sig = bool(int(G[f"mlp_{tag}_sigmoid"]))
    m = MLP(in_dim=Ws[0].shape[1], num_layers=n, layer_width=Ws[0].shape[0], out_dim=Ws[-1].shape[0], activation=nn.ReLU(),
            out_activation=nn.Sigmoid() if sig else None, implementation="hip")
    sd = {}
    for i in range(n):
        sd[f"layers.{i}.weight"], sd[f"layers.{i}.bias"] = Ws[i], t(G[f"mlp_{tag}_b{i}"])
    m.load_state_dict(sd)
    m.to(dev)
    x = t(G[f"mlp_{tag}_x"]).to(dev).requires_grad_(True)
    y = m(x)
    close(y, G[f"mlp_{tag}_y"], rtol=1e-4, atol=1e-5)
    (y * t(G[f"mlp_{tag}_cot"]).to(dev)).sum().backward()
    close(x.grad, G[f"mlp_{tag}_gx"], rtol=1e-4, atol=1e-5)
    for i in range(n):
        close(m.layers[i].weight.grad, G[f"mlp_{tag}_gW{i}"], rtol=1e-4, atol=2e-5)
        close(m.layers[i].bias.grad, G[f"mlp_{tag}_gb{i}"], rtol=1e-4, atol=2e-5)


def test_tcnn_implementation_warns_once():
    import warnings

    from presight_amd import components

    components._WARNED_TCNN = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        components.SHEncoding(levels=4, implementation="tcnn")
        components.SHEncoding(levels=4, implementation="tcnn+fp32")
    assert len([x for x in w if "not table-compatible" in str(x.message)]) == 1


@pytest.mark.parametrize("raw", [-40.0, -20.0, -15.0, -1.0, 0.0, 1.0, 15.0, 20.0, 40.0])
def test_trunc_exp_clamp_branch_through_the_fused_fields(dev, raw):
    """a9 (ns/field_components/activations.py:28-52): forward exp(x), backward g * exp(clamp(x, -15, 15)).  The density
    pre-activation is pinned to `raw` (zero last-layer weights, bias = raw), so sigma = exp(raw) * selector for every point
    and d(loss)/d(bias) = sum(selector * dsigma) * exp(clamp(raw)): beyond +-15 the gradient must use the CLAMPED exponent
    while the forward does not — in the proposal field (vector-ALU head) and in the main field (MFMA head)."""
    from presight_amd.fields import PropNetDensityField, iNGPField

    torch.manual_seed(0)
    aabb = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
    g = torch.Generator().manual_seed(1)
    pos = ((torch.rand(1000, 3, generator=g) - 0.5) * 3).to(dev)  # no scene contraction: points outside the AABB have selector 0
    dsig = torch.rand(1000, generator=g).to(dev)
    expect_fwd = float(np.exp(np.float32(raw)))
    expect_grad = float(np.exp(np.float32(min(max(raw, -15.0), 15.0))))
    prop = PropNetDensityField(aabb, num_levels=8, features_per_level=1, log2_hashmap_size=10, max_res=1024, spatial_distortion=None,
                               implementation="hip").to(dev)
    main = iNGPField(aabb, num_levels=16, features_per_level=2, log2_hashmap_size=10, max_res=2048, use_semantics=True,
                     appearance_embedding_dim=16, spatial_distortion=None, implementation="hip").to(dev)
    with torch.no_grad():
        last = prop.mlp_base[1].layers[1]
        last.weight.zero_()
        last.bias.fill_(raw)
        lm = main.mlp_base_mlp.layers[1]
        lm.weight[0].zero_()
        lm.bias[0] = raw
    for field, bias in ((prop, prop.mlp_base[1].layers[1].bias), (main, main.mlp_base_mlp.layers[1].bias)):
        u, sel = field.points(pos=pos)
        sigma = field.evaluate(u, sel) if field is prop else field.evaluate(u, sel, None, None, 1, want_rgb=False, want_sem=False)[0]
        torch.testing.assert_close(sigma, sel * expect_fwd, rtol=2e-6, atol=0)
        assert 0 < int(sel.sum()) < 1000
        field.zero_grad(set_to_none=True)
        (sigma * dsig).sum().backward()
        want = float((sel * dsig).sum()) * expect_grad
        got = float(bias.grad[0])
        assert abs(got - want) <= 2e-5 * abs(want), (raw, got, want)


def test_losses_on_real_sampler_bins(dev):
    """interlevel (z-anti-aliased) and distortion losses on the bins of the reference's real sampler chain
    (tests/golden/losses_real.npz): value and gradients within 1e-3 of the reference"""
    from presight_amd import losses as L

    class _RS:
        def __init__(self, sbins):
            self.sbins = sbins

    G = load_golden("losses_real")
    wl = [t(G[f"w{i}"]).to(dev).requires_grad_(True) for i in range(3)]
    rs = [_RS(t(G[f"sbins{i}"]).to(dev)) for i in range(3)]
    il = L.z_anti_aliasing_interlevel_loss(wl, rs, (0.03, 0.003))
    close(il, G["interlevel"], rtol=1e-3, atol=0)
    g = torch.autograd.grad(il, wl[:2], retain_graph=True)
    for got, key in zip(g, ("g_interlevel_w0", "g_interlevel_w1")):
        ref = t(G[key])
        err = float((got.cpu().reshape(ref.shape) - ref).abs().max()) / float(ref.abs().max())
        assert err <= 1e-3, (key, err)
    dl = L.distortion_loss(wl, rs)
    close(dl, G["distortion"], rtol=2e-4, atol=0)
    (g2,) = torch.autograd.grad(dl, wl[2])
    close(g2, G["g_distortion_w2"], rtol=1e-3, atol=1e-8)
