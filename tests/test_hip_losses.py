"""GPU parity of the fused per-ray loss kernels against the reference-generated fixture (tests/golden/losses.npz)."""
import numpy as np
import pytest
import torch

from conftest import t

pytestmark = pytest.mark.gpu


class _RS:
    def __init__(self, sbins):
        self.sbins = sbins


def close(a, b, rtol=1e-4, atol=1e-6):
    a = a.detach().cpu()
    b = t(b) if isinstance(b, np.ndarray) else b.detach().cpu()
    torch.testing.assert_close(a.to(b.dtype).reshape(b.shape), b, rtol=rtol, atol=atol)


def close_interlevel_grad(got, ref, wp, n_total, rtol=2e-3, ws_err=1e-6):
    """d/dwp of max(ws-wp,0)^2/(wp+1e-5) has slope ~2/(wp+1e-5) in ws: a rounding-level error `ws_err` of the
    resampled histogram (values O(1) accumulated over ~130 fp32 terms) is amplified by up to 2e5 where wp == 0.
    Allow exactly that conditioning, nothing more."""
    got, ref, wp = got.detach().cpu(), ref.detach().cpu(), wp.detach().cpu()
    allowed = rtol * ref.abs() + ws_err * (2.0 / (wp + 1e-5)) / n_total + 1e-9
    bad = (got - ref).abs() > allowed
    assert not bool(bad.any()), (int(bad.sum()), float(((got - ref).abs() / allowed).max()))


def test_losses_golden(gold_losses):
    from presight_amd import losses as L

    G = gold_losses
    dev = torch.device("cuda:0")
    raw = [t(G[f"w{i}_raw"]).to(dev).requires_grad_(True) for i in range(3)]
    wl = [w / w.sum(-1, keepdim=True) * 0.9 for w in raw]
    rs = [_RS(t(G[f"bins{i}"]).to(dev)) for i in range(3)]
    # The fixture's bins are sorted uniform randoms: some are ~1e-5 wide, so the blurred-histogram cumsums carry
    # values ~1e4 and fp32 summation ORDER alone moves the result by ~1e-3 (torch's own CPU cumsum vs a sequential
    # one differ by that much).  Hold the value to 5e-3 and require 99 % of the gradient entries within 1 % of the
    # gradient scale; test_interlevel_vs_oracle_well_conditioned below is the tight check.
    il = L.z_anti_aliasing_interlevel_loss(wl, rs, (0.03, 0.003))
    close(il, G["interlevel"], rtol=5e-3)
    g = torch.autograd.grad(il, raw[:2], retain_graph=True)
    for got, key in zip(g, ("g_interlevel_w0", "g_interlevel_w1")):
        ref = t(G[key])
        bad = (got.cpu() - ref).abs() > 1e-2 * ref.abs().max()
        assert bad.float().mean() < 0.01, (key, float(bad.float().mean()))
    dl = L.distortion_loss(wl, rs)
    close(dl, G["distortion"], rtol=2e-4)
    (g2,) = torch.autograd.grad(dl, raw[2])
    close(g2, G["g_distortion_w2"], rtol=1e-3, atol=1e-7)
    acc = t(G["acc"]).to(dev).requires_grad_(True)
    sl = L.sky_loss(acc, t(G["sky_mask"]).to(dev))
    close(sl, G["sky_loss"])
    close(torch.autograd.grad(sl, acc)[0], G["g_sky"])
    pred = t(G["sem_pred"]).to(dev).requires_grad_(True)
    sm = L.semantic_loss(pred, t(G["sem_tgt"]).to(dev))
    close(sm, G["sem_loss"])
    close(torch.autograd.grad(sm, pred)[0], G["g_sem"])


def test_interlevel_vs_oracle_well_conditioned():
    """Bins with bounded-below widths (fp32-well-conditioned) and rays with exactly-zero weights, which exercise the
    flat-cdf tie rules of torch.max/min indices: tight comparison of value and gradient."""
    from oracle import nerf_oracle as O
    from presight_amd import losses as L

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    R = 64

    def bins(S):
        b = (torch.arange(S + 1)[None, :] + (torch.rand(R, S + 1, generator=g) - 0.5) * 0.6) / S
        b[:, 0], b[:, -1] = 0.0, 1.0
        return b

    bl = [bins(128), bins(64), bins(64)]
    wl = [torch.rand(R, S, generator=g) / S for S in (128, 64, 64)]
    wl[2][:, 10:30] = 0.0
    wl[2][:8] = 0.0
    wl[2][8:16, :60] = 0.0
    wl[0][:, ::3] = 0.0
    ref_w = [w.clone().requires_grad_(True) for w in wl]
    ref = O.interlevel_loss_zaa(ref_w, bl, (0.03, 0.003))
    gr = torch.autograd.grad(ref, ref_w[:2])
    dw = [w.to(dev).requires_grad_(True) for w in wl]
    out = L.z_anti_aliasing_interlevel_loss(dw, [_RS(b.to(dev)) for b in bl], (0.03, 0.003))
    close(out, ref, rtol=2e-4)
    gd = torch.autograd.grad(out, dw[:2])
    for a, b, w in zip(gd, gr, wl[:2]):
        close_interlevel_grad(a, b, w, w.numel(), ws_err=3e-6)
    d_ref = O.distortion_loss(bl[2], ref_w[2])
    d_out = L.distortion_loss(dw, [_RS(b.to(dev)) for b in bl])
    close(d_out, d_ref, rtol=2e-4)
