"""The weight-space and per-ray pieces of the factored render node (csrc/factored.hip) one by one against their
closed forms in plain torch fp32 on the CPU (SH basis from the pinned oracle): merged linear layers, the per-ray output layer of
the semantic head, the per-ray term of the colour head's first layer -- forward values and every gradient they hand back."""
import ctypes

import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _call(name, *args):
    from presight_amd._lib import check, lib

    check(getattr(lib(), name)(*args), name)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def test_merge_linear_forward_and_backward(dev):
    g = torch.Generator().manual_seed(1)
    O_, K, I = 64, 64, 32
    W0, b0 = torch.randn(O_, K, generator=g), torch.randn(O_, generator=g)
    We, be = torch.randn(K, I, generator=g), torch.randn(K, generator=g)
    dWm, dbm = torch.randn(O_, I, generator=g), torch.randn(O_, generator=g)
    ref = [t.clone().requires_grad_(True) for t in (W0, b0, We, be)]
    Wm_r, bm_r = ref[0] @ ref[2], ref[0] @ ref[3] + ref[1]
    ((Wm_r * dWm).sum() + (bm_r * dbm).sum()).backward()
    d = [t.to(dev) for t in (W0, b0, We, be, dWm, dbm)]
    Wm, bm = torch.empty(O_, I, device=dev), torch.empty(O_, device=dev)
    _call("ps_merge_linear_fwd", _p(d[0]), _p(d[1]), _p(d[2]), _p(d[3]), O_, K, I, _p(Wm), _p(bm), _stream())
    torch.testing.assert_close(Wm.cpu(), Wm_r.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(bm.cpu(), bm_r.detach(), rtol=1e-5, atol=1e-5)
    grads = [torch.full_like(t, 0.5).to(dev) for t in (W0, b0, We, be)]  # the entry ADDS into its destinations
    _call("ps_merge_linear_bwd", _p(d[4]), _p(d[5]), _p(d[0]), _p(d[2]), _p(d[3]), O_, K, I, _p(grads[0]), _p(grads[1]), _p(grads[2]),
          _p(grads[3]), _stream())
    for got, r in zip(grads, ref):
        torch.testing.assert_close(got.cpu() - 0.5, r.grad, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("R", [1, 63, 130, 1000])
def test_semantic_output_layer_per_ray(dev, R):
    g = torch.Generator().manual_seed(2)
    H, acc = torch.randn(R, 64, generator=g), torch.rand(R, generator=g)
    W, b, d = torch.randn(64, 64, generator=g) * 0.2, torch.randn(64, generator=g), torch.randn(R, 64, generator=g)
    Hr, ar, Wr, br = (t.clone().requires_grad_(True) for t in (H, acc, W, b))
    sem_r = Hr @ Wr.T + br[None, :] * ar[:, None]  # W (sum_n w_n s_n) + b sum_n w_n
    (sem_r * d).sum().backward()
    Hd, ad, Wd, bd, dd = (t.to(dev) for t in (H, acc, W, b, d))
    sem = torch.empty(R, 64, device=dev)
    _call("ps_sem_out_fwd", _p(Hd), _p(ad), _p(Wd), _p(bd), R, 64, _p(sem), _stream())
    torch.testing.assert_close(sem.cpu(), sem_r.detach(), rtol=1e-5, atol=1e-5)
    v, cray = torch.empty(R, 64, device=dev), torch.empty(R, device=dev)
    dW, db = torch.zeros(64, 64, device=dev), torch.zeros(64, device=dev)
    _call("ps_sem_out_bwd", _p(dd), _p(Hd), _p(ad), _p(Wd), _p(bd), R, 64, _p(v), _p(cray), _p(dW), _p(db), _stream())
    torch.testing.assert_close(v.cpu(), Hr.grad, rtol=1e-5, atol=1e-5)          # W^T d(sem): the per-ray gradient of the composited activations
    torch.testing.assert_close(cray.cpu(), ar.grad, rtol=1e-5, atol=1e-5)       # <d(sem), b>: joins d(accumulation)
    s = float(Wr.grad.abs().max())
    torch.testing.assert_close(dW.cpu() / s, Wr.grad / s, rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-5, atol=1e-4 * float(br.grad.abs().max()))


@pytest.mark.parametrize("R,S,A,HC", [(1, 16, 16, 64), (77, 64, 16, 64), (300, 32, 0, 32), (129, 48, 5, 64)])
def test_colour_head_per_ray_term(dev, R, S, A, HC):
    """c_ray = W0[:, SH16] SH(dir) + W0[:, app] app (ns/fields/PreSight/ingp_field.py:239-262: both are constant along a ray);
    backward from the per-16-sample-block sums the field kernel hands over"""
    g = torch.Generator().manual_seed(3)
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    app = torch.randn(R, A, generator=g) if A else None
    W0 = torch.randn(HC, 31 + A, generator=g) * 0.3
    dpart = torch.randn(R * S // 16, HC, generator=g)  # per-block gradients of the term
    W0r = W0.clone().requires_grad_(True)
    appr = app.clone().requires_grad_(True) if A else None
    x = O.sh4_of_direction(dirs)
    c_ref = x @ W0r[:, :16].T + (appr @ W0r[:, 31:].T if A else 0.0)
    d_ray = dpart.reshape(R, S // 16, HC).sum(1)
    (c_ref * d_ray).sum().backward()
    dd, W0d = dirs.to(dev), W0.to(dev)
    appd = app.to(dev) if A else None
    c = torch.empty(R, HC, device=dev)
    _call("ps_ray_colour_fwd", _p(dd), _p(appd), _p(W0d), R, A, HC, _p(c), _stream())
    torch.testing.assert_close(c.cpu(), c_ref.detach(), rtol=1e-5, atol=1e-5)
    dW0 = torch.full((HC, 31 + A), 0.25, device=dev)  # += into the SH / appearance columns only
    dapp = torch.empty(R, A, device=dev) if A else None
    _call("ps_ray_colour_bwd", _p(dpart.to(dev)), _p(dd), _p(appd), _p(W0d), R, S, A, HC, _p(dW0), _p(dapp), _stream())
    got = dW0.cpu() - 0.25
    s = float(W0r.grad.abs().max())
    torch.testing.assert_close(got / s, W0r.grad / s, rtol=1e-5, atol=2e-6)
    assert float(got[:, 16:31].abs().max()) == 0.0  # the geometry columns belong to the field kernel
    if A:
        torch.testing.assert_close(dapp.cpu(), appr.grad, rtol=1e-5, atol=1e-5)
