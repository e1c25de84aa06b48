"""stand-alone timing of the composite kernels at the cfg-2 shape (65536 rays x 64 samples x 64 semantic channels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from presight_amd import ops

dev = torch.device("cuda:0")
R, S, C = 65536, 64, 64
g = torch.Generator(device=dev).manual_seed(0)
eb = torch.sort(torch.rand(R, S + 1, device=dev, generator=g) * 10 + 0.1, dim=-1).values
w = (torch.rand(R, S, device=dev, generator=g) / S).requires_grad_(True)
rgb_s, sem_s = torch.rand(R, S, 3, device=dev, generator=g), torch.randn(R, S, C, device=dev, generator=g)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


gb = sem_s.numel() * 4 / 1e9
t = timed(lambda: sem_s.sum())
print(f"torch sum over sem_s        {t * 1e3:8.1f} us  {gb / t * 1e3:6.0f} GB/s")
t = timed(lambda: ops.composite(w.detach(), eb, rgb_s, sem_s))
print(f"composite forward           {t * 1e3:8.1f} us  {gb / t * 1e3:6.0f} GB/s (semantic rows only)")
t = timed(lambda: ops.composite(w.detach(), eb, None, None))
print(f"composite forward, no rows  {t * 1e3:8.1f} us")
out = ops.composite(w, eb, rgb_s, sem_s)
cot = [torch.randn_like(o) for o in out]
loss = sum((o * c).sum() for o, c in zip(out, cot) if o.requires_grad)
t = timed(lambda: torch.autograd.grad(loss, w, retain_graph=True))
print(f"composite backward (d_w)    {t * 1e3:8.1f} us  {gb / t * 1e3:6.0f} GB/s")
