"""Timing and a bit-level fingerprint of the interlevel-loss kernel on cfg-2-sized inputs (65 536 rays, S = 64, Sp = 128 and 64): run once
per library (PRESIGHT_HIP_LIB) and compare the printed fingerprints -- a rewrite of the kernel's searches must not change a bit.
    python tools/dbg/interlevel_ab.py"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from presight_amd._lib import check, lib  # noqa: E402
from presight_amd.ops import _p, _stream  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
R, S = 65536, 64


def edges(n):
    d = torch.rand(R, n, device=dev, generator=g) ** 3 + 1e-4
    d[:, ::7] *= 0.01  # some nearly coincident edges
    e = torch.cumsum(d, 1)
    return torch.cat([torch.zeros(R, 1, device=dev), e], 1) / e[:, -1:]


c = edges(S).contiguous()
w = torch.rand(R, S, device=dev, generator=g) ** 4
w = (w / w.sum(1, keepdim=True)).contiguous()
for Sp, r in ((128, 0.03), (64, 0.003)):
    cp = edges(Sp).contiguous()
    cp[:, 1:-1] += (torch.rand(R, Sp - 1, device=dev, generator=g) - 0.5) * 1e-3
    cp = torch.sort(cp, 1).values.contiguous()
    cp[: R // 16] = c[: R // 16, :: max(1, S // Sp)][:, : Sp + 1] if Sp <= S else cp[: R // 16]  # edges that coincide with main edges
    wp = torch.rand(R, Sp, device=dev, generator=g).contiguous() * 0.02
    per_ray, dwp = torch.empty(R, device=dev), torch.empty(R, Sp, device=dev)
    run = lambda: check(lib().ps_interlevel_loss(_p(c), _p(w), _p(cp), _p(wp), R, S, Sp, float(r), _p(per_ray), _p(dwp), _stream()), "il")  # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record()
    torch.cuda.synchronize()
    h = hashlib.sha256(per_ray.cpu().numpy().tobytes() + dwp.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"Sp {Sp}: {a.elapsed_time(b) / 20 * 1e3:.1f} us per launch, sum {float(per_ray.double().sum()):.9e}, fingerprint {h}, finite {bool(torch.isfinite(dwp).all())}")
