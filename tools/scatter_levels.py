"""Per-level cost of the binned table backward on the real sample distribution of a cfg-2 training step:
captures (u, d(features)) of the main field's scatter, then runs ps_grid_scatter_binned level by level (L=1)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from presight_amd import field_ops as FO  # noqa: E402
from presight_amd._lib import check, lib  # noqa: E402


def main():
    steps_before = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    dev = torch.device("cuda", 0)
    model, scene = bench.build_model(dev, seed=42)
    tr = bench.Trainer(model, scene, 1)
    batches = bench.make_batches(scene, dev, 4, 0)
    for i in range(steps_before):
        tr.step(batches[i % 4])
    cap = {}
    orig = FO._scatter

    def spy(u, dfeat, scalings, g, tshape, sink=None):
        if g.features_per_level == 2:
            cap.update(u=u.clone(), dfeat=dfeat.clone(), scalings=scalings.clone(), g=g)
        return orig(u, dfeat, scalings, g, tshape, sink)

    FO._scatter = spy
    tr.step(batches[0])
    FO._scatter = orig
    u, dfeat, sc, g = cap["u"], cap["dfeat"], cap["scalings"], cap["g"]
    N = u.shape[0]
    L, F, l2t = g.num_levels, g.features_per_level, g.log2_hashmap_size
    ws = torch.empty(lib().ps_grid_scatter_workspace(L, F, l2t, N) + 4096, dtype=torch.uint8, device=dev)
    out = torch.empty((1 << l2t) * L, F, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    t_all = timed(lambda: check(lib().ps_grid_scatter_binned(u.data_ptr(), dfeat.data_ptr(), sc.data_ptr(), L, F, l2t, N, N * F,
                                                             out.data_ptr(), 0, ws.data_ptr(), s), "scatter"))
    print(f"after {steps_before} steps: all {L} levels together {t_all:.3f} ms (N={N})")
    tot = 0.0
    for l in range(L):
        sl = sc[l:l + 1].contiguous()
        plane = dfeat[l].contiguous()
        t = timed(lambda: check(lib().ps_grid_scatter_binned(u.data_ptr(), plane.data_ptr(), sl.data_ptr(), 1, F, l2t, N, N * F,
                                                             out.data_ptr(), 0, ws.data_ptr(), s), "scatter"))
        tot += t
        print(f"  level {l:2d} res {int(sl[0]):5d}: {t:.3f} ms")
    print(f"  sum of single-level runs {tot:.3f} ms")


if __name__ == "__main__":
    main()
