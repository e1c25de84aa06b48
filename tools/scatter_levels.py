"""Per-level cost of the hash encode and of the binned table backward on the real sample distribution of a cfg-2 training step, for
the main table AND the two proposal tables: captures (u, d(features), table) of every field's scatter, then runs ps_grid_encode and
ps_grid_scatter_binned level by level (L = 1; the scatter writes a plain gradient, without the fused Adam step), and prints the
records per level (4 x-pair records per point and level; the share that the accumulate pass could merge because consecutive records of
a lane hit the same row pair is what the coarse levels' times show).     gpurun -- python tools/scatter_levels.py [steps_before]"""
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
    caps = []
    orig = FO._scatter

    def spy(u, dfeat, scalings, g, tshape, sink=None, counts=None, ws_with_absmax=None, sink_owner=None):
        caps.append(dict(u=u.clone(), dfeat=dfeat.clone(), scalings=scalings.clone(), g=g, table=sink_owner.detach().clone(),
                         counts=None if counts is None else counts.clone(), absmax=ws_with_absmax is not None))
        return orig(u, dfeat, scalings, g, tshape, sink, counts, ws_with_absmax, sink_owner)

    FO._scatter = spy
    tr.step(batches[0])
    FO._scatter = orig
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

    for cap in caps:
        u, dfeat, sc, g = cap["u"], cap["dfeat"], cap["scalings"], cap["g"]
        N = u.shape[0]
        L, F, l2t = g.num_levels, g.features_per_level, g.log2_hashmap_size
        ws = torch.empty(lib().ps_grid_scatter_workspace(L, F, l2t, N) + 4096, dtype=torch.uint8, device=dev)
        out = torch.empty((1 << l2t) * L, F, device=dev)
        items = lib().ps_grid_scatter_items(L, F, l2t, 1)
        cnt = cap["counts"]

        def run(uu, df, scl, LL, phase, i0=0, i1=-1, c=None):
            check(lib().ps_grid_scatter_binned_part(uu.data_ptr(), df.data_ptr(), scl.data_ptr(), LL, F, l2t, N, N * F, out.data_ptr(), 0,
                                                    0 if c is None else c.data_ptr(), 0, ws.data_ptr(), phase, i0, i1, s), "scatter")

        t_all = timed(lambda: run(u, dfeat, sc, L, 3, c=cnt))
        t_prep = timed(lambda: run(u, dfeat, sc, L, 1, c=cnt))
        t_acc = timed(lambda: run(u, dfeat, sc, L, 2))
        table = cap["table"].reshape(L, 1 << l2t, F)
        feat = torch.empty(L, N, F, device=dev)

        def enc(tb, scl, LL, f):
            check(lib().ps_grid_encode(u.data_ptr(), tb.data_ptr(), scl.data_ptr(), LL, F, l2t, N, N * F, f.data_ptr(), 0, s), "encode")

        t_enc = timed(lambda: enc(table, sc, L, feat))
        print(f"after {steps_before} steps: L{L} F{F} T2^{l2t} N={N} counts_from_fwd={cnt is not None}: encode (inference form, all levels) {t_enc:.3f} ms; "
              f"table backward all {t_all:.3f} ms = prepare {t_prep:.3f} + accumulate {t_acc:.3f}")
        per = items // L
        tot = 0.0
        for l in range(L):
            sl = sc[l:l + 1].contiguous()
            plane = dfeat[l].contiguous()
            tp = timed(lambda: run(u, plane, sl, 1, 1))
            ta = timed(lambda: run(u, plane, sl, 1, 2))
            t1 = timed(lambda: run(u, plane, sl, 1, 2, 0, 1))  # slice 0 of the level alone (dense levels: all of it)
            tot += tp + ta
            te = timed(lambda: enc(table[l].contiguous(), sl, 1, feat[:1]))
            rows = min((int(sl[0]) + 2) ** 3, 1 << l2t)
            print(f"  level {l:2d} res {int(sl[0]):5d} (<= {rows:8d} live rows): encode {te:.3f}  prepare (count+write) {tp:.3f}  accumulate {ta:.3f} ms "
                  f"(slice 0 alone {t1:.3f}; {per} slices; {4 * N / 1e6:.1f} M records)")
        print(f"  sum of single-level runs {tot:.3f} ms")


if __name__ == "__main__":
    main()
