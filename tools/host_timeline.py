"""How host-bound is one training step?  (1) enqueue time vs GPU time per step, (2) the same step replayed from a
hipGraph (torch.cuda.CUDAGraph capture of Trainer.step) = the step time with zero launch overhead.
    python tools/host_timeline.py [--graph]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    model, scene = bench.build_model(dev, seed=42)
    tr = bench.Trainer(model, scene, 1)
    batches = bench.make_batches(scene, dev, 4, 0)
    for i in range(5):
        tr.step(batches[i % 4])
    torch.cuda.synchronize()
    enq = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        a = time.perf_counter()
        tr.step(batches[i % 4])
        enq.append(time.perf_counter() - a)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"eager: enqueue {t_enq / args.steps * 1e3:.2f} ms/step (min {min(enq) * 1e3:.2f}, max {max(enq) * 1e3:.2f}), "
          f"wall {t_all / args.steps * 1e3:.2f} ms/step")
    if args.graph:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for i in range(3):
                tr.step(batches[0])
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            tr.step(batches[0])
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.replay()
        torch.cuda.synchronize()
        print(f"graph replay: {(time.perf_counter() - t0) / args.steps * 1e3:.2f} ms/step")


if __name__ == "__main__":
    main()
