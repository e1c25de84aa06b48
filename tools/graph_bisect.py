"""Which part of a training step survives hipGraph capture + replay?  Each stage runs in its own process.
    python tools/graph_bisect.py            # runs all stages as children
    python tools/graph_bisect.py <stage>    # fwd | fwdloss | bwd | full"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stage(name):
    import torch

    import bench
    from presight_amd import ops
    from presight_amd.rays import RayBundle

    dev = torch.device("cuda", 0)
    model, scene = bench.build_model(dev, seed=42)
    tr = bench.Trainer(model, scene, 1)
    batch = bench.make_batches(scene, dev, 1, 0)[0]
    m, s = model, scene
    variant = name.split("_")[1] if "_" in name else ""
    name = name.split("_")[0]
    if variant == "noemb":
        for e in (m.appearance_embedding, m.video_embedding):
            for p in e.parameters():
                p.requires_grad_(False)
    prop_grads = variant != "noprop"
    if variant == "unfused":
        m.fused_render = False

    def body():
        if name == "full":
            tr.step(batch)
            return
        m.train()
        tr.grads.zero_()
        o, d, pa, dn = ops.generate_rays(batch["ray_indices"], s["c2w"], s["fx"], s["fy"], s["cx"], s["cy"])
        rb = RayBundle(o, d, pa, camera_indices=batch["ray_indices"][:, 0:1],
                       metadata={"video_id": batch["video_ids"][:, None], "directions_norm": dn})
        m.proposal_sampler._steps_since_update = (1 << 30) if prop_grads else 0
        if name == "fwd":
            with torch.no_grad():
                m(rb)
            return
        out = m(rb)
        loss = sum(m.get_loss_dict(out, batch).values())
        if name == "bwd":
            (loss * 1024.0).backward()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    torch.cuda.synchronize()
    print(name, "captured", flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print(f"{name}: replay ok, {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        stage(sys.argv[1])
    else:
        for st in ("bwd_noemb", "bwd_noprop", "bwd_unfused"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), st], capture_output=True, text=True, timeout=280)
            tail = (r.stdout + r.stderr).strip().splitlines()[-3:]
            print(f"== {st}: rc={r.returncode}", *tail, sep="\n   ", flush=True)
