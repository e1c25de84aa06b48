"""Why is the gated forward of the routed production tile at 0.34 matrix-pipe busy (prior extraction, r05 PMC)?  One 8 M-point lattice
slab through extract.query_priors at three thresholds: nothing passes the gate (base MLP only), the bench's threshold (densest 10 %),
everything passes -- main_field_fwd region time and executed / visited tiles each.  If the 10 % case costs much more than 0.9 x (none) +
0.1 x (all), the time is load imbalance: the sub-field workgroups are dealt by point count, the semantic head runs where the dense points are."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from presight_amd import extract, prof  # noqa: E402
from presight_amd import field_ops as F  # noqa: E402

dev = torch.device("cuda:0")
model, scene = bench.build_model(dev, seed=42, config="cfg3")
model.eval()
aabb = bench.tile_aabb(scene)
probe = extract.dense_tile_query(model, aabb, res=64, density_threshold=-1.0)
thr = float(torch.quantile(probe["densities"][:: max(1, probe["densities"].numel() // 100000)], 0.9))
for slab in (0, 5, 10):
    pts = extract.lattice_points(aabb, 512, slab * (1 << 23), 1 << 23, dev)
    for name, t in (("none", 1e30), ("10 %", thr), ("all", -1.0)):
        extract.query_priors(model, pts, t)
        F.GATE_STATS = torch.zeros(2, device=dev, dtype=torch.int64)
        prof.enable(True)
        for _ in range(3):
            extract.query_priors(model, pts, t)
        k = prof.summary()
        prof.enable(False)
        st = F.GATE_STATS.tolist()
        F.GATE_STATS = None
        print(f"slab {slab:2d} gate {name:5s}: main_field_fwd {k['main_field_fwd'][1]:.3f} ms, head tiles {st[0] / max(st[1], 1):.3f}, encode "
              f"{k.get('grid_encode_L10F4', (0, 0))[1]:.3f} ms", flush=True)
