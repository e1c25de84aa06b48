"""Micro-benchmarks of the operator-level kernels at BASELINE cfg-2 sizes (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from presight_amd import ops
from oracle import nerf_oracle as O

dev = torch.device("cuda:0")

def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n): fn()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n

R = 65536
for name, L, F, l2t, mx, S in [("main L16F2 T19", 16, 2, 19, 2048, 64), ("prop L8F1 T20", 8, 1, 20, 1024, 128), ("prod L10F4 T20", 10, 4, 20, 16384, 64)]:
    N = R * S
    sc = O.hash_scalings(L, 16, mx).to(dev)
    table = (torch.rand((1 << l2t) * L, F, device=dev) * 2 - 1) * 1e-3
    # ray-coherent points: origins + t*dirs inside the unit cube
    o = torch.rand(R, 1, 3, device=dev) * 0.2 + 0.4
    d = torch.nn.functional.normalize(torch.randn(R, 1, 3, device=dev), dim=-1)
    tt = torch.linspace(0, 0.4, S, device=dev).view(1, S, 1)
    x = (o + d * tt).clamp(0, 1).reshape(-1, 3).contiguous()
    out = torch.empty(N, L * F, device=dev)
    dt = torch.zeros_like(table)
    from presight_amd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: check(lib().ps_hashgrid_fwd(x.data_ptr(), table.data_ptr(), sc.data_ptr(), L, F, l2t, N, out.data_ptr(), st), "f")
    b = lambda: check(lib().ps_hashgrid_bwd(x.data_ptr(), out.data_ptr(), sc.data_ptr(), L, F, l2t, N, dt.data_ptr(), st), "b")
    tf, tb = timeit(f), timeit(b)
    gathers = N * L * 8
    print(f"hashgrid {name}: N={N} fwd {tf:.3f} ms ({gathers/tf/1e6:.1f} G gathers/s, {gathers*F*4/tf/1e6:.0f} GB/s alg)  bwd {tb:.3f} ms ({gathers*F/tb/1e6:.1f} G atomics/s)")
    xr = torch.rand(N, 3, device=dev)
    f2 = lambda: check(lib().ps_hashgrid_fwd(xr.data_ptr(), table.data_ptr(), sc.data_ptr(), L, F, l2t, N, out.data_ptr(), st), "f")
    b2 = lambda: check(lib().ps_hashgrid_bwd(xr.data_ptr(), out.data_ptr(), sc.data_ptr(), L, F, l2t, N, dt.data_ptr(), st), "b")
    tf, tb = timeit(f2), timeit(b2)
    print(f"   random points: fwd {tf:.3f} ms  bwd {tb:.3f} ms ({gathers*F/tb/1e6:.1f} G atomics/s)")

for dims, act in [([32, 64, 80], None), ([64, 64, 64, 64], None), ([47, 64, 64, 3], "sigmoid"), ([8, 64, 1], None)]:
    N = R * 64
    layers = [((torch.rand(dims[i + 1], dims[i], device=dev) - 0.5) * 0.3, torch.rand(dims[i + 1], device=dev) - 0.5) for i in range(len(dims) - 1)]
    x = torch.randn(N, dims[0], device=dev)
    spec = ops.mlp_spec(dims)
    packed = spec.pack(layers, dev)
    y = torch.empty(N, dims[-1], device=dev)
    from presight_amd._lib import lib, check
    st = torch.cuda.current_stream().cuda_stream
    act_i = 1 if act else 0
    f = lambda: check(lib().ps_mlp_fwd(x.data_ptr(), packed.data_ptr(), y.data_ptr(), N, dims[0], dims[1], dims[-1], spec.nl, act_i, st), "f")
    gpart = torch.empty(256, spec.g_total, device=dev)
    dx = torch.empty_like(x)
    b = lambda: check(lib().ps_mlp_bwd(x.data_ptr(), y.data_ptr(), packed.data_ptr(), dx.data_ptr(), gpart.data_ptr(), N, dims[0], dims[1], dims[-1], spec.nl, act_i, st), "b")
    tf, tb = timeit(f), timeit(b)
    mac = sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    print(f"mlp {dims}: N={N} fwd {tf:.3f} ms ({2*mac*N/tf/1e9:.1f} TFLOP/s)  bwd {tb:.3f} ms ({3*2*mac*N/tb/1e9:.1f} TFLOP/s incl recompute)")
