"""phase timing of the main backward (profiling build: tools/build_variant.sh timing -DPS_TIMING; PRESIGHT_HIP_LIB=.../lib_timing.so)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from presight_amd._lib import lib

dev = torch.device("cuda:0")
model, scene = bench.build_model(dev, 42)
tr = bench.Trainer(model, scene, 1)
b = bench.make_batches(scene, dev, 2, 0)
for i in range(3):
    tr.step(b[i % 2])
torch.cuda.synchronize()
h = ctypes.CDLL(os.environ["PRESIGHT_HIP_LIB"])
out = (ctypes.c_ulonglong * 16)()
h.ps_debug_timing(out, 1)
outf = (ctypes.c_ulonglong * 16)()
h.ps_debug_timing_fwd(outf, 1)
outr = (ctypes.c_ulonglong * 16)()
h.ps_debug_timing_rgb(outr, 1)
n = 4
for i in range(n):
    tr.step(b[i % 2])
torch.cuda.synchronize()
h.ps_debug_timing(out, 0)
if os.environ.get("PRESIGHT_MAIN_BWD_SPLIT", "1") == "0":
    names = ["loop top / tile select", "loads + prep (x, h1, zb, dzb)", "semantic head backward", "colour head backward", "base backward + d(feature) stores"]
else:  # main_bwd_sem_kernel
    names = ["loop top", "load issue + d(sem) gather", "LZ: writes + dX (+ operand read-back)", "LZ: dW", "L1: relu + writes + dX", "L1: dW",
             "L0: relu + writes + dX", "L0: dW", "store d(head input)"]
tot = sum(out[i] for i in range(len(names)))
tiles = 65536 * 64 / 32
for i, nm in enumerate(names):
    print(f"{nm:40s} {out[i] / n / tiles:10.0f} clk/tile   {100 * out[i] / tot:5.1f} %")
print(f"{'total':40s} {tot / n / tiles:10.0f} clk/tile (s_memtime ticks; 1024 waves x 128 tiles)")

h.ps_debug_timing_fwd(outf, 0)
namesf = ["loop top", "-", "base MLP", "stores h1 zb sigma", "semantic MLP", "stores s1 s2 sem", "consume the next tile's inputs (the wait)",
          "colour input + MLP", "stores c1 c2 co rgb, fetch"]
totf = sum(outf[i] for i in range(len(namesf)))
tiles_per_wave = 65536 * 64 / 32 / 1024
print("main_fwd_kernel (1024 waves, one per SIMD)")
for i, nm in enumerate(namesf):
    print(f"{nm:40s} {outf[i] / n / 1024 / tiles_per_wave:10.0f} clk/tile   {100 * outf[i] / totf:5.1f} %")
print(f"{'total':40s} {totf / n / 1024 / tiles_per_wave:10.0f} clk/tile")

h.ps_debug_timing_rgb(outr, 0)
namesr = ["loop top", "issue loads c1 zb0 dirs app", "LZ (64->16): dX", "LZ: stage c2 + dW", "L1: relu + dX", "L1: stage c1 + dW",
          "L0: build colour input + dX", "L0: dW", "consume head, fetch next head", "d(appearance) sums + atomics, d(sigma), store"]
totr = sum(outr[i] for i in range(len(namesr)))
print("main_bwd_rgb_kernel (1024 waves x 128 tiles)")
for i, nm in enumerate(namesr):
    print(f"{nm:48s} {outr[i] / n / tiles:10.0f} clk/tile   {100 * outr[i] / totr:5.1f} %")
print(f"{'total':48s} {totr / n / tiles:10.0f} clk/tile")
