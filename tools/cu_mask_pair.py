"""Two-stream / CU-mask co-run experiment: can the MFMA-bound family of the cfg-2 step (main MLP forward / backward)
and the memory-bound family (hash encodes, table backward = bin + accumulate + the table's Adam) run BESIDE each other in less than
their sum when each gets its own share of the compute units?

The real kernels on real operands: one training step of the bench model is run with `field_ops._apply` recording the inputs of the
three field nodes; the nodes' forward / backward are then re-run at will through a stand-in ctx object, split into

    A1 main MLP backward   (sem_out_bwd, main_bwd_sem, composite_bwd, weights_bwd, main_bwd_rgb + base, ray_colour_bwd, unpack, merge_bwd)
    A2 main MLP forward    (merge, pack, ray_colour, main_fwd_kernel, composite_fwd, sem_out, clip)
    B1 main table backward (absmax + bin_kernel<2> + accumulate_kernel<2> incl. the table's Adam step)
    B2 main hash encode    (grid_encode L16 F2, with the record counts)
    B3 proposal field 0 forward (field encode L8 F1 + prop_fwd, 128 samples / ray)
    B4 proposal field 0 backward (prop_bwd + bin<1> + accumulate<1> + Adam)

and timed alone on the whole chip, alone on a masked stream (hipExtStreamCreateWithCUMask, the first X mask bits = X / 8 CUs of every
XCD), and as pairs on two streams: unmasked, and A on X CUs beside B on the other 256 - X, X in {160, 192, 224}.

    python tools/cu_mask_pair.py [rays]        (65536 = the full batch, 32768 = one half of a two-half pipeline)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from presight_amd import field_ops as FO  # noqa: E402
from presight_amd import ops  # noqa: E402

N_CU = 256


class Ctx:
    """stand-in for the autograd ctx of the field nodes (their forward / backward are static methods)"""

    def __init__(self, n):
        self.needs_input_grad = [True] * n
        self.saved_tensors = ()

    def save_for_backward(self, *t):
        self.saved_tensors = t

    def mark_non_differentiable(self, *a):
        pass


def masked_stream(dev, bits):
    """stream restricted to the compute units whose mask bit is set (bit i: XCD i % 8, then shader engine / CU round robin -- the KFD deals
    mask bits over the XCDs first, so the first X bits are X / 8 CUs of every XCD)"""
    hip = ctypes.CDLL("libamdhip64.so")
    words = (ctypes.c_uint32 * (N_CU // 32))()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), N_CU // 32, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value, device=dev)


def main():
    rays = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    reps = int(os.environ.get("REPS", "6"))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ops.SIDE_STREAM = False
    model, scene = bench.build_model(dev, 42, "cfg2")
    tr = bench.Trainer(model, scene, 1)
    batch = bench.make_batches(scene, dev, 1, 0, rays=rays)[0]
    for _ in range(2):
        tr.step(batch)
    calls = []
    orig_apply = FO._apply

    def rec_apply(fn, *args):
        calls.append((fn, args))
        return orig_apply(fn, *args)

    FO._apply = rec_apply
    import presight_amd.fields as fields_mod

    for m in (fields_mod,):  # (modules that imported the name)
        if hasattr(m, "_apply"):
            m._apply = rec_apply
    tr.step(batch)
    FO._apply = orig_apply
    torch.cuda.synchronize()
    names = [fn.__name__ for fn, _ in calls]
    print("recorded nodes:", names, flush=True)
    main_fn, main_args = next(c for c in calls if c[0].__name__ == "_MainFieldRenderF")
    prop_fn, prop_args = next(c for c in calls if c[0].__name__ == "_PropField")  # proposal level 0 (128 samples / ray)
    main_args = tuple(a.detach() if torch.is_tensor(a) and not isinstance(a, torch.nn.Parameter) else a for a in main_args)
    prop_args = tuple(a.detach() if torch.is_tensor(a) and not isinstance(a, torch.nn.Parameter) else a for a in prop_args)
    R = rays
    S = main_args[4]
    tables = [p for n, p in model.named_parameters() if n.endswith("hash_table")]

    def rearm():
        tr.opt.fused_armed = True
        for p in tables:
            p._ps_touched, p._ps_fused_done = False, False

    # ---- the pieces
    real_encode, real_scatter = FO._encode, FO._scatter
    cache = {}

    def enc_cached(u, table, scalings, g, count=False):
        return cache[("enc", g)]

    def scatter_capture(*a, **k):
        cache["scatter"] = (a, k)
        return None

    def scatter_skip(*a, **k):
        return None

    mctx = Ctx(len(main_args))
    with torch.no_grad():
        feat_counts = real_encode(main_args[0], FO._f32(main_args[7]), main_args[8], main_args[9], count=True)
        cache[("enc", main_args[9])] = feat_counts
        FO._encode = enc_cached
        main_fn.forward(mctx, *main_args)
        g_up = (torch.randn(R, 3, device=dev) * 1e-3, torch.randn(R, 1, device=dev) * 1e-3, None, torch.randn(R, 1, device=dev) * 1e-3,
                torch.randn(R, 64, device=dev) * 1e-3, torch.randn(R, S, device=dev) * 1e-3)
        FO._scatter = scatter_capture
        main_fn.backward(mctx, *g_up)
        sc_main = cache["scatter"]
        pctx = Ctx(len(prop_args))
        FO._encode = real_encode
        FO._scatter = real_scatter
        rearm()
        prop_fn.forward(pctx, *prop_args)
        dsig = torch.randn(prop_args[0].shape[0], device=dev) * 1e-3
    torch.cuda.synchronize()

    def A1():
        FO._encode, FO._scatter = enc_cached, scatter_skip
        main_fn.backward(mctx, *g_up)

    def A2():
        FO._encode, FO._scatter = enc_cached, scatter_skip
        main_fn.forward(Ctx(len(main_args)), *main_args)

    def B1():
        rearm()
        a, k = sc_main
        real_scatter(*a, **k)

    def B2():
        real_encode(main_args[0], FO._f32(main_args[7]), main_args[8], main_args[9], count=True)

    def B3():
        FO._encode, FO._scatter = real_encode, real_scatter
        prop_fn.forward(Ctx(len(prop_args)), *prop_args)

    def B4():
        FO._encode, FO._scatter = real_encode, real_scatter
        rearm()
        prop_fn.backward(pctx, dsig)

    work = {"A1 main MLP bwd": A1, "A2 main MLP fwd": A2, "B1 main table bwd+Adam": B1, "B2 main encode": B2, "B3 prop0 fwd (encode+mlp)": B3,
            "B4 prop0 bwd (mlp+table+Adam)": B4}
    cur = torch.cuda.current_stream()

    def run(pairs):
        """pairs: [(fn, stream)]; all start together behind a sleep on the default stream; -> ms per repetition until ALL are done"""
        with torch.no_grad():
            for fn, st in pairs:  # warm (allocator pools, workspaces of this stream)
                with torch.cuda.stream(st):
                    fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(30_000_000)  # ~12 ms: the host enqueues everything meanwhile
            e0.record(cur)
            for fn, st in pairs:
                st.wait_event(e0)
            for _ in range(reps):
                for fn, st in pairs:
                    with torch.cuda.stream(st):
                        fn()
            for fn, st in pairs:
                ev = torch.cuda.Event()
                ev.record(st)
                cur.wait_event(ev)
            e1.record(cur)
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    full = torch.cuda.Stream(device=dev)
    full2 = torch.cuda.Stream(device=dev)
    alone = {}
    print(f"\n== alone on the whole chip ({rays} rays, ms per call, {reps} back-to-back calls)")
    for name, fn in work.items():
        alone[name] = run([(fn, full)])
        print(f"  {name:32s} {alone[name]:7.3f}", flush=True)
    masks = {}
    for X in (128, 160, 192, 224):
        masks[X] = (masked_stream(dev, range(X)), masked_stream(dev, range(X, N_CU)))
    print("\n== alone on a masked stream: ms (x slowdown vs whole chip)")
    for name, fn in work.items():
        row = []
        for X in (128, 160, 192, 224):
            st = masks[X][0] if name.startswith("A") else masks[X][1]
            cu = X if name.startswith("A") else N_CU - X
            t = run([(fn, st)])
            row.append(f"{cu:3d} CUs {t:7.3f} (x{t / alone[name]:.2f})")
        print(f"  {name:32s} " + " | ".join(row), flush=True)
    print("\n== pairs: sum alone | two unmasked streams | A on X CUs beside B on 256 - X   (ratio to the sum)")
    for an in [n for n in work if n.startswith("A")]:
        for bn in [n for n in work if n.startswith("B")]:
            s = alone[an] + alone[bn]
            t_un = run([(work[an], full), (work[bn], full2)])
            row = [f"sum {s:6.3f}", f"unmasked {t_un:6.3f} ({t_un / s:.2f})"]
            for X in (128, 160, 192, 224):
                t = run([(work[an], masks[X][0]), (work[bn], masks[X][1])])
                row.append(f"X={X} {t:6.3f} ({t / s:.2f})")
            print(f"  {an[:2]}+{bn[:2]}  " + " | ".join(row), flush=True)
    # XCD split: A on 6 XCDs, B on the other 2 (their own L2s)
    xa = masked_stream(dev, [b for b in range(N_CU) if b % 8 < 6])
    xb = masked_stream(dev, [b for b in range(N_CU) if b % 8 >= 6])
    print("\n== XCD split (A: XCDs 0-5 = 192 CUs, B: XCDs 6-7 = 64 CUs with their own L2)")
    for an in [n for n in work if n.startswith("A")]:
        for bn in ("B1 main table bwd+Adam", "B3 prop0 fwd (encode+mlp)"):
            s = alone[an] + alone[bn]
            t = run([(work[an], xa), (work[bn], xb)])
            print(f"  {an[:2]}+{bn[:2]}  sum {s:6.3f} | xcd split {t:6.3f} ({t / s:.2f})", flush=True)


if __name__ == "__main__":
    main()
