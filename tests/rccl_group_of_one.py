"""Child process of tests/test_hip_dist.py::test_rccl_group_of_one_runs_the_exchange: the trainer's bucketed exchange issued through a
process group of ONE rank over RCCL (two ranks cannot share a GPU under RCCL; the two-rank functional run of a one-GPU box goes over
gloo).  Sum over one rank / 1 is the identity, so the exchanged gradient buffer of a step must equal that of a trainer that exchanges nothing
(up to the float atomics of a few per-ray gradients).  Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch.distributed as dist

    import bench
    from presight_amd.dist import COMM_LOG, exchanging, init_from_env
    from test_hip_dist import _tiny_model

    K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rank, local_rank, world = init_from_env("cuda")
    assert world == 1 and dist.is_initialized() and dist.get_backend() == "nccl" and exchanging()
    dev = torch.device("cuda", 0)
    out = {"backend": dist.get_backend(), "K": K}
    for mode in ("allreduce", "sharded", "sparse"):
        os.environ["PRESIGHT_EXCHANGE_WORLD_OF_ONE"] = "1"
        model_a, scene = _tiny_model(dev, K=K)
        tr_a = bench.Trainer(model_a, scene, 1, exchange=mode)
        assert tr_a.exchange == mode and not tr_a.grads.dry and not tr_a.fused_table_adam and len(tr_a.grads._buckets) >= 5
        os.environ["PRESIGHT_EXCHANGE_WORLD_OF_ONE"] = "0"
        model_b, _ = _tiny_model(dev, K=K)
        tr_b = bench.Trainer(model_b, scene, 1, fused_table_adam=False)
        assert tr_b.grads.dry or not tr_b.grads._buckets
        os.environ["PRESIGHT_EXCHANGE_WORLD_OF_ONE"] = "1"
        batches = bench.make_batches(scene, dev, 3, 0, rays=512)
        jit = [[torch.rand(512, 1, device=dev) for _ in range(3)] for _ in range(3)]
        c0 = tr_a.grads.stats["collectives"]
        worst, losses = 0.0, []
        for i in range(3):
            for tr, flag in ((tr_a, "1"), (tr_b, "0")):
                os.environ["PRESIGHT_EXCHANGE_WORLD_OF_ONE"] = flag
                ld, _ = tr.step(dict(batches[i], jitter=jit[i]))
                losses.append(float(sum(ld.values())))
            torch.cuda.synchronize()
            if i == 0:
                # same parameters, same batch: the exchanged gradient buffer (sum over one rank / 1) against the plain one, parameter by
                # parameter.  (Not bit for bit: a few per-ray gradients are float atomics, two RUNS of the same step differ by ~1e-6 of
                # the largest entry, and Adam with eps = 1e-15 turns that into +-lr on entries whose gradient is noise -- so the
                # comparison is made on the gradients of the first step, not on the parameters after it.)
                fa, fb = tr_a.grads, tr_b.grads

                def by_name(fg, model):  # (the record exchange orders the flat buffer differently: tables first)
                    names = {id(p): n for n, p in model.named_parameters()}
                    return {names[id(p)]: fg.flat[o:o + p.numel()] for p, o in zip(fg.params, fg.offsets)}

                ga_all, gb_all = by_name(fa, model_a), by_name(fb, model_b)
                assert set(ga_all) == set(gb_all)
                for name, gb in gb_all.items():
                    ga = ga_all[name]
                    ref = float(gb.abs().max())
                    err = float((ga - gb).abs().max()) / ref if ref > 0 else float(ga.abs().max())
                    if err > worst:
                        worst, out[f"{mode}_worst_parameter"] = err, name
                # the hash tables' gradients are int64 fixed-point sums: the record exchange (one run per slice here) must reproduce the
                # plain binned backward BIT FOR BIT
                out[f"{mode}_tables_bit_equal"] = all(torch.equal(ga_all[n], gb_all[n]) for n in gb_all if n.endswith("hash_table"))
        os.environ["PRESIGHT_EXCHANGE_WORLD_OF_ONE"] = "1"
        out[mode] = {"collectives": tr_a.grads.stats["collectives"] - c0, "buckets": len(tr_a.grads._buckets), "gradient_rel_err": worst,
                     "in_backward": tr_a.grads.stats.get("in_backward", 0), "losses_finite": all(x == x and abs(x) < 1e30 for x in losses),
                     "loss_a_b_last": losses[-2:]}
    out["comm_log_kinds"] = sorted({ln.split()[1] for ln in COMM_LOG.tail(256)})
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
