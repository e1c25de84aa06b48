"""GPU tests of the optimizer / gradient-exchange plumbing around the hot path: per-parameter Adam step counts
(torch.optim.Adam semantics, ns/engine/optimizers.py:133-140), the multi-range Adam launch, the bucketed exchange's
bookkeeping on the GPU (side stream, single-contribution guard) and bench.py's multi-rank path (two ranks: on two GPUs
over RCCL when the box has them, otherwise time-slicing GPU 0 over gloo)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def test_adam_intermittent_gradients_match_torch(dev):
    """A parameter without a gradient in a step is skipped AND keeps its own step count (bias corrections), like
    torch.optim.Adam after zero_grad(set_to_none=True): proposal nets off-schedule, sub-fields without samples."""
    from presight_amd.dist import FlatGrads
    from presight_amd.optim import HipAdam

    torch.manual_seed(0)
    shapes = [(37, 3), (64,), (5, 5, 5), (1024, 2)]
    ref_p = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    hip_p = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=1e-2, eps=1e-15, weight_decay=1e-5)
    fg = FlatGrads(hip_p)
    opt = HipAdam(hip_p, lr=1e-2, eps=1e-15, weight_decay=1e-5, flat_grads=fg)
    gen = torch.Generator().manual_seed(1)
    schedule = [[0, 1, 2, 3], [0, 3], [0, 3], [0, 1, 3], [2], [0, 1, 2, 3], [0, 3]]  # parameter 1 ~ "proposal net", 2 ~ "idle sub-field"
    for touched in schedule:
        ref.zero_grad(set_to_none=True)
        fg.zero_()
        for i in touched:
            g = torch.randn(*shapes[i], generator=gen)
            ref_p[i].grad = g.clone()
            hip_p[i].grad.copy_(g.to(dev))
            hip_p[i]._ps_touched = True
        ref.step()
        opt.step()
    assert opt.steps == [6, 3, 3, 6]
    for a, b in zip(hip_p, ref_p):
        torch.testing.assert_close(a.detach().cpu(), b.detach(), rtol=2e-6, atol=2e-7)
    # an untouched parameter was not decayed either
    assert opt.state_dict()["steps"] == [6, 3, 3, 6]


def test_adam_ranges_launch_equals_single_range_launches(dev):
    """ps_adam_step_ranges (one launch, per-range bias corrections in the kernel argument) == ps_adam_step per range"""
    from presight_amd._lib import check, lib
    import ctypes

    torch.manual_seed(2)
    n = 4096 * 5 + 8
    p0, g = torch.randn(n, device=dev), torch.randn(n, device=dev)
    m0, v0 = torch.rand(n, device=dev) * 0.1, torch.rand(n, device=dev) * 0.01
    ranges = [(0, 4096, 1), (4096, 4, 7), (8192, 10000, 3), (4096 * 5, 8, 2)] + [(4096 * 5 - 4 * (k + 1) * 8, 28, 5 + k) for k in range(40)]
    a = [t.clone() for t in (p0, m0, v0)]
    b = [t.clone() for t in (p0, m0, v0)]
    s = torch.cuda.current_stream().cuda_stream
    for st, cnt, step in ranges:
        check(lib().ps_adam_step(a[0].data_ptr() + 4 * st, g.data_ptr() + 4 * st, a[1].data_ptr() + 4 * st, a[2].data_ptr() + 4 * st, cnt,
                                 1e-2, 0.9, 0.999, 1e-15, 1e-5, step, 1.0, s), "ps_adam_step")
    k = len(ranges)
    check(lib().ps_adam_step_ranges(b[0].data_ptr(), g.data_ptr(), b[1].data_ptr(), b[2].data_ptr(), k,
                                    (ctypes.c_int64 * k)(*[r[0] for r in ranges]), (ctypes.c_int64 * k)(*[r[1] for r in ranges]),
                                    (ctypes.c_int * k)(*[r[2] for r in ranges]), None, None, None, 0, 1e-2, 0.9, 0.999, 1e-15, 1e-5, 1.0, s),
          "ps_adam_step_ranges")
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert not torch.equal(a[0], p0)


def test_adam_grad_scale_factor_is_applied_before_the_weight_decay(dev):
    """The kernel's grad_scale argument (the reference's NON-default update_grad_scaler=True branch: GradScaler.step unscales the
    gradients before Adam adds the weight decay, ns/engine/optimizers.py:118-131): grad_scale = 1/1024 inside the kernel ==
    torch.optim.Adam on g / 1024 (with eps = 1e-15 Adam itself is scale invariant, the weight decay is not).  The DEFAULT path
    (grad_scale 1.0 on the scaled gradients, ns/engine/trainer.py:481-486) is pinned by the reference's own 24-iteration run:
    tests/test_hip_trainer.py."""
    from presight_amd.dist import FlatGrads
    from presight_amd.optim import HipAdam

    torch.manual_seed(3)
    ref_p = [torch.nn.Parameter(torch.randn(300, 4) * 3), torch.nn.Parameter(torch.randn(77))]
    hip_p = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=1e-2, eps=1e-15, weight_decay=1e-2)  # a large decay makes a wrong order of operations visible
    fg = FlatGrads(hip_p)
    opt = HipAdam(hip_p, lr=1e-2, eps=1e-15, weight_decay=1e-2, flat_grads=fg, grad_scale=1.0 / 1024)
    wrong = [p.detach().clone() for p in ref_p]
    for it in range(4):
        fg.zero_()
        for a, b in zip(ref_p, hip_p):
            g = torch.randn_like(a) * 1e-3
            a.grad = g.clone()
            b.grad.copy_((g * 1024).to(dev))
            b._ps_touched = True
        ref.step()
        opt.step()
    for a, b, w in zip(hip_p, ref_p, wrong):
        torch.testing.assert_close(a.detach().cpu(), b.detach(), rtol=3e-6, atol=3e-7)
    # the single-tensor entry point takes the same factor
    solo = HipAdam([torch.nn.Parameter(wrong[0].clone().to(dev))], lr=1e-2, eps=1e-15, weight_decay=1e-2, grad_scale=0.5)
    chk = torch.optim.Adam([torch.nn.Parameter(wrong[0].clone())], lr=1e-2, eps=1e-15, weight_decay=1e-2)
    g = torch.randn_like(wrong[0])
    solo.params[0].grad = (2 * g).to(dev)
    chk.param_groups[0]["params"][0].grad = g.clone()
    solo.step()
    chk.step()
    torch.testing.assert_close(solo.params[0].detach().cpu(), chk.param_groups[0]["params"][0].detach(), rtol=3e-6, atol=3e-7)


def test_adam_skips_groups_whose_device_flag_is_down(dev):
    """Device-decided groups (FlatGrads.define_groups): a group whose flag is 0 keeps parameters, both moments and its step
    count BIT-identical; flagged groups advance their own device-side step counts and match torch.optim.Adam."""
    from presight_amd.dist import FlatGrads
    from presight_amd.optim import HipAdam

    torch.manual_seed(4)
    shapes = [(64, 2), (33,), (128, 4), (7, 7), (50,)]
    ref_p = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    hip_p = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=1e-2, eps=1e-15, weight_decay=1e-5)
    fg = FlatGrads(hip_p)
    fg.define_groups([[hip_p[0], hip_p[1]], [hip_p[2], hip_p[3]]])  # two "sub-fields"; parameter 4 is host-decided
    opt = HipAdam(hip_p, lr=1e-2, eps=1e-15, weight_decay=1e-5, flat_grads=fg)
    gen = torch.Generator().manual_seed(5)
    live = [[0, 1], [1], [], [0], [0, 1], [1]]  # which groups received samples in each step
    for step, groups in enumerate(live):
        ref.zero_grad(set_to_none=True)
        fg.zero_()
        snap = [(p.detach().clone(), m.clone(), v.clone()) for p, m, v in zip(hip_p, opt.exp_avg, opt.exp_avg_sq)]
        for i in range(5):
            gid = {0: 0, 1: 0, 2: 1, 3: 1}.get(i)
            hip_p[i]._ps_touched = True  # the routed nodes mark every sub-field on the host; the device flag decides
            if gid is None or gid in groups:
                g = torch.randn(*shapes[i], generator=gen)
                ref_p[i].grad = g.clone()
                hip_p[i].grad.copy_(g.to(dev))
        for gid in groups:
            fg.group_flags[gid] = 1
        ref.step()
        opt.step()
        for i in range(4):
            if {0: 0, 1: 0, 2: 1, 3: 1}[i] not in groups:
                assert torch.equal(hip_p[i].detach(), snap[i][0]) and torch.equal(opt.exp_avg[i], snap[i][1]) and torch.equal(opt.exp_avg_sq[i], snap[i][2])
    assert opt.param_steps() == [3, 3, 4, 4, 6] and opt.state_dict()["steps"] == [3, 3, 4, 4, 6]
    for a, b in zip(hip_p, ref_p):
        torch.testing.assert_close(a.detach().cpu(), b.detach(), rtol=2e-6, atol=2e-7)
    # checkpoint round trip of the device-side step counts
    sd = opt.state_dict()
    fg.group_steps.zero_()
    opt.load_state_dict(sd)
    assert opt.param_steps() == [3, 3, 4, 4, 6]


def _tiny_model(dev, K=1, **conf_kw):
    import bench
    from presight_amd.model import NerfactoNuscMSModel, NerfactoNuscMSModelConfig

    torch.manual_seed(0)
    conf = NerfactoNuscMSModelConfig(near_plane=0.005, far_plane=50.0, piecewise_sampler_threshold=5.0, implementation="hip",
                                     use_lidar_loss=False, num_levels=2, features_per_level=2, log2_hashmap_size=12, base_res=16, max_res=64,
                                     hidden_dim=32, hidden_dim_color=32,
                                     proposal_net_args_list=[dict(features_per_level=1, log2_hashmap_size=12, num_levels=2, base_res=16,
                                                                  max_res=32, hidden_dim=32, use_linear=False),
                                                             dict(features_per_level=1, log2_hashmap_size=12, num_levels=2, base_res=16,
                                                                  max_res=64, hidden_dim=32, use_linear=False)], **conf_kw)
    scene = bench.make_scene(48, 2, K=K)
    model = NerfactoNuscMSModel(conf, num_train_cameras=48, num_train_videos=2, dino_to_rgb=None, centroids=scene["centroids"],
                                aabbs=scene["aabbs"]).to(dev)
    scene = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
    return model, scene


@pytest.mark.parametrize("K", [1, 4])
def test_trainer_buckets_on_gpu_single_rank(dev, K, monkeypatch):
    """The trainer's bucketed exchange armed without a process group (PRESIGHT_DRY_OVERLAP: launches are bookkeeping only).  Flat
    order = completion order: proposal network 1, proposal network 0 (side stream), the fields' small parameters (complete before
    the main table backward starts), then the main hash table as 2 level-group pieces (K = 1) / the K tables in sub-field groups
    (routed tile), one accumulate launch per piece.  EVERY bucket is handed over during backward, strictly in bucket order; every
    parameter receives exactly one contribution, and a second in-place contribution after the hand-over raises."""
    import bench
    from presight_amd.ops import mark_touched

    monkeypatch.setenv("PRESIGHT_DRY_OVERLAP", "1")
    model, scene = _tiny_model(dev, K=K)
    tr = bench.Trainer(model, scene, 1)
    fg = tr.grads
    n_table = 2 if K == 1 else 4  # tiny model: 2 levels -> 2 pieces; K = 4 tables -> 4 groups of one
    assert tr.group_names == ["proposal_networks", "proposal_networks", "fields"] + ["fields"] * (1 if K == 1 else n_table) + ["fields"]
    assert len(fg._buckets) == 4 + n_table and fg.dry
    if K == 1:
        assert [b["part"] for b in fg._buckets[3:5]] == [(0, 2), (1, 2)]
    order = []
    launch = fg._launch
    fg._launch = lambda b: (order.append((b["index"], fg._in_finish)), launch(b))[1]
    batches = bench.make_batches(scene, dev, 2, 0, rays=512)
    fg.record_timeline = True
    for i in range(2):
        order.clear()
        tr.step(batches[i])
        assert order == [(j, False) for j in range(4 + n_table)], order  # bucket order, all of them BEFORE finish_exchange
    assert all(b["launched"] and b["phase"] == "backward" for b in fg._buckets)
    tl = fg.timeline_summary()
    assert len(tl) == 4 + n_table and all(t["handed_over_in_backward"] == 2 for t in tl)
    assert all(t["ms_before_backward_end"] >= 0.0 for t in tl)
    assert tl[2]["ms_before_backward_end"] > tl[2 + n_table]["ms_before_backward_end"]  # the MLP bucket is ready before the last table piece
    with pytest.raises(RuntimeError, match="second gradient"):
        mark_touched([fg.params[0]])
    assert all(s == 2 for s in tr.opt.param_steps())
    # off-schedule step: proposal networks get no gradient -> their buckets are skipped, their Adam step count stays
    tr.update_props_every_step = False
    tr.step_idx = 50000
    model.proposal_sampler.step_cb(50000)  # past the "first 10 steps always update" rule of ray_samplers.py:586
    model.proposal_sampler._steps_since_update = 0
    order.clear()
    tr.step(batches[0])
    assert [j for j, _ in order] == list(range(2, 4 + n_table))
    n_prop = fg.bucket_params[1][1]
    steps = tr.opt.param_steps()
    assert all(s == 2 for s in steps[:n_prop]) and all(s == 3 for s in steps[n_prop:])


@pytest.mark.parametrize("K", [1, 4])
def test_pipelined_adam_equals_the_plain_step(dev, K):
    """Trainer.pipeline_adam (single process, opt-in): the proposal networks' Adam right behind backward, the fields' Adam + the
    clearing of their gradients on a second stream, the compute stream waiting for it at model.param_gate("fields").  On the SAME
    gradients (two steps' worth, written into the flat buffers; for the routed tile with some device-decided sub-field groups flagged
    and some not) the pipelined optimizer leaves parameters, both moments, every step count and the cleared gradient buffer
    BIT-identical to the plain one."""
    import bench
    from presight_amd.ops import mark_touched

    results = []
    for pipelined in (False, True):
        model, scene = _tiny_model(dev, K=K)
        tr = bench.Trainer(model, scene, 1, fused_table_adam=False)  # (the pipelined step is the alternative to the fused table update)
        tr.pipeline_adam = pipelined
        fg = tr.grads
        g = torch.Generator().manual_seed(7)
        for it in range(2):
            pipe = tr._begin_step()
            torch.cuda.synchronize()
            assert float(fg.flat.abs().max()) == 0.0  # whoever was responsible has cleared every gradient
            touched = [p for i, p in enumerate(fg.params) if not (it == 1 and tr.group_names[_bucket_of(fg, i)] == "proposal_networks")]
            for p in touched:  # (the padding between parameters is never written, as in training)
                p.grad.copy_((torch.randn(p.shape, generator=g) * 1e-3).to(dev))
            mark_touched(touched, groups_on_device=True)  # (second step: proposal networks off schedule)
            if fg.n_groups:
                flags = torch.zeros(fg.n_groups, dtype=torch.int32)
                flags[::2] = 1  # every other sub-field "received samples"
                fg.group_flags.copy_(flags.to(dev))
            tr._optimizer_step(pipe)
        if pipelined:
            assert tr._pipe is not None and tr._pipe["event"] is not None
            model.param_gate("fields")  # the compute stream waits for the fields' piece here
        p_now = tr.opt.flat[0].clone()  # (read on the compute stream, behind the gate)
        tr._begin_step()
        torch.cuda.synchronize()
        results.append((p_now, tr.opt.flat[2].clone(), tr.opt.flat[3].clone(), tr.opt.param_steps(), fg.flat.clone()))
    a, b = results
    assert a[3] == b[3] and len(set(a[3])) > 1
    for j in (0, 1, 2, 4):
        assert torch.equal(a[j], b[j]), j
    assert float(a[4].abs().max()) == 0.0


def _bucket_of(fg, i):
    off = fg.offsets[i]
    return next(j for j, (a0, a1) in enumerate(fg.bucket_ranges) if a0 <= off < a1)


def _run_bench_two_ranks(extra, timeout=300, attempts=2, min_buckets=4):
    """bench.py --gpus 2 in a fresh child process.  A run that hangs or times out is NEVER a skip: the per-rank collective logs
    (presight_amd.dist.CommLog) are compared first -- different issued sequences = an ordering bug in the exchange = failure at
    once --, then the run is repeated ONCE in a new process (the two-ranks-on-one-GPU-over-gloo harness hung twice in ~125 runs
    with identical logs on both ranks); a second hang fails the test."""
    two_gpus = torch.cuda.device_count() >= 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if not two_gpus:  # functional run on a one-GPU box: both ranks share GPU 0, host-staged gloo transport
        env.update(PRESIGHT_SINGLE_DEVICE="1", PRESIGHT_DIST_BACKEND="gloo")
    # a rank that is still running after 200 s writes every thread's stack to gpurun_out/ and exits
    dump_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(dump_dir, exist_ok=True)
    dump = os.path.join(dump_dir, "dp2_hang_rank{rank}.txt")
    comm = os.path.join(dump_dir, "dp2_comm_rank{rank}.log")
    env.update(PRESIGHT_HANG_DUMP="200", PRESIGHT_HANG_DUMP_FILE=dump, PRESIGHT_COMM_LOG=comm)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rays", "4096",
           "--no-cpu-baseline"] + extra
    path = lambda pat, r: pat.replace("{rank}", str(r))  # noqa: E731
    read_logs = lambda: [open(path(comm, r)).read().splitlines() if os.path.exists(path(comm, r)) else [] for r in (0, 1)]  # noqa: E731
    failures = []
    for attempt in range(attempts):
        for r in (0, 1):
            for pat in (dump, comm):
                if os.path.exists(path(pat, r)):
                    os.remove(path(pat, r))
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)
            rc, tail = res.returncode, res.stdout[-1500:] + res.stderr[-3000:]
        except subprocess.TimeoutExpired as e:
            rc, tail = None, f"timeout after {timeout} s\n{(e.stderr or b'')[-2000:]}"
        hung = [r for r in (0, 1) if os.path.exists(path(dump, r)) and os.path.getsize(path(dump, r)) > 0]
        if rc == 0:
            break
        logs = read_logs()
        k = min(len(logs[0]), len(logs[1]))
        assert logs[0][:k] == logs[1][:k], (f"two-rank run hung / failed with DIFFERENT collective sequences on the two ranks (first difference at line "
                                            f"{next(i for i in range(k) if logs[0][i] != logs[1][i])}): {tail}")
        assert rc is None or hung, tail  # an ordinary failure (non-zero exit without a hang) is a failure
        stacks = open(path(dump, hung[0])).read()[:1500] if hung else ""
        failures.append(f"attempt {attempt}: hang / timeout, collective logs agree over {k} lines ({len(logs[0])} / {len(logs[1])} issued)\n{stacks}\n{tail}")
    else:
        pytest.fail("two-rank run hung in every attempt:\n" + "\n".join(failures))
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["comm"]["ranks"] == 2
    assert line["comm"]["backend"] == ("nccl" if two_gpus else "gloo")
    logs = read_logs()
    assert logs[0] == logs[1] and len(logs[0]) > 10, "the ranks issued different collective sequences"
    # overlap by construction: of the gradient buckets of a step, all but (at most) the last leave while backward is still running
    # (a record bucket of the sparse exchange shows up as its first collective, the MAX all-reduce of the per-level maxima)
    grad = [ln for ln in logs[0] if (" all_reduce " in ln or " reduce_scatter " in ln or " all_reduce_max_levels " in ln) and "bucket=" in ln
            and "bucket=-" not in ln]
    steps_seen = sorted({ln.split("step=")[1].split()[0] for ln in grad})
    for st in steps_seen[1:-1]:
        mine = [ln for ln in grad if f"step={st} " in ln + " "]
        assert len(mine) >= min_buckets and sum("phase=backward" in ln for ln in mine) >= len(mine) - 1, mine
    for r in (0, 1):
        os.remove(path(comm, r))
    line["_hang_retries"] = len(failures)
    return line


def test_bench_two_ranks_allreduce():
    """bench.py --gpus 2 starts its own two ranks; replicas stay bit-identical after bucketed, overlapped all-reduces"""
    line = _run_bench_two_ranks(["--exchange", "allreduce"])
    assert line["replicas_max_abs_diff"] == 0.0
    assert line["config"]["rays_per_gpu"] == 4096 and line["scaling"] == "weak"
    assert line["other_scaling"]["scaling"] == "strong" and line["other_scaling"]["rays_per_gpu"] == 2048


def test_bench_two_ranks_sharded_strong():
    """reduce-scatter + Adam on the owned shard + all-gather left in flight under the next step's sampling; strong scaling
    splits the global batch R // world (ns/data/PreSight/my_datamanager.py:203-212)"""
    line = _run_bench_two_ranks(["--exchange", "sharded", "--scaling", "strong", "--global-depth-clip"])
    assert line["replicas_max_abs_diff"] == 0.0
    assert line["config"]["rays_per_gpu"] == 2048 and line["config"]["exchange"] == "sharded"


def test_bench_two_ranks_sparse_record_exchange():
    """exchange = sparse (SURVEY.md 8e "sparse exchange of touched rows"): the hash tables' gradients travel as the binned backward's
    records to the owner of their table slice (all-to-all), the owner accumulates both ranks' runs in int64 and updates its shard; MLP
    gradients are reduce-scattered as in the sharded mode; the parameters return by the same all-gather.  Replicas stay bit-identical
    and the line reports the record bytes on the link next to the dense table gradient's."""
    # (an off-schedule step exchanges three buckets: the main MLPs, the main tables' records -- ONE bucket, not level-group pieces --, the tail)
    line = _run_bench_two_ranks(["--exchange", "sparse", "--scaling", "strong"], min_buckets=3)
    assert line["replicas_max_abs_diff"] == 0.0
    assert line["config"]["exchange"] == "sparse"
    rec, dense = line["comm"]["record_bytes_on_link_per_rank_per_step"], line["comm"]["dense_table_gradient_bytes"]
    # records on the link: (N - 1) / N of 4 x-pair records of (2 + F) words per (point, level) -- 2048 rays per rank at cfg 2: ~140 MB, MORE
    # than this small model's dense gradient (134 MB): the record exchange pays off where the tables dwarf the batch (production tile at
    # 8192 rays per rank: ~1 GB of records against 3.3 GB of dense reduce-scatter traffic), the line reports both figures
    pts = 2048 * (64 * 16 * 4 * 16 + 128 * 8 * 4 * 12 + 64 * 8 * 4 * 12)
    assert 0.3 * pts / 2 < rec <= pts / 2 * 1.001 and abs(dense / (4.0 * (16 * 2 ** 19 * 2 + 2 * 8 * 2 ** 20)) - 1.0) < 1e-3, (rec, pts, dense)


@pytest.mark.parametrize("K", [1, 4])
def test_rccl_group_of_one_runs_the_exchange(K):
    """The bucketed exchange through RCCL itself on a one-GPU box: a process group of ONE rank (backend nccl = RCCL) with
    PRESIGHT_EXCHANGE_WORLD_OF_ONE=1, so that every bucket all-reduce / reduce-scatter, the parameter all-gather of the sharded mode and
    the routed tile's flag all-reduce are issued for real -- on the side stream, during backward, behind the two-stream backward's
    events -- and the exchanged gradient buffer (sum over one rank / 1) must equal that of a trainer that exchanges nothing (to the 1e-6
    of the float atomics in a few per-ray gradients; Adam with eps = 1e-15 amplifies that noise, so gradients are compared, not parameters)."""
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("PRESIGHT_DIST_BACKEND", "PRESIGHT_SINGLE_DEVICE")}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
               PRESIGHT_EXCHANGE_WORLD_OF_ONE="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_group_of_one.py"), str(K)], capture_output=True, text=True,
                         env=env, timeout=300)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["backend"] == "nccl"
    for mode in ("allreduce", "sharded", "sparse"):
        # 3 steps x (>= 5 gradient buckets [+ the parameter all-gathers of the sharded mode]); all but the tail handed over in backward
        assert out[mode]["collectives"] >= 3 * out[mode]["buckets"] * (2 if mode != "allreduce" else 1), out
        assert out[f"{mode}_tables_bit_equal"], out  # (integer accumulation: the exchanged table gradient equals the plain one exactly)
        assert out[mode]["in_backward"] >= 3 * (out[mode]["buckets"] - 1) and out[mode]["gradient_rel_err"] < 1e-4, out
        assert out[mode]["losses_finite"] and abs(out[mode]["loss_a_b_last"][0] - out[mode]["loss_a_b_last"][1]) < 0.05 * abs(out[mode]["loss_a_b_last"][1]), out
    want = ({"all_reduce", "reduce_scatter", "all_gather_params", "all_to_all_records", "all_to_all_item_runs", "all_reduce_max_levels"}
            | ({"all_reduce_max_flags"} if K > 1 else set()))
    assert want <= set(out["comm_log_kinds"]), out


def test_chunk_feed_yields_the_reference_loader_batches(dev):
    """ChunkFeed over a device-resident chunk == the reference's DataLoader(ImageChunk, DistributedSampler) batch for batch
    (tests/golden/datafeed.npz: ImageChunk.__getitem__ + default collate, generated from the reference), for one rank of one and
    of three; the next chunk is swapped in when the pass is exhausted (drop_last)"""
    from conftest import load_golden, t
    from presight_amd.datafeed import ChunkFeed

    G = load_golden("datafeed")
    raw = {k: t(G[k]) for k in ("rgbs", "skies", "depths", "features", "pixel_indices", "image_indices", "video_ids", "widths")}
    calls = []

    def load(i):
        calls.append(i)
        return raw  # host tensors: pinned + uploaded on the feed's side stream

    for world, rank in ((1, 0), (3, 1)):
        calls.clear()
        feed = ChunkFeed(load, batch_size=96, device=dev, world=world, rank=rank, seed=0)
        n = int(G[f"n_batches_w{world}r{rank}"])
        for b in range(n):
            got = feed.next_batch()
            for name, key in (("rgb", "rgb"), ("sky", "sky"), ("depth", "depth"), ("features", "features"), ("video_id", "video_id"),
                              ("ray_index", "ray_indices")):
                ref = t(G[f"b_{name}_w{world}r{rank}"])[b]
                assert torch.equal(got[key].cpu().reshape(ref.shape).to(ref.dtype), ref), (world, rank, b, name)
        assert feed.chunks_loaded == 1
        first_of_next = feed.next_batch()  # the pass is exhausted (drop_last): chunk 1 takes over, same data -> same first batch
        assert feed.chunks_loaded == 2 and feed.chunk_index == 1
        assert torch.equal(first_of_next["rgb"].cpu(), t(G[f"b_rgb_w{world}r{rank}"])[0])
        feed.close()
        assert calls[:2] == [0, 1]


@pytest.mark.parametrize("K", [1, 4])
def test_fused_table_adam_is_the_separate_step_bit_for_bit(dev, K):
    """Trainer(world=1) applies the hash tables' Adam step inside their table backward (ps_grid_scatter_binned_adam, routed tile:
    ps_grid_scatter_binned_ms_adam): from equal parameters, ONE iteration leaves every table, both its moments and its step count
    bit-equal to backward -> optimizer.step() (the table gradient is integer-accumulated, both paths run csrc/adam_core.hpp), the
    fused path never writes the tables' gradients, a backward pass outside Trainer.step is left alone (plain gradients), and a second
    gradient contribution to an already-updated table raises instead of being lost."""
    import bench
    from presight_amd import field_ops as FO

    runs = []
    for fused in (True, False):
        model, scene = _tiny_model(dev, K=K)
        tr = bench.Trainer(model, scene, 1, fused_table_adam=fused)
        assert tr.fused_table_adam == fused
        batches = bench.make_batches(scene, dev, 2, 0, rays=512)
        torch.manual_seed(11)
        tr.step(batches[0])
        runs.append((model, tr, batches))
    (ma, ta, batches), (mb, tb, _) = runs
    names = {id(p): n for n, p in ma.named_parameters()}
    tables = [i for i, p in enumerate(ta.opt.params) if names[id(p)].endswith("hash_table")]
    assert len(tables) == 3 * K and all(getattr(ta.opt.params[i], "_ps_fused_adam", None) is ta.opt for i in tables)
    assert ta.opt.param_steps() == tb.opt.param_steps()
    moved = 0
    for i in tables:
        assert torch.equal(ta.opt.params[i], tb.opt.params[i]), names[id(ta.opt.params[i])]
        assert torch.equal(ta.opt.exp_avg[i], tb.opt.exp_avg[i]) and torch.equal(ta.opt.exp_avg_sq[i], tb.opt.exp_avg_sq[i])
        assert float(ta.opt.params[i].grad.abs().max()) == 0.0
        moved += int(float(ta.opt.exp_avg_sq[i].abs().max()) > 0)
    assert moved >= 3  # (K = 4: a sub-field the 512 rays never reach stays untouched on both sides)
    # a backward pass the trainer has not armed: plain table gradients, nothing is updated
    from presight_amd import ops
    from presight_amd.rays import RayBundle

    b = batches[1]
    ta.grads.zero_()
    before = [ta.opt.params[i].detach().clone() for i in tables]
    o, d, pa, dn = ops.generate_rays(b["ray_indices"], ta.scene["c2w"], ta.scene["fx"], ta.scene["fy"], ta.scene["cx"], ta.scene["cy"])
    rb = RayBundle(o, d, pa, camera_indices=b["ray_indices"][:, 0:1], metadata={"video_id": b["video_ids"][:, None], "directions_norm": dn})
    ma.train()
    sum(ma.get_loss_dict(ma(rb), b).values()).backward()
    assert all(torch.equal(x, ta.opt.params[i]) for x, i in zip(before, tables))
    assert sum(float(ta.opt.params[i].grad.abs().max()) > 0 for i in tables) >= 3
    # a table that has been updated inside its backward cannot take another contribution in the same step
    ta.opt.params[tables[0]]._ps_fused_done = True
    with pytest.raises(RuntimeError, match="second gradient contribution"):
        FO._refuse_second_contribution([ta.opt.params[tables[0]]])
    ta.grads.zero_()
    assert ta.opt.params[tables[0]]._ps_fused_done is False


def test_shared_proposal_network_keeps_the_separate_table_update(dev):
    """use_same_proposal_network=True (a reference-supported config, nerfacto_nusc_ms.py:263): the one proposal network is evaluated in
    both proposal iterations, so its table receives TWO gradient contributions per step -- it must stay out of the fused table update
    (which needs exactly one); the default single-process Trainer steps, the main table is still fused, and the result equals the
    trainer with the fused update off (tables bit-equal after the first iteration: integer-accumulated gradients, shared element update)."""
    import bench

    runs = []
    for fused in (None, False):
        model, scene = _tiny_model(dev, use_same_proposal_network=True)
        assert len(model.proposal_networks) == 1
        tr = bench.Trainer(model, scene, 1, fused_table_adam=fused)
        batches = bench.make_batches(scene, dev, 2, 0, rays=512)
        torch.manual_seed(11)
        tr.step(batches[0])
        runs.append((model, tr, batches))
    (ma, ta, batches), (mb, tb, _) = runs
    assert ta.fused_table_adam and not tb.fused_table_adam
    names = {id(p): n for n, p in ma.named_parameters()}
    fused = [names[id(p)] for p in ta.opt.params if getattr(p, "_ps_fused_adam", None) is ta.opt]
    assert len(fused) == 1 and fused[0].startswith("field."), fused
    assert ta.opt.param_steps() == tb.opt.param_steps()
    for i, p in enumerate(ta.opt.params):
        if names[id(p)].endswith("hash_table"):
            assert torch.equal(p, tb.opt.params[i]), names[id(p)]
            assert float(ta.opt.exp_avg_sq[i].abs().max()) > 0, names[id(p)]
    torch.manual_seed(12)
    ta.step(batches[1])  # (and a second iteration: the proposal table's two contributions accumulate, nothing raises)


def test_an_iteration_that_raises_after_a_fused_update_blocks_the_trainer(dev):
    """The fused table update is applied DURING backward: if the iteration raises afterwards, tables are one optimizer step ahead of
    every other parameter.  The trainer refuses to continue silently (clear_failure() after restoring a checkpoint re-enables it)."""
    import bench

    model, scene = _tiny_model(dev)
    tr = bench.Trainer(model, scene, 1)
    assert tr.fused_table_adam
    batches = bench.make_batches(scene, dev, 2, 0, rays=512)
    tr.step(batches[0])
    real = tr.grads.finish_exchange

    def boom():
        raise ValueError("injected")

    tr.grads.finish_exchange = boom
    with pytest.raises(ValueError, match="injected"):
        tr.step(batches[1])
    tr.grads.finish_exchange = real
    assert not tr.opt.fused_armed
    with pytest.raises(RuntimeError, match="one step apart"):
        tr.step(batches[1])
    tr.clear_failure()
    tr.step(batches[1])


def test_second_trainer_takes_the_routed_groups_over(dev):
    """two Trainers on one routed model: the newer one owns the device-decided sub-field groups; the older one raises instead of
    silently never updating the routed sub-fields"""
    import bench

    model, scene = _tiny_model(dev, K=4)
    t1 = bench.Trainer(model, scene, 1)
    batches = bench.make_batches(scene, dev, 1, 0, rays=512)
    t1.step(batches[0])
    t2 = bench.Trainer(model, scene, 1)
    with pytest.raises(RuntimeError, match="taken over"):
        t1.step(batches[0])
    t2.step(batches[0])
