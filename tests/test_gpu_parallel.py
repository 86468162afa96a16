"""Data-parallel step with the real HIP kernels: two processes share cuda:0 and exchange gradients over gloo (RCCL needs
one GPU per rank, which the test box does not have).  Checks that DP training on a split batch equals single-process
training on the whole batch, for every gradient path (autograd accumulate, direct deposit, side stream)."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CFG = dict(patch_size=4, window_size=64, shift_size=32, shift_strategy="nest_roll", rel_pos_bias="flat", embed_dim=64,
           depths=[2, 2], num_heads=[2, 4], drop_path_rate=0.0)
SPEC = dict(dim_in=12 * 32 * 32, f_in=3, f_out=12, base_pix=12, class_names=[])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    sys.path.insert(0, ROOT)
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch.swin_hp_transformer import SwinHPTransformerConfig, SwinHPTransformerSys
    torch.manual_seed(11)
    m = SwinHPTransformerSys(SwinHPTransformerConfig(**CFG), DataSpec(**SPEC)).to("cuda:0")
    m.compute_dtype = torch.bfloat16
    return m


def _data():
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 256, (4, 3, SPEC["dim_in"]), generator=g).float()
    y = torch.randint(0, 12, (4, SPEC["dim_in"]), generator=g)
    return x.to("cuda:0"), y.to("cuda:0")


def _train(model, dp, xs, ys, steps=2):
    from heal_swin_amd.losses import seg_loss
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    for _ in range(steps):
        dp.zero_grad()
        seg_loss(model(xs), ys).backward()
        dp.finish()
        opt.step()
    torch.cuda.synchronize()
    return [p.detach().float().cpu().numpy().copy() for p in model.parameters()]


def _worker(rank, world, port, kw, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from heal_swin_amd.parallel import GradBucketAllReduce
    model = _build()
    x, y = _data()
    dp = GradBucketAllReduce(model.parameters(), bucket_bytes=256 << 10, **kw)  # several buckets
    assert dp.world == 2 and len(dp.buckets) > 1
    out = _train(model, dp, x.chunk(world)[rank], y.chunk(world)[rank])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kw", [dict(direct_wgrad=False), dict(direct_wgrad=True), dict(async_wgrad=True)])
def test_two_rank_dp_equals_single_process(kw):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kw, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single process on the full batch (mean CE over the whole batch == mean of the two half-batch means)
    sys.path.insert(0, ROOT)
    from heal_swin_amd.parallel import GradBucketAllReduce
    model = _build()
    x, y = _data()
    dp = GradBucketAllReduce(model.parameters(), direct_wgrad=False)
    ref = _train(model, dp, x, y)
    dp.remove()
    import numpy as np
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)  # replicas identical
    worst = 0.0
    for a, b in zip(res[0], ref):
        worst = max(worst, float(np.abs(a - b).max()) / max(1e-3, float(np.abs(b).max())))
    assert worst < 2e-2, worst  # bf16 activations: half-batch vs full-batch reduction order differs


def _rccl_worker(port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    sys.path.insert(0, ROOT)
    from heal_swin_amd.parallel import GradBucketAllReduce
    model = _build()
    x, y = _data()
    dp = GradBucketAllReduce(model.parameters(), bucket_bytes=256 << 10, exchange_single_rank=True)
    assert len(dp.buckets) > 1 and dp._exchange
    launched = []
    real = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (launched.append(1), real(*a, **k))[1]
    out = _train(model, dp, x, y)
    dist.all_reduce = real
    assert len(launched) == 2 * len(dp.buckets), (len(launched), len(dp.buckets))  # every bucket, both steps
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_backend_single_rank_matches_local():
    """The RCCL ("nccl") code path of bench.py --gpus N: asynchronous bucket all-reduces on RCCL's stream, launched while
    the backward is still depositing later buckets.  One rank is all a 1-GPU box allows; the result must equal the
    process-group-free run bit for bit (sum over one rank, scale 1)."""
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from heal_swin_amd.parallel import GradBucketAllReduce
    model = _build()
    x, y = _data()
    dp = GradBucketAllReduce(model.parameters(), bucket_bytes=256 << 10)
    ref = _train(model, dp, x, y)
    dp.remove()
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)


def test_bench_script_multi_rank_path_on_one_gpu():
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, RANK / LOCAL_RANK / WORLD_SIZE from the
    environment, barrier + max-over-ranks timing, one JSON line from rank 0) -- with the two ranks sharing this box's single
    GPU over gloo (HS_BENCH_SHARED_GPU=1) because RCCL needs one GPU per rank."""
    import json
    import subprocess
    env = dict(os.environ, HS_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "tiny"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0
    assert out["scaling"] == "weak" and out["config"]["global_batch"] == 2 * out["config"]["batch_per_gpu"]
    assert out["config"]["parallelism"] == "dp2" and "cpu_baseline" not in out and "roofline" in out


def _ddp_worker(rank, world, port, dtype_name, q):
    """The reference's own trainer wiring (train.py:182-189: Lightning `ddp` = one process per GPU, the module wrapped in
    torch.nn.parallel.DistributedDataParallel with find_unused_parameters=False; training/optimizer.py:57-66: torch.optim.Adam)
    around the drop-in module -- no GradBucketAllReduce, no FlatAdam, no fused loss."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from torch.nn.parallel import DistributedDataParallel
    model = _build()
    model.compute_dtype = getattr(torch, dtype_name)
    ddp = DistributedDataParallel(model, device_ids=[0], find_unused_parameters=False)
    x, y = _data()
    out = _train_torch(ddp, model, x.chunk(world)[rank], y.chunk(world)[rank])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _train_torch(callable_model, model, xs, ys, steps=3):
    loss_fn = torch.nn.CrossEntropyLoss()  # model_lightning_swin_hp.py:39-45
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(steps):
        opt.zero_grad()  # (set_to_none=True: the PyTorch / Lightning default)
        loss = loss_fn(callable_model(xs), ys.long())
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    return losses, [p.detach().float().cpu().numpy().copy() for p in model.parameters()]


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_torch_ddp_wrapper_with_torch_adam_equals_single_process(dtype_name):
    """INTEGRATION.md level 1 as the reference's trainer would run it: SwinHPTransformerSys inside torch's DistributedDataParallel
    (two ranks on this box's one GPU, gloo), nn.CrossEntropyLoss, torch.optim.Adam, three steps.  The replicas must stay identical
    and equal single-process training on the whole batch (to the reduction-order noise of the half batches)."""
    import numpy as np
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, dtype_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model = _build()
    model.compute_dtype = getattr(torch, dtype_name)
    x, y = _data()
    ref_losses, ref = _train_torch(model, model, x, y)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b)  # replicas identical after three steps
    # the mean of the two half-batch losses is the full-batch loss (equal halves)
    tol_l = 2e-5 if dtype_name == "float32" else 5e-3
    for s, (l0, l1, lr) in enumerate(zip(res[0][0], res[1][0], ref_losses)):
        assert abs(0.5 * (l0 + l1) - lr) <= tol_l * abs(lr), (s, l0, l1, lr)
    # Adam normalises every update to ~lr whatever the gradient's size, so a gradient element inside the rounding noise moves by a
    # different amount in the two runs: the comparison is on the MOVEMENT of three steps as a whole (cosine), and in fp32 -- where
    # the noise is the reduction order of fp32 sums -- also on its worst element
    init = [p.detach().float().cpu().numpy().copy() for p in _build().parameters()]
    d_ddp = np.concatenate([(a - i).ravel() for a, i in zip(res[0][1], init)]).astype(np.float64)
    d_ref = np.concatenate([(b - i).ravel() for b, i in zip(ref, init)]).astype(np.float64)
    cos = float(d_ddp @ d_ref / (np.linalg.norm(d_ddp) * np.linalg.norm(d_ref)))
    assert np.linalg.norm(d_ref) > 0
    assert cos >= (0.999 if dtype_name == "float32" else 0.97), cos
    if dtype_name == "float32":
        worst = float(np.abs(d_ddp - d_ref).max() / np.abs(d_ref).max())
        assert worst <= 0.1, worst


def test_bench_script_eight_ranks_on_one_gpu():
    """The first real 8-GPU run must not be the first time eight ranks meet: bench.py --gpus 8 as the driver launches it, the eight
    ranks sharing this box's GPU over gloo at the tiny size, bf16 wire format.  Checks what a mis-wired exchange would break:
    every rank arrives (max-over-ranks timing of 8 entries), the bucket order / hooks do not deadlock, the GEMM policy is agreed by
    all ranks (a MAX all-reduce: one dissenting rank would hang the next collective), the timeline diagnostic is there, and the
    line says rccl_ranks == 8.  The strong-scaling reading (global batch 8 -> 1 per rank) runs the same way."""
    import json
    import subprocess
    env = dict(os.environ, HS_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra, per_gpu, scaling in ((["--batch", "1"], 1, "weak"), (["--batch", "8", "--strong-scaling"], 1, "strong")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
               "--workload", "tiny", "--comm-dtype", "bf16"] + extra
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        out = json.loads(lines[0])
        assert out["n_gpus"] == 8 and out["value"] > 0 and out["scaling"] == scaling
        assert out["config"]["parallelism"] == "dp8" and out["config"]["batch_per_gpu"] == per_gpu and out["config"]["global_batch"] == 8
        rc = out["rccl"]
        assert rc["rccl_ranks"] == 8 and len(rc["step_ms_per_rank"]) == 8 and len(rc["allreduce_ms_per_step_standalone_per_rank"]) == 8
        assert rc["buckets"] >= 1 and rc["allreduce_bytes_per_step"] > 0 and "bf16" in rc["exchange"]
        assert rc["gemm_policy"] is not None and set(rc["gemm_policy"]["trial_ms_per_step"]) == {"own", "per_shape_with_library"}
        tl = rc["bucket_timeline_rank0"]
        assert isinstance(tl, list) and len(tl) >= rc["buckets"] and all("ms" in e and "launched_from" in e for e in tl), tl


def test_bench_script_drop_in_two_ranks_on_one_gpu():
    """`bench.py --drop-in --gpus 2`: the level-1 integration as a multi-rank run -- torch's DistributedDataParallel around the module, no gradient
    sink -- through the same launcher path as the driver's runs (two ranks sharing the GPU over gloo)."""
    import json
    import subprocess
    env = dict(os.environ, HS_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "tiny", "--drop-in"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and "drop-in" in out["config"]["step"] and "DistributedDataParallel" in out["config"]["step"]
    assert out["rccl"]["rccl_ranks"] == 2 and "DistributedDataParallel" in out["rccl"]["exchange"] and out["rccl"]["gemm_policy"] is None


def _graphed_dp_worker(rank, world, port, graphed, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from heal_swin_amd.graphs import GraphedTrainStep
    from heal_swin_amd.losses import seg_loss
    from heal_swin_amd.optim import FlatAdam
    from heal_swin_amd.parallel import GradBucketAllReduce
    model = _build().train()
    x, y = _data()
    xs, ys = x.chunk(world)[rank], y.chunk(world)[rank]
    dp = GradBucketAllReduce(model.parameters(), bucket_bytes=256 << 10)
    opt = FlatAdam(model.parameters(), dp, lr=1e-3, model=model)
    losses = []
    if graphed:
        step = GraphedTrainStep(model, lambda out, t: seg_loss(out, t), opt, xs, ys, warmup=2, grad_sink=dp)
        for _ in range(3):
            losses.append(float(step(xs, ys)))
    else:
        for _ in range(2 + 3):  # the graphed run's two warm-up steps are real training steps too
            dp.zero_grad()
            loss = seg_loss(model(xs), ys)
            loss.backward()
            dp.finish()
            opt.step()
            losses.append(float(loss.detach()))
        losses = losses[2:]
    torch.cuda.synchronize()
    q.put((rank, losses, [p.detach().float().cpu().numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_graphed_train_step_under_data_parallelism_equals_the_eager_dp_step():
    """GraphedTrainStep with world > 1: [graph: zero_grad, forward, loss, backward into the local buckets] -> eager bucket exchange ->
    [graph: FlatAdam].  Two ranks on one GPU over gloo: same losses and bit-identical parameters as the eager data-parallel step (the
    exchange sums the same buckets in the same order; only its timing relative to the backward differs), replicas identical."""
    import numpy as np
    import torch.multiprocessing as mp
    res = {}
    for graphed in (False, True):
        world, port = 2, _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_graphed_dp_worker, args=(r, world, port, graphed, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=300) for _ in range(world)]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        res[graphed] = {r: (l, ps) for r, l, ps in got}
    for graphed in (False, True):
        for a, b in zip(res[graphed][0][1], res[graphed][1][1]):
            assert np.array_equal(a, b), "replicas diverged"
    for r in (0, 1):
        assert res[True][r][0] == res[False][r][0], (res[True][r][0], res[False][r][0])
    for a, b in zip(res[True][0][1], res[False][0][1]):
        assert np.array_equal(a, b)


def test_bench_script_graph_replay_with_two_ranks_on_one_gpu():
    """`bench.py --graph --gpus 2`: the step as two HIP graphs around the eager bucket exchange, through the driver's launcher path (two
    ranks sharing the GPU over gloo); the line says so and the loss is the eager run's."""
    import json
    import subprocess
    env = dict(os.environ, HS_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = {}
    for extra in ([], ["--graph"]):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--workload", "tiny"] + extra
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        outs[bool(extra)] = json.loads(lines[0])
    g, e = outs[True], outs[False]
    assert g["config"]["launch"] == "hip graph replay" and e["config"]["launch"] == "eager" and g["n_gpus"] == 2
    assert g["rccl"]["rccl_ranks"] == 2 and g["rccl"]["gemm_policy"] is not None
    # the graphed run takes 1 (graph warm-up) step more before its timed region than the eager one: compare loosely
    assert abs(g["config"]["final_loss"] - e["config"]["final_loss"]) < 0.05 * abs(e["config"]["final_loss"]), (g["config"]["final_loss"], e["config"]["final_loss"])
