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
