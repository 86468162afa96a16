"""Data-parallel gradient exchange (heal_swin_amd.parallel) on CPU with the gloo backend, world_size 2: bucketed
all-reduce launched from post-accumulate hooks must equal single-process training on the concatenated batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.GELU(), torch.nn.LayerNorm(40), torch.nn.Linear(40, 7))


def _worker(rank, world, port, bucket_bytes, q, comm_dtype=None):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from heal_swin_amd.parallel import GradBucketAllReduce

    torch.set_num_threads(1)
    model = _model()
    dp = GradBucketAllReduce(model.parameters(), bucket_bytes=bucket_bytes, comm_dtype=comm_dtype)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 8, 12, generator=g)  # 3 steps, global batch 8
    y = torch.randn(3, 8, 7, generator=g)
    for step in range(3):
        dp.zero_grad()
        xs, ys = x[step].chunk(world)[rank], y[step].chunk(world)[rank]
        torch.nn.functional.mse_loss(model(xs), ys).backward()
        dp.finish()
        opt.step()
    q.put((rank, [p.detach().numpy().copy() for p in model.parameters()], len(dp.buckets)))
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes,comm_dtype", [(64 << 20, None), (1024, None), (1024, torch.bfloat16)])  # one bucket / many small buckets / bf16 wire format
def test_dp_matches_single_process(bucket_bytes, comm_dtype):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_bytes, q, comm_dtype)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the full batch
    model = _model()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 8, 12, generator=g)
    y = torch.randn(3, 8, 7, generator=g)
    for step in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(model(x[step]), y[step]).backward()
        opt.step()
    # bf16 wire format: each exchanged gradient carries 2^-9 relative rounding; three Adam steps at lr 1e-2 move a weight by at
    # most 3e-2, of which the rounding can flip a few per cent
    atol, rtol = (1e-6, 1e-5) if comm_dtype is None else (3e-3, 0.0)
    for (rank, params, nb) in results:
        for a, b in zip(params, model.parameters()):
            assert torch.allclose(torch.from_numpy(a), b.detach(), atol=atol, rtol=rtol), rank
    assert results[0][2] == (1 if bucket_bytes > 1e6 else results[0][2]) and (bucket_bytes > 1e6 or results[0][2] > 1)
    for a, b in zip(results[0][1], results[1][1]):
        assert (a == b).all()  # replicas stay bit-identical


def test_single_process_is_a_noop_wrapper():
    sys.path.insert(0, ROOT)
    from heal_swin_amd.parallel import GradBucketAllReduce

    model = _model()
    dp = GradBucketAllReduce(model.parameters())
    assert dp.world == 1
    dp.zero_grad()
    model(torch.randn(4, 12)).sum().backward()
    dp.finish()
    flat = torch.cat([p.grad.reshape(-1) for p in reversed(list(model.parameters()))])
    assert torch.equal(flat, dp.buckets[0])  # .grad tensors are views into the flat bucket


def test_skipped_finish_is_reported():
    """Two backward() calls without finish() in between (easy to do on one GPU, where nothing is exchanged) used to let the
    direct-deposit gradients pile up silently; the second backward now raises."""
    sys.path.insert(0, ROOT)
    from heal_swin_amd.parallel import GradBucketAllReduce

    model = _model()
    dp = GradBucketAllReduce(model.parameters())
    dp.zero_grad()
    model(torch.randn(4, 12)).sum().backward()
    with pytest.raises(RuntimeError, match="finish"):
        model(torch.randn(4, 12)).sum().backward()
    dp.remove()


def _accum_worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from heal_swin_amd.parallel import GradBucketAllReduce

    torch.set_num_threads(1)
    model = _model()
    dp = GradBucketAllReduce(model.parameters(), bucket_bytes=1024)
    opt = torch.optim.SGD(model.parameters(), lr=1e-1)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 2, 8, 12, generator=g)  # 2 steps x 2 micro-batches, global micro-batch 8
    y = torch.randn(2, 2, 8, 7, generator=g)
    err = None
    try:
        for step in range(2):
            if mode == "set_to_none":
                opt.zero_grad(set_to_none=True)  # the PyTorch / Lightning default: .grad views dropped, detected at the next backward
            else:
                dp.zero_grad()
            for mb in range(2):
                xs, ys = x[step, mb].chunk(world)[rank], y[step, mb].chunk(world)[rank]
                if mode == "no_no_sync":  # accumulation WITHOUT no_sync(): the second backward must raise, not diverge
                    torch.nn.functional.mse_loss(model(xs), ys).backward()
                    dp.finish()
                elif mb == 0:
                    with dp.no_sync():
                        torch.nn.functional.mse_loss(model(xs), ys).backward()
                        dp.finish()
                else:
                    torch.nn.functional.mse_loss(model(xs), ys).backward()
                    dp.finish()
            assert all(p.grad.data_ptr() == dp._views[p].data_ptr() for p in dp.params)  # still bucket views
            opt.step()
    except RuntimeError as e:
        err = str(e)
    q.put((rank, [p.detach().numpy().copy() for p in model.parameters()], err))
    dist.destroy_process_group()


def _run_accum(mode):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


@pytest.mark.parametrize("mode", ["dp_zero_grad", "set_to_none"])
def test_gradient_accumulation_under_no_sync_matches_single_process(mode):
    """Two micro-batches per step (the first under no_sync()): the replicas must end up with the gradients of the sum of
    both micro-batch losses averaged over the ranks -- i.e. equal single-process training on the un-split micro-batches --
    whether the caller zeroes with dp.zero_grad() or with optimizer.zero_grad(set_to_none=True)."""
    results = _run_accum(mode)
    model = _model()
    opt = torch.optim.SGD(model.parameters(), lr=1e-1)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 2, 8, 12, generator=g)
    y = torch.randn(2, 2, 8, 7, generator=g)
    for step in range(2):
        opt.zero_grad()
        for mb in range(2):
            torch.nn.functional.mse_loss(model(x[step, mb]), y[step, mb]).backward()
        opt.step()
    for rank, params, err in results:
        assert err is None, err
        for a, b in zip(params, model.parameters()):
            assert torch.allclose(torch.from_numpy(a), b.detach(), atol=1e-6, rtol=1e-5), rank
    for a, b in zip(results[0][1], results[1][1]):
        assert (a == b).all()


def test_second_backward_on_exchanged_gradients_raises():
    for rank, params, err in _run_accum("no_no_sync"):
        assert err is not None and "no_sync" in err, err


def test_other_models_are_not_touched_by_the_sink():
    """grad_buffer() answers only for the parameters registered with this instance."""
    sys.path.insert(0, ROOT)
    from heal_swin_amd.parallel import GradBucketAllReduce

    model, other = _model(), _model()
    dp = GradBucketAllReduce(model.parameters())
    p_in, p_out = next(model.parameters()), next(other.parameters())
    assert dp.grad_buffer(p_out) is None
    assert dp.grad_buffer(p_in).data_ptr() == p_in.grad.data_ptr()
