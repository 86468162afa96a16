"""optim.FlatAdam (csrc/adam.hip) against torch.optim.Adam / AdamW -- the optimizer the reference's trainer builds
(heal_swin/training/optimizer.py:57-66) -- on the same parameters and gradients, and its hand-over of the bf16 parameter copies
to the model's forward (ops.ParamCastCache)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _toy(seed=0):
    torch.manual_seed(seed)
    shapes = [(64, 32), (64,), (7, 5, 3), (1,), (129, 33), (1000,)]  # odd sizes: bucket tails that are no multiple of 4
    return [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]


@pytest.mark.parametrize("wd,decoupled", [(0.0, False), (0.05, False), (0.05, True)])
def test_flat_adam_follows_torch_adam(wd, decoupled):
    from heal_swin_amd.optim import FlatAdam
    from heal_swin_amd.parallel import GradBucketAllReduce
    a, b = _toy(), _toy()
    ref_cls = torch.optim.AdamW if decoupled else torch.optim.Adam
    ref = ref_cls(a, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    dp = GradBucketAllReduce(b, bucket_bytes=16 << 10, direct_wgrad=False)  # several buckets
    try:
        opt = FlatAdam(b, dp, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd, decoupled_weight_decay=decoupled)
        assert all(torch.equal(x, y) for x, y in zip(a, b)), "moving the parameters into the flat buffers must not change them"
        g = torch.Generator(device=DEV).manual_seed(1)
        for it in range(12):
            for x, y in zip(a, b):
                grad = torch.randn(x.shape, generator=g, device=DEV) * (0.1 + it)
                x.grad = grad.clone()
                y.grad.copy_(grad)      # the sink's bucket view
            ref.step()
            opt.step()
            for x, y in zip(a, b):
                err = float((x - y).abs().max()) / (float(x.abs().max()) + 1e-12)
                assert err < 2e-6, (it, tuple(x.shape), err, x.flatten()[:4].tolist(), y.flatten()[:4].tolist(),
                                    ref.state[x]["exp_avg"].flatten()[:4].tolist(), opt.state[y]["exp_avg"].flatten()[:4].tolist(),
                                    ref.state[x]["exp_avg_sq"].flatten()[:4].tolist(), opt.state[y]["exp_avg_sq"].flatten()[:4].tolist())
        for x, y in zip(a, b):  # moments in torch.optim.Adam's state layout
            # (a few ulp after 12 steps: hipcc contracts g + wd p and the moment updates into FMAs, torch's foreach kernels do not)
            assert float((ref.state[x]["exp_avg"] - opt.state[y]["exp_avg"]).abs().max()) <= 5e-6 * float(ref.state[x]["exp_avg"].abs().max())
            assert float((ref.state[x]["exp_avg_sq"] - opt.state[y]["exp_avg_sq"]).abs().max()) <= 5e-6 * float(ref.state[x]["exp_avg_sq"].abs().max())
        assert int(opt.state[b[0]]["step"]) == 12
    finally:
        dp.remove()


def test_flat_adam_state_dict_round_trip_and_tensor_lr():
    from heal_swin_amd.optim import FlatAdam
    from heal_swin_amd.parallel import GradBucketAllReduce
    b, c = _toy(3), _toy(3)
    dpb, dpc = GradBucketAllReduce(b, direct_wgrad=False), GradBucketAllReduce(c, direct_wgrad=False)
    try:
        lr = torch.tensor(1e-2, device=DEV)
        ob, oc = FlatAdam(b, dpb, lr=lr), FlatAdam(c, dpc, lr=1e-2)
        g = torch.Generator(device=DEV).manual_seed(2)

        def grads():
            for x, y in zip(b, c):
                gr = torch.randn(x.shape, generator=g, device=DEV)
                x.grad.copy_(gr)
                y.grad.copy_(gr)
        for _ in range(3):
            grads()
            ob.step()
            oc.step()
        assert all(torch.equal(x, y) for x, y in zip(b, c)), "tensor lr and float lr must agree"
        sd = copy.deepcopy(ob.state_dict())
        # a fresh optimizer over fresh (equal) parameters resumes from the state dict
        d = [torch.nn.Parameter(x.detach().clone()) for x in b]
        dpd = GradBucketAllReduce(d, direct_wgrad=False)
        try:
            od = FlatAdam(d, dpd, lr=1e-2)
            od.load_state_dict(sd)
            assert int(od._step) == 3
            grads()
            for x, y in zip(c, d):
                y.grad.copy_(x.grad)
            oc.step()
            od.step()
            for x, y in zip(c, d):
                assert float((x - y).abs().max()) <= 1e-6 * float(x.abs().max()), tuple(x.shape)
            assert od.state[d[0]]["exp_avg"].data_ptr() >= od._flat_m[dpd._where[d[0]]].data_ptr(), "state must stay in the flat buffer"
        finally:
            dpd.remove()
    finally:
        dpb.remove()
        dpc.remove()


def test_flat_adam_keeps_the_models_bf16_copies_current():
    """A training step with FlatAdam: the next forward must see the UPDATED weights without a copy pass (the step kernel wrote
    the bf16 shadows), equal to what torch.optim.Adam + the cast cache's own refresh give."""
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch.swin_hp_transformer import SwinHPTransformerConfig, SwinHPTransformerSys
    from heal_swin_amd.optim import FlatAdam
    from heal_swin_amd.parallel import GradBucketAllReduce
    spec = DataSpec(dim_in=12 * 32 * 32, f_in=3, f_out=12, base_pix=12, class_names=[])
    cfg = SwinHPTransformerConfig(patch_size=4, window_size=64, shift_size=32, rel_pos_bias="flat", embed_dim=64, depths=[2, 2],
                                  num_heads=[2, 4], drop_path_rate=0.0)
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randint(0, 256, (2, 3, spec.dim_in), generator=g, device=DEV).float()
    y = torch.randint(0, 12, (2, spec.dim_in), generator=g, device=DEV, dtype=torch.uint8)
    losses = {}
    for kind in ("torch", "flat"):
        torch.manual_seed(0)
        model = SwinHPTransformerSys(cfg, spec).to(DEV).train()
        model.compute_dtype = torch.bfloat16
        dp = GradBucketAllReduce(model.parameters())
        try:
            opt = (FlatAdam(model.parameters(), dp, lr=1e-3, model=model) if kind == "flat"
                   else torch.optim.Adam(model.parameters(), lr=1e-3, fused=True))
            copies = []
            orig = torch._foreach_copy_
            torch._foreach_copy_ = lambda *a, **k: (copies.append(1), orig(*a, **k))[1]
            try:
                ls = []
                for _ in range(4):
                    dp.zero_grad()
                    loss = model.forward_seg_loss(x, y)
                    loss.backward()
                    dp.finish()
                    opt.step()
                    ls.append(float(loss))
            finally:
                torch._foreach_copy_ = orig
            losses[kind] = ls
            if kind == "flat":
                assert len(copies) <= 1, f"the cast cache ran {len(copies)} copy passes; FlatAdam should have made them unnecessary"
                cache = model.__dict__["_cast_cache"]
                assert cache.all_external
                for p, sh in zip(cache.params, cache.shadows):
                    assert torch.equal(sh, p.detach().to(torch.bfloat16))
            else:
                assert len(copies) == 4
        finally:
            dp.remove()
    assert losses["flat"][0] == losses["torch"][0]
    for a, b in zip(losses["torch"], losses["flat"]):
        assert abs(a - b) <= 2e-3 * abs(a), (losses["torch"], losses["flat"])
    assert losses["flat"][-1] < losses["flat"][0]


def test_flat_adam_in_a_captured_training_step():
    """The whole step (zero_grad, fwd, loss, bwd, finish, FlatAdam.step) in one HIP graph: replays advance the device-side step
    counter and follow the eager optimizer."""
    from heal_swin_amd.optim import FlatAdam
    from heal_swin_amd.parallel import GradBucketAllReduce
    torch.manual_seed(0)
    lin_a, lin_b = torch.nn.Linear(32, 16).to(DEV), torch.nn.Linear(32, 16).to(DEV)
    lin_b.load_state_dict(lin_a.state_dict())
    xs = torch.randn(8, 64, 32, device=DEV)
    dpa, dpb = GradBucketAllReduce(lin_a.parameters(), direct_wgrad=False), GradBucketAllReduce(lin_b.parameters(), direct_wgrad=False)
    try:
        oa, ob = FlatAdam(lin_a.parameters(), dpa, lr=1e-2), FlatAdam(lin_b.parameters(), dpb, lr=1e-2)
        static_x = xs[0].clone()

        def step(lin, dp, opt, inp):
            dp.zero_grad()
            loss = lin(inp).square().mean()
            loss.backward()
            dp.finish()
            opt.step()
            return loss.detach()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step(lin_b, dpb, ob, static_x)
        torch.cuda.current_stream().wait_stream(side)
        step(lin_a, dpa, oa, xs[0])
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step(lin_b, dpb, ob, static_x)
        step(lin_a, dpa, oa, xs[0])   # (the capture itself does not execute; this pairs with the first replay below)
        static_x.copy_(xs[0])
        graph.replay()
        for i in range(1, 6):
            step(lin_a, dpa, oa, xs[i])
            static_x.copy_(xs[i])
            graph.replay()
        torch.cuda.synchronize()
        assert int(oa._step) == int(ob._step) == 7
        for p, q in zip(lin_a.parameters(), lin_b.parameters()):
            assert float((p - q).abs().max()) <= 1e-6 * float(p.abs().max())
    finally:
        dpa.remove()
        dpb.remove()
