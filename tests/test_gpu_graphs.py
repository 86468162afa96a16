"""`heal_swin_amd.graphs.GraphedTrainStep`: a whole training step (zero_grad, forward, CE loss, backward, Adam) replayed from
one HIP graph gives exactly the eager step's losses and parameters (the library's kernels are deterministic and launch on the
capture stream), with and without the direct-deposit gradient sink."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(seed=0):
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch.swin_hp_transformer import SwinHPTransformerConfig, SwinHPTransformerSys

    cfg = dict(patch_size=4, window_size=64, shift_size=32, shift_strategy="nest_roll", rel_pos_bias="flat", embed_dim=64,
               depths=[2, 2], num_heads=[2, 4], mlp_ratio=4.0, qkv_bias=True, qk_scale=None, use_cos_attn=False, drop_rate=0.0,
               attn_drop_rate=0.0, drop_path_rate=0.0, use_v2_norm_placement=False, ape=False)
    spec = dict(dim_in=8 * 16 * 16, f_in=3, f_out=12, base_pix=8, class_names=[])
    torch.manual_seed(seed)
    model = SwinHPTransformerSys(SwinHPTransformerConfig(**cfg), DataSpec(**spec)).cuda().train()
    model.compute_dtype = torch.bfloat16
    return model, spec


def _batches(spec, n, batch=2):
    g = torch.Generator(device="cuda").manual_seed(7)
    return [(torch.randint(0, 256, (batch, 3, spec["dim_in"]), generator=g, device="cuda", dtype=torch.uint8),
             torch.randint(0, spec["f_out"], (batch, spec["dim_in"]), generator=g, device="cuda", dtype=torch.uint8)) for _ in range(n)]


@pytest.mark.parametrize("sink", [False, True])
def test_graphed_step_equals_eager_step(sink):
    from heal_swin_amd.graphs import GraphedTrainStep
    from heal_swin_amd.losses import seg_loss
    from heal_swin_amd.parallel import GradBucketAllReduce

    losses, finals = {}, {}
    for mode in ("eager", "graph"):
        model, spec = _build()
        data = _batches(spec, 5)
        dp = GradBucketAllReduce(model.parameters()) if sink else None
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)
        out = []
        if mode == "graph":
            # the constructor's two warm-up steps train on the example batch (the capture itself records, it does not execute)
            step = GraphedTrainStep(model, seg_loss, opt, data[0][0], data[0][1], warmup=2, pre_forward=lambda x: x.float(), grad_sink=dp)
            for x, y in data[1:]:
                out.append(float(step(x, y)))
        else:
            def eager(x, y):
                if dp is not None:
                    dp.zero_grad()
                else:
                    opt.zero_grad(set_to_none=False)
                loss = seg_loss(model(x.float()), y)
                loss.backward()
                if dp is not None:
                    dp.finish()
                opt.step()
                return float(loss)
            for _ in range(2):  # the same two warm-up steps
                eager(*data[0])
            for x, y in data[1:]:
                out.append(eager(x, y))
        losses[mode] = out
        finals[mode] = {k: v.detach().clone() for k, v in model.named_parameters()}
        if dp is not None:
            dp.remove()
    assert losses["eager"] == losses["graph"], (losses["eager"], losses["graph"])
    for k, v in finals["eager"].items():
        assert torch.equal(v, finals["graph"][k]), k


def test_eager_forward_between_replays_sees_the_current_weights():
    """replay -> eval -> replay -> eval: every eager (no-grad) forward between graph replays must use the parameters as the
    last replay left them.  The bf16 weight copies are keyed on the parameters' Python-side version counters, which a replay
    does not bump: GraphedTrainStep invalidates them after each replay."""
    from heal_swin_amd.graphs import GraphedTrainStep
    from heal_swin_amd.losses import seg_loss

    evals = {}
    for mode in ("eager", "graph"):
        model, spec = _build()
        data = _batches(spec, 5)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, fused=True, capturable=True)
        xe = data[0][0].float()
        out = []

        def evaluate():
            model.eval()
            with torch.no_grad():
                y = model(xe).float().clone()
            model.train()
            return y

        if mode == "graph":
            step = GraphedTrainStep(model, seg_loss, opt, data[0][0], data[0][1], warmup=2, pre_forward=lambda x: x.float())
            for x, y in data[1:]:
                step(x, y)
                out.append(evaluate())
        else:
            def eager(x, y):
                opt.zero_grad(set_to_none=False)
                loss = seg_loss(model(x.float()), y)
                loss.backward()
                opt.step()
            for _ in range(2):
                eager(*data[0])
            for x, y in data[1:]:
                eager(x, y)
                out.append(evaluate())
        evals[mode] = out
    for i, (a, b) in enumerate(zip(evals["eager"], evals["graph"])):
        assert torch.equal(a, b), f"eval after replay {i}: max diff {float((a - b).abs().max())}"
    assert not torch.equal(evals["graph"][0], evals["graph"][-1])  # the weights did move between the evaluations


def test_graphed_step_refuses_what_it_cannot_capture():
    from heal_swin_amd.graphs import GraphedTrainStep
    from heal_swin_amd.losses import seg_loss

    model, spec = _build()
    (x, y), = _batches(spec, 1)
    with pytest.raises(ValueError, match="capturable"):
        GraphedTrainStep(model, seg_loss, torch.optim.Adam(model.parameters(), lr=1e-3), x, y)


def _build_stochastic(v2, seed=3):
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch.swin_hp_transformer import SwinHPTransformerConfig, SwinHPTransformerSys

    cfg = dict(patch_size=4, window_size=64, shift_size=32, shift_strategy="nest_roll", rel_pos_bias="flat", embed_dim=96,
               depths=[2, 2], num_heads=[3, 6], mlp_ratio=4.0, qkv_bias=True, qk_scale=None, use_cos_attn=v2, drop_rate=0.1,
               attn_drop_rate=0.1, drop_path_rate=0.1, use_v2_norm_placement=v2, ape=False)
    spec = dict(dim_in=8 * 16 * 16, f_in=3, f_out=12, base_pix=8, class_names=[])
    torch.manual_seed(seed)
    model = SwinHPTransformerSys(SwinHPTransformerConfig(**cfg), DataSpec(**spec)).cuda().train()
    model.compute_dtype = torch.bfloat16
    return model, spec


@pytest.mark.parametrize("v2", [False, True])
def test_graphed_step_with_dropout_draws_new_masks_on_every_replay(v2):
    """Dropout / DropPath under replay: the kernels' host-drawn seeds are frozen into the graph, the library's replay counter
    (hs_set_seed_epoch) is not.  With lr = 0 the parameters never move, so the loss of a replay depends on the masks alone:
    consecutive replays of the same batch must differ (frozen masks would repeat bit-identically), the counter must advance by one per
    step, and an eager evaluation afterwards must still equal the no-dropout forward (eval mode is untouched)."""
    from heal_swin_amd import _lib
    from heal_swin_amd.graphs import GraphedTrainStep
    from heal_swin_amd.losses import seg_loss

    model, spec = _build_stochastic(v2)
    (x, y), = _batches(spec, 1)
    opt = torch.optim.Adam(model.parameters(), lr=0.0, capturable=True)
    assert not _lib.lib.hs_get_seed_epoch()
    step = GraphedTrainStep(model, lambda out, t: seg_loss(out, t), opt, x, y, pre_forward=lambda t: t.float())
    try:
        assert _lib.lib.hs_get_seed_epoch()
        e0 = int(step._epoch.item())
        losses = [float(step(x, y)) for _ in range(6)]
        assert int(step._epoch.item()) == e0 + 6
        assert all(l == l and abs(l) < 1e3 for l in losses)
        assert len(set(losses)) == len(losses), losses  # six different mask draws
        spread = max(losses) - min(losses)
        assert spread < 0.2 * abs(losses[0]), losses    # ... of the same network on the same batch
        with pytest.raises(ValueError, match="another graphed step with dropout"):
            GraphedTrainStep(model, seg_loss, opt, x, y, pre_forward=lambda t: t.float())
    finally:
        step.close()
    assert not _lib.lib.hs_get_seed_epoch()


def test_seed_epoch_reaches_every_mask_generator():
    """hs_set_seed_epoch: one static device pointer per translation unit with stochastic kernels.  For each of them -- elementwise GELU,
    LayerNorm, the GEMM epilogue, the fused Mlp block, the MFMA / fp32-MFMA / generic attention kernels -- the mask of a fixed seed is
    unchanged with the counter at 0, changes with the counter at 1, and is reproducible there."""
    from heal_swin_amd import _lib, ops
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(5)
    C = 96
    x = torch.randn(4, 256, C, generator=g, device=dev).to(torch.bfloat16)
    w1 = torch.randn(4 * C, C, generator=g, device=dev) * 0.1
    w2 = torch.randn(C, 4 * C, generator=g, device=dev) * 0.1
    b1, b2 = torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev)
    lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    qkv = torch.randn(1, 1024, 3 * 64, generator=g, device=dev)
    hs = torch.full((2,), 0.2, device=dev)
    qkv_odd = torch.randn(1, 256, 3 * 48, generator=g, device=dev)  # head_dim 24: the fp32-VALU path
    cases = {
        "gelu": lambda: ops.GeluDropoutFn.apply(x.float(), 0.3, 11),
        "layernorm": lambda: ops.layer_norm(x, lw, lb, residual=None, row_scale=None, drop_p=0.3, seed=12),
        "gemm_nt epilogue": lambda: ops.mlp(x, w1, b1, w2, b2, drop_p=0.3, seed=13),
        "fused mlp": lambda: ops.fused_mlp_block(x, lw, lb, w1, b1, w2, b2, post_norm=True, row_scale=None, drop_p=0.3, seeds=(14, 15)),
        "attention mfma": lambda: ops.window_attn_core(qkv.to(torch.bfloat16), None, hs, None, 0, None, 2, 64, False, attn_drop=0.3, seed=16),
        "attention mfma f32": lambda: ops.window_attn_core(qkv, None, hs, None, 0, None, 2, 64, False, attn_drop=0.3, seed=17),
        "attention generic": lambda: ops.window_attn_core(qkv_odd, None, hs, None, 0, None, 2, 64, False, attn_drop=0.3, seed=18),
    }
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    for name, run in cases.items():
        base = run().float().clone()
        _lib.check(_lib.lib.hs_set_seed_epoch(_lib.ptr(counter)), "hs_set_seed_epoch")
        try:
            counter.zero_()
            assert torch.equal(run().float(), base), f"{name}: counter 0 must not change the mask"
            counter.fill_(1)
            a = run().float().clone()
            b = run().float().clone()
            assert torch.equal(a, b), f"{name}: not reproducible at counter 1"
            assert not torch.equal(a, base), f"{name}: the counter did not reach this kernel"
        finally:
            _lib.lib.hs_set_seed_epoch(None)
        assert torch.equal(run().float(), base), f"{name}: unregistering must restore the plain seed"
