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
    model.config.drop_path_rate = 0.1
    with pytest.raises(ValueError, match="drop_path_rate"):
        GraphedTrainStep(model, seg_loss, torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True), x, y)
