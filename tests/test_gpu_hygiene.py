"""Round-2 robustness items on the GPU: odd window sizes, dropout key mixing, side-stream hand-over, gradient deposit under
optimizer.zero_grad(set_to_none=True), the decoder head's leaf gradient, the bf16 weight-copy cache, bench.py self-launch."""
import json
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest
import torch

from _util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_attn(qkv, nH, Ws, scale):
    """plain windowed softmax attention on [B, N, 3C] (no bias / mask / shift): swin_hp_transformer.py:131-171 without proj"""
    B, N, C3 = qkv.shape
    C = C3 // 3
    hd = C // nH
    q, k, v = qkv.reshape(B, N // Ws, Ws, 3, nH, hd).permute(3, 0, 1, 4, 2, 5)
    a = torch.softmax((q * scale) @ k.transpose(-2, -1), dim=-1)
    return (a @ v).permute(0, 1, 3, 2, 4).reshape(B, N, C)


@pytest.mark.parametrize("Ws,nH,hd", [(8, 2, 16), (32, 3, 32), (128, 2, 32), (16, 1, 128), (64, 2, 128)])
def test_attention_accepts_any_power_of_two_window_and_head_dim_128(Ws, nH, hd):
    """The reference only asserts a power-of-two window (hp_windowing.py:16); windows that are not 4^k (nest_roll with 8, 32,
    128, or a last stage clamped to 8 * 4^k tokens) and head_dim up to 128 run forward AND backward (fp32-VALU kernels)."""
    from heal_swin_amd import ops
    C = nH * hd
    g = torch.Generator().manual_seed(Ws + hd)
    qkv = torch.randn(2, 4 * Ws, 3 * C, generator=g)
    dy = torch.randn(2, 4 * Ws, C, generator=g)
    scale = hd ** -0.5
    ref_in = qkv.clone().requires_grad_(True)
    ref = _oracle_attn(ref_in, nH, Ws, scale)
    ref.backward(dy)
    x = qkv.to(DEV).requires_grad_(True)
    hs = torch.full((nH,), scale, device=DEV)
    y = ops.window_attn_core(x, None, hs, None, 0, None, nH, Ws, False)
    y.backward(dy.to(DEV))
    assert_close(y, ref, 1e-4, "attn out")
    assert_close(x.grad, ref_in.grad, 1e-4, "attn dqkv")


def test_model_with_window_32_runs():
    """nest_roll with a window that is not 4^k and no relative-position bias: legal in the reference."""
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch import swin_hp_transformer as M
    from oracle import model as OM
    import types
    cfg = dict(patch_size=4, window_size=32, shift_size=16, shift_strategy="nest_roll", rel_pos_bias=None, embed_dim=32,
               depths=[2, 2], num_heads=[2, 4], mlp_ratio=4.0, qkv_bias=True, qk_scale=None, use_cos_attn=False, drop_rate=0.0,
               attn_drop_rate=0.0, drop_path_rate=0.0, use_v2_norm_placement=False, ape=False)
    spec = dict(dim_in=8 * 16 * 16, f_in=3, f_out=5, base_pix=8, class_names=[])
    torch.manual_seed(0)
    model = M.SwinHPTransformerSys(M.SwinHPTransformerConfig(**cfg), DataSpec(**spec))
    sd = {k: v.clone() for k, v in model.state_dict().items() if not k.endswith("attn_mask")}
    x = torch.randint(0, 256, (2, 3, spec["dim_in"])).float()
    y_ref = OM.forward(sd, types.SimpleNamespace(**cfg), types.SimpleNamespace(**spec), x)
    y = model.to(DEV)(x.to(DEV))
    assert_close(y, y_ref, 1e-3, "logits window 32")


def test_dropout_masks_of_different_seeds_are_uncorrelated():
    """seeds must select unrelated mask sequences, not XOR-translated windows of one sequence."""
    from heal_swin_amd import ops
    n = 1 << 20
    x = torch.ones(n, device=DEV) * 3.0  # gelu(3) > 0: survivors are visible
    masks = []
    for seed in (1, 2, 3, 1 << 32, (1 << 32) + 1, 0x12345678_9ABCDEF0):
        y = ops.GeluDropoutFn.apply(x, 0.5, seed)
        masks.append((y > 0).float().cpu().numpy())
    for m in masks:
        assert abs(m.mean() - 0.5) < 5e-3
    for i in range(len(masks)):
        for j in range(i + 1, len(masks)):
            a, b = masks[i] - masks[i].mean(), masks[j] - masks[j].mean()
            corr = float((a * b).mean() / (a.std() * b.std()))
            assert abs(corr) < 6e-3, (i, j, corr)  # ~6 sigma of 1/sqrt(n)
            # and no small shift aligns them (the old construction made mask(seed b) = mask(seed a) at index ^ const)
            for sh in (1, 2, 64, 4096):
                c2 = float((a[sh:] * b[:-sh]).mean() / (a.std() * b.std()))
                assert abs(c2) < 6e-3, (i, j, sh, c2)
    # XOR-translate check: for seeds differing in the low word only, mask_b[i] == mask_a[i ^ k] must NOT hold for small k
    a, b = masks[0], masks[1]
    idx = np.arange(n)
    for k in (2, 4, 6):  # element pairs share a hash: translate by whole pairs
        assert (a[idx ^ k] == b).mean() < 0.52


def test_dropout_mask_elements_of_a_chunk_are_independent():
    """The elementwise generator (csrc/hs_device.h: ElemRng) derives the 8 masks of a chunk from ONE 32-bit chunk key by four fixed
    multipliers: every element is dropped with probability p, every pair of elements of a chunk -- and neighbours across chunks --
    independently (correlation within 6 sigma of 0), and triples jointly at p^3."""
    from heal_swin_amd import ops
    n = 1 << 23
    x = torch.ones(n, device=DEV) * 3.0
    for p, seed in ((0.1, 5), (0.5, 0xDEADBEEF_00000007)):
        keep = (ops.GeluDropoutFn.apply(x, p, seed) > 0).view(-1, 8).double()
        rows = keep.shape[0]
        mean = keep.mean(0)
        assert float((mean - (1 - p)).abs().max()) < 6 * (p * (1 - p) / rows) ** 0.5 + 1e-4, mean
        z = (keep - mean) / keep.std(0)
        corr = (z.t() @ z / rows).cpu().numpy()
        off = corr - np.eye(8)
        assert np.abs(off).max() < 6 / rows ** 0.5, (p, np.abs(off).max())
        zc = (z[:-1].t() @ z[1:] / (rows - 1)).cpu().numpy()  # element a of chunk c against element b of chunk c + 1
        assert np.abs(zc).max() < 6 / rows ** 0.5, (p, np.abs(zc).max())
        drop = 1 - keep
        for tri in ((0, 1, 2), (0, 2, 4), (1, 3, 7), (2, 3, 6), (4, 5, 6)):
            joint = float((drop[:, tri[0]] * drop[:, tri[1]] * drop[:, tri[2]]).mean())
            sigma = (p ** 3 * (1 - p ** 3) / rows) ** 0.5
            assert abs(joint - p ** 3) < 6 * sigma + 1e-6, (p, tri, joint, p ** 3)


def test_attention_dropout_keys_of_a_row_are_independent():
    """The attention generator (csrc/window_attn.h: DropRng) derives the 64 key masks of a (head, query) row from two 32-bit row keys
    by one compile-time multiplier per key pair.  Masks are read off the kernel itself (q = k = 0: uniform probabilities; V = one-hot
    over 32 of the keys): every key dropped with probability p, all key pairs of a row uncorrelated."""
    from heal_swin_amd import ops
    B, nH, Ws, hd = 1, 2, 64, 32
    N, C = 64 * 1024, 64
    p = 0.3
    masks = []
    for half in (0, 1):
        qkv = torch.zeros(B, N, 3 * C, device=DEV, dtype=torch.bfloat16)
        tok = torch.arange(N, device=DEV)
        key = tok % Ws
        sel = (key // hd) == half
        for h in range(nH):
            qkv[0, tok[sel], 2 * C + h * hd + (key[sel] % hd)] = 1.0
        o = ops.window_attn_core(qkv, None, torch.ones(nH, device=DEV), None, 0, None, nH, Ws, False, attn_drop=p, seed=0xABCDEF12_3456789)
        masks.append((o.float().view(N, nH, hd) > 0).permute(1, 0, 2).reshape(nH * N, hd))  # [row, key within the half]
    keep = torch.cat(masks, 1).double()  # [rows, 64 keys]
    rows = keep.shape[0]
    mean = keep.mean(0)
    assert float((mean - (1 - p)).abs().max()) < 6 * (p * (1 - p) / rows) ** 0.5 + 1e-4, mean
    z = (keep - mean) / keep.std(0)
    corr = (z.t() @ z / rows).cpu().numpy() - np.eye(Ws)
    assert np.abs(corr).max() < 6 / rows ** 0.5, np.abs(corr).max()
    zr = (z[:-1].t() @ z[1:] / (rows - 1)).cpu().numpy()  # consecutive rows
    assert np.abs(zr).max() < 6 / rows ** 0.5, np.abs(zr).max()


def test_to_device_side_stream_handover():
    from heal_swin_amd import data
    imgs = torch.randint(0, 256, (4, 3, 1 << 20), dtype=torch.uint8).pin_memory()
    masks = torch.randint(0, 12, (4, 1 << 20), dtype=torch.uint8).pin_memory()
    copy_stream = torch.cuda.Stream()
    for _ in range(3):
        busy = torch.randn(4096, 4096, device=DEV) @ torch.randn(4096, 4096, device=DEV)  # keep the main stream busy
        batch = data.to_device((imgs, masks), DEV, stream=copy_stream)
        assert batch.ready is not None
        a, b = batch.wait()
        s = a.long().sum() + b.long().sum()  # consumer on the current stream
        assert int(s) == int(imgs.long().sum() + masks.long().sum())
        del busy
    plain = data.to_device((imgs, masks), DEV)
    assert plain.ready is None and torch.equal(plain.wait()[0].cpu(), imgs)


def _small_model():
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch import swin_hp_transformer as M
    cfg = dict(patch_size=4, window_size=64, shift_size=32, shift_strategy="nest_roll", rel_pos_bias="flat", embed_dim=64,
               depths=[2, 2], num_heads=[2, 4], drop_path_rate=0.0)
    spec = DataSpec(dim_in=12 * 32 * 32, f_in=3, f_out=12, base_pix=12, class_names=[])
    torch.manual_seed(11)
    m = M.SwinHPTransformerSys(M.SwinHPTransformerConfig(**cfg), spec).to(DEV)
    m.compute_dtype = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 256, (2, 3, spec.dim_in), generator=g).float().to(DEV)
    y = torch.randint(0, 12, (2, spec.dim_in), generator=g).to(DEV)
    return m, x, y


def test_direct_deposit_survives_zero_grad_set_to_none_and_head_is_a_leaf():
    """optimizer.zero_grad(set_to_none=True) (PyTorch / Lightning default) must not drop the model to the slow gradient path:
    after every step each .grad is again a view into the flat buckets, the results equal the dp.zero_grad() run bit for
    bit, the decoder head (a Conv1d weight used as a matrix) receives its gradient as a leaf, and no warning is raised."""
    from heal_swin_amd.losses import seg_loss
    from heal_swin_amd.parallel import GradBucketAllReduce
    outs = {}
    for mode in ("dp", "none"):
        model, x, y = _small_model()
        dp = GradBucketAllReduce(model.parameters(), bucket_bytes=256 << 10)
        opt = torch.optim.SGD(model.parameters(), lr=1e-2)
        with warnings.catch_warnings():
            warnings.simplefilter("error")  # the round-1 head path read .grad of a non-leaf: UserWarning
            for _ in range(3):
                if mode == "dp":
                    dp.zero_grad()
                else:
                    opt.zero_grad(set_to_none=True)
                    assert all(p.grad is None for p in model.parameters())
                seg_loss(model(x), y).backward()
                dp.finish()
                assert all(p.grad is not None and p.grad.data_ptr() == dp._views[p].data_ptr() for p in dp.params)
                head = model.decoder.output.weight
                assert head.is_leaf and float(head.grad.abs().max()) > 0
                opt.step()
        torch.cuda.synchronize()
        outs[mode] = [p.detach().clone() for p in model.parameters()]
        dp.remove()
    for a, b in zip(outs["dp"], outs["none"]):
        assert torch.equal(a, b)


def test_param_cast_cache_sees_data_swaps_and_invalidate():
    model, x, y = _small_model()
    model.eval()
    with torch.no_grad():
        y0 = model(x).float()
        w = model.layers[0].blocks[0].mlp.fc1.weight
        w.data = w.data * 0.5          # new storage: caught through data_ptr
        y1 = model(x).float()
        assert not torch.equal(y0, y1)
        w.data.mul_(2.0)               # in place through .data: invisible to the version counter ...
        model.invalidate_param_casts()  # ... so the caller says so
        y2 = model(x).float()
    assert torch.equal(y0, y2)


def test_bench_self_launches_multi_rank():
    """`python bench.py --gpus 2` with NO launcher around it: re-executes itself under torch.distributed.run (two ranks share
    this box's single GPU over gloo: HS_BENCH_SHARED_GPU=1) and reports the exchange in the JSON line."""
    env = dict(os.environ, HS_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "tiny"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["value"] > 0
    assert out["rccl"]["rccl_ranks"] == 2 and len(out["rccl"]["allreduce_ms_per_step_standalone_per_rank"]) == 2
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["frac"] > 0 and out["roofline"]["mfma"]["frac"] > 0
    assert len(out["rccl"]["step_ms_per_rank"]) == 2
