#!/usr/bin/env python3
"""Throughput benchmark of the HEAL-SWIN hot path (fwd + loss + bwd + gradient exchange + Adam step).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0: BASELINE.json's metric (images/s, whole job), plus
  roofline      fused shift+window-attention kernels (fwd + bwd launches of the timed steps), algorithmic bytes
                / measured launch time against the HBM peak (HIP events on the launch stream)
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's forward) timed on this host's cores on a bounded
                sample of the same model (rank 0, N = 1 only)
Data is synthetic (uint8-range images, random labels), weights are random-init.
"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense, 2495 TF measured)

WORKLOADS = {
    # BASELINE.json configs[2]: HEAL-SWIN-B, nside 256, 12 base pixels (full sphere), window 64 (nest_roll: the only
    # shift valid for 12 base pixels), 12 classes
    "B256": dict(name="HEAL-SWIN-B nside=256 base_pix=12 window=64 nest_roll(shift 32) seg 12 classes",
                 nside=256, base_pix=12, f_out=12, fwd_gflop_per_image=3799.23,
                 cfg=dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=64, shift_size=32,
                          shift_strategy="nest_roll", rel_pos_bias="flat")),
    # the paper's run config (run_configs/segmentation/swin_hp_synwoodscape_large_plus_AD_train_run_config.py:35-96)
    "T256": dict(name="HEAL-SWIN-T (paper) nside=256 base_pix=8 window=64 ring_shift(4) cos-attn v2-norm seg 12 classes",
                 nside=256, base_pix=8, f_out=12, fwd_gflop_per_image=722.26,
                 cfg=dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=64, shift_size=4,
                          shift_strategy="ring_shift", rel_pos_bias="flat", use_cos_attn=True, use_v2_norm_placement=True)),
    # BASELINE.json configs[1]
    "T128": dict(name="HEAL-SWIN-T nside=128 base_pix=8 window=64 nest_roll(shift 32) seg 12 classes",
                 nside=128, base_pix=8, f_out=12, fwd_gflop_per_image=180.56,
                 cfg=dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=64, shift_size=32,
                          shift_strategy="nest_roll", rel_pos_bias="flat")),
    # BASELINE.json configs[4]: depth-estimation head (f_out = 1, masked L1 loss), fp32 -- the `depth_fp32` companion of the line
    "D256": dict(name="HEAL-SWIN-T depth head nside=256 base_pix=8 window=64 nest_roll(shift 32) f_out=1, L1 over finite targets",
                 nside=256, base_pix=8, f_out=1, fwd_gflop_per_image=721.15, task="depth",
                 cfg=dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=64, shift_size=32,
                          shift_strategy="nest_roll", rel_pos_bias="flat")),
    # BASELINE.json configs[0] shape (plumbing)
    "tiny": dict(name="HEAL-SWIN-tiny nside=32 base_pix=4 window=16", nside=32, base_pix=4, f_out=12, fwd_gflop_per_image=0.67,
                 cfg=dict(embed_dim=48, depths=[2, 2, 2], num_heads=[3, 6, 12], window_size=16, shift_size=8,
                          shift_strategy="nest_roll", rel_pos_bias="flat")),
}


COMPANION_WORKLOADS = ("T128", "T256")  # the reference's own model (BASELINE configs[1] and the paper config): companion lines


def full_cfg(cfg):
    base = dict(patch_size=4, window_size=64, shift_size=32, shift_strategy="nest_roll", rel_pos_bias="flat", embed_dim=96,
                depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], mlp_ratio=4.0, qkv_bias=True, qk_scale=None, use_cos_attn=False,
                drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, use_v2_norm_placement=False, ape=False)
    base.update(cfg)
    return base


def build_model(wl, nside=None):
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch.swin_hp_transformer import SwinHPTransformerConfig, SwinHPTransformerSys

    nside = nside or wl["nside"]
    cfg = full_cfg(wl["cfg"])
    spec = dict(dim_in=wl["base_pix"] * nside * nside, f_in=3, f_out=wl["f_out"], base_pix=wl["base_pix"], class_names=[])
    torch.manual_seed(0)
    model = SwinHPTransformerSys(SwinHPTransformerConfig(**cfg), DataSpec(**spec))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("relative_position_bias_table"):
                p.normal_(0, 0.02)
    return model, cfg, spec


def _traffic_table(records):
    table = {}
    for r in records:
        tag = r["kernel"].replace("hs_", "")
        table.setdefault((tag, int(r["algorithmic_bytes"])), []).append(r["hbm_bytes_corrected"])
    return table


def measure_pmc_traffic(workload, batch, timeout_s=240):
    """HBM bytes of the attention launches MEASURED on this box: two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE -- separate
    runs, only --kernel-trace beside --pmc, as MI355X_MICROARCH.md prescribes) over tools/bench_attn.py at this workload's stage
    shapes, reduced by tools/attn_pmc_traffic.py (KiB units, FETCH_SIZE x 2 for wide coalesced reads on gfx950).  Returns the
    records, or None when rocprofv3 is not on PATH / a pass fails (the caller then falls back to the committed profile)."""
    import glob
    import shutil
    import subprocess
    import tempfile

    if not shutil.which("rocprofv3"):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import attn_pmc_traffic as APT
    except Exception:  # noqa: BLE001
        return None
    work = tempfile.mkdtemp(prefix="hs_bench_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(work, c), "-o", "t", "--",
                   sys.executable, os.path.join(ROOT, "tools", "bench_attn.py"), "--iters", "2", "--workload", workload, "--batch", str(batch)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
        f = glob.glob(os.path.join(work, "FETCH_SIZE", "**", "*counter_collection.csv"), recursive=True)
        w = glob.glob(os.path.join(work, "WRITE_SIZE", "**", "*counter_collection.csv"), recursive=True)
        k = glob.glob(os.path.join(work, "FETCH_SIZE", "**", "*kernel_trace.csv"), recursive=True)
        if not (f and w and k):
            return None
        return APT.records(f[0], w[0], k[0], WORKLOADS[workload], batch)
    except Exception:  # noqa: BLE001
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def pmc_traffic_per_launch(attn_agg, records=None):
    """HBM bytes per launch, averaged over this run's launch mix by matching each launch's algorithmic byte count against
    `records` (measured in this run, measure_pmc_traffic) or, without them, against the committed passes of the same kernels
    and shapes (profiles/r04b_attn_pmc_hbm_traffic.json); (None, source) if a launch shape is not covered."""
    source = "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/bench_attn.py at the same stage shapes (after the timed region)"
    if records is None:
        path = os.path.join(ROOT, "profiles", "r04b_attn_pmc_hbm_traffic.json")
        if not os.path.exists(path):
            return None, None
        records = json.load(open(path))["records"]
        source = "profiles/r04b_attn_pmc_hbm_traffic.json (committed rocprofv3 --pmc passes of the same kernels and shapes; looked up: rocprofv3 unavailable or failed in this run)"
    table = _traffic_table(records)
    tot, n = 0.0, 0
    for tag, a in attn_agg.items():
        for nbytes, cnt in a[4].items():
            vals = table.get((tag, int(nbytes)))
            if not vals:
                return None, None
            tot += cnt * sum(vals) / len(vals)
            n += cnt
    return (tot / n if n else None), source


def usable_cores():
    """CPU threads this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU boxes expose
    256 hardware threads but a 16-CPU quota; running 256 threads against it throttles everything)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_timing(wl, nside, batch, warmup, iters):
    """Seconds per fwd + CE + bwd of the CPU oracle on `batch` images at `nside` (list of timed iterations after warm-up)."""
    from oracle import model as OM  # the checker, used here as the reported CPU baseline ("port")

    model, cfg, spec = build_model(wl, nside)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()
          if not k.endswith("attn_mask")}
    del model
    g = torch.Generator().manual_seed(0)
    x = torch.randint(0, 256, (batch, 3, spec["dim_in"]), generator=g).float()
    y = torch.randint(0, spec["f_out"], (batch, spec["dim_in"]), generator=g)
    cfg_ns, spec_ns = types.SimpleNamespace(**cfg), types.SimpleNamespace(**spec)
    times = []
    for it in range(warmup + iters):
        t0 = time.time()
        loss = OM.seg_loss(OM.forward(sd, cfg_ns, spec_ns, x), y)
        loss.backward()
        if it >= warmup:
            times.append(time.time() - t0)
        for v in sd.values():
            v.grad = None
    return times


def host_memory_gb():
    """Memory this process may use: MemAvailable capped by the cgroup limit (GiB)."""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = int(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            cur = int(open("/sys/fs/cgroup/memory.current").read())
            lim_free = (int(lim) - cur) / 2 ** 30
            avail = lim_free if avail is None else min(avail, lim_free)
    except (OSError, ValueError):
        pass
    return avail if avail is not None else 16.0


def _peak_rss_gb():
    import resource
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20


def cpu_baseline(wl, budget_s=30.0):
    """Times the CPU oracle (forward + CE loss + backward over all parameters; a port of the reference's forward, fp32) on
    this host's usable cores (SURVEY 8d protocol).  Headline workload: ONE image, at the workload's own nside when one
    iteration fits the time budget (about 10-30 s of CPU work) AND the oracle's autograd graph fits the host memory (it holds
    every [windows, heads, 64, 64] score tensor: both are calibrated at a small nside and extrapolated x4 per doubling),
    otherwise at the largest nside that does, rescaled by the pixel ratio (cost is linear in the pixel count for windowed
    attention) -- `sample` says which.  BASELINE configs[0] (tiny) and configs[1] (T, nside 128) are timed at full size
    (2 warm-up + 3 timed iterations) and the paper config (T, nside 256, 8 base pixels, ring_shift + cosine + v2) with ONE timed
    iteration, as 8d prescribes."""
    cores = usable_cores()
    torch.set_num_threads(cores)
    mem_gb = host_memory_gb()

    def fit(w, budget):
        """Largest nside <= the workload's whose single iteration fits `budget` seconds and 60 % of the host memory:
        (nside, estimated seconds, estimated GiB)."""
        L = len(w["cfg"]["depths"])
        nside = min(w["nside"], 16 * 2 ** (L - 1))  # >= one 64-token window per base-pixel quartet at the last stage
        rss0 = _peak_rss_gb()
        t = min(_oracle_timing(w, nside, 1, 1, 1))  # calibration (after one warm-up: thread pool, allocator)
        gb = max(_peak_rss_gb() - rss0, 0.05)
        while nside * 2 <= w["nside"] and t * 4 * 1.2 < budget and gb * 4 * 1.3 < 0.6 * mem_gb:
            nside, t, gb = nside * 2, t * 4, gb * 4
        return nside, t, gb

    # headline: full nside with two timed iterations if it fits, else 1 warm-up + 3 timed at the reduced nside
    nside, t_est, gb_est = fit(wl, budget_s)
    gb_full = gb_est * (wl["nside"] / nside) ** 2
    if nside == wl["nside"]:
        # (a full-size iteration is ~35 s on 16 threads: TWO timed iterations -- the first also pays thread-pool / allocator warm-up,
        # the sample string carries mean and min)
        times = _oracle_timing(wl, nside, 1, 0, 2) if t_est * 4 > budget_s else _oracle_timing(wl, nside, 1, 1, 3)
    else:
        n2, t2 = nside, t_est
        while n2 > 16 and t2 * 4 * 1.2 > budget_s:  # 1 warm-up + 3 timed iterations inside the budget
            n2, t2 = n2 // 2, t2 / 4
        nside = n2
        times = _oracle_timing(wl, nside, 1, 1, 3)
    t = sum(times) / len(times)
    scale = (wl["nside"] / nside) ** 2
    how = (f"at the workload's own nside={nside} (no rescaling)" if scale == 1 else
           f"at nside={nside}, rescaled x{1 / scale:.4g} to nside={wl['nside']} by pixel count (a full-size iteration: "
           f"estimated {t * scale:.0f} s and {gb_full:.0f} GiB of autograd state)")
    out = {"value": 1.0 / (t * scale), "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": cpu_model_name(),
           "host_memory_GiB_available": round(mem_gb, 1),
           "sample": f"oracle (CPU restatement of the reference forward) fwd+CE+bwd, fp32, 1 image {how}; "
                     f"{len(times)} timed iteration(s) (mean {t:.2f}s, min {min(times):.2f}s on {cores} threads)"}
    other = {}
    for key, batch in (("tiny", 1), ("T128", 1)):  # BASELINE configs[0] and configs[1] at full size
        w = WORKLOADS[key]
        ts = _oracle_timing(w, w["nside"], batch, 2, 3)
        other[key] = {"workload": w["name"], "images_per_s": batch * len(ts) / sum(ts), "s_per_iter": [round(v, 3) for v in ts],
                      "batch": batch, "iters": "2 warm-up + 3 timed"}
    # the paper config: ONE timed iteration at its own size, memory permitting
    w = WORKLOADS["T256"]
    n_p, t_p, gb_p = fit(w, 60.0)
    ts = _oracle_timing(w, n_p, 1, 0, 1)
    sc = (w["nside"] / n_p) ** 2
    other["T256_paper"] = {"workload": w["name"], "images_per_s": 1.0 / (ts[0] * sc), "s_per_iter": [round(ts[0], 3)], "batch": 1,
                           "nside_timed": n_p, "rescaled": sc != 1,
                           "iters": "1 timed iteration, no warm-up (SURVEY 8d: reference anchor 67 s on 8 cores of the survey container)"}
    out["other_configs"] = other
    return out


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="B256", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (weak scaling)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-shape table of the event-timed launches to stderr")
    ap.add_argument("--no-fp32-companion", action="store_true", help="skip the fp32 run of the same workload (N = 1 only)")
    ap.add_argument("--no-graph-companion", action="store_true", help="skip the HIP-graph replay of the same workload (N = 1 only)")
    ap.add_argument("--unfused-loss", action="store_true",
                    help="call the model and losses.seg_loss separately (logits materialised) instead of model.forward_seg_loss")
    ap.add_argument("--no-companions", action="store_true", help="skip the HEAL-SWIN-T companion workloads (T128, T256; N = 1 only)")
    ap.add_argument("--no-pmc-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 counter passes after the timed region (N = 1 only); look it up in profiles/ instead")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (fwd + loss + bwd + Adam) in one HIP graph and replay it (single GPU, no "
                         "dropout); removes host launch latency, which dominates the small workloads")
    ap.add_argument("--paper-drop-rates", action="store_true",
                    help="train with the paper's drop_rate = attn_drop_rate = drop_path_rate = 0.1 instead of 0")
    ap.add_argument("--no-tuned-gemm", action="store_true", help="ignore the shipped TunableOp results for the library GEMMs")
    ap.add_argument("--tune-gemm", metavar="CSV", default=None,
                    help="(maintenance) run PyTorch TunableOp tuning over this workload's library GEMMs during the warm-up and "
                         "write the results file CSV (copy it to heal_swin_amd/tuning/); the timed numbers of such a run are not a benchmark")
    ap.add_argument("--async-wgrad", action="store_true", help="run the Linear weight-gradient kernels on a side stream")
    ap.add_argument("--torch-adam", action="store_true",
                    help="step torch.optim.Adam(fused=True) instead of heal_swin_amd.optim.FlatAdam (same arithmetic; A/B runs)")
    ap.add_argument("--use-checkpoint", action="store_true",
                    help="SwinHPTransformerConfig.use_checkpoint = True (swin_hp_transformer.py:541-542: every block's forward recomputed in the "
                         "backward): the activation-memory policy of the reference, with its price")
    ap.add_argument("--strong-scaling", action="store_true",
                    help="--batch is the GLOBAL batch, split over the ranks (BASELINE's 'batch=8 at 1/2/4/8' read as a fixed total; SURVEY 8d "
                         "asks for both readings).  Default: --batch per GPU (weak scaling, the reference's per-process batch, train.py:34-41)")
    ap.add_argument("--drop-in", action="store_true",
                    help="INTEGRATION.md level 1 exactly: only the module is swapped -- autograd's own .grad accumulation (no gradient "
                         "sink, torch's DistributedDataParallel when N > 1), nn.CrossEntropyLoss on the materialised logits, "
                         "torch.optim.Adam as training/optimizer.py:57-66 builds it, eager")
    ap.add_argument("--reserved-cus", default="auto",
                    help="compute units the chip-filling launches leave free for RCCL (multiple of 8; auto: 16 when N > 1, else 0)")
    ap.add_argument("--comm-dtype", default="fp32", choices=["fp32", "bf16"], help="wire format of the gradient buckets")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)  # does not return
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.strong_scaling:
        if args.batch % world:
            raise SystemExit(f"--strong-scaling: the global batch {args.batch} does not split over {world} ranks")
        args.batch //= world  # from here on: images per GPU
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    # test hook (tests/test_gpu_parallel.py): several ranks on ONE GPU over gloo exercise this script's multi-rank path on a
    # 1-GPU box; the driver's runs use one GPU per rank over RCCL
    shared_gpu = os.environ.get("HS_BENCH_SHARED_GPU") == "1"
    dev_index = 0 if shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the only mode the host driver supports
        if rank == 0 and not shared_gpu:
            # one line with the RCCL version -- on STDERR (RCCL logs to stdout by default; stdout carries the JSON line alone)
            if "NCCL_DEBUG" not in os.environ:
                os.environ["NCCL_DEBUG"] = "VERSION"
                os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        # RCCL's ring kernels are long-lived workgroups that share the chip with the backward.  A kernel that fills all 256 CUs
        # loses 45-65 % when even 8 of them hold a foreign 128-VGPR workgroup (profiles/archive_r01_r04/r03_cu_contention.json), so the exchange
        # gets a bounded number of channels and the library leaves that many CUs free (GradBucketAllReduce(reserved_cus=...)).
        # 596 MB of gradients per 160 ms step need < 10 GB/s: 16 channels are ample.  Both can be overridden from the environment.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
        t_init = time.perf_counter()
        if shared_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
        # diagnosability of the first real multi-GPU run: what every rank sees, on stderr (the JSON line stays alone on stdout)
        env = {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_VISIBLE", "ROCR_VISIBLE", "MASTER_"))}
        # a mis-bound exchange must fail loudly BEFORE anything is timed: five all-reduces of ones have to give the world size
        for _ in range(5):
            probe = torch.ones(1 << 16, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize(dev)
            if not bool((probe == float(world)).all()):
                raise SystemExit(f"[bench rank {rank}] warm exchange check failed: all_reduce(ones) gave {float(probe[0])}, expected {world}")
        print(f"[bench rank {rank}/{world}] device {dev_index}: {torch.cuda.get_device_name(dev_index)}, backend "
              f"{dist.get_backend()}, init {time.perf_counter() - t_init:.2f}s, env {env}", file=sys.stderr, flush=True)

    # Library GEMMs (forward / input-gradient of the Linear layers): load the per-shape hipBLASLt/rocBLAS solution choices
    # tuned once on an MI355X with PyTorch TunableOp (tuning itself stays OFF here; a validator mismatch -- other ROCm,
    # other GPU -- makes PyTorch ignore the file and fall back to the library heuristic).
    tuned = os.path.join(ROOT, "heal_swin_amd", "tuning", f"tunableop_gfx950_{args.workload}_bs{args.batch}_{args.dtype}.csv")
    gemm_selection = "TunableOp tuning run" if args.tune_gemm else "default heuristic"
    if args.tune_gemm:
        os.makedirs(os.path.dirname(os.path.abspath(args.tune_gemm)), exist_ok=True)
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_filename(args.tune_gemm, insert_device_ordinal=False)
        # longer measurements than the defaults (30 ms / 100 iterations) and operands rotated through a buffer larger than
        # L2 + MALL, so that candidates are ranked on HBM-resident behaviour as in the real step
        torch.cuda.tunable.set_max_tuning_duration(int(os.environ.get("HS_TUNE_MS", "100")))
        torch.cuda.tunable.set_max_tuning_iterations(int(os.environ.get("HS_TUNE_ITERS", "200")))
        torch.cuda.tunable.set_rotating_buffer_size(int(os.environ.get("HS_TUNE_ROTATE_MB", "512")))
    elif os.path.exists(tuned) and not args.no_tuned_gemm:
        # one results file per process: the headline workload's picks plus those of the companions this run will execute (the
        # fp32 re-run of the same workload, the fp32 depth-head config), merged into a scratch file
        extra = [os.path.join(os.path.dirname(tuned), f"tunableop_gfx950_{args.workload}_bs{args.batch}_fp32.csv"),
                 os.path.join(os.path.dirname(tuned), "tunableop_gfx950_D256_bs2_fp32.csv")]
        extra += [os.path.join(os.path.dirname(tuned), f"tunableop_gfx950_{w}_bs{args.batch}_bf16.csv") for w in COMPANION_WORKLOADS]
        merged = tuned
        have = [f for f in extra if os.path.exists(f) and f != tuned]
        if have and args.dtype == "bf16":
            import tempfile
            lines = open(tuned).read().splitlines()
            for f in have:
                lines += [ln for ln in open(f).read().splitlines() if ln and not ln.startswith("Validator")]
            fd, merged = tempfile.mkstemp(prefix=f"tunableop_merged_r{rank}_", suffix=".csv")
            with os.fdopen(fd, "w") as fh:
                fh.write("\n".join(lines) + "\n")
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(False)
        torch.cuda.tunable.set_filename(merged, insert_device_ordinal=False)
        ok = torch.cuda.tunable.read_file(merged)
        # PyTorch ignores a results file whose Validator lines (ROCm / hipBLASLt / GPU identification) do not match this process:
        # say what was actually loaded, not that a file exists
        want = sum(1 for ln in open(merged).read().splitlines() if ln and not ln.startswith("Validator"))
        try:
            loaded = len(torch.cuda.tunable.get_results())
        except Exception:  # noqa: BLE001
            loaded = 0
        gemm_selection = (f"TunableOp results file: {loaded} of {want} entries loaded" if (ok and loaded) else
                          "default heuristic (TunableOp results file present but rejected: validator mismatch)")

    wl = WORKLOADS[args.workload]
    if args.paper_drop_rates:
        wl = dict(wl, cfg=dict(wl["cfg"], drop_rate=0.1, attn_drop_rate=0.1, drop_path_rate=0.1),
                  name=wl["name"] + " drop 0.1/0.1/0.1")
    if args.use_checkpoint:
        wl = dict(wl, cfg=dict(wl["cfg"], use_checkpoint=True), name=wl["name"] + " use_checkpoint")
    ctx = types.SimpleNamespace(args=args, wl=wl, dev=dev, world=world, rank=rank, shared_gpu=shared_gpu)
    res = run_workload(ctx, args.dtype, args.steps, args.warmup, timing=not args.no_kernel_timing)
    elapsed = res.elapsed

    if rank == 0:
        images = args.batch * world * args.steps
        out = {
            "metric": "images/sec fwd+bwd, HEAL-SWIN nside=256 seg, batch=8 at 1/2/4/8 MI355X",
            "value": images / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if args.strong_scaling else "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": wl["name"], "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}",
                       "step": ("drop-in (INTEGRATION.md level 1): module swapped only -- fwd, nn.CrossEntropyLoss on the logits, bwd with autograd's "
                                "own .grad accumulation" + (" inside torch DistributedDataParallel" if world > 1 else "") + ", torch.optim.Adam") if args.drop_in else
                               "fwd + CE loss + bwd + grad all-reduce + Adam" + ("" if args.unfused_loss else " (loss fused into the decoder tail: model.forward_seg_loss)"),
                       "optimizer": "torch.optim.Adam (training/optimizer.py:57-66)" if args.drop_in else
                                    "torch.optim.Adam(fused=True)" if args.torch_adam else "heal_swin_amd.optim.FlatAdam (torch.optim.Adam arithmetic on flat buffers)",
                       "launch": "hip graph replay" if args.graph else "eager",
                       "params_M": res.params_m, "final_loss": res.loss, "peak_device_memory_GB": res.peak_gb,
                       "library_gemm_selection": gemm_selection,
                       "own_or_library_gemm": _gemm_tuner_record()},
        }
        # whole-step model FLOPs (analytic forward count == FlopCounterMode; backward = 2x) against the dense bf16 peak
        if wl.get("fwd_gflop_per_image"):
            tf = 3 * wl["fwd_gflop_per_image"] * images / elapsed / 1e3
            out["model_flops"] = {"fwd_GFLOP_per_image": wl["fwd_gflop_per_image"], "achieved_TFLOPs": tf,
                                  "frac_of_dense_bf16_peak": tf / (MFMA_PEAK_TFLOPS * world) if args.dtype == "bf16" else None}
        if world > 1:
            out["rccl"] = res.rccl
        if res.timings:
            out["roofline"] = roofline_of(res.timings, elapsed, detail=args.kernel_table)
            if args.kernel_table:
                per = out["roofline"]["per_kernel"]
                for tag, v in sorted(per.items(), key=lambda kv: -kv[1]["share_of_step"]):
                    print(f"{1e3 * v['share_of_step'] * elapsed / args.steps:8.3f} ms/step {v['launches'] / args.steps:6.1f} x {v['avg_us']:8.1f} us "
                          f"{v['TFLOP/s']:7.0f} TF/s {v['GB/s']:6.0f} GB/s  {tag}", file=sys.stderr)
    # the reference trains in fp32 (training/train_config.py:95 `precision: int = 32`): same workload, same batch, fp32
    # activations, a few steps -- so the reference's own precision is measured next to the bf16 headline
    if world == 1 and args.dtype == "bf16" and not args.no_fp32_companion and not args.graph and not args.tune_gemm:
        try:  # (a companion line must never cost the headline line: its failure is recorded under its key)
            k = max(2, min(args.steps, 8))
            r32 = run_workload(ctx, "fp32", k, 2, timing=False)
            out["fp32"] = {"value": args.batch * k / r32.elapsed, "unit": "images/s", "ms_per_step": 1e3 * r32.elapsed / k, "steps": k,
                           "warmup": 2, "batch_per_gpu": args.batch, "workload": wl["name"], "final_loss": r32.loss,
                           "peak_device_memory_GB": r32.peak_gb,
                           "gemm": "bf16x3: every Linear product as one bf16 GEMM of three-fold depth over hi / lo splits, fp32 accumulation "
                                   "(ops.FP32_GEMM, csrc/split3.hip; logits 7e-6 of the oracle at this size, tests/test_gpu_baseline_configs.py)",
                           "note": "fp32 activations and MFMA-f32 attention kernels; library GEMMs with " + (
                               "TunableOp picks" if os.path.exists(tuned.replace("_bf16.csv", "_fp32.csv")) and not args.no_tuned_gemm else "the default heuristic")}
            # the exact-fp32 form of the same step (library fp32 GEMMs, v_mfma_f32 weight gradients): the reference for the line above
            from heal_swin_amd import ops as _ops
            prev_mode, _ops.FP32_GEMM = _ops.FP32_GEMM, "strict"
            try:
                rs = run_workload(ctx, "fp32", 3, 1, timing=False)
            finally:
                _ops.FP32_GEMM = prev_mode
            out["fp32"]["strict_fp32_gemm"] = {"value": args.batch * 3 / rs.elapsed, "unit": "images/s", "ms_per_step": 1e3 * rs.elapsed / 3, "steps": 3,
                                               "warmup": 1, "final_loss": rs.loss, "note": "HS_FP32_GEMM=strict"}
        except Exception as e:  # noqa: BLE001
            out.setdefault('fp32', {})
            out['fp32'] = {**(out['fp32'] if isinstance(out['fp32'], dict) else {}), "error": f"{type(e).__name__}: {str(e)[:300]}"}
            import gc
            gc.collect()
            torch.cuda.empty_cache()
    # BASELINE configs[4] at its stated size next to it: HEAL-SWIN-T, nside 256, 8 base pixels, depth head (f_out = 1), fp32, masked
    # L1 loss; batch 2 per GPU as in the reference's run configs (run_configs/*/..._train_run_config.py: batch_size 2)
    if world == 1 and args.dtype == "bf16" and args.workload == "B256" and not args.no_fp32_companion and not args.graph and not args.tune_gemm:
        try:  # (a companion line must never cost the headline line: its failure is recorded under its key)
            dctx = types.SimpleNamespace(**{**vars(ctx), "wl": WORKLOADS["D256"], "batch": 2})
            kd = 5
            rd = run_workload(dctx, "fp32", kd, 2, timing=False)
            out["depth_fp32"] = {"value": 2 * kd / rd.elapsed, "unit": "images/s", "ms_per_step": 1e3 * rd.elapsed / kd, "steps": kd, "warmup": 2,
                                 "batch_per_gpu": 2, "workload": WORKLOADS["D256"]["name"], "final_loss": rd.loss,
                                 "model_TFLOPs": 3 * WORKLOADS["D256"]["fwd_gflop_per_image"] * 2 * kd / rd.elapsed / 1e3,
                                 "note": "BASELINE configs[4]: fp32 activations (the reference's precision), MFMA-f32 attention / weight-gradient "
                                         "kernels, fp32 library GEMMs; parity at this size: tests/test_gpu_model.py::test_depth_head_fp32_*[256]"}
        except Exception as e:  # noqa: BLE001
            out.setdefault('depth_fp32', {})
            out['depth_fp32'] = {**(out['depth_fp32'] if isinstance(out['depth_fp32'], dict) else {}), "error": f"{type(e).__name__}: {str(e)[:300]}"}
            import gc
            gc.collect()
            torch.cuda.empty_cache()
    # what INTEGRATION.md level 1 delivers when NOTHING but the import line of the reference changes: no gradient sink (autograd
    # accumulates .grad, AccumulateGrad launches included), nn.CrossEntropyLoss on the materialised fp32 logits, torch.optim.Adam as
    # the reference builds it, eager -- in the reference's precision (fp32) and with `model.compute_dtype = torch.bfloat16`
    if world == 1 and args.dtype == "bf16" and not args.drop_in and not args.no_companions and not args.graph and not args.tune_gemm:
        try:  # (a companion line must never cost the headline line: its failure is recorded under its key)
            dctx = types.SimpleNamespace(**{**vars(ctx), "args": argparse.Namespace(**{**vars(args), "drop_in": True})})
            out["drop_in"] = {"workload": wl["name"], "batch_per_gpu": args.batch, "unit": "images/s",
                              "what": "only `from heal_swin_amd.models_torch.swin_hp_transformer import ...` differs from the reference's trainer: "
                                      "plain .grad accumulation, nn.CrossEntropyLoss(logits, masks.long()), torch.optim.Adam, eager"}
            for tag, dt, k in (("bf16", "bf16", 6), ("fp32", "fp32", 3)):
                rdi = run_workload(dctx, dt, k, 2, timing=False)
                out["drop_in"][tag] = {"value": args.batch * k / rdi.elapsed, "ms_per_step": 1e3 * rdi.elapsed / k, "steps": k, "warmup": 2,
                                       "final_loss": rdi.loss, "peak_device_memory_GB": rdi.peak_gb}
            out["drop_in"]["bf16"]["of_the_headline_step"] = (elapsed / args.steps) / (out["drop_in"]["bf16"]["ms_per_step"] * 1e-3)
        except Exception as e:  # noqa: BLE001
            out.setdefault('drop_in', {})
            out['drop_in'] = {**(out['drop_in'] if isinstance(out['drop_in'], dict) else {}), "error": f"{type(e).__name__}: {str(e)[:300]}"}
            import gc
            gc.collect()
            torch.cuda.empty_cache()
    # the reference's activation-memory policy (use_checkpoint: every block recomputed in the backward) on the same workload: what the
    # 110 GB (bf16) / 260 GB (fp32) of the lines above shrink to, and what the recomputation costs
    if world == 1 and args.dtype == "bf16" and not args.use_checkpoint and not args.drop_in and not args.no_companions and not args.graph and not args.tune_gemm:
        try:  # (a companion line must never cost the headline line: its failure is recorded under its key)
            wck = dict(wl, cfg=dict(wl["cfg"], use_checkpoint=True), name=wl["name"] + " use_checkpoint")
            kctx = types.SimpleNamespace(**{**vars(ctx), "wl": wck})
            out["use_checkpoint"] = {"workload": wck["name"], "batch_per_gpu": args.batch, "unit": "images/s",
                                     "what": "SwinHPTransformerConfig.use_checkpoint=True (swin_hp_transformer.py:541-542), everything else as the headline / fp32 lines"}
            for tag, dt, k in (("bf16", "bf16", 5), ("fp32", "fp32", 3)):
                if tag == "fp32" and args.no_fp32_companion:
                    continue
                rck = run_workload(kctx, dt, k, 2, timing=False)
                out["use_checkpoint"][tag] = {"value": args.batch * k / rck.elapsed, "ms_per_step": 1e3 * rck.elapsed / k, "steps": k, "warmup": 2,
                                              "final_loss": rck.loss, "peak_device_memory_GB": rck.peak_gb}
        except Exception as e:  # noqa: BLE001
            out.setdefault('use_checkpoint', {})
            out['use_checkpoint'] = {**(out['use_checkpoint'] if isinstance(out['use_checkpoint'], dict) else {}), "error": f"{type(e).__name__}: {str(e)[:300]}"}
            import gc
            gc.collect()
            torch.cuda.empty_cache()
    # the same step replayed from ONE HIP graph (heal_swin_amd.graphs): what host launch latency costs the eager line above
    if world == 1 and args.dtype == "bf16" and not args.graph and not args.no_graph_companion and not args.paper_drop_rates and not args.tune_gemm:
        try:  # (a companion line must never cost the headline line: its failure is recorded under its key)
            gctx = types.SimpleNamespace(**{**vars(ctx), "args": argparse.Namespace(**{**vars(args), "graph": True})})
            kg = max(2, min(args.steps, 8))
            rg = run_workload(gctx, "bf16", kg, 2, timing=False)
            out["graph_replay"] = {"value": args.batch * kg / rg.elapsed, "unit": "images/s", "ms_per_step": 1e3 * rg.elapsed / kg, "steps": kg,
                                   "eager_over_graph": (elapsed / args.steps) / (rg.elapsed / kg),
                                   "note": "whole step (zero_grad, fwd, CE, bwd, Adam) captured once and replayed; the headline `value` stays the "
                                           "eager step because the roofline brackets need per-launch HIP events"}
        except Exception as e:  # noqa: BLE001
            out.setdefault('graph_replay', {})
            out['graph_replay'] = {**(out['graph_replay'] if isinstance(out['graph_replay'], dict) else {}), "error": f"{type(e).__name__}: {str(e)[:300]}"}
            import gc
            gc.collect()
            torch.cuda.empty_cache()
    # the reference's own model next to the headline: BASELINE configs[1] (T @ nside 128) and the paper config (T @ nside 256,
    # ring_shift + cosine attention + v2 norms), bf16, same batch, eager AND replayed from one HIP graph (these steps are short
    # enough for host launch latency to matter: the faster of the two is the value, `launch` says which)
    if world == 1 and args.dtype == "bf16" and args.workload == "B256" and not args.no_companions and not args.graph and not args.tune_gemm:
        try:  # (a companion line must never cost the headline line: its failure is recorded under its key)
            out["companions"] = {}
            for key in COMPANION_WORKLOADS:
                w = WORKLOADS[key]
                cctx = types.SimpleNamespace(**{**vars(ctx), "wl": w})
                re_ = run_workload(cctx, "bf16", 8, 2, timing=True)
                gctx = types.SimpleNamespace(**{**vars(cctx), "args": argparse.Namespace(**{**vars(args), "graph": True})})
                rg_ = run_workload(gctx, "bf16", 8, 2, timing=False)
                best = min(re_.elapsed, rg_.elapsed)
                tf = 3 * w["fwd_gflop_per_image"] * args.batch * 8 / best / 1e3
                rl = roofline_of(re_.timings, re_.elapsed)
                out["companions"][key] = {
                    "workload": w["name"], "value": args.batch * 8 / best, "unit": "images/s", "batch_per_gpu": args.batch, "steps": 8, "warmup": 2,
                    "launch": "hip graph replay" if rg_.elapsed < re_.elapsed else "eager",
                    "ms_per_step_eager": 1e3 * re_.elapsed / 8, "ms_per_step_graph": 1e3 * rg_.elapsed / 8, "final_loss": re_.loss,
                    "achieved_TFLOPs": tf, "frac_of_dense_bf16_peak": tf / MFMA_PEAK_TFLOPS,
                    "attention_roofline": {"bound": "hbm", "achieved": rl["achieved"], "peak": rl["peak"], "unit": "GB/s", "frac": rl["frac"],
                                           "avg_launch_us": rl["avg_launch_us"], "launches": rl["launches"], "measured_in": "the eager run"}}
            # the paper's ACTUAL training configuration: the same model with drop_rate = attn_drop_rate = drop_path_rate = 0.1
            # (run_configs/segmentation/swin_hp_woodscape_train_run_config.py:50-51 + the config default :715), eager and replayed (the
            # replay draws new masks per step through the library's seed counter, hs_set_seed_epoch)
            w = WORKLOADS["T256"]
            wd = dict(w, cfg=dict(w["cfg"], drop_rate=0.1, attn_drop_rate=0.1, drop_path_rate=0.1), name=w["name"] + " drop 0.1/0.1/0.1")
            dctx = types.SimpleNamespace(**{**vars(ctx), "wl": wd, "args": argparse.Namespace(**{**vars(args), "paper_drop_rates": True})})
            rd_ = run_workload(dctx, "bf16", 8, 2, timing=False)
            gdctx = types.SimpleNamespace(**{**vars(dctx), "args": argparse.Namespace(**{**vars(args), "paper_drop_rates": True, "graph": True})})
            rdg_ = run_workload(gdctx, "bf16", 8, 2, timing=False)
            base = out["companions"]["T256"]
            bestd = min(rd_.elapsed, rdg_.elapsed)
            out["companions"]["T256_paper_drop"] = {
                "workload": wd["name"], "value": args.batch * 8 / bestd, "unit": "images/s", "batch_per_gpu": args.batch, "steps": 8, "warmup": 2,
                "launch": "hip graph replay" if rdg_.elapsed < rd_.elapsed else "eager",
                "ms_per_step_eager": 1e3 * rd_.elapsed / 8, "ms_per_step_graph": 1e3 * rdg_.elapsed / 8, "final_loss": rd_.loss,
                "ratio_to_no_drop_eager": base["ms_per_step_eager"] / (1e3 * rd_.elapsed / 8),
                "note": "in-kernel counter-based dropout / DropPath: attention probabilities, GELU epilogues, LayerNorm kernels, the fused "
                        "stage-0 Mlp block (train-mode form); the fused WindowAttention module kernel has no stochastic form (time-neutral at nH = 3)"}
        except Exception as e:  # noqa: BLE001
            out.setdefault('companions', {})
            out['companions'] = {**(out['companions'] if isinstance(out['companions'], dict) else {}), "error": f"{type(e).__name__}: {str(e)[:300]}"}
            import gc
            gc.collect()
            torch.cuda.empty_cache()
    if rank == 0 and res.timings and world == 1 and not args.no_kernel_timing:
        # the WindowAttention MODULE (qkv Linear + fused core + proj Linear, forward + input gradients + weight gradients) at the
        # step's shapes: north_star's ">= 40 % MFMA utilisation in windowed attention" is a statement about this unit, not about
        # the 32-flop/B core kernels the `roofline` object above brackets
        try:
            out["roofline"]["module"] = module_roofline(ctx)
        except Exception as e:  # noqa: BLE001
            out["roofline"]["module"] = {"error": repr(e)}
        if not args.no_pmc_traffic:
            recs = measure_pmc_traffic(args.workload, args.batch)
            if recs is not None:
                out["roofline"] = {**roofline_of(res.timings, elapsed, detail=args.kernel_table, traffic_records=recs),
                                   "module": out["roofline"]["module"]}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def roofline_of(timings, elapsed, detail=False, traffic_records=None):
    """attention roofline = attention flops of the timed launches (fwd + bwd) / their measured time against the
    dense bf16 MFMA peak; the HBM view of the same launches (algorithmic bytes / time against 8 TB/s) is kept beside it
    because a core-only attention kernel (32 flop/B) is bandwidth-bound by construction."""
    agg = {}
    for tag, s, e, nbytes, flops in timings:
        a = agg.setdefault(tag, [0.0, 0, 0, 0, {}])
        a[0] += s.elapsed_time(e) * 1e-3
        a[1] += nbytes
        a[2] += flops
        a[3] += 1
        a[4][nbytes] = a[4].get(nbytes, 0) + 1
    # the attention CORE launches (hs_window_attn_fwd / _bwd: the kernel the review names).  Blocks whose forward runs inside the
    # fused module kernel (hs_window_attn_module_fwd_train: norm1 + qkv + core + proj + residual, stage 0) have no core forward
    # launch; that kernel is reported beside the aggregate with its own algorithmic bytes, not mixed into it
    attn = {t: a for t, a in agg.items() if t in ("window_attn_fwd", "window_attn_bwd")}
    fused = agg.get("window_attn_module_fwd_train")
    tot_t = sum(a[0] for a in attn.values())
    tot_b = sum(a[1] for a in attn.values())
    tot_f = sum(a[2] for a in attn.values())
    launches = sum(a[3] for a in attn.values())
    tf = tot_f / tot_t / 1e12
    gbs = tot_b / tot_t / 1e9
    # SURVEY 8d counts the attention-core flops of fwd + bwd as 3 x forward (backward = 2 x for contractions); the kernels'
    # own count (tot_f) includes the backward's recomputed score tile (forward 4, backward 10 units of B N C Ws)
    fwd_f = sum(a[2] for t, a in attn.items() if t.endswith("_fwd"))
    bwd_f = sum(a[2] for t, a in attn.items() if t.endswith("_bwd"))
    tf_8d = (fwd_f + 0.8 * bwd_f) / tot_t / 1e12  # (the backward's own count is 10 units of B N C Ws: 8 of them are 2 x forward)
    traffic, traffic_source = pmc_traffic_per_launch({t: a for t, a in attn.items() if t in ("window_attn_fwd", "window_attn_bwd")},
                                                     traffic_records)
    return {
        "kernel": " + ".join(sorted(attn)) + " (fused shift / window partition / attention / reverse)",
        # a core-only attention kernel is 32 flop/B: HBM is the bound that applies; the MFMA fraction is the
        # north_star's target metric and is carried beside it
        "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "fused_module_forward": None if fused is None else {
            "kernel": "window_attn_module_fwd_train (norm1 + qkv + shift / window attention + proj + residual in one launch, writes what the "
                      "backward reads; replaces the core forward launch of these blocks)",
            "launches": fused[3], "avg_launch_us": 1e6 * fused[0] / fused[3], "algorithmic_GBs": fused[1] / fused[0] / 1e9,
            "frac_of_hbm_peak": fused[1] / fused[0] / 1e9 / HBM_PEAK_GBS, "module_TFLOPs": fused[2] / fused[0] / 1e12,
            "algorithmic_bytes": "x in (twice: residual) + out + LayerNorm(x) + qkv + attention output = 9 C * 2 B per token"},
        # every attention launch of the step in ONE figure: the core launches above plus the fused module forwards (each with its own
        # algorithmic bytes) -- sum of bytes over sum of launch times
        "all_attention_launches": {"launches": launches + (fused[3] if fused else 0),
                                   "achieved": (tot_b + (fused[1] if fused else 0)) / (tot_t + (fused[0] if fused else 0)) / 1e9, "unit": "GB/s",
                                   "frac": (tot_b + (fused[1] if fused else 0)) / (tot_t + (fused[0] if fused else 0)) / 1e9 / HBM_PEAK_GBS},
        "algorithmic_bytes_per_launch": tot_b / launches,
        # what a pure read stream reaches on this chip (tools/microbench/fill_rate.hip, 1 GB working set, every CU
        # streaming: profiles/archive_r01_r04/r02_microbench_fill_rate.txt) -- the vendor figure above is the contract's denominator
        "measured_read_ceiling_GBs": 6300.0, "frac_of_measured_ceiling": gbs / 6300.0,
        "mfma": {"achieved": tf_8d, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf_8d / MFMA_PEAK_TFLOPS,
                 "flop_count": "SURVEY 8d: attention-core flops, fwd + bwd = 3 x forward",
                 "achieved_counting_recompute": tf, "frac_counting_recompute": tf / MFMA_PEAK_TFLOPS},
        "algorithmic_flops_per_launch": tot_f / launches,
        "avg_launch_us": 1e6 * tot_t / launches, "launches": launches, "share_of_step": tot_t / elapsed,
        "per_kernel": {tag: {"launches": a[3], "avg_us": 1e6 * a[0] / a[3], "GB/s": a[1] / a[0] / 1e9,
                             "TFLOP/s": a[2] / a[0] / 1e12, "share_of_step": a[0] / elapsed}
                       for tag, a in (agg if detail else fold_gemm_tags(agg)).items()},
    }


def module_roofline(ctx, reps=3):
    """Event-timed forward + backward of every distinct WindowAttention module of the workload (the block's attention branch:
    qkv Linear -> fused shift / window / attention core -> proj Linear, with their input- and weight-gradient kernels) on random
    bf16 activations of the step's shapes, weighted by the number of blocks of that shape per step.  Standalone launches on the
    idle GPU after the timed region (inside the step these kernels are interleaved with the norms and the MLP)."""
    from heal_swin_amd.models_torch.swin_hp_transformer import SwinTransformerBlock

    wl, dev, batch = ctx.wl, ctx.dev, getattr(ctx, "batch", None) or ctx.args.batch
    model, cfg, spec = build_model(wl)
    model = model.to(dev).train()
    model.compute_dtype = torch.bfloat16
    groups = {}
    for m in model.modules():
        if isinstance(m, SwinTransformerBlock):
            groups.setdefault((m.dim, m.input_resolution, m._shifted), []).append(m)
    tot_s, tot_flop, per = 0.0, 0.0, []
    for (C, N, shifted), blks in sorted(groups.items()):
        blk = blks[0]
        x = torch.randn(batch, N, C, device=dev, dtype=torch.bfloat16, requires_grad=True)
        dy = torch.randn(batch, N, C, device=dev, dtype=torch.bfloat16)
        ts = []
        for it in range(reps + 1):
            for p_ in blk.parameters():
                p_.grad = None
            x.grad = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            blk._attention_branch(x).backward(dy)
            e1.record()
            torch.cuda.synchronize(dev)
            if it:
                ts.append(e0.elapsed_time(e1) * 1e-3)
        t = min(ts)
        fwd_flop = batch * (8.0 * N * C * C + 4.0 * N * blk.window_size * C)  # qkv 6NC^2 + proj 2NC^2 + QK^T and PV
        tot_s += t * len(blks)
        tot_flop += 3.0 * fwd_flop * len(blks)
        per.append({"C": C, "tokens": N, "shifted": bool(shifted), "blocks": len(blks), "ms_fwd_bwd": 1e3 * t,
                    "TFLOP/s": 3.0 * fwd_flop / t / 1e12})
        del x, dy
    del model
    torch.cuda.empty_cache()
    tf = tot_flop / tot_s / 1e12
    return {"kernel": "WindowAttention module: qkv Linear + window_attn core + proj Linear, fwd + dgrad + wgrad", "bound": "mfma",
            "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS,
            "module_fwd_GFLOP_per_image": tot_flop / 3.0 / batch / 1e9, "ms_per_step": 1e3 * tot_s,
            "flop_count": "3 x forward (8 N C^2 + 4 N Ws C per block), SURVEY 8d",
            "measured": f"standalone, HIP events, min of {reps} after 1 warm-up per distinct (C, tokens, shifted)", "per_shape": per}


def fold_gemm_tags(agg):
    """The GEMM launches are tagged per shape (`--kernel-table`); the JSON line carries one entry per family."""
    out = {}
    for tag, a in agg.items():
        fam = "hs_gemm_nt" if tag.startswith("hs_gemm_nt") else ("library_gemm (hipBLASLt)" if tag.startswith("lib ") else tag)
        o = out.setdefault(fam, [0.0, 0, 0, 0, {}])
        o[0] += a[0]
        o[1] += a[1]
        o[2] += a[2]
        o[3] += a[3]
    return out


def _gemm_tuner_record():
    """What decided hs_gemm_nt-or-library for the bias / residual products of this process (ops.GemmTuner: first-call trials)."""
    from heal_swin_amd import ops
    t = ops.GEMM_TUNER
    return {"tuner": bool(ops.GEMM_TUNE), "rule": "GELU / GELU' epilogues: hs_gemm_nt; bias / residual products: a first-call trial of both on "
                                                  "synthetic operands per (rows bucket, n, k), the class rule where no trial ran",
            "trials_us_own_vs_library": {f"m~2^{k[0]} n={k[1]} k={k[2]}": list(v) + ["own" if t.picks[k] else "library"] for k, v in sorted(t.trials.items())}}


class _NoSink:
    """--drop-in: stands where the gradient sink stands in the step, and does what a plain trainer does there."""
    buckets = ()
    opt = None

    def zero_grad(self):
        self.opt.zero_grad()  # set_to_none=True, the PyTorch / Lightning default

    def finish(self):
        pass

    def remove(self):
        pass


def run_workload(ctx, dtype_name, steps, warmup, timing):
    """Build the model / optimizer / gradient exchange for `dtype_name`, run `warmup` untimed and `steps` timed steps
    (barrier + synchronize on both sides, MAX over ranks) and release everything again."""
    undo = []  # process-wide registrations of this run (the library's replay counter): withdrawn on EVERY way out, a failed
    try:       # capture included -- the companions swallow exceptions and carry on in this process
        return _run_workload(ctx, dtype_name, steps, warmup, timing, undo)
    finally:
        for fn in reversed(undo):
            fn()


def _run_workload(ctx, dtype_name, steps, warmup, timing, undo):
    import gc

    import torch.distributed as dist
    from heal_swin_amd import ops
    from heal_swin_amd.losses import depth_l1_loss, seg_loss
    from heal_swin_amd.parallel import GradBucketAllReduce

    args, wl, dev, world, rank = ctx.args, ctx.wl, ctx.dev, ctx.world, ctx.rank
    batch = getattr(ctx, "batch", None) or args.batch
    torch.cuda.reset_peak_memory_stats(dev)
    model, cfg, spec = build_model(wl)
    model = model.to(dev).train()
    model.compute_dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    drop_in = bool(getattr(args, "drop_in", False))
    ddp = None
    if drop_in:
        # what the reference's trainer does around the module and nothing of this package besides it
        if args.graph:
            raise SystemExit("--drop-in is the eager trainer path; it is not captured")
        dp = _NoSink()
        if world > 1:
            ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], find_unused_parameters=False)  # train.py:187
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)  # training/optimizer.py:57-66
        dp.opt = opt
    else:
        dp = GradBucketAllReduce(model.parameters(), async_wgrad=args.async_wgrad,
                                 reserved_cus=args.reserved_cus if args.reserved_cus == "auto" else int(args.reserved_cus),
                                 comm_dtype=torch.bfloat16 if args.comm_dtype == "bf16" else None)
    if drop_in:
        pass
    elif args.torch_adam:
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True, capturable=args.graph)  # ref: training/optimizer.py:57-66
    else:
        # the same Adam on flat parameter / moment buffers laid out like the gradient buckets: one launch per bucket, which also
        # writes the bf16 parameter copies the next forward reads (heal_swin_amd/optim.py, csrc/adam.hip)
        from heal_swin_amd.optim import FlatAdam
        opt = FlatAdam(model.parameters(), dp, lr=1e-4, model=model if dtype_name == "bf16" else None)

    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    imgs = torch.randint(0, 256, (batch, 3, spec["dim_in"]), generator=g, device=dev, dtype=torch.uint8)
    if wl.get("task") == "depth":
        # positive depths with ~4 % infinite (background) targets, as in the data set
        labels = torch.randn((batch, spec["dim_in"]), generator=g, device=dev).abs() * 10
        labels[torch.rand((batch, spec["dim_in"]), generator=g, device=dev) < 0.04] = float("inf")
        loss_fn = depth_l1_loss  # training/loss_depth_regression.py:41-53
    else:
        labels = torch.randint(0, spec["f_out"], (batch, spec["dim_in"]), generator=g, device=dev, dtype=torch.uint8)
        loss_fn = seg_loss

    fused_loss = wl.get("task") != "depth" and not args.unfused_loss and not drop_in
    if drop_in and wl.get("task") != "depth":
        ce = torch.nn.CrossEntropyLoss()  # model_lightning_swin_hp.py:39-45
        loss_fn = lambda logits, y: ce(logits, y.long())  # noqa: E731  (`masks.long()`, :104-111)
    net = ddp if ddp is not None else model

    def step():
        dp.zero_grad()
        if drop_in:
            loss = loss_fn(net(imgs.float()), labels)
        elif fused_loss:  # model + the caller's CrossEntropyLoss in one call: the loss rides on the decoder tail's kernels
            loss = model.forward_seg_loss(imgs.float(), labels)
        else:
            logits = model(imgs.float())  # the caller's `.float()` (model_lightning_swin_hp.py:61)
            loss = loss_fn(logits, labels)
        loss.backward()
        dp.finish()
        opt.step()
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        step()
    sync()
    gemm_policy = None
    if world > 1 and dtype_name == "bf16" and not drop_in:
        # With CUs reserved for RCCL the sink routes every bf16 Linear to hs_gemm_nt (whose grids honour the reservation), which costs
        # ~4 % on an idle chip and saves 16 % if the exchange's kernels do stay resident (profiles/archive_r01_r04/r04_cu_contention.json).  Which of the
        # two this node's exchange looks like is MEASURED here instead of assumed: three steps under each policy (max over ranks), the
        # faster one runs the timed region.  Every rank takes the same decision (the times are all-reduced).
        trial = {}
        for pref in (True, False):
            ops.RT.prefer_own_gemm = pref
            step()
            sync()
            t0 = time.perf_counter()
            for _ in range(3):
                step()
            sync()
            t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            trial[pref] = float(t.item()) / 3
        ops.RT.prefer_own_gemm = trial[True] <= trial[False]
        gemm_policy = {"all_linear_products_on_hs_gemm_nt": bool(ops.RT.prefer_own_gemm),
                       "trial_ms_per_step": {"own": round(1e3 * trial[True], 3), "per_shape_with_library": round(1e3 * trial[False], 3)},
                       "note": "decided at run time from 3 + 3 untimed steps under the live exchange"}
    if args.tune_gemm:
        torch.cuda.tunable.tuning_enable(False)  # every shape was met during the warm-up; PyTorch writes the file at exit
    epoch = None
    if args.graph and world > 1:
        # collectives are not captured: two graphs around the sink's eager exchange (heal_swin_amd.graphs.GraphedTrainStep does the same)
        if args.paper_drop_rates or drop_in:
            raise SystemExit("--graph with --gpus > 1: no dropout replay counter / drop-in wiring on this path")

        def fwd_bwd_local():
            with dp.no_sync():
                dp.zero_grad()
                loss = model.forward_seg_loss(imgs.float(), labels) if fused_loss else loss_fn(model(imgs.float()), labels)
                loss.backward()
                dp.finish()
            return loss
        graph, graph_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = fwd_bwd_local()
        dp.finish()
        with torch.cuda.graph(graph_opt):
            opt.step()

        def step():  # noqa: F811
            graph.replay()
            dp.finish()
            graph_opt.replay()
            return static_loss

        eager_step = None
        step()
        sync()
    elif args.graph:
        eager_step = step
        if args.paper_drop_rates:
            # the kernels' host-drawn dropout seeds are frozen into the graph; the library's replay counter is not (hs_set_seed_epoch:
            # every mask generator adds counter x odd constant to its seed, the captured step ends with counter += 1)
            from heal_swin_amd import _lib as _L
            epoch = torch.zeros(1, dtype=torch.int64, device=dev)

            def _unregister(keep=epoch):  # (holds the tensor: the library's pointer is cleared before the memory can go)
                torch.cuda.synchronize(dev)
                _L.lib.hs_set_seed_epoch(None)
            undo.append(_unregister)
            _L.check(_L.lib.hs_set_seed_epoch(_L.ptr(epoch)), "hs_set_seed_epoch")
            plain_step = step

            def eager_step():
                out_ = plain_step()
                epoch.add_(1)
                return out_
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_loss = eager_step()

        def step():  # noqa: F811
            graph.replay()
            return static_loss

        step()
        sync()
    if timing:
        ops.KERNEL_TIMINGS = []
        # the roofline object needs the attention launches only; bracketing every GEMM as well (--kernel-table) costs ~2 % of
        # the step in event packets
        ops.TIMED_PREFIXES = None if args.kernel_table else ("window_attn",)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    sync()
    elapsed = time.perf_counter() - t0
    timings, ops.KERNEL_TIMINGS = ops.KERNEL_TIMINGS, None
    rccl = None
    if world > 1:
        assert dist.get_world_size() == world == args.gpus
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        own = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(own, t)
        step_ms_per_rank = [round(1e3 * float(v.item()) / steps, 3) for v in own]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the exchange on its own: every gradient bucket all-reduced back to back, timed per rank with events
        reps = 5
        if drop_in:  # (torch DDP owns its buckets: time an all-reduce of the same volume)
            dp.buckets = [torch.zeros(sum(p.numel() for p in model.parameters()), device=dev)]
        for flat in dp.buckets:
            dist.all_reduce(flat)
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for flat in dp.buckets:
                dist.all_reduce(flat)
        e1.record()
        torch.cuda.synchronize(dev)
        mine = torch.tensor([e0.elapsed_time(e1) / reps], device=dev, dtype=torch.float64)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        nbytes = sum(f.numel() * 4 for f in dp.buckets)
        ms = [float(v.item()) for v in per_rank]
        try:
            rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            rccl_version = None
        # one more (untimed) step with the sink's timeline on: when, in milliseconds after the step began, did each gradient bucket
        # start its exchange (from a hook during the backward, or only in finish()), and when had all of them been waited for
        timeline = None
        if not drop_in:
            dp.record_timeline(True)
            t_begin = torch.cuda.Event(enable_timing=True)
            t_begin.record()
            step()
            timeline = dp.timeline_ms(t_begin)
            dp.record_timeline(False)
        rccl = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": rccl_version,
                "bucket_timeline_rank0": timeline,
                "step_ms_per_rank": step_ms_per_rank, "buckets": len(dp.buckets),
                "allreduce_bytes_per_step": nbytes, "allreduce_ms_per_step_standalone_per_rank": [round(v, 3) for v in ms],
                "allreduce_bus_GBps": 2 * (world - 1) / world * nbytes / (max(ms) * 1e-3) / 1e9,
                "reserved_cus": int(__import__("heal_swin_amd")._lib.lib.hs_get_reserved_cus()), "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"),
                "gemm_policy": gemm_policy,
                "exchange": ("torch.nn.parallel.DistributedDataParallel (its own buckets and hooks)" if drop_in else
                             f"{args.comm_dtype} wire format of fp32 flat buckets, async all-reduce launched from gradient hooks during backward")}
    res = types.SimpleNamespace(elapsed=elapsed, loss=float(loss.item()), timings=timings if rank == 0 else None, rccl=rccl,
                                params_m=round(sum(p.numel() for p in model.parameters()) / 1e6, 2),
                                peak_gb=round(torch.cuda.max_memory_allocated(dev) / 1e9, 1))
    dp.remove()
    del model, dp, opt, imgs, labels, loss, step, net, ddp
    if args.graph:
        del graph, static_loss, eager_step
        if world > 1:
            del graph_opt
    del epoch  # (its registration with the library is withdrawn by run_workload's `undo`, which still holds the tensor)
    gc.collect()
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
