#!/usr/bin/env python3
"""Benchmarks of the MI355X hot path (BASELINE.json).  Default = the headline: SimCLR ResNet-18 training throughput
on synthetic wafer maps.

    python bench.py --gpus N --steps K --warmup W [--workload simclr_r18]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workloads (`--workload`; each prints ONE JSON line with the same schema, rank 0):
  simclr_r18        BASELINE configs[1] (default).  One step = fused two-view augmentation of 256 wafers per GPU
                    (written in the stem's space-to-depth layout) -> ResNet-18 forward/backward on 2 x 256 images
                    (3 x 224 x 224, bf16, BN statistics per view) -> SimCLR head -> NT-Xent (in-batch negatives) ->
                    (N > 1: bucketed RCCL all-reduce of the flat gradient arena, overlapped with the backward stages)
                    -> fused SGD.  The step replays captured hipGraphs.
  dino_vit_tiny     BASELINE configs[2]: DINO, ViT-Tiny/16, 2 x 224^2 + 6 x 96^2 crops, 64 wafers per GPU, AdamW.
  dino_vit_small    the reference's own DINOViT (ViT-S/16), same step.
  mae_vit_small_16  BASELINE configs[3]: MAE ViT-S/16, 75 % mask, 64 wafers per GPU (MixedWM38-sized 52 x 52 maps).
  mae_vit_b_32      the reference's own MAE (ViT-B/32).
  knn_allpairs      BASELINE configs[4]: 811 457 x 128 bf16 embeddings row-sharded over the ranks, ONE all-gather of the
                    shards, then every rank ranks its own rows against the whole bank (top-8); step = one batch of
                    1024 queries per rank; metric queries/s.
All device work runs in the hand-written HIP kernels of libwafer_hip.so; inputs are resident in HBM before the timed
region.  The timed region holds EXACTLY K steps between two fences (barrier + synchronize), max over ranks.

Extra objects on the line:
  roofline     : dominant kernels of the workload (conv implicit-GEMM for simclr_r18; Linear GEMMs + attention for
                 the transformers; the streaming pairwise-dot kernel for knn): algorithmic FLOPs (bytes) / summed
                 HIP-event durations of those launches on the launch stream, measured live on ONE eager step run
                 right AFTER the timed region (a timing event is a barrier packet: bracketing ~60 launches inside the
                 timed steps would cost throughput), against the dense bf16 MFMA peak (HBM peak for knn).
  cpu_baseline : the torch-CPU oracle (oracle/) on BASELINE configs[0] (bs 32, fp32), bounded sample, rank 0, N = 1:
                 augmentation and model seconds split; plus the oracle knn_predict on a 100 k x 128 slice.
  knn, augment : (simclr_r18, N = 1) the two secondary kernels of the north-star, timed with HIP events:
                 811 457 x 128 top-8 for 64 / 256 / 1024 queries in bf16 and f32 (stream + select, formula-(ii) HBM
                 fraction and dense-view TFLOP/s), and the fused augmentation kernel (views/s, HBM fraction).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0           # HBM3E, same table
# HBM-side bytes per launch of the dominant kernels (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes of this
# script, FETCH_SIZE doubled per the gfx950 correction, calibrated on sgd_step) live in profiles/traffic.json together
# with the commit and the kernel-source digest they were measured at: see traffic_entry().  Algorithmic bytes of a conv
# launch (every operand once): 230 MB.
R18_GFLOP_PER_SAMPLE = 21.76    # ResNet-18 fwd 3.627 GFLOP x 3 (fwd+bwd) x 2 views
KNN_N, KNN_D, KNN_K = 811457, 128, 8


# --------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline(seconds_budget: float = 22.0):
    """Oracle SimCLR step (torch CPU float32), BASELINE configs[0]: 1 000 synthetic wafers, bs 32, two views, SGD
    lr 0.06 x 32/256.  Two warm-up steps, then as many timed steps as fit the budget (at most one epoch = 31)."""
    import numpy as np
    import torch

    from oracle import augment as oa
    from oracle import resnet as orn
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.heads import SimCLRProjectionHead
    from ssl_wafermap_amd.models import create_model

    torch.manual_seed(0)
    cores = torch.get_num_threads()
    wafers, _ = synthetic_wafers(1000, seed=1234)
    backbone, head = create_model("resnet18", num_classes=0), SimCLRProjectionHead(512, 512, 128)
    sd = {"backbone." + k: v.clone() for k, v in backbone.state_dict().items()}
    sd.update({"projection_head." + k: v.clone() for k, v in head.state_dict().items()})
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and "running" not in k}
    state = dict(sd)
    state.update(params)
    bufs = {}
    rng = np.random.default_rng(0)
    bs = 32
    t_aug = t_model = 0.0

    def step(i, timed):
        nonlocal t_aug, t_model
        t0 = time.perf_counter()
        idx = (np.arange(bs) + i * bs) % len(wafers)
        views = []
        for _ in range(2):
            imgs = []
            for s in idx:
                w = wafers[s]
                op = oa.OP_DIENOISE if rng.random() < 0.5 else oa.OP_DPW
                d = oa.ViewDecision(op=op, noise_seed=int(rng.integers(1 << 31)),
                                    dpw_hw=oa.dpw_dims(*w.shape, oa.dpw_scale(w.shape, rng.beta(0.5, 1.5))),
                                    rot90=rng.random() < 0.5, vflip=rng.random() < 0.5, hflip=rng.random() < 0.5)
                imgs.append(oa.augment_view(w, d))
            views.append(torch.from_numpy(np.stack(imgs)))
        t1 = time.perf_counter()
        for p in params.values():
            p.grad = None
        loss, _ = orn.simclr_loss(views[0], views[1], state, 0.5, True)
        loss.backward()
        with torch.no_grad():
            orn.sgd_step({k: p for k, p in params.items()}, {k: p.grad for k, p in params.items()}, bufs,
                         lr=6e-2 * bs / 256)
        t2 = time.perf_counter()
        if timed:
            t_aug += t1 - t0
            t_model += t2 - t1
        return float(loss.detach())

    step(0, False)
    step(1, False)
    t0 = time.perf_counter()
    n = 0
    while n < 31:
        step(n + 2, True)
        n += 1
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    return {"value": round(bs * n / dt, 3), "unit": "imgs/sec", "cores": cores, "kind": "port",
            "augment_s": round(t_aug, 2), "model_s": round(t_model, 2),
            "sample": f"{n} SimCLR ResNet-18 steps at bs {bs} over 1000 synthetic wafers after 2 warm-up steps (fp32 torch "
                      f"CPU oracle; numpy augmentation in process, {t_aug:.1f} s of {dt:.1f} s)"}


def cpu_knn_baseline():
    """Oracle knn_predict (torch.mm + topk, fp32) on a 100 000 x 128 slice, 256 queries per call, k 8."""
    import torch

    from oracle import knn as ok

    g = torch.Generator().manual_seed(7)
    n, d, bq = 100_000, KNN_D, 256
    bank = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).t().contiguous()
    labels = torch.randint(0, 9, (n,), generator=g)
    q = bank.t()[:bq].contiguous()
    ok.knn_predict(q, bank, labels, 9, KNN_K, 0.1)
    t0 = time.perf_counter()
    reps = 0
    while reps < 50 and time.perf_counter() - t0 < 4.0:
        ok.knn_predict(q, bank, labels, 9, KNN_K, 0.1)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    qps = bq / dt
    return {"value": round(qps, 1), "unit": "queries/sec against 100000 x 128 (fp32)", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{reps} calls of oracle knn_predict, 256 queries x 100 000 rows, k 8",
            "extrapolated_allpairs_811457_s": round((KNN_N / qps) * (KNN_N / n), 1)}


# --------------------------------------------------------------------------------------------- secondary kernels
def knn_object(dev):
    """811 457 x 128, top-8: wm_knn_topk (stream + select) for 64 / 256 / 1024 queries, bf16 and f32, HIP events."""
    import torch

    from ssl_wafermap_amd import functional as F

    g = torch.Generator(device=dev).manual_seed(7)
    bank32 = torch.nn.functional.normalize(torch.randn(KNN_N, KNN_D, generator=g, device=dev), dim=1)
    bank16 = bank32.bfloat16()
    out = {"n": KNN_N, "d": KNN_D, "k": KNN_K, "rows": []}
    for dtype, bank in (("bf16", bank16), ("f32", bank32)):
        s = bank.element_size()
        for bq in (64, 256, 1024):
            q = bank[1000:1000 + bq].contiguous()
            for _ in range(3):
                F.knn_topk(q, bank, KNN_K)
            reps = 20
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                F.knn_topk(q, bank, KNN_K)
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / reps
            bytes_ = KNN_N * KNN_D * s + bq * KNN_D * s + bq * KNN_K * 8      # SURVEY 8d formula (ii), one batch
            # many batches, pipelined over HIP streams (functional.knn_topk_batched): the selection kernel of one batch
            # under the streaming kernel of the next -- the rate an embedding-retrieval job sees
            nb = 256 if bq <= 256 else 32   # (enough batches that the first streaming kernel and the last selection are amortised)
            qq = bank[2000:2000 + nb * bq].contiguous()
            F.knn_topk_batched(qq, bank, KNN_K, batch=bq)
            torch.cuda.synchronize()
            us_p = float("inf")
            for _ in range(3):  # best of three calls of nb batches each (host jitter shows in a single call)
                a.record()
                F.knn_topk_batched(qq, bank, KNN_K, batch=bq)
                b.record()
                torch.cuda.synchronize()
                us_p = min(us_p, a.elapsed_time(b) * 1e3 / nb)
            out["rows"].append({"dtype": dtype, "queries": bq, "us_per_batch": round(us, 1),
                                "pipelined_us_per_batch": round(us_p, 1),
                                "pipelined_hbm_frac": round(bytes_ / us_p / 1e3 / HBM_PEAK_GBS, 3),
                                "hbm_GBs": round(bytes_ / us / 1e3, 1), "hbm_frac": round(bytes_ / us / 1e3 / HBM_PEAK_GBS, 3),
                                "dense_TFLOPs": round(2.0 * bq * KNN_N * KNN_D / us / 1e6, 1),
                                "allpairs_s": round(us * 1e-6 * (KNN_N / bq), 3),
                                "traffic": traffic_entry("knn_b64")[0] if (dtype == "bf16" and bq == 64) else None})
    del bank32, bank16
    torch.cuda.empty_cache()
    return out


def augment_object(dev, ds, B):
    """The fused two-view augmentation kernel alone: 2 x B views of 224^2 in the stem's bf16 layout."""
    import numpy as np
    import torch

    rng = np.random.default_rng(0)
    tr = ds.transform
    from ssl_wafermap_amd.transforms.augmentations import PARAM_DTYPE

    params = tr.sample(ds.store, np.arange(B), rng)
    # the decisions are uploaded once (as graph.GraphedTrainStep keeps them in a static device buffer): the timed
    # loop is the kernel launch alone
    pdev = [torch.from_numpy(np.ascontiguousarray(p).view(np.uint8).reshape(-1).copy()).to(dev) for p in params]
    assert all(p.dtype == PARAM_DTYPE for p in params)
    for _ in range(3):
        tr.launch(ds.store, params, B, "s2d_bf16", params_dev=pdev)
    reps = 50
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def eager():
        for _ in range(reps):
            tr.launch(ds.store, params, B, "s2d_bf16", params_dev=pdev)

    # the launches replayed from one hipGraph (as in the training step): a Python call per launch costs about as much
    # host time as the kernel takes on the device, and the figure then follows the host's speed (46 - 57 us box to box)
    run, how = eager, "eager launches back to back"
    try:
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            eager()
        run, how = g.replay, f"{reps} launches replayed from one hipGraph"
        run()
    except Exception as e:   # (a build whose launch path cannot be captured: the eager loop)
        print(f"[bench] augment: graph capture unavailable ({type(e).__name__}: {e})", file=sys.stderr)
        torch.cuda.synchronize()
    a.record()
    run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / reps
    # what a pure store stream of the same bytes takes on this device (torch fill kernel over the same tensor)
    buf = tr.launch(ds.store, params, B, "s2d_bf16", params_dev=pdev).stacked
    a.record()
    for _ in range(reps):
        buf.fill_(1.0)
    b.record()
    torch.cuda.synchronize()
    fill_us = a.elapsed_time(b) * 1e3 / reps
    views = 2 * B
    hw = float(np.mean(ds.store.heights_np[:B].astype(np.int64) * ds.store.widths_np[:B]))
    bytes_ = views * (hw + 3 * 224 * 224 * 2)       # read H*W uint8, write 3 x 224 x 224 bf16 per view
    written = views * 112 * 112 * 16 * 2            # what the s2d layout actually stores (16 channels, 12 used)
    return {"views": views, "us_per_launch": round(us, 1), "views_per_sec": round(views / us * 1e6, 0),
            "algorithmic_GBs": round(bytes_ / us / 1e3, 1), "hbm_frac": round(bytes_ / us / 1e3 / HBM_PEAK_GBS, 3),
            "written_GBs": round(written / us / 1e3, 1), "fill_same_tensor_us": round(fill_us, 1), "includes": how + ", decisions resident on the device"}


# --------------------------------------------------------------------------------------------- self-launch (N > 1)
def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher (reference switch: scripts/WM811k_benchmark.py:78-85, `devices` +
    strategy="ddp"): this process -- which has touched neither torch nor the GPU -- starts the N ranks as CHILDREN
    through torch.distributed.run (one process per GPU, RCCL over xGMI), lets rank 0's JSON line through on stdout and
    returns the launcher's exit status.  Never an exec: a process is only ever replaced before anything GPU-side exists,
    and a child is simpler to reason about."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's cross-process buffers need it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def dry_run(args) -> None:
    """The launch / rendezvous / collective plumbing of an N-rank run without any kernel: every rank joins the
    process group (RCCL on GPUs, gloo on a CPU-only box or under WM_DIST_BACKEND=gloo), all-reduces a one, rank 0
    prints a line in the bench schema with "dry_run": true."""
    import torch
    import torch.distributed as dist

    from ssl_wafermap_amd import distributed as wdist

    rank, world, local = wdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ranks, backend = 1, "none"
    if world > 1:
        backend = dist.get_backend()
        dev = torch.device("cuda", 0 if os.environ.get("WM_SINGLE_DEVICE") == "1" else local) if backend == "nccl" else torch.device("cpu")
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        ranks = int(one.item())
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "imgs/sec (SimCLR ResNet18, bs=256, 224^2)", "value": None, "unit": "imgs/sec",
                          "n_gpus": world, "steps": 0, "warmup": 0, "dry_run": True, "rccl_ranks": ranks,
                          "backend": backend, "config": {"workload": args.workload}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def source_digest() -> str:
    """sha256 over the kernel sources: committed PMC traffic figures name the digest they were measured at."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted((ROOT / "self-supervised-wafermaps_amd" / "csrc").glob("*.h*")):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:12]


def traffic_entry(key):
    """(bytes per launch, source, commit, stale) of a committed rocprofv3 --pmc measurement (profiles/traffic.json,
    written by tools/pmc_traffic.py --json): a PMC pass cannot run inside the timed region, so the line carries the
    committed figure, the commit and kernel-source digest it was measured at, and says so when the kernels have
    changed since (`traffic_stale`)."""
    try:
        ent = json.loads((ROOT / "profiles" / "traffic.json").read_text()).get(key)
    except (OSError, ValueError):
        ent = None
    if not ent:
        return None, None, None, None
    stale = ent.get("kernel_digest") != source_digest()
    if stale:
        print(f"[bench] warning: roofline.traffic for {key!r} was measured at commit {ent.get('commit')} (kernel digest "
              f"{ent.get('kernel_digest')}); the kernel sources have changed since -- re-run tools/pmc_bench.sh", file=sys.stderr)
    return ent["bytes_per_launch"], ent.get("source"), ent.get("commit"), stale


# --------------------------------------------------------------------------------------------- workloads
def make_vit_workload(name, dev, B, world):
    """DINO / MAE steps on synthetic wafers -> (dataset, model, transform format, GFLOP per sample, label)."""
    from ssl_wafermap_amd.data import WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.models import MAE, DINOViT
    from ssl_wafermap_amd.transforms import BaseViewTransform, MultiCropTransform

    def vit_gflop(d, tokens, layers=12, patch=16, mlp=4):
        per_layer = 2.0 * (tokens * d * d * (4 + 2 * mlp) + 2 * tokens * tokens * d)
        return (layers * per_layer + 2.0 * (tokens - 1) * (patch * patch * 3) * d) / 1e9

    if name.startswith("dino"):
        bb = "vit_tiny" if name.endswith("tiny") else "vit_small"
        d = 192 if bb == "vit_tiny" else 384
        wafers, labels = synthetic_wafers(2048, seed=1)
        ds = WaferMapDataset(wafers, labels, transform=MultiCropTransform(), device=dev)
        model = DINOViT(None, 9, batch_size=B * world, log_rep_std=False, backbone=bb)
        g224, g96 = vit_gflop(d, 197), vit_gflop(d, 37)
        head = 2.0 * (d * 2048 + 2048 * 2048 + 2048 * 256 + 256 * 2048) / 1e9
        gflop = 2 * (g224 + head) + 3 * (2 * g224 + 6 * g96 + 8 * head)   # teacher: 2 fwd; student: 8 crops fwd + bwd
        label = f"DINO {'ViT-Tiny' if bb == 'vit_tiny' else 'ViT-S'}/16, {B} wafers/GPU/step, 2x224^2 + 6x96^2 crops, AdamW"
        cfg = "BASELINE.json configs[2]" if bb == "vit_tiny" else "the reference's DINOViT"
        return ds, model, gflop, f"{label} ({cfg})"
    bb = "vit_small_16" if "small" in name else "vit_b_32"
    fixed = 52 if bb == "vit_small_16" else None                            # MixedWM38 maps are 52 x 52
    wafers, labels = synthetic_wafers(2048, seed=1, fixed_size=fixed)
    ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(n_views=1, denoise=bb == "vit_small_16"), device=dev)
    model = MAE(None, 9, batch_size=B * world, log_rep_std=False, backbone=bb)
    d, tokens, patch = (384, 197, 16) if bb == "vit_small_16" else (768, 50, 32)
    keep = int(tokens * 0.25)
    enc = vit_gflop(d, keep, patch=patch) + 2.0 * (tokens - 1 - keep) * patch * patch * 3 * d / 1e9
    dec = vit_gflop(512, tokens, layers=1, patch=1) + 2.0 * (keep * d * 512 + (tokens - keep) * 512 * patch * patch * 3) / 1e9
    gflop = 3 * (enc + dec)
    label = f"MAE {'ViT-S/16' if bb == 'vit_small_16' else 'ViT-B/32'}, 75 % mask, {B} wafers/GPU/step, AdamW"
    cfg = "BASELINE.json configs[3]" if bb == "vit_small_16" else "the reference's MAE"
    return ds, model, gflop, f"{label} ({cfg})"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="simclr_r18",
                    choices=["simclr_r18", "dino_vit_tiny", "dino_vit_small", "mae_vit_small_16", "mae_vit_b_32", "knn_allpairs"])
    ap.add_argument("--batch", type=int, default=None, help="wafers per GPU per step (default: 256 SimCLR, 64 DINO / MAE)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the knn / augment objects of the default line")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step into hipGraphs")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: one graph + one exposed all-reduce (no stages)")
    ap.add_argument("--sharded", action="store_true",
                    help="knn_allpairs: the bank stays row-sharded (no all-gather of the embeddings); every step all-gathers "
                         "the ranks' query batches, searches the local shard and merges the per-rank top-k lists")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch, rendezvous and one all-reduce only (no kernel): checks the N-rank plumbing, also without a GPU")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher set the rank environment: this (GPU-free) process starts the ranks itself and relays their result
        raise SystemExit(launch_ranks(args))
    if args.dry_run:
        return dry_run(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    from ssl_wafermap_amd import distributed as wdist
    from ssl_wafermap_amd import ops

    rank, world, local = wdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if os.environ.get("WM_SINGLE_DEVICE") == "1":
        local = 0  # rehearsal of the N > 1 path on a one-GPU box (with WM_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, warmup, steps):
        """W untimed + exactly K timed steps between two fences; -> (wall s max over ranks, GPU-event ms per step,
        the value step_fn returned last)."""
        last = None
        for i in range(warmup):
            last = step_fn(i)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        t0 = time.perf_counter()
        ev0.record()
        for i in range(steps):
            last = step_fn(warmup + i)
        ev1.record()
        fence()
        dt = time.perf_counter() - t0
        gpu_ms = ev0.elapsed_time(ev1) / steps
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, gpu_ms, last

    def finish(out):
        from ssl_wafermap_amd import graph as wgraph
        if wgraph.HOST_TIMES and rank == 0:   # WM_STEP_HOST_TIMES=1: where the host spends a step (diagnostic, stderr)
            n = max(wgraph.HOST_TIMES.pop("steps", 1), 1)
            print("[bench] host ms per step: " + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in wgraph.HOST_TIMES.items()),
                  file=sys.stderr)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()

    def check_finite(loss):
        v = float(loss.detach())
        if not np.isfinite(v):
            # a throughput figure of a network that has diverged is not a training throughput
            print(f"[bench] rank {rank}: non-finite loss {v} after {args.warmup} + {args.steps} steps", file=sys.stderr)
            if world > 1:
                dist.destroy_process_group()
            raise SystemExit(3)
        return v

    def roofline_from(timer, sampled_steps, kernel_label, traffic_key=None, overhead_ms=0.0):
        traffic, traffic_source, traffic_commit, traffic_stale = traffic_entry(traffic_key) if traffic_key else (None,) * 4
        summ = timer.summary()   # (never corrected: the bracket overhead is reported beside it, below)
        work = sum(v["work"] for v in summ.values())
        ms = sum(v["ms"] for v in summ.values())
        ach = work / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        r = {"bound": "mfma", "kernel": kernel_label, "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
             "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
             "launches": int(sum(v["launches"] for v in summ.values())), "sampled_steps": sampled_steps,
             "kernel_ms_per_step": round(ms / max(sampled_steps, 1), 3),
             "by_kernel": {k: {"TFLOP/s": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 2),
                               "ms_per_step": round(v["ms"] / max(sampled_steps, 1), 3)} for k, v in summ.items()}}
        if traffic_source:
            r["traffic_source"], r["traffic_commit"], r["traffic_stale"] = traffic_source, traffic_commit, traffic_stale
        if overhead_ms > 0.0:
            # what the two timing events of a bracket add to a short launch, calibrated on a dependent chain of two GEMMs
            # (ops.KernelTimer.bracket_overhead_ms) -- an UPPER bound of the inflation (the calibration chain overlaps its
            # launches better than a training step does): `frac` above is the uncorrected, conservative figure
            r["bracket_overhead_us_per_launch"] = round(overhead_ms * 1e3, 2)
            ms2 = max(ms - overhead_ms * r["launches"], 1e-9)
            r["frac_less_bracket_overhead"] = round(work / (ms2 * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)
        return r

    # ------------------------------------------------------------------------------------------ kNN all-pairs
    if args.workload == "knn_allpairs":
        from ssl_wafermap_amd import functional as F

        bq = args.batch or 1024
        per = -(-KNN_N // world)
        lo, hi = rank * per, min(KNN_N, (rank + 1) * per)
        g = torch.Generator(device=dev).manual_seed(7 + rank)
        shard = torch.nn.functional.normalize(torch.randn(hi - lo, KNN_D, generator=g, device=dev), dim=1).bfloat16()
        if args.sharded:
            # the memory-scalable form (distributed.sharded_knn_topk, what KNNBenchmarkModule uses when the bank does not
            # fit one GPU): per step one all-gather of the query batches [world * bq, d], a local top-k against the shard
            # with global indices, one all-gather of the [world * bq, k] candidate lists, merge (wm_knn_merge)
            def sstep(i):
                o = (i * bq) % max(hi - lo - bq, 1)
                q = shard[o:o + bq].contiguous()
                if world > 1:
                    qs = [torch.empty_like(q) for _ in range(world)]
                    dist.all_gather(qs, q)
                    q = torch.cat(qs)
                return wdist.sharded_knn_topk(q, shard, KNN_K, lo)

            dt, gpu_ms, (sim, idx) = timed(sstep, args.warmup, args.steps)
            mine = sim[rank * bq:(rank + 1) * bq]
            if not (bool((mine[:, 0] > 0.99).all()) and bool((sim[:, :-1] >= sim[:, 1:]).all())):
                raise SystemExit("knn_allpairs --sharded: self-retrieval / sortedness check failed")
            qps = bq * world * args.steps / dt
            us = gpu_ms * 1e3
            nq = bq * world
            bytes_ = (hi - lo) * KNN_D * 2 + nq * KNN_D * 2 + nq * KNN_K * 8   # per rank and step
            hbm = nq <= 256
            ach = bytes_ / us / 1e3 if hbm else 2.0 * nq * (hi - lo) * KNN_D / us / 1e6
            finish({"metric": "queries/sec (all-pairs cosine kNN top-8, 811457 x 128 bf16)", "value": round(qps, 1),
                    "unit": "queries/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": round(1e3 * dt / args.steps, 4), "gpu_ms_per_step": round(gpu_ms, 4),
                    "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                    "config": {"workload": f"all-pairs kNN top-{KNN_K}, {KNN_N} x {KNN_D} bf16 rows sharded over {world} ranks, bank "
                                           f"kept sharded: per step all-gather of {world} x {bq} queries, local top-k, all-gather "
                                           "of the candidate lists, merge (BASELINE.json configs[4])",
                               "sharded": True, "queries_per_step_per_gpu": bq, "bank_rows_per_gpu": hi - lo,
                               "allpairs_extrapolated_s": round(KNN_N / qps, 3),
                               "curve": "unmeasured beyond the GPUs of this run (the driver's SCALE runs use the default form)"},
                    "roofline": {"bound": "hbm" if hbm else "mfma", "kernel": "knn_stream_b128 + knn_select + knn_merge_lists",
                                 "achieved": round(ach, 1), "peak": HBM_PEAK_GBS if hbm else MFMA_BF16_PEAK_TFLOPS,
                                 "unit": "GB/s" if hbm else "TFLOP/s",
                                 "frac": round(ach / (HBM_PEAK_GBS if hbm else MFMA_BF16_PEAK_TFLOPS), 4), "traffic": None}})
            return
        if world > 1:   # the one exchange of the path: every rank ends up with the whole 208-MB bank
            pad = torch.zeros(per, KNN_D, dtype=torch.bfloat16, device=dev)
            pad[: hi - lo] = shard
            parts = [torch.empty_like(pad) for _ in range(world)]
            t0 = time.perf_counter()
            dist.all_gather(parts, pad)
            torch.cuda.synchronize()
            gather_s = time.perf_counter() - t0
            bank = torch.cat(parts)[:KNN_N].contiguous()
        else:
            gather_s, bank = 0.0, shard
        nq_local = hi - lo

        # A step = `group` query batches issued by ONE C call (functional.knn_topk_batched -> wm_knn_topk_many): consecutive
        # batches alternate between HIP streams, so the latency-bound selection of batch i runs under the streaming kernel
        # of batch i + 1, and no per-batch host work sits between the launches (one Python call per batch made this
        # figure follow the host's speed: 44 us per 64-query step on one box, 87 on another, same kernels).
        n_lanes = int(os.environ.get("WM_KNN_LANES", "3" if bq <= 64 else "2"))
        group = max(1, 1024 // bq)

        def step_join(i):
            o = lo + (i * bq * group) % max(nq_local - bq * group, 1)
            q = bank[o:o + bq * group]
            if args.no_overlap:
                outs = [F.knn_topk(q[j * bq:(j + 1) * bq], bank, KNN_K) for j in range(group)]
                return torch.cat([t[0] for t in outs]), torch.cat([t[1] for t in outs])
            return F.knn_topk_batched(q, bank, KNN_K, batch=bq, lanes=n_lanes)   # (joins its lanes before returning)

        dt, gpu_ms, (sim, idx) = timed(step_join, args.warmup, args.steps)
        ok = bool((sim[:, 0] > 0.99).all()) and bool((sim[:, :-1] >= sim[:, 1:]).all())
        if not ok:
            raise SystemExit("knn_allpairs: self-retrieval / sortedness check failed")
        qps = bq * group * world * args.steps / dt
        us = gpu_ms * 1e3 / group          # per query batch
        bytes_ = KNN_N * KNN_D * 2 + bq * KNN_D * 2 + bq * KNN_K * 8
        finish({"metric": "queries/sec (all-pairs cosine kNN top-8, 811457 x 128 bf16)", "value": round(qps, 1),
                "unit": "queries/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * dt / args.steps, 4), "gpu_ms_per_step": round(gpu_ms, 4), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": f"all-pairs kNN top-{KNN_K}, {KNN_N} x {KNN_D} bf16 rows sharded over {world} ranks, one "
                                       f"all-gather of the shards, {group} batches of {bq} queries per rank per step "
                                       "(BASELINE.json configs[4])",
                           "queries_per_step_per_gpu": bq * group, "batches_per_step": group, "queries_per_batch": bq,
                           "us_per_batch": round(us, 2), "all_gather_s": round(gather_s, 4),
                           "allpairs_extrapolated_s": round(KNN_N / qps, 3)},
                "roofline": {"bound": "hbm" if bq <= 256 else "mfma", "kernel": "knn_stream_b128 + knn_select (wm_knn_topk)",
                             "achieved": round(bytes_ / us / 1e3, 1) if bq <= 256 else round(2.0 * bq * KNN_N * KNN_D / us / 1e6, 1),
                             "peak": HBM_PEAK_GBS if bq <= 256 else MFMA_BF16_PEAK_TFLOPS, "unit": "GB/s" if bq <= 256 else "TFLOP/s",
                             "frac": round((bytes_ / us / 1e3 / HBM_PEAK_GBS) if bq <= 256 else (2.0 * bq * KNN_N * KNN_D / us / 1e6 / MFMA_BF16_PEAK_TFLOPS), 4),
                             "traffic": traffic_entry("knn_b64")[0] if bq == 64 else None}})
        return

    # ------------------------------------------------------------------------------------------ training workloads
    from ssl_wafermap_amd.data import WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.graph import GraphedTrainStep
    from ssl_wafermap_amd.models import SimCLR
    from ssl_wafermap_amd.transforms import BaseViewTransform

    def run_training(workload, B, warmup, steps, roof_steps, keep=False):
        """One training workload: (warm-up, hipGraph capture, W + K replayed steps between fences, `roof_steps` eager
        steps with HIP-event brackets around the MFMA launches) -> (result dict, dataset or None)."""
        simclr = workload == "simclr_r18"
        torch.manual_seed(0)
        if simclr:
            wafers, labels = synthetic_wafers(4096, seed=1234 + rank)
            ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(), device=dev)
            model = SimCLR(None, 9, batch_size=B * world, max_epochs=150, gather_distributed=False)
            gflop, label = R18_GFLOP_PER_SAMPLE, ("SimCLR ResNet-18, 256 wafers/GPU/step, two 3x224x224 views, NT-Xent in-batch "
                                                  "negatives, SGD (BASELINE.json configs[1])")
            FMT, metric = "s2d_bf16", "imgs/sec (SimCLR ResNet18, bs=256, 224^2)"
        else:
            ds, model, gflop, label = make_vit_workload(workload, dev, B, world)
            FMT, metric = "nhwc_bf16", f"imgs/sec ({workload})"
        model = model.to(dev).train()
        (opt,), _ = model.configure_optimizers()
        sync = wdist.GradSync(opt)
        wdist.broadcast_state(model, opt)
        rng = np.random.default_rng(rank)

        def eager_step(i):
            idx = (np.arange(B) + i * B) % len(ds)
            batch = ds.get_batch(idx, rng, fmt=FMT)
            opt.zero_grad()
            loss = model.training_step(batch, i)
            loss.backward()
            sync.start()
            sync.wait()
            opt.step()
            return loss

        graphed = None
        for i in range(min(max(warmup, 1), 3)):
            eager_step(i)
        if not args.no_graph:
            ok = 1
            try:
                stages = bool(getattr(model, "backward_stages", False)) and world > 1 and not args.no_overlap
                graphed = GraphedTrainStep(model, opt, ds, B, fmt=FMT, stages=stages).capture(np.arange(B), rng, sync)
            except Exception as e:  # capture is an optimisation: report and continue eagerly
                print(f"[bench] hipGraph capture failed, running eagerly: {type(e).__name__}: {e}", file=sys.stderr)
                graphed, ok = None, 0
                # a failed capture leaves torch's side stream current (torch.cuda.graph's exit raised before restoring it)
                torch.cuda.set_stream(torch.cuda.default_stream(dev))
            if world > 1:
                # every rank must take the same path: a rank replaying staged graphs and a rank stepping eagerly would
                # issue different collectives
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    graphed = None

        def step(i):
            if graphed is None:
                return eager_step(i)
            return graphed.step((np.arange(B) + i * B) % len(ds), rng, sync)

        dt, gpu_ms, loss = timed(step, warmup, steps)
        final_loss = check_finite(loss)
        rccl_ranks, ar_exposed = 1, 0.0
        if world > 1:
            one = torch.ones(1, device=dev)
            dist.all_reduce(one)                       # every rank of the job took part in a collective
            rccl_ranks = int(one.item())
            # exposed cost of the gradient exchange = the same steps with and without it, measured AFTER the timed region
            # (without the exchange the replicas drift apart: nothing from here on is a training result)
            m = max(5, min(20, steps))

            def step_local(i):
                if graphed is None:
                    idx = (np.arange(B) + i * B) % len(ds)
                    opt.zero_grad()
                    loss_ = model.training_step(ds.get_batch(idx, rng, fmt=FMT), i)
                    loss_.backward()
                    opt.step()
                    return loss_
                return graphed.step((np.arange(B) + i * B) % len(ds), rng, None)

            dt_with, _, _ = timed(step, 0, m)
            dt_without, _, _ = timed(step_local, 0, m)
            ar_exposed = max(0.0, (dt_with - dt_without) / m * 1e3)

        # ---- after the timed region: host-side preparation cost of a step, and eager steps with event brackets
        t0 = time.perf_counter()
        for i in range(20):
            ds.transform.sample(ds.store, (np.arange(B) + i * B) % len(ds), rng)
        host_prepare_ms = (time.perf_counter() - t0) / 20 * 1e3
        roof = None
        if not args.no_kernel_timer and roof_steps > 0:
            kl = ("conv_igemm + conv3x3_patch + conv_wgrad (implicit-GEMM, bf16 MFMA)" if simclr else
                  "Linear GEMMs (conv_igemm 1x1 fwd / dgrad, conv_wgrad) + attn_fwd / attn_bwd, bf16 MFMA")
            tkey = "simclr_r18" if simclr else workload
            timing, roof = None, None
            def external_events_ok():
                # (torch on ROCm 7.0 refuses external events -- "External events are disallowed in rocm" --, so today this
                # is False on the MI355X image and the eager brackets below are what runs; kept for builds that allow them)
                try:
                    torch.cuda.Event(enable_timing=True, external=True).record()
                    return True
                except Exception:
                    return False

            if graphed is not None and os.environ.get("WM_ROOFLINE_GRAPH", "1") != "0" and external_events_ok():
                # (a) the brackets INSIDE a replayed hipGraph: a second capture of the step with the timing events as
                # event-record nodes (external events), replayed `roof_steps` times -- the configuration the timed region
                # ran, no host gaps between the short launches of the transformer steps
                try:
                    timer = ops.KernelTimer(external=True)
                    g2 = GraphedTrainStep(model, opt, ds, B, fmt=FMT, stages=False).capture(np.arange(B), rng, sync, timer=timer)
                    for j in range(roof_steps):
                        check_finite(g2.step((np.arange(B) + (warmup + steps + j) * B) % len(ds), rng, sync))
                        torch.cuda.synchronize()
                        timer.accumulate()
                    roof = roofline_from(timer, roof_steps, kl, tkey)
                    timing = "event-record nodes inside a replayed hipGraph of the step"
                    del g2
                except Exception as e:
                    print(f"[bench] graph-timed roofline unavailable ({type(e).__name__}: {e}); eager brackets instead", file=sys.stderr)
                    ops.TIMER = None
                    torch.cuda.set_stream(torch.cuda.default_stream(dev))
                    roof = None
            if roof is None:
                # (b) eager steps with one bracket per launch; for the transformer steps (~400 launches of 10 - 60 us) the
                # bracket's own cost is calibrated on a dependent chain of two GEMMs of the block's MLP shapes and taken out.
                # The timed region runs independent parts of the step as parallel branches (SimCLR: the two views through
                # the backbone; DINO: the teacher beside the student): a bracket around a launch that shares the chip with
                # another branch's launches measures the contention, not the kernel.  The brackets are therefore taken
                # with the branches switched off -- every launch alone on the device, as in the rocprofv3 tables under
                # profiles/ (same switches) -- and `timing` says so.
                saved_env = {k: os.environ.get(k) for k in ("WM_VIEW_BRANCHES", "WM_DINO_TEACHER_STREAM")}
                os.environ["WM_VIEW_BRANCHES"] = "0"        # (both switches are read per call)
                os.environ["WM_DINO_TEACHER_STREAM"] = "0"
                timer = ops.KernelTimer()
                ops.TIMER = timer
                try:
                    for j in range(roof_steps):
                        check_finite(eager_step(warmup + steps + j))
                finally:
                    ops.TIMER = None
                    for k, v in saved_env.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
                torch.cuda.synchronize()
                over = 0.0
                if not simclr:
                    d_model = model.backbone.embed_dim if hasattr(model.backbone, "embed_dim") else model.backbone.hidden_dim
                    xs = [torch.randn(64 * 197, d_model, device=dev).bfloat16()]
                    w1 = torch.randn(4 * d_model, d_model, device=dev) * 0.02
                    w2 = torch.randn(d_model, 4 * d_model, device=dev) * 0.02
                    bufs = {}

                    def fc1():
                        bufs["h"] = ops.linear(xs[0], w1)

                    def fc2():
                        bufs["y"] = ops.linear(bufs["h"], w2)

                    with torch.no_grad():
                        over = ops.KernelTimer.bracket_overhead_ms([fc1, fc2])
                roof = roofline_from(timer, roof_steps, kl, tkey, overhead_ms=over)
                timing = ("one HIP-event bracket per launch on eager steps, every launch alone on the device (the parallel "
                          "branches of the timed region switched off for these steps)")
            roof["timing"] = timing
        imgs = B * world * steps
        value = imgs / dt
        res = {
            "metric": metric, "value": round(value, 2), "unit": "imgs/sec", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(1e3 * dt / steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": label, "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "hip_graph": graphed is not None,
                       "backward_stage_graphs": len(graphed.graphs) if graphed is not None else 0,
                       # independent parts of the step on parallel branches of the graph (two streams)
                       "parallel_branches": ("two views through the backbone"
                                             if getattr(getattr(model, "backbone", None), "_branches", None) is not None
                                             and os.environ.get("WM_VIEW_BRANCHES", "1") != "0" else
                                             "teacher beside student" if (hasattr(model, "teacher_backbone")
                                             and os.environ.get("WM_DINO_TEACHER_STREAM", "1") != "0") else "none"),
                       "model_tflop_per_step_per_gpu": round(gflop * B / 1e3, 3),
                       "model_mfma_frac": round(value / world * gflop / 1e3 / MFMA_BF16_PEAK_TFLOPS, 4)},
            "final_loss": round(final_loss, 4),
            "rccl_ranks": rccl_ranks,                    # result of an all-reduce of ones over the job's process group
            "all_reduce_ms_exposed": round(ar_exposed, 3),  # step time with minus without the gradient exchange (after the timed region)
            "gpu_ms_per_step": round(gpu_ms, 3),          # HIP events around the K steps on the launch stream
            "host_prepare_ms_per_step": round(host_prepare_ms, 3),  # drawing + checking the step's augmentation decisions
            "roofline": roof,
        }
        del graphed
        model = opt = sync = None
        torch.cuda.empty_cache()
        return res, (ds if keep else None)

    simclr = args.workload == "simclr_r18"
    B = args.batch or (256 if simclr else 64)
    out, ds = run_training(args.workload, B, args.warmup, args.steps, 0 if args.no_kernel_timer else 3, keep=simclr)
    if world == 1 and simclr and not args.no_secondary:
        out["augment"] = augment_object(dev, ds, B)
        del ds
        torch.cuda.empty_cache()
        # the transformer workload of the north-star (BASELINE configs[2]) inside the default line: DINO ViT-Tiny/16,
        # 64 wafers, a short replayed run + the GEMM / attention roofline of eager steps
        vit, _ = run_training("dino_vit_tiny", 64, 3, 10, 2)
        out["vit"] = {"workload": vit["config"]["workload"], "imgs_per_sec": vit["value"], "ms_per_step": vit["ms_per_step"],
                      "steps": vit["steps"], "hip_graph": vit["config"]["hip_graph"], "final_loss": vit["final_loss"],
                      "model_mfma_frac": vit["config"]["model_mfma_frac"], "roofline": vit["roofline"]}
        out["knn"] = knn_object(dev)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
        if simclr and not args.no_secondary:
            out["knn"]["cpu_baseline"] = cpu_knn_baseline()
    finish(out)


if __name__ == "__main__":
    main()
