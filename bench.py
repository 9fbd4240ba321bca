#!/usr/bin/env python3
"""Headline benchmark: SimCLR ResNet-18 training throughput on synthetic wafer maps (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch: fused two-view augmentation of 256 wafers per
GPU (written in the stem's space-to-depth layout) -> ResNet-18 forward/backward on 2 x 256 images
(3 x 224 x 224, bf16, BN statistics per view) ->
SimCLR projection head -> NT-Xent (in-batch negatives) -> (N > 1: flat RCCL all-reduce of the
gradient arena) -> fused SGD.  All device work runs in the hand-written HIP kernels of
libwafer_hip.so; inputs (the ragged uint8 wafer store) are resident in HBM before the timed region.

Prints ONE JSON line (rank 0): metric imgs/sec = wafers (not views) per second, whole job.
  roofline     : the conv implicit-GEMM kernels (fwd + dgrad + wgrad), algorithmic FLOPs / the summed
                 HIP-event durations of those launches, vs dense bf16 MFMA peak.  The events are
                 recorded INSIDE the timed region on the launch stream, on the LAST timed step only
                 (or every `--timer-every`-th): a timing event is a barrier packet on ROCm and
                 bracketing all ~60 conv launches of every step costs ~20 % throughput, so the
                 bracketed step runs eagerly (about twice a graph replay's time: with the default 100
                 timed steps it costs ~1 % of `value`) and the others replay the captured hipGraph.
  cpu_baseline : the torch-CPU oracle (oracle/) running BASELINE configs[0] (bs 32, fp32) on the
                 host cores for a bounded number of steps (rank 0, N = 1 only).
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
# HBM-side bytes per conv launch (mean over the 65 implicit-GEMM launches of a step), measured with
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of this same script, FETCH_SIZE doubled per the
# gfx950 correction (calibrated on sgd_step: 138.1 / 92.0 MB measured vs 138.0 / 92.0 MB algorithmic):
# profiles/r01_hbm_traffic_simclr_r18_v2.md.  A PMC pass cannot run inside the timed region, so this is the
# committed measurement, not a live one.  Algorithmic bytes (every operand once): 230 MB per launch.
# (The counter sits behind L2: repeats served by the 256-MB Infinity Cache are included.)
CONV_TRAFFIC_BYTES_PER_LAUNCH = 518.0e6
R18_GFLOP_PER_SAMPLE = 21.76    # ResNet-18 fwd 3.627 GFLOP x 3 (fwd+bwd) x 2 views


def cpu_baseline(seconds_budget: float = 25.0):
    """Oracle SimCLR step (torch CPU float32), BASELINE configs[0]: bs 32, two views, SGD."""
    import numpy as np
    import torch

    from oracle import augment as oa
    from oracle import resnet as orn
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.heads import SimCLRProjectionHead
    from ssl_wafermap_amd.models import create_model

    torch.manual_seed(0)
    cores = torch.get_num_threads()
    wafers, _ = synthetic_wafers(64, seed=1234)
    backbone, head = create_model("resnet18", num_classes=0), SimCLRProjectionHead(512, 512, 128)
    sd = {"backbone." + k: v.clone() for k, v in backbone.state_dict().items()}
    sd.update({"projection_head." + k: v.clone() for k, v in head.state_dict().items()})
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and "running" not in k}
    state = dict(sd)
    state.update(params)
    bufs = {}
    rng = np.random.default_rng(0)
    bs = 32

    def step(i):
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
        for p in params.values():
            p.grad = None
        loss, _ = orn.simclr_loss(views[0], views[1], state, 0.5, True)
        loss.backward()
        with torch.no_grad():
            orn.sgd_step({k: p for k, p in params.items()}, {k: p.grad for k, p in params.items()}, bufs,
                         lr=6e-2 * bs / 256)
        return float(loss)

    step(0)  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        step(n + 1)
        n += 1
        if time.perf_counter() - t0 > seconds_budget or n >= 8:
            break
    dt = time.perf_counter() - t0
    return {"value": round(bs * n / dt, 3), "unit": "imgs/sec", "cores": cores, "kind": "port",
            "sample": f"{n} SimCLR ResNet-18 steps at bs {bs} (fp32, torch CPU oracle incl. numpy augmentation), "
                      f"{dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="wafers per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--timer-every", type=int, default=0,
                    help="bracket the conv launches on every n-th timed step (0: on the last timed step only)")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step into a hipGraph")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from ssl_wafermap_amd import distributed as wdist
    from ssl_wafermap_amd import ops
    from ssl_wafermap_amd.data import WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.models import SimCLR
    from ssl_wafermap_amd.transforms import BaseViewTransform

    rank, world, local = wdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if os.environ.get("WM_SINGLE_DEVICE") == "1":
        local = 0  # rehearsal of the N > 1 path on a one-GPU box (with WM_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    B = args.batch
    wafers, labels = synthetic_wafers(4096, seed=1234 + rank)
    ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(), device=dev)
    torch.manual_seed(0)
    model = SimCLR(None, 9, batch_size=B * world, max_epochs=150, gather_distributed=False).to(dev).train()
    (opt,), _ = model.configure_optimizers()
    sync = wdist.GradSync(opt)
    rng = np.random.default_rng(rank)
    # the augmentation kernel writes the 2x2 space-to-depth layout the stem convolution consumes (no layout pass)
    FMT = "s2d_bf16"

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

    def step(i):
        # sampled steps (kernel timer on) run eagerly: timing events cannot be captured
        if graphed is None or ops.TIMER is not None:
            return eager_step(i)
        return graphed.step((np.arange(B) + i * B) % len(ds), rng, sync)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    if not args.no_graph:
        try:
            from ssl_wafermap_amd.graph import GraphedTrainStep

            graphed = GraphedTrainStep(model, opt, ds, B, fmt=FMT).capture(np.arange(B), rng, sync)
            for i in range(2):
                step(i)
        except Exception as e:  # capture is an optimisation: report and continue eagerly
            print(f"[bench] hipGraph capture failed, running eagerly: {type(e).__name__}: {e}", file=sys.stderr)
            graphed = None
    timer = None if args.no_kernel_timer else ops.KernelTimer()
    timed_steps = 0
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if args.timer_every > 0:
            sample = timer is not None and i % args.timer_every == args.timer_every - 1
        else:
            sample = timer is not None and i == args.steps - 1
        ops.TIMER = timer if sample else None
        timed_steps += int(sample)
        loss = step(args.warmup + i)
    ops.TIMER = None
    host_dt = time.perf_counter() - t0  # time the host needed to enqueue the K steps (no sync yet)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    if not np.isfinite(final_loss):
        # a throughput figure of a network that has diverged is not a training throughput
        print(f"[bench] rank {rank}: non-finite loss {final_loss} after {args.warmup} + {args.steps} steps", file=sys.stderr)
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(3)

    if rank == 0:
        imgs = B * world * args.steps
        value = imgs / dt
        roof = None
        if timer is not None:
            summ = timer.summary()
            work = sum(v["work"] for v in summ.values())
            ms = sum(v["ms"] for v in summ.values())
            ach = work / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            roof = {"bound": "mfma", "kernel": "conv_igemm + conv3x3_patch + conv_wgrad (implicit-GEMM, bf16 MFMA)",
                    "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": CONV_TRAFFIC_BYTES_PER_LAUNCH,
                    "traffic_source": "profiles/r01_hbm_traffic_simclr_r18_v2.md (rocprofv3 --pmc, separate passes)",
                    "launches": int(sum(v["launches"] for v in summ.values())),
                    "sampled_steps": timed_steps,
                    "kernel_ms_per_step": round(ms / max(timed_steps, 1), 3),
                    "by_kernel": {k: {"TFLOP/s": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 2),
                                      "ms_per_step": round(v["ms"] / max(timed_steps, 1), 3)} for k, v in summ.items()}}
        out = {
            "metric": "imgs/sec (SimCLR ResNet18, bs=256, 224^2)",
            "value": round(value, 2), "unit": "imgs/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "SimCLR ResNet-18, 256 wafers/GPU/step, two 3x224x224 views, NT-Xent in-batch "
                                   "negatives, SGD (BASELINE.json configs[1])",
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}",
                       "hip_graph": graphed is not None,
                       "model_tflop_per_step_per_gpu": round(R18_GFLOP_PER_SAMPLE * B / 1e3, 3),
                       "model_mfma_frac": round(value / world * R18_GFLOP_PER_SAMPLE / 1e3 / MFMA_BF16_PEAK_TFLOPS, 4)},
            "final_loss": round(final_loss, 4),
            "host_enqueue_ms_per_step": round(1e3 * host_dt / args.steps, 3),
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
