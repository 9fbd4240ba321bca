"""Step time of the non-headline models (DINO ViT-S/16, MAE ViT-B/32) at the reference batch size,
synthetic wafers, fused augmentation -> forward/backward -> AdamW, one GPU.

    python tools/bench_models.py --model dino --batch 256 --steps 10 --warmup 3
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ssl_wafermap_amd.data import WaferMapDataset  # noqa: E402
from ssl_wafermap_amd.data.synthetic import synthetic_wafers  # noqa: E402
from ssl_wafermap_amd.models import MAE, DINOViT  # noqa: E402
from ssl_wafermap_amd.transforms import BaseViewTransform, MultiCropTransform  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["dino", "mae"], default="dino")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    wafers, labels = synthetic_wafers(2048, seed=1)
    if a.model == "dino":
        tf = MultiCropTransform()
        model = DINOViT(None, 9, batch_size=a.batch, log_rep_std=False)
        # per sample: ViT-S/16 4.6 GFLOP fwd at 197 tokens, ~0.8 at 37; teacher 2 fwd, student (2+6) fwd+bwd
        gflop = 2 * 4.6 + 3 * (2 * 4.6 + 6 * 0.8)
    else:
        tf = BaseViewTransform(n_views=1) if "n_views" in BaseViewTransform.__init__.__code__.co_varnames else BaseViewTransform()
        model = MAE(None, 9, batch_size=a.batch, log_rep_std=False)
        gflop = 3 * (0.17 * 12 / 12 * 8.7 / 4 + 0.35)  # rough: encoder on 12 of 50 tokens + decoder on 50
    ds = WaferMapDataset(wafers, labels, transform=tf, device=dev)
    model = model.to(dev).train()
    (opt,), _ = model.configure_optimizers()
    rng = np.random.default_rng(0)

    def step(i):
        idx = (np.arange(a.batch) + i * a.batch) % len(ds)
        batch = ds.get_batch(idx, rng)
        if a.model == "mae":
            batch = (batch[0][:1], batch[1])
        opt.zero_grad()
        loss = model.training_step(batch, i)
        loss.backward()
        opt.step()
        return loss

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"model": a.model, "batch": a.batch, "ms_per_step": round(dt * 1e3, 2),
                      "imgs_per_sec": round(a.batch / dt, 1), "approx_TFLOPs": round(gflop * a.batch / dt / 1e3, 1),
                      "loss": round(float(loss), 4), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))


if __name__ == "__main__":
    main()
