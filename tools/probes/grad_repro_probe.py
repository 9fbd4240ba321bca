#!/usr/bin/env python3
"""Run-to-run reproducibility of ONE SimCLR step's gradients (same weights, same decisions): eager vs eager,
graph replay vs graph replay, eager vs graph; per-parameter relative L2 differences (largest first)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ssl_wafermap_amd.data import WaferMapDataset  # noqa: E402
from ssl_wafermap_amd.data.synthetic import synthetic_wafers  # noqa: E402
from ssl_wafermap_amd.graph import GraphedTrainStep  # noqa: E402
from ssl_wafermap_amd.models import SimCLR  # noqa: E402
from ssl_wafermap_amd.transforms import BaseViewTransform  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wafers, labels = synthetic_wafers(256, seed=11)


def run(graph):
    ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(), device="cuda:0")
    torch.manual_seed(5)
    model = SimCLR(None, 9, batch_size=2 * B, max_epochs=10, log_rep_std=False).to("cuda:0").train()
    (opt,), _ = model.configure_optimizers()
    names = [n for n, p in model.named_parameters()]
    p1 = ds.transform.sample(ds.store, np.arange(B), np.random.default_rng(1))
    p2 = ds.transform.sample(ds.store, np.arange(B) + B, np.random.default_rng(2))
    if graph:
        g = GraphedTrainStep(model, opt, ds, B, warmup=1, fmt="s2d_bf16")
        g.capture(np.arange(B), np.random.default_rng(1))
        g._upload(p2)
        g.graph.replay()
        loss = g.loss
    else:
        opt.zero_grad()
        model.training_step((ds.transform.launch(ds.store, p1, B, "s2d_bf16"), None), 0).backward()
        opt.step()
        opt.zero_grad()
        loss = model.training_step((ds.transform.launch(ds.store, p2, B, "s2d_bf16"), None), 0)
        loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters()}


def cmp(a, b, label):
    (la, ga), (lb, gb) = a, b
    tot = float(torch.cat([(ga[k] - gb[k]).flatten() for k in ga]).norm() / torch.cat([gb[k].flatten() for k in gb]).norm())
    rows = sorted(((float((ga[k] - gb[k]).norm() / (gb[k].norm() + 1e-30)), k) for k in ga), reverse=True)
    print(f"{label}: loss {la:.6f} vs {lb:.6f}; all gradients rel L2 diff {tot:.3e}; worst: "
          + ", ".join(f"{k} {v:.2e}" for v, k in rows[:4]))


e1, e2, g1, g2 = run(False), run(False), run(True), run(True)
cmp(e1, e2, "eager vs eager")
cmp(g1, g2, "graph vs graph")
cmp(g1, e1, "graph vs eager")
e3 = run(False)
cmp(e3, e1, "eager (after graphs) vs eager")
