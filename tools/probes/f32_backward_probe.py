"""Per-tensor gradient error of the float32 preset's SimCLR step against the oracle (debugging aid)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle import resnet as orn
from ssl_wafermap_amd import precision
from ssl_wafermap_amd.data import WaferMapDataset
from ssl_wafermap_amd.data.synthetic import synthetic_wafers
from ssl_wafermap_amd.models import SimCLR
from ssl_wafermap_amd.transforms import BaseViewTransform, augment_views

DEV = "cuda:0"
B = 32
wafers, labels = synthetic_wafers(128, seed=7)
ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(), device=DEV)
torch.manual_seed(0)
model = SimCLR(None, 9, batch_size=B, max_epochs=150, log_rep_std=False).to(DEV).train()
(opt,), _ = model.configure_optimizers()
sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
names = [k for k, _ in model.named_parameters()]
for k in names:
    sd[k].requires_grad_(True)
rng = np.random.default_rng(5)
params = ds.transform.sample(ds.store, np.arange(B), rng)
v = augment_views(ds.store, params[0], fmt="nchw_f32", n_slots=2 * B)
opt.zero_grad()
with precision.precision("float32"):
    loss = model.training_step(((v[:B], v[B:]), None), 0)
    loss.backward()
vc = v.cpu()
ref, _ = orn.simclr_loss(vc[:B], vc[B:], sd, 0.5, True)
ref.backward()
print("loss", float(loss), float(ref))
for k, p in model.named_parameters():
    g, r = p.grad.float().cpu(), sd[k].grad
    print(f"{float((g - r).norm() / r.norm().clamp_min(1e-20)):.3e}  |ref| {float(r.norm()):.3e}  |got| {float(g.norm()):.3e}  {k} {tuple(p.shape)}")

# ---- three optimiser steps: per-step loss error and the worst parameter difference after each update
lr = opt.param_groups[0]["lr"]
bufs = {}
opt.step()
with torch.no_grad():
    orn.sgd_step({k: sd[k] for k in names}, {k: sd[k].grad for k in names}, bufs, lr=lr)
for i in range(1, 4):
    worst = max(((float((p.detach().float().cpu() - sd[k].detach()).abs().max() / sd[k].detach().abs().max().clamp_min(1e-12)), k)
                 for k, p in model.named_parameters()))
    print(f"after update {i}: worst parameter difference (relative max) {worst[0]:.3e} at {worst[1]}")
    idx = (np.arange(B) + i * B) % len(ds)
    params = ds.transform.sample(ds.store, idx, rng)
    v = augment_views(ds.store, params[0], fmt="nchw_f32", n_slots=2 * B)
    opt.zero_grad()
    with precision.precision("float32"):
        loss = model.training_step(((v[:B], v[B:]), None), i)
        loss.backward()
    vc = v.cpu()
    for k in names:
        sd[k].grad = None
    ref, _ = orn.simclr_loss(vc[:B], vc[B:], sd, 0.5, True)
    ref.backward()
    print(f"step {i + 1}: loss {float(loss.detach()):.7f} oracle {float(ref.detach()):.7f} rel {abs(float(loss.detach()) - float(ref.detach())) / float(ref.detach()):.2e}")
    gw = max(((float((p.grad.float().cpu() - sd[k].grad).norm() / sd[k].grad.norm().clamp_min(1e-20)), k) for k, p in model.named_parameters()))
    print(f"   worst gradient (relative L2) {gw[0]:.3e} at {gw[1]}")
    if i == 1:
        for k, p in model.named_parameters():
            gg, r = p.grad.float().cpu(), sd[k].grad
            print(f"      {float((gg - r).norm() / r.norm().clamp_min(1e-20)):.3e}  |ref| {float(r.norm()):.3e}  {k}")
    opt.step()
    with torch.no_grad():
        orn.sgd_step({k: sd[k] for k in names}, {k: sd[k].grad for k in names}, bufs, lr=lr)
