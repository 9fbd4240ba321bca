"""Which Python lines launch the remaining non-library GPU work of a training step (device-to-device copies, ATen
fill / add / cat kernels)?  torch.profiler with stacks over two eager steps of a bench workload.
    python tools/probes/find_aten_kernels.py dino_vit_tiny 64"""
import collections
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402

workload, B = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
torch.manual_seed(0)
if workload == "simclr_r18":
    from ssl_wafermap_amd.data import WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.models import SimCLR
    from ssl_wafermap_amd.transforms import BaseViewTransform

    wafers, labels = synthetic_wafers(2048, seed=1)
    ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(), device=dev)
    model, fmt = SimCLR(None, 9, batch_size=B, max_epochs=150), "s2d_bf16"
else:
    ds, model, _, _ = bench.make_vit_workload(workload, dev, B, 1)
    fmt = "nhwc_bf16"
model = model.to(dev).train()
(opt,), _ = model.configure_optimizers()
rng = np.random.default_rng(0)


def step(i):
    batch = ds.get_batch((np.arange(B) + i * B) % len(ds), rng, fmt=fmt)
    opt.zero_grad()
    loss = model.training_step(batch, i)
    loss.backward()
    opt.step()
    if hasattr(model, "on_train_batch_end"):
        model.on_train_batch_end()
    return loss


for i in range(3):
    step(i)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(2):
        step(3 + i)
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    n = ev.name
    if not (n.startswith("aten::") or "Memcpy" in n or "Memset" in n):
        continue
    if n in ("aten::empty", "aten::empty_like", "aten::view", "aten::reshape", "aten::as_strided", "aten::detach", "aten::slice",
             "aten::select", "aten::empty_strided", "aten::_unsafe_view", "aten::alias", "aten::t", "aten::transpose",
             "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::narrow", "aten::permute", "aten::result_type",
             "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::detach_", "aten::to",
             "aten::contiguous", "aten::view_as", "aten::unbind", "aten::split", "aten::chunk", "aten::flatten",
             "aten::resolve_conj", "aten::resolve_neg", "aten::set_", "aten::stride", "aten::size"):
        continue
    frames = [f for f in (ev.stack or []) if "ssl_wafermap_amd" in f or "self-supervised" in f or "bench" in f]
    where = frames[0].strip()[-110:] if frames else "(autograd engine / no python frame)"
    agg[(n, where)] += 1
for (n, where), c in sorted(agg.items(), key=lambda t: -t[1])[:70]:
    print(f"{c / 2:7.1f}/step  {n:32s} {where}")
