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

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(2):
        step(3 + i)
    torch.cuda.synchronize()
agg = collections.Counter()
LAUNCHERS = ("aten::copy_", "aten::fill_", "aten::add_", "aten::add", "aten::mul", "aten::mul_", "aten::cat", "aten::index",
             "aten::_to_copy", "aten::zero_", "aten::div", "aten::div_", "aten::sum", "aten::mean", "aten::clone",
             "aten::index_select", "aten::gather", "aten::scatter_", "aten::sub", "aten::neg", "aten::where")
for ev in prof.events():
    if ev.name not in LAUNCHERS and "Memcpy" not in ev.name:
        continue
    # a kernel-launching ATen op: who called it?  (the chain of enclosing profiler ranges: autograd nodes, custom Functions)
    chain, par = [], ev.cpu_parent
    while par is not None and len(chain) < 6:
        chain.append(par.name)
        par = par.cpu_parent
    shape = ""
    try:
        shape = str(ev.input_shapes)[:60]
    except Exception:
        pass
    agg[(ev.name, " < ".join(chain)[:150], shape)] += 1
for (n, where, shape), c in sorted(agg.items(), key=lambda t: -t[1])[:80]:
    print(f"{c / 2:6.1f}/step  {n:16s} {shape:60s} {where}")
