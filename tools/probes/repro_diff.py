"""Which gradients differ between two runs of the same SimCLR step (bit-reproducibility hunt)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ssl_wafermap_amd.data import WaferMapDataset  # noqa: E402
from ssl_wafermap_amd.data.synthetic import synthetic_wafers  # noqa: E402
from ssl_wafermap_amd.models import SimCLR  # noqa: E402
from ssl_wafermap_amd.transforms import BaseViewTransform  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def run():
    wafers, labels = synthetic_wafers(1024, seed=1234)
    ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(), device="cuda:0")
    torch.manual_seed(0)
    model = SimCLR(None, 9, batch_size=B, max_epochs=150, log_rep_std=False).to("cuda:0").train()
    (opt,), _ = model.configure_optimizers()
    rng = np.random.default_rng(7)
    batch = ds.get_batch(np.arange(B), rng, fmt="s2d_bf16")
    opt.zero_grad()
    acts = {}

    def hook(name):
        def f(mod, inp, out):
            o = out[0] if isinstance(out, tuple) else out
            acts[name] = o.detach().float().clone()
            if o.requires_grad:
                o.register_hook(lambda g, n=name: acts.__setitem__("grad:" + n, g.detach().float().clone()))
        return f

    for n, m in model.named_modules():
        if n and n.count(".") <= 2:
            m.register_forward_hook(hook(n))
    loss = model.training_step(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()}, acts


la, ga, aa = run()
lb, gb, ab = run()
print("loss", la, lb, la == lb)
for n in ga:
    if not torch.equal(ga[n], gb[n]):
        d = (ga[n] - gb[n]).abs().max().item()
        print(f"param grad differs: {n:50s} max abs {d:.3e} of {ga[n].abs().max().item():.3e}")
for n in aa:
    if n in ab and not torch.equal(aa[n], ab[n]):
        print(f"activation/grad differs: {n:50s} max abs {(aa[n] - ab[n]).abs().max().item():.3e}")
print("done")
