"""Four-stage ring of conv_igemm (small grids, long reductions) against the two-stage loop: bit-equality and time."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from ssl_wafermap_amd import ops, vit_ops

DEV = "cuda:0"
torch.manual_seed(0)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rows, c, k in [(512, 2048, 2048), (512, 2048, 256), (512, 256, 2048), (128, 512, 2048), (512, 512, 512), (256, 512, 128),
                   (1024, 2048, 2048), (512, 2048, 192), (4096, 1024, 512)]:
    x = (torch.randn(rows, c, device=DEV) * 0.5).bfloat16().requires_grad_(True)
    w = torch.nn.Parameter(torch.randn(k, c, device=DEV) * 0.03)
    b = torch.nn.Parameter(torch.randn(k, device=DEV) * 0.1)
    dy = (torch.randn(rows, k, device=DEV) * 0.1).bfloat16()
    res = {}
    for deep in ("0", "1"):
        os.environ["WM_CONV_DEEP"] = deep
        y0 = ops.linear(x, w)
        y1 = vit_ops.linear(x, w, b)
        (dx,) = torch.autograd.grad(ops.linear(x, w), x, dy)
        t_f = timed(lambda: ops.linear(x.detach(), w))
        t_b = timed(lambda: vit_ops.linear(x.detach(), w, b))
        xg = x.detach().requires_grad_(True)
        yy = ops.linear(xg, w)
        t_d = timed(lambda: torch.autograd.grad(yy, xg, dy, retain_graph=True))
        res[deep] = (y0.detach().clone(), y1.detach().clone(), dx.clone(), t_f, t_b, t_d)
    a, d = res["0"], res["1"]
    same = all(torch.equal(a[i], d[i]) for i in range(3))
    ref = (x.detach().float() @ w.detach().float().t())
    err = float((d[0].float() - ref).abs().max() / ref.abs().max())
    print(f"rows {rows:5d} C {c:5d} K {k:5d}: bit-identical {same}  rel err vs f32 {err:.2e}  "
          f"fwd {a[3]:.1f} -> {d[3]:.1f} us, fwd+bias {a[4]:.1f} -> {d[4]:.1f} us, dgrad(+wgrad) {a[5]:.1f} -> {d[5]:.1f} us")
