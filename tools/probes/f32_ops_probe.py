"""float32-preset ops with backward vs torch CPU autograd (debugging aid)."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from ssl_wafermap_amd import f32path

DEV = "cuda:0"
g = torch.Generator().manual_seed(0)


def rel(a, b):
    return float((a.float().cpu() - b).norm() / b.norm().clamp_min(1e-20))


for (n, c, h, k, r, st, pad) in [(3, 16, 12, 32, 3, 1, 1), (3, 16, 12, 32, 3, 2, 1), (2, 64, 14, 70, 1, 2, 0), (2, 3, 20, 64, 7, 2, 3), (4, 64, 8, 64, 3, 1, 1)]:
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(k, c, r, r, generator=g) * 0.1
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, st, pad)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    yd = f32path.conv2d(xd, wd, st, pad)
    yd.backward(dy.to(DEV))
    print(f"conv {c}->{k} {r}x{r}/{st}: y {rel(yd.detach(), yr.detach()):.2e} dx {rel(xd.grad, xr.grad):.2e} dw {rel(wd.grad, wr.grad):.2e}")

for relu, res in ((True, False), (True, True), (False, False)):
    x = torch.randn(8, 32, 6, 6, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.1
    r_ = torch.randn(8, 32, 6, 6, generator=g)
    xr, gr, br, rr = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), r_.clone().requires_grad_(True)
    parts = []
    for xp, rp in zip(xr.chunk(2), rr.chunk(2)):
        o = F.batch_norm(xp, None, None, gr, br, True, 0.1, 1e-5)
        if res:
            o = o + rp
        parts.append(F.relu(o) if relu else o)
    yr = torch.cat(parts)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xd, gd, bd, rd = x.to(DEV).requires_grad_(True), gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True), r_.to(DEV).requires_grad_(True)
    rm, rv, nb = torch.zeros(32, device=DEV), torch.ones(32, device=DEV), torch.tensor(0, device=DEV)
    yd = f32path.batch_norm(xd, gd, bd, rm, rv, True, rd if res else None, relu, 1e-5, 0.1, 2, nb)
    yd.backward(dy.to(DEV))
    print(f"bn relu={relu} res={res}: y {rel(yd.detach(), yr.detach()):.2e} dx {rel(xd.grad, xr.grad):.2e} dgamma {rel(gd.grad, gr.grad):.2e} "
          f"dbeta {rel(bd.grad, br.grad):.2e}" + (f" dres {rel(rd.grad, rr.grad):.2e}" if res else ""))

x = torch.randn(3, 8, 9, 9, generator=g)
xr = x.clone().requires_grad_(True)
yr = F.max_pool2d(xr, 3, 2, 1)
dy = torch.randn(yr.shape, generator=g)
yr.backward(dy)
xd = x.to(DEV).requires_grad_(True)
yd = f32path.max_pool3x3s2(xd)
yd.backward(dy.to(DEV))
print(f"maxpool: y {rel(yd.detach(), yr.detach()):.2e} dx {rel(xd.grad, xr.grad):.2e}")
xr = x.clone().requires_grad_(True)
yr = xr.mean((2, 3))
dy = torch.randn(yr.shape, generator=g)
yr.backward(dy)
xd = x.to(DEV).requires_grad_(True)
yd = f32path.global_avg_pool(xd)
yd.backward(dy.to(DEV))
print(f"gap: y {rel(yd.detach(), yr.detach()):.2e} dx {rel(xd.grad, xr.grad):.2e}")
x = torch.randn(40, 48, generator=g)
w = torch.randn(24, 48, generator=g) * 0.1
xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
yr = F.linear(xr, wr)
dy = torch.randn(yr.shape, generator=g)
yr.backward(dy)
xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
yd = f32path.linear(xd, wd)
yd.backward(dy.to(DEV))
print(f"linear: y {rel(yd.detach(), yr.detach()):.2e} dx {rel(xd.grad, xr.grad):.2e} dw {rel(wd.grad, wr.grad):.2e}")
