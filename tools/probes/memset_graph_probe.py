#!/usr/bin/env python3
"""Does a hipMemsetAsync captured into a hipGraph clear its target on every replay?
wm_ntxent_bwd(split > 1) = memset(dzn) + a kernel that adds partial row gradients with f32 atomics.
The output buffer is poisoned before each replay; a replay that does not clear it leaves NaN behind."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ssl_wafermap_amd import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
b, d = 256, 128
zn = torch.nn.functional.normalize(torch.randn(2 * b, d, device=dev), dim=1)
lse, rows = F.ntxent_forward(zn, zn, b, b, 0, 0.5)
ref = F.ntxent_backward(zn, zn, lse, b, b, 0, 0.5, 1.0 / (2 * b)).clone()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        F.ntxent_backward(zn, zn, lse, b, b, 0, 0.5, 1.0 / (2 * b))
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = F.ntxent_backward(zn, zn, lse, b, b, 0, 0.5, 1.0 / (2 * b))
bad = 0
for i in range(50):
    out.fill_(float("nan") if i % 2 else 1000.0)
    g.replay()
    torch.cuda.synchronize()
    err = (out - ref).abs().max().item()
    if not (err < 1e-6):
        bad += 1
        if bad < 5:
            print(f"replay {i}: max|out - ref| = {err}")
print(f"{bad} of 50 replays left the poisoned output in place" if bad else "all 50 replays cleared the output")
