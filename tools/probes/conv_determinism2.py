"""conv3x3_patch race hunt, part 2: holes (sentinel survives) or wrong values?  Where?  From which batch size on?"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ssl_wafermap_amd import _lib  # noqa: E402
from ssl_wafermap_amd._lib import check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
H, C, K = 56, 64, 64
for N in (64, 128, 256, 512):
    x = torch.randn(N, H, H, C, generator=g, device=dev).bfloat16()
    wk = (torch.randn(K, 3, 3, C, generator=g, device=dev) * 0.05).bfloat16()
    geom = (N, H, H, C, K, 3, 3, H, H, 1, 1)
    outs = []
    for rep in range(5):
        y = torch.full((N, H, H, K), float("nan"), device=dev, dtype=torch.bfloat16)
        check(lib.wm_conv2d_fwd(ptr(x), ptr(wk), ptr(y), *geom, st), "f")
        torch.cuda.synchronize()
        outs.append(y)
    nan = [int(torch.isnan(o).sum()) for o in outs]
    # majority vote as the reference
    ref = torch.stack([o.float() for o in outs]).median(0).values
    bad = [(o.float() != ref) for o in outs]
    nbad = [int(b.sum()) for b in bad]
    line = f"N={N}: NaN holes {nan}; elements off the median {nbad}"
    b = bad[0] | bad[1] | bad[2] | bad[3] | bad[4]
    if int(b.sum()):
        idx = b.nonzero()
        n_, h_, w_, k_ = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
        tile = (n_ * 49 + (h_ // 8) * 7 + (w_ // 8))
        blocks = torch.unique(tile // 2)
        line += (f"; distinct blocks touched {blocks.numel()}; first blocks {blocks[:12].tolist()}; "
                 f"k range {int(k_.min())}-{int(k_.max())}; in-tile rows {torch.unique(h_ % 8).tolist()} cols {torch.unique(w_ % 8).tolist()}")
        d = (outs[0].float() - ref)[bad[0]]
        if d.numel():
            line += f"; |diff| max {float(d.abs().max()):.3e} vs |ref| max {float(ref.abs().max()):.3e}"
    print(line, flush=True)
