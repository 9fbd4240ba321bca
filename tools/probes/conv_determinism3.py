"""conv3x3_patch race hunt, part 3: inside the failing blocks, which (pixel, channel) are wrong?"""
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
H, C, K, N = 56, 64, 64, 256
x = torch.randn(N, H, H, C, generator=g, device=dev).bfloat16()
wk = (torch.randn(K, 3, 3, C, generator=g, device=dev) * 0.05).bfloat16()
geom = (N, H, H, C, K, 3, 3, H, H, 1, 1)
outs = []
for rep in range(5):
    y = torch.empty((N, H, H, K), device=dev, dtype=torch.bfloat16)
    check(lib.wm_conv2d_fwd(ptr(x), ptr(wk), ptr(y), *geom, st), "f")
    torch.cuda.synchronize()
    outs.append(y.float())
ref = torch.stack(outs).median(0).values
shown = 0
for r, o in enumerate(outs):
    bad = (o != ref)
    if not int(bad.sum()):
        continue
    idx = bad.nonzero()
    tile = idx[:, 0] * 49 + (idx[:, 1] // 8) * 7 + (idx[:, 2] // 8)
    for blk in torch.unique(tile // 2)[:4].tolist():
        m = (tile // 2) == blk
        sub = idx[m]
        t_in = (tile[m] % 2)
        pix = (sub[:, 1] % 8) * 8 + (sub[:, 2] % 8)
        ks = sub[:, 3]
        desc = []
        for t in (0, 1):
            mm = t_in == t
            if int(mm.sum()):
                pk = {}
                for p, k in zip(pix[mm].tolist(), ks[mm].tolist()):
                    pk.setdefault(p, []).append(k)
                desc.append(f"tile{t}: {len(pk)} pixels; channels per pixel "
                            + ", ".join(f"p{p}(r{p // 8}c{p % 8}):{min(v)}-{max(v)}#{len(v)}" for p, v in sorted(pk.items())[:10]))
        print(f"rep {r} block {blk}: {int(m.sum())} wrong; " + " | ".join(desc), flush=True)
        shown += 1
    if shown >= 10:
        break
