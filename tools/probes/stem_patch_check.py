"""conv_stem_patch against conv_igemm<128,64,2,0> (WM_CONV_PATCH=0) on the same stem inputs: run twice with the env
switch and compare the saved outputs (bit-exact expected: same MFMA sequence)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ssl_wafermap_amd import _lib  # noqa: E402
from ssl_wafermap_amd._lib import check, ptr  # noqa: E402

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
out = {}
for (N, H) in ((6, 112), (2, 48), (3, 16)):
    x = torch.randn(N, H, H, 16, generator=g, device="cuda").bfloat16()
    x[..., 12:] = 0
    w = (torch.randn(64, 4, 4, 16, generator=g, device="cuda") * 0.05).bfloat16()
    y = torch.empty(N, H, H, 64, device="cuda", dtype=torch.bfloat16)
    nb = int(lib.wm_conv2d_fwd_stats_tiles(N, H, H, 16, 64, 4, 4, H, H, 1, 2, rpg)) if rpg else 1
    stat = torch.zeros(2, nb, 2, 64, device="cuda")  # [G][slots][stat][C] per-workgroup partial sums
    rpg = N * H * H // 2 if (N * H * H // 2) % (H * H) == 0 and (N * H * H // 2) % 128 == 0 else 0
    check(lib.wm_conv2d_fwd(ptr(x), ptr(w), ptr(y), N, H, H, 16, 64, 4, 4, H, H, 1, 2, st), "f")
    out[f"y{N}_{H}"] = y.cpu()
    if rpg:
        y2 = torch.empty_like(y)
        check(lib.wm_conv2d_fwd_stats(ptr(x), ptr(w), ptr(y2), N, H, H, 16, 64, 4, 4, H, H, 1, 2, ptr(stat), nb, rpg, st), "fs")
        out[f"ys{N}_{H}"] = y2.cpu()
        out[f"stat{N}_{H}"] = stat.sum(1).cpu()
        ref = y2.float().reshape(2, -1, 64)
        out[f"statref{N}_{H}"] = torch.stack([ref.sum(1), (ref * ref).sum(1)], 1).cpu()
torch.cuda.synchronize()
torch.save(out, sys.argv[1])
