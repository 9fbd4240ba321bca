"""Do repeated launches of the layer1 convolution kernels give bit-identical outputs? (race hunt)"""
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
for (N, H, C, K) in ((512, 56, 64, 64), (64, 56, 64, 64), (512, 28, 128, 128)):
    x = torch.randn(N, H, H, C, generator=g, device=dev).bfloat16()
    wk = (torch.randn(K, 3, 3, C, generator=g, device=dev) * 0.05).bfloat16()
    geom = (N, H, H, C, K, 3, 3, H, H, 1, 1)
    G = 2
    rpg = N * H * H // G
    T = int(lib.wm_conv2d_fwd_stats_tiles(*geom, rpg))
    outs, stats = [], []
    for rep in range(6):
        y = torch.empty(N, H, H, K, device=dev, dtype=torch.bfloat16)
        s = torch.zeros(G, T, 2, K, device=dev)
        if rep % 2 == 0:
            check(lib.wm_conv2d_fwd(ptr(x), ptr(wk), ptr(y), *geom, st), "f")
        else:
            check(lib.wm_conv2d_fwd_stats(ptr(x), ptr(wk), ptr(y), *geom, ptr(s), T, rpg, st), "fs")
            stats.append(s.clone())
        # something else in between (as in a training step): a big streaming kernel
        torch.randn(64 << 20, device=dev).sum().item()
        outs.append(y.clone())
    torch.cuda.synchronize()
    same = [bool(torch.equal(outs[0], o)) for o in outs[1:]]
    nd = [int((outs[0] != o).sum()) for o in outs[1:]]
    print(f"fwd N={N} H={H} C={C}: outputs equal to first: {same}  differing elements {nd}; "
          f"stats equal: {[bool(torch.equal(stats[0], t)) for t in stats[1:]]}", flush=True)
    # dgrad
    wc = (torch.randn(C, 3, 3, K, generator=g, device=dev) * 0.05).bfloat16()
    dys = []
    for rep in range(4):
        dx = torch.empty(N, H, H, C, device=dev, dtype=torch.bfloat16)
        check(lib.wm_conv2d_dgrad(ptr(y), ptr(wc), ptr(dx), *geom, st), "d")
        torch.randn(64 << 20, device=dev).sum().item()
        dys.append(dx.clone())
    print(f"dgrad: equal to first: {[bool(torch.equal(dys[0], o)) for o in dys[1:]]}", flush=True)
