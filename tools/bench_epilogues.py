"""What the fused epilogues cost on their own: conv forward with / without the BatchNorm statistics, dgrad with / without
the BatchNorm-backward epilogue (wm_conv2d_dgrad_bnstat), per ResNet-18 shape at batch 512, HIP events, cold-ish
(every launch streams tensors larger than what the previous one left in L2).
    python tools/bench_epilogues.py > gpurun_out/epilogues.txt"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ssl_wafermap_amd import _lib, ops  # noqa: E402
from ssl_wafermap_amd._lib import check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
N, G = 512, 2
SHAPES = [("layer1 3x3 64", 64, 56, 64, 3, 1, 1), ("layer2 3x3 128", 128, 28, 128, 3, 1, 1), ("layer2.0 3x3 s2", 64, 56, 128, 3, 2, 1),
          ("layer3 3x3 256", 256, 14, 256, 3, 1, 1), ("layer4 3x3 512", 512, 7, 512, 3, 1, 1)]


def timeit(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for name, C, H, K, R, stride, pad in SHAPES:
    P = (H + 2 * pad - R) // stride + 1
    x = torch.randn(N, H, H, C, generator=g, device=dev).bfloat16()
    y = torch.randn(N, P, P, K, generator=g, device=dev).bfloat16()
    wk = (torch.randn(K, R, R, C, generator=g, device=dev) * 0.05).bfloat16()
    wc = (torch.randn(C, R, R, K, generator=g, device=dev) * 0.05).bfloat16()
    dx, res = torch.empty_like(x), torch.randn_like(x.float()).bfloat16()
    bn_y = torch.randn_like(x.float()).bfloat16()
    mean, invstd = torch.zeros(G, C, device=dev), torch.ones(G, C, device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    geom = (N, H, H, C, K, R, R, P, P, stride, pad)
    nbf = int(lib.wm_conv2d_fwd_stats_tiles(*geom, N * P * P // G))
    nbb = N * H * H // G // 128
    stats = torch.empty(G * max(nbf, nbb) * 2 * max(C, K), device=dev)
    t = {}
    t["fwd"] = timeit(lambda: check(lib.wm_conv2d_fwd(ptr(x), ptr(wk), ptr(y), *geom, st), "f"))
    t["fwd+stats"] = timeit(lambda: check(lib.wm_conv2d_fwd_stats(ptr(x), ptr(wk), ptr(y), *geom, ptr(stats), nbf, N * P * P // G, st), "fs"))
    t["dgrad"] = timeit(lambda: check(lib.wm_conv2d_dgrad(ptr(y), ptr(wc), ptr(dx), *geom, st), "d"))
    t["dgrad+res"] = timeit(lambda: check(lib.wm_conv2d_dgrad_add(ptr(y), ptr(wc), ptr(res), ptr(dx), *geom, st), "da"))
    if lib.wm_conv2d_dgrad_bnstat_ok(*geom, G):
        t["dgrad+bnb(remask)"] = timeit(lambda: check(lib.wm_conv2d_dgrad_bnstat(
            ptr(y), ptr(wc), 0, ptr(dx), *geom, ptr(bn_y), 0, ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), G, ptr(stats), nbb, st), "b1"))
        t["dgrad+res+bnb(x mask)"] = timeit(lambda: check(lib.wm_conv2d_dgrad_bnstat(
            ptr(y), ptr(wc), ptr(res), ptr(dx), *geom, ptr(bn_y), ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), G, ptr(stats), nbb, st), "b2"))
    print(f"{name:18s} " + "  ".join(f"{k} {v:7.1f}us" for k, v in t.items()), flush=True)
