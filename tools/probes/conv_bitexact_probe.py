"""Run a fixed set of conv / Linear launches through the C ABI and save the outputs: run once per library build and
compare the files (kernel-restructuring changes that must be bit-identical)."""
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
# (N, H, C, K, R, stride, pad)
for (N, H, C, K, R, stride, pad) in ((8, 14, 256, 256, 3, 1, 1), (6, 28, 128, 128, 3, 1, 1), (4, 7, 512, 512, 3, 1, 1),
                                     (5, 56, 64, 128, 3, 2, 1), (5, 56, 64, 128, 1, 2, 0), (3, 13, 64, 64, 3, 1, 1),
                                     (2, 9, 128, 192, 5, 1, 2), (7, 10, 64, 64, 1, 1, 0)):
    P = (H + 2 * pad - R) // stride + 1
    x = torch.randn(N, H, H, C, generator=g, device="cuda").bfloat16()
    dy = torch.randn(N, P, P, K, generator=g, device="cuda").bfloat16()
    wk = (torch.randn(K, R, R, C, generator=g, device="cuda") * 0.05).bfloat16()
    wc = (torch.randn(C, R, R, K, generator=g, device="cuda") * 0.05).bfloat16()
    y = torch.empty(N, P, P, K, device="cuda", dtype=torch.bfloat16)
    dx = torch.empty_like(x)
    res = torch.randn(N, H, H, C, generator=g, device="cuda").bfloat16()
    geom = (N, H, H, C, K, R, R, P, P, stride, pad)
    check(lib.wm_conv2d_fwd(ptr(x), ptr(wk), ptr(y), *geom, st), "f")
    check(lib.wm_conv2d_dgrad(ptr(dy), ptr(wc), ptr(dx), *geom, st), "d")
    tag = f"{N}_{H}_{C}_{K}_{R}_{stride}"
    out["y" + tag], out["dx" + tag] = y.cpu(), dx.cpu()
    check(lib.wm_conv2d_dgrad_add(ptr(dy), ptr(wc), ptr(res), ptr(dx), *geom, st), "da")
    out["dxa" + tag] = dx.cpu()
for (rows, C, K) in ((1000, 384, 1152), (777, 192, 192), (300, 1536, 384), (129, 64, 64), (64, 2048, 128)):
    x = torch.randn(rows, C, generator=g, device="cuda").bfloat16()
    dy = torch.randn(rows, K, generator=g, device="cuda").bfloat16()
    wk = (torch.randn(K, C, generator=g, device="cuda") * 0.05).bfloat16()
    wc = (torch.randn(C, K, generator=g, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(K, generator=g, device="cuda")
    res = torch.randn(rows, K, generator=g, device="cuda").bfloat16()
    y = torch.empty(rows, K, device="cuda", dtype=torch.bfloat16)
    pre = torch.empty_like(y)
    dx = torch.empty_like(x)
    geom = (rows, 1, 1, C, K, 1, 1, 1, 1, 1, 0)
    check(lib.wm_conv2d_fwd_bias_res(ptr(x), ptr(wk), ptr(bias), ptr(res), ptr(y), *geom, st), "l")
    out[f"ly{rows}_{C}_{K}"] = y.cpu()
    check(lib.wm_linear_bias_gelu_fwd(ptr(x), ptr(wk), ptr(bias), ptr(pre), ptr(y), rows, C, K, st), "g")
    out[f"lg{rows}_{C}_{K}"], out[f"lp{rows}_{C}_{K}"] = y.cpu(), pre.cpu()
    check(lib.wm_conv2d_dgrad(ptr(dy), ptr(wc), ptr(dx), *geom, st), "ld")
    out[f"ldx{rows}_{C}_{K}"] = dx.cpu()
    prex = torch.randn(rows, C, generator=g, device="cuda").bfloat16()
    check(lib.wm_linear_dgrad_gelu(ptr(dy), ptr(wc), ptr(prex), ptr(dx), rows, C, K, st), "ldg")
    out[f"ldg{rows}_{C}_{K}"] = dx.cpu()
torch.cuda.synchronize()
torch.save(out, sys.argv[1])
