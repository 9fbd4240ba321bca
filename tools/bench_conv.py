"""Per-layer conv micro-benchmark: every distinct ResNet-18 convolution at batch 512 (2 views x 256),
forward / dgrad / wgrad, HIP-event timed on the launch stream.  `--only i` runs one shape (PMC runs)."""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ssl_wafermap_amd import _lib  # noqa: E402
from ssl_wafermap_amd._lib import check, ptr  # noqa: E402

# (name, C, H, W, K, R, stride, pad, count per step)
SHAPES = [
    ("stem 4x4x16 s2d", 16, 112, 112, 64, 4, 1, 2, 1),
    ("layer1 3x3 64->64", 64, 56, 56, 64, 3, 1, 1, 4),
    ("layer2.0 3x3/2 64->128", 64, 56, 56, 128, 3, 2, 1, 1),
    ("layer2 ds 1x1/2 64->128", 64, 56, 56, 128, 1, 2, 0, 1),
    ("layer2 3x3 128->128", 128, 28, 28, 128, 3, 1, 1, 3),
    ("layer3.0 3x3/2 128->256", 128, 28, 28, 256, 3, 2, 1, 1),
    ("layer3 ds 1x1/2 128->256", 128, 28, 28, 256, 1, 2, 0, 1),
    ("layer3 3x3 256->256", 256, 14, 14, 256, 3, 1, 1, 3),
    ("layer4.0 3x3/2 256->512", 256, 14, 14, 512, 3, 2, 1, 1),
    ("layer4 ds 1x1/2 256->512", 256, 14, 14, 512, 1, 2, 0, 1),
    ("layer4 3x3 512->512", 512, 7, 7, 512, 3, 1, 1, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--modes", default="fwd,dgrad,wgrad")
    ap.add_argument("--lib", default=None, help="another build of the library (ablation builds: -DWM_CONV_ABLATE=n)")
    a = ap.parse_args()
    lib = _lib.load(a.lib) if a.lib else _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    for i, (name, C, H, W, K, R, stride, pad, cnt) in enumerate(SHAPES):
        if a.only >= 0 and i != a.only:
            continue
        N = a.n
        P = (H + 2 * pad - R) // stride + 1 if C != 16 else H
        Q = P if C != 16 else W
        x = torch.randn(N, H, W, C, generator=g, device=dev).bfloat16()
        y = torch.randn(N, P, Q, K, generator=g, device=dev).bfloat16()
        wk = (torch.randn(K, R, R, C, generator=g, device=dev) * 0.05).bfloat16()
        wc = (torch.randn(C, R, R, K, generator=g, device=dev) * 0.05).bfloat16()
        ns = int(lib.wm_conv2d_wgrad_splits(N, H, W, C, K, R, R, P, Q, stride, pad))
        dw = torch.empty(max(ns, 1), K, R, R, C, device=dev)  # split-K slabs
        dx = torch.empty_like(x)
        geom = (N, H, W, C, K, R, R, P, Q, stride, pad)
        flops = 2.0 * N * P * Q * K * R * R * C
        res = {"shape": name, "GFLOP": round(flops / 1e9, 1)}
        # BatchNorm-backward epilogue forms of dgrad (wm_conv2d_dgrad_bnstat): the step's 15 ReLU'd BatchNorm layers
        G = 2
        bnb_ok = C != 16 and bool(lib.wm_conv2d_dgrad_bnstat_ok(*geom, G))
        if bnb_ok:
            tiles = N * H * W // G // 128
            bn_y = torch.randn(N, H, W, C, generator=g, device=dev).bfloat16()
            resid = torch.randn(N, H, W, C, generator=g, device=dev).bfloat16()
            mask = torch.randint(0, 256, (N * H * W, C // 8), generator=g, device=dev, dtype=torch.uint8)
            gam, bet = torch.rand(C, generator=g, device=dev) + 0.5, torch.randn(C, generator=g, device=dev) * 0.1
            mean, istd = torch.randn(G, C, generator=g, device=dev) * 0.1, torch.rand(G, C, generator=g, device=dev) + 0.5
            slots = torch.empty(G, tiles, 2, C, device=dev)

            def bnb(with_res):
                return check(lib.wm_conv2d_dgrad_bnstat(ptr(y), ptr(wc), ptr(resid) if with_res else 0, ptr(dx), *geom, ptr(bn_y), 0,
                                                        ptr(mask) if with_res else 0, ptr(gam), ptr(bet), ptr(mean), ptr(istd), G,
                                                        ptr(slots), tiles, st), "dgrad_bnstat")
        stat_tiles = int(lib.wm_conv2d_fwd_stats_tiles(*geom, N * P * Q // 2)) if (N * P * Q // 2) % 128 == 0 else 0
        fslots = torch.empty(2, max(stat_tiles, 1), 2, K, device=dev)
        calls = {
            "fwd_stats": lambda: check(lib.wm_conv2d_fwd_stats(ptr(x), ptr(wk), ptr(y), *geom, ptr(fslots), stat_tiles,
                                                               N * P * Q // 2, st), "fwd_stats"),
            "dgrad_bnb": lambda: bnb(False),
            "dgrad_bnb_res": lambda: bnb(True),
            "fwd": lambda: check(lib.wm_conv2d_fwd(ptr(x), ptr(wk), ptr(y), *geom, st), "fwd"),
            "dgrad": lambda: check(lib.wm_conv2d_dgrad(ptr(y), ptr(wc), ptr(dx), *geom, st), "dgrad"),
            "wgrad": lambda: check(lib.wm_conv2d_wgrad(ptr(y), ptr(x), ptr(dw), *geom, st), "wgrad"),
        }
        for mode in a.modes.split(","):
            if mode.startswith("dgrad") and C == 16:
                continue
            if mode.startswith("dgrad_bnb") and not bnb_ok:
                continue
            if mode == "fwd_stats" and stat_tiles <= 0:
                continue
            f = calls[mode]
            for _ in range(2):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.reps * 1e3
            res[mode + "_us"] = round(us, 1)
            res[mode + "_TF"] = round(flops / us / 1e6, 0)
            tot[mode] = tot.get(mode, 0.0) + us * cnt
        print(json.dumps(res), flush=True)
    if a.only < 0:
        print(json.dumps({"per_step_ms": {k: round(v / 1e3, 3) for k, v in tot.items()},
                          "sum_ms": round(sum(tot.values()) / 1e3, 3)}))


if __name__ == "__main__":
    main()
