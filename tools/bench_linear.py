"""Linear-layer GEMM micro-benchmark: the transformer shapes of DINO ViT-S/16 and ViT-Tiny/16 at 64 wafers per step
(global crops 2 x 64 x 197 rows, local crops 6 x 64 x 37 rows), forward (+bias), dgrad, wgrad (+bias), fused GELU
forward / dgrad; HIP-event timed on the launch stream."""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ssl_wafermap_amd import _lib  # noqa: E402
from ssl_wafermap_amd._lib import check, ptr  # noqa: E402

ROWS_G, ROWS_L = 2 * 64 * 197, 6 * 64 * 37
# (name, rows, C (in), K (out))
SHAPES = []
for tag, e in (("S", 384), ("T", 192)):
    for rn, rows in (("glob", ROWS_G), ("loc", ROWS_L)):
        SHAPES += [
            (f"{tag} {rn} qkv", rows, e, 3 * e),
            (f"{tag} {rn} proj", rows, e, e),
            (f"{tag} {rn} fc1", rows, e, 4 * e),
            (f"{tag} {rn} fc2", rows, 4 * e, e),
        ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = {}
    for name, rows, C, K in SHAPES:
        if a.only and a.only not in name:
            continue
        x = torch.randn(rows, C, generator=g, device=dev).bfloat16()
        y = torch.randn(rows, K, generator=g, device=dev).bfloat16()
        pre = torch.empty_like(y)
        res = torch.randn(rows, K, generator=g, device=dev).bfloat16()
        wk = (torch.randn(K, C, generator=g, device=dev) * 0.05).bfloat16()
        wc = (torch.randn(C, K, generator=g, device=dev) * 0.05).bfloat16()
        bias = torch.randn(K, generator=g, device=dev)
        ns = int(lib.wm_conv2d_wgrad_splits(rows, 1, 1, C, K, 1, 1, 1, 1, 1, 0))
        dw = torch.empty(max(ns, 1), K, C, device=dev)  # split-K slabs
        db = torch.empty(max(ns, 1), K, device=dev)
        dx = torch.empty_like(x)
        geom = (rows, 1, 1, C, K, 1, 1, 1, 1, 1, 0)
        flops = 2.0 * rows * C * K
        calls = {
            "fwd": lambda: check(lib.wm_conv2d_fwd_bias_res(ptr(x), ptr(wk), ptr(bias), ptr(res), ptr(y), *geom, st), "f"),
            "dgrad": lambda: check(lib.wm_conv2d_dgrad(ptr(y), ptr(wc), ptr(dx), *geom, st), "d"),
            "wgrad": lambda: check(lib.wm_conv2d_wgrad_bias(ptr(y), ptr(x), ptr(dw), ptr(db), *geom, st), "w"),
        }
        if "fc1" in name:
            calls["fwd_gelu"] = lambda: check(lib.wm_linear_bias_gelu_fwd(ptr(x), ptr(wk), ptr(bias), ptr(pre), ptr(y), rows, C, K, st), "g")
        if "fc2" in name:
            # dgrad of fc2 x gelu'(pre): dy [rows, K] -> dpre [rows, C]
            prex = torch.randn(rows, C, generator=g, device=dev).bfloat16()
            calls["dgrad_gelu"] = lambda: check(lib.wm_linear_dgrad_gelu(ptr(y), ptr(wc), ptr(prex), ptr(dx), rows, C, K, st), "dg")
        out = {"shape": name, "rows": rows, "C": C, "K": K, "GFLOP": round(flops / 1e9, 1)}
        for mode, fn in calls.items():
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.reps
            out[mode] = f"{us:.1f}us {flops / us / 1e6:.0f}TF"
            tot[mode] = tot.get(mode, 0.0) + us
        print(json.dumps(out), flush=True)
    print(json.dumps({"total_us": {k: round(v, 1) for k, v in tot.items()}}))


if __name__ == "__main__":
    main()
