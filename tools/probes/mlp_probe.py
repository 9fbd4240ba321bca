"""Fused MLP forward (wm_mlp_fused_fwd) against the two-launch path, HIP-event timed; checks bit-identity."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from ssl_wafermap_amd import _lib  # noqa: E402
from ssl_wafermap_amd._lib import check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
C, H = 192, 768
for rows in (25216, 39424, 8192, 300):
    x = torch.randn(rows, C, device=dev).bfloat16()
    w1 = (torch.randn(H, C, device=dev) * 0.05).bfloat16()
    w2 = (torch.randn(C, H, device=dev) * 0.05).bfloat16()
    b1, b2 = torch.randn(H, device=dev), torch.randn(C, device=dev)
    res = torch.randn(rows, C, device=dev).bfloat16()
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    pre, h = torch.empty(rows, H, device=dev).bfloat16(), torch.empty(rows, H, device=dev).bfloat16()

    def fused():
        check(lib.wm_mlp_fused_fwd(ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(res), ptr(y1), rows, C, H, st), "fused")

    def two():
        check(lib.wm_linear_bias_gelu_fwd(ptr(x), ptr(w1), ptr(b1), ptr(pre), ptr(h), rows, C, H, st), "fc1")
        check(lib.wm_conv2d_fwd_bias_res(ptr(h), ptr(w2), ptr(b2), ptr(res), ptr(y2), rows, 1, 1, H, C, 1, 1, 1, 1, 1, 0, st), "fc2")

    out = []
    for fn in (fused, two):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / 40)
    print(f"rows {rows}: fused {out[0]:.1f} us, two launches {out[1]:.1f} us, identical {bool(torch.equal(y1, y2))}", flush=True)
