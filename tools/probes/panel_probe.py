"""Where does the time of the 192-wide panel Linear go?  WM_PANEL_DEBUG ablations (1 no stores, 2 no MFMAs, 4 no
staging, 8 x rows from cache) x WM_PANEL_SPLIT x rows, against conv_igemm (WM_LINEAR_PANEL=0); HIP-event timed."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from ssl_wafermap_amd import _lib  # noqa: E402
from ssl_wafermap_amd._lib import check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def run(rows, n, gelu, env):
    for k in ("WM_PANEL_DEBUG", "WM_PANEL_SPLIT", "WM_LINEAR_PANEL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    x = torch.randn(rows, 192, device=dev).bfloat16()
    w = (torch.randn(n, 192, device=dev) * 0.05).bfloat16()
    bias = torch.randn(n, device=dev)
    y = torch.empty(rows, n, device=dev).bfloat16()
    pre = torch.empty_like(y)
    if gelu:
        return timed(lambda: check(lib.wm_linear_bias_gelu_fwd(ptr(x), ptr(w), ptr(bias), ptr(pre), ptr(y), rows, 192, n, st), "g"))
    geom = (rows, 1, 1, 192, n, 1, 1, 1, 1, 1, 0)
    return timed(lambda: check(lib.wm_conv2d_fwd_bias_res(ptr(x), ptr(w), ptr(bias), 0, ptr(y), *geom, st), "f"))


for rows in (25216, 39424, 8192):
    for n, gelu in ((576, False), (768, True), (192, False)):
        out = [f"rows {rows} N {n} gelu {int(gelu)}:"]
        out.append(f"igemm {run(rows, n, gelu, {'WM_LINEAR_PANEL': '0'}):.1f}")
        for split in ("0", "1", "3"):
            out.append(f"panel split{split} {run(rows, n, gelu, {'WM_PANEL_SPLIT': split}):.1f}")
        for dbg in (1, 2, 4, 8, 3, 7, 15):
            out.append(f"dbg{dbg} {run(rows, n, gelu, {'WM_PANEL_DEBUG': str(dbg)}):.1f}")
        print(" | ".join(out), flush=True)
