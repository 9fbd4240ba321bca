"""Race screen for the counted-wait DMA rings of linear_panel (panel.hip) and mlp_fused_fwd (mlp.hip): many launches at
the large shapes, under the memory load of back-to-back launches, every output compared with the conv_igemm path's."""
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
bad = 0
for rows, n, gelu in ((39424, 576, False), (39424, 768, True), (25216, 576, False), (25216, 768, True), (50000, 1152, False)):
    x = torch.randn(rows, 192, device=dev).bfloat16()
    w = (torch.randn(n, 192, device=dev) * 0.05).bfloat16()
    bias = torch.randn(n, device=dev)
    y, pre = torch.empty(rows, n, device=dev).bfloat16(), torch.empty(rows, n, device=dev).bfloat16()
    geom = (rows, 1, 1, 192, n, 1, 1, 1, 1, 1, 0)

    def run():
        if gelu:
            check(lib.wm_linear_bias_gelu_fwd(ptr(x), ptr(w), ptr(bias), ptr(pre), ptr(y), rows, 192, n, st), "g")
        else:
            check(lib.wm_conv2d_fwd_bias_res(ptr(x), ptr(w), ptr(bias), 0, ptr(y), *geom, st), "f")

    os.environ["WM_LINEAR_PANEL"] = "0"
    run()
    torch.cuda.synchronize()
    ref_y, ref_pre = y.clone(), pre.clone()
    os.environ["WM_LINEAR_PANEL"] = "1"
    n_bad = 0
    for it in range(150):
        y.fill_(float("nan"))
        run()
        if not torch.equal(y, ref_y) or (gelu and not torch.equal(pre, ref_pre)):
            n_bad += 1
    print(f"rows {rows} N {n} gelu {int(gelu)}: {n_bad} of 150 launches differ", flush=True)
    bad += n_bad
# fused MLP against itself (first launch as the reference)
for rows in (25216, 32768):
    x = torch.randn(rows, 192, device=dev).bfloat16()
    w1 = (torch.randn(768, 192, device=dev) * 0.05).bfloat16()
    w2 = (torch.randn(192, 768, device=dev) * 0.05).bfloat16()
    b1, b2 = torch.randn(768, device=dev), torch.randn(192, device=dev)
    y = torch.empty_like(x)
    check(lib.wm_mlp_fused_fwd(ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(x), ptr(y), rows, 192, 768, st), "m")
    torch.cuda.synchronize()
    ref = y.clone()
    n_bad = 0
    for it in range(150):
        y.fill_(float("nan"))
        check(lib.wm_mlp_fused_fwd(ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(x), ptr(y), rows, 192, 768, st), "m")
        if not torch.equal(y, ref):
            n_bad += 1
    print(f"fused MLP rows {rows}: {n_bad} of 150 launches differ", flush=True)
    bad += n_bad
print("TOTAL differing launches:", bad)
