"""attention backward, one block per (image, head) against the role-split kernel (WM_ATTN_BWD_ROLES=0 / 1)."""
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
for B, S, H, hd in ((128, 197, 3, 64), (128, 197, 6, 64), (64, 197, 6, 64), (512, 50, 12, 64), (64, 197, 16, 32)):
    qkv = torch.randn(B * S, 3 * H * hd, device=dev).bfloat16()
    out = torch.empty(B * S, H * hd, device=dev).bfloat16()
    lse = torch.empty(B, H, S, device=dev)
    dout = torch.randn(B * S, H * hd, device=dev).bfloat16()
    check(lib.wm_attention_fwd(ptr(qkv), B, S, H, hd, hd ** -0.5, ptr(out), ptr(lse), st), "f")
    res = {}
    for flag in ("0", "1"):
        os.environ["WM_ATTN_BWD_ROLES"] = flag
        dq = torch.empty_like(qkv)
        fn = lambda: check(lib.wm_attention_bwd(ptr(qkv), ptr(out), ptr(dout), ptr(lse), B, S, H, hd, hd ** -0.5, ptr(dq), st), "b")
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[flag] = (e0.elapsed_time(e1) * 1e3 / 30, dq.float())
    rel = float((res["0"][1] - res["1"][1]).norm() / res["0"][1].norm())
    print(f"B {B} S {S} H {H} hd {hd}: one block per head {res['0'][0]:.1f} us, roles {res['1'][0]:.1f} us, relative difference {rel:.2e}", flush=True)
