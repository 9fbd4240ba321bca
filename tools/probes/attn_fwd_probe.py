"""attention forward: query strips of a head on one block or split over two (WM_ATTN_FWD_SPLIT is read once per process:
run this script once per setting)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from ssl_wafermap_amd import _lib  # noqa: E402
from ssl_wafermap_amd._lib import check, ptr  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for B, S, H, hd in ((128, 197, 3, 64), (128, 197, 6, 64), (384, 37, 3, 64), (64, 197, 16, 32)):
    qkv = torch.randn(B * S, 3 * H * hd, device=dev).bfloat16()
    out = torch.empty(B * S, H * hd, device=dev).bfloat16()
    lse = torch.empty(B, H, S, device=dev)
    fn = lambda: check(lib.wm_attention_fwd(ptr(qkv), B, S, H, hd, hd ** -0.5, ptr(out), ptr(lse), st), "f")
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"B {B} S {S} H {H} hd {hd}: {e0.elapsed_time(e1) * 1e3 / 40:.1f} us  checksum {float(out.float().abs().sum()):.6e}", flush=True)
