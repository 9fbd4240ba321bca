"""Is the dgrad instantiation (MODE 3) of conv_igemm slower than the forward one (MODE 0) on the SAME operands and
geometry?  Times wm_conv2d_fwd and wm_conv2d_dgrad on 256->256 and 512->512 3x3 stride-1 layers with the roles of the
tensors swapped so both launches read the same bytes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ssl_wafermap_amd import _lib  # noqa: E402
from ssl_wafermap_amd._lib import check, ptr  # noqa: E402

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)


def t(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for (N, H, C) in ((512, 14, 256), (512, 7, 512), (512, 28, 128)):
    src = torch.randn(N, H, H, C, generator=g, device="cuda").bfloat16()
    w = (torch.randn(C, 3, 3, C, generator=g, device="cuda") * 0.05).bfloat16()
    dst = torch.empty_like(src)
    geom = (N, H, H, C, C, 3, 3, H, H, 1, 1)
    f = t(lambda: check(lib.wm_conv2d_fwd(ptr(src), ptr(w), ptr(dst), *geom, st), "f"))
    d = t(lambda: check(lib.wm_conv2d_dgrad(ptr(src), ptr(w), ptr(dst), *geom, st), "d"))
    fl = 2.0 * N * H * H * C * 9 * C
    print(f"{C}ch {H}x{H}: fwd-mode {f:.1f} us ({fl / f / 1e6:.0f} TF)   dgrad-mode {d:.1f} us ({fl / d / 1e6:.0f} TF)")
