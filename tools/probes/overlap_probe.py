"""Do an MFMA-bound convolution and an HBM-bound BatchNorm apply pass overlap when issued on two streams?
(layer2-sized tensors of ONE view: 256 images x 28 x 28 x 128)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from ssl_wafermap_amd import ops

DEV = "cuda:0"
torch.manual_seed(0)


def mk(n, c, h):
    x = ops.to_nhwc_bf16(torch.randn(n, c, h, h, device=DEV))
    w = torch.nn.Parameter(torch.randn(c, c, 3, 3, device=DEV) * 0.05)
    g, b = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    return x, w, g, b, rm, rv


def run(label, n, c, h, reps=20):
    xa, wa, *_ = mk(n, c, h)
    xb, _, g, b, rm, rv = mk(n, c, h)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def conv():
        return ops.conv2d(xa, wa, 1, 1)

    def bn():
        return ops.batch_norm(xb, g, b, rm, rv, True, relu=True)

    with torch.no_grad():
        for f in (conv, bn):
            for _ in range(3):
                f()
        torch.cuda.synchronize()

        def timed(fa, fb, ra, rb):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            s1.wait_stream(torch.cuda.current_stream())
            s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s1):
                for _ in range(ra):
                    fa()
            with torch.cuda.stream(s2):
                for _ in range(rb):
                    fb()
            torch.cuda.current_stream().wait_stream(s1)
            torch.cuda.current_stream().wait_stream(s2)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3

        nothing = lambda: None
        tc = timed(conv, nothing, reps, 0)
        # as many BN passes as take about the conv time
        tb1 = timed(nothing, bn, 0, reps)
        rb = max(1, int(round(reps * tc / tb1)))
        tb = timed(nothing, bn, 0, rb)
        both = timed(conv, bn, reps, rb)
    print(f"{label}: conv x{reps} {tc:.0f} us, bn x{rb} {tb:.0f} us, both streams {both:.0f} us  (sum {tc + tb:.0f}, max {max(tc, tb):.0f})")


run("layer2 one view (256 x 28 x 28 x 128)", 256, 128, 28)
run("layer3 one view (256 x 14 x 14 x 256)", 256, 256, 14)
run("layer1 one view (256 x 56 x 56 x 64)", 256, 64, 56)
run("layer2 both views (512 x 28 x 28 x 128)", 512, 128, 28)
