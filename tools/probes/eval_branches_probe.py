"""Eval-mode ResNet-18 forward (embedding inference) with the batch as two half-batch branches vs one stream."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from ssl_wafermap_amd import ops
from ssl_wafermap_amd.models.resnet import create_model

DEV = "cuda:0"
torch.manual_seed(0)
m = create_model("resnet18").to(DEV).eval()
for n in (256, 512):
    x = ops.to_nhwc_bf16(torch.randn(n, 3, 224, 224, device=DEV))
    res = {}
    for mode in ("0", "1"):
        os.environ["WM_EVAL_BRANCHES"] = mode
        with torch.no_grad():
            for _ in range(3):
                y = m(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                y = m(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g.replay()
            e0.record()
            for _ in range(20):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            res[mode] = (e0.elapsed_time(e1) / 20, y.float().clone())
    print(f"batch {n}: one stream {res['0'][0]:.3f} ms, two branches {res['1'][0]:.3f} ms, outputs equal {torch.equal(res['0'][1], res['1'][1])}")
