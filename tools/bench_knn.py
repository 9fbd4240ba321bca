"""kNN retrieval micro-benchmark: 811 457 x 128 bank, query batches Bq.
Reports the streaming-HBM figure (formula ii: bank bytes per query batch / kernel time) and the
dense-contraction figure (2*Bq*N*D FLOP / time).  Timed with HIP events on the launch stream."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ssl_wafermap_amd import functional as F  # noqa: E402

N, D, K = 811457, 128, 8
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(7)
bank32 = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1)
out = []
for dtype in (torch.bfloat16, torch.float32):
    bank = bank32.to(dtype).contiguous()
    for bq in (64, 128, 256, 1024):
        q = bank[:bq].contiguous()
        ws = torch.empty(F._lib.load().wm_knn_topk_workspace_bytes(bq, N, D, K), dtype=torch.uint8, device=dev)
        for _ in range(3):
            F.knn_topk(q, bank, K, workspace=ws)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        a.record()
        for _ in range(reps):
            F.knn_topk(q, bank, K, workspace=ws)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        byts = N * D * bank.element_size() + bq * D * bank.element_size() + bq * K * 8
        out.append({"dtype": str(dtype).split(".")[-1], "Bq": bq, "ms": round(ms, 4),
                    "stream_GBps": round(byts / ms / 1e6, 1), "frac_of_8TBps": round(byts / ms / 1e6 / 8000, 3),
                    "TFLOPs": round(2.0 * bq * N * D / ms / 1e9, 1)})
        print(json.dumps(out[-1]), flush=True)
