"""kNN 811 457 x 128 bf16, 64 queries per batch: one call, and query batches pipelined over HIP streams
(functional.knn_topk_batched: all launches queued by one C call) -- us per batch and fraction of the 8 TB/s roofline."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ssl_wafermap_amd import functional as F  # noqa: E402

N, D, K = 811457, 128, 8
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(7)
bank = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1).bfloat16().contiguous()


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for bq in (64, 128):
    byts = N * D * 2 + bq * D * 2 + bq * K * 8
    q = bank[:bq].contiguous()
    ws = torch.empty(F._lib.load().wm_knn_topk_workspace_bytes(bq, N, D, K), dtype=torch.uint8, device=dev)
    one = timeit(lambda: F.knn_topk(q, bank, K, workspace=ws), 30)
    nb = 48
    qq = bank[2000:2000 + nb * bq].contiguous()
    row = {"queries": bq, "one_call_us": round(one, 1), "one_call_hbm_frac": round(byts / one / 1e3 / 8000, 3)}
    for lanes in (2, 3, 4):
        us = timeit(lambda: F.knn_topk_batched(qq, bank, K, batch=bq, lanes=lanes), 5) / nb
        row[f"lanes{lanes}_us"] = round(us, 1)
        row[f"lanes{lanes}_hbm_frac"] = round(byts / us / 1e3 / 8000, 3)
    print(json.dumps(row), flush=True)
