"""knn_topk_batched timed as bench.py's knn object does (one call after one warm-up) and over several calls."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from ssl_wafermap_amd import functional as F  # noqa: E402
N, D, K, bq, nb = 811457, 128, 8, 64, 48
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
bank = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1).bfloat16()
qq = bank[2000:2000 + nb * bq].contiguous()
F.knn_topk_batched(qq, bank, K, batch=bq)
for reps in (1, 1, 1, 5, 5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        F.knn_topk_batched(qq, bank, K, batch=bq)
    b.record()
    torch.cuda.synchronize()
    print(reps, "calls:", round(a.elapsed_time(b) * 1e3 / nb / reps, 1), "us per batch")
