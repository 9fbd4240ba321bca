"""Workload for a kernel trace of the PIPELINED kNN mode (bench.py knn.rows[0].pipelined_*): 811 457 x 128 bf16 bank, 64 queries per
batch, 256 batches per call round-robin on three HIP streams (functional.knn_topk_batched -> wm_knn_topk_many), 3 calls.
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/knn_pipe -o p -- python3 tools/knn_pipelined_trace.py
    python3 tools/summarize_knn_trace.py gpurun_out/knn_pipe profiles/r04_knn_pipelined_trace.md"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ssl_wafermap_amd import functional as F  # noqa: E402

N, D, K, BQ, NB = 811457, 128, 8, 64, 256
dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(7)
bank = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device=dev), dim=1).bfloat16().contiguous()
qq = bank[2000:2000 + NB * BQ].contiguous()
for _ in range(3):
    sim, idx = F.knn_topk_batched(qq, bank, K, batch=BQ)
    torch.cuda.synchronize()
assert bool((sim[:, 0] > 0.99).all())   # every query is a bank row: it finds itself
print("ok")
