"""One kNN configuration, a few launches (target of rocprofv3 --pmc passes)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ssl_wafermap_amd import functional as F  # noqa: E402

bq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
N, D, K = 811457, 128, 8
g = torch.Generator(device="cuda").manual_seed(7)
bank = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device="cuda"), dim=1).to(dt).contiguous()
q = bank[:bq].contiguous()
for _ in range(5):
    F.knn_topk(q, bank, K)
torch.cuda.synchronize()
