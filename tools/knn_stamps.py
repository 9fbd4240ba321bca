"""Phase timestamps of the d=128 bf16 kNN stream kernel (WM_KNN_DEBUG=4; timing only, results are wrong)."""
import os
import sys
from pathlib import Path

os.environ["WM_KNN_DEBUG"] = os.environ.get("WM_KNN_DEBUG", "4")
import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from ssl_wafermap_amd import functional as F  # noqa: E402

N, D, K = 811457, 128, 8
bq = 64
g = torch.Generator(device="cuda").manual_seed(7)
bank = torch.nn.functional.normalize(torch.randn(N, D, generator=g, device="cuda"), dim=1).bfloat16().contiguous()
q = bank[:bq].contiguous()
ws = torch.zeros(F._lib.load().wm_knn_topk_workspace_bytes(bq, N, D, K), dtype=torch.uint8, device="cuda")
for rep in range(4):
    F.knn_topk(q, bank, K, workspace=ws)
    torch.cuda.synchronize()
    nsl = int(os.environ.get('NSL', '488'))
    cand = 64 * nsl * 8
    idx = ws[: cand * 4].cpu().numpy().view(np.int32)[: nsl * 8].view(np.uint64).reshape(nsl, 4).astype(np.int64)
    t = (idx - idx[:, 0].min()) / 100.0  # us at 100 MHz
    print(f"rep {rep}: start {t[:,0].min():.2f}..{t[:,0].max():.2f}  loop-begin {t[:,1].min():.2f}..{t[:,1].max():.2f} "
          f"(mean {t[:,1].mean():.2f})  loop-end {t[:,2].min():.2f}..{t[:,2].max():.2f} (mean {t[:,2].mean():.2f})  "
          f"done {t[:,3].min():.2f}..{t[:,3].max():.2f} (mean {t[:,3].mean():.2f}); per-block: prologue "
          f"{(t[:,1]-t[:,0]).mean():.2f} loop {(t[:,2]-t[:,1]).mean():.2f} tail {(t[:,3]-t[:,2]).mean():.2f}")
    if rep == 3:
        for x in range(8):
            m = np.arange(nsl) % 8 == x
            print(f"  xcd {x}: start {t[m,0].mean():.2f} loop {(t[m,2]-t[m,1]).mean():.2f} (min {(t[m,2]-t[m,1]).min():.2f} max {(t[m,2]-t[m,1]).max():.2f}) end {t[m,3].max():.2f}")
        half = np.arange(nsl) < 256
        print(f"  first 256 blocks: start {t[half,0].mean():.2f} loop {(t[half,2]-t[half,1]).mean():.2f}; rest: start {t[~half,0].mean():.2f} loop {(t[~half,2]-t[~half,1]).mean():.2f}")
    if rep == 3:
        sel = ws[: cand * 4].cpu().numpy().view(np.int32).reshape(64, nsl * 8)[:, :10].copy().view(np.uint64).astype(np.int64)
        ts = (sel - sel[:, :1]) / 100.0
        print("  select phases (us since block start): heads %.2f  groups %.2f  rescored %.2f  done %.2f; block starts spread %.2f" % (
            ts[:, 1].mean(), ts[:, 2].mean(), ts[:, 3].mean(), ts[:, 4].mean(), (sel[:, 0].max() - sel[:, 0].min()) / 100.0))
