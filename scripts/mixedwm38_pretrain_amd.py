#!/usr/bin/env python3
"""MixedWM38 self-supervised pre-training on MI355X: the reference's scripts/MixedWM38_pretrain.py main() (:566-654)
on the HIP path, step for step.

    python scripts/mixedwm38_pretrain_amd.py [--full] [--data-root /path/to/reference/data] [--models MAE,DINOViT]
                                             [--max-epochs N] [--batch-size 64] [--out DIR] [--mae-backbone vit_small_16]
    torchrun --nproc-per-node N scripts/mixedwm38_pretrain_amd.py ...      (data parallel, reference `distributed=True`)

What the reference does, and where it is here:
  :82-87    subset mode: train_5_split.pkl.xz; full: train_data.pkl.xz          -> load_data()
  :89-93    LightlyDataset over WaferMapDataset(waferMap, failureType codes): no dataset transform, the collate
            function augments                                                    -> WaferMapDataset(transform=None)
  :95-103   collate functions, all with denoise=True (3 x 3 median as the first-stage alternative of DPW)
  :106-135  get_data_loader(): the collate function by model class              -> get_data_loader()
  :566-649  for each model: seed, loader, model(), Trainer.fit (no validation), run record, results.csv -> main()
Model classes are this repo's (same names / hyper-parameters as the script's: DINOViT with batch_norm=False heads
:146-151, MAE on torchvision's ViT-B/32 geometry :261-281, ...).  `--mae-backbone vit_small_16` selects BASELINE.json
configs[3] (MAE ViT-S/16) instead of the reference's ViT-B/32.  Not carried over: TensorBoard logger, ModelCheckpoint.

Data: `--data-root` = the reference's `data/` directory.  Without it the subset mode runs from the data-only fixture
tests/golden/mixedwm38_train_1_split.npz (381 maps; the reference's own subset file is train_5_split).
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

logs_root_dir = "mixed_wm38_pretrain"
MODEL_NAMES = ["SwaV", "MSN", "DCLW", "VICReg", "MAE", "BYOL", "DINOViT"]   # reference order, :567-581


def load_data(subset: bool, data_root):
    from ssl_wafermap_amd.data.store import WaferStore

    if data_root:
        import pandas as pd

        name = "train_5_split" if subset else "train_data"
        df = pd.read_pickle(Path(data_root) / "processed/MixedWM38" / f"{name}.pkl.xz")
        return WaferStore(df.waferMap.tolist()), df.failureType.factorize(sort=True)[0].astype(np.int64)
    if not subset:
        raise SystemExit("--full needs --data-root (train_data.pkl.xz is not shipped as a fixture)")
    store, labels = WaferStore.load(ROOT / "tests/golden/mixedwm38_train_1_split.npz")
    return store, np.asarray(labels).astype(np.int64)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="reference `subset = False`: train_data.pkl.xz, 150 epochs")
    ap.add_argument("--data-root", default=None)
    ap.add_argument("--models", default=",".join(MODEL_NAMES))
    ap.add_argument("--max-epochs", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--n-runs", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--limit-train-batches", type=int, default=None)
    ap.add_argument("--mae-backbone", default="vit_b_32", choices=["vit_b_32", "vit_small_16"])
    ap.add_argument("--log-every", type=int, default=50)
    args = ap.parse_args(argv)

    import pandas as pd
    import torch

    import ssl_wafermap_amd.models as zoo
    from ssl_wafermap_amd import distributed as wdist
    from ssl_wafermap_amd.data import WaferCollateLoader, WaferMapDataset
    from ssl_wafermap_amd.trainer import Trainer
    from ssl_wafermap_amd.transforms import (WaferDINOCOllateFunction, WaferImageCollateFunction, WaferMAECollateFunction2,
                                             WaferMSNCollateFunction, WaferSwaVCollateFunction)

    subset = not args.full
    max_epochs = args.max_epochs if args.max_epochs is not None else (5 if subset else 150)
    rank, world, local = wdist.init_from_env()
    batch_size = args.batch_size // world if world > 1 else args.batch_size   # reference :74-77
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    store, labels = load_data(subset, args.data_root)
    store.to(dev)
    dataset_train_ssl = WaferMapDataset(store, labels, transform=None, device=dev)
    collate_fn = WaferImageCollateFunction(denoise=True)
    dino_collate_fn = WaferDINOCOllateFunction(denoise=True)
    mae_collate_fn = WaferMAECollateFunction2(denoise=True)
    msn_collate_fn = WaferMSNCollateFunction(denoise=True)
    swav_collate_fn = WaferSwaVCollateFunction(denoise=True)

    def get_data_loader(batch_size, model, seed):
        col_fn = {"DINOViT": dino_collate_fn, "MAE": mae_collate_fn, "MSN": msn_collate_fn,
                  "SwaV": swav_collate_fn}.get(model, collate_fn)
        return WaferCollateLoader(dataset_train_ssl, batch_size, col_fn, shuffle=True, drop_last=True, seed=seed,
                                  rank=rank, world_size=world)

    version = time.strftime("version_%Y%m%d_%H%M%S")
    out_root = Path(args.out) if args.out else Path(logs_root_dir) / "wafermaps" / version
    results = {}
    for model_name in [m for m in args.models.split(",") if m]:
        runs = []
        for seed in range(args.n_runs):
            np.random.seed(seed)
            torch.manual_seed(seed)
            dataloader_train_ssl = get_data_loader(batch_size, model_name, seed)
            kw = dict(batch_size=args.batch_size, max_epochs=max_epochs)
            if model_name == "DINOViT":
                kw["batch_norm"] = False            # :146-151
            if model_name == "MAE":
                kw["backbone"] = args.mae_backbone
            model = getattr(zoo, model_name)(None, 9, **kw).to(dev)
            log_dir = out_root / model_name
            if rank == 0:
                log_dir.mkdir(parents=True, exist_ok=True)
            trainer = Trainer(max_epochs=max_epochs, limit_train_batches=args.limit_train_batches,
                              log_every_n_steps=args.log_every)
            torch.cuda.reset_peak_memory_stats()
            start = time.time()
            trainer.fit(model, train_dataloaders=dataloader_train_ssl)
            torch.cuda.synchronize()
            end = time.time()
            run = {
                "model": model_name,
                "batch_size": dataloader_train_ssl.batch_size,
                "epochs": max_epochs,
                "params": sum(p.numel() for p in model.parameters() if p.requires_grad) / 1_000_000,
                "runtime": end - start,
                "gpu_memory_usage": torch.cuda.max_memory_allocated() / (1024 ** 3),
                "seed": seed,
                "final_train_loss_ssl": trainer.history[-1]["train_loss_ssl"],
            }
            runs.append(run)
            if rank == 0:
                print(run, flush=True)
                pd.DataFrame(runs).to_csv(log_dir / "results.csv", index=False)
                pd.DataFrame(trainer.loss_log, columns=["step", "loss", "rep_std"]).to_csv(log_dir / "loss_log.csv", index=False)
            del model, trainer
            torch.cuda.empty_cache()
        results[model_name] = runs
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return results


if __name__ == "__main__":
    main()
