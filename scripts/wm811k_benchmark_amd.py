#!/usr/bin/env python3
"""WM-811K self-supervised benchmark on MI355X: the reference's scripts/WM811k_benchmark.py main()
(:1033-1187) on the HIP path, step for step.

    python scripts/wm811k_benchmark_amd.py [--full] [--data-root /path/to/reference/data] [--models SimCLR,MoCo]
                                           [--max-epochs N] [--batch-size 64] [--out DIR] [--limit-train-batches N]
    torchrun --nproc-per-node N scripts/wm811k_benchmark_amd.py ...      (data parallel, reference `distributed=True`)

What the reference does, and where it is here:
  :87-104   dummy mode: train_20_split.pkl.xz, stratified 80/20 train_test_split(random_state=42); full mode:
            train_data / val_data pickles                                   -> load_data()
  :107-108  kNN bank / test datasets with the inference transform           -> dataset_train_kNN, dataset_test
  :113-157  per-model SSL transform                                          -> create_dataset_train_ssl()
  :160-195  three DataLoaders (SSL shuffle + drop_last; kNN / test in order) -> get_data_loaders()
  :1033-1150 for each model, for each seed: seed, loaders, model, fit with per-epoch kNN validation, run record,
            confusion_matrix.npz, results.csv                                -> main()
  :1152-1187 results table                                                   -> print_table()
Not carried over: TensorBoard logger, ModelCheckpoint (out of scope); AMP flag (the HIP
path computes in bf16 with float32 master weights throughout).

Data: `--data-root` = the reference's `data/` directory (reads the *.pkl.xz files through data/ingest.py).
Without it the dummy mode runs from the data-only fixture tests/golden/wm811k_train_20_split.npz (the same 12 449
wafers, converted by tests/golden/make_reference_data.py), so the script runs on a box without the reference.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

logs_root_dir = "benchmark_logs"
memory_bank_size = 4096
knn_k = 5      # reference: "sweep of knn_k values leads to best performance at k=5"
knn_t = 0.1
classes = 9


def load_data(dummy: bool, data_root: str | None):
    """-> (store_train, y_train, store_val, y_val): host-side WaferStores + int64 label arrays."""
    from sklearn.model_selection import train_test_split

    from ssl_wafermap_amd.data.ingest import read_wafer_pickle
    from ssl_wafermap_amd.data.store import WaferStore

    if dummy:
        if data_root:
            store, y = read_wafer_pickle(Path(data_root) / "processed/WM811K/train_20_split.pkl.xz")
        else:
            store, y = WaferStore.load(ROOT / "tests/golden/wm811k_train_20_split.npz")
        y = np.asarray(y).astype(np.int64)
        # the reference splits the pandas Series; splitting the positions with the same arguments selects the
        # same wafers in the same order (sklearn permutes positions, not values)
        i_train, i_val = train_test_split(np.arange(len(store)), test_size=0.2, random_state=42, stratify=y)
        return store.subset(i_train), y[i_train], store.subset(i_val), y[i_val]
    if not data_root:
        raise SystemExit("--full needs --data-root (train_data.pkl.xz / val_data.pkl.xz are not shipped as fixtures)")
    st, yt = read_wafer_pickle(Path(data_root) / "processed/WM811K/train_data.pkl.xz")
    sv, yv = read_wafer_pickle(Path(data_root) / "processed/WM811K/val_data.pkl.xz")
    return st, np.asarray(yt).astype(np.int64), sv, np.asarray(yv).astype(np.int64)


def seed_everything(seed: int):
    import random

    import torch

    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


MODEL_NAMES = ["SimCLR", "MoCo", "DCLW", "SwaV", "BYOL", "SimSiam", "FastSiam", "DINO", "DINOViT", "VICReg",
               "BarlowTwins", "MSN", "PMSN", "MAE", "SimMIM", "SupervisedR18"]   # reference order, :1034-1059


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="reference `dummy = False`: full train/val pickles, 150 epochs")
    ap.add_argument("--data-root", default=None)
    ap.add_argument("--models", default=",".join(MODEL_NAMES))
    ap.add_argument("--max-epochs", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--n-runs", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--limit-train-batches", type=int, default=None)
    ap.add_argument("--graph", action="store_true", help="replay the SimCLR-family step from a captured hipGraph")
    ap.add_argument("--log-every", type=int, default=50, help="Lightning's log_every_n_steps (loss / rep_std rows)")
    args = ap.parse_args(argv)

    import pandas as pd
    import torch

    import ssl_wafermap_amd.models as zoo
    from ssl_wafermap_amd import distributed as wdist
    from ssl_wafermap_amd.data import WaferLoader, WaferMapDataset
    from ssl_wafermap_amd.trainer import Trainer
    from ssl_wafermap_amd.transforms import BaseViewTransform, InferenceTransform, MultiCropTransform

    dummy = not args.full
    max_epochs = args.max_epochs if args.max_epochs is not None else (2 if dummy else 150)
    rank, world, local = wdist.init_from_env()
    distributed = world > 1
    batch_size = args.batch_size // world if distributed else args.batch_size   # reference :78-81
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    st_train, y_train, st_val, y_val = load_data(dummy, args.data_root)
    st_train.to(dev)
    st_val.to(dev)
    dataset_train_kNN = WaferMapDataset(st_train, y_train, InferenceTransform(), device=dev)
    dataset_test = WaferMapDataset(st_val, y_val, InferenceTransform(), device=dev)
    base_transform = BaseViewTransform()
    fastsiam_transform = BaseViewTransform(n_views=4)
    mae_transform = BaseViewTransform(n_views=1)
    multicrop_transform = MultiCropTransform()
    model_to_transform = {
        "BarlowTwins": base_transform, "BYOL": base_transform, "DCLW": base_transform, "DINO": multicrop_transform,
        "DINOViT": multicrop_transform, "FastSiam": fastsiam_transform, "MAE": mae_transform, "MoCo": base_transform,
        "MSN": multicrop_transform, "PMSN": multicrop_transform, "SimCLR": base_transform, "SimMIM": mae_transform,
        "SimSiam": base_transform, "SwaV": multicrop_transform, "VICReg": base_transform,
        "SupervisedR18": mae_transform,  # the single transform composition (one view)
    }

    def create_dataset_train_ssl(name):
        return WaferMapDataset(st_train, y_train, transform=model_to_transform[name], device=dev)

    def get_data_loaders(batch_size, dataset_train_ssl, seed):
        train = WaferLoader(dataset_train_ssl, batch_size, shuffle=True, drop_last=True, seed=seed, rank=rank,
                            world_size=world)
        knn = WaferLoader(dataset_train_kNN, args.batch_size, shuffle=False, drop_last=False)
        test = WaferLoader(dataset_test, args.batch_size, shuffle=False, drop_last=False)
        return train, knn, test

    version = time.strftime("version_%Y%m%d_%H%M%S")
    out_root = Path(args.out) if args.out else Path(logs_root_dir) / "wafermaps" / version
    bench_results = {}
    for model_name in [m for m in args.models.split(",") if m]:
        Benchmark = getattr(zoo, model_name)
        runs = []
        for seed in range(args.n_runs):
            seed_everything(seed)
            dataset_train_ssl = create_dataset_train_ssl(model_name)
            dataloader_train_ssl, dataloader_train_kNN, dataloader_test = get_data_loaders(batch_size, dataset_train_ssl, seed)
            # batch_size -> the reference's module-level lr_factor = batch_size / 256 (:71, the GLOBAL batch);
            # max_epochs -> the cosine schedules' horizon (module-level in the reference)
            kw = dict(knn_k=knn_k, knn_t=knn_t, batch_size=args.batch_size, max_epochs=max_epochs)
            benchmark_model = Benchmark(dataloader_train_kNN, classes, **kw).to(dev)
            log_dir = out_root / (model_name if args.n_runs <= 1 else f"{model_name}/run{seed}")
            if rank == 0:
                log_dir.mkdir(parents=True, exist_ok=True)
            trainer = Trainer(max_epochs=max_epochs, limit_train_batches=args.limit_train_batches, use_graph=args.graph,
                              log_every_n_steps=args.log_every)
            torch.cuda.reset_peak_memory_stats()
            start = time.time()
            trainer.fit(benchmark_model, train_dataloaders=dataloader_train_ssl, val_dataloaders=dataloader_test)
            torch.cuda.synchronize()
            end = time.time()
            run = {
                "model": model_name,
                "batch_size": dataloader_train_ssl.batch_size,
                "epochs": max_epochs,
                "params": sum(p.numel() for p in benchmark_model.parameters() if p.requires_grad) / 1_000_000,
                "max_accuracy": benchmark_model.max_accuracy,
                "max_f1": benchmark_model.max_f1,
                "runtime": end - start,
                "gpu_memory_usage": torch.cuda.max_memory_allocated() / (1024 ** 3),
                "seed": seed,
            }
            runs.append(run)
            if rank == 0:
                print(run, flush=True)
                np.savez_compressed(log_dir / "confusion_matrix.npz",
                                    confusion_matrix=np.stack(benchmark_model.confusion_matrix))
                pd.DataFrame(runs).to_csv(log_dir / "results.csv", index=False)
                pd.DataFrame(trainer.history).to_csv(log_dir / "history.csv", index=False)
                # the reference logs these scalars to TensorBoard (train_loss_ssl, rep_std at log_every_n_steps)
                pd.DataFrame(trainer.loss_log, columns=["step", "loss", "rep_std"]).to_csv(log_dir / "loss_log.csv", index=False)
            del benchmark_model, trainer
            torch.cuda.empty_cache()
        bench_results[model_name] = runs
    if rank == 0:
        print_table(bench_results, args.batch_size)
    if distributed:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return bench_results


def print_table(bench_results, batch_size):
    header = (f"| {'':<13} | {'Batch Size':>10} | {'Epochs':>6} | {'#param.':>9} "
              f"| {'KNN Test Accuracy':>18} | {'KNN Test F1':>16} | {'Time':>10} | {'Peak GPU Usage':>14} |")
    print("-" * len(header))
    print(header)
    print("-" * len(header))
    for model, results in bench_results.items():
        runtime = np.array([r["runtime"] for r in results]).mean() / 60
        accuracy = np.array([r["max_accuracy"] for r in results])
        f1 = np.array([r["max_f1"] for r in results])
        mem = np.array([r["gpu_memory_usage"] for r in results]).max()
        epochs = int(np.array([r["epochs"] for r in results]).mean())
        params = results[0]["params"]
        acc_msg = f"{accuracy.mean():>8.3f} +- {accuracy.std():>4.3f}" if len(accuracy) > 1 else f"{accuracy.mean():>18.3f}"
        f1_msg = f"{f1.mean():>8.3f} +- {f1.std():>4.3f}" if len(f1) > 1 else f"{f1.mean():>16.3f}"
        print(f"| {model:<13} | {batch_size:>10} | {epochs:>6} | {params:>8.1f}M | {acc_msg} | {f1_msg} "
              f"| {runtime:>6.1f} Min | {mem:>8.1f} GByte |", flush=True)
    print("-" * len(header))


if __name__ == "__main__":
    main()
