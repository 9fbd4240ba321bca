"""A minimal stand-in for the slice of pytorch_lightning.Trainer the reference uses
(scripts/WM811k_benchmark.py:1097-1114): fit(model, train_loader, val_dataloaders) with per-epoch
kNN validation, epoch-stepped LR schedulers and optional data-parallel gradient averaging."""
from __future__ import annotations

import time
from typing import Optional

import torch

from . import distributed as wdist


class Trainer:
    def __init__(self, max_epochs: int = 1, log_every_n_steps: int = 50, limit_train_batches: Optional[int] = None,
                 verbose: bool = True):
        self.max_epochs = max_epochs
        self.log_every_n_steps = log_every_n_steps
        self.limit_train_batches = limit_train_batches
        self.verbose = verbose and wdist.rank() == 0
        self.global_step = 0
        self.history = []

    def fit(self, model, train_dataloaders, val_dataloaders=None):
        opts, scheds = model.configure_optimizers()
        opt = opts[0]
        sync = wdist.GradSync(opt)
        for epoch in range(self.max_epochs):
            model.current_epoch = epoch
            model.train()
            if hasattr(train_dataloaders, "set_epoch"):
                train_dataloaders.set_epoch(epoch)
            t0 = time.time()
            for bi, batch in enumerate(train_dataloaders):
                if self.limit_train_batches is not None and bi >= self.limit_train_batches:
                    break
                opt.zero_grad()
                loss = model.training_step(batch, bi)
                loss.backward()
                sync.start()
                sync.wait()
                opt.step()
                self.global_step += 1
                if self.verbose and self.global_step % self.log_every_n_steps == 0:
                    print(f"epoch {epoch} step {self.global_step} loss {loss.item():.4f}", flush=True)
            for s in scheds:
                s.step()
            rec = {"epoch": epoch, "train_time_s": time.time() - t0,
                   "train_loss_ssl": float(model.logged.get("train_loss_ssl", float("nan")))}
            if val_dataloaders is not None:
                self.validate(model, val_dataloaders)
                rec["knn_accuracy"] = model.logged.get("knn_accuracy")
                rec["knn_f1"] = model.logged.get("knn_f1")
            self.history.append(rec)
            if self.verbose:
                print(rec, flush=True)
        return self.history

    @torch.no_grad()
    def validate(self, model, val_dataloaders):
        model.eval()
        model.on_validation_epoch_start()
        for bi, batch in enumerate(val_dataloaders):
            model.validation_step(batch, bi)
        model.on_validation_epoch_end()
        model.train()
