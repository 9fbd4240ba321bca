"""A minimal stand-in for the slice of pytorch_lightning.Trainer the reference uses
(scripts/WM811k_benchmark.py:1097-1114, scripts/MixedWM38_pretrain.py:611-627): fit(model, train_loader,
val_dataloaders) with per-epoch kNN validation, epoch-stepped LR schedulers and data-parallel gradient
averaging (one process per GPU; the replicas are aligned from rank 0 before the first step).

`use_graph=True` replays the step of models that declare `graph_safe = True` (no host-side scalar that
changes from step to step inside training_step) from a captured hipGraph (graph.GraphedTrainStep): the
loader then only draws indices and augmentation decisions; images are produced inside the replay."""
from __future__ import annotations

import time
from typing import Optional

import torch

from . import distributed as wdist


class Trainer:
    def __init__(self, max_epochs: int = 1, log_every_n_steps: int = 50, limit_train_batches: Optional[int] = None,
                 verbose: bool = True, use_graph: bool = False, overlap_grad_sync: bool = True,
                 sync_batchnorm: bool = False):
        self.max_epochs = max_epochs
        self.log_every_n_steps = log_every_n_steps
        self.limit_train_batches = limit_train_batches
        self.verbose = verbose and wdist.rank() == 0
        self.use_graph = use_graph
        self.overlap_grad_sync = overlap_grad_sync
        self.sync_batchnorm = sync_batchnorm  # the reference's module-level flag (scripts/WM811k_benchmark.py:62,1103)
        self.global_step = 0
        self.current_epoch = 0
        self.history = []
        self.loss_log = []   # (global_step, loss) every log_every_n_steps, like the reference's TensorBoard scalars

    def _graphed(self, model, opt, loader, sync):
        """GraphedTrainStep for this (model, loader) or None when the step cannot be captured."""
        if not (self.use_graph and getattr(model, "graph_safe", False) and hasattr(loader, "iter_indices")):
            return None
        if self.sync_batchnorm and wdist.world_size() > 1:
            return None  # a statistics exchange inside every BatchNorm: the step runs eagerly
        if not getattr(loader, "drop_last", False) and len(loader.dataset) % (loader.batch_size * getattr(loader, "world_size", 1)):
            # a captured step has ONE batch shape: a short final batch cannot be replayed
            raise ValueError("Trainer(use_graph=True) needs drop_last=True on the training loader (or a dataset size "
                             "divisible by the global batch): the captured step replays one batch shape")
        from .graph import GraphedTrainStep

        idx, rng = next(iter(loader.iter_indices()))
        fmt = "s2d_bf16" if getattr(model, "stem_takes_s2d", False) else loader.fmt
        stages = getattr(model, "backward_stages", None) if (self.overlap_grad_sync and wdist.world_size() > 1) else None
        return GraphedTrainStep(model, opt, loader.dataset, loader.batch_size, fmt=fmt, stages=stages).capture(idx, rng, sync)

    def fit(self, model, train_dataloaders, val_dataloaders=None):
        opts, scheds = model.configure_optimizers()
        opt = opts[0]
        if self.sync_batchnorm:
            from .nn import convert_sync_batchnorm

            convert_sync_batchnorm(model)
        sync = wdist.GradSync(opt)
        wdist.broadcast_state(model, opt)  # replicas start from rank 0's weights, buffers, teachers and banks
        if wdist.world_size() > 1 and not getattr(train_dataloaders, "drop_last", True):
            raise ValueError("data-parallel fit needs drop_last=True on the training loader: every rank must run "
                             "the same number of equally sized steps (the reference's loaders use drop_last=True)")
        graphed = None
        for epoch in range(self.max_epochs):
            self.current_epoch = model.current_epoch = epoch
            model.train()
            if hasattr(train_dataloaders, "set_epoch"):
                train_dataloaders.set_epoch(epoch)
            t0 = time.time()
            if epoch == 0:
                graphed = self._graphed(model, opt, train_dataloaders, sync)
            batches = train_dataloaders.iter_indices() if graphed is not None else train_dataloaders
            for bi, batch in enumerate(batches):
                if self.limit_train_batches is not None and bi >= self.limit_train_batches:
                    break
                if graphed is not None:
                    loss = graphed.step(batch[0], batch[1], sync)
                else:
                    opt.zero_grad()
                    loss = model.training_step(batch, bi)
                    loss.backward()
                    sync.start()
                    sync.wait()
                    opt.step()
                self.global_step += 1
                if self.global_step % self.log_every_n_steps == 0:
                    lv = float(loss.detach())
                    rs = model.logged.get("rep_std") if hasattr(model, "logged") else None
                    self.loss_log.append((self.global_step - 1, lv, None if rs is None else float(rs)))
                    if self.verbose:
                        print(f"epoch {epoch} step {self.global_step} loss {lv:.4f}", flush=True)
            for s in scheds:
                s.step()
            torch.cuda.synchronize()
            rec = {"epoch": epoch, "train_time_s": time.time() - t0,
                   "train_loss_ssl": float(model.logged.get("train_loss_ssl", model.logged.get("train_loss", float("nan"))))}
            if val_dataloaders is not None:
                tv = time.time()
                self.validate(model, val_dataloaders)
                torch.cuda.synchronize()
                rec["val_time_s"] = time.time() - tv
                rec["knn_accuracy"] = model.logged.get("knn_accuracy")
                rec["knn_f1"] = model.logged.get("knn_f1")
            self.history.append(rec)
            if self.verbose:
                print(rec, flush=True)
        return self.history

    @torch.no_grad()
    def validate(self, model, val_dataloaders):
        wdist.sync_bn_buffers(model)  # rank 0's running statistics everywhere, as DDP's broadcast_buffers
        model.eval()
        model.on_validation_epoch_start()
        # data parallel: a model that can shard its evaluation hands each rank its share of the batches
        batches = model.shard_eval_batches(val_dataloaders) if hasattr(model, "shard_eval_batches") else enumerate(val_dataloaders)
        for bi, batch in batches:
            model.validation_step(batch, bi)
        model.on_validation_epoch_end()
        model.train()
