"""lightly.utils.scheduler equivalents (host-side LR schedules)."""
import math

import torch


def cosine_warmup_factor(epoch: int, warmup_epochs: int, max_epochs: int) -> float:
    if epoch < warmup_epochs:
        return (epoch + 1) / warmup_epochs
    return 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (max_epochs - warmup_epochs)))


class CosineWarmupScheduler(torch.optim.lr_scheduler.LambdaLR):
    """Linear warm-up then cosine decay, stepped per epoch (reference use:
    scripts/WM811k_benchmark.py:599-601)."""

    def __init__(self, optimizer, warmup_epochs, max_epochs, last_epoch=-1, verbose=False):
        self.warmup_epochs, self.max_epochs = warmup_epochs, max_epochs
        super().__init__(optimizer, lr_lambda=lambda e: cosine_warmup_factor(e, warmup_epochs, max_epochs),
                         last_epoch=last_epoch)
