"""Projection heads with lightly's constructor signatures and `layers.N.*` state_dict keys
(lightly.models.modules.heads; reference use scripts/WM811k_benchmark.py:233)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import functional as F_hip
from . import nn as hnn
from . import ops, vit_ops
from . import precision as _precision


class ProjectionHead(nn.Module):
    """blocks: (in, out, batch_norm module or None, activation module or None); Linear has a bias
    only without batch norm, as in lightly."""

    def __init__(self, blocks: List[Tuple[int, int, Optional[nn.Module], Optional[nn.Module]]]):
        super().__init__()
        layers = []
        self._plan = []
        for in_dim, out_dim, bn, act in blocks:
            lin = hnn.Linear(in_dim, out_dim, bias=not bool(bn))
            layers.append(lin)
            i_lin = len(layers) - 1
            i_bn = None
            if bn:
                layers.append(bn)
                i_bn = len(layers) - 1
            if act:
                layers.append(act)
            self._plan.append((i_lin, i_bn, act))
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        for i_lin, i_bn, act in self._plan:
            lin = self.layers[i_lin]
            if i_bn is not None:
                x = lin(x)
                if isinstance(act, hnn.GELU):
                    x = act(self.layers[i_bn](x, relu=False))
                else:
                    x = self.layers[i_bn](x, relu=act is not None)
            elif isinstance(act, hnn.GELU):
                x = lin(x, act=vit_ops.ACT_GELU)
            elif isinstance(act, hnn.ReLU):
                x = lin(x, act=vit_ops.ACT_RELU)
            elif act is not None:
                raise NotImplementedError(f"ProjectionHead: activation {type(act).__name__} has no HIP path")
            else:
                x = lin(x)
        return x


class SimCLRProjectionHead(ProjectionHead):
    """lightly SimCLRProjectionHead(input_dim, hidden_dim, output_dim, num_layers=2, batch_norm=True):
    Linear-BN-ReLU (x num_layers-1), Linear-BN  (the SimCLR v2 form, lightly's default since 1.2; the reference calls
    it with the defaults, scripts/WM811k_benchmark.py:233).  batch_norm=False is the v1 form: the Linear layers carry
    a bias and the ReLU rides in the GEMM epilogue."""

    def __init__(self, input_dim: int = 2048, hidden_dim: int = 2048, output_dim: int = 128, num_layers: int = 2,
                 batch_norm: bool = True):
        def bn(dim):
            return hnn.BatchNorm1d(dim) if batch_norm else None

        blocks = [(input_dim, hidden_dim, bn(hidden_dim), hnn.ReLU())]
        for _ in range(2, num_layers):
            blocks.append((hidden_dim, hidden_dim, bn(hidden_dim), hnn.ReLU()))
        blocks.append((hidden_dim, output_dim, bn(output_dim), None))
        super().__init__(blocks)


class MoCoProjectionHead(ProjectionHead):
    """lightly MoCoProjectionHead: Linear(+bias)-ReLU, Linear(+bias)  (reference: heads.MoCoProjectionHead(512, 2048,
    128), scripts/WM811k_benchmark.py:298)."""

    def __init__(self, input_dim: int = 2048, hidden_dim: int = 2048, output_dim: int = 128):
        super().__init__([(input_dim, hidden_dim, None, hnn.ReLU()), (hidden_dim, output_dim, None, None)])


class MSNProjectionHead(ProjectionHead):
    """lightly MSNProjectionHead(input_dim=768, hidden_dim=2048, output_dim=256): (Linear-BN-GELU) x2, Linear(+bias)
    (reference: MSNProjectionHead(384), scripts/WM811k_benchmark.py:678)."""

    def __init__(self, input_dim: int = 768, hidden_dim: int = 2048, output_dim: int = 256):
        super().__init__([(input_dim, hidden_dim, hnn.BatchNorm1d(hidden_dim), hnn.GELU()),
                          (hidden_dim, hidden_dim, hnn.BatchNorm1d(hidden_dim), hnn.GELU()),
                          (hidden_dim, output_dim, None, None)])


class SwaVProjectionHead(ProjectionHead):
    """lightly SwaVProjectionHead: Linear-BN-ReLU, Linear(+bias)  (reference: (512, 2048, 128), :830)."""

    def __init__(self, input_dim: int = 2048, hidden_dim: int = 2048, output_dim: int = 128):
        super().__init__([(input_dim, hidden_dim, hnn.BatchNorm1d(hidden_dim), hnn.ReLU()),
                          (hidden_dim, output_dim, None, None)])


class SwaVPrototypes(nn.Module):
    """lightly SwaVPrototypes(input_dim, n_prototypes): a bias-free Linear (state_dict key `layers.weight`) whose
    rows `normalize()` puts on the unit sphere (reference: (128, 3000), :831, :846)."""

    def __init__(self, input_dim: int = 128, n_prototypes: int = 3000):
        super().__init__()
        self.layers = hnn.Linear(input_dim, n_prototypes, bias=False)

    def forward(self, x):
        return self.layers(x)

    @torch.no_grad()
    def normalize(self) -> None:
        w = self.layers.weight
        w.copy_(F_hip.l2_normalize(w.detach().contiguous()))
        ops.bump_weight_epoch()


class BarlowTwinsProjectionHead(ProjectionHead):
    """lightly BarlowTwinsProjectionHead: (Linear-BN-ReLU) x2, Linear(+bias)  (reference: (512, 2048, 2048), :363, :400)."""

    def __init__(self, input_dim: int = 2048, hidden_dim: int = 8192, output_dim: int = 8192):
        super().__init__([(input_dim, hidden_dim, hnn.BatchNorm1d(hidden_dim), hnn.ReLU()),
                          (hidden_dim, hidden_dim, hnn.BatchNorm1d(hidden_dim), hnn.ReLU()),
                          (hidden_dim, output_dim, None, None)])


class BYOLProjectionHead(ProjectionHead):
    """lightly BYOLProjectionHead: Linear-BN-ReLU, Linear(+bias) (reference: (512, 4096, 256), :437)."""

    def __init__(self, input_dim: int = 2048, hidden_dim: int = 4096, output_dim: int = 256):
        super().__init__([(input_dim, hidden_dim, hnn.BatchNorm1d(hidden_dim), hnn.ReLU()),
                          (hidden_dim, output_dim, None, None)])


class BYOLPredictionHead(BYOLProjectionHead):
    """lightly BYOLPredictionHead: same form (reference: (256, 4096, 256), :438)."""


class SimSiamProjectionHead(ProjectionHead):
    """lightly SimSiamProjectionHead: (Linear-BN-ReLU) x2, Linear-BN(affine=False) (reference: (512, 2048, 2048), :611)."""

    def __init__(self, input_dim: int = 2048, hidden_dim: int = 2048, output_dim: int = 2048):
        super().__init__([(input_dim, hidden_dim, hnn.BatchNorm1d(hidden_dim), hnn.ReLU()),
                          (hidden_dim, hidden_dim, hnn.BatchNorm1d(hidden_dim), hnn.ReLU()),
                          (hidden_dim, output_dim, hnn.BatchNorm1d(output_dim, affine=False), None)])


class SimSiamPredictionHead(ProjectionHead):
    """lightly SimSiamPredictionHead: Linear-BN-ReLU, Linear(+bias) (reference: (2048, 512, 2048), :612)."""

    def __init__(self, input_dim: int = 2048, hidden_dim: int = 512, output_dim: int = 2048):
        super().__init__([(input_dim, hidden_dim, hnn.BatchNorm1d(hidden_dim), hnn.ReLU()),
                          (hidden_dim, output_dim, None, None)])


class DINOProjectionHead(ProjectionHead):
    """lightly.models.modules.heads.DINOProjectionHead (reference: scripts/WM811k_benchmark.py:553-559,
    MixedWM38_pretrain.py:146-152): Linear-[BN]-GELU x2, Linear -> bottleneck, L2-normalise,
    weight-normalised bias-free Linear with the gain frozen at 1.  As in lightly, ONE BatchNorm1d
    instance serves both hidden blocks (state_dict keys layers.1.* and layers.4.* alias)."""

    def __init__(self, input_dim: int = 2048, hidden_dim: int = 2048, bottleneck_dim: int = 256, output_dim: int = 65536,
                 batch_norm: bool = False, freeze_last_layer: int = -1, norm_last_layer: bool = True):
        bn = hnn.BatchNorm1d(hidden_dim) if batch_norm else None
        super().__init__([(input_dim, hidden_dim, bn, hnn.GELU()), (hidden_dim, hidden_dim, bn, hnn.GELU()),
                          (hidden_dim, bottleneck_dim, None, None)])
        for m in self.modules():
            if isinstance(m, hnn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        self.freeze_last_layer = freeze_last_layer
        self.last_layer = _WeightNormLinear(bottleneck_dim, output_dim, train_gain=not norm_last_layer)

    def cancel_last_layer_gradients(self, current_epoch: int) -> None:
        if current_epoch >= self.freeze_last_layer:
            return
        for p in self.last_layer.parameters():
            if p.grad is not None:
                # in place: with the fused optimisers p.grad is a view of the flat gradient arena, and dropping
                # the view would detach the parameter from the arena-based step and the all-reduce for good
                p.grad.zero_()

    def forward(self, x):
        x = super().forward(x)
        # the normalised bottleneck feeds the last layer's GEMM: bf16 rows with a bf16 backward (no framework cast passes
        # around the GEMM); float32 throughout under the float32 preset
        x = F_hip.l2_normalize(x, differentiable_bf16=not _precision.is_f32())
        return self.last_layer(x)


class _ColScale(torch.autograd.Function):
    """y = x * g (per output column): the gain of a weight-normalised Linear applied to the product with the
    unit-norm rows (wm_colscale_fwd / wm_colscale_bwd)."""

    @staticmethod
    def forward(ctx, x, g):
        from . import _lib
        from ._lib import check, ptr, stream_ptr

        x = x.to(torch.bfloat16).contiguous()
        rows, c = x.shape
        gf = g.detach().reshape(-1).float().contiguous()
        y = torch.empty_like(x)
        check(_lib.load().wm_colscale_fwd(ptr(x), ptr(gf), rows, c, ptr(y), stream_ptr()), "wm_colscale_fwd")
        ctx.save_for_backward(x, gf)
        ctx.gshape = g.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        from ._lib import check, ptr, stream_ptr

        x, gf = ctx.saved_tensors
        rows, c = x.shape
        dy = dy.to(torch.bfloat16).contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty(c, dtype=torch.float32, device=x.device)
        check(_lib.load().wm_colscale_bwd(ptr(x), ptr(gf), ptr(dy), rows, c, ptr(dx), ptr(dg), 0, stream_ptr()),
              "wm_colscale_bwd")
        return dx, dg.view(ctx.gshape)


class _WeightNormLinear(nn.Module):
    """torch.nn.utils.weight_norm(nn.Linear(in, out, bias=False)) with parameters weight_g [out, 1]
    (filled with 1) and weight_v [out, in]: W = g * v / ||v||_row.  norm_last_layer=True (lightly's default, the
    reference's call) freezes g at 1; with a trainable gain the product with the unit rows is scaled per column."""

    def __init__(self, in_features: int, out_features: int, train_gain: bool = False):
        super().__init__()
        v = torch.empty(out_features, in_features)
        nn.init.kaiming_uniform_(v, a=5 ** 0.5)
        self.weight_g = nn.Parameter(torch.ones(out_features, 1), requires_grad=train_gain)
        self.weight_v = nn.Parameter(v)
        self.train_gain = train_gain

    def forward(self, x):
        w = F_hip.l2_normalize(self.weight_v, eps=0.0)
        y = ops.linear(x, w)
        if self.train_gain:
            y = _ColScale.apply(y, self.weight_g)
        return y
