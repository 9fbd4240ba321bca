"""nn.Module wrappers (parameters in torch/timm layout, compute in the HIP kernels)."""
from __future__ import annotations

import math

import os

import torch
import torch.nn as nn

from . import ops, vit_ops


class Conv2d(nn.Module):
    """Bias-free convolution; weight float32 [out, in, kh, kw] (state_dict key `weight`)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")  # timm resnet init

    def forward(self, x, stats=None, groups=1, passthrough=False):
        if passthrough:
            return ops.conv2d_passthrough(x, self.weight, self.stride, self.padding, stats=stats, groups=groups)
        return ops.conv2d(x, self.weight, self.stride, self.padding, stats=stats, groups=groups)


class StemConv(Conv2d):
    """ResNet stem: 7x7 stride 2 pad 3 on 3 channels (space-to-depth formulation inside)."""

    def __init__(self, out_channels=64):
        super().__init__(3, out_channels, 7, 2, 3)

    def forward(self, x, stats=None, groups=1):
        return ops.stem_conv(x, self.weight, stats=stats, groups=groups)


class _BatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.affine = num_features, eps, momentum, affine
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:  # constants that stay out of the state_dict, like torch's affine=False
            self.register_buffer("weight", torch.ones(num_features), persistent=False)
            self.register_buffer("bias", torch.zeros(num_features), persistent=False)
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.sync, self.process_group = False, None  # see convert_sync_batchnorm

    def _synced(self) -> bool:
        if not (self.sync and self.training):
            return False
        import torch.distributed as dist

        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1

    def stats_buffer(self, groups: int = 0, which: str = "_stat_buf"):
        """The StatSlots object (ops) of this BatchNorm for sums produced in a convolution's epilogue: the producing
        convolution's forward (batch statistics) or, `which="_stat_buf_bwd"`, the consuming convolution's dgrad
        (backward sums).  Per-tile slots written with plain stores, summed in order by the finalize kernels."""
        if ops.current_branch():   # a second view's pass running beside the first on its own stream: its own slots
            which = f"{which}_b{ops.current_branch()}"
        slots = getattr(self, which, None)
        if slots is None:
            slots = ops.StatSlots(self.num_features)
            setattr(self, which, slots)
        return slots

    def forward(self, x, residual=None, relu=False, stats=None):
        # num_batches_tracked (+1 per forward call = + the number of statistics groups) is incremented inside
        # the statistics kernel: 22 one-block torch kernels per ResNet-18 step otherwise
        if self._synced():
            return ops.sync_batch_norm(x, self.weight, self.bias, self.running_mean, self.running_var, residual=residual,
                                       relu=relu, eps=self.eps, momentum=self.momentum, stats=stats,
                                       num_batches_tracked=self.num_batches_tracked, process_group=self.process_group)
        bwd = None
        if self.training and relu and x.dim() == 4 and torch.is_grad_enabled():
            bwd = self.stats_buffer(ops.current_bn_groups(), "_stat_buf_bwd")
        rm, rv, mom = self._running()
        return ops.batch_norm(x, self.weight, self.bias, rm, rv, self.training,
                              residual=residual, relu=relu, eps=self.eps, momentum=mom, stats=stats,
                              num_batches_tracked=self.num_batches_tracked, bwd_stats=bwd)

    def _running(self):
        """(running_mean, running_var, momentum) this pass updates.  On a side branch (ops.branch: the second view of a
        siamese step running beside the first on its own stream) the launch must not read-modify-write the buffers the main
        branch updates at the same time: it stores its batch statistics (momentum 1) into the module's branch buffers, and
        ViewBranches.merge() folds them into the running statistics behind the join -- in the order of the reference's two
        forward calls (first view, then second)."""
        if ops.current_branch() and self.training:
            br = getattr(self, "_branch_running", None)
            if br is None:
                raise RuntimeError("BatchNorm on a side branch without branch buffers: build nn.ViewBranches(model) first")
            return br[0], br[1], 1.0
        return self.running_mean, self.running_var, self.momentum


class BatchNorm2d(_BatchNorm):
    def forward_relu_maxpool(self, x, stats=None):
        """maxpool3x3s2(relu(self(x))) fused (ResNet stem)."""
        if self._synced():  # the statistics need the exchange between the two halves of the fused kernel pair
            return ops.max_pool3x3s2(self.forward(x, relu=True, stats=stats))
        rm, rv, mom = self._running()
        return ops.bn_relu_maxpool(x, self.weight, self.bias, rm, rv, self.training,
                                   eps=self.eps, momentum=mom, stats=stats,
                                   num_batches_tracked=self.num_batches_tracked)


class BatchNorm1d(_BatchNorm):
    pass


class ViewBranches:
    """The two views of a siamese training step through `root` (a backbone) as two PARALLEL BRANCHES: the first view on
    the caller's stream, the second on a side stream (inside a hipGraph capture: two branches of the graph), joined
    before whatever consumes both outputs.  Both views used to go through every kernel as one batch with per-view
    BatchNorm statistics; as branches, one view's HBM-bound BatchNorm / pooling passes meet the other view's MFMA-bound
    convolutions on the chip (tools/probes/overlap_probe.py: a convolution and a BatchNorm pass of equal length on two
    streams take 0.76 of their sum).  Same arithmetic per view.  What a module caches for "its" launches is kept per
    branch (statistics slots: stats_buffer; scratch: per stream), parameter gradients of the side branch join the pass's
    ordered fold instead of adding into the slots directly (ops._BatchNorm.backward), and the side branch's batch
    statistics reach the running statistics through merge().

    The BatchNorm layers' running statistics are moved into ONE flat buffer (the modules' buffers become views of it),
    with a second flat buffer of the same layout for the side branch's batch statistics: merge() is one launch."""

    _ALIGN = 64

    def __init__(self, root: nn.Module):
        bns = [m for m in root.modules() if isinstance(m, _BatchNorm) and m.running_mean is not None]
        if not bns:
            raise ValueError("ViewBranches: no BatchNorm layers")
        moms = {m.momentum for m in bns}
        if len(moms) != 1 or None in moms:
            raise ValueError("ViewBranches: the BatchNorm layers must share one momentum")
        self.momentum = float(moms.pop())
        dev = bns[0].running_mean.device
        offs, total = [], 0
        for m in bns:
            for _ in range(2):
                offs.append(total)
                total += (m.num_features + self._ALIGN - 1) // self._ALIGN * self._ALIGN
        self.real = torch.zeros(total, dtype=torch.float32, device=dev)
        self.tmp = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for i, m in enumerate(bns):
                c = m.num_features
                om, ov = offs[2 * i], offs[2 * i + 1]
                self.real[om:om + c].copy_(m.running_mean)
                self.real[ov:ov + c].copy_(m.running_var)
                m.running_mean.data = self.real[om:om + c]
                m.running_var.data = self.real[ov:ov + c]
                m._branch_running = (self.tmp[om:om + c], self.tmp[ov:ov + c])
        self.n = total
        self.side = torch.cuda.Stream(device=dev)
        self.modules = bns
        self.convs = [m for m in root.modules() if isinstance(m, Conv2d)]

    def prepare(self) -> None:
        """Before the fork, on the main stream: every bf16 kernel layout of the root's weights that is rebuilt lazily at
        its first use after an update (the stem's space-to-depth form; everything, in the very first step).  Left to the
        branches, one would launch the rebuild on its stream and the other read the buffer without waiting for it."""
        for m in self.convs:
            if isinstance(m, StemConv):
                ops._WCACHE.get(m.weight, kind="stem")
            else:
                ops._WCACHE.get(m.weight, need_crsk=True)

    def __deepcopy__(self, memo):
        # a copy of the model gets no branch state (streams are not copyable; the copy's BatchNorm buffers are its own
        # tensors): it builds its own at its first branched step
        return None

    def valid(self) -> bool:
        """The modules' buffers are still the views made here (a .to() / load of new tensors would replace them)."""
        m = self.modules[0]
        return m.running_mean.data_ptr() == self.real.data_ptr() and m.running_mean.device == self.real.device

    def merge(self) -> None:
        """running <- (1 - momentum) * running + momentum * (side branch's batch statistics), all layers in one launch;
        call on the main stream behind the join."""
        from . import _lib
        from ._lib import check, stream_ptr

        check(_lib.load().wm_ema_update(self.real.data_ptr(), self.tmp.data_ptr(), self.n, 1.0 - self.momentum, stream_ptr()),
              "wm_ema_update(bn running statistics)")


def convert_sync_batchnorm(module: nn.Module, process_group=None) -> nn.Module:
    """torch.nn.SyncBatchNorm.convert_sync_batchnorm for this package's BatchNorm layers (what Lightning's
    `Trainer(sync_batchnorm=True)` does in the reference, scripts/WM811k_benchmark.py:1103): in training mode, with a
    process group of more than one rank, every BatchNorm takes its batch statistics over all ranks.  In place; the
    module is returned for chaining.  State dict and eval mode are unchanged."""
    for m in module.modules():
        if isinstance(m, _BatchNorm):
            m.sync, m.process_group = True, process_group
    return module


class Linear(nn.Module):
    """nn.Linear: weight float32 [out, in], optional bias; x bf16 [rows, in].  The bias, an optional
    GELU and an optional residual add run as one epilogue pass after the GEMM kernel."""

    def __init__(self, in_features, out_features, bias=False):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))  # nn.Linear default
        if bias:
            bound = 1 / math.sqrt(in_features)
            self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)

    def forward(self, x, act: int = vit_ops.ACT_NONE, residual=None):
        if self.out_features % 64:
            # classifier heads (9 / 8 outputs): the GEMM kernel works on 64-column tiles, so the weight is
            # zero-padded to the next multiple and the logits sliced back (gradients flow through both)
            if residual is not None:
                raise NotImplementedError("Linear: residual with out_features % 64 != 0")
            pad = 64 - self.out_features % 64
            w = torch.cat([self.weight, self.weight.new_zeros(pad, self.in_features)], dim=0)
            b = None if self.bias is None else torch.cat([self.bias, self.bias.new_zeros(pad)])
            y = ops.linear(x, w) if b is None and act == vit_ops.ACT_NONE else vit_ops.linear(x, w, b, act, None)
            return y[:, : self.out_features]
        if self.bias is None and act == vit_ops.ACT_NONE and residual is None:
            return ops.linear(x, self.weight)
        return vit_ops.linear(x, self.weight, self.bias, act, residual)


class LayerNorm(nn.Module):
    def __init__(self, normalized_shape, eps=1e-5):
        super().__init__()
        self.normalized_shape, self.eps = int(normalized_shape), eps
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))

    def forward(self, x):
        return vit_ops.layer_norm(x, self.weight, self.bias, self.eps)

    def forward_skip(self, x):
        """(LayerNorm(x), x) as one autograd node: hand the second value to the branch's residual add, and the
        gradients of both uses of x are combined inside the LayerNorm backward launch."""
        if os.environ.get("WM_LN_SKIP", "1") == "0":  # A/B switch: two autograd nodes, gradients added by autograd
            return vit_ops.layer_norm(x, self.weight, self.bias, self.eps), x
        return vit_ops.layer_norm_skip(x, self.weight, self.bias, self.eps)


class GELU(nn.Module):
    """Marker module: exact (erf) GELU, fused into the preceding Linear's epilogue pass or applied
    by ProjectionHead after a BatchNorm."""

    def forward(self, x):
        return vit_ops.bias_act(x, None, vit_ops.ACT_GELU)


class ReLU(nn.Module):
    """Marker module: the activation itself is fused into the preceding BatchNorm kernel."""

    def forward(self, x):
        raise RuntimeError("ReLU is fused into the preceding batch-norm launch; do not call it directly")
