"""ResNet-18 backbone with timm's module tree and state_dict keys
(`timm.create_model("resnet18", num_classes=0)`, reference scripts/WM811k_benchmark.py:231):
conv1, bn1, layer{1-4}.{0,1}.{conv1,bn1,conv2,bn2}, layer{2-4}.0.downsample.{0,1}; output [N, 512]."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .. import nn as hnn
from .. import ops


def conv_bn(conv, bn, x, relu=False, residual=None):
    """conv -> BatchNorm (+residual) (+ReLU); in training the convolution's epilogue accumulates the
    BN statistics so the tensor is not re-read for them."""
    if bn.training:
        g = ops.current_bn_groups()
        st = bn.stats_buffer(g)
        return bn(conv(x, stats=st, groups=g), residual=residual, relu=relu, stats=st)
    return bn(conv(x), residual=residual, relu=relu)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = hnn.Conv2d(inplanes, planes, 3, stride, 1)
        self.bn1 = hnn.BatchNorm2d(planes)
        self.conv2 = hnn.Conv2d(planes, planes, 3, 1, 1)
        self.bn2 = hnn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(hnn.Conv2d(inplanes, planes, 1, stride, 0), hnn.BatchNorm2d(planes))
        nn.init.zeros_(self.bn2.weight)  # timm zero_init_last

    def forward(self, x):
        if self.downsample is not None and torch.is_grad_enabled() and x.requires_grad:
            # the block input feeds conv1 and the 1x1 downsample: conv1 hands it on, so the downsample's
            # input gradient comes back to conv1's backward and is added in its dgrad epilogue
            g = ops.current_bn_groups()
            st = self.bn1.stats_buffer(g) if self.bn1.training else None
            y, x_alias = self.conv1(x, stats=st, groups=g, passthrough=True)
            out = self.bn1(y, relu=True, stats=st)
            shortcut = conv_bn(self.downsample[0], self.downsample[1], x_alias)
        elif self.downsample is not None:
            out = conv_bn(self.conv1, self.bn1, x, relu=True)
            shortcut = conv_bn(self.downsample[0], self.downsample[1], x)
        elif torch.is_grad_enabled() and x.requires_grad:
            # identity shortcut: conv1 hands its input back as the residual so that the shortcut's
            # gradient is added in conv1's dgrad epilogue (no separate add kernel in the backward)
            g = ops.current_bn_groups()
            st = self.bn1.stats_buffer(g) if self.bn1.training else None
            y, shortcut = self.conv1(x, stats=st, groups=g, passthrough=True)
            out = self.bn1(y, relu=True, stats=st)
        else:
            out = conv_bn(self.conv1, self.bn1, x, relu=True)
            shortcut = x
        return conv_bn(self.conv2, self.bn2, out, relu=True, residual=shortcut)


class ResNet18(nn.Module):
    num_features = 512

    def __init__(self, num_classes: int = 0):
        super().__init__()
        if num_classes != 0:
            raise NotImplementedError("only the headless backbone (num_classes=0) is built")
        self.conv1 = hnn.StemConv(64)
        self.bn1 = hnn.BatchNorm2d(64)
        self.layer1 = nn.Sequential(BasicBlock(64, 64), BasicBlock(64, 64))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256))
        self.layer4 = nn.Sequential(BasicBlock(256, 512, 2), BasicBlock(512, 512))

    def forward_features(self, x):
        # stem: conv (+ fused BN statistics) -> BN + ReLU + max-pool in one pass
        if self.bn1.training:
            g = ops.current_bn_groups()
            st = self.bn1.stats_buffer(g)
            x = self.bn1.forward_relu_maxpool(self.conv1(x, stats=st, groups=g), stats=st)
        else:
            x = self.bn1.forward_relu_maxpool(self.conv1(x))
        x = self.layer1(x)
        x = self.layer2(x)
        # stage boundaries of the backward pass (identity unless graph.GraphedTrainStep records them): layer4 + head
        # hold 3/4 of the gradient bytes and finish first, so their all-reduce hides under layers 3..1
        x = self.layer3(ops.cut_point(x, "layer3"))
        return self.layer4(ops.cut_point(x, "layer4"))

    view_branches = True   # WM_VIEW_BRANCHES=0 (read per call) or this attribute: both statistic groups in one stream

    def forward(self, x):
        if self._branch_ok(x):
            return self._forward_branches(x)
        return ops.global_avg_pool(self.forward_features(x))

    def _branch_ok(self, x) -> bool:
        """A training-mode pass over TWO statistic groups (ops.bn_groups(2): the two views of a siamese step, which the
        reference runs as forward(x0); forward(x1)) goes through the network as two parallel branches (nn.ViewBranches):
        on the GPU, in the bf16 preset, not already inside a branch.  So does an inference pass over an even batch of at
        least 64 images (two half batches)."""
        from .. import precision

        # inference (embedding dump, kNN bank build / validation: eval mode, no gradient): the batch as two half-batch
        # branches -- no statistics, no gradients, bit-identical outputs; 1.90 -> 1.74 ms per 256 images, 3.53 -> 3.18 per 512
        # (tools/probes/eval_branches_probe.py).  WM_EVAL_BRANCHES=0: one stream.
        if os.environ.get("WM_EVAL_BRANCHES", "1") != "0" and not self.training and not torch.is_grad_enabled() \
                and x.is_cuda and x.shape[0] % 2 == 0 and x.shape[0] >= 64 and ops.current_branch() == 0 \
                and self.view_branches and not precision.is_f32():
            if getattr(self, "_branches", None) is None or not self._branches.valid():
                if torch.cuda.is_current_stream_capturing():
                    return False
                try:
                    self._branches = hnn.ViewBranches(self)
                except ValueError:
                    self.view_branches = False
                    return False
            return True
        if not (self.view_branches and self.training and x.is_cuda and ops.current_bn_groups() == 2
                and x.shape[0] % 2 == 0 and ops.current_branch() == 0):
            return False
        if os.environ.get("WM_VIEW_BRANCHES", "1") == "0" or precision.is_f32():
            return False
        vb = getattr(self, "_branches", None)
        if vb is None or not vb.valid():
            if torch.cuda.is_current_stream_capturing():
                return False   # (built by the eager warm-up steps that precede every capture)
            try:
                vb = self._branches = hnn.ViewBranches(self)
            except ValueError:   # (layers without running statistics or with differing momenta: one stream)
                self.view_branches = False
                return False
        # synchronised BatchNorm exchanges statistics between the ranks inside every layer: two branches would issue those
        # collectives from two streams, in an order that may differ from rank to rank
        return not any(m._synced() for m in vb.modules)

    def _forward_branches(self, x):
        vb = self._branches
        b = x.shape[0] // 2
        cur = torch.cuda.current_stream(x.device)
        vb.prepare()
        vb.side.wait_stream(cur)
        with ops.bn_groups(1):
            f0 = ops.global_avg_pool(self.forward_features(x[:b]))
            with torch.cuda.stream(vb.side), ops.branch(1):
                f1 = ops.global_avg_pool(self.forward_features(x[b:]))
        cur.wait_stream(vb.side)
        if self.training:
            vb.merge()
        return ops.stack_rows(f0.flatten(start_dim=1), f1.flatten(start_dim=1))


def create_model(name: str, num_classes: int = 0, pretrained: bool = False, **kw):
    """The one timm entry point the reference's hot path uses."""
    if name != "resnet18":
        raise NotImplementedError(f"model {name!r} is not built (only resnet18 is on the hot path)")
    if pretrained:
        raise NotImplementedError("no network: pretrained weights cannot be fetched")
    return ResNet18(num_classes=num_classes)
