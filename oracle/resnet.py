"""CPU oracle for the SimCLR ResNet-18 training step.  TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

torch-CPU float32 restatement of what the reference executes through third-party modules:
  timm.create_model("resnet18", num_classes=0)       scripts/WM811k_benchmark.py:231  
  heads.SimCLRProjectionHead(512, 512, 128)          scripts/WM811k_benchmark.py:233  
  SimCLR.forward / training_step                     scripts/WM811k_benchmark.py:236-248
  torch.optim.SGD(lr 6e-2*bs/256, m 0.9, wd 5e-4)    scripts/WM811k_benchmark.py:250-255
timm (pinned 0.8.19.dev0) and lightly (unpinned) are not installed here and the reference holds no
test or golden vector for them: PARITY UNPINNED upstream.  The functions below take a state_dict
with timm's keys, so the HIP path and the oracle always run on identical weights.
"""
import torch
import torch.nn.functional as F


def _bn(x, sd, prefix, training, relu=False, residual=None, momentum=0.1, eps=1e-5):
    y = F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                     sd[prefix + ".bias"], training, momentum, eps)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def _block(x, sd, p, stride, training):
    out = F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)
    out = _bn(out, sd, p + ".bn1", training, relu=True)
    out = F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1)
    shortcut = x
    if (p + ".downsample.0.weight") in sd:
        shortcut = F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0)
        shortcut = _bn(shortcut, sd, p + ".downsample.1", training)
    return _bn(out, sd, p + ".bn2", training, relu=True, residual=shortcut)


def resnet18_features(x, sd, training=True, prefix=""):
    """x [N,3,H,W] float32 -> [N,512].  `sd` maps timm keys to tensors; running stats are updated
    in place when training (as nn.BatchNorm2d does)."""
    g = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    x = F.conv2d(x, g["conv1.weight"], None, 2, 3)
    x = _bn(x, g, "bn1", training, relu=True)
    x = F.max_pool2d(x, 3, 2, 1)
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        x = _block(x, g, f"layer{li}.0", stride, training)
        x = _block(x, g, f"layer{li}.1", 1, training)
    return F.adaptive_avg_pool2d(x, 1).flatten(1)


def simclr_head(f, sd, training=True, prefix="projection_head."):
    g = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    h = F.linear(f, g["layers.0.weight"])
    h = _bn(h, g, "layers.1", training, relu=True)
    h = F.linear(h, g["layers.3.weight"])
    return _bn(h, g, "layers.4", training)


def simclr_loss(x0, x1, sd, temperature=0.5, training=True):
    """training_step of the reference: two separate forwards, then NTXentLoss."""
    from .ntxent import ntxent_lightly

    f0 = resnet18_features(x0, sd, training, "backbone.")
    z0 = simclr_head(f0, sd, training)
    f1 = resnet18_features(x1, sd, training, "backbone.")
    z1 = simclr_head(f1, sd, training)
    return ntxent_lightly(z0, z1, temperature), (f0, f1, z0, z1)


def sgd_step(params, grads, bufs, lr, momentum=0.9, weight_decay=5e-4):
    """torch.optim.SGD.step (no dampening / nesterov): in place."""
    for k in params:
        d = grads[k] + weight_decay * params[k]
        if bufs.get(k) is None:
            bufs[k] = d.clone()
        else:
            bufs[k].mul_(momentum).add_(d)
        params[k].sub_(lr * bufs[k])


def moco_head(f, sd, prefix="projection_head."):
    """lightly MoCoProjectionHead: Linear(+bias)-ReLU, Linear(+bias) (layers.0, layers.2)."""
    x = F.relu(F.linear(f, sd[prefix + "layers.0.weight"], sd[prefix + "layers.0.bias"]))
    return F.linear(x, sd[prefix + "layers.2.weight"], sd[prefix + "layers.2.bias"])


def _bn1d(x, sd, key, training, groups=1, relu=False, momentum=0.1, eps=1e-5):
    """BatchNorm1d with statistics per view group (what per-view forward calls compute); affine
    parameters are optional (lightly's SimSiam projection head ends with BatchNorm1d(affine=False))."""
    w, b = sd.get(key + ".weight"), sd.get(key + ".bias")
    parts = [F.batch_norm(p, sd[key + ".running_mean"], sd[key + ".running_var"], w, b, training, momentum, eps)
             for p in x.chunk(groups)]
    y = torch.cat(parts)
    return F.relu(y) if relu else y


def byol_head(f, sd, prefix, training=True, groups=1):
    """lightly BYOLProjectionHead / BYOLPredictionHead: Linear-BN-ReLU (layers.0-2), Linear+bias (layers.3)."""
    x = _bn1d(F.linear(f, sd[prefix + "layers.0.weight"]), sd, prefix + "layers.1", training, groups, relu=True)
    return F.linear(x, sd[prefix + "layers.3.weight"], sd[prefix + "layers.3.bias"])


def simsiam_projection_head(f, sd, prefix="projection_head.", training=True, groups=1):
    """(Linear-BN-ReLU) x2 (layers.0-5), Linear-BN(affine=False) (layers.6-7)."""
    x = _bn1d(F.linear(f, sd[prefix + "layers.0.weight"]), sd, prefix + "layers.1", training, groups, relu=True)
    x = _bn1d(F.linear(x, sd[prefix + "layers.3.weight"]), sd, prefix + "layers.4", training, groups, relu=True)
    return _bn1d(F.linear(x, sd[prefix + "layers.6.weight"]), sd, prefix + "layers.7", training, groups)


def neg_cosine(x0, x1, eps=1e-8):
    return -F.cosine_similarity(x0, x1, dim=1, eps=eps).mean()


def lars_step(params, grads, bufs, lr, momentum=0.9, weight_decay=0.0, trust_coeff=0.001, eps=1e-8):
    """timm.optim.lars.Lars (dampening 0, no nesterov, no trust clip, always_adapt False), in place."""
    for k, p in params.items():
        g = grads[k].clone()
        if weight_decay != 0:
            w_norm, g_norm = p.norm(2.0), g.norm(2.0)
            ratio = trust_coeff * w_norm / (g_norm + w_norm * weight_decay + eps)
            ratio = torch.where(w_norm > 0, torch.where(g_norm > 0, ratio, torch.ones_like(ratio)), torch.ones_like(ratio))
            g = (g + weight_decay * p) * ratio
        buf = bufs.get(k)
        buf = g.clone() if buf is None else buf.mul_(momentum).add_(g)
        bufs[k] = buf
        p.add_(buf, alpha=-lr)
