"""Float32 ("parity") preset: tensor-level wrappers over the wm_f32_* entry points (csrc/f32path.hip).

The convolution / BatchNorm / pooling / Linear ops (the ResNet-18 + projection-head path) have backward passes, so a whole SimCLR
optimiser step runs under the preset; the transformer-specific ops (LayerNorm, attention, fused activations, the DINO / MSE losses) are
forward-only and say so when differentiated.  Activations are float32: images / feature maps [N, C, H, W] in channels_last memory (NHWC), token and
feature matrices [rows, C]; parameters are used in their float32 master layout.  torch moves data here (cat, index
gather / scatter, reshape); every FLOP runs in the HIP kernels.  See precision.py for why the preset exists."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2


class _NoBackward(torch.autograd.Function):
    """Marks a float32-preset result: differentiating through it raises (the preset is forward-only)."""

    @staticmethod
    def forward(ctx, y, *deps):
        return y.view_as(y)

    @staticmethod
    def backward(ctx, *g):
        raise NotImplementedError("the float32 (parity) preset is forward-only: run the backward pass under the bf16 preset")


def _mark(y: torch.Tensor, *deps) -> torch.Tensor:
    if torch.is_grad_enabled() and any(torch.is_tensor(d) and d.requires_grad for d in deps):
        return _NoBackward.apply(y, *[d for d in deps if torch.is_tensor(d)])
    return y


def _cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _lib.WaferHipError(f"{what}: the HIP path needs a device tensor (no CPU fallback)")


def _f32(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    return t.detach().to(torch.float32).contiguous()


def as_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[N, C, H, W] of any float dtype / layout -> float32 channels_last."""
    if x.dim() != 4:
        raise ValueError("expected a 4-D [N,C,H,W] tensor")
    return x.detach().to(torch.float32).contiguous(memory_format=torch.channels_last)


_WS = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    # per (device, stream): two passes may be in flight on different streams (the DINO teacher beside the student)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WS[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    return ws


class _Conv2d(torch.autograd.Function):
    """y = act(conv(x, w) + bias) + residual on NHWC float32; backward (act = none): input gradient, weight gradient (pixel
    ranges summed in a fixed order), bias gradient = column sums, residual gradient = dy."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, stride, padding, act):
        xs = as_nhwc(x)
        n, c, h, w = xs.shape
        k, c2, r, s = weight.shape
        if c2 != c:
            raise ValueError(f"conv2d: input channels {c} vs weight {tuple(weight.shape)}")
        p, q = (h + 2 * padding - r) // stride + 1, (w + 2 * padding - s) // stride + 1
        y = torch.empty((n, p, q, k), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
        lib = _lib.load()
        ws = _workspace(lib.wm_f32_conv2d_workspace_bytes(c, k, r, s), x.device)
        res = as_nhwc(residual) if residual is not None else None
        wf = _f32(weight)
        check(lib.wm_f32_conv2d_fwd(ptr(xs), ptr(wf), ptr(_f32(bias)), ptr(res), ptr(y), n, h, w, c, k, r, s, p, q, stride,
                                    padding, int(act), ptr(ws), ws.numel(), stream_ptr()), "wm_f32_conv2d_fwd")
        ctx.save_for_backward(xs, wf)
        ctx.geom = (n, h, w, c, k, r, s, p, q, stride, padding)
        ctx.act, ctx.has_bias, ctx.has_res = int(act), bias is not None, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.act != ACT_NONE:
            raise NotImplementedError("float32 preset: no backward through a fused activation (the transformer path is forward-only)")
        xs, wf = ctx.saved_tensors
        n, h, w, c, k, r, s, p, q, stride, padding = ctx.geom
        dy = as_nhwc(dy)
        lib = _lib.load()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dy.device).permute(0, 3, 1, 2)
            ws = _workspace(lib.wm_f32_conv2d_workspace_bytes(c, k, r, s), dy.device)
            check(lib.wm_f32_conv2d_dgrad(ptr(dy), ptr(wf), ptr(dx), n, h, w, c, k, r, s, p, q, stride, padding, ptr(ws),
                                          ws.numel(), stream_ptr()), "wm_f32_conv2d_dgrad")
        if ctx.needs_input_grad[1]:
            dw = torch.empty((k, c, r, s), dtype=torch.float32, device=dy.device)
            ws = _workspace(lib.wm_f32_conv2d_wgrad_workspace_bytes(n, p, q, c, k, r, s), dy.device)
            check(lib.wm_f32_conv2d_wgrad(ptr(dy), ptr(xs), ptr(dw), n, h, w, c, k, r, s, p, q, stride, padding, ptr(ws),
                                          ws.numel(), stream_ptr()), "wm_f32_conv2d_wgrad")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty((k,), dtype=torch.float32, device=dy.device)
            check(lib.wm_f32_colsum(ptr(dy), n * p * q, k, ptr(db), stream_ptr()), "wm_f32_colsum")
        return dx, dw, db, (dy if ctx.has_res else None), None, None, None


def conv2d(x, weight, stride=1, padding=0, bias=None, act=ACT_NONE, residual=None):
    _cuda(x, "conv2d(float32)")
    return _Conv2d.apply(x, weight, bias, residual, int(stride), int(padding), int(act))


def linear(x, weight, bias=None, act=ACT_NONE, residual=None):
    """act(x @ W^T + bias) + residual on [rows, C]: the 1 x 1 convolution on a 1 x 1 image (same kernels, same backward)."""
    _cuda(x, "linear(float32)")
    if x.dim() != 2 or weight.dim() != 2 or x.shape[1] != weight.shape[1]:
        raise ValueError(f"linear: x {tuple(x.shape)} vs weight {tuple(weight.shape)}")
    rows, c = x.shape
    k = weight.shape[0]
    res4 = residual.reshape(rows, k, 1, 1) if residual is not None else None
    y = _Conv2d.apply(x.reshape(rows, c, 1, 1), weight.reshape(k, c, 1, 1), bias, res4, 1, 0, int(act))
    return y.reshape(rows, k)


class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, residual, gamma, beta, running_mean, running_var, training, relu, eps, momentum, groups, counter):
        four = y.dim() == 4
        ys = as_nhwc(y) if four else _f32(y)
        if four:
            n, c, h, w = ys.shape
            rows = n * h * w
        else:
            rows, c = ys.shape
        g = groups if training else 1
        if rows % g:
            raise ValueError("batch_norm: rows not divisible by groups")
        res = None
        if residual is not None:
            res = as_nhwc(residual) if four else _f32(residual)
        out = torch.empty_like(ys)
        mean = torch.empty((g, c), dtype=torch.float32, device=y.device)
        invstd = torch.empty_like(mean)
        lib = _lib.load()
        ws = _workspace(lib.wm_f32_bn_workspace_bytes(rows, c, g), y.device)
        check(lib.wm_f32_bn_fwd(ptr(ys), ptr(res), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                ptr(counter) if training else 0, rows, c, g, int(bool(training)), float(eps),
                                float(momentum), int(bool(relu)), ptr(mean), ptr(invstd), ptr(out), ptr(ws), ws.numel(),
                                stream_ptr()), "wm_f32_bn_fwd")
        ctx.save_for_backward(ys, out if relu else None, mean, invstd, _f32(gamma))
        ctx.meta = (rows, c, g, bool(training), residual is not None, gamma is not None and gamma.requires_grad)
        return out

    @staticmethod
    def backward(ctx, dout):
        ys, out, mean, invstd, gamma = ctx.saved_tensors
        rows, c, g, training, has_res, affine = ctx.meta
        if not training:
            raise NotImplementedError("batch_norm (float32 preset): backward through eval-mode statistics is not implemented")
        dout = as_nhwc(dout) if dout.dim() == 4 else _f32(dout)
        lib = _lib.load()
        dy = torch.empty_like(ys)
        dz = torch.empty_like(ys) if has_res else None
        dgamma = torch.empty((c,), dtype=torch.float32, device=ys.device)
        dbeta = torch.empty_like(dgamma)
        ws = _workspace(lib.wm_f32_bn_workspace_bytes(rows, c, g), ys.device)
        check(lib.wm_f32_bn_bwd(ptr(ys), ptr(dout), ptr(out), ptr(gamma), ptr(mean), ptr(invstd), rows, c, g, ptr(dgamma),
                                ptr(dbeta), ptr(dy), ptr(dz), ptr(ws), ws.numel(), stream_ptr()), "wm_f32_bn_bwd")
        return dy, dz, (dgamma if affine else None), (dbeta if affine else None), None, None, None, None, None, None, None, None


def batch_norm(y, gamma, beta, running_mean, running_var, training, residual=None, relu=False, eps=1e-5, momentum=0.1,
               groups=1, num_batches_tracked=None):
    _cuda(y, "batch_norm(float32)")
    return _BatchNorm.apply(y, residual, gamma, beta, running_mean, running_var, bool(training), bool(relu), float(eps),
                            float(momentum), int(groups), num_batches_tracked)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xs = as_nhwc(x)
        n, c, h, w = xs.shape
        p, q = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((n, p, q, c), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
        check(_lib.load().wm_f32_maxpool3x3s2(ptr(xs), n, h, w, c, ptr(y), stream_ptr()), "wm_f32_maxpool3x3s2")
        ctx.save_for_backward(xs)
        return y

    @staticmethod
    def backward(ctx, dy):
        (xs,) = ctx.saved_tensors
        n, c, h, w = xs.shape
        dy = as_nhwc(dy)
        dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dy.device).permute(0, 3, 1, 2)
        check(_lib.load().wm_f32_maxpool3x3s2_bwd(ptr(xs), ptr(dy), n, h, w, c, ptr(dx), stream_ptr()), "wm_f32_maxpool3x3s2_bwd")
        return dx


def max_pool3x3s2(x):
    return _MaxPool.apply(x)


class _Gap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xs = as_nhwc(x)
        n, c, h, w = xs.shape
        y = torch.empty((n, c), dtype=torch.float32, device=x.device)
        check(_lib.load().wm_f32_gap(ptr(xs), n, h * w, c, ptr(y), stream_ptr()), "wm_f32_gap")
        ctx.geom = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.geom
        dy = _f32(dy)
        dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dy.device).permute(0, 3, 1, 2)
        check(_lib.load().wm_f32_gap_bwd(ptr(dy), n, h * w, c, ptr(dx), stream_ptr()), "wm_f32_gap_bwd")
        return dx


def global_avg_pool(x):
    return _Gap.apply(x)


def layer_norm(x, gamma, beta, eps=1e-6):
    _cuda(x, "layer_norm(float32)")
    xs = _f32(x)
    rows, c = xs.shape
    y = torch.empty_like(xs)
    check(_lib.load().wm_f32_layernorm(ptr(xs), ptr(_f32(gamma)), ptr(_f32(beta)), float(eps), rows, c, ptr(y), stream_ptr()),
          "wm_f32_layernorm")
    return _mark(y, x, gamma, beta)


def bias_act(x, bias=None, act=ACT_NONE, residual=None):
    _cuda(x, "bias_act(float32)")
    xs = _f32(x)
    shape = xs.shape
    xs = xs.reshape(-1, shape[-1])
    rows, c = xs.shape
    y = torch.empty_like(xs)
    res = _f32(residual).reshape(rows, c) if residual is not None else None
    check(_lib.load().wm_f32_bias_act(ptr(xs), ptr(_f32(bias)), ptr(res), int(act), rows, c, ptr(y), stream_ptr()),
          "wm_f32_bias_act")
    return _mark(y.view(shape), x, bias, residual)


def attention(qkv, batch, seq, heads, scale=None, head_dim=64):
    _cuda(qkv, "attention(float32)")
    q = _f32(qkv)
    if q.shape != (batch * seq, 3 * heads * head_dim):
        raise ValueError(f"attention: qkv {tuple(q.shape)} vs batch {batch} seq {seq} heads {heads} x {head_dim}")
    out = torch.empty((batch * seq, heads * head_dim), dtype=torch.float32, device=qkv.device)
    sc = float(scale) if scale is not None else head_dim ** -0.5
    check(_lib.load().wm_f32_attention(ptr(q), batch, seq, heads, head_dim, sc, ptr(out), stream_ptr()), "wm_f32_attention")
    return _mark(out, qkv)


def attention_segments(qkv, segments, heads, scale=None, head_dim=64):
    outs, off = [], 0
    for n, seq in segments:
        outs.append(attention(qkv[off:off + n * seq], n, seq, heads, scale, head_dim))
        off += n * seq
    return torch.cat(outs, dim=0)


def patch_embed(images, weight):
    """[N,3,S,S] images, conv weight [D,3,p,p] -> patch rows [N * (S/p)^2, D] (no bias; bias_act adds it)."""
    p = weight.shape[-1]
    y = conv2d(images, weight, stride=p, padding=0)          # logical [N, D, g, g], memory [N][g][g][D]
    n, d, g, _ = y.shape
    return y.permute(0, 2, 3, 1).reshape(n * g * g, d)


def tokens_assemble(patches, cls, pos, n, np_):
    """[N * np, D] patch rows -> [N * (np + 1), D] token rows: class token first, positional embedding added."""
    d = patches.shape[1]
    tok = torch.cat([_f32(cls).reshape(1, 1, d).expand(n, 1, d), _f32(patches).reshape(n, np_, d)], dim=1)
    posx = _f32(pos).reshape(1, np_ + 1, d).expand(n, np_ + 1, d).contiguous()
    return bias_act(tok.reshape(n * (np_ + 1), d), None, ACT_NONE, residual=posx.reshape(n * (np_ + 1), d))


def gather_rows(x, idx, batch, seq):
    """out[b * K + j] = x[b * seq + idx[b, j]]."""
    xs = _f32(x)
    c = xs.shape[1]
    k = idx.shape[1]
    src = xs.reshape(batch, seq, c)
    return torch.gather(src, 1, idx.long().unsqueeze(-1).expand(batch, k, c)).reshape(batch * k, c)


def scatter_rows(base, src, idx, batch, seq):
    b, s = _f32(base), _f32(src)
    c = b.shape[-1]
    k = idx.shape[1]
    out = b.reshape(batch, seq, c).clone()
    out.scatter_(1, idx.long().unsqueeze(-1).expand(batch, k, c), s.reshape(batch, k, c))
    return out.reshape(batch * seq, c)


def softmax_rows(x, subtract=None, inv_temp=1.0, log=False):
    xs = _f32(x)
    rows, d = xs.shape
    y = torch.empty_like(xs)
    check(_lib.load().wm_f32_softmax_rows(ptr(xs), ptr(_f32(subtract).reshape(-1)) if subtract is not None else 0, float(inv_temp),
                                          int(bool(log)), rows, d, ptr(y), stream_ptr()), "wm_f32_softmax_rows")
    return y


def _reduce(a, b, mode, scale):
    a = _f32(a).reshape(-1)
    b = _f32(b).reshape(-1) if b is not None else None
    out = torch.empty((), dtype=torch.float32, device=a.device)
    check(_lib.load().wm_f32_reduce(ptr(a), ptr(b), a.numel(), mode, float(scale), ptr(out), stream_ptr()), "wm_f32_reduce")
    return out


def mse_loss(pred, target):
    return _mark(_reduce(pred, target, 1, 1.0 / pred.numel()), pred)


def l1_loss(pred, target):
    return _mark(_reduce(pred, target, 2, 1.0 / pred.numel()), pred)


def dino_loss(student, probs, n_student_views, n_teacher_views, batch, student_temp):
    """lightly DINOLoss: mean over the (teacher view, student view != teacher view, sample) pairs of the cross entropy."""
    logq = softmax_rows(student, None, 1.0 / student_temp, log=True)
    d = logq.shape[1]
    pair = torch.empty(n_teacher_views * n_student_views * batch, dtype=torch.float32, device=logq.device)
    check(_lib.load().wm_f32_pair_ce(ptr(_f32(probs)), ptr(logq), n_teacher_views, n_student_views, batch, d, ptr(pair),
                                     stream_ptr()), "wm_f32_pair_ce")
    n_terms = n_teacher_views * n_student_views - min(n_teacher_views, n_student_views)
    return _mark(_reduce(pair, None, 0, 1.0 / (n_terms * batch)), student)


def dino_center_update(center, teacher, momentum):
    t = _f32(teacher)
    check(_lib.load().wm_f32_center_update(ptr(center), ptr(t), t.shape[0], t.shape[1], float(momentum), stream_ptr()),
          "wm_f32_center_update")
