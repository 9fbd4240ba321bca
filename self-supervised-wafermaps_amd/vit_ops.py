"""Autograd pairing of the vision-transformer kernels (include/wafer_hip.h, "Vision-transformer path").

Token activations are bf16 [rows, C] (rows = images x tokens, token-major); parameters float32.
The Linear layers themselves run on the implicit-GEMM kernel (`ops.linear`, a 1x1 convolution on a
1x1 image); this module adds what surrounds them in a ViT block: LayerNorm, bias / GELU / residual
epilogues, attention, patch embedding, token assembly, row gather / scatter, the DINO loss and MSE.
No CPU fallback: every function raises when the HIP library is missing.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _lib, f32path, ops
from . import precision as _precision
from ._lib import check, ptr, stream_ptr
from .ops import _arena_grad, _need_cuda

ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2


def _bf16_rows(x: torch.Tensor) -> torch.Tensor:
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    return x.contiguous()


def _zeros(shape, dtype, device) -> torch.Tensor:
    """torch.zeros by the library's own fill kernel (keeps framework launches out of the captured step)."""
    t = torch.empty(shape, dtype=dtype, device=device)
    nbytes = t.numel() * t.element_size()
    if nbytes % 4 or not t.is_cuda:
        return t.zero_()
    check(_lib.load().wm_fill_zero(ptr(t), nbytes, stream_ptr()), "wm_fill_zero")
    return t


_INDEX_CACHE = {}
_INDEX_PINNED = set()   # keys looked up while a hipGraph was being captured: their device address is baked into the graph


def cached_index(key, build) -> torch.Tensor:
    """A constant int64 index tensor (class-token rows, patch rows of a token grid) built once per (shape, device).
    An entry used during a hipGraph capture is never evicted (the captured kernels read it at every replay: an evicted
    tensor's block would return to the caching allocator and be recycled under the graph)."""
    t = _INDEX_CACHE.get(key)
    capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    if t is None:
        if len(_INDEX_CACHE) > 256:
            for k in [k for k in _INDEX_CACHE if k not in _INDEX_PINNED][:128]:
                del _INDEX_CACHE[k]
        t = _INDEX_CACHE[key] = build()
    if capturing:
        _INDEX_PINNED.add(key)
    return t


def _grad_target(p: torch.Tensor):
    """(buffer the kernels accumulate into, value to hand to autograd)."""
    slot = _arena_grad(p)
    if slot is not None:
        return slot, None
    g = torch.zeros_like(p, dtype=torch.float32)
    return g, g


def _part_buffer(owner: torch.Tensor, attr: str, numel: int, device) -> torch.Tensor:
    """A persistent f32 slot buffer of `owner` (one per use of the parameter inside one backward pass)."""
    idx = getattr(owner, "_hip_pending", 0)
    bufs = getattr(owner, attr, None)
    if bufs is None:
        bufs = []
        setattr(owner, attr, bufs)
    while len(bufs) <= idx:
        bufs.append(None)
    part = bufs[idx]
    if part is None or part.numel() != numel or part.device != device:
        part = bufs[idx] = torch.empty(numel, dtype=torch.float32, device=device)
    return part


def _ln_backward(x, dy, gamma, beta, mean, rstd, dskip):
    """LayerNorm backward (+ the residual-path gradient `dskip` of a pre-norm block) -> (dx, dgamma, dbeta) with the
    parameter gradients None when they went straight into a fused optimiser's gradient arena.  Round 3: with arena slots
    the per-block channel sums are STORED as slots [2][blocks][C] (wm_layernorm_bwd_parts) and added in order by the
    backward pass's batched fold (ops.fold_wgrads) -- no f32 atomics, bit-reproducible; WM_LN_SLOTS=0: the atomic form."""
    rows, c = x.shape
    lib = _lib.load()
    dx = torch.empty_like(x)
    sg, sb = _arena_grad(gamma), _arena_grad(beta)
    if sg is not None and sb is not None and c % 4 == 0 and os.environ.get("WM_LN_SLOTS", "1") != "0":
        nb = int(lib.wm_layernorm_bwd_blocks(rows, c))
        part = _part_buffer(gamma, "_hip_ln_parts", 2 * nb * c, x.device)   # (applied twice in one pass: two buffers)
        check(lib.wm_layernorm_bwd_parts(ptr(x), ptr(dy), ptr(gamma), ptr(mean), ptr(rstd), rows, c,
                                         ptr(dskip) if dskip is not None else 0, ptr(dx), ptr(part), stream_ptr()),
              "wm_layernorm_bwd_parts")
        ops._queue_fold(gamma, part[: nb * c], nb, sg, 1, c, 1)
        ops._queue_fold(beta, part[nb * c:], nb, sb, 1, c, 1)
        return dx, None, None
    dg, dg_ret = _grad_target(gamma)
    db, db_ret = _grad_target(beta)
    if dskip is None:
        check(lib.wm_layernorm_bwd(ptr(x), ptr(dy), ptr(gamma), ptr(mean), ptr(rstd), rows, c, ptr(dx), ptr(dg),
                                   ptr(db), stream_ptr()), "wm_layernorm_bwd")
    else:
        check(lib.wm_layernorm_bwd_add(ptr(x), ptr(dy), ptr(gamma), ptr(mean), ptr(rstd), rows, c, ptr(dskip), ptr(dx),
                                       ptr(dg), ptr(db), stream_ptr()), "wm_layernorm_bwd_add")
    return dx, dg_ret, db_ret


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _need_cuda(x, "layer_norm")
        x = _bf16_rows(x)
        rows, c = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        check(_lib.load().wm_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), eps, rows, c, ptr(y), ptr(mean), ptr(rstd),
                                           stream_ptr()), "wm_layernorm_fwd")
        ctx.save_for_backward(x, mean, rstd)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        dx, dg_ret, db_ret = _ln_backward(x, _bf16_rows(dy), gamma, beta, mean, rstd, None)
        return dx, dg_ret, db_ret, None


def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """nn.LayerNorm over the last dim of bf16 [rows, C]."""
    if _precision.is_f32():
        return f32path.layer_norm(x, gamma, beta, eps)
    return _LayerNorm.apply(x, gamma, beta, float(eps))


class _LayerNormSkip(torch.autograd.Function):
    """(LayerNorm(x), x): the two uses of x in a pre-norm residual block (x -> LN -> f -> + x) as ONE autograd node, so
    the gradients of both uses meet inside the LayerNorm backward kernel (wm_layernorm_bwd_add) instead of in a
    separate accumulation pass over the activation (two of those per transformer block and step otherwise)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _need_cuda(x, "layer_norm_skip")
        x = _bf16_rows(x)
        rows, c = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        check(_lib.load().wm_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), eps, rows, c, ptr(y), ptr(mean), ptr(rstd),
                                           stream_ptr()), "wm_layernorm_fwd")
        ctx.save_for_backward(x, mean, rstd)
        ctx.params = (gamma, beta)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        rows, c = x.shape
        lib = _lib.load()
        if dy is None:  # the normalised branch was not used
            return dskip, None, None, None
        dx, dg_ret, db_ret = _ln_backward(x, _bf16_rows(dy), gamma, beta, mean, rstd,
                                          _bf16_rows(dskip) if dskip is not None else None)
        return dx, dg_ret, db_ret, None


def layer_norm_skip(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-6):
    """(LayerNorm(x), x) for a pre-norm residual block; use the second value as the residual operand."""
    if _precision.is_f32():
        return f32path.layer_norm(x, gamma, beta, eps), x
    return _LayerNormSkip.apply(x, gamma, beta, float(eps))


class _BiasAct(torch.autograd.Function):
    """out = act(x + bias) (+ residual)."""

    @staticmethod
    def forward(ctx, x, bias, residual, act):
        _need_cuda(x, "bias_act")
        x = _bf16_rows(x)
        rows, c = x.shape
        if residual is not None:
            residual = _bf16_rows(residual)
        y = torch.empty_like(x)
        check(_lib.load().wm_bias_act_fwd(ptr(x), ptr(bias), ptr(residual), act, rows, c, ptr(y), stream_ptr()),
              "wm_bias_act_fwd")
        ctx.act = act
        ctx.bias = bias
        ctx.has_res = residual is not None
        ctx.save_for_backward(x if act != ACT_NONE else None)
        ctx.shape = (rows, c)
        return y

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        rows, c = ctx.shape
        dout = _bf16_rows(dout)
        bias = ctx.bias
        db = db_ret = None
        lib = _lib.load()
        slot = _arena_grad(bias) if (bias is not None and bias.requires_grad) else None
        if slot is not None and c % 4 == 0 and os.environ.get("WM_LN_SLOTS", "1") != "0":
            # bias gradient as per-block slots added in order by the pass's batched fold (no f32 atomics)
            nb = int(lib.wm_colsum_blocks(rows, c))
            part = _part_buffer(bias, "_hip_bias_parts", nb * c, dout.device)
            dx = dout if ctx.act == ACT_NONE else torch.empty_like(dout)
            check(lib.wm_bias_act_bwd_parts(ptr(x) if ctx.act != ACT_NONE else 0, ptr(bias) if ctx.act != ACT_NONE else 0,
                                            ptr(dout), ctx.act, rows, c, ptr(dx) if ctx.act != ACT_NONE else 0, ptr(part),
                                            stream_ptr()), "wm_bias_act_bwd_parts")
            ops._queue_fold(bias, part, nb, slot, 1, c, 1)
            return dx, None, (dout if ctx.has_res else None), None
        if bias is not None and bias.requires_grad:
            db, db_ret = _grad_target(bias)
        if ctx.act == ACT_NONE:
            if db is not None:
                check(lib.wm_bias_act_bwd(0, 0, ptr(dout), ACT_NONE, rows, c, 0, ptr(db), stream_ptr()), "wm_bias_act_bwd")
            dx = dout
        else:
            dx = torch.empty_like(dout)
            check(lib.wm_bias_act_bwd(ptr(x), ptr(bias), ptr(dout), ctx.act, rows, c, ptr(dx), ptr(db), stream_ptr()),
                  "wm_bias_act_bwd")
        return dx, db_ret, (dout if ctx.has_res else None), None


def bias_act(x: torch.Tensor, bias: Optional[torch.Tensor], act: int = ACT_NONE,
             residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    if _precision.is_f32():
        return f32path.bias_act(x, bias, int(act), residual)
    return _BiasAct.apply(x, bias, residual, int(act))


class _LinearBias(torch.autograd.Function):
    """out = x @ W^T + bias (+ residual).  Backward: the bias gradient rides in the weight-gradient
    launch (wm_conv2d_wgrad_bias), so dout is read by exactly two kernels (dgrad, wgrad)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        _need_cuda(x, "linear")
        x = _bf16_rows(x)
        rows, c = x.shape
        k = weight.shape[0]
        if weight.shape[1] != c:
            raise ValueError(f"linear: x {tuple(x.shape)} vs weight {tuple(weight.shape)}")
        train = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)
        krsc, _ = ops._WCACHE.get(weight, kind="linear", need_crsk=train and x.requires_grad)
        lib = _lib.load()
        y = torch.empty((rows, k), dtype=torch.bfloat16, device=x.device)
        if residual is not None:
            residual = _bf16_rows(residual)
        # bias and residual are added in the GEMM kernel's epilogue
        check(ops._run("gemm_fwd", 2.0 * rows * c * k, lib.wm_conv2d_fwd_bias_res, ptr(x), ptr(krsc), ptr(bias),
                       ptr(residual), ptr(y), rows, 1, 1, c, k, 1, 1, 1, 1, 1, 0, stream_ptr()),
              "wm_conv2d_fwd_bias_res(linear)")
        ctx.save_for_backward(x)
        ctx.params = (weight, bias)
        ctx.geom = (rows, c, k)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        weight, bias = ctx.params
        rows, c, k = ctx.geom
        dout = _bf16_rows(dout)
        lib = _lib.load()
        dx = dw = db_ret = None
        if ctx.needs_input_grad[0]:
            _, crsk = ops._WCACHE.get(weight, kind="linear", need_crsk=True)
            dx = torch.empty((rows, c), dtype=torch.bfloat16, device=dout.device)
            check(ops._run("gemm_dgrad", 2.0 * rows * c * k, lib.wm_conv2d_dgrad, ptr(dout), ptr(crsk), ptr(dx), rows, 1, 1,
                           c, k, 1, 1, 1, 1, 1, 0, stream_ptr()), "wm_conv2d_dgrad(linear)")
        want_b = bias is not None and bias.requires_grad
        if ctx.needs_input_grad[1] or want_b:
            slabs, bslabs, ns = ops.wgrad(dout, x, weight, rows, 1, 1, c, k, 1, 1, 1, 1, 1, 0, bias_k=k if want_b else 0,
                                          name="gemm_wgrad")
            dw, db_ret = ops.wgrad_deliver(weight, slabs, ns, k, c, 1, 1, bias if want_b else None, bslabs)
        return dx, dw, db_ret, (dout if ctx.has_res else None)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
           residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(x @ W^T + bias) (+ residual) on bf16 [rows, C]: the GEMM kernel, then one epilogue pass."""
    if _precision.is_f32():
        return f32path.linear(x, weight, bias, int(act), residual)
    if act == ACT_NONE and (bias is not None or residual is not None):
        return _LinearBias.apply(x, weight, bias, residual)
    y = ops.linear(x, weight)
    if bias is None and act == ACT_NONE and residual is None:
        return y
    return bias_act(y, bias, act, residual)


class _ConstMatmul(torch.autograd.Function):
    """m @ p for a CONSTANT float32 matrix m [M, K] and a parameter-side float32 p ([K, N], or [1, K, N]: the positional
    embedding parameter itself): wm_matmul_f32 forward, m^T @ dout backward.  When `p` is a parameter whose gradient slot
    a fused optimiser owns, the backward adds its product into the slot with the library's own kernel and hands autograd
    None: a parameter used by several nodes (the positional embedding: directly by the 224 x 224 crops, through this
    resize by the 96 x 96 crops) would otherwise cost one framework add kernel per use."""

    @staticmethod
    def forward(ctx, m, p):
        _need_cuda(p, "const_matmul")
        m = m.contiguous().float()
        p2 = p.detach().reshape(-1, p.shape[-1]).contiguous().float()
        out = torch.empty((m.shape[0], p2.shape[1]), dtype=torch.float32, device=p.device)
        check(_lib.load().wm_matmul_f32(ptr(m), ptr(p2), ptr(out), m.shape[0], p2.shape[1], m.shape[1], 0, stream_ptr()),
              "wm_matmul_f32")
        ctx.save_for_backward(m)
        ctx.p = p
        return out

    @staticmethod
    def backward(ctx, dout):
        (m,) = ctx.saved_tensors
        p = ctx.p
        dout = dout.contiguous().float()
        dp = torch.empty((m.shape[1], dout.shape[1]), dtype=torch.float32, device=dout.device)
        # dp [K, N] = m^T [K, M] @ dout [M, N]: op(a) = a^T with a = m stored [M][K]
        check(_lib.load().wm_matmul_f32(ptr(m), ptr(dout), ptr(dp), m.shape[1], dout.shape[1], m.shape[0], 1, stream_ptr()),
              "wm_matmul_f32")
        slot = _arena_grad(p) if p.is_leaf else None
        if slot is not None:
            check(_lib.load().wm_wgrad_finalize(ptr(dp), 1, 1, dp.numel(), 1, 1, ptr(slot), 1, stream_ptr()), "wm_wgrad_finalize(add)")
            return None, None
        return None, dp.reshape(p.shape)


def const_matmul(m: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    return _ConstMatmul.apply(m, p)


class _MlpGelu(torch.autograd.Function):
    """y = fc2(gelu(fc1(x))) (+ residual), four GEMM-shaped launches forward + backward each carrying the
    element-wise work in its epilogue: fc1 + bias + GELU (pre-activation kept), fc2 + bias + residual; backward:
    wgrad(+bias) of fc2, dgrad of fc2 x gelu'(pre), wgrad(+bias) of fc1, dgrad of fc1."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual):
        _need_cuda(x, "mlp")
        x = _bf16_rows(x)
        rows, c = x.shape
        hid, out = w1.shape[0], w2.shape[0]
        if w1.shape[1] != c or w2.shape[1] != hid or hid % 64 or out % 64 or c % 64:
            raise ValueError(f"mlp: x {tuple(x.shape)}, fc1 {tuple(w1.shape)}, fc2 {tuple(w2.shape)}")
        lib = _lib.load()
        k1, _ = ops._WCACHE.get(w1, kind="linear", need_crsk=True)
        k2, _ = ops._WCACHE.get(w2, kind="linear", need_crsk=True)
        pre = torch.empty((rows, hid), dtype=torch.bfloat16, device=x.device)
        h = torch.empty_like(pre)
        check(ops._run("gemm_fwd", 2.0 * rows * c * hid, lib.wm_linear_bias_gelu_fwd, ptr(x), ptr(k1), ptr(b1), ptr(pre),
                       ptr(h), rows, c, hid, stream_ptr()), "wm_linear_bias_gelu_fwd")
        y = torch.empty((rows, out), dtype=torch.bfloat16, device=x.device)
        if residual is not None:
            residual = _bf16_rows(residual)
        check(ops._run("gemm_fwd", 2.0 * rows * hid * out, lib.wm_conv2d_fwd_bias_res, ptr(h), ptr(k2), ptr(b2),
                       ptr(residual), ptr(y), rows, 1, 1, hid, out, 1, 1, 1, 1, 1, 0, stream_ptr()),
              "wm_conv2d_fwd_bias_res(fc2)")
        ctx.save_for_backward(x, pre, h)
        ctx.params = (w1, b1, w2, b2)
        ctx.geom = (rows, c, hid, out)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pre, h = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        rows, c, hid, out = ctx.geom
        dy = _bf16_rows(dy)
        lib = _lib.load()
        rets = {}

        def wgrad(dout, inp, w, b, kk, cc, tag):
            want_b = b.requires_grad
            slabs, bslabs, ns = ops.wgrad(dout, inp, w, rows, 1, 1, cc, kk, 1, 1, 1, 1, 1, 0, bias_k=kk if want_b else 0,
                                          name="gemm_wgrad")
            rets[tag] = ops.wgrad_deliver(w, slabs, ns, kk, cc, 1, 1, b if want_b else None, bslabs)

        wgrad(dy, h, w2, b2, out, hid, "fc2")
        _, c2 = ops._WCACHE.get(w2, kind="linear", need_crsk=True)
        dpre = torch.empty_like(pre)
        check(ops._run("gemm_dgrad", 2.0 * rows * hid * out, lib.wm_linear_dgrad_gelu, ptr(dy), ptr(c2), ptr(pre),
                       ptr(dpre), rows, hid, out, stream_ptr()), "wm_linear_dgrad_gelu")
        wgrad(dpre, x, w1, b1, hid, c, "fc1")
        dx = None
        if ctx.needs_input_grad[0]:
            _, c1 = ops._WCACHE.get(w1, kind="linear", need_crsk=True)
            dx = torch.empty((rows, c), dtype=torch.bfloat16, device=dy.device)
            check(ops._run("gemm_dgrad", 2.0 * rows * c * hid, lib.wm_conv2d_dgrad, ptr(dpre), ptr(c1), ptr(dx), rows, 1, 1,
                           c, hid, 1, 1, 1, 1, 1, 0, stream_ptr()), "wm_conv2d_dgrad(fc1)")
        return dx, rets["fc1"][0], rets["fc1"][1], rets["fc2"][0], rets["fc2"][1], (dy if ctx.has_res else None)


def mlp_gelu(x: torch.Tensor, fc1_weight: torch.Tensor, fc1_bias: torch.Tensor, fc2_weight: torch.Tensor,
             fc2_bias: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fc2(gelu(fc1(x))) (+ residual) on bf16 [rows, C] with biases: the transformer MLP block.
    When no gradient is being recorded (DINO teacher, validation, embedding inference) and the shape is served
    (wm_mlp_fused_fwd_ok: C = 192, ViT-Tiny), the whole block is ONE launch with the hidden activation kept in LDS
    (wm_mlp_fused_fwd) -- bit-identical to the two-launch path."""
    if _precision.is_f32():
        h = f32path.linear(x, fc1_weight, fc1_bias, ACT_GELU)
        return f32path.linear(h, fc2_weight, fc2_bias, ACT_NONE, residual)
    needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in
                                                 (x, fc1_weight, fc1_bias, fc2_weight, fc2_bias, residual))
    if not needs_grad and os.environ.get("WM_MLP_FUSED", "1") != "0":
        lib = _lib.load()
        rows, c = x.shape
        hid = fc1_weight.shape[0]
        if x.is_cuda and x.dim() == 2 and fc2_weight.shape[0] == c and lib.wm_mlp_fused_fwd_ok(rows, c, hid):
            xb = _bf16_rows(x)
            k1, _ = ops._WCACHE.get(fc1_weight, kind="linear")
            k2, _ = ops._WCACHE.get(fc2_weight, kind="linear")
            y = torch.empty((rows, c), dtype=torch.bfloat16, device=x.device)
            rb = _bf16_rows(residual) if residual is not None else None
            check(ops._run("gemm_fwd", 4.0 * rows * c * hid, lib.wm_mlp_fused_fwd, ptr(xb), ptr(k1), ptr(fc1_bias.detach()),
                           ptr(k2), ptr(fc2_bias.detach()), ptr(rb), ptr(y), rows, c, hid, stream_ptr()), "wm_mlp_fused_fwd")
            return y
    return _MlpGelu.apply(x, fc1_weight, fc1_bias, fc2_weight, fc2_bias, residual)


def _no_grad_for(*tensors) -> bool:
    return not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors))


def ln_linear(x: torch.Tensor, ln_weight: torch.Tensor, ln_bias: torch.Tensor, eps: float, weight: torch.Tensor,
              bias: Optional[torch.Tensor]):
    """Linear(LayerNorm(x)) as ONE launch (wm_ln_linear_fwd: the normalised rows exist only as register fragments), for
    passes that record no gradient and shapes the kernel serves (192-wide rows: ViT-Tiny's norm1 -> qkv).  Returns None
    when it does not apply: the caller runs the two launches."""
    if _precision.is_f32():
        return None
    if os.environ.get("WM_LN_FUSED", "1") == "0" or not _no_grad_for(x, ln_weight, ln_bias, weight, bias):
        return None
    if not (x.is_cuda and x.dim() == 2):
        return None
    rows, c = x.shape
    n = weight.shape[0]
    lib = _lib.load()
    if weight.shape[1] != c or not lib.wm_ln_linear_fwd_ok(rows, c, n):
        return None
    xb = _bf16_rows(x)
    krsc, _ = ops._WCACHE.get(weight, kind="linear")
    y = torch.empty((rows, n), dtype=torch.bfloat16, device=x.device)
    check(ops._run("gemm_fwd", 2.0 * rows * c * n, lib.wm_ln_linear_fwd, ptr(xb), ptr(ln_weight.detach()), ptr(ln_bias.detach()),
                   float(eps), ptr(krsc), ptr(bias.detach()) if bias is not None else 0, ptr(y), rows, c, n, stream_ptr()),
          "wm_ln_linear_fwd")
    return y


def ln_mlp_gelu(x: torch.Tensor, ln_weight: torch.Tensor, ln_bias: torch.Tensor, eps: float, fc1_weight: torch.Tensor,
                fc1_bias: torch.Tensor, fc2_weight: torch.Tensor, fc2_bias: torch.Tensor):
    """x + fc2(gelu(fc1(LayerNorm(x)))): the second half of a pre-norm block as ONE launch (wm_ln_mlp_fused_fwd), under the
    same conditions as ln_linear; None when it does not apply."""
    if _precision.is_f32():
        return None
    if os.environ.get("WM_LN_FUSED", "1") == "0" or os.environ.get("WM_MLP_FUSED", "1") == "0":
        return None
    if not _no_grad_for(x, ln_weight, ln_bias, fc1_weight, fc1_bias, fc2_weight, fc2_bias) or not (x.is_cuda and x.dim() == 2):
        return None
    rows, c = x.shape
    hid = fc1_weight.shape[0]
    lib = _lib.load()
    if fc2_weight.shape[0] != c or not lib.wm_mlp_fused_fwd_ok(rows, c, hid):
        return None
    xb = _bf16_rows(x)
    k1, _ = ops._WCACHE.get(fc1_weight, kind="linear")
    k2, _ = ops._WCACHE.get(fc2_weight, kind="linear")
    y = torch.empty((rows, c), dtype=torch.bfloat16, device=x.device)
    check(ops._run("gemm_fwd", 4.0 * rows * c * hid, lib.wm_ln_mlp_fused_fwd, ptr(xb), ptr(ln_weight.detach()),
                   ptr(ln_bias.detach()), float(eps), ptr(k1), ptr(fc1_bias.detach()), ptr(k2), ptr(fc2_bias.detach()), ptr(xb),
                   ptr(y), rows, c, hid, stream_ptr()), "wm_ln_mlp_fused_fwd")
    return y


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, b, s, h, hd, scale):
        _need_cuda(qkv, "attention")
        qkv = _bf16_rows(qkv)
        if qkv.shape != (b * s, 3 * h * hd):
            raise ValueError(f"attention: qkv {tuple(qkv.shape)} vs B={b} S={s} H={h} head_dim={hd}")
        out = torch.empty((b * s, h * hd), dtype=torch.bfloat16, device=qkv.device)
        lse = torch.empty((b, h, s), dtype=torch.float32, device=qkv.device)
        check(ops._run("attn_fwd", 4.0 * b * h * s * s * hd, _lib.load().wm_attention_fwd, ptr(qkv), b, s, h, hd, scale,
                       ptr(out), ptr(lse), stream_ptr()), "wm_attention_fwd")
        ctx.save_for_backward(qkv, out, lse)
        ctx.geom = (b, s, h, hd, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        b, s, h, hd, scale = ctx.geom
        dout = _bf16_rows(dout)
        dqkv = torch.empty_like(qkv)
        check(ops._run("attn_bwd", 8.0 * b * h * s * s * hd, _lib.load().wm_attention_bwd, ptr(qkv), ptr(out), ptr(dout),
                       ptr(lse), b, s, h, hd, scale, ptr(dqkv), stream_ptr()), "wm_attention_bwd")
        return dqkv, None, None, None, None, None


def attention(qkv: torch.Tensor, batch: int, seq: int, heads: int, scale: Optional[float] = None,
              head_dim: int = 64) -> torch.Tensor:
    """softmax(scale q k^T) v per head; qkv bf16 [B*S, 3*H*hd] as the qkv Linear emits it -> [B*S, H*hd]
    (head_dim 64: ViT-S/16, ViT-B/32; 32: the MAE decoder's 512 / 16)."""
    if _precision.is_f32():
        return f32path.attention(qkv, int(batch), int(seq), int(heads), scale, int(head_dim))
    return _Attention.apply(qkv, int(batch), int(seq), int(heads), int(head_dim),
                            float(head_dim ** -0.5 if scale is None else scale))


class _AttentionSegments(torch.autograd.Function):
    """Attention over SEGMENTS of one row-concatenated token tensor: segment g holds b_g sequences of s_g tokens (DINO's
    multi-crop student: 2 x B global crops of 197 tokens and 6 x B local crops of 37).  One attention launch per
    segment, straight into / out of row slices of the shared buffers, so every other layer of the block (LayerNorm,
    qkv / proj / MLP GEMMs and their weight gradients) runs ONCE over all rows instead of once per resolution."""

    @staticmethod
    def forward(ctx, qkv, segments, h, hd, scale):
        _need_cuda(qkv, "attention_segments")
        qkv = _bf16_rows(qkv)
        rows = sum(b * s for b, s in segments)
        if qkv.shape != (rows, 3 * h * hd):
            raise ValueError(f"attention_segments: qkv {tuple(qkv.shape)} vs segments {segments} H={h} head_dim={hd}")
        out = torch.empty((rows, h * hd), dtype=torch.bfloat16, device=qkv.device)
        lses, off = [], 0
        lib = _lib.load()
        for b, s in segments:
            lse = torch.empty((b, h, s), dtype=torch.float32, device=qkv.device)
            check(ops._run("attn_fwd", 4.0 * b * h * s * s * hd, lib.wm_attention_fwd, ptr(qkv[off:off + b * s]), b, s, h,
                           hd, scale, ptr(out[off:off + b * s]), ptr(lse), stream_ptr()), "wm_attention_fwd")
            lses.append(lse)
            off += b * s
        ctx.save_for_backward(qkv, out, *lses)
        ctx.geom = (tuple(segments), h, hd, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, *lses = ctx.saved_tensors
        segments, h, hd, scale = ctx.geom
        dout = _bf16_rows(dout)
        dqkv = torch.empty_like(qkv)
        lib = _lib.load()
        off = 0
        for (b, s), lse in zip(segments, lses):
            sl = slice(off, off + b * s)
            check(ops._run("attn_bwd", 8.0 * b * h * s * s * hd, lib.wm_attention_bwd, ptr(qkv[sl]), ptr(out[sl]),
                           ptr(dout[sl]), ptr(lse), b, s, h, hd, scale, ptr(dqkv[sl]), stream_ptr()), "wm_attention_bwd")
            off += b * s
        return dqkv, None, None, None, None


def attention_segments(qkv: torch.Tensor, segments, heads: int, scale: Optional[float] = None,
                       head_dim: int = 64) -> torch.Tensor:
    """attention() over row-concatenated segments [(batch, seq), ...] of one qkv tensor."""
    if _precision.is_f32():
        return f32path.attention_segments(qkv, segments, int(heads), scale, int(head_dim))
    segments = [(int(b), int(s)) for b, s in segments]
    if len(segments) == 1:
        return attention(qkv, segments[0][0], segments[0][1], heads, scale, head_dim)
    return _AttentionSegments.apply(qkv, segments, int(heads), int(head_dim),
                                    float(head_dim ** -0.5 if scale is None else scale))


class _PatchEmbed(torch.autograd.Function):
    """Conv2d(3, D, kernel = stride = p) as patchify + GEMM; images need no gradient."""

    @staticmethod
    def forward(ctx, images, weight):
        _need_cuda(images, "patch_embed")
        x = ops._as_nhwc(images)
        n, c, s, s2 = x.shape
        d, c2, p, p2 = weight.shape
        if c != 3 or c2 != 3 or s != s2 or p != p2 or s % p:
            raise ValueError(f"patch_embed: images {tuple(images.shape)} vs weight {tuple(weight.shape)}")
        g = s // p
        lib = _lib.load()
        rows = torch.empty((n * g * g, p * p * 3), dtype=torch.bfloat16, device=x.device)
        check(lib.wm_patchify(ptr(x), n, s, p, ptr(rows), stream_ptr()), "wm_patchify")
        krsc, _ = ops._WCACHE.get(weight, kind="conv")  # [D][p][p][3] == the patch row order
        y = torch.empty((n * g * g, d), dtype=torch.bfloat16, device=x.device)
        k = p * p * 3
        check(lib.wm_conv2d_fwd(ptr(rows), ptr(krsc), ptr(y), n * g * g, 1, 1, k, d, 1, 1, 1, 1, 1, 0, stream_ptr()),
              "wm_conv2d_fwd(patch_embed)")
        ctx.save_for_backward(rows)
        ctx.weight = weight
        ctx.geom = (n * g * g, k, d, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        (rows,) = ctx.saved_tensors
        weight = ctx.weight
        m, k, d, p = ctx.geom
        dw = None
        if ctx.needs_input_grad[1]:
            dy = _bf16_rows(dy)
            lib = _lib.load()
            # rows [m][k = p*p*3] x dy [m][d] as a 1 x 1 convolution; the slabs are [d][p][p][3] = KRSC of the patch filter
            slabs, _, ns = ops.wgrad(dy, rows, weight, m, 1, 1, k, d, 1, 1, 1, 1, 1, 0, name="gemm_wgrad")
            ops.side_join()   # the finalize below reads the slabs on this stream
            slot = _arena_grad(weight)
            if slot is not None:
                check(lib.wm_wgrad_finalize(ptr(slabs), ns, d, 3, p, p, ptr(slot), 1, stream_ptr()), "wm_wgrad_finalize")
            else:
                dw = torch.empty((d, 3, p, p), dtype=torch.float32, device=dy.device)
                check(lib.wm_wgrad_finalize(ptr(slabs), ns, d, 3, p, p, ptr(dw), 0, stream_ptr()), "wm_wgrad_finalize")
        return None, dw


def patch_embed(images: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """images [N,3,S,S] (bf16 channels_last) x weight [D,3,p,p] -> bf16 [N*(S/p)^2, D] (no bias)."""
    if _precision.is_f32():
        return f32path.patch_embed(images, weight)
    return _PatchEmbed.apply(images, weight)


def _tokens_assemble_fwd(patches, cls, pos, n, np_, out):
    """wm_tokens_assemble of one group into `out` (a [n * (np + 1), d] bf16 row range)."""
    d = patches.shape[1]
    cls_c, pos_c = cls.detach().reshape(-1).contiguous(), pos.detach().reshape(-1, d).contiguous()
    if pos_c.shape[0] != np_ + 1 or patches.shape[0] != n * np_:
        raise ValueError(f"tokens_assemble: {tuple(patches.shape)} patches, pos {tuple(pos.shape)}, N={n}, np={np_}")
    check(_lib.load().wm_tokens_assemble(ptr(patches), ptr(cls_c), ptr(pos_c), n, np_, d, ptr(out), stream_ptr()),
          "wm_tokens_assemble")


def _tokens_assemble_bwd(dx, cls_p, pos_p, n, np_, d):
    """Gradients of one group: (dpatch, dcls or None, dpos or None) -- None where the sums went straight into a fused
    optimiser's gradient slot by the library's own add (the class token / positional embedding PARAMETERS are used once
    per crop resolution: autograd's accumulation would be one framework add kernel per use and parameter)."""
    s = np_ + 1
    lib = _lib.load()
    # sum over the images of the [s * d] token rows: per-block slots added in order (no f32 atomics)
    dpos = torch.empty(s * d, dtype=torch.float32, device=dx.device)
    nb = int(lib.wm_colsum_blocks(n, s * d))
    part = torch.empty(nb * s * d, dtype=torch.float32, device=dx.device)
    check(lib.wm_bias_act_bwd_parts(0, 0, ptr(dx), ACT_NONE, n, s * d, 0, ptr(part), stream_ptr()), "wm_bias_act_bwd_parts")
    check(lib.wm_wgrad_finalize(ptr(part), nb, 1, s * d, 1, 1, ptr(dpos), 0, stream_ptr()), "wm_wgrad_finalize")
    # the patch rows (every token but the class token of each image) by the row-gather kernel: a strided slice +
    # reshape would be an ATen copy kernel
    idx = cached_index(("patch_rows", n, np_, str(dx.device)),
                       lambda: torch.arange(1, s, dtype=torch.int64, device=dx.device).repeat(n, 1).contiguous())
    dpatch = torch.empty((n * np_, d), dtype=torch.bfloat16, device=dx.device)
    check(lib.wm_gather_rows(ptr(dx), ptr(idx), n, s, np_, d, ptr(dpatch), stream_ptr()), "wm_gather_rows")
    dcls, dposr = dpos[:d].reshape(cls_p.shape), dpos.reshape(pos_p.shape)
    sc = _arena_grad(cls_p) if cls_p.is_leaf else None
    sp = _arena_grad(pos_p) if pos_p.is_leaf else None
    if sc is not None:
        check(lib.wm_wgrad_finalize(ptr(dpos), 1, 1, d, 1, 1, ptr(sc), 1, stream_ptr()), "wm_wgrad_finalize(add cls)")
        dcls = None
    if sp is not None:
        check(lib.wm_wgrad_finalize(ptr(dpos), 1, 1, s * d, 1, 1, ptr(sp), 1, stream_ptr()), "wm_wgrad_finalize(add pos)")
        dposr = None
    return dpatch, dcls, dposr


class _TokensAssemble(torch.autograd.Function):
    @staticmethod
    def forward(ctx, patches, cls, pos, n, np_):
        _need_cuda(patches, "tokens_assemble")
        patches = _bf16_rows(patches)
        d = patches.shape[1]
        out = torch.empty((n * (np_ + 1), d), dtype=torch.bfloat16, device=patches.device)
        _tokens_assemble_fwd(patches, cls, pos, n, np_, out)
        ctx.geom = (n, np_, d)
        ctx.params = (cls, pos)
        return out

    @staticmethod
    def backward(ctx, dx):
        n, np_, d = ctx.geom
        dpatch, dcls, dpos = _tokens_assemble_bwd(_bf16_rows(dx), ctx.params[0], ctx.params[1], n, np_, d)
        return dpatch, dcls, dpos, None, None


class _TokensAssembleMulti(torch.autograd.Function):
    """tokens_assemble of SEVERAL groups (DINO's crop resolutions) into ONE row-concatenated token tensor: every group
    writes its row range of the shared buffer, so no concatenation pass follows (forward) and the gradient of a group is
    a row range of the incoming gradient (backward)."""

    @staticmethod
    def forward(ctx, cls, geoms, *flat):
        groups = [(flat[2 * i], flat[2 * i + 1]) for i in range(len(geoms))]
        _need_cuda(groups[0][0], "tokens_assemble_multi")
        d = groups[0][0].shape[1]
        rows = [n * (np_ + 1) for n, np_ in geoms]
        out = torch.empty((sum(rows), d), dtype=torch.bfloat16, device=groups[0][0].device)
        off = 0
        for (patches, pos), (n, np_), r in zip(groups, geoms, rows):
            _tokens_assemble_fwd(_bf16_rows(patches), cls, pos, n, np_, out[off:off + r])
            off += r
        ctx.geoms, ctx.d, ctx.cls = geoms, d, cls
        ctx.pos = [pos for _, pos in groups]
        return out

    @staticmethod
    def backward(ctx, dx):
        dx = _bf16_rows(dx)
        grads, dcls_sum, off = [], None, 0
        for (n, np_), pos in zip(ctx.geoms, ctx.pos):
            r = n * (np_ + 1)
            dpatch, dcls, dpos = _tokens_assemble_bwd(dx[off:off + r], ctx.cls, pos, n, np_, ctx.d)
            off += r
            grads += [dpatch, dpos]
            if dcls is not None:   # (no gradient slot: plain tensors; the framework adds them)
                dcls_sum = dcls if dcls_sum is None else dcls_sum + dcls
        return (dcls_sum, None, *grads)


def tokens_assemble_multi(groups, cls: torch.Tensor) -> torch.Tensor:
    """groups: [(patches [n * np, d], pos, n, np), ...] -> bf16 [sum n * (np + 1), d], the groups' token rows in order."""
    if _precision.is_f32():
        return torch.cat([f32path.tokens_assemble(p, cls, pos, int(n), int(np_)) for p, pos, n, np_ in groups], dim=0)
    geoms = tuple((int(n), int(np_)) for _, _, n, np_ in groups)
    flat = []
    for p, pos, _, _ in groups:
        flat += [p, pos]
    return _TokensAssembleMulti.apply(cls, geoms, *flat)


def tokens_assemble(patches: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, n: int, np_: int) -> torch.Tensor:
    """[cls + pos[0]; patches + pos[1:]] per image: bf16 [N*np, D] -> [N*(np+1), D]."""
    if _precision.is_f32():
        return f32path.tokens_assemble(patches, cls, pos, int(n), int(np_))
    return _TokensAssemble.apply(patches, cls, pos, int(n), int(np_))


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, b, s):
        _need_cuda(x, "gather_rows")
        x = _bf16_rows(x)
        c = x.shape[1]
        idx = idx.to(torch.int64).contiguous()
        k = idx.shape[1]
        out = torch.empty((b * k, c), dtype=torch.bfloat16, device=x.device)
        check(_lib.load().wm_gather_rows(ptr(x), ptr(idx), b, s, k, c, ptr(out), stream_ptr()), "wm_gather_rows")
        ctx.save_for_backward(idx)
        ctx.geom = (b, s, k, c)
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        b, s, k, c = ctx.geom
        dy = _bf16_rows(dy)
        dx = _zeros((b * s, c), torch.bfloat16, dy.device)
        check(_lib.load().wm_scatter_rows(ptr(dy), ptr(idx), b, s, k, c, ptr(dx), stream_ptr()), "wm_scatter_rows")
        return dx, None, None, None


def gather_rows(x: torch.Tensor, idx: torch.Tensor, batch: int, seq: int) -> torch.Tensor:
    """lightly get_at_index: x bf16 [B*S, C], idx int64 [B, K] (distinct per row) -> [B*K, C]."""
    if _precision.is_f32():
        return f32path.gather_rows(x, idx, int(batch), int(seq))
    return _GatherRows.apply(x, idx, int(batch), int(seq))


class _ScatterRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, base, src, idx, b, s):
        _need_cuda(base, "scatter_rows")
        base, src = _bf16_rows(base), _bf16_rows(src)
        c = base.shape[1]
        idx = idx.to(torch.int64).contiguous()
        k = idx.shape[1]
        out = base.clone()
        check(_lib.load().wm_scatter_rows(ptr(src), ptr(idx), b, s, k, c, ptr(out), stream_ptr()), "wm_scatter_rows")
        ctx.save_for_backward(idx)
        ctx.geom = (b, s, k, c)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        b, s, k, c = ctx.geom
        dout = _bf16_rows(dout)
        lib = _lib.load()
        dsrc = torch.empty((b * k, c), dtype=torch.bfloat16, device=dout.device)
        check(lib.wm_gather_rows(ptr(dout), ptr(idx), b, s, k, c, ptr(dsrc), stream_ptr()), "wm_gather_rows")
        dbase = dout.clone()
        zeros = torch.zeros_like(dsrc)
        check(lib.wm_scatter_rows(ptr(zeros), ptr(idx), b, s, k, c, ptr(dbase), stream_ptr()), "wm_scatter_rows")
        return dbase, dsrc, None, None, None


def scatter_rows(base: torch.Tensor, src: torch.Tensor, idx: torch.Tensor, batch: int, seq: int) -> torch.Tensor:
    """lightly set_at_index: copy of base [B*S, C] with rows idx [B, K] replaced by src [B*K, C]."""
    if _precision.is_f32():
        return f32path.scatter_rows(base, src, idx, int(batch), int(seq))
    return _ScatterRows.apply(base, src, idx, int(batch), int(seq))


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, l1=False):
        _need_cuda(pred, "mse_loss")
        pred, target = _bf16_rows(pred), _bf16_rows(target)
        if pred.shape != target.shape or pred.numel() % 8:
            raise ValueError(f"mse_loss: {tuple(pred.shape)} vs {tuple(target.shape)}")
        loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred)
        fn = _lib.load().wm_l1_fwd_bwd if l1 else _lib.load().wm_mse_fwd_bwd
        check(fn(ptr(pred), ptr(target), pred.numel(), ptr(loss), ptr(dpred), stream_ptr()),
              "wm_l1_fwd_bwd" if l1 else "wm_mse_fwd_bwd")
        ctx.save_for_backward(dpred)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        d = (dpred.float() * g).to(torch.bfloat16)
        return d, (-d if ctx.needs_input_grad[1] else None), None


def mse_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """nn.MSELoss() (mean) on bf16 tensors; the gradient is produced in the same pass."""
    if _precision.is_f32():
        return f32path.mse_loss(pred, target)
    return _MSE.apply(pred, target, False)


def l1_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """nn.L1Loss() (mean) on bf16 tensors."""
    if _precision.is_f32():
        return f32path.l1_loss(pred, target)
    return _MSE.apply(pred, target, True)


class _DinoLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, probs, vs, vt, b, temp_s):
        _need_cuda(student, "dino_loss")
        student = _bf16_rows(student)
        d = student.shape[1]
        if student.shape[0] != vs * b or tuple(probs.shape) != (vt * b, d) or probs.dtype != torch.float32:
            raise ValueError(f"dino_loss: student {tuple(student.shape)}, probs {tuple(probs.shape)} {probs.dtype}")
        loss = _zeros((1,), torch.float32, student.device)
        dstudent = torch.empty_like(student)
        check(_lib.load().wm_dino_loss_fwd_bwd(ptr(student), ptr(probs.contiguous()), vs, vt, b, d, temp_s, ptr(loss),
                                               ptr(dstudent), stream_ptr()), "wm_dino_loss_fwd_bwd")
        ctx.save_for_backward(dstudent)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dstudent,) = ctx.saved_tensors
        if dstudent.numel() % 8 == 0 and g.numel() == 1 and g.is_cuda:
            # the loss's upstream gradient stays on the device: one launch instead of cast, multiply, cast
            out = torch.empty_like(dstudent)
            gs = g.detach().reshape(1).to(torch.float32)
            check(_lib.load().wm_scale_bf16(ptr(dstudent), dstudent.numel(), ptr(gs), ptr(out), stream_ptr()), "wm_scale_bf16")
            return out, None, None, None, None, None
        return (dstudent.float() * g).to(torch.bfloat16), None, None, None, None, None


def dino_teacher_probs(teacher: torch.Tensor, center: torch.Tensor, temp: float) -> torch.Tensor:
    """softmax((teacher - center) / temp) per row: bf16 [rows, D] -> float32 [rows, D] (no gradient)."""
    if _precision.is_f32():
        return f32path.softmax_rows(teacher.detach(), center, 1.0 / float(temp))
    _need_cuda(teacher, "dino_teacher_probs")
    teacher = _bf16_rows(teacher.detach())
    rows, d = teacher.shape
    probs = torch.empty((rows, d), dtype=torch.float32, device=teacher.device)
    check(_lib.load().wm_dino_teacher_probs(ptr(teacher), ptr(center.reshape(-1).contiguous()), float(temp), rows, d,
                                            ptr(probs), stream_ptr()), "wm_dino_teacher_probs")
    return probs


def dino_loss(student: torch.Tensor, probs: torch.Tensor, n_student_views: int, n_teacher_views: int, batch: int,
              student_temp: float = 0.1) -> torch.Tensor:
    """Mean over (teacher view t, student view s != t) and batch of -<p_t, log_softmax(student_s / T)>."""
    if _precision.is_f32():
        return f32path.dino_loss(student, probs, int(n_student_views), int(n_teacher_views), int(batch), float(student_temp))
    return _DinoLoss.apply(student, probs, int(n_student_views), int(n_teacher_views), int(batch), float(student_temp))


def dino_center_update(center: torch.Tensor, teacher: torch.Tensor, momentum: float) -> None:
    """center <- m center + (1 - m) mean_rows(teacher), in place (float32 [D] / [1, D])."""
    if _precision.is_f32():
        return f32path.dino_center_update(center, teacher.detach(), float(momentum))
    teacher = _bf16_rows(teacher.detach())
    rows, d = teacher.shape
    check(_lib.load().wm_dino_center_update(ptr(teacher), rows, d, float(momentum), ptr(center), stream_ptr()),
          "wm_dino_center_update")


class _SoftCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, student, probs, views, b, temp_s):
        _need_cuda(student, "soft_cross_entropy")
        student = _bf16_rows(student)
        d = student.shape[1]
        if student.shape[0] != views * b or tuple(probs.shape) != (b, d) or probs.dtype != torch.float32:
            raise ValueError(f"soft_cross_entropy: student {tuple(student.shape)}, probs {tuple(probs.shape)} {probs.dtype}")
        loss = torch.zeros(1, dtype=torch.float32, device=student.device)
        ds = torch.empty_like(student)
        check(_lib.load().wm_soft_cross_entropy_fwd_bwd(ptr(student), ptr(probs.contiguous()), views, b, d, temp_s, ptr(loss),
                                                        ptr(ds), stream_ptr()), "wm_soft_cross_entropy_fwd_bwd")
        ctx.save_for_backward(ds)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (ds,) = ctx.saved_tensors
        return (ds.float() * g).to(torch.bfloat16), None, None, None, None


def soft_cross_entropy(student: torch.Tensor, probs: torch.Tensor, views: int, batch: int, temperature: float) -> torch.Tensor:
    """mean over (view, sample) of -<probs_b, log_softmax(student_vb / T)>; student bf16 [views*B, D] view-major."""
    return _SoftCE.apply(student, probs, int(views), int(batch), float(temperature))


class _MeanEntropyReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, log_prior, temperature):
        _need_cuda(logits, "mean_entropy_reg")
        logits = _bf16_rows(logits)
        n, k = logits.shape
        loss = torch.zeros(1, dtype=torch.float32, device=logits.device)
        dl = torch.empty((n, k), dtype=torch.float32, device=logits.device)
        ws = torch.empty(k, dtype=torch.float32, device=logits.device)
        check(_lib.load().wm_mean_entropy_reg_fwd_bwd(ptr(logits), ptr(log_prior), n, k, temperature, ptr(loss), ptr(dl),
                                                      ptr(ws), stream_ptr()), "wm_mean_entropy_reg_fwd_bwd")
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return (dl * g).to(torch.bfloat16), None, None


def mean_entropy_reg(logits: torch.Tensor, temperature: float, log_prior: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sum_k m_k (log m_k - log prior_k) with m the batch mean of softmax(logits / T)."""
    return _MeanEntropyReg.apply(logits, log_prior, float(temperature))
