"""Losses with lightly's call signatures, computed by the wafer_hip kernels.

NTXentLoss mirrors lightly.loss.NTXentLoss as the reference constructs and calls it
(scripts/WM811k_benchmark.py:234,246: `NTXentLoss()` -> temperature 0.5, no memory bank,
gather_distributed False; `loss = criterion(z0, z1)`)."""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn

from . import functional as F_hip


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _all_gather_rows(x: torch.Tensor) -> torch.Tensor:
    """[r, ...] per rank -> [world*r, ...] (rank-major), no gradient (the kernels provide the
    GatherLayer gradient analytically)."""
    out = [torch.empty_like(x) for _ in range(_world())]
    dist.all_gather(out, x.contiguous())
    return torch.cat(out, dim=0)


class _NTXentCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, zn, b_local, temperature, gather):
        world = _world() if gather else 1
        rank = dist.get_rank() if world > 1 else 0
        b_global = b_local * world
        if world > 1:
            zall = torch.cat([_all_gather_rows(zn[:b_local]), _all_gather_rows(zn[b_local:])], dim=0)
        else:
            zall = zn
        lse, rows = F_hip.ntxent_forward(zn, zall, b_local, b_global, rank * b_local, temperature)
        if world > 1:
            lse_all = torch.cat([_all_gather_rows(lse[:b_local]), _all_gather_rows(lse[b_local:])], dim=0)
        else:
            lse_all = lse
        ctx.save_for_backward(zn, zall, lse_all)
        ctx.meta = (b_local, b_global, rank * b_local, temperature)
        return F_hip.vector_mean(rows)

    @staticmethod
    def backward(ctx, grad_out):
        zn, zall, lse_all = ctx.saved_tensors
        b_local, b_global, off, temperature = ctx.meta
        dzn = F_hip.ntxent_backward(zn, zall, lse_all, b_local, b_global, off, temperature, 1.0 / (2 * b_local))
        return dzn * grad_out, None, None, None


class _NTXentProjections(torch.autograd.Function):
    """L2 normalisation + NT-Xent on the STACKED projections z [2 b, d] (view-major, bf16 or float32) as one autograd
    node: what `NTXentLoss.forward(z[:b], z[b:])` computes, without the slice / cat / cast round trip of two separately
    passed halves (per step: a cat, a cast, two zero-filled gradient buffers, two copies and an add from autograd's
    slice backward), and with the loss's incoming gradient applied inside the normalisation's backward kernel."""

    @staticmethod
    def forward(ctx, z, b_local, temperature, gather):
        from . import _lib
        from ._lib import check, ptr, stream_ptr

        _lib.require_gpu(z)
        z = z.contiguous()
        rows, d = z.shape
        zn = torch.empty((rows, d), dtype=torch.float32, device=z.device)
        inv = torch.empty((rows,), dtype=torch.float32, device=z.device)
        check(_lib.load().wm_l2_normalize(ptr(z), _lib.dtype_code(z), rows, d, 1e-12, ptr(zn), _lib.WM_F32, ptr(inv),
                                          stream_ptr()), "wm_l2_normalize")
        world = _world() if gather else 1
        rank = dist.get_rank() if world > 1 else 0
        b_global = b_local * world
        zall = torch.cat([_all_gather_rows(zn[:b_local]), _all_gather_rows(zn[b_local:])], dim=0) if world > 1 else zn
        lse, rows_loss = F_hip.ntxent_forward(zn, zall, b_local, b_global, rank * b_local, temperature)
        lse_all = torch.cat([_all_gather_rows(lse[:b_local]), _all_gather_rows(lse[b_local:])], dim=0) if world > 1 else lse
        ctx.save_for_backward(zn, inv, zall, lse_all)
        ctx.meta = (b_local, b_global, rank * b_local, temperature, z.dtype)
        return F_hip.vector_mean(rows_loss)

    @staticmethod
    def backward(ctx, grad_out):
        from . import _lib
        from ._lib import check, ptr, stream_ptr

        zn, inv, zall, lse_all = ctx.saved_tensors
        b_local, b_global, off, temperature, in_dtype = ctx.meta
        dzn = F_hip.ntxent_backward(zn, zall, lse_all, b_local, b_global, off, temperature, 1.0 / (2 * b_local))
        out_dtype = in_dtype if in_dtype in (torch.float32, torch.bfloat16) else torch.float32
        dz = torch.empty(zn.shape, dtype=out_dtype, device=zn.device)
        g = grad_out.detach().reshape(-1).to(torch.float32)
        check(_lib.load().wm_l2_normalize_bwd(ptr(dzn), _lib.WM_F32, ptr(zn), ptr(inv), zn.shape[0], zn.shape[1], ptr(dz),
                                              _lib.dtype_code(dz), ptr(g), stream_ptr()), "wm_l2_normalize_bwd")
        return dz.to(in_dtype), None, None, None


def stacked_views(z: torch.Tensor, b: int):
    """(z[:b], z[b:]) of the projections of two views computed in ONE pass, each half remembering the stacked tensor: a
    loss that is handed exactly these two objects (NTXentLoss without a memory bank) works on `z` itself."""
    z0, z1 = z[:b], z[b:]
    z0._hip_stacked = z1._hip_stacked = z
    return z0, z1


def _stacked_parent(out0: torch.Tensor, out1: torch.Tensor):
    p = getattr(out0, "_hip_stacked", None)
    if p is None or p is not getattr(out1, "_hip_stacked", None) or not p.is_cuda:
        return None
    b = out0.shape[0]
    if p.dim() != 2 or p.shape[0] != 2 * b or out1.shape[0] != b or not p.is_contiguous():
        return None
    if out0.data_ptr() != p.data_ptr() or out1.data_ptr() != p.data_ptr() + b * p.shape[1] * p.element_size():
        return None
    return p


class _NTXentBank(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, bank, temperature):
        from . import _lib
        from ._lib import check, ptr, stream_ptr

        b, d = q.shape
        loss = torch.zeros(1, dtype=torch.float32, device=q.device)
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        check(_lib.load().wm_ntxent_bank_fwd_bwd(ptr(q), ptr(k), ptr(bank), b, d, bank.shape[1], temperature, ptr(loss),
                                                 ptr(dq), ptr(dk), stream_ptr()), "wm_ntxent_bank_fwd_bwd")
        ctx.save_for_backward(dq, dk)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dq, dk = ctx.saved_tensors
        return dq * g, dk * g, None, None


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, weight):
        from . import _lib
        from ._lib import check, dtype_code, ptr, stream_ptr

        if logits.dtype not in (torch.float32, torch.bfloat16):
            logits = logits.float()
        logits = logits.contiguous()
        b, c = logits.shape
        acc = torch.zeros(2, dtype=torch.float32, device=logits.device)
        dl = torch.empty((b, c), dtype=torch.float32, device=logits.device)
        check(_lib.load().wm_cross_entropy_fwd_bwd(ptr(logits), dtype_code(logits), ptr(labels.to(torch.int64).contiguous()),
                                                   ptr(weight), b, c, ptr(acc), ptr(dl), stream_ptr()),
              "wm_cross_entropy_fwd_bwd")
        ctx.save_for_backward(dl, acc)
        ctx.dtype = logits.dtype
        return acc[0] / acc[1]

    @staticmethod
    def backward(ctx, g):
        dl, acc = ctx.saved_tensors
        return (dl * (g / acc[1])).to(ctx.dtype), None, None


class CrossEntropyLoss(nn.Module):
    """torch.nn.CrossEntropyLoss(weight=None, reduction="mean") on [B, C] logits and int64 labels (the
    reference's linear probe, src/ssl_wafermap/models/evals.py:20; F.nll_loss(F.log_softmax(.)) in its
    supervised baseline is the same function)."""

    def __init__(self, weight: torch.Tensor = None):
        super().__init__()
        self.register_buffer("weight", None if weight is None else torch.as_tensor(weight, dtype=torch.float32).clone())

    def forward(self, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        if logits.dim() != 2 or labels.shape != logits.shape[:1]:
            raise ValueError("CrossEntropyLoss expects [B, C] logits and [B] labels")
        return _CrossEntropy.apply(logits, labels, self.weight)


class _BCEWithLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, pos_weight):
        from . import _lib
        from ._lib import check, dtype_code, ptr, stream_ptr

        if logits.dtype not in (torch.float32, torch.bfloat16):
            logits = logits.float()
        logits = logits.contiguous()
        b, c = logits.shape
        loss = torch.zeros(1, dtype=torch.float32, device=logits.device)
        dl = torch.empty((b, c), dtype=torch.float32, device=logits.device)
        check(_lib.load().wm_bce_logits_fwd_bwd(ptr(logits), dtype_code(logits), ptr(target.float().contiguous()),
                                                ptr(pos_weight), b, c, ptr(loss), ptr(dl), stream_ptr()),
              "wm_bce_logits_fwd_bwd")
        ctx.save_for_backward(dl)
        ctx.dtype = logits.dtype
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return (dl * g).to(ctx.dtype), None, None


class BCEWithLogitsLoss(nn.Module):
    """torch.nn.BCEWithLogitsLoss(pos_weight) (the multi-label probe on MixedWM38, evals.py:93)."""

    def __init__(self, pos_weight: torch.Tensor = None):
        super().__init__()
        self.register_buffer("pos_weight",
                             None if pos_weight is None else torch.as_tensor(pos_weight, dtype=torch.float32).clone())

    def forward(self, logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if logits.shape != target.shape or logits.dim() != 2:
            raise ValueError("BCEWithLogitsLoss expects [B, C] logits and targets")
        return _BCEWithLogits.apply(logits, target, self.pos_weight)


class _NegCosine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, x1, eps):
        from . import _lib
        from ._lib import check, dtype_code, ptr, stream_ptr

        if x0.dtype != x1.dtype or x0.dtype not in (torch.float32, torch.bfloat16):
            x0, x1 = x0.float(), x1.float()
        x0, x1 = x0.contiguous(), x1.contiguous()
        b, d = x0.shape
        loss = torch.zeros(1, dtype=torch.float32, device=x0.device)
        need0, need1 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d0 = torch.empty((b, d), dtype=torch.float32, device=x0.device) if need0 else None
        d1 = torch.empty((b, d), dtype=torch.float32, device=x0.device) if need1 else None
        check(_lib.load().wm_neg_cosine_fwd_bwd(ptr(x0), ptr(x1), dtype_code(x0), b, d, eps, ptr(loss), ptr(d0), ptr(d1),
                                                stream_ptr()), "wm_neg_cosine_fwd_bwd")
        ctx.save_for_backward(d0, d1)
        ctx.dtypes = (x0.dtype, x1.dtype)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        d0, d1 = ctx.saved_tensors
        return (None if d0 is None else (d0 * g).to(ctx.dtypes[0]), None if d1 is None else (d1 * g).to(ctx.dtypes[1]),
                None)


class NegativeCosineSimilarity(nn.Module):
    """lightly.loss.NegativeCosineSimilarity (BYOL / SimSiam in the reference, scripts/WM811k_benchmark.py:446,613):
    -mean(cosine_similarity(x0, x1, dim, eps)); dim must be the feature dim of [B, D] inputs."""

    def __init__(self, dim: int = 1, eps: float = 1e-8):
        super().__init__()
        if dim != 1:
            raise NotImplementedError("NegativeCosineSimilarity: only dim=1 of [batch, dim] inputs is built")
        self.dim, self.eps = dim, float(eps)

    def forward(self, x0: torch.Tensor, x1: torch.Tensor) -> torch.Tensor:
        if x0.shape != x1.shape or x0.dim() != 2:
            raise ValueError("NegativeCosineSimilarity expects two [batch, dim] tensors of equal shape")
        return _NegCosine.apply(x0, x1, self.eps)


class NTXentLoss(nn.Module):
    """Contrastive cross-entropy (lightly.loss.NTXentLoss: same constructor and call).

    memory_bank_size = 0 (SimCLR): all pairs of the two views' batches, optionally gathered across
    ranks.  memory_bank_size > 0 (MoCo): logits [<out0, out1>, out0 . bank] / T with label 0; the bank is
    lightly's [dim, size] FIFO of L2-normalised `out1` rows, initialised with normalised Gaussian
    columns on first use and updated AFTER the negatives for this call were taken, only when out0
    requires grad."""

    def __init__(self, temperature: float = 0.5, memory_bank_size: int = 0, gather_distributed: bool = False):
        super().__init__()
        if abs(temperature) < 1e-8:
            raise ValueError(f"Illegal temperature: abs({temperature}) < 1e-8")
        if memory_bank_size < 0:
            raise ValueError(f"Illegal memory bank size {memory_bank_size}")
        self.temperature = float(temperature)
        self.gather_distributed = bool(gather_distributed)
        self.size = int(memory_bank_size)
        self.register_buffer("bank", torch.empty(0, dtype=torch.float32), persistent=False)
        self.register_buffer("bank_ptr", torch.zeros(1, dtype=torch.long), persistent=False)
        self._ptr = 0  # host copy of bank_ptr (the enqueue offsets are host-side slicing)

    @torch.no_grad()
    def _init_memory_bank(self, dim: int, device) -> None:
        bank = torch.randn(dim, self.size, device=device)
        self.bank = torch.nn.functional.normalize(bank, dim=0).contiguous()
        self.bank_ptr = torch.zeros(1, dtype=torch.long, device=device)
        self._ptr = 0

    @torch.no_grad()
    def _dequeue_and_enqueue(self, batch: torch.Tensor) -> None:
        b = batch.shape[0]
        ptr_ = self._ptr
        if ptr_ + b >= self.size:
            self.bank[:, ptr_:] = batch[: self.size - ptr_].T
            self._ptr = 0
        else:
            self.bank[:, ptr_:ptr_ + b] = batch.T
            self._ptr = ptr_ + b
        self.bank_ptr[0] = self._ptr

    def forward(self, out0: torch.Tensor, out1: torch.Tensor) -> torch.Tensor:
        if out0.shape != out1.shape or out0.dim() != 2:
            raise ValueError("NTXentLoss expects two [batch, dim] tensors of equal shape")
        b = out0.shape[0]
        if self.size > 0:
            # (lightly takes this branch whenever the bank returns negatives and never reaches its gather code then: with a
            # memory bank `gather_distributed` has no effect -- SURVEY Appendix A.1; the reference's MoCo leaves it False)
            q = F_hip.l2_normalize(out0.float().contiguous())
            k = F_hip.l2_normalize(out1.float().contiguous())
            if self.bank.numel() == 0 or self.bank.shape[0] != q.shape[1] or self.bank.device != q.device:
                self._init_memory_bank(q.shape[1], q.device)
            loss = _NTXentBank.apply(q, k, self.bank, self.temperature)  # the kernel has read the bank ...
            if out0.requires_grad:
                self._dequeue_and_enqueue(k.detach())                     # ... before this call's keys enter it
            return loss
        parent = _stacked_parent(out0, out1)
        if parent is not None and parent.dtype in (torch.float32, torch.bfloat16):
            return _NTXentProjections.apply(parent, b, self.temperature, self.gather_distributed)
        z = torch.cat([out0, out1], dim=0).float().contiguous()
        zn = F_hip.l2_normalize(z)
        return _NTXentCore.apply(zn, b, self.temperature, self.gather_distributed)


class _DCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z0, z1, temperature, sigma, weighted):
        from . import _lib
        from ._lib import check, ptr, stream_ptr

        b, d = z0.shape
        lib = _lib.load()
        need = lib.wm_dcl_workspace_bytes(b)
        ws = torch.empty(need, dtype=torch.uint8, device=z0.device)
        loss = torch.zeros(1, dtype=torch.float32, device=z0.device)
        d0, d1 = torch.empty_like(z0), torch.empty_like(z1)
        check(lib.wm_dcl_fwd_bwd(ptr(z0), ptr(z1), b, d, temperature, sigma, int(weighted), ptr(loss), ptr(d0), ptr(d1),
                                 ptr(ws), need, stream_ptr()), "wm_dcl_fwd_bwd")
        ctx.save_for_backward(d0, d1)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        d0, d1 = ctx.saved_tensors
        return d0 * g, d1 * g, None, None, None


class DCLLoss(nn.Module):
    """lightly.loss.DCLLoss (decoupled contrastive learning): the positive term is taken out of both
    denominators.  `weight_fn` is not a free callable here: DCLWLoss is the weighted form."""

    def __init__(self, temperature: float = 0.1, gather_distributed: bool = False):
        super().__init__()
        if gather_distributed and _world() > 1:
            raise NotImplementedError("DCLLoss(gather_distributed=True) is not built")
        self.temperature, self.sigma, self.weighted = float(temperature), 0.5, False

    def forward(self, out0: torch.Tensor, out1: torch.Tensor) -> torch.Tensor:
        if out0.shape != out1.shape or out0.dim() != 2 or out0.shape[0] < 2:
            raise ValueError("DCLLoss expects two [batch >= 2, dim] tensors of equal shape")
        z0 = F_hip.l2_normalize(out0.float().contiguous())
        z1 = F_hip.l2_normalize(out1.float().contiguous())
        return _DCL.apply(z0, z1, self.temperature, self.sigma, self.weighted)


class DCLWLoss(DCLLoss):
    """lightly.loss.DCLWLoss() as the reference's DCLW model builds it (scripts/WM811k_benchmark.py:265):
    temperature 0.1, negative von Mises-Fisher weights with sigma 0.5 on the positive term."""

    def __init__(self, temperature: float = 0.1, sigma: float = 0.5, gather_distributed: bool = False):
        super().__init__(temperature, gather_distributed)
        self.sigma, self.weighted = float(sigma), True


class _CrossCorrBarlow(torch.autograd.Function):
    """loss(za_n, zb_n) on the cross-correlation of two standardised [B, D] bf16 projections.  The [D, D]
    matrix is the weight-gradient GEMM of the implicit-GEMM kernel (sum over rows of za (x) zb), its
    gradient goes back through two ordinary GEMMs against the saved d loss / d matrix."""

    @staticmethod
    def forward(ctx, za, zb, scale, lambda_param, gather=False):
        from . import _lib, ops
        from ._lib import check, ptr, stream_ptr

        za, zb = za.to(torch.bfloat16).contiguous(), zb.to(torch.bfloat16).contiguous()
        b, d = za.shape
        lib = _lib.load()
        raw = ops.rows_outer_product(za, zb)  # [d][d] cross-correlation
        if gather and _world() > 1:
            # lightly: c = c / world_size; dist.all_reduce(c) -- the matrix of the GLOBAL batch (each rank standardises
            # with its own statistics); all_reduce is not differentiable, so the gradient reaches the local
            # projections through the local term only: d loss / d raw_local = d loss / d c_global * scale / world,
            # which is what the kernel returns for the summed matrix with the scale divided by the world size
            dist.all_reduce(raw)
            scale = scale / _world()
        loss = torch.zeros(1, dtype=torch.float32, device=za.device)
        draw = torch.empty_like(raw)
        check(lib.wm_barlow_twins_fwd_bwd(ptr(raw), d, scale, lambda_param, 1.0, ptr(loss), ptr(draw), stream_ptr()),
              "wm_barlow_twins_fwd_bwd")
        ctx.save_for_backward(za, zb, draw)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        from . import ops

        za, zb, draw = ctx.saved_tensors
        # raw[i][j] = sum_n za[n][i] zb[n][j]:  dza = zb @ draw^T (Linear with weight draw), dzb = za @ draw
        dza = ops.linear(zb, draw).float() * g
        dzb = ops.linear(za, draw.t().contiguous()).float() * g
        return dza.to(torch.bfloat16), dzb.to(torch.bfloat16), None, None, None


class BarlowTwinsLoss(nn.Module):
    """lightly.loss.BarlowTwinsLoss(lambda_param=5e-3): both projections are standardised over the batch
    (mean 0, unbiased std 1), c = za^T zb / N, loss = sum_i (c_ii - 1)^2 + lambda sum_{i != j} c_ij^2."""

    def __init__(self, lambda_param: float = 5e-3, gather_distributed: bool = False):
        super().__init__()
        self.lambda_param = float(lambda_param)
        self.gather_distributed = bool(gather_distributed)  # reference: scripts/WM811k_benchmark.py:364-366

    def forward(self, z_a: torch.Tensor, z_b: torch.Tensor) -> torch.Tensor:
        from . import ops

        if z_a.shape != z_b.shape or z_a.dim() != 2 or z_a.shape[0] < 2:
            raise ValueError("BarlowTwinsLoss expects two [batch >= 2, dim] tensors of equal shape")
        n, d = z_a.shape
        ones = torch.ones(d, dtype=torch.float32, device=z_a.device)
        zeros = torch.zeros(d, dtype=torch.float32, device=z_a.device)
        rm, rv = torch.zeros_like(zeros), torch.ones_like(ones)
        # batch standardisation = BatchNorm without affine, eps 0 (biased variance); the unbiased std of
        # lightly's (z - mean) / std turns into the factor (N - 1) / N on the correlation matrix
        with ops.bn_groups(1):
            za = ops.batch_norm(z_a, ones, zeros, rm, rv, True, eps=0.0)
            zb = ops.batch_norm(z_b, ones, zeros, rm.clone(), rv.clone(), True, eps=0.0)
        return _CrossCorrBarlow.apply(za, zb, (n - 1) / (n * n), self.lambda_param, self.gather_distributed)


class _VICRegBranch(torch.autograd.Function):
    """variance term * mu + covariance term * nu of ONE projection z [N, D] (bf16)."""

    @staticmethod
    def forward(ctx, z, mu, nu, eps):
        from . import _lib, ops
        from ._lib import check, dtype_code, ptr, stream_ptr

        z = z.to(torch.bfloat16).contiguous()
        n, d = z.shape
        lib = _lib.load()
        mean = torch.zeros(d, dtype=torch.float32, device=z.device)
        var = torch.zeros(d, dtype=torch.float32, device=z.device)
        check(lib.wm_colstats(ptr(z), dtype_code(z), n, d, ptr(mean), ptr(var), stream_ptr()), "wm_colstats")
        zc = torch.empty_like(z)
        check(lib.wm_center_columns(ptr(z), ptr(mean), n, d, ptr(zc), stream_ptr()), "wm_center_columns")
        vloss = torch.zeros(1, dtype=torch.float32, device=z.device)
        coef = torch.empty(d, dtype=torch.float32, device=z.device)
        check(lib.wm_vicreg_variance(ptr(var), n, d, eps, ptr(vloss), ptr(coef), stream_ptr()), "wm_vicreg_variance")
        raw = ops.rows_outer_product(zc, zc)  # [d][d] covariance
        closs = torch.zeros(1, dtype=torch.float32, device=z.device)
        draw = torch.empty_like(raw)
        check(lib.wm_barlow_twins_fwd_bwd(ptr(raw), d, 1.0 / (n - 1), 1.0 / d, 0.0, ptr(closs), ptr(draw), stream_ptr()),
              "wm_barlow_twins_fwd_bwd(covariance)")
        ctx.save_for_backward(zc, coef, draw)
        ctx.w = (mu, nu)
        return mu * vloss[0] + nu * closs[0]

    @staticmethod
    def backward(ctx, g):
        from . import ops

        zc, coef, draw = ctx.saved_tensors
        mu, nu = ctx.w
        # covariance: raw = zc^T zc -> d/dzc = zc @ (draw + draw^T) = 2 zc @ draw (draw is symmetric)
        dzc = 2.0 * nu * ops.linear(zc, draw).float() + mu * coef * zc.float()
        dz = dzc - dzc.mean(dim=0, keepdim=True)  # through the centring
        return (dz * g).to(torch.bfloat16), None, None, None


class VICRegLoss(nn.Module):
    """lightly.loss.VICRegLoss(lambda_param=25, mu_param=25, nu_param=1, eps=1e-4): invariance (MSE between the
    branches) + variance hinge on the per-dimension std + squared off-diagonal covariance."""

    def __init__(self, lambda_param: float = 25.0, mu_param: float = 25.0, nu_param: float = 1.0,
                 gather_distributed: bool = False, eps: float = 0.0001):
        super().__init__()
        if gather_distributed and _world() > 1:
            raise NotImplementedError("VICRegLoss(gather_distributed=True) is not built")
        self.lambda_param, self.mu_param, self.nu_param, self.eps = lambda_param, mu_param, nu_param, eps

    def forward(self, z_a: torch.Tensor, z_b: torch.Tensor) -> torch.Tensor:
        from . import vit_ops

        if z_a.shape != z_b.shape or z_a.dim() != 2 or z_a.shape[0] < 2:
            raise ValueError("VICRegLoss expects two [batch >= 2, dim] tensors of equal shape")
        inv = vit_ops.mse_loss(z_a, z_b)
        branch = _VICRegBranch.apply(z_a, 0.5 * self.mu_param, self.nu_param, self.eps) + \
            _VICRegBranch.apply(z_b, 0.5 * self.mu_param, self.nu_param, self.eps)
        return self.lambda_param * inv + branch


def sinkhorn(out: torch.Tensor, iterations: int = 3, epsilon: float = 0.05, gather_distributed: bool = False) -> torch.Tensor:
    """lightly.loss.swav_loss.sinkhorn: [B, K] prototype scores -> [B, K] float32 assignment (rows sum to 1)."""
    from . import _lib
    from ._lib import check, dtype_code, ptr, stream_ptr

    x = out.detach()
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    x = x.contiguous()
    b_local = x.shape[0]
    gathered = gather_distributed and _world() > 1
    if gathered:
        # lightly all-reduces the total and the per-prototype sums of every iteration: that IS the Sinkhorn iteration
        # on the global batch's score matrix (the per-sample normalisation is local to a column).  One all-gather of
        # the [B, K] scores (0.8 MB at 64 x 3000) instead of 1 + iterations small all-reduces inside the loop, then
        # this rank's rows of the global assignment.
        x = _all_gather_rows(x).contiguous()
    b, k = x.shape
    q = torch.empty((b, k), dtype=torch.float32, device=x.device)
    ws = torch.empty(b + k, dtype=torch.float32, device=x.device)
    check(_lib.load().wm_sinkhorn(ptr(x), dtype_code(x), b, k, float(epsilon), int(iterations), ptr(q), ptr(ws),
                                  stream_ptr()), "wm_sinkhorn")
    if gathered:
        r = dist.get_rank()
        q = q[r * b_local:(r + 1) * b_local].contiguous()
    return q


class SwaVLoss(nn.Module):
    """lightly.loss.SwaVLoss(temperature=0.1, sinkhorn_iterations=3, sinkhorn_epsilon=0.05): for every
    high-resolution crop i the Sinkhorn assignment q_i of its prototype scores is the target of all OTHER
    crops' softmax(scores / T); mean over the pairs and the batch.  That is the DINO loss kernel with q_i as
    the teacher probabilities (same pairing rule, same normalisation)."""

    def __init__(self, temperature: float = 0.1, sinkhorn_iterations: int = 3, sinkhorn_epsilon: float = 0.05,
                 sinkhorn_gather_distributed: bool = False):
        super().__init__()
        self.sinkhorn_gather_distributed = bool(sinkhorn_gather_distributed)  # reference: WM811k_benchmark.py:834-836
        self.temperature = temperature
        self.sinkhorn_iterations = sinkhorn_iterations
        self.sinkhorn_epsilon = sinkhorn_epsilon

    def forward(self, high_resolution_outputs, low_resolution_outputs, queue_outputs=None):
        from . import vit_ops

        if queue_outputs is not None:
            raise NotImplementedError("SwaVLoss: queue_outputs is not built (the reference passes none)")
        high, low = list(high_resolution_outputs), list(low_resolution_outputs)
        b = high[0].shape[0]
        probs = torch.cat([sinkhorn(h, self.sinkhorn_iterations, self.sinkhorn_epsilon, self.sinkhorn_gather_distributed)
                           for h in high], dim=0)
        student = torch.cat(high + low, dim=0)
        return vit_ops.dino_loss(student, probs, len(high) + len(low), len(high), b, self.temperature)


class MSNLoss(nn.Module):
    """lightly.loss.MSNLoss (Masked Siamese Networks; the reference's MSN, scripts/WM811k_benchmark.py:684,705):
    anchors [V*B, D] (view-major), targets [B, D], prototypes [K, D].  Targets: softmax of the prototype
    cosines / T, sharpened (power 1 / target_sharpen_temperature) and balanced by Sinkhorn; loss = mean
    cross-entropy of the anchors' prototype distribution against them + regularization_weight * sum m log m
    (mean-entropy maximisation) with m the mean anchor distribution.  Gradients reach the anchors only,
    through every term (the reference passes `prototypes.data`)."""

    def __init__(self, temperature: float = 0.1, sinkhorn_iterations: int = 3, regularization_weight: float = 1.0,
                 gather_distributed: bool = False):
        super().__init__()
        self.gather_distributed = bool(gather_distributed)  # lightly: the target probabilities' Sinkhorn is global
        self.temperature = temperature
        self.sinkhorn_iterations = sinkhorn_iterations
        self.regularization_weight = regularization_weight

    def _log_prior(self, k: int, device):
        return None

    def forward(self, anchors: torch.Tensor, targets: torch.Tensor, prototypes: torch.Tensor,
                target_sharpen_temperature: float = 0.25) -> torch.Tensor:
        from . import ops, vit_ops

        b, k = targets.shape[0], prototypes.shape[0]
        views = anchors.shape[0] // b
        if anchors.shape[0] != views * b or anchors.shape[1] != targets.shape[1] != prototypes.shape[1]:
            raise ValueError("MSNLoss: anchors [V*B, D], targets [B, D], prototypes [K, D]")
        protos = F_hip.l2_normalize(prototypes.detach().float().contiguous())
        a = F_hip.l2_normalize(anchors.float().contiguous())
        with torch.no_grad():
            t = F_hip.l2_normalize(targets.detach().float().contiguous())
            t_cos = ops.linear(t, protos)  # [B, K] cosines (bf16)
            # sharpen(softmax(cos / T), Ts) = softmax(cos / (T Ts)); Sinkhorn works on it up to row factors
            if self.sinkhorn_iterations > 0:
                q = sinkhorn(t_cos.float() / (self.temperature * target_sharpen_temperature), self.sinkhorn_iterations, 1.0,
                             self.gather_distributed)
            else:
                q = vit_ops.dino_teacher_probs(t_cos, torch.zeros(k, device=t.device),
                                               self.temperature * target_sharpen_temperature)
        a_cos = ops.linear(a, protos)   # [V*B, K], gradient to the anchors through the GEMM's dgrad
        loss = vit_ops.soft_cross_entropy(a_cos, q, views, b, self.temperature)
        if self.regularization_weight > 0:
            loss = loss + self.regularization_weight * vit_ops.mean_entropy_reg(a_cos, self.temperature,
                                                                                self._log_prior(k, a.device))
        return loss


class PMSNLoss(MSNLoss):
    """lightly.loss.PMSNLoss: the regulariser is the KL divergence of the mean anchor distribution to a
    power-law prior over the prototypes, prior_k proportional to 1 / k^power_law_exponent (k = 1..K)."""

    def __init__(self, temperature: float = 0.1, sinkhorn_iterations: int = 3, regularization_weight: float = 1.0,
                 power_law_exponent: float = 0.25, gather_distributed: bool = False):
        super().__init__(temperature, sinkhorn_iterations, regularization_weight, gather_distributed)
        self.power_law_exponent = power_law_exponent

    def _log_prior(self, k: int, device):
        prior = 1.0 / torch.arange(1, k + 1, dtype=torch.float64) ** self.power_law_exponent
        return torch.log(prior / prior.sum()).float().to(device)


class DINOLoss(nn.Module):
    """lightly.loss.DINOLoss as the reference calls it (scripts/WM811k_benchmark.py:564,586:
    `DINOLoss(output_dim=2048)`, `criterion(teacher_out, student_out, epoch=...)`).

    teacher_out / student_out: lists of [B, D] tensors (the reference's form) or already stacked
    view-major [V*B, D] tensors with `batch=B`.  Views with the same list index are the same crop
    and are skipped (lightly zeroes the diagonal of the [teacher view, student view] loss matrix).
    The centre is updated after the loss, from this step's teacher outputs; under
    torch.distributed the batch mean is all-reduced first, as lightly does."""

    def __init__(self, output_dim: int = 65536, warmup_teacher_temp: float = 0.04, teacher_temp: float = 0.04,
                 warmup_teacher_temp_epochs: int = 30, student_temp: float = 0.1, center_momentum: float = 0.9):
        super().__init__()
        self.warmup_teacher_temp_epochs = warmup_teacher_temp_epochs
        self.teacher_temp = teacher_temp
        self.student_temp = student_temp
        self.center_momentum = center_momentum
        self.register_buffer("center", torch.zeros(1, 1, output_dim))
        self._center_mean = None      # graph-replayed data-parallel steps: local batch centre awaiting its all-reduce
        self._center_pending = False
        self.teacher_temp_schedule = torch.linspace(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs)

    def forward(self, teacher_out, student_out, epoch: int, batch: int = None):
        from . import vit_ops

        if isinstance(teacher_out, (list, tuple)):
            vt, batch = len(teacher_out), teacher_out[0].shape[0]
            teacher = torch.cat([t.detach() for t in teacher_out], dim=0)
        else:
            teacher, vt = teacher_out.detach(), teacher_out.shape[0] // batch
        if isinstance(student_out, (list, tuple)):
            vs = len(student_out)
            student = torch.cat(list(student_out), dim=0)
        else:
            student, vs = student_out, student_out.shape[0] // batch
        temp = float(self.teacher_temp_schedule[epoch]) if epoch < self.warmup_teacher_temp_epochs else self.teacher_temp
        probs = vit_ops.dino_teacher_probs(teacher, self.center, temp)
        loss = vit_ops.dino_loss(student, probs, vs, vt, batch, self.student_temp)
        self.update_center(teacher)
        return loss

    @torch.no_grad()
    def finish_center_update(self) -> None:
        """The deferred half of update_center for graph-replayed data-parallel steps (see there)."""
        if not getattr(self, "_center_pending", False) or self._center_mean is None:
            return
        mean = self._center_mean.clone()
        dist.all_reduce(mean)
        mean /= _world()
        self.center.mul_(self.center_momentum).add_(mean, alpha=1 - self.center_momentum)

    @torch.no_grad()
    def update_center(self, teacher: torch.Tensor) -> None:
        from . import vit_ops

        if _world() > 1:
            # lightly: batch_center = mean over (views, batch), all-reduced and divided by world size
            mean = teacher.float().mean(dim=0, keepdim=True)
            if self._center_mean is None:
                # allocated by the first EAGER step (the capture's warm-up): not from a graph's private pool
                self._center_mean = torch.zeros_like(self.center)
            if torch.cuda.is_available() and teacher.is_cuda and torch.cuda.is_current_stream_capturing():
                # inside a hipGraph capture (graph.GraphedTrainStep): the collective cannot be part of the graph (gloo
                # is a host call; RCCL inside a captured step is not relied on).  The local mean is parked in a
                # persistent buffer by the graph; finish_center_update() -- called by the step after the replay --
                # does the exchange and the moving average.
                self._center_mean.copy_(mean.view_as(self.center))
                self._center_pending = True   # stays set: every REPLAY parks a new mean without running this code again
                return
            self._center_pending = False      # an eager step: nothing is parked
            dist.all_reduce(mean)
            mean /= _world()
            self.center.mul_(self.center_momentum).add_(mean.view_as(self.center), alpha=1 - self.center_momentum)
        else:
            vit_ops.dino_center_update(self.center, teacher, self.center_momentum)
