"""CPU oracle for NT-Xent.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates lightly.loss.NTXentLoss (third-party `lightly`, UNPINNED, not installed here) from its
published algorithm (SimCLR, Chen et al. 2020, arXiv:2002.05709, with lightly's logits layout, see
SURVEY.md Appendix A.1), anchored on the reference call site scripts/WM811k_benchmark.py:234,246.
Parity status: PARITY UNPINNED upstream; cross-checked against a float64 closed form (tests/).
"""
import torch
import torch.nn.functional as F


def ntxent_lightly(out0, out1, temperature=0.5, out0_large=None, out1_large=None, rank=0):
    """The [2B, 2B-1] logits construction, step by step as lightly does it.  With *_large given
    (all-gathered, rank-major) this is the gather_distributed=True branch for `rank`."""
    batch_size = out0.shape[0]
    out0 = F.normalize(out0, dim=1)
    out1 = F.normalize(out1, dim=1)
    if out0_large is None:
        out0_large, out1_large = out0, out1
        diag_mask = torch.eye(batch_size, dtype=torch.bool, device=out0.device)
    else:
        out0_large = F.normalize(out0_large, dim=1)
        out1_large = F.normalize(out1_large, dim=1)
        world = out0_large.shape[0] // batch_size
        diag_mask = torch.zeros(batch_size, batch_size * world, dtype=torch.bool, device=out0.device)
        diag_mask[:, rank * batch_size : (rank + 1) * batch_size] = torch.eye(batch_size, dtype=torch.bool, device=out0.device)
    logits_00 = torch.einsum("nc,mc->nm", out0, out0_large) / temperature
    logits_01 = torch.einsum("nc,mc->nm", out0, out1_large) / temperature
    logits_10 = torch.einsum("nc,mc->nm", out1, out0_large) / temperature
    logits_11 = torch.einsum("nc,mc->nm", out1, out1_large) / temperature
    logits_00 = logits_00[~diag_mask].view(batch_size, -1)
    logits_11 = logits_11[~diag_mask].view(batch_size, -1)
    logits_0100 = torch.cat([logits_01, logits_00], dim=1)
    logits_1011 = torch.cat([logits_10, logits_11], dim=1)
    logits = torch.cat([logits_0100, logits_1011], dim=0)
    labels = torch.arange(batch_size, dtype=torch.long, device=out0.device) + rank * batch_size
    labels = labels.repeat(2)
    return F.cross_entropy(logits, labels, reduction="mean")


def ntxent_closed_form_f64(out0, out1, temperature=0.5):
    z = F.normalize(torch.cat([out0, out1]).double(), dim=1)
    n = z.shape[0]
    b = n // 2
    s = z @ z.t() / temperature
    s.fill_diagonal_(float("-inf"))
    pos = torch.cat([torch.arange(b, n), torch.arange(0, b)])
    return (torch.logsumexp(s, dim=1) - s[torch.arange(n), pos]).mean()


def ntxent_memory_bank(out0, out1, bank, temperature=0.1):
    """lightly NTXentLoss.forward with memory_bank_size > 0 (the reference's MoCo,
    scripts/WM811k_benchmark.py:305-307,337-338): positives <out0, out1>, negatives out0 . bank [D, K],
    logits [B, 1 + K] / T, labels 0."""
    out0 = F.normalize(out0, dim=1)
    out1 = F.normalize(out1, dim=1)
    sim_pos = torch.einsum("nc,nc->n", out0, out1).unsqueeze(-1)
    sim_neg = torch.einsum("nc,ck->nk", out0, bank)
    logits = torch.cat([sim_pos, sim_neg], dim=1) / temperature
    labels = torch.zeros(logits.shape[0], dtype=torch.long, device=out0.device)
    return F.cross_entropy(logits, labels, reduction="mean")


def memory_bank_enqueue(bank, ptr, batch):
    """lightly MemoryBankModule._dequeue_and_enqueue: returns the new pointer; writes normalised keys as
    columns; a batch that reaches the end fills the tail and wraps the pointer to 0."""
    size = bank.shape[1]
    b = batch.shape[0]
    if ptr + b >= size:
        bank[:, ptr:] = batch[: size - ptr].T.detach()
        return 0
    bank[:, ptr:ptr + b] = batch.T.detach()
    return ptr + b


def dcl_loss(out0, out1, temperature=0.1, sigma=None):
    """lightly DCLLoss (sigma None) / DCLWLoss (negative von Mises-Fisher weights, sigma 0.5), no gathering:
    per direction -w <a_i,b_i>/T + logsumexp_{k!=i} <a_i,a_k>/T + logsumexp_{k!=i} <a_i,b_k>/T, mean over rows,
    mean of the two directions.  Restated from the DCL paper (Yeh et al. 2021, arXiv:2110.06848, eq. 6-7) and
    lightly's layout; PARITY UNPINNED upstream."""
    out0 = F.normalize(out0, dim=1)
    out1 = F.normalize(out1, dim=1)

    def weights(a, b):
        if sigma is None:
            return 1.0
        sim = torch.einsum("nm,nm->n", a.detach(), b.detach()) / sigma
        return 2 - a.shape[0] * F.softmax(sim, dim=0)

    def one(a, b):
        n = a.shape[0]
        sim_aa = a @ a.t() / temperature
        sim_ab = a @ b.t() / temperature
        positive = -sim_ab.diagonal() * weights(a, b)
        eye = torch.eye(n, dtype=torch.bool, device=a.device)
        neg_aa = torch.logsumexp(sim_aa[~eye].view(n, -1), dim=1)
        neg_ab = torch.logsumexp(sim_ab[~eye].view(n, -1), dim=1)
        return (positive + neg_aa + neg_ab).mean()

    return 0.5 * (one(out0, out1) + one(out1, out0))


def barlow_twins_loss(z_a, z_b, lambda_param=5e-3):
    """lightly BarlowTwinsLoss (Zbontar et al. 2021, arXiv:2103.03230 Algorithm 1 with lightly's scaling):
    standardise over the batch with the unbiased std, c = z_a^T z_b / N, on-diagonal (c_ii - 1)^2 plus
    lambda times the squared off-diagonal.  PARITY UNPINNED upstream."""
    n, d = z_a.shape
    za = (z_a - z_a.mean(0)) / z_a.std(0)
    zb = (z_b - z_b.mean(0)) / z_b.std(0)
    c = za.t() @ zb / n
    on = (torch.diagonal(c) - 1).pow(2).sum()
    off = c.pow(2).sum() - torch.diagonal(c).pow(2).sum()
    return on + lambda_param * off


def vicreg_loss(z_a, z_b, lambda_param=25.0, mu_param=25.0, nu_param=1.0, eps=1e-4):
    """lightly VICRegLoss (Bardes et al. 2022, arXiv:2105.04906, with lightly's normalisations): MSE invariance,
    hinge on sqrt(var + eps) averaged over the two branches, squared off-diagonal covariance / D summed over
    the branches.  PARITY UNPINNED upstream."""
    def variance(x):
        return torch.mean(F.relu(1.0 - torch.sqrt(x.var(dim=0) + eps)))

    def covariance(x):
        x = x - x.mean(dim=0)
        n, d = x.shape
        cov = x.t() @ x / (n - 1)
        off = cov.pow(2).sum() - torch.diagonal(cov).pow(2).sum()
        return off / d

    return (lambda_param * F.mse_loss(z_a, z_b) + mu_param * 0.5 * (variance(z_a) + variance(z_b))
            + nu_param * (covariance(z_a) + covariance(z_b)))


def sinkhorn(out, iterations=3, epsilon=0.05):
    """lightly.loss.swav_loss.sinkhorn (single process), after Caron et al. 2020 (arXiv:2006.09882, Listing 1)."""
    q = torch.exp(out / epsilon).t()
    q = q / q.sum()
    k, b = q.shape
    for _ in range(iterations):
        q = q / q.sum(dim=1, keepdim=True) / k
        q = q / q.sum(dim=0, keepdim=True) / b
    return (q * b).t()


def swav_loss(high, low, temperature=0.1, iterations=3, epsilon=0.05):
    """lightly SwaVLoss.forward without a queue.  PARITY UNPINNED upstream."""
    n_crops = len(high) + len(low)
    loss = 0.0
    for i in range(len(high)):
        with torch.no_grad():
            q = sinkhorn(high[i].detach(), iterations, epsilon)
        sub = 0.0
        for v in range(len(high)):
            if v != i:
                sub = sub - torch.mean(torch.sum(q * F.log_softmax(high[v] / temperature, dim=1), dim=1))
        for v in range(len(low)):
            sub = sub - torch.mean(torch.sum(q * F.log_softmax(low[v] / temperature, dim=1), dim=1))
        loss = loss + sub / (n_crops - 1)
    return loss / len(high)


def msn_loss(anchors, targets, prototypes, temperature=0.1, sinkhorn_iterations=3, regularization_weight=1.0,
             target_sharpen_temperature=0.25, power_law_exponent=None):
    """lightly MSNLoss / PMSNLoss (Assran et al. 2022, arXiv:2204.07141; PMSN: Assran et al. 2023, arXiv:2210.07277),
    single process.  power_law_exponent None -> MSN's mean-entropy maximisation, else PMSN's KL to the power-law
    prior.  PARITY UNPINNED upstream."""
    num_views = anchors.shape[0] // targets.shape[0]
    anchors = F.normalize(anchors, dim=1)
    targets = F.normalize(targets, dim=1)
    prototypes = F.normalize(prototypes, dim=1)
    anchor_probs = F.softmax(anchors @ prototypes.t() / temperature, dim=1)
    with torch.no_grad():
        tp = F.softmax(targets @ prototypes.t() / temperature, dim=1)
        tp = tp ** (1.0 / target_sharpen_temperature)
        tp = tp / tp.sum(dim=1, keepdim=True)
        if sinkhorn_iterations > 0:
            q = tp.t()
            q = q / q.sum()
            k, b = q.shape
            for _ in range(sinkhorn_iterations):
                q = q / q.sum(dim=1, keepdim=True) / k
                q = q / q.sum(dim=0, keepdim=True) / b
            tp = (q * b).t()
        tp = tp.repeat(num_views, 1)  # view-major: all samples of view 0, then view 1, ...
    loss = torch.mean(torch.sum(-tp * torch.log(anchor_probs), dim=1))
    if regularization_weight > 0:
        m = anchor_probs.mean(dim=0)
        if power_law_exponent is None:
            reg = torch.sum(m * torch.log(m))
        else:
            prior = 1.0 / torch.arange(1, m.shape[0] + 1, dtype=torch.float64) ** power_law_exponent
            prior = (prior / prior.sum()).to(m.dtype)
            reg = torch.sum(m * (torch.log(m) - torch.log(prior)))
        loss = loss + regularization_weight * reg
    return loss
