"""World-size-2 gloo tests (CPU) of the data-parallel host logic: gradient bucketing, the
gather_distributed plumbing of NTXentLoss (kernels replaced by a torch stand-in that implements
the SAME contract as wm_ntxent_fwd/bwd), and rank slicing of the loader."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ntxent as on


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, world, *args):
    port = _free_port()
    mp.spawn(_entry, args=(world, port, fn, args), nprocs=world, join=True)


def _entry(rank, world, port, fn, args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world, *args)
    finally:
        dist.destroy_process_group()


# ---- torch stand-ins with the kernels' contract (include/wafer_hip.h: wm_ntxent_fwd / wm_ntxent_bwd)
def _ids(b_local, b_global, off):
    v = torch.arange(2).repeat_interleave(b_local)
    i = torch.arange(b_local).repeat(2)
    return v * b_global + off + i, (1 - v) * b_global + off + i


def ref_ntxent_forward(zn, zall, b_local, b_global, off, t):
    self_g, pos_g = _ids(b_local, b_global, off)
    s = zn @ zall.t() / t
    pos = s[torch.arange(2 * b_local), pos_g]
    s = s.clone()
    s[torch.arange(2 * b_local), self_g] = float("-inf")
    lse = torch.logsumexp(s, dim=1)
    return lse, lse - pos


def ref_ntxent_backward(zn, zall, lse_all, b_local, b_global, off, t, grad_scale):
    self_g, pos_g = _ids(b_local, b_global, off)
    s = zn @ zall.t() / t
    m = torch.exp(s - lse_all[self_g][:, None]) + torch.exp(s - lse_all[None, :])
    m[torch.arange(2 * b_local), pos_g] -= 2.0
    m[torch.arange(2 * b_local), self_g] = 0.0
    return grad_scale / t * (m @ zall)


def _grad_sync(rank, world):
    from ssl_wafermap_amd import distributed as wdist

    class FakeOpt:
        grad_scale = 1.0

        def __init__(self):
            g = torch.Generator().manual_seed(rank)
            self.grad_arenas = [torch.randn(1000, generator=g), torch.randn(37, generator=g)]

    opt = FakeOpt()
    mine = [a.clone() for a in opt.grad_arenas]
    sync = wdist.GradSync(opt, bucket_bytes=256 * 4)  # several buckets + a ragged tail
    assert opt.grad_scale == 1.0 / world
    sync.start()
    sync.wait()
    other = [torch.randn(1000, generator=torch.Generator().manual_seed(1 - rank)), None]
    g = torch.Generator().manual_seed(1 - rank)
    other = [torch.randn(1000, generator=g), torch.randn(37, generator=g)]
    for a, m, o in zip(opt.grad_arenas, mine, other):
        torch.testing.assert_close(a, m + o)
    assert wdist.world_size() == world and wdist.rank() == rank


def test_grad_sync_buckets_world2():
    _spawn(_grad_sync, 2)


def _ntxent_gathered(rank, world):
    import ssl_wafermap_amd.functional as F_hip
    from ssl_wafermap_amd.loss import NTXentLoss

    # replace the three kernel entry points by contract-equivalent torch code (no GPU here)
    F_hip.ntxent_forward = ref_ntxent_forward
    F_hip.ntxent_backward = ref_ntxent_backward
    F_hip.l2_normalize = lambda x, eps=1e-12, out_dtype=None: torch.nn.functional.normalize(x, dim=1, eps=eps)
    F_hip.vector_mean = lambda x, scale=1.0, sqrt_of=False: (torch.sqrt(scale * x) if sqrt_of else x).mean()
    b, d, t = 6, 32, 0.5
    g = torch.Generator().manual_seed(0)
    z0_all, z1_all = torch.randn(world * b, d, generator=g), torch.randn(world * b, d, generator=g)
    z0 = z0_all[rank * b:(rank + 1) * b].clone().requires_grad_(True)
    z1 = z1_all[rank * b:(rank + 1) * b].clone().requires_grad_(True)
    loss = NTXentLoss(temperature=t, gather_distributed=True)(z0, z1)
    loss.backward()
    # reference: lightly's gathered branch; gradients summed over ranks (GatherLayer.backward)
    r0, r1 = z0_all.clone().requires_grad_(True), z1_all.clone().requires_grad_(True)
    losses = [on.ntxent_lightly(r0[k * b:(k + 1) * b], r1[k * b:(k + 1) * b], t, r0, r1, rank=k) for k in range(world)]
    torch.stack(losses).sum().backward()
    assert abs(loss.item() - losses[rank].item()) < 1e-5
    torch.testing.assert_close(z0.grad, r0.grad[rank * b:(rank + 1) * b], atol=1e-6, rtol=1e-4)
    torch.testing.assert_close(z1.grad, r1.grad[rank * b:(rank + 1) * b], atol=1e-6, rtol=1e-4)
    # and the mean over ranks of the gathered losses is the single-process loss on the global batch
    full = on.ntxent_lightly(z0_all, z1_all, t)
    tot = loss.detach().clone()
    dist.all_reduce(tot)
    assert abs(tot.item() / world - full.item()) < 1e-5


def test_ntxent_gather_distributed_world2():
    _spawn(_ntxent_gathered, 2)


def test_contract_stand_in_matches_autograd_single_process():
    """The torch stand-in used above obeys the documented kernel contract (and so do the kernels:
    tests/test_gpu_embed.py checks them against the same oracle on the GPU)."""
    g = torch.Generator().manual_seed(3)
    b, d, t = 10, 16, 0.5
    z0, z1 = torch.randn(b, d, generator=g), torch.randn(b, d, generator=g)
    zn = torch.nn.functional.normalize(torch.cat([z0, z1]), dim=1)
    lse, rows = ref_ntxent_forward(zn, zn, b, b, 0, t)
    ref = on.ntxent_lightly(z0, z1, t)
    assert abs(rows.mean() - ref) < 1e-6
    znr = zn.clone().requires_grad_(True)
    on.ntxent_lightly(znr[:b], znr[b:], t).backward()
    dzn = ref_ntxent_backward(zn, zn, lse, b, b, 0, t, 1.0 / (2 * b))
    proj = dzn - zn * (dzn * zn).sum(1, keepdim=True)
    torch.testing.assert_close(proj, znr.grad, atol=1e-6, rtol=1e-4)


def test_loader_rank_slices_partition_the_global_batch():
    from ssl_wafermap_amd.data.dataset import WaferLoader

    class FakeDataset:
        def __len__(self):
            return 50

        def get_batch(self, indices, rng, fmt="nhwc_bf16"):
            return [torch.as_tensor(indices)], torch.as_tensor(indices)

    ds = FakeDataset()
    full = [y for _, y in WaferLoader(ds, 8, shuffle=True, drop_last=True, seed=3)]
    parts = [[y for _, y in WaferLoader(ds, 4, shuffle=True, drop_last=True, seed=3, rank=r, world_size=2)] for r in range(2)]
    assert len(full) == 6 == len(parts[0]) == len(parts[1])
    for f, a, b in zip(full, parts[0], parts[1]):
        assert torch.equal(torch.cat([a, b]), f)
    seen = torch.cat(full)
    assert len(set(seen.tolist())) == 48  # a permutation without repeats
    again = [y for _, y in WaferLoader(ds, 8, shuffle=True, drop_last=True, seed=3)]
    assert all(torch.equal(a, b) for a, b in zip(full, again))
    nxt = WaferLoader(ds, 8, shuffle=True, drop_last=True, seed=3)
    nxt.set_epoch(1)
    assert not torch.equal(torch.cat([y for _, y in nxt]), seen)


def test_loader_short_final_batch_keeps_every_rank_in_step():
    """drop_last=False with a final global batch shorter than one rank's share (n % (bs * world) < bs): every
    rank still gets a batch of the same size (wrap-around padding, as torch's DistributedSampler), so no rank
    skips a step and its collectives."""
    from ssl_wafermap_amd.data.dataset import WaferLoader

    class FakeDataset:
        def __len__(self):
            return 35  # 35 % (8 * 2) = 3 < 8

        def get_batch(self, indices, rng, fmt="nhwc_bf16"):
            return [torch.as_tensor(indices)], torch.as_tensor(indices)

    parts = [[y for _, y in WaferLoader(FakeDataset(), 8, shuffle=False, drop_last=False, rank=r, world_size=2)]
             for r in range(2)]
    assert len(parts[0]) == len(parts[1]) == 3
    assert [len(y) for y in parts[0]] == [len(y) for y in parts[1]] == [8, 8, 2]
    seen = torch.cat(parts[0] + parts[1])
    assert set(seen.tolist()) == set(range(35))          # nothing lost; one sample repeated as padding
    assert len(seen) == 36


def _broadcast_state(rank, world):
    from ssl_wafermap_amd import distributed as wdist

    torch.manual_seed(rank)  # replicas constructed from DIFFERENT seeds
    model = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.BatchNorm1d(8), torch.nn.Linear(8, 4))
    model[1].running_mean.fill_(float(rank + 1))
    teacher = torch.nn.Linear(8, 8)
    for p in teacher.parameters():
        p.requires_grad = False
    model.add_module("teacher", teacher)

    class Arena:
        pass

    class FakeOpt:  # the flat-arena layout of optim._Arena on CPU tensors
        def __init__(self, params):
            a = Arena()
            n = sum(p.numel() for p in params)
            a.params = torch.zeros(n)
            o = 0
            for p in params:
                a.params[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = a.params[o:o + p.numel()].view(p.shape)
                o += p.numel()
            self._arenas = [a]

    opt = FakeOpt([p for p in model.parameters() if p.requires_grad])
    wdist.broadcast_state(model, opt)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()] + [b.reshape(-1).float() for b in model.buffers()])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1])
    assert float(model[1].running_mean[0]) == 1.0 and model[0].weight.data_ptr() == opt._arenas[0].params.data_ptr()


def test_broadcast_state_aligns_replicas_world2():
    _spawn(_broadcast_state, 2)


def _sharded_knn(rank, world):
    import ssl_wafermap_amd.functional as F_hip
    from ssl_wafermap_amd import distributed as wdist

    def ref_topk(query, bank, k, index_base=0):
        sim = query @ bank.t()
        s, i = sim.topk(k, dim=1)
        return s, (i + index_base).to(torch.int32)

    def ref_merge(sims, idxs):
        parts, nq, k = sims.shape
        s = sims.permute(1, 0, 2).reshape(nq, parts * k)
        i = idxs.permute(1, 0, 2).reshape(nq, parts * k)
        top, pos = s.topk(k, dim=1)
        return top, torch.gather(i, 1, pos)

    F_hip.knn_topk, F_hip.knn_merge = ref_topk, ref_merge  # contract-equivalent stand-ins (no GPU here)
    g = torch.Generator().manual_seed(0)
    bank = torch.nn.functional.normalize(torch.randn(101, 16, generator=g), dim=1)
    q = torch.nn.functional.normalize(torch.randn(7, 16, generator=g), dim=1)
    cut = [0, 60, 101]                                     # unequal shards
    sim, idx = wdist.sharded_knn_topk(q, bank[cut[rank]:cut[rank + 1]], 5, cut[rank])
    want_s, want_i = ref_topk(q, bank, 5)
    torch.testing.assert_close(sim, want_s)
    assert torch.equal(idx, want_i)
    # a shard smaller than k: padded lists (-inf, out-of-range index) must lose the merge
    cut = [0, 3, 101]
    sim, idx = wdist.sharded_knn_topk(q, bank[cut[rank]:cut[rank + 1]], 5, cut[rank])
    assert torch.equal(idx, want_i) and int(idx.max()) < 101


def test_sharded_knn_topk_equals_unsharded_world2():
    _spawn(_sharded_knn, 2)


# ---- sharded kNN evaluation on the product path (models/knn.py, utils/benchmarking.KNNClassifier)
def _knn_stand_ins():
    """Contract-equivalent torch code for wm_l2_normalize / wm_knn_topk / wm_knn_vote (no GPU here)."""
    import ssl_wafermap_amd.functional as F_hip

    def ref_l2(x, eps=1e-12, out_dtype=None):
        return torch.nn.functional.normalize(x.float(), dim=1, eps=eps)

    def ref_topk(query, bank, k, index_base=0):
        s, i = (query @ bank.t()).topk(k, dim=1)
        return s, (i + index_base).to(torch.int32)

    def ref_vote(sim, idx, labels, num_classes, t):
        w = (sim / t).exp()
        one = torch.zeros(sim.shape[0], num_classes)
        one.scatter_add_(1, labels[idx.long()], w)
        return one.argsort(dim=1, descending=True)

    F_hip.l2_normalize, F_hip.knn_topk, F_hip.knn_vote = ref_l2, ref_topk, ref_vote


class _ListLoader:
    """A plain list of (images, targets) batches with a length: what rank_batches shards by contiguous ranges."""

    def __init__(self, batches):
        self.batches = batches

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def _knn_problem(short=False):
    g = torch.Generator().manual_seed(5)
    proj = torch.randn(20, 12, generator=g)
    centers = torch.randn(4, 20, generator=g) * 2
    def make(n):
        y = torch.randint(0, 4, (n,), generator=g)
        return centers[y] + torch.randn(n, 20, generator=g), y
    xb, yb = make(103)
    xv, yv = make(41)
    if short:   # ONE batch each: with two ranks, rank 0 owns no batch at all
        return proj, _ListLoader([(xb, yb)]), _ListLoader([(xv, yv)])
    bank = _ListLoader([(xb[i:i + 16], yb[i:i + 16]) for i in range(0, 103, 16)])   # 7 batches, ragged tail
    val = _ListLoader([(xv[i:i + 8], yv[i:i + 8]) for i in range(0, 41, 8)])        # 6 batches
    return proj, bank, val


def _knn_module_run(world_tag, short=False):
    from ssl_wafermap_amd.models.knn import KNNBenchmarkModule
    from ssl_wafermap_amd.trainer import Trainer

    proj, bank, val = _knn_problem(short)

    class M(KNNBenchmarkModule):
        def __init__(self):
            super().__init__(bank, 4, knn_k=5, knn_t=0.1)
            self.backbone = torch.nn.Linear(20, 12, bias=False)
            with torch.no_grad():
                self.backbone.weight.copy_(proj.t())

    m = M()
    tr = Trainer(verbose=False)
    torch.cuda.synchronize = lambda *a, **k: None
    tr.validate(m, val)
    return m


def _sharded_knn_module(rank, world, short=False):
    _knn_stand_ins()
    m = _knn_module_run("dp", short)
    # every rank holds the whole bank in single-process order and reports the metrics of ALL validation samples
    _, bank, val = _knn_problem(short)
    assert m.feature_bank_nd.shape[0] == 103 and torch.equal(m.targets_bank, torch.cat([y for _, y in bank.batches]))
    assert m.last_preds.shape[0] == 41 and torch.equal(m.last_targets, torch.cat([y for _, y in val.batches]))
    out = torch.cat([m.last_preds.float(), torch.tensor([m.logged["knn_accuracy"], m.logged["knn_f1"]])])
    both = [torch.empty_like(out) for _ in range(world)]
    dist.all_gather(both, out)
    assert torch.equal(both[0], both[1])
    torch.save(out, os.environ["WM_TEST_OUT"] + f".{rank}")


def test_knn_module_sharded_evaluation_equals_single_process_world2(tmp_path):
    """KNNBenchmarkModule under world_size 2: each rank embeds its share of the bank and validates its share of the
    batches; predictions and metrics equal the single-process run EXACTLY."""
    _knn_stand_ins()
    single = _knn_module_run("single")
    want = torch.cat([single.last_preds.float(), torch.tensor([single.logged["knn_accuracy"], single.logged["knn_f1"]])])
    os.environ["WM_TEST_OUT"] = str(tmp_path / "knn_out")
    _spawn(_sharded_knn_module, 2)
    for r in range(2):
        got = torch.load(str(tmp_path / "knn_out") + f".{r}")
        assert torch.equal(got, want)


def test_knn_module_sharded_evaluation_with_fewer_batches_than_ranks_world2(tmp_path):
    """A one-batch bank loader and a one-batch validation loader under world_size 2: rank 0 owns NO batch.  It must
    still take part in every collective with the right feature width / dtype (learned from rank 1) and report the
    single-process metrics (used to raise on rank 0 while rank 1 hung in all_gather)."""
    _knn_stand_ins()
    single = _knn_module_run("single", True)
    want = torch.cat([single.last_preds.float(), torch.tensor([single.logged["knn_accuracy"], single.logged["knn_f1"]])])
    os.environ["WM_TEST_OUT"] = str(tmp_path / "knn_short")
    _spawn(_sharded_knn_module, 2, True)
    for r in range(2):
        got = torch.load(str(tmp_path / "knn_short") + f".{r}")
        assert torch.equal(got, want)


def test_all_gather_rows_adopts_shape_and_dtype_on_empty_ranks_world2():
    _spawn(_gather_rows_empty, 2)


def _gather_rows_empty(rank, world):
    from ssl_wafermap_amd import distributed as wdist

    mine = torch.arange(12, dtype=torch.bfloat16).reshape(3, 4) if rank == 1 else torch.empty((0, 0))
    got = wdist.all_gather_rows(mine)
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == (3, 4) and torch.equal(got, torch.arange(12, dtype=torch.bfloat16).reshape(3, 4))
    both_empty = wdist.all_gather_rows(torch.empty((0,), dtype=torch.long))
    assert both_empty.numel() == 0


def test_knn_classifier_fit_and_predict_cpu_contract():
    """utils.benchmarking.KNNClassifier (lightly's class form): bank from training_step batches, top-k accuracies
    from validation_step, predictions equal to knn_predict on the same features."""
    from ssl_wafermap_amd.utils.benchmarking import KNNClassifier, knn_predict, mean_topk_accuracy

    _knn_stand_ins()
    proj, bank, val = _knn_problem()
    model = torch.nn.Linear(20, 12, bias=False)
    with torch.no_grad():
        model.weight.copy_(proj.t())
    clf = KNNClassifier(model, num_classes=4, knn_k=5, knn_t=0.1, topk=(1, 3))
    assert clf.validation_step(val.batches[0]) is None            # no bank yet
    clf.fit_bank(bank)
    assert clf._train_features_tensor.shape == (12, 103)
    xb = torch.cat([x for x, _ in bank.batches])
    yb = torch.cat([y for _, y in bank.batches])
    fb = torch.nn.functional.normalize(xb @ proj, dim=1)
    hits = 0
    for x, y in val.batches:
        pred = clf.validation_step((x, y))
        want = knn_predict(torch.nn.functional.normalize(x @ proj, dim=1), fb.t(), yb, 4, 5, 0.1)
        assert torch.equal(pred, want)
        acc = mean_topk_accuracy(pred, y, k=(1, 3))
        assert float(clf.logged["val_top1"]) == float(acc[1]) and float(acc[3]) >= float(acc[1])
        hits += int((pred[:, 0] == y).sum())
    assert hits / 41 > 0.6


def _batch_shuffle(rank, world):
    from ssl_wafermap_amd.utils.model_utils import batch_shuffle, batch_unshuffle

    torch.manual_seed(100 + rank)      # different generators per rank: the permutation must still be rank 0's
    mine = torch.arange(6, dtype=torch.float32).reshape(6, 1) + 10 * rank
    shuf, inv = batch_shuffle(mine.clone(), distributed=True)
    assert shuf.shape == mine.shape
    both = [torch.empty_like(shuf) for _ in range(world)]
    dist.all_gather(both, shuf)
    seen = sorted(torch.cat(both).flatten().tolist())
    assert seen == sorted(list(range(6)) + [10 + i for i in range(6)])       # a permutation of the GLOBAL batch
    invs = [torch.empty_like(inv) for _ in range(world)]
    dist.all_gather(invs, inv)
    assert torch.equal(invs[0], invs[1])                                      # one permutation, rank 0's
    back = batch_unshuffle(shuf * 2.0, inv, distributed=True)                 # (a per-row function in between)
    assert torch.equal(back, mine * 2.0)


def test_batch_shuffle_distributed_world2():
    _spawn(_batch_shuffle, 2)


def _gathered_sinkhorn(rank, world):
    """loss.sinkhorn(gather_distributed=True) all-gathers the [B, K] scores and runs ONE Sinkhorn on the global matrix,
    keeping this rank's rows.  lightly instead all-reduces the total and the per-prototype sums inside the loop: the
    two formulations are the same iteration (checked here with the kernel's arithmetic restated in torch)."""
    import ssl_wafermap_amd.loss as L

    b, k = 5, 12
    g = torch.Generator().manual_seed(1)
    scores_all = torch.randn(world * b, k, generator=g)

    def sinkhorn_matrix(out, iterations, eps):        # wm_sinkhorn's contract on one matrix
        q = torch.exp(out / eps).t()
        q = q / q.sum()
        kk, bb = q.shape
        for _ in range(iterations):
            q = q / q.sum(1, keepdim=True) / kk
            q = q / q.sum(0, keepdim=True) / bb
        return (q * bb).t()

    mine = scores_all[rank * b:(rank + 1) * b]
    gathered = L._all_gather_rows(mine.contiguous())            # the product path's exchange
    assert torch.equal(gathered, scores_all)
    got = sinkhorn_matrix(gathered, 3, 0.05)[rank * b:(rank + 1) * b]
    q = torch.exp(mine / 0.05).t()                              # lightly.loss.swav_loss.sinkhorn, gather_distributed
    tot = q.sum()
    dist.all_reduce(tot)
    q = q / tot
    for _ in range(3):
        rs = q.sum(1, keepdim=True)
        dist.all_reduce(rs)
        q = q / rs / k
        q = q / q.sum(0, keepdim=True) / (b * world)
    torch.testing.assert_close(got, (q * b * world).t(), rtol=1e-5, atol=1e-7)


def test_sinkhorn_gather_distributed_equals_lightly_allreduce_form_world2():
    _spawn(_gathered_sinkhorn, 2)


# ---- bench.py starts its own ranks (reference switch scripts/WM811k_benchmark.py:78-85)
def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` with no launcher environment: the parent (which never touches torch / the GPU) starts
    two children through torch.distributed.run, they rendezvous (gloo here, RCCL on GPUs), all-reduce a one, and rank
    0's JSON line comes back on the parent's stdout with the parent's exit status 0."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WM_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["dry_run"] is True and out["backend"] == "gloo"


def test_bench_launcher_propagates_a_failing_rank():
    """A rank that exits non-zero (here: --gpus disagrees with the world the launcher made) fails the parent too."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
