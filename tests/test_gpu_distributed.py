"""GPU: the data-parallel training step with a LIVE gradient exchange, two ranks on the one GPU of the test box
(gloo carries the collectives: RCCL refuses two ranks per device; the RCCL path differs only in the backend
string).  Covers broadcast_state, GraphedTrainStep with staged backward graphs + GradSync.start_range, and the
staged graphs against the single-graph step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _entry(rank, world, port, fn, args):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world, *args)
    finally:
        dist.destroy_process_group()


def _setup(rank, B, seed_model):
    from ssl_wafermap_amd.data import WaferLoader, WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.models import SimCLR
    from ssl_wafermap_amd.transforms import BaseViewTransform

    wafers, labels = synthetic_wafers(96, seed=11)
    ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(), device="cuda:0")
    torch.manual_seed(seed_model)
    model = SimCLR(None, 9, batch_size=2 * B, max_epochs=10, log_rep_std=False).to("cuda:0").train()
    (opt,), _ = model.configure_optimizers()
    return ds, model, opt, WaferLoader


def _staged_dp(rank, world, staged):
    from ssl_wafermap_amd import distributed as wdist
    from ssl_wafermap_amd.graph import GraphedTrainStep

    B = 8
    ds, model, opt, WaferLoader = _setup(rank, B, seed_model=100 + rank)   # replicas built from DIFFERENT seeds
    sync = wdist.GradSync(opt, bucket_bytes=4 << 20)
    wdist.broadcast_state(model, opt)
    loader = WaferLoader(ds, B, shuffle=True, drop_last=True, seed=3, rank=rank, world_size=world)
    it = loader.iter_indices()
    idx, rng = next(it)
    g = GraphedTrainStep(model, opt, ds, B, warmup=1, fmt="s2d_bf16", stages=staged).capture(idx, rng, sync)
    assert g.staged == bool(staged) and len(g.graphs) == (3 if staged else 1)
    losses = []
    first = None
    for k in range(3):
        idx, rng = next(it)
        losses.append(float(g.step(idx, rng, sync).detach()))
        if k == 0:  # state after the first replayed step: what the staged / single-graph comparison uses
            first = torch.cat([opt._arenas[0].params, opt._arenas[0].momentum]).clone()
    assert all(np.isfinite(losses))
    flat = torch.cat([opt._arenas[0].params, opt._arenas[0].momentum]).clone()
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), "replicas diverged"
    n_par = opt._arenas[0].params.numel()
    bufs = torch.cat([b.reshape(-1).float() for b in model.buffers()])
    assert torch.isfinite(bufs).all()
    if rank == 0:
        torch.save({"params": first[:n_par].cpu(), "momentum": first[n_par:].cpu(), "losses": losses},
                   os.environ["WM_TEST_OUT"] + f".{int(bool(staged))}")


def test_staged_graph_data_parallel_world2(tmp_path):
    """Two ranks, staged backward graphs with bucketed all-reduce between them: replicas bit-identical after 3
    steps, and the same weights (to f32-atomics noise) as the unstaged single-graph exchange."""
    os.environ["WM_TEST_OUT"] = str(tmp_path / "dp")
    for staged in (True, False):
        mp.spawn(_entry, args=(2, _free_port(), _staged_dp, (staged,)), nprocs=2, join=True)
    a = torch.load(str(tmp_path / "dp") + ".1")
    b = torch.load(str(tmp_path / "dp") + ".0")
    # Round 3: no floating-point atomics are left on the training path (BatchNorm statistics as per-tile slots, split-K
    # weight gradients as slabs folded in order, NT-Xent backward partials summed in order), so two runs of the same
    # schedule are bit-identical and the staged exchange (all-reduce of arena ranges under the next stage) gives the
    # same sums as the whole-arena exchange: the weights and momentum buffers of the two runs agree to the last bit
    # (round 2 accepted 0.35 on the momentum and 15 % on the step-3 loss: f32-atomics noise).
    from parity_log import parity

    parity("staged vs single-graph data-parallel run, losses of 3 steps (relative, worst)",
           float(np.max(np.abs(np.array(a["losses"]) - np.array(b["losses"])) / np.abs(np.array(b["losses"])))), 0.0,
           note="bit-identical")
    rel_p = float((a["params"] - b["params"]).norm() / b["params"].norm())
    rel_m = float((a["momentum"] - b["momentum"]).norm() / b["momentum"].norm())
    parity("staged vs single-graph data-parallel run, parameters after step 1 (relative L2)", rel_p, 0.0, note="bit-identical")
    parity("staged vs single-graph data-parallel run, momentum after step 1 (relative L2)", rel_m, 0.0, note="bit-identical")


def test_staged_graphs_equal_the_single_graph_step():
    """One process: the three-stage capture (cut before layer4 and layer3) computes the same gradients as the
    single-graph capture on identical decisions and weights."""
    from ssl_wafermap_amd.graph import GraphedTrainStep

    B = 8
    grads, losses = [], []
    for staged in (True, False, False):
        ds, model, opt, _ = _setup(0, B, seed_model=5)
        g = GraphedTrainStep(model, opt, ds, B, warmup=1, fmt="s2d_bf16", stages=staged)
        g.capture(np.arange(B), np.random.default_rng(1))
        assert g.staged == staged
        if staged:
            n = opt.grad_arenas[0].numel()
            assert 0 < g.bounds[1] < g.bounds[0] < n and (n - g.bounds[0]) > 0.7 * n  # layer4 + head: 3/4 of the bytes
        g._upload(g.tr.sample(ds.store, np.arange(B) + B, np.random.default_rng(2)))
        for gr in g.graphs:
            gr.replay()
        torch.cuda.synchronize()
        grads.append(opt.grad_arenas[0].clone())
        losses.append(float(g.loss.detach()))

    def rel(a, b):
        return float((a - b).norm() / b.norm())

    from parity_log import parity

    # two captures of the same schedule: bit-identical (no atomics); the staged capture runs the same kernels on the
    # same data in the same order, only cut into three graphs
    assert torch.equal(grads[1], grads[2]) and losses[1] == losses[2]
    print(f"staged vs single graph {rel(grads[0], grads[1]):.3e}")
    parity("staged vs single-graph capture, gradient arena (relative L2)", rel(grads[0], grads[1]), 0.0, note="bit-identical")
    parity("staged vs single-graph capture, loss (relative)", abs(losses[0] - losses[1]) / abs(losses[1]), 0.0, note="bit-identical")


def _sync_bn_case(rank, world, out_path):
    """Each rank runs a conv -> BN(+ReLU) -> conv -> BN(+shortcut, ReLU) stack and a BatchNorm1d on ITS half of the
    batch with synchronised statistics; rank 0 then repeats the computation alone on the whole batch."""
    from ssl_wafermap_amd import nn as wnn
    from ssl_wafermap_amd import ops

    def build():
        torch.manual_seed(3)
        conv1, bn1 = wnn.Conv2d(64, 64, 3, padding=1), wnn.BatchNorm2d(64)
        conv2, bn2 = wnn.Conv2d(64, 64, 3, padding=1), wnn.BatchNorm2d(64)
        bn1d = wnn.BatchNorm1d(256)
        mods = torch.nn.ModuleList([conv1, bn1, conv2, bn2, bn1d]).to("cuda:0").train()
        with torch.no_grad():
            for m in (bn1, bn2, bn1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
        return mods

    def run(mods, x, v, dy, dv):
        conv1, bn1, conv2, bn2, bn1d = mods
        x = x.clone().requires_grad_(True)
        v = v.clone().requires_grad_(True)
        st = bn1.stats_buffer(1)  # statistics slots written by the convolution's epilogue, as the ResNet blocks do
        h = bn1(conv1(x, stats=st, groups=1), relu=True, stats=st)
        y = bn2(conv2(h), residual=x, relu=True)  # statistics computed from the tensor
        z = bn1d(v)
        torch.autograd.backward([y, z], [dy, dv])
        grads = [p.grad.clone() for p in mods.parameters()]
        bufs = [b.clone() for b in mods.buffers()]
        return y.detach(), z.detach(), x.grad, v.grad, grads, bufs

    g = torch.Generator().manual_seed(21)
    N = 8  # whole batch; 4 images per rank
    x = ops.to_nhwc_bf16(torch.randn(N, 64, 16, 16, generator=g).to("cuda:0"))
    v = torch.randn(N * 4, 256, generator=g).to("cuda:0").bfloat16()
    dy = ops.to_nhwc_bf16(torch.randn(N, 64, 16, 16, generator=g).to("cuda:0"))
    dv = torch.randn(N * 4, 256, generator=g).to("cuda:0").bfloat16()
    h, hv = N // world, N * 4 // world
    sl, slv = slice(rank * h, (rank + 1) * h), slice(rank * hv, (rank + 1) * hv)

    mods = wnn.convert_sync_batchnorm(build())
    assert all(m.sync for m in mods if isinstance(m, (wnn.BatchNorm2d, wnn.BatchNorm1d)))
    y, z, dx, dvv, grads, bufs = run(mods, x[sl], v[slv], dy[sl], dv[slv])
    # parameter gradients are local sums (torch.nn.SyncBatchNorm's too): the whole-batch gradient is their sum
    for t in grads:
        dist.all_reduce(t)
    parts = [y, z, dx, dvv]
    gathered = []
    for t in parts:
        both = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(both, t.contiguous())
        gathered.append(torch.cat(both, 0))
    # running statistics must agree between the ranks to the bit
    for b in bufs:
        both = [torch.empty_like(b) for _ in range(world)]
        dist.all_gather(both, b)
        assert torch.equal(both[0], both[1])
    # eval mode ignores the flag
    mods.eval()
    with torch.no_grad():
        mods[1](x[sl])
    if rank == 0:
        ref_mods = build()
        ry, rz, rdx, rdv, rgrads, rbufs = run(ref_mods, x, v, dy, dv)
        torch.save({"got": [t.float().cpu() for t in gathered], "ref": [t.float().cpu() for t in (ry, rz, rdx, rdv)],
                    "got_g": [t.float().cpu() for t in grads], "ref_g": [t.float().cpu() for t in rgrads],
                    "got_b": [t.float().cpu() for t in bufs], "ref_b": [t.float().cpu() for t in rbufs]}, out_path)


def test_sync_batchnorm_world2_equals_whole_batch_statistics(tmp_path):
    """convert_sync_batchnorm (the reference's sync_batchnorm flag): two ranks with half the batch each produce the
    activations, input gradients, parameter gradients (summed) and running statistics of ONE BatchNorm over the whole
    batch.  The only differences are summation order (per-rank totals rounded to f32 before the exchange) and bf16
    rounding of outputs that sit on a rounding boundary."""
    from parity_log import parity

    out = str(tmp_path / "syncbn.pt")
    mp.spawn(_entry, args=(2, _free_port(), _sync_bn_case, (out,)), nprocs=2, join=True)
    d = torch.load(out)
    names = ["BN2d stack output", "BN1d output", "input gradient (4-D)", "input gradient (2-D)"]
    for n, a, b in zip(names, d["got"], d["ref"]):
        parity(f"SyncBN world 2 vs whole-batch BN, {n} (relative L2)", float((a - b).norm() / b.norm()), 6.5e-6,
               note="measured 4.7e-10 / 0 / 3.1e-6 / 0: per-rank totals are rounded to f32 before the exchange")
    worst = max(float((a - b).norm() / b.norm().clamp_min(1e-12)) for a, b in zip(d["got_g"], d["ref_g"]))
    parity("SyncBN world 2 vs whole-batch BN, parameter gradients summed over ranks (relative L2, worst tensor)", worst, 2.3e-6)  # measured 1.14e-6
    worst_b = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)) for a, b in zip(d["got_b"], d["ref_b"]))
    parity("SyncBN world 2 vs whole-batch BN, running statistics (relative max)", worst_b, 2.2e-7)  # measured 1.1e-7


def _sync_bn_wide_case(rank, world, out_path):
    """BYOL's head width (lightly BYOLProjectionHead: Linear - BatchNorm1d(4096) - ReLU, reference
    scripts/WM811k_benchmark.py:437-438) under convert_sync_batchnorm, two view groups per rank: the wide
    (C > 2048) form of the wm_bn_sync_* entry points."""
    from ssl_wafermap_amd import nn as wnn
    from ssl_wafermap_amd import ops

    C, b = 4096, 12   # rows per rank and view

    def build():
        torch.manual_seed(5)
        bn = wnn.BatchNorm1d(C).to("cuda:0").train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
        return bn

    def run(bn, v, dv):
        v = v.clone().requires_grad_(True)
        with ops.bn_groups(2):
            z = bn(v, relu=True)
        z.backward(dv)
        return z.detach(), v.grad, [p.grad.clone() for p in bn.parameters()], [t.clone() for t in bn.buffers()]

    g = torch.Generator().manual_seed(22)
    # [view][rank][row][C]: a rank sees its rows of both views, view-major; the whole batch is view-major over all ranks
    full = (torch.randn(2, world, b, C, generator=g) * 1.5 + 0.3).to("cuda:0").bfloat16()
    dfull = torch.randn(2, world, b, C, generator=g).to("cuda:0").bfloat16()
    mine, dmine = full[:, rank].reshape(2 * b, C), dfull[:, rank].reshape(2 * b, C)
    bn = wnn.convert_sync_batchnorm(build())
    z, dv, grads, bufs = run(bn, mine, dmine)
    for t in grads:
        dist.all_reduce(t)
    outs = []
    for t in (z, dv):
        both = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(both, t.contiguous())
        outs.append(torch.stack([x.reshape(2, b, C) for x in both], dim=1).reshape(2 * world * b, C))
    if rank == 0:
        rz, rdv, rgrads, rbufs = run(build(), full.reshape(2 * world * b, C), dfull.reshape(2 * world * b, C))
        torch.save({"got": [t.float().cpu() for t in outs], "ref": [rz.float().cpu(), rdv.float().cpu()],
                    "got_g": [t.float().cpu() for t in grads], "ref_g": [t.float().cpu() for t in rgrads],
                    "got_b": [t.float().cpu() for t in bufs], "ref_b": [t.float().cpu() for t in rbufs]}, out_path)


def test_sync_batchnorm_wide_world2_equals_whole_batch_statistics(tmp_path):
    from parity_log import parity

    out = str(tmp_path / "syncbn_wide.pt")
    mp.spawn(_entry, args=(2, _free_port(), _sync_bn_wide_case, (out,)), nprocs=2, join=True)
    d = torch.load(out)
    for n, a, b in zip(["output", "input gradient"], d["got"], d["ref"]):
        # the unsynchronised wide kernel takes a two-pass variance, the synchronised one sum / sum of squares (it has to:
        # the totals cross ranks): outputs on a bf16 rounding boundary may land on the other side
        parity(f"wide SyncBN (4096 channels) world 2 vs whole-batch BN, {n} (relative L2)", float((a - b).norm() / b.norm()), 1e-4)  # measured 0 / 3.1e-7 (one bf16 flip of one element would be ~1e-5)
    worst = max(float((a - b).norm() / b.norm().clamp_min(1e-12)) for a, b in zip(d["got_g"], d["ref_g"]))
    parity("wide SyncBN world 2 vs whole-batch BN, parameter gradients summed over ranks (relative L2, worst)", worst, 1e-6)  # measured 1.0e-7
    worst_b = max(float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-12)) for a, b in zip(d["got_b"], d["ref_b"]))
    parity("wide SyncBN world 2 vs whole-batch BN, running statistics (relative max)", worst_b, 1e-6)  # measured 1.5e-7


def _barlow_gather_case(rank, world, out_path):
    """lightly's BarlowTwinsLoss(gather_distributed=True) (reference scripts/WM811k_benchmark.py:364-366): every rank
    standardises ITS half of the batch, c = z_a^T z_b / N_local / world, all_reduce(c); the gradient reaches the local
    projections through the local term only.  Restated here in float32 on the same bf16-exact inputs."""
    from ssl_wafermap_amd.loss import BarlowTwinsLoss

    g = torch.Generator().manual_seed(31)
    n, d = 64, 256
    a = (torch.randn(world * n, d, generator=g) * 1.5 + 0.3).bfloat16().float()
    b = (a + 0.7 * torch.randn(world * n, d, generator=g)).bfloat16().float()
    a, b = a[rank * n:(rank + 1) * n], b[rank * n:(rank + 1) * n]

    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    za = (ar - ar.mean(0)) / ar.std(0)
    zb = (br - br.mean(0)) / br.std(0)
    c_local = za.T @ zb / n / world
    parts = [torch.empty_like(c_local) for _ in range(world)]
    dist.all_gather(parts, c_local.detach().contiguous())
    c = c_local + sum(p for i, p in enumerate(parts) if i != rank)   # the other ranks' terms carry no gradient
    inv = c.diagonal().add(-1).pow(2).sum()
    off = (c - torch.diag_embed(c.diagonal())).pow(2).sum()
    ref = inv + 5e-3 * off
    ref.backward()

    ad, bd = a.to("cuda:0").bfloat16().requires_grad_(True), b.to("cuda:0").bfloat16().requires_grad_(True)
    loss = BarlowTwinsLoss(gather_distributed=True).to("cuda:0")(ad, bd)
    loss.backward()
    both = [torch.empty(1, device="cuda:0") for _ in range(world)]
    dist.all_gather(both, loss.detach().reshape(1))
    # every rank evaluates the loss on the same all-reduced matrix (its scalar is an atomically summed f32: last bits)
    assert abs(float(both[0]) - float(both[1])) <= 1e-6 * abs(float(both[0])), (float(both[0]), float(both[1]))
    local = BarlowTwinsLoss(gather_distributed=False).to("cuda:0")(ad.detach(), bd.detach())
    assert abs(float(local) - float(loss.detach())) > 1e-3 * abs(float(loss.detach())), "gathered == local loss?"
    ca = torch.nn.functional.cosine_similarity(ad.grad.float().cpu().flatten(), ar.grad.flatten(), dim=0)
    cb = torch.nn.functional.cosine_similarity(bd.grad.float().cpu().flatten(), br.grad.flatten(), dim=0)
    torch.save({"loss": float(loss.detach()), "ref": float(ref.detach()), "cos": min(float(ca), float(cb)),
                "gmax": float((ad.grad.float().cpu() - ar.grad).abs().max() / ar.grad.abs().max())}, f"{out_path}.{rank}")


def test_barlow_twins_gather_distributed_world2(tmp_path):
    from parity_log import parity

    out = str(tmp_path / "barlow")
    mp.spawn(_entry, args=(2, _free_port(), _barlow_gather_case, (out,)), nprocs=2, join=True)
    for r in range(2):
        d = torch.load(f"{out}.{r}")
        # standardised projections are stored in bf16 before the correlation GEMM: 2^-9 on each factor (the
        # single-process test accepts the same 2 %)
        parity(f"Barlow Twins gather_distributed world 2, rank {r}: loss vs float32 restatement (relative)",
               abs(d["loss"] - d["ref"]) / abs(d["ref"]), 2e-2)
        parity(f"Barlow Twins gather_distributed world 2, rank {r}: projection gradients (cosine, worse of the two)",
               d["cos"], 0.995, higher=True)
        parity(f"Barlow Twins gather_distributed world 2, rank {r}: projection gradients (max |err| / max |ref|)",
               d["gmax"], 0.05)


def _dino_center_case(rank, world, graphed, out_path):
    """Two data-parallel DINO ViT-Tiny steps: the loss's centre is a cross-rank quantity (lightly all-reduces the batch
    centre).  Under hipGraph replay the collective is deferred to `post_graph_step` (loss.DINOLoss.finish_center_update)."""
    from ssl_wafermap_amd import distributed as wdist
    from ssl_wafermap_amd.data import WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.graph import GraphedTrainStep
    from ssl_wafermap_amd.models import DINOViT
    from ssl_wafermap_amd.transforms import MultiCropTransform

    B = 4
    wafers, labels = synthetic_wafers(64, seed=5)
    ds = WaferMapDataset(wafers, labels, transform=MultiCropTransform(), device="cuda:0")
    torch.manual_seed(7)
    model = DINOViT(None, 9, batch_size=B * world, log_rep_std=False, backbone="vit_tiny").to("cuda:0").train()
    (opt,), _ = model.configure_optimizers()
    sync = wdist.GradSync(opt)
    wdist.broadcast_state(model, opt)
    rng = np.random.default_rng(100 + rank)
    idx = [np.arange(B) + (2 * i + rank) * B for i in range(3)]
    if graphed:
        g = GraphedTrainStep(model, opt, ds, B, warmup=1, fmt="nhwc_bf16").capture(idx[0], np.random.default_rng(1), sync)
        assert model.criterion._center_pending and model.criterion._center_mean is not None
        for i in range(2):
            g.step(idx[i], rng, sync)
    else:
        for i in range(2):
            batch = ds.get_batch(idx[i], rng, fmt="nhwc_bf16")
            opt.zero_grad()
            loss = model.training_step(batch, i)
            loss.backward()
            sync.start()
            sync.wait()
            opt.step()
    torch.cuda.synchronize()
    c = model.criterion.center.detach().float().reshape(-1)
    both = [torch.empty_like(c) for _ in range(world)]
    dist.all_gather(both, c)
    assert torch.equal(both[0], both[1]), "the centre is a global quantity: identical on every rank"
    assert float(c.abs().max()) > 0
    if rank == 0:
        torch.save(c.cpu(), f"{out_path}.{int(graphed)}")


def test_dino_center_under_graph_replay_equals_eager_world2(tmp_path):
    """the deferred cross-rank half of the DINO centre update (graph replay) against the eager step, two ranks."""
    from parity_log import parity

    out = str(tmp_path / "center")
    for graphed in (True, False):
        mp.spawn(_entry, args=(2, _free_port(), _dino_center_case, (graphed, out)), nprocs=2, join=True)
    a, b = torch.load(out + ".1"), torch.load(out + ".0")
    # same decisions, weights and batches; the ViT path keeps f32 atomics in the LayerNorm parameter gradients, so the
    # second step's teacher differs in the last bits between any two runs
    parity("DINO centre after 2 data-parallel steps, graph replay vs eager (relative L2)", float((a - b).norm() / b.norm()), 1e-4,
           note="measured 0 (identical); the bound leaves room for the f32-atomic LayerNorm gradient sums of the ViT path")
