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
    # the first replayed step starts from identical state; later steps start from states that already differ by the
    # run-to-run gradient noise described below (observed up to 2.8 % in the loss at step 3 with 8 wafers per rank)
    np.testing.assert_allclose(a["losses"][0], b["losses"][0], rtol=2e-2)
    np.testing.assert_allclose(a["losses"][1:], b["losses"][1:], rtol=0.15)
    # Two runs of the SAME step differ by 4-10 % in the gradients (f32 atomics reorder BatchNorm / wgrad sums in the
    # last bit, bf16 roundings downstream flip, and at random init BatchNorm-bias gradients are sums of cancelling
    # terms: tools/probes/grad_repro_probe.py, profiles/r02_experiments.md), so the momentum buffers -- accumulated
    # gradients -- of two correct runs agree only to that level, and the runs drift further apart with every step
    # (0.46 after four): compared right after the first replayed step; the weights, at lr 0.004, to 1e-3
    rel_p = float((a["params"] - b["params"]).norm() / b["params"].norm())
    rel_m = float((a["momentum"] - b["momentum"]).norm() / b["momentum"].norm())
    assert rel_p < 1e-3 and rel_m < 0.35, (rel_p, rel_m)


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

    # run-to-run noise of the SAME (unstaged) step: f32 atomics in wgrad / BN statistics reorder, bf16 roundings
    # downstream flip, and at batch 8 one step after a weight update that is a few per cent of the gradient norm
    noise = rel(grads[1], grads[2])
    print(f"staged vs single graph {rel(grads[0], grads[1]):.4f}, single graph vs itself {noise:.4f}")
    # (the loss itself carries that noise: BatchNorm statistics are f32-atomic sums; 1.2e-3 seen between two captures)
    assert abs(losses[0] - losses[1]) < 2.0 * abs(losses[1] - losses[2]) + 3e-3 * abs(losses[1])
    assert float(torch.nn.functional.cosine_similarity(grads[0], grads[1], dim=0)) > 0.998
    # two captures of the same schedule can come out bit-identical (noise ~ 0) while a different launch schedule
    # reorders the atomics: the 4-10 % band of tools/probes/grad_repro_probe.py is the bound, the cosine the check
    assert rel(grads[0], grads[1]) < max(2.0 * noise, 0.12)
