"""GPU: long-run stability of the headline configuration (BASELINE cfg 2) and its agreement with the
float32 oracle over the first optimiser steps.

Round 1's default `bench.py` run (110 graph-replayed steps at bs 256) ended with a NaN loss.  Root cause
(profiles/r02_nan_root_cause.md): the NT-Xent backward cleared its gradient buffer with hipMemsetAsync and
added into it with f32 atomics; captured as a memset NODE inside the ~400-node hipGraph of the training step
that clear was not reliably applied, so the atomics landed on stale pool memory.  The library now issues no
memset (tests/test_abi.py checks the sources) and these tests run the configuration that failed.
Reference step: scripts/WM811k_benchmark.py:242-255."""
import numpy as np
import pytest
import torch
from parity_log import parity

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(B, n_wafers, seed=1234):
    from ssl_wafermap_amd.data import WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.models import SimCLR
    from ssl_wafermap_amd.transforms import BaseViewTransform

    wafers, labels = synthetic_wafers(n_wafers, seed=seed)
    ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(), device=DEV)
    torch.manual_seed(0)
    model = SimCLR(None, 9, batch_size=B, max_epochs=150).to(DEV).train()
    (opt,), _ = model.configure_optimizers()
    return ds, model, opt


def test_ntxent_backward_captured_through_autograd_is_replay_stable():
    """loss.backward() of a captured step runs on torch's autograd worker thread; every replay must return
    the gradient of ITS input, independent of what the output buffer held before."""
    from ssl_wafermap_amd.loss import NTXentLoss

    torch.manual_seed(0)
    b, d = 256, 128
    z = torch.randn(2 * b, d, device=DEV).bfloat16().requires_grad_(True)
    crit = NTXentLoss()
    out = torch.zeros(2 * b, d, device=DEV)

    def body():
        loss = crit(z[:b], z[b:])
        (g,) = torch.autograd.grad(loss, z)
        out.copy_(g)
        return loss

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        loss = body()
    for i in range(12):
        with torch.no_grad():
            z.copy_(torch.randn(2 * b, d, device=DEV, generator=None).bfloat16())
        graph.replay()
        got, got_loss = out.clone(), float(loss.detach())
        want_loss = crit(z[:b], z[b:])
        (want,) = torch.autograd.grad(want_loss, z)
        assert torch.isfinite(got).all()
        torch.testing.assert_close(got, want.float(), atol=2e-6, rtol=1e-2)  # the gradient is returned in z's dtype (bf16): one rounding of 2^-9 per element
        assert abs(got_loss - float(want_loss)) < 1e-5


def test_bs256_graph_replayed_training_stays_finite_and_falls():
    """>= 300 graph-replayed steps at bs 256 (the run that went to NaN between steps 40 and 170 in round 1)."""
    from ssl_wafermap_amd.graph import GraphedTrainStep

    B, steps = 256, 320
    ds, model, opt = _setup(B, 4096)
    rng = np.random.default_rng(0)
    for i in range(3):
        batch = ds.get_batch((np.arange(B) + i * B) % len(ds), rng, fmt="s2d_bf16")
        opt.zero_grad()
        model.training_step(batch, i).backward()
        opt.step()
    g = GraphedTrainStep(model, opt, ds, B, fmt="s2d_bf16").capture(np.arange(B), rng)
    arena = opt.grad_arenas[0]
    losses = []
    for i in range(steps):
        loss = g.step((np.arange(B) + i * B) % len(ds), rng)
        if i % 10 == 9 or i == 0:
            losses.append(float(loss.detach()))
            assert np.isfinite(losses[-1]), f"non-finite loss at replayed step {i}: {losses}"
            assert bool(torch.isfinite(arena).all()), f"non-finite gradient at replayed step {i}"
    params = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert bool(torch.isfinite(params).all())
    first, last = np.mean(losses[:3]), np.mean(losses[-3:])
    assert last < first - 0.4, (first, last)          # measured: 5.2 -> 4.6 over 320 steps
    assert last < np.log(2 * B - 1) - 1.0              # well below the uniform-similarity plateau ln(511) = 6.24


def test_bs32_first_steps_track_the_float32_oracle():
    """Same weights, same augmentation decisions, same SGD: the graph-replayed HIP path (bf16 activations)
    against the float32 torch-CPU oracle over the first optimiser steps (reference
    scripts/WM811k_benchmark.py:242-255).  The views the oracle consumes are the augmentation kernel's
    output for the same decisions (bit-exact vs oracle/augment.py: tests/test_gpu_augment.py), rounded to
    bf16 as the HIP model sees them."""
    from oracle import resnet as orn
    from ssl_wafermap_amd.graph import GraphedTrainStep
    from ssl_wafermap_amd.transforms import augment_views

    B, steps = 32, 5
    ds, model, opt = _setup(B, 256, seed=7)
    lr = opt.param_groups[0]["lr"]
    assert abs(lr - 6e-2 * B / 256) < 1e-12
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    names = [k for k, _ in model.named_parameters()]
    for k in names:
        sd[k].requires_grad_(True)
    bufs = {}
    # the capture's warm-up steps are undone (GraphedTrainStep.capture(restore=True)): the first replay starts from the
    # initial weights, exactly like the oracle
    cap_rng, rng, rng_o = np.random.default_rng(99), np.random.default_rng(5), np.random.default_rng(5)
    g = GraphedTrainStep(model, opt, ds, B, warmup=1, fmt="s2d_bf16").capture(np.arange(B), cap_rng)
    tr = ds.transform

    def oracle_step(idx, params):
        v = augment_views(ds.store, params[0], fmt="nchw_f32", n_slots=2 * B).bfloat16().float().cpu()
        for k in names:
            sd[k].grad = None
        loss, _ = orn.simclr_loss(v[:B], v[B:], sd, 0.5, True)
        loss.backward()
        with torch.no_grad():
            orn.sgd_step({k: sd[k] for k in names}, {k: sd[k].grad for k in names}, bufs, lr=lr)
        return float(loss.detach())

    got, want = [], []
    for i in range(steps):
        idx = (np.arange(B) + i * B) % len(ds)
        got.append(float(g.step(idx, rng).detach()))
        want.append(oracle_step(idx, tr.sample(ds.store, idx, rng_o)))
    print("hip   ", got)
    print("oracle", want)
    rel = np.abs(np.array(got) - np.array(want)) / np.abs(want)
    parity("SimCLR bs 32 graph-replayed step 1 loss vs float32 oracle (relative)", rel[0], 6e-4)  # measured 3.0e-4
    parity("SimCLR bs 32 graph-replayed steps 1-5 loss vs float32 oracle (relative, worst)", rel.max(), 3e-2,
           note="five optimiser steps of bf16-vs-float32 drift")


def test_bs64_whole_step_loss_matches_the_float32_oracle():
    """BASELINE cfg 2's step at the reference batch size (64 wafers, two views, T 0.5) on identical weights and
    identical augmentation decisions: the bf16 HIP step's loss against the float32 oracle.  north_star asks 1e-4
    relative: that holds at the loss KERNEL on identical embeddings (test_gpu_embed.py, 1e-5); through 18 bf16
    layers the whole step lands at a few 1e-4 (measured 1e-4 .. 5e-4 over seeds), bounded here at 1e-3."""
    from oracle import resnet as orn
    from ssl_wafermap_amd.transforms import augment_views

    B = 64
    rels = []
    for seed in (3, 4):
        ds, model, opt = _setup(B, 128, seed=seed)
        sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
        params = ds.transform.sample(ds.store, np.arange(B), np.random.default_rng(seed))
        views = ds.transform.launch(ds.store, params, B, "s2d_bf16")
        opt.zero_grad()
        loss = model.training_step((views, None), 0)
        v = augment_views(ds.store, params[0], fmt="nchw_f32", n_slots=2 * B).bfloat16().float().cpu()
        ref, _ = orn.simclr_loss(v[:B], v[B:], sd, 0.5, True)
        rels.append(abs(float(loss.detach()) - float(ref)) / abs(float(ref)))
    print("whole-step loss, relative error vs float32 oracle at bs 64:", rels)
    parity("SimCLR whole step loss at bs 64 vs float32 oracle (relative, worst of 2 seeds)", max(rels), 3e-4,
           note="north_star asks 1e-4: holds at the loss kernel (1e-5); 18 bf16 layers in front of it")


def test_bs256_step_is_bit_reproducible():
    """Round-2 verdict: two runs of the SAME step differed by 4-10 % in the gradients (f32 atomics in the BatchNorm
    statistics, the split-K weight gradients and the NT-Xent backward reorder sums in the last bit; bf16 roundings
    flip downstream).  None of those reductions uses an atomic any more (per-tile statistics slots and split-K slabs
    written with plain stores and added in a fixed order): the same weights, wafers and decisions give bit-identical losses, gradients and updates --
    eagerly and under hipGraph replay."""
    from ssl_wafermap_amd.graph import GraphedTrainStep

    B = 256

    def run(graph: bool):
        ds, model, opt = _setup(B, 1024)
        rng = np.random.default_rng(7)
        out = []
        if graph:
            g = GraphedTrainStep(model, opt, ds, B, fmt="s2d_bf16").capture(np.arange(B), np.random.default_rng(1))
        for i in range(3):
            idx = (np.arange(B) + i * B) % len(ds)
            if graph:
                loss = g.step(idx, rng)
            else:
                batch = ds.get_batch(idx, rng, fmt="s2d_bf16")
                opt.zero_grad()
                loss = model.training_step(batch, i)
                loss.backward()
                opt.step()
            torch.cuda.synchronize()
            out.append((float(loss.detach()), opt.grad_arenas[0].clone(),
                        torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()))
        return out

    for graph in (False, True):
        a, b = run(graph), run(graph)
        for (la, ga, pa), (lb, gb, pb) in zip(a, b):
            assert la == lb, (graph, la, lb)
            assert torch.equal(ga, gb), (graph, float((ga - gb).abs().max()))
            assert torch.equal(pa, pb)
    # the restore after the capture warm-up makes the replayed run the eager run, step for step
    e, g = run(False), run(True)
    for (le, ge, pe), (lg, gg, pg) in zip(e, g):
        assert le == lg and torch.equal(ge, gg) and torch.equal(pe, pg)


def test_view_branches_compute_the_single_stream_step(monkeypatch):
    """SimCLR's two views as two parallel branches (nn.ViewBranches, the default) against both views batched through one
    stream (WM_VIEW_BRANCHES=0), same weights, wafers and decisions, two optimiser steps at bs 128: the per-view arithmetic is
    the same, so the loss is bit-identical in the first step and the step-1 gradients differ only by the order in which the
    two views' float32 sums meet (6.8e-6 relative, bound 2e-5: the per-view launches split their pixel ranges differently from
    the batched one); from the second step on the bf16 roundings downstream of those last
    bits move the loss in its 5th digit (measured 1.3e-5; bound 1e-4, the distance of two bf16 runs that differ in one
    rounding); parameters and BatchNorm running statistics after two steps <= 1e-4; the batch counters agree exactly (they
    count forward calls: two per step)."""
    B = 128

    def run(mode: str):
        monkeypatch.setenv("WM_VIEW_BRANCHES", mode)
        ds, model, opt = _setup(B, 512)
        rng = np.random.default_rng(11)
        losses = []
        for i in range(2):
            batch = ds.get_batch((np.arange(B) + i * B) % len(ds), rng, fmt="s2d_bf16")
            opt.zero_grad()
            loss = model.training_step(batch, i)
            loss.backward()
            if i == 0:
                grads = opt.grad_arenas[0].clone()
            opt.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        assert (getattr(model.backbone, "_branches", None) is not None) == (mode == "1")
        params = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        bn = [m for m in model.backbone.modules() if hasattr(m, "running_mean") and m.running_mean is not None]
        stats = torch.cat([torch.cat([m.running_mean.reshape(-1), m.running_var.reshape(-1)]) for m in bn])
        counts = [int(m.num_batches_tracked) for m in bn]
        return losses, grads, params, stats, counts

    one, two = run("0"), run("1")
    assert one[0][0] == two[0][0], (one[0], two[0])
    assert abs(one[0][1] - two[0][1]) <= 1e-4 * abs(one[0][1]), (one[0], two[0])

    def rel(a, b):
        return float((a - b).norm() / b.norm())

    from parity_log import parity

    parity("SimCLR bs 128, views as parallel branches vs one stream: step-1 gradients (relative L2)", rel(two[1], one[1]), 2e-5)  # measured 6.8e-6
    parity("SimCLR bs 128, views as parallel branches vs one stream: parameters after two steps (relative L2)",
           rel(two[2], one[2]), 1e-4)
    parity("SimCLR bs 128, views as parallel branches vs one stream: BatchNorm running statistics after two steps (relative L2)",
           rel(two[3], one[3]), 1e-4)
    assert one[4] == two[4] and set(one[4]) == {4}, (one[4][:4], two[4][:4])


@pytest.mark.parametrize("which", ["dino_vit_tiny", "mae_vit_small_16"])
def test_transformer_steps_are_bit_reproducible_in_their_gradients(which):
    """Round 3, second half: the transformer steps' parameter gradients no longer pass through f32 atomics either
    (LayerNorm and bias column sums as per-block slots added by the pass's ordered fold, positional-embedding sums by an
    ordered finalize; Linear weight gradients were slabs already).  Two runs of the same two steps: identical gradient
    arenas and parameters.  (The scalar loss VALUES of the DINO / MSE kernels are still atomically summed: compared to
    1e-6.)"""
    from ssl_wafermap_amd.data import WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.models import MAE, DINOViT
    from ssl_wafermap_amd.transforms import BaseViewTransform, MultiCropTransform

    B = 8

    def run():
        torch.manual_seed(3)
        if which == "dino_vit_tiny":
            wafers, labels = synthetic_wafers(64, seed=2)
            ds = WaferMapDataset(wafers, labels, transform=MultiCropTransform(), device=DEV)
            model = DINOViT(None, 9, batch_size=B, log_rep_std=False, backbone="vit_tiny")
        else:
            wafers, labels = synthetic_wafers(64, seed=2, fixed_size=52)
            ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(n_views=1, denoise=True), device=DEV)
            model = MAE(None, 9, batch_size=B, log_rep_std=False, backbone="vit_small_16")
        model = model.to(DEV).train()
        (opt,), _ = model.configure_optimizers()
        rng = np.random.default_rng(5)
        gen = torch.Generator(device=DEV).manual_seed(11)
        out = []
        for i in range(2):
            batch = ds.get_batch(np.arange(B) + i * B, rng, fmt="nhwc_bf16")
            opt.zero_grad()
            loss = model.training_step(batch, i, generator=gen) if which != "dino_vit_tiny" else model.training_step(batch, i)
            loss.backward()
            opt.step()
            torch.cuda.synchronize()
            out.append((float(loss.detach()), torch.cat([a.clone() for a in opt.grad_arenas]),
                        torch.cat([p.detach().reshape(-1).float() for p in model.parameters()]).clone()))
        return out

    a, b = run(), run()
    for (la, ga, pa), (lb, gb, pb) in zip(a, b):
        assert abs(la - lb) <= 1e-6 * abs(la), (la, lb)
        assert torch.equal(ga, gb), float((ga - gb).abs().max())
        assert torch.equal(pa, pb)
