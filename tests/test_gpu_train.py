"""GPU: the training loop pieces — fused SGD parity, loader determinism, a short SimCLR fit with
kNN validation through the reference-shaped module API."""
import numpy as np
import pytest
import torch
from parity_log import parity

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_fused_sgd_matches_torch_sgd():
    from ssl_wafermap_amd import optim

    torch.manual_seed(0)
    shapes = [(64, 3, 7, 7), (64,), (128, 64, 3, 3), (5,), (512, 512)]
    ref = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref]
    o_ref = torch.optim.SGD(ref, lr=0.06, momentum=0.9, weight_decay=5e-4)
    o_hip = optim.SGD(mine, lr=0.06, momentum=0.9, weight_decay=5e-4)
    sched_ref = torch.optim.lr_scheduler.CosineAnnealingLR(o_ref, 10)
    sched_hip = torch.optim.lr_scheduler.CosineAnnealingLR(o_hip, 10)
    for it in range(4):
        o_ref.zero_grad()
        o_hip.zero_grad()
        for p, q in zip(ref, mine):
            g = torch.randn(p.shape, generator=torch.Generator().manual_seed(it * 10 + p.numel() % 7))
            p.grad = g.clone()
            q.grad.add_(g.to(DEV))  # accumulates into the arena view, as autograd does
        o_ref.step()
        o_hip.step()
        sched_ref.step()
        sched_hip.step()
        for p, q in zip(ref, mine):
            torch.testing.assert_close(q.detach().cpu(), p.detach(), atol=1e-6, rtol=1e-6)
    assert all(q.grad.data_ptr() >= o_hip.grad_arenas[0].data_ptr() for q in mine)


def test_loader_is_deterministic_and_rank_sliced():
    from ssl_wafermap_amd.data import WaferLoader, WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.transforms import BaseViewTransform, InferenceTransform

    wafers, labels = synthetic_wafers(40, seed=3)
    ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(), device=DEV)
    a = [(v.stacked.clone(), y.clone()) for v, y in WaferLoader(ds, 8, shuffle=True, drop_last=True, seed=5)]
    b = [(v.stacked.clone(), y.clone()) for v, y in WaferLoader(ds, 8, shuffle=True, drop_last=True, seed=5)]
    assert len(a) == 5 and all(torch.equal(x[0], z[0]) and torch.equal(x[1], z[1]) for x, z in zip(a, b))
    assert a[0][0].shape == (16, 3, 224, 224) and a[0][0].dtype == torch.bfloat16
    # two ranks of 4 see the two halves of the global batch of 8 (labels identify the samples)
    r0 = [y for _, y in WaferLoader(ds, 4, shuffle=True, drop_last=True, seed=5, rank=0, world_size=2)]
    r1 = [y for _, y in WaferLoader(ds, 4, shuffle=True, drop_last=True, seed=5, rank=1, world_size=2)]
    for (_, y), y0, y1 in zip(a, r0, r1):
        assert torch.equal(torch.cat([y0, y1]), y)
    dv = WaferMapDataset(wafers, labels, transform=InferenceTransform(), device=DEV)
    x, y = next(iter(WaferLoader(dv, 10)))
    assert x.shape == (10, 3, 224, 224) and y.shape == (10,)


def test_simclr_fit_with_knn_validation():
    from ssl_wafermap_amd.data import WaferLoader, WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.models import SimCLR
    from ssl_wafermap_amd.trainer import Trainer
    from ssl_wafermap_amd.transforms import BaseViewTransform, InferenceTransform

    wafers, labels = synthetic_wafers(96, seed=1)
    ds_ssl = WaferMapDataset(wafers[:64], labels[:64], transform=BaseViewTransform(), device=DEV)
    ds_knn = WaferMapDataset(wafers[:64], labels[:64], transform=InferenceTransform(), device=DEV)
    ds_val = WaferMapDataset(wafers[64:], labels[64:], transform=InferenceTransform(), device=DEV)
    torch.manual_seed(0)
    model = SimCLR(WaferLoader(ds_knn, 16), 9, knn_k=5, knn_t=0.1, batch_size=16, max_epochs=3).to(DEV)
    w0 = model.backbone.conv1.weight.detach().clone()
    hist = Trainer(max_epochs=3, verbose=False).fit(model, WaferLoader(ds_ssl, 16, shuffle=True, drop_last=True, seed=0),
                                                    WaferLoader(ds_val, 16))
    assert len(hist) == 3
    losses = [h["train_loss_ssl"] for h in hist]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] + 0.2
    assert 0.0 <= hist[-1]["knn_accuracy"] <= 1.0 and 0.0 <= hist[-1]["knn_f1"] <= 1.0
    assert model.confusion_matrix[-1].shape == (9, 9)
    assert not torch.equal(w0, model.backbone.conv1.weight.detach())
    assert torch.isfinite(model.logged["rep_std"]).all()
    # loss at init is near log(2B-1) for random embeddings (reference band: 3.73 at bs 64 ~ log 127 = 4.84 upper bound)
    assert losses[0] < np.log(2 * 16 - 1) + 0.5


def test_graphed_step_matches_eager():
    """The hipGraph-captured step (zero_grad -> augmentation -> fwd -> bwd) + eager SGD follows the
    same loss trajectory as the eager loop on identical decisions (f32 atomics in wgrad make the
    low bits order-dependent, so the comparison is a tolerance)."""
    from ssl_wafermap_amd.data import WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.graph import GraphedTrainStep
    from ssl_wafermap_amd.models import SimCLR
    from ssl_wafermap_amd.transforms import BaseViewTransform

    wafers, labels = synthetic_wafers(64, seed=2)
    B = 16

    def run(graph):
        ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(), device=DEV)
        torch.manual_seed(0)
        model = SimCLR(None, 9, batch_size=B, max_epochs=10).to(DEV).train()
        (opt,), _ = model.configure_optimizers()
        rng = np.random.default_rng(5)
        losses = []
        g = GraphedTrainStep(model, opt, ds, B, warmup=1) if graph else None
        if graph:
            g.capture(np.arange(B), np.random.default_rng(99))
        else:  # the capture consumes 1 warm-up + 0 recorded optimiser steps on the same decisions
            p = ds.transform.sample(ds.store, np.arange(B), np.random.default_rng(99))
            opt.zero_grad()
            loss = model.training_step((ds.transform.launch(ds.store, p, B), None), 0)
            loss.backward()
            opt.step()
        for i in range(4):
            idx = (np.arange(B) + i * B) % 64
            if graph:
                losses.append(float(g.step(idx, rng)))
            else:
                p = ds.transform.sample(ds.store, idx, rng)
                opt.zero_grad()
                loss = model.training_step((ds.transform.launch(ds.store, p, B), None), i)
                loss.backward()
                opt.step()
                losses.append(float(loss))
        return losses

    eager, graphed = run(False), run(True)
    assert all(np.isfinite(eager)) and all(np.isfinite(graphed))
    np.testing.assert_allclose(graphed, eager, rtol=5e-2)


def test_moco_step_matches_oracle_and_keeps_the_bank_order():
    """MoCo (reference scripts/WM811k_benchmark.py:289-351) on identical weights: symmetric loss against
    the float32 oracle, momentum encoders moved by the EMA, the bank holding the first term's keys
    before the second term reads it, and a falling loss under SGD."""
    import math

    from oracle import ntxent as ont
    from oracle import resnet as orn
    from ssl_wafermap_amd import ops
    from ssl_wafermap_amd.models import MoCo

    torch.manual_seed(0)
    b = 16
    model = MoCo(None, 9, batch_size=b, memory_bank_size=256, log_rep_std=False).to(DEV).train()
    (opt,), _ = model.configure_optimizers()
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(b, 3, 224, 224, generator=g).bfloat16().float()
    x1 = (x0 + 0.5 * torch.randn(b, 3, 224, 224, generator=g)).bfloat16().float()
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    model.criterion._init_memory_bank(128, torch.device(DEV))
    bank = model.criterion.bank.cpu().clone()

    # oracle: EMA first (teacher == student at init, so it is a no-op), then the two terms in order
    def enc(x, bb, hd):
        f = orn.resnet18_features(x, sd, True, prefix=bb + ".")
        return orn.moco_head(f, sd, prefix=hd + ".")

    q0, q1 = enc(x0, "backbone", "projection_head"), enc(x1, "backbone", "projection_head")
    with torch.no_grad():
        k1, k0 = enc(x1, "backbone_momentum", "projection_head_momentum"), enc(x0, "backbone_momentum", "projection_head_momentum")
    l1 = ont.ntxent_memory_bank(q0, k1, bank, 0.1)
    ptr = ont.memory_bank_enqueue(bank, 0, torch.nn.functional.normalize(k1, dim=1))
    l2 = ont.ntxent_memory_bank(q1, k0, bank, 0.1)
    ont.memory_bank_enqueue(bank, ptr, torch.nn.functional.normalize(k0, dim=1))
    ref = 0.5 * (l1 + l2)

    batch = ((ops.to_nhwc_bf16(x0.to(DEV)), ops.to_nhwc_bf16(x1.to(DEV))), None)
    opt.zero_grad()
    loss = model.training_step(batch, 0)
    loss.backward()
    parity("MoCo step loss vs float32 oracle (relative)", abs(float(loss.detach()) - float(ref)) / abs(float(ref)), 2e-4)  # measured 9.8e-5
    assert int(model.criterion.bank_ptr) == 2 * b
    got_bank = model.criterion.bank.cpu()
    cos = torch.nn.functional.cosine_similarity(got_bank[:, :2 * b].T, bank[:, :2 * b].T, dim=1)
    assert float(cos.min()) > 0.995  # bf16 encoder vs float32 oracle keys, in the same slots and order
    assert torch.equal(got_bank[:, 2 * b:], bank[:, 2 * b:])
    # a few more steps (the same batch re-enters the bank as negatives, so the loss need not fall): the
    # optimiser moves the query encoder and the momentum encoder follows it by the EMA
    w0 = model.backbone.conv1.weight.detach().clone()
    m0 = model.backbone_momentum.conv1.weight.detach().clone()
    for i in range(4):
        opt.step()
        opt.zero_grad()
        loss = model.training_step(batch, i + 1)
        loss.backward()
    assert math.isfinite(float(loss.detach()))
    dw = (model.backbone.conv1.weight.detach() - w0).abs().max()
    dm = (model.backbone_momentum.conv1.weight.detach() - m0).abs().max()
    assert float(dw) > 0 and 0 < float(dm) < float(dw)


@pytest.mark.parametrize("name", ["simsiam", "byol"])
def test_siamese_steps_match_oracle(name):
    """SimSiam / BYOL (reference scripts/WM811k_benchmark.py:605-640, 429-488) on identical weights: loss
    against the float32 oracle, then a few SGD steps stay finite and move the loss."""
    import math

    from oracle import resnet as orn
    from ssl_wafermap_amd import ops
    from ssl_wafermap_amd.models import BYOL, SimSiam

    torch.manual_seed(0)
    b = 16
    model = (SimSiam(None, 9, log_rep_std=False) if name == "simsiam" else BYOL(None, 9, batch_size=b, log_rep_std=False))
    model = model.to(DEV).train()
    (opt,), _ = model.configure_optimizers()
    g = torch.Generator().manual_seed(2)
    x0 = torch.randn(b, 3, 224, 224, generator=g).bfloat16().float()
    x1 = (x0 + 0.5 * torch.randn(b, 3, 224, 224, generator=g)).bfloat16().float()
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}

    def feats(x, bb):
        return orn.resnet18_features(x, sd, True, prefix=bb + ".")

    if name == "simsiam":
        z0 = orn.simsiam_projection_head(feats(x0, "backbone"), sd)
        z1 = orn.simsiam_projection_head(feats(x1, "backbone"), sd)
        p0, p1 = orn.byol_head(z0, sd, "prediction_head."), orn.byol_head(z1, sd, "prediction_head.")
        ref = 0.5 * (orn.neg_cosine(z0.detach(), p1) + orn.neg_cosine(z1.detach(), p0))
    else:
        p0 = orn.byol_head(orn.byol_head(feats(x0, "backbone"), sd, "projection_head."), sd, "prediction_head.")
        p1 = orn.byol_head(orn.byol_head(feats(x1, "backbone"), sd, "projection_head."), sd, "prediction_head.")
        with torch.no_grad():
            z0 = orn.byol_head(feats(x0, "backbone_momentum"), sd, "projection_head_momentum.")
            z1 = orn.byol_head(feats(x1, "backbone_momentum"), sd, "projection_head_momentum.")
        ref = 0.5 * (orn.neg_cosine(p0, z1) + orn.neg_cosine(p1, z0))
    batch = ((ops.to_nhwc_bf16(x0.to(DEV)), ops.to_nhwc_bf16(x1.to(DEV))), None)
    opt.zero_grad()
    loss = model.training_step(batch, 0)
    loss.backward()
    parity(f"{name} step loss vs float32 oracle (absolute; -cosine in [-1, 1])", abs(float(loss.detach()) - float(ref)), 6e-4)  # measured 3.1e-4 (simsiam), 2.1e-4 (byol)
    first = float(loss.detach())
    for i in range(6):
        opt.step()
        opt.zero_grad()
        loss = model.training_step(batch, i + 1)
        loss.backward()
    assert math.isfinite(float(loss.detach())) and float(loss.detach()) < first  # -cos falls on a repeated batch


def test_dino_resnet_step_matches_oracle():
    """DINO on ResNet-18 (reference :491-543): 2 global + 2 local crops, BatchNorm statistics per crop."""
    from oracle import resnet as orn
    from oracle import vit as ov
    from ssl_wafermap_amd import ops
    from ssl_wafermap_amd.models import DINO

    torch.manual_seed(0)
    b = 8
    model = DINO(None, 9, batch_size=b, log_rep_std=False).to(DEV).train()
    (opt,), _ = model.configure_optimizers()
    g = torch.Generator().manual_seed(3)
    views = [torch.randn(b, 3, 224, 224, generator=g).bfloat16().float() for _ in range(2)] + \
            [torch.randn(b, 3, 96, 96, generator=g).bfloat16().float() for _ in range(2)]
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    t_hd = {k[len("teacher_head."):]: v for k, v in sd.items() if k.startswith("teacher_head.")}
    s_hd = {k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}
    with torch.no_grad():
        t_out = [ov.dino_head(orn.resnet18_features(v, sd, True, prefix="teacher_backbone."), t_hd) for v in views[:2]]
        s_out = [ov.dino_head(orn.resnet18_features(v, sd, True, prefix="backbone."), s_hd) for v in views]
        ref, _ = ov.dino_loss(t_out, s_out, torch.zeros(1, 1, 2048), 0.04, 0.1)
    opt.zero_grad()
    loss = model.training_step(([ops.to_nhwc_bf16(v.to(DEV)) for v in views], None), 0)
    loss.backward()
    opt.step()
    assert abs(float(loss.detach()) - float(ref)) <= 1e-2 * abs(float(ref)), (float(loss), float(ref))


def test_barlow_twins_step_matches_oracle_and_lars_moves_it():
    import math

    from oracle import ntxent as ont
    from oracle import resnet as orn
    from ssl_wafermap_amd import ops
    from ssl_wafermap_amd.models import BarlowTwins

    torch.manual_seed(0)
    b = 32
    model = BarlowTwins(None, 9, batch_size=b, log_rep_std=False).to(DEV).train()
    (opt,), _ = model.configure_optimizers()
    g = torch.Generator().manual_seed(4)
    x0 = torch.randn(b, 3, 224, 224, generator=g).bfloat16().float()
    x1 = (x0 + 0.5 * torch.randn(b, 3, 224, 224, generator=g)).bfloat16().float()
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}

    def proj(x):
        f = orn.resnet18_features(x, sd, True, prefix="backbone.")
        h = orn._bn1d(torch.nn.functional.linear(f, sd["projection_head.layers.0.weight"]), sd, "projection_head.layers.1", True, 1, relu=True)
        h = orn._bn1d(torch.nn.functional.linear(h, sd["projection_head.layers.3.weight"]), sd, "projection_head.layers.4", True, 1, relu=True)
        return torch.nn.functional.linear(h, sd["projection_head.layers.6.weight"], sd["projection_head.layers.6.bias"])

    ref = ont.barlow_twins_loss(proj(x0), proj(x1))
    batch = ((ops.to_nhwc_bf16(x0.to(DEV)), ops.to_nhwc_bf16(x1.to(DEV))), None)
    opt.zero_grad()
    loss = model.training_step(batch, 0)
    loss.backward()
    parity("Barlow Twins step loss vs float32 oracle (relative)", abs(float(loss.detach()) - float(ref)) / abs(float(ref)), 2e-4)  # measured 1.0e-4
    first = float(loss.detach())
    for grp in opt.param_groups:
        grp["lr"] = 0.2  # past the warm-up factor of epoch 0
    for i in range(6):
        opt.step()
        opt.zero_grad()
        loss = model.training_step(batch, i + 1)
        loss.backward()
    assert math.isfinite(float(loss.detach())) and float(loss.detach()) < first


def test_swav_step_runs_and_matches_oracle_loss():
    from oracle import ntxent as ont
    from oracle import resnet as orn
    from ssl_wafermap_amd import ops
    from ssl_wafermap_amd.models import SwaV

    torch.manual_seed(0)
    b = 8
    model = SwaV(None, 9, batch_size=b, log_rep_std=False).to(DEV).train()
    (opt,), _ = model.configure_optimizers()
    g = torch.Generator().manual_seed(5)
    views = [torch.randn(b, 3, 224, 224, generator=g).bfloat16().float() for _ in range(2)] + \
            [torch.randn(b, 3, 96, 96, generator=g).bfloat16().float() for _ in range(2)]
    model.prototypes.normalize()
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}

    def scores(x):
        f = orn.resnet18_features(x, sd, True, prefix="backbone.")
        z = orn.byol_head(f, sd, "projection_head.")  # Linear-BN-ReLU, Linear(+bias): the same layer layout
        return torch.nn.functional.linear(torch.nn.functional.normalize(z, dim=1), sd["prototypes.layers.weight"])

    with torch.no_grad():
        outs = [scores(v) for v in views]
        ref = ont.swav_loss(outs[:2], outs[2:])
    opt.zero_grad()
    loss = model.training_step(([ops.to_nhwc_bf16(v.to(DEV)) for v in views], None), 0)
    loss.backward()
    opt.step()
    # exp(score / 0.05) amplifies the bf16 encoder's score error 20-fold inside Sinkhorn
    parity("SwaV step loss vs float32 oracle (relative)", abs(float(loss.detach()) - float(ref)) / abs(float(ref)), 1e-4)  # measured 4.7e-5
    norms = model.prototypes.layers.weight.detach().norm(dim=1)
    assert norms.shape == (3000,)


ZOO = ["SupervisedR18", "SimCLR", "DCLW", "MoCo", "BarlowTwins", "VICReg", "BYOL", "DINO", "DINOViT", "SimSiam", "FastSiam",
       "MSN", "PMSN", "SwaV", "MAE", "SimMIM"]


@pytest.mark.parametrize("name", ZOO)
def test_every_benchmark_model_trains_and_validates_through_the_trainer(name):
    """The reference's main() loop (scripts/WM811k_benchmark.py:1033-1150) for every model class of the script:
    its transform, one short epoch through Trainer.fit, kNN validation on the backbone features."""
    import ssl_wafermap_amd.models as zoo
    from ssl_wafermap_amd.data import WaferLoader, WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.trainer import Trainer
    from ssl_wafermap_amd.transforms import BaseViewTransform, InferenceTransform, MultiCropTransform

    wafers, labels = synthetic_wafers(48, seed=7)
    multi = name in ("DINO", "DINOViT", "SwaV", "MSN", "PMSN")
    one_view = name in ("MAE", "SimMIM", "SupervisedR18")
    if multi:
        tf = MultiCropTransform(n_local_views=2)
    elif name == "FastSiam":
        tf = BaseViewTransform(n_views=4)
    else:
        tf = BaseViewTransform(n_views=1 if one_view else 2)
    ds_ssl = WaferMapDataset(wafers[:32], labels[:32], transform=tf, device=DEV)
    ds_knn = WaferMapDataset(wafers[:32], labels[:32], transform=InferenceTransform(), device=DEV)
    ds_val = WaferMapDataset(wafers[32:], labels[32:], transform=InferenceTransform(), device=DEV)
    torch.manual_seed(0)
    cls = getattr(zoo, name)
    model = cls(WaferLoader(ds_knn, 16), 9, knn_k=5, knn_t=0.1, batch_size=8, max_epochs=2).to(DEV)
    if name in ("DINOViT", "MSN", "PMSN", "MAE", "SimMIM"):  # shrink the transformers: this is a plumbing test
        from ssl_wafermap_amd.models.mae import MAEBackbone
        from ssl_wafermap_amd.models.vit import VisionTransformer
        import copy as _copy

        if name == "DINOViT":
            model.backbone = VisionTransformer(patch_size=16, embed_dim=384, depth=1, num_heads=6).to(DEV)
            model.teacher_backbone = _copy.deepcopy(model.backbone)
            for p_ in model.teacher_backbone.parameters():
                p_.requires_grad = False
        elif name in ("MSN", "PMSN"):
            model.backbone = MAEBackbone(224, 16, 1, 6, 384, 1536).to(DEV)
            model.anchor_backbone = _copy.deepcopy(model.backbone)
            for p_ in model.backbone.parameters():
                p_.requires_grad = False
        else:
            model.backbone = MAEBackbone(224, 32, 1, 12, 768, 3072).to(DEV)
    if name == "SupervisedR18":
        train = WaferLoader(ds_ssl, 8, shuffle=True, drop_last=True, seed=0)
    else:
        train = WaferLoader(ds_ssl, 8, shuffle=True, drop_last=True, seed=0)
    hist = Trainer(max_epochs=1, limit_train_batches=2, verbose=False).fit(model, train, WaferLoader(ds_val, 16))
    assert len(hist) == 1
    key = "train_loss" if name == "SupervisedR18" else "train_loss_ssl"
    assert np.isfinite(float(model.logged[key]))
    assert 0.0 <= hist[-1]["knn_accuracy"] <= 1.0


@pytest.mark.gpu
def test_augmentation_writes_the_stem_layout_directly():
    """fmt "s2d_bf16" of the augmentation kernel == wm_image_to_s2d of its "nhwc_bf16" output, bit for bit, and
    a SimCLR step on it gives the same loss (the ResNet stem takes the space-to-depth tensor as is)."""
    import numpy as np

    from ssl_wafermap_amd import _lib, ops
    from ssl_wafermap_amd._lib import check, ptr, stream_ptr
    from ssl_wafermap_amd.data import WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.models import SimCLR
    from ssl_wafermap_amd.transforms import BaseViewTransform

    dev = torch.device("cuda:0")
    wafers, labels = synthetic_wafers(32, seed=5)
    ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(), device=dev)
    idx = np.arange(8)
    (a0, a1), _ = ds.get_batch(idx, np.random.default_rng(3), fmt="nhwc_bf16")
    (s0, s1), _ = ds.get_batch(idx, np.random.default_rng(3), fmt="s2d_bf16")
    assert s0.shape == (8, 16, 112, 112) and s0.dtype == torch.bfloat16
    for a, s in ((a0, s0), (a1, s1)):
        ref = torch.empty((8, 112, 112, 16), dtype=torch.bfloat16, device=dev)
        check(_lib.load().wm_image_to_s2d(a.data_ptr(), _lib.WM_IMG_NHWC_BF16, 8, 224, 224, ptr(ref), stream_ptr()),
              "wm_image_to_s2d")
        assert torch.equal(s.permute(0, 2, 3, 1).contiguous(), ref)
    torch.manual_seed(0)
    model = SimCLR(None, 9, batch_size=8, log_rep_std=False).to(dev).train()
    l_nhwc = float(model.training_step(((a0, a1), None), 0))
    l_s2d = float(model.training_step(((s0, s1), None), 0))
    assert l_nhwc == l_s2d


@pytest.mark.parametrize("kind", ["sgd", "adamw", "lars"])
def test_optimizer_state_dict_round_trip_resumes_the_run(kind):
    """save -> load into a FRESH optimiser over fresh copies of the parameters -> the next steps equal the
    uninterrupted run bit for bit (momentum / moments / AdamW step counters live in private arenas, not in
    torch's self.state); SGD and AdamW checkpoints are also interchangeable with torch.optim's."""
    from ssl_wafermap_amd import optim

    torch.manual_seed(0)
    shapes = [(64, 3, 7, 7), (64,), (128, 64, 3, 3), (5,)]
    init = [torch.randn(s) for s in shapes]

    def make(params):
        if kind == "sgd":
            return optim.SGD(params, lr=0.06, momentum=0.9, weight_decay=5e-4)
        if kind == "adamw":
            return optim.AdamW(params, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05)
        return optim.LARS(params, lr=0.2, momentum=0.9, weight_decay=1.5e-6)

    def grads(it):
        return [torch.randn(s, generator=torch.Generator().manual_seed(100 * it + i)) for i, s in enumerate(shapes)]

    def run(opt, params, its):
        for it in its:
            opt.zero_grad()
            for p, g in zip(params, grads(it)):
                p.grad.add_(g.to(DEV))
            opt.step()

    pa = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    oa = make(pa)
    run(oa, pa, range(3))
    sd = oa.state_dict()
    weights = [p.detach().clone() for p in pa]
    assert len(sd["state"]) == len(shapes)
    run(oa, pa, range(3, 6))

    pb = [torch.nn.Parameter(w.clone()) for w in weights]
    ob = make(pb)
    ob.load_state_dict(sd)
    run(ob, pb, range(3, 6))
    for a, b in zip(pa, pb):
        if kind == "lars":  # the per-parameter norms are summed with f32 atomics: last-bit differences run to run
            torch.testing.assert_close(a.detach(), b.detach(), atol=1e-6, rtol=1e-6)
        else:
            assert torch.equal(a.detach(), b.detach())

    if kind in ("sgd", "adamw"):  # the same checkpoint drives torch.optim to the same place
        pc = [torch.nn.Parameter(w.clone().cpu()) for w in weights]
        oc = (torch.optim.SGD(pc, lr=0.06, momentum=0.9, weight_decay=5e-4) if kind == "sgd"
              else torch.optim.AdamW(pc, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.05))
        oc.load_state_dict({"state": {k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in sd["state"].items()},
                            "param_groups": oc.state_dict()["param_groups"]})
        for it in range(3, 6):
            oc.zero_grad()
            for p, g in zip(pc, grads(it)):
                p.grad = g.clone()
            oc.step()
        for a, c in zip(pa, pc):
            torch.testing.assert_close(a.detach().cpu(), c.detach(), atol=2e-6, rtol=2e-6)
