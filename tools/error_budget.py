"""Error budget of the whole-step losses against the float32 oracle (GPU box; test infrastructure).

For each of the three steps the parity tests compare -- SimCLR ResNet-18 at bs 64 (reference
scripts/WM811k_benchmark.py:236-248), DINO ViT-Tiny (:578-588) and MAE ViT-S/16 (:902-947) -- the HIP forward is run
ONCE with every stage's output captured; row s of the table is the loss the float32 oracle produces when it CONTINUES
from the HIP path's output of stage s (everything before s in bf16 HIP kernels, everything after in float32).  The
difference between consecutive rows is what stage s adds to the loss error (signed, first order); the last row is the
HIP step's own loss.

    python tools/error_budget.py [simclr] [dino] [mae] [--out gpurun_out/error_budget.md]
"""
from __future__ import annotations

import argparse
import copy
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
DEV = "cuda:0"


def _bf(x):
    return x.bfloat16().float()


class Capture:
    """Ordered record of module outputs (forward hooks) / inputs (pre hooks)."""

    def __init__(self):
        self.rec, self.handles = [], []

    def out(self, module, name):
        self.handles.append(module.register_forward_hook(lambda m, a, o, n=name: self.rec.append((n, self._t(o)))))

    def inp(self, module, name):
        self.handles.append(module.register_forward_pre_hook(lambda m, a, n=name: self.rec.append((n, self._t(a[0])))))

    @staticmethod
    def _t(o):
        if isinstance(o, (tuple, list)):
            o = o[0]
        return o.detach().float()

    def close(self):
        for h in self.handles:
            h.remove()


def _table(title, rows, ref):
    """rows: [(stage, loss)] -> markdown lines with the signed relative error and the stage's increment."""
    out = [f"### {title}", "", f"float32 oracle loss: {ref:.7f}", "",
           "| HIP through stage | loss (oracle continues in f32) | (loss - ref) / ref | added by this stage |",
           "|---|---:|---:|---:|"]
    prev = 0.0
    for name, loss in rows:
        rel = (loss - ref) / ref
        out.append(f"| {name} | {loss:.7f} | {rel:+.2e} | {rel - prev:+.2e} |")
        prev = rel
    out.append("")
    return out


# ------------------------------------------------------------------------------------------------ SimCLR ResNet-18
def budget_simclr(B=64, seed=3):
    from oracle import resnet as orn
    from oracle.ntxent import ntxent_lightly
    from ssl_wafermap_amd.data import WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.models import SimCLR
    from ssl_wafermap_amd.transforms import BaseViewTransform, augment_views

    wafers, labels = synthetic_wafers(128, seed=seed)
    ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(), device=DEV)
    torch.manual_seed(0)
    model = SimCLR(None, 9, batch_size=B, max_epochs=150).to(DEV).train()
    sd0 = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    params = ds.transform.sample(ds.store, np.arange(B), np.random.default_rng(seed))
    views = ds.transform.launch(ds.store, params, B, "s2d_bf16")
    bb, hd = model.backbone, model.projection_head
    cap = Capture()
    cap.inp(bb.layer1, "stem (conv1 + bn1 + relu + maxpool)")
    for li in (1, 2, 3, 4):
        for bi in (0, 1):
            cap.out(getattr(bb, f"layer{li}")[bi], f"layer{li}.{bi}")
    cap.out(bb, "global average pool")
    cap.out(hd.layers[0], "head Linear 1")
    cap.out(hd.layers[1], "head BatchNorm1d 1 + ReLU")
    cap.out(hd.layers[3], "head Linear 2")
    cap.out(hd.layers[4], "head BatchNorm1d 2 (= z)")
    loss_hip = float(model.training_step((views, None), 0).detach())
    cap.close()

    v = augment_views(ds.store, params[0], fmt="nchw_f32", n_slots=2 * B).bfloat16().float().cpu()

    def sd():
        return {k: t.clone() for k, t in sd0.items()}

    def g(prefix, s):
        return {k[len(prefix):]: t for k, t in s.items() if k.startswith(prefix)}

    def stem(x, s):
        b = g("backbone.", s)
        x = F.conv2d(x, b["conv1.weight"], None, 2, 3)
        return F.max_pool2d(orn._bn(x, b, "bn1", True, relu=True), 3, 2, 1)

    fns = [stem]
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        fns.append(lambda x, s, li=li, stride=stride: orn._block(x, g("backbone.", s), f"layer{li}.0", stride, True))
        fns.append(lambda x, s, li=li: orn._block(x, g("backbone.", s), f"layer{li}.1", 1, True))
    fns.append(lambda x, s: F.adaptive_avg_pool2d(x, 1).flatten(1))
    fns.append(lambda x, s: F.linear(x, s["projection_head.layers.0.weight"]))
    fns.append(lambda x, s: orn._bn(x, g("projection_head.", s), "layers.1", True, relu=True))
    fns.append(lambda x, s: F.linear(x, s["projection_head.layers.3.weight"]))
    fns.append(lambda x, s: orn._bn(x, g("projection_head.", s), "layers.4", True))

    def continue_from(stage, x2b):
        """x2b: the stacked [2B, ...] output of stage `stage` (-1: the images); per-view BatchNorm statistics."""
        zs = []
        for half in (x2b[:B], x2b[B:]):
            s, x = sd(), half
            for fn in fns[stage + 1:]:
                x = fn(x, s)
            zs.append(x)
        return float(ntxent_lightly(zs[0], zs[1], 0.5))

    with torch.no_grad():
        ref = continue_from(-1, v)
        rows = []
        assert len(cap.rec) == len(fns), (len(cap.rec), len(fns))
        for i, (name, t) in enumerate(cap.rec):
            rows.append((name, continue_from(i, t.cpu())))
    rows.append(("NT-Xent (= the HIP step's loss)", loss_hip))
    return _table(f"SimCLR ResNet-18, bs {B}, two 224x224 views, seed {seed}", rows, ref)


# ------------------------------------------------------------------------------------------------ DINO ViT-Tiny
def budget_dino(b=8):
    from oracle import vit as ov
    from ssl_wafermap_amd import ops
    from ssl_wafermap_amd.models import DINOViT

    torch.manual_seed(0)
    model = DINOViT(None, 9, batch_size=8, max_epochs=10, log_rep_std=False, backbone="vit_tiny")
    with torch.no_grad():
        for p_ in model.backbone.parameters():
            if p_.dim() == 1:
                p_.add_(torch.randn_like(p_) * 0.05)
    model.teacher_backbone = copy.deepcopy(model.backbone)
    for p_ in model.teacher_backbone.parameters():
        p_.requires_grad = False
    nh = 3
    model = model.to(DEV).train()
    g = torch.Generator().manual_seed(11)
    views = [_bf(torch.randn(b, 3, 224, 224, generator=g)) for _ in range(2)] + \
            [_bf(torch.randn(b, 3, 96, 96, generator=g)) for _ in range(2)]
    sd = {k: v.detach().clone().float() for k, v in model.state_dict().items()}
    s_bb = {k[len("backbone."):]: v.clone() for k, v in sd.items() if k.startswith("backbone.")}
    s_hd = {k[len("head."):]: v.clone() for k, v in sd.items() if k.startswith("head.")}
    t_bb = {k[len("teacher_backbone."):]: v.clone() for k, v in sd.items() if k.startswith("teacher_backbone.")}
    t_hd = {k[len("teacher_head."):]: v.clone() for k, v in sd.items() if k.startswith("teacher_head.")}
    fl = lambda d: {k: v for k, v in d.items() if v.is_floating_point() and "running" not in k}
    ov.update_momentum(s_bb, t_bb, 0.99)
    ov.update_momentum(fl(s_hd), fl(t_hd), 0.99)
    vd = [v.to(DEV) for v in views]
    segs = [(2 * b, 197), (2 * b, 37)]
    depth = len(model.backbone.blocks)
    bn = "layers.1.weight" in s_hd and s_hd["layers.1.weight"].dim() == 1

    def hd():
        return {k: v.clone() for k, v in s_hd.items()}

    def tokens(x):
        w = s_bb["patch_embed.proj.weight"]
        p = w.shape[-1]
        t = F.conv2d(x, w, s_bb["patch_embed.proj.bias"], stride=p).flatten(2).transpose(1, 2)
        return torch.cat([s_bb["cls_token"].expand(t.shape[0], -1, -1), t], dim=1) + ov.pos_embed_for(s_bb["pos_embed"], x.shape[-1] // p)

    def split(rows):
        out, off = [], 0
        for n, s in segs:
            out.append(rows[off:off + n * s].reshape(n, s, -1))
            off += n * s
        return out

    # student stages on the list of per-resolution token tensors / the [4b, D] features / head activations
    st = [("patch embedding + class token + positions", lambda x, h: [tokens(torch.cat(vd[:2])), tokens(torch.cat(vd[2:]))])]
    for i in range(depth):
        st.append((f"blocks.{i}", lambda ts, h, i=i: [ov.block(t, s_bb, f"blocks.{i}", nh) for t in ts]))
    st.append(("final LayerNorm of the class tokens", lambda ts, h: torch.cat(
        [F.layer_norm(t[:, 0], (t.shape[-1],), s_bb["norm.weight"], s_bb["norm.bias"], 1e-6) for t in ts])))

    def bn_(x, key, h):
        return torch.cat([F.batch_norm(p, h[key + ".running_mean"], h[key + ".running_var"], h[key + ".weight"],
                                       h[key + ".bias"], True, 0.1, 1e-5) for p in x.chunk(4)])

    if bn:
        st.append(("head Linear 1 + BatchNorm1d + GELU", lambda y, h: F.gelu(bn_(F.linear(y, h["layers.0.weight"]), "layers.1", h))))
        st.append(("head Linear 2 + BatchNorm1d + GELU", lambda y, h: F.gelu(bn_(F.linear(y, h["layers.3.weight"]), "layers.4", h))))
        st.append(("head Linear 3 (bottleneck)", lambda y, h: F.linear(y, h["layers.6.weight"], h["layers.6.bias"])))
    else:
        st.append(("head Linear 1 + GELU", lambda y, h: F.gelu(F.linear(y, h["layers.0.weight"], h["layers.0.bias"]))))
        st.append(("head Linear 2 + GELU", lambda y, h: F.gelu(F.linear(y, h["layers.2.weight"], h["layers.2.bias"]))))
        st.append(("head Linear 3 (bottleneck)", lambda y, h: F.linear(y, h["layers.4.weight"], h["layers.4.bias"])))

    def last(y, h):
        v = h["last_layer.weight_v"]
        return F.linear(F.normalize(y, dim=-1), h["last_layer.weight_g"] * v / v.norm(dim=1, keepdim=True))

    st.append(("L2 normalise + weight-normalised last layer", last))

    with torch.no_grad():
        t_out_ref = [ov.dino_head(ov.vit_features(v, t_bb, nh), {k: v_.clone() for k, v_ in t_hd.items()}, training=True) for v in vd[:2]]

    def loss_of(student, teacher):
        return float(ov.dino_loss(teacher, list(student.chunk(4)), torch.zeros(1, 1, 2048, device=DEV), 0.04, 0.1)[0])

    def continue_from(stage, x):
        h = hd()
        for _, fn in st[stage + 1:]:
            x = fn(x, h)
        return x

    # ---- HIP, captured
    cap = Capture()
    bbm, hdm = model.backbone, model.head
    cap.inp(bbm.blocks[0], st[0][0])
    for i in range(depth):
        cap.out(bbm.blocks[i], f"blocks.{i}")
    cap.out(bbm.norm, st[depth + 1][0])
    act = [m for m in hdm.layers if type(m).__name__ == "GELU"]
    if bn:
        cap.out(act[0], st[depth + 2][0])
        cap.out(act[1], st[depth + 3][0])
        cap.out(hdm.layers[6], st[depth + 4][0])
    else:
        cap.out(hdm.layers[0], st[depth + 2][0])
        cap.out(hdm.layers[2], st[depth + 3][0])
        cap.out(hdm.layers[4], st[depth + 4][0])
    cap.out(hdm.last_layer, st[depth + 5][0])
    tcap = Capture()
    tcap.out(model.teacher_head, "teacher")
    batch = ([ops.to_nhwc_bf16(v) for v in vd], None)
    loss_hip = float(model.training_step(batch, 0).detach())
    cap.close(), tcap.close()
    rec = [r for r in cap.rec]
    assert len(rec) == len(st), (len(rec), len(st), [r[0] for r in rec])
    t_out_hip = list(tcap.rec[0][1].chunk(2))

    with torch.no_grad():
        ref = loss_of(continue_from(-1, None), t_out_ref)
        rows = [("teacher only (EMA + 2 global crops; student in f32)", loss_of(continue_from(-1, None), t_out_hip))]
        base = (rows[0][1] - ref)
        for i, (name, t) in enumerate(rec):
            x = split(t) if i <= depth else t
            rows.append((name + " [student; teacher f32]", loss_of(continue_from(i, x), t_out_ref)))
    rows.append(("DINO loss kernel, teacher + student in HIP (= the step's loss)", loss_hip))
    out = _table(f"DINO ViT-Tiny/16 (12 blocks), {b} wafers, 2 x 224 + 2 x 96 crops", rows[1:-1], ref)
    out.insert(-1, f"| teacher only in HIP (student f32) | {rows[0][1]:.7f} | {base / ref:+.2e} | -- |")
    out.insert(-1, f"| whole step in HIP | {loss_hip:.7f} | {(loss_hip - ref) / ref:+.2e} | -- |")
    return out


# ------------------------------------------------------------------------------------------------ MAE ViT-S/16
def budget_mae(b=8):
    from oracle import vit as ov
    from ssl_wafermap_amd import ops
    from ssl_wafermap_amd.models import MAE
    from ssl_wafermap_amd.utils import get_at_index, patchify, random_token_mask

    torch.manual_seed(0)
    model = MAE(None, 9, batch_size=8, log_rep_std=False, backbone="vit_small_16")
    seq, ps, eh, dh = 197, 16, 6, 16
    with torch.no_grad():
        model.mask_token.normal_(std=0.02)
        for p_ in model.parameters():
            if p_.dim() == 1:
                p_.add_(torch.randn_like(p_) * 0.02)
    model = model.to(DEV).train()
    g = torch.Generator().manual_seed(4)
    images = _bf(torch.randn(b, 3, 224, 224, generator=g)).to(DEV)
    keep, mask = random_token_mask((b, seq), 0.75, generator=g)
    keep, mask = keep.to(DEV), mask.to(DEV)
    sd = {k: v.detach().clone().float() for k, v in model.state_dict().items()}
    e = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    d = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    depth = len(model.backbone.encoder.layers)
    ce, cd = e["encoder.pos_embedding"].shape[-1], d["pos_embedding"].shape[-1]

    def tokens(_):
        p = e["conv_proj.weight"].shape[-1]
        t = F.conv2d(images, e["conv_proj.weight"], e["conv_proj.bias"], stride=p).flatten(2).transpose(1, 2)
        t = torch.cat([e["class_token"].expand(b, -1, -1), t], dim=1) + e["encoder.pos_embedding"]
        return torch.gather(t, 1, keep.unsqueeze(-1).expand(-1, -1, ce))

    def to_decoder(x):
        xm = sd["mask_token"].repeat(b, seq, 1).scatter(1, keep.unsqueeze(-1).expand(-1, -1, cd), x)
        return xm + d["pos_embedding"]

    patches = ov.lightly_patchify(images, ps)
    target = torch.gather(patches, 1, (mask - 1).unsqueeze(-1).expand(-1, -1, patches.shape[-1]))
    st = [("patch embedding + class token + positions + gather of the kept tokens", tokens)]
    for i in range(depth):
        st.append((f"encoder block {i}", lambda x, i=i: ov.tv_block(x, e, f"encoder.layers.encoder_layer_{i}", eh)))
    st.append(("encoder LayerNorm", lambda x: F.layer_norm(x, (ce,), e["encoder.ln.weight"], e["encoder.ln.bias"], 1e-6)))
    st.append(("decoder embed Linear", lambda x: F.linear(x, d["decoder_embed.weight"], d["decoder_embed.bias"])))
    st.append(("mask-token scatter + decoder positions + decoder block",
               lambda x: ov.tv_block(to_decoder(x), d, "layers.encoder_layer_0", dh)))
    st.append(("decoder LayerNorm", lambda x: F.layer_norm(x, (cd,), d["ln.weight"], d["ln.bias"], 1e-6)))
    st.append(("gather of the masked tokens + prediction Linear", lambda x: F.linear(
        torch.gather(x, 1, mask.unsqueeze(-1).expand(-1, -1, cd)), d["decoder_pred.weight"], d["decoder_pred.bias"])))

    def continue_from(stage, x):
        for _, fn in st[stage + 1:]:
            x = fn(x)
        return float(F.mse_loss(x, target))

    cap = Capture()
    enc, dec = model.backbone.encoder, model.decoder
    cap.inp(enc.layers[0], st[0][0])
    for i in range(depth):
        cap.out(enc.layers[i], f"encoder block {i}")
    cap.out(enc.ln, "encoder LayerNorm")
    cap.out(dec.decoder_embed, "decoder embed Linear")
    cap.out(dec.layers[0], st[depth + 3][0])
    cap.out(dec.ln, "decoder LayerNorm")
    cap.out(dec.decoder_pred, st[depth + 5][0])
    x_enc = model.forward_encoder(ops.to_nhwc_bf16(images), keep)
    pred = model.forward_decoder(x_enc, keep, mask)
    tgt = get_at_index(patchify(ops.to_nhwc_bf16(images), ps), mask - 1)
    loss_hip = float(model.criterion(pred, tgt).detach())
    cap.close()
    assert len(cap.rec) == len(st), (len(cap.rec), len(st), [r[0] for r in cap.rec])
    nk = keep.shape[1]
    with torch.no_grad():
        ref = continue_from(-1, None)
        rows = []
        for i, (name, t) in enumerate(cap.rec):
            if i <= depth + 1:
                x = t.reshape(b, nk, ce)
            elif i == depth + 2:
                x = t.reshape(b, nk, cd)
            elif i in (depth + 3, depth + 4):
                x = t.reshape(b, seq, cd)
            else:
                x = t.reshape(b, mask.shape[1], -1)
            rows.append((name, continue_from(i, x)))
    rows.append(("MSE kernel (= the HIP step's loss)", loss_hip))
    return _table(f"MAE ViT-S/16 (12 blocks, 49 of 197 tokens kept), {b} wafers", rows, ref)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["simclr", "dino", "mae"])
    ap.add_argument("--out", default="gpurun_out/error_budget.md")
    ap.add_argument("--seeds", type=int, nargs="*", default=[3, 4])
    a = ap.parse_args()
    lines = ["# Error budget of the whole-step losses vs the float32 oracle", "",
             "`python tools/error_budget.py` on one MI355X.  Row s: the float32 oracle continues from the HIP path's output of",
             "stage s; the last column is what the stage adds to the relative loss error (signed).", ""]
    if "simclr" in a.which:
        for s in a.seeds:
            lines += budget_simclr(64, s)
    if "dino" in a.which:
        lines += budget_dino()
    if "mae" in a.which:
        lines += budget_mae()
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
