"""CPU oracle for the vision-transformer path (DINO ViT-S/16, MAE pieces).  TEST INFRASTRUCTURE ONLY
(oracle/__init__.py).

torch float32 restatement of what the reference executes through third-party code:
  torch.hub "facebookresearch/dino:main" dino_vits16     scripts/WM811k_benchmark.py:548-550
  lightly heads.DINOProjectionHead(384, 2048, 256, 2048)  :553-559                           
  lightly.loss.DINOLoss(output_dim=2048)                  :564,586                           
  lightly utils.update_momentum                           :579-581                           
  torch.optim.AdamW(lr 1.5e-4*bs/256, wd 0.05, (0.9,0.95)) :591-598
Neither dino (floating branch) nor lightly (unpinned) is installed here and the reference holds no
test or golden vector for them: PARITY UNPINNED upstream.  Every function takes a state_dict with
the upstream key names so the HIP path and the oracle run on identical weights; tensors may live on
any device (the GPU tests run the oracle on the GPU in float32 to keep them fast).
"""
import math

import torch
import torch.nn.functional as F


def pos_embed_for(pos_embed, g_new):
    """dino interpolate_pos_encoding for a g_new x g_new patch grid."""
    n = pos_embed.shape[1] - 1
    if g_new * g_new == n:
        return pos_embed
    g = int(math.sqrt(n))
    dim = pos_embed.shape[-1]
    sf = (g_new + 0.1) / g
    patch = F.interpolate(pos_embed[:, 1:].reshape(1, g, g, dim).permute(0, 3, 1, 2), scale_factor=(sf, sf), mode="bicubic")
    patch = patch.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat([pos_embed[:, :1], patch], dim=1)


def attention(x, sd, p, heads):
    b, s, c = x.shape
    qkv = F.linear(x, sd[p + ".qkv.weight"], sd.get(p + ".qkv.bias")).reshape(b, s, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * ((c // heads) ** -0.5)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(b, s, c)
    return F.linear(out, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def block(x, sd, p, heads, eps=1e-6):
    c = x.shape[-1]
    x = x + attention(F.layer_norm(x, (c,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps), sd, p + ".attn", heads)
    h = F.layer_norm(x, (c,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps)
    h = F.gelu(F.linear(h, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"]))
    return x + F.linear(h, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])


def vit_features(x, sd, heads, prefix="", eps=1e-6):
    """x [N,3,S,S] float32 -> class-token features [N, D] (dino VisionTransformer.forward)."""
    g = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    w = g["patch_embed.proj.weight"]
    p = w.shape[-1]
    t = F.conv2d(x, w, g["patch_embed.proj.bias"], stride=p).flatten(2).transpose(1, 2)
    n = t.shape[0]
    t = torch.cat([g["cls_token"].expand(n, -1, -1), t], dim=1) + pos_embed_for(g["pos_embed"], x.shape[-1] // p)
    depth = 1 + max(int(k.split(".")[1]) for k in g if k.startswith("blocks."))
    for i in range(depth):
        t = block(t, g, f"blocks.{i}", heads, eps)
    c = t.shape[-1]
    return F.layer_norm(t, (c,), g["norm.weight"], g["norm.bias"], eps)[:, 0]


def dino_head(y, sd, prefix="", training=True, groups=1, momentum=0.1, eps=1e-5):
    """lightly DINOProjectionHead.  `groups`: the batch is `groups` views stacked view-major and
    BatchNorm statistics are taken per view (what separate per-view forward calls compute)."""
    g = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    has_bn = "layers.1.weight" in g and g["layers.1.weight"].dim() == 1

    def bn(x, key):
        parts = []
        for part in x.chunk(groups):
            parts.append(F.batch_norm(part, g[key + ".running_mean"], g[key + ".running_var"], g[key + ".weight"],
                                      g[key + ".bias"], training, momentum, eps))
        return torch.cat(parts)

    if has_bn:  # layers: 0 Linear, 1 BN, 2 GELU, 3 Linear, 4 BN (same module as 1), 5 GELU, 6 Linear
        x = F.gelu(bn(F.linear(y, g["layers.0.weight"]), "layers.1"))
        x = F.gelu(bn(F.linear(x, g["layers.3.weight"]), "layers.4"))
        x = F.linear(x, g["layers.6.weight"], g["layers.6.bias"])
    else:       # layers: 0 Linear, 1 GELU, 2 Linear, 3 GELU, 4 Linear
        x = F.gelu(F.linear(y, g["layers.0.weight"], g["layers.0.bias"]))
        x = F.gelu(F.linear(x, g["layers.2.weight"], g["layers.2.bias"]))
        x = F.linear(x, g["layers.4.weight"], g["layers.4.bias"])
    x = F.normalize(x, dim=-1, p=2)
    v = g["last_layer.weight_v"]
    w = g["last_layer.weight_g"] * v / v.norm(dim=1, keepdim=True)
    return F.linear(x, w)


def dino_loss(teacher_out, student_out, center, teacher_temp=0.04, student_temp=0.1):
    """lightly DINOLoss.forward on lists of [B, D] tensors; returns (loss, batch centre [1,1,D])."""
    t = torch.stack(list(teacher_out))
    t_out = F.softmax((t - center) / teacher_temp, dim=-1)
    s = torch.stack(list(student_out))
    s_out = F.log_softmax(s / student_temp, dim=-1)
    loss = -torch.einsum("tbd,sbd->ts", t_out, s_out)
    loss.fill_diagonal_(0)
    n_terms = loss.numel() - loss.diagonal().numel()
    batch_size = t.shape[1]
    return loss.sum() / (n_terms * batch_size), torch.mean(t, dim=(0, 1), keepdim=True)


def adamw_step(params, grads, state, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
    """torch.optim.AdamW (no amsgrad), in place on the dict of tensors."""
    b1, b2 = betas
    for k, p in params.items():
        g = grads[k]
        m, v = state.setdefault(k, (torch.zeros_like(p), torch.zeros_like(p)))
        p.mul_(1 - lr * weight_decay)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))


def update_momentum(params, params_ema, m):
    for k in params_ema:
        params_ema[k].mul_(m).add_(params[k], alpha=1 - m)


# ---------------------------------------------------------------------------------------- MAE
# torchvision.models.vit_b_32 + lightly masked_autoencoder.MAEBackbone / MAEDecoder as the reference's
# MAE drives them (scripts/WM811k_benchmark.py:876-957).  PARITY UNPINNED upstream.

def tv_block(x, sd, p, heads, eps=1e-6):
    """torchvision EncoderBlock (dropout 0): nn.MultiheadAttention(batch_first) + MLP, pre-norm."""
    c = x.shape[-1]
    h = F.layer_norm(x, (c,), sd[p + ".ln_1.weight"], sd[p + ".ln_1.bias"], eps)
    b, s, _ = h.shape
    qkv = F.linear(h, sd[p + ".self_attention.in_proj_weight"], sd[p + ".self_attention.in_proj_bias"])
    q, k, v = qkv.reshape(b, s, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    a = ((q @ k.transpose(-2, -1)) * ((c // heads) ** -0.5)).softmax(-1) @ v
    a = a.transpose(1, 2).reshape(b, s, c)
    x = x + F.linear(a, sd[p + ".self_attention.out_proj.weight"], sd[p + ".self_attention.out_proj.bias"])
    h = F.layer_norm(x, (c,), sd[p + ".ln_2.weight"], sd[p + ".ln_2.bias"], eps)
    h = F.gelu(F.linear(h, sd[p + ".mlp.0.weight"], sd[p + ".mlp.0.bias"]))
    return x + F.linear(h, sd[p + ".mlp.3.weight"], sd[p + ".mlp.3.bias"])


def _tv_layers(x, sd, prefix, heads):
    n = 1 + max(int(k[len(prefix):].split(".")[0].rsplit("_", 1)[1]) for k in sd if k.startswith(prefix + "encoder_layer_"))
    for i in range(n):
        x = tv_block(x, sd, f"{prefix}encoder_layer_{i}", heads)
    return x


def lightly_patchify(images, p):
    n, c, h, w = images.shape
    g = h // p
    x = images.reshape(n, c, g, p, g, p)
    return torch.einsum("nchpwq->nhwpqc", x).reshape(n, g * g, p * p * c)


def mae_encode(images, sd, idx_keep, heads=12, prefix="backbone."):
    g = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    p = g["conv_proj.weight"].shape[-1]
    t = F.conv2d(images, g["conv_proj.weight"], g["conv_proj.bias"], stride=p).flatten(2).transpose(1, 2)
    n = t.shape[0]
    t = torch.cat([g["class_token"].expand(n, -1, -1), t], dim=1) + g["encoder.pos_embedding"]
    if idx_keep is not None:
        t = torch.gather(t, 1, idx_keep.unsqueeze(-1).expand(-1, -1, t.shape[-1]))
    t = _tv_layers(t, g, "encoder.layers.", heads)
    c = t.shape[-1]
    return F.layer_norm(t, (c,), g["encoder.ln.weight"], g["encoder.ln.bias"], 1e-6)


def mae_loss(images, sd, idx_keep, idx_mask, enc_heads=12, dec_heads=16):
    """The reference's MAE.training_step for given token indices (:925-954)."""
    x_enc = mae_encode(images, sd, idx_keep, enc_heads)
    d = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    x_dec = F.linear(x_enc, d["decoder_embed.weight"], d["decoder_embed.bias"])
    b, seq = images.shape[0], d["pos_embedding"].shape[1]
    c = x_dec.shape[-1]
    x_masked = sd["mask_token"].repeat(b, seq, 1)
    x_masked = x_masked.scatter(1, idx_keep.unsqueeze(-1).expand(-1, -1, c), x_dec)
    t = _tv_layers(x_masked + d["pos_embedding"], d, "layers.", dec_heads)
    t = F.layer_norm(t, (c,), d["ln.weight"], d["ln.bias"], 1e-6)
    pred = torch.gather(t, 1, idx_mask.unsqueeze(-1).expand(-1, -1, c))
    pred = F.linear(pred, d["decoder_pred.weight"], d["decoder_pred.bias"])
    p = sd["backbone.conv_proj.weight"].shape[-1]
    patches = lightly_patchify(images, p)
    target = torch.gather(patches, 1, (idx_mask - 1).unsqueeze(-1).expand(-1, -1, patches.shape[-1]))
    return F.mse_loss(pred, target)


def simmim_loss(images, sd, idx_mask, heads=12):
    """The reference's SimMIM.training_step for given masked indices (scripts/WM811k_benchmark.py:979-1014):
    tokens (class token prepended, NO positional embedding yet) -> masked ones replaced by the mask token ->
    encoder (adds the positional embedding) -> Linear decoder on the masked tokens -> L1 against the patches."""
    g = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    p = g["conv_proj.weight"].shape[-1]
    t = F.conv2d(images, g["conv_proj.weight"], g["conv_proj.bias"], stride=p).flatten(2).transpose(1, 2)
    n = t.shape[0]
    t = torch.cat([g["class_token"].expand(n, -1, -1), t], dim=1)
    c = t.shape[-1]
    idx = idx_mask.unsqueeze(-1).expand(-1, -1, c)
    t = t.scatter(1, idx, sd["mask_token"].expand(n, idx_mask.shape[1], c))
    t = t + g["encoder.pos_embedding"]
    t = _tv_layers(t, g, "encoder.layers.", heads)
    t = F.layer_norm(t, (c,), g["encoder.ln.weight"], g["encoder.ln.bias"], 1e-6)
    pred = F.linear(torch.gather(t, 1, idx), sd["decoder.weight"], sd["decoder.bias"])
    patches = lightly_patchify(images, p)
    target = torch.gather(patches, 1, (idx_mask - 1).unsqueeze(-1).expand(-1, -1, patches.shape[-1]))
    return F.l1_loss(pred, target)
