"""Masked auto-encoder on ViT-B/32: the reference's MAE (scripts/WM811k_benchmark.py:876-963;
MixedWM38_pretrain.py:257-340) with lightly's `masked_autoencoder.MAEBackbone / MAEDecoder` and
torchvision's `vit_b_32` module trees and state_dict keys:

  backbone: class_token, conv_proj.{weight,bias}, encoder.pos_embedding,
            encoder.layers.encoder_layer_N.{ln_1, self_attention.{in_proj_weight,in_proj_bias,out_proj},
            ln_2, mlp.{0,3}}.*, encoder.ln.*
  decoder:  decoder_embed.*, pos_embedding, layers.encoder_layer_N.*, ln.*, decoder_pred.*

One step: random 75 % token mask -> encoder on the 12 kept tokens (class token always kept) ->
decoder embed -> scatter into mask tokens -> 1-block decoder on all 50 -> predict the 38 masked
patches -> MSE against patchify(images, 32).  Tokens travel as bf16 [B*S, C] rows.
"""
from __future__ import annotations

from collections import OrderedDict
from types import SimpleNamespace

import torch
import torch.nn as nn

from .. import nn as hnn
from .. import ops, optim, vit_ops
from ..utils import debug, model_utils, scheduler
from .knn import KNNBenchmarkModule
from .vit import _PatchConv


class _SelfAttention(nn.Module):
    """torch.nn.MultiheadAttention's parameters (in_proj_weight [3D, D], in_proj_bias, out_proj)."""

    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        if dim % num_heads or dim // num_heads not in (32, 64):
            raise ValueError("the attention kernel is built for head dims 64 and 32")
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dim))
        self.out_proj = hnn.Linear(dim, dim, bias=True)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def forward(self, x, batch, seq, residual):
        qkv = vit_ops.linear(x, self.in_proj_weight, self.in_proj_bias)
        a = vit_ops.attention(qkv, batch, seq, self.num_heads, head_dim=self.head_dim)
        return self.out_proj(a, residual=residual)


class _Dropout0(nn.Module):
    """Dropout(p = 0) placeholder keeping torchvision's Sequential indices (mlp.0, mlp.3)."""

    def forward(self, x):
        return x


class EncoderBlock(nn.Module):
    """torchvision.models.vision_transformer.EncoderBlock (pre-norm, eps 1e-6, dropout 0)."""

    def __init__(self, num_heads: int, hidden_dim: int, mlp_dim: int):
        super().__init__()
        self.ln_1 = hnn.LayerNorm(hidden_dim, eps=1e-6)
        self.self_attention = _SelfAttention(hidden_dim, num_heads)
        self.ln_2 = hnn.LayerNorm(hidden_dim, eps=1e-6)
        self.mlp = nn.Sequential(hnn.Linear(hidden_dim, mlp_dim, bias=True), hnn.GELU(), _Dropout0(),
                                 hnn.Linear(mlp_dim, hidden_dim, bias=True), _Dropout0())
        for m in (self.mlp[0], self.mlp[3]):
            nn.init.xavier_uniform_(m.weight)
            nn.init.normal_(m.bias, std=1e-6)

    def forward(self, x, batch, seq):
        h, skip = self.ln_1.forward_skip(x)
        x = self.self_attention(h, batch, seq, residual=skip)
        h, skip = self.ln_2.forward_skip(x)
        return vit_ops.mlp_gelu(h, self.mlp[0].weight, self.mlp[0].bias, self.mlp[3].weight, self.mlp[3].bias, skip)


class _Encoder(nn.Module):
    def __init__(self, seq_length: int, num_layers: int, num_heads: int, hidden_dim: int, mlp_dim: int):
        super().__init__()
        self.pos_embedding = nn.Parameter(torch.empty(1, seq_length, hidden_dim).normal_(std=0.02))
        self.layers = nn.Sequential(OrderedDict(
            (f"encoder_layer_{i}", EncoderBlock(num_heads, hidden_dim, mlp_dim)) for i in range(num_layers)))
        self.ln = hnn.LayerNorm(hidden_dim, eps=1e-6)

    def run_layers(self, x, batch, seq):
        for blk in self.layers:
            x = blk(x, batch, seq)
        return self.ln(x)


def vit_b_32():
    """Geometry of torchvision.models.vit_b_32() (the reference builds it only to hand it to
    MAEBackbone.from_vit, scripts/WM811k_benchmark.py:881,888)."""
    return SimpleNamespace(image_size=224, patch_size=32, num_layers=12, num_heads=12, hidden_dim=768, mlp_dim=3072,
                           seq_length=50)


def vit_small_16():
    """ViT-S/16 geometry (384-d, 6 heads, 197 tokens): BASELINE.json configs[3], MAE on MixedWM38."""
    return SimpleNamespace(image_size=224, patch_size=16, num_layers=12, num_heads=6, hidden_dim=384, mlp_dim=1536,
                           seq_length=197)


def vit_tiny_16():
    return SimpleNamespace(image_size=224, patch_size=16, num_layers=12, num_heads=3, hidden_dim=192, mlp_dim=768,
                           seq_length=197)


VIT_GEOMETRIES = {"vit_b_32": vit_b_32, "vit_small_16": vit_small_16, "vit_tiny_16": vit_tiny_16}


class MAEBackbone(nn.Module):
    def __init__(self, image_size=224, patch_size=32, num_layers=12, num_heads=12, hidden_dim=768, mlp_dim=3072):
        super().__init__()
        self.image_size, self.patch_size, self.hidden_dim = image_size, patch_size, hidden_dim
        self.seq_length = (image_size // patch_size) ** 2 + 1
        self.conv_proj = _PatchConv(hidden_dim, patch_size)
        self.class_token = nn.Parameter(torch.zeros(1, 1, hidden_dim))
        self.encoder = _Encoder(self.seq_length, num_layers, num_heads, hidden_dim, mlp_dim)

    @classmethod
    def from_vit(cls, vit) -> "MAEBackbone":
        return cls(vit.image_size, vit.patch_size, vit.num_layers, vit.num_heads, vit.hidden_dim, vit.mlp_dim)

    def _pos_for(self, g_new: int) -> torch.Tensor:
        """Positional embedding for a g_new x g_new patch grid (lightly's MAEEncoder.interpolate_pos_encoding:
        the same bicubic resize as dino's; MSN's 96^2 focal crops need it)."""
        import math

        pos = self.encoder.pos_embedding
        n = pos.shape[1] - 1
        if g_new * g_new == n:
            return pos
        g = int(math.sqrt(n))
        key = (g, g_new)
        cache = self.__dict__.setdefault("_interp", {})
        m = cache.get(key)
        if m is None or m.device != pos.device:
            basis = torch.eye(n, dtype=torch.float32).reshape(1, g, g, n).permute(0, 3, 1, 2)
            sf = (g_new + 0.1) / g
            out = nn.functional.interpolate(basis, scale_factor=(sf, sf), mode="bicubic")
            full = torch.zeros(g_new * g_new + 1, n + 1, dtype=torch.float32)   # row 0: the class-token position passes through
            full[0, 0] = 1.0
            full[1:, 1:] = out.permute(0, 2, 3, 1).reshape(g_new * g_new, n)
            m = full.contiguous().to(pos.device)
            cache[key] = m
        return vit_ops.const_matmul(m, pos).unsqueeze(0)

    def images_to_tokens(self, images, add_pos: bool = True):
        """[B,3,S,S] -> bf16 [B*seq, D] rows: class token prepended; the positional embedding is added here
        (add_pos) or later by `encode_tokens` (lightly's encoder adds it: SimMIM masks tokens before that)."""
        n, g = images.shape[0], images.shape[-1] // self.patch_size
        patches = self.conv_proj(images)
        pos = self._pos_for(g)
        if not add_pos:
            pos = torch.zeros_like(pos)
        return vit_ops.tokens_assemble(patches, self.class_token, pos, n, g * g)

    def encode_tokens(self, tokens, batch: int):
        """lightly MAEEncoder.forward(tokens): + positional embedding, encoder blocks, final LayerNorm."""
        seq, c = self.seq_length, self.hidden_dim
        pos = self.encoder.pos_embedding.to(ops.act_dtype()).expand(batch, seq, c).reshape(batch * seq, c)
        t = vit_ops.bias_act(tokens.reshape(batch * seq, c), None, vit_ops.ACT_NONE, residual=pos)
        return self.encoder.run_layers(t, batch, seq).view(batch, seq, c)

    def encode(self, images, idx_keep=None):
        """All kept tokens after the encoder: [B, K, D] (lightly MAEBackbone.encode)."""
        n = images.shape[0]
        tok = self.images_to_tokens(images)
        seq = (images.shape[-1] // self.patch_size) ** 2 + 1
        if idx_keep is not None:
            tok = vit_ops.gather_rows(tok, idx_keep, n, seq)
            seq = idx_keep.shape[1]
        out = self.encoder.run_layers(tok, n, seq)
        return out.view(n, seq, self.hidden_dim)

    def forward(self, images, idx_keep=None):
        """Class-token feature [B, D] (what the kNN evaluation embeds)."""
        return self.encode(images, idx_keep)[:, 0]


class MAEDecoder(nn.Module):
    def __init__(self, seq_length: int, num_layers: int, num_heads: int, embed_input_dim: int, hidden_dim: int,
                 mlp_dim: int, out_dim: int, dropout: float = 0.0, attention_dropout: float = 0.0):
        super().__init__()
        if dropout or attention_dropout:
            raise NotImplementedError("MAEDecoder: dropout > 0 has no HIP path (the reference uses 0)")
        self.seq_length, self.hidden_dim = seq_length, hidden_dim
        self.decoder_embed = hnn.Linear(embed_input_dim, hidden_dim, bias=True)
        self.pos_embedding = nn.Parameter(torch.empty(1, seq_length, hidden_dim).normal_(std=0.02))
        self.layers = nn.Sequential(OrderedDict(
            (f"encoder_layer_{i}", EncoderBlock(num_heads, hidden_dim, mlp_dim)) for i in range(num_layers)))
        self.ln = hnn.LayerNorm(hidden_dim, eps=1e-6)
        self.decoder_pred = hnn.Linear(hidden_dim, out_dim, bias=True)
        for m in (self.decoder_embed, self.decoder_pred):
            nn.init.xavier_uniform_(m.weight)
            nn.init.constant_(m.bias, 0.0)

    def embed(self, x):
        b, k, c = x.shape
        return self.decoder_embed(x.reshape(b * k, c)).view(b, k, self.hidden_dim)

    def decode(self, x):
        b, s, c = x.shape
        pos = self.pos_embedding.to(ops.act_dtype()).expand(b, s, c).reshape(b * s, c)
        t = vit_ops.bias_act(x.reshape(b * s, c), None, vit_ops.ACT_NONE, residual=pos)
        for blk in self.layers:
            t = blk(t, b, s)
        return self.ln(t).view(b, s, c)

    def predict(self, x):
        b, k, c = x.shape
        return self.decoder_pred(x.reshape(b * k, c)).view(b, k, -1)

    def forward(self, x):
        return self.predict(self.decode(self.embed(x)))


masked_autoencoder = SimpleNamespace(MAEBackbone=MAEBackbone, MAEDecoder=MAEDecoder)


class MAE(KNNBenchmarkModule):
    def __init__(self, dataloader_kNN=None, num_classes=9, batch_size: int = 64, max_epochs: int = 150,
                 log_rep_std: bool = True, backbone: str = "vit_b_32", **kwargs):
        super().__init__(dataloader_kNN, num_classes, **kwargs)
        decoder_dim = 512
        # "vit_b_32" = the reference's torchvision.models.vit_b_32() (both scripts); "vit_small_16" = BASELINE.json
        # configs[3]: 197 tokens, 49 kept at mask ratio 0.75, 16 x 16 x 3 = 768 predicted values per masked patch
        if backbone not in VIT_GEOMETRIES:
            raise ValueError(f"MAE: unknown backbone {backbone!r} (have {sorted(VIT_GEOMETRIES)})")
        vit = VIT_GEOMETRIES[backbone]()
        self.warmup_epochs = 40 if max_epochs >= 800 else 20
        self.mask_ratio = 0.75
        self.patch_size = vit.patch_size
        self.sequence_length = vit.seq_length
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_dim))
        self.backbone = MAEBackbone.from_vit(vit)
        self.decoder = MAEDecoder(seq_length=vit.seq_length, num_layers=1, num_heads=16, embed_input_dim=vit.hidden_dim,
                                  hidden_dim=decoder_dim, mlp_dim=decoder_dim * 4, out_dim=vit.patch_size ** 2 * 3,
                                  dropout=0, attention_dropout=0)
        self.lr_factor = batch_size / 256
        self.max_epochs = max_epochs
        self.log_rep_std = log_rep_std

    def criterion(self, pred, target):
        return vit_ops.mse_loss(pred.reshape(-1, pred.shape[-1]), target.reshape(-1, target.shape[-1]))

    def forward_encoder(self, images, idx_keep=None):
        out = self.backbone.encode(images, idx_keep)
        if self.log_rep_std:
            self.log("rep_std", debug.std_of_l2_normalized(out.detach().flatten(1)))
        return out

    def forward_decoder(self, x_encoded, idx_keep, idx_mask):
        batch_size = x_encoded.shape[0]
        x_decode = self.decoder.embed(x_encoded)
        x_masked = model_utils.repeat_token(self.mask_token.to(ops.act_dtype()), (batch_size, self.sequence_length))
        x_masked = model_utils.set_at_index(x_masked, idx_keep, x_decode)
        x_decoded = self.decoder.decode(x_masked)
        x_pred = model_utils.get_at_index(x_decoded, idx_mask)
        return self.decoder.predict(x_pred)

    def training_step(self, batch, batch_idx, generator: torch.Generator = None):
        images = batch[0]
        if not (torch.is_tensor(images) and images.dim() == 4):
            images = images[0]  # the list of views of a one-view transform (the reference's form)
        batch_size = images.shape[0]
        idx_keep, idx_mask = model_utils.random_token_mask(size=(batch_size, self.sequence_length),
                                                           mask_ratio=self.mask_ratio, device=images.device,
                                                           generator=generator)
        x_encoded = self.forward_encoder(images, idx_keep)
        x_pred = self.forward_decoder(x_encoded, idx_keep, idx_mask)
        patches = model_utils.patchify(images, self.patch_size)
        target = model_utils.get_at_index(patches, idx_mask - 1)  # patches carry no class token
        loss = self.criterion(x_pred, target)
        self.log("train_loss_ssl", loss)
        return loss

    def configure_optimizers(self):
        opt = optim.AdamW(self.parameters(), lr=1.5e-4 * self.lr_factor, weight_decay=0.05, betas=(0.9, 0.95))
        cosine = scheduler.CosineWarmupScheduler(opt, self.warmup_epochs, self.max_epochs)
        return [opt], [cosine]


class SimMIM(KNNBenchmarkModule):
    """SimMIM on ViT-B/32 (scripts/WM811k_benchmark.py:960-1030): every token goes through the encoder, the
    masked ones replaced by a learned mask token BEFORE the positional embedding is added; a Linear decoder
    predicts the masked patches; L1 loss; AdamW lr 8e-4 x bs/256."""

    def __init__(self, dataloader_kNN=None, num_classes=9, batch_size: int = 64, max_epochs: int = 150, **kwargs):
        kwargs.pop("log_rep_std", None)
        super().__init__(dataloader_kNN, num_classes, **kwargs)
        vit = vit_b_32()
        self.warmup_epochs = 40 if max_epochs >= 800 else 20
        self.mask_ratio = 0.75
        self.patch_size = vit.patch_size
        self.sequence_length = vit.seq_length
        self.mask_token = nn.Parameter(torch.zeros(1, 1, vit.hidden_dim))
        self.backbone = MAEBackbone.from_vit(vit)
        self.decoder = hnn.Linear(vit.hidden_dim, vit.patch_size ** 2 * 3, bias=True)
        self.lr_factor = batch_size / 256
        self.max_epochs = max_epochs

    def criterion(self, pred, target):
        return vit_ops.l1_loss(pred.reshape(-1, pred.shape[-1]), target.reshape(-1, target.shape[-1]))

    def forward_encoder(self, images, batch_size, idx_mask):
        tokens = self.backbone.images_to_tokens(images, add_pos=False).view(batch_size, self.sequence_length, -1)
        tokens_masked = model_utils.mask_at_index(tokens, idx_mask, self.mask_token)
        return self.backbone.encode_tokens(tokens_masked, batch_size)

    def forward_decoder(self, x_encoded):
        b, k, c = x_encoded.shape
        return self.decoder(x_encoded.reshape(b * k, c)).view(b, k, -1)

    def training_step(self, batch, batch_idx, generator: torch.Generator = None):
        images = batch[0]
        if not (torch.is_tensor(images) and images.dim() == 4):
            images = images[0]
        batch_size = images.shape[0]
        _, idx_mask = model_utils.random_token_mask(size=(batch_size, self.sequence_length), mask_ratio=self.mask_ratio,
                                                    device=images.device, generator=generator)
        x_encoded = self.forward_encoder(images, batch_size, idx_mask)
        x_out = self.forward_decoder(model_utils.get_at_index(x_encoded, idx_mask))
        patches = model_utils.patchify(images, self.patch_size)
        target = model_utils.get_at_index(patches, idx_mask - 1)
        loss = self.criterion(x_out, target)
        self.log("train_loss_ssl", loss)
        return loss

    def configure_optimizers(self):
        opt = optim.AdamW(self.parameters(), lr=8e-4 * self.lr_factor, weight_decay=0.05, betas=(0.9, 0.999))
        return [opt], [scheduler.CosineWarmupScheduler(opt, self.warmup_epochs, self.max_epochs)]
