"""Vision transformer on the HIP kernels, with facebookresearch/dino's module tree and state_dict
keys (`cls_token`, `pos_embed`, `patch_embed.proj.*`, `blocks.N.{norm1,attn.qkv,attn.proj,norm2,
mlp.fc1,mlp.fc2}.*`, `norm.*`) — the backbone the reference loads with
`torch.hub.load("facebookresearch/dino:main", "dino_vits16", pretrained=False)`
(scripts/WM811k_benchmark.py:548-550; MixedWM38_pretrain.py:141-143).

Tokens travel as bf16 [images * tokens, dim] (token-major rows), which is what every kernel on this
path takes; an image batch is bf16 channels_last as the augmentation kernel emits it.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn as nn

from .. import nn as hnn
from .. import vit_ops


class Attention(nn.Module):
    def __init__(self, dim: int, num_heads: int, qkv_bias: bool = True):
        super().__init__()
        if dim != num_heads * 64:
            raise ValueError("the attention kernel is built for head dim 64 (ViT-S/16: 6 x 64, ViT-B: 12 x 64)")
        self.num_heads = num_heads
        self.scale = 64 ** -0.5
        self.qkv = hnn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = hnn.Linear(dim, dim, bias=True)

    def forward(self, x, batch: int, seq: int, residual, segments=None, qkv=None):
        if qkv is None:
            qkv = self.qkv(x)
        if segments is not None:  # several (batch, seq) groups row-concatenated in x
            a = vit_ops.attention_segments(qkv, segments, self.num_heads, self.scale)
        else:
            a = vit_ops.attention(qkv, batch, seq, self.num_heads, self.scale)
        return self.proj(a, residual=residual)


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = hnn.Linear(dim, hidden, bias=True)
        self.act = hnn.GELU()
        self.fc2 = hnn.Linear(hidden, dim, bias=True)

    def forward(self, x, residual):
        return vit_ops.mlp_gelu(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, residual)


class Block(nn.Module):
    """Pre-norm block: x += proj(attn(norm1(x))); x += fc2(gelu(fc1(norm2(x)))).  The residual adds
    ride in the epilogue pass of proj / fc2."""

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = True, eps: float = 1e-6):
        super().__init__()
        self.norm1 = hnn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.norm2 = hnn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, batch: int, seq: int, segments=None):
        # passes that record no gradient (EMA teacher, inference) on shapes the fused kernels serve: LayerNorm folded
        # into the qkv GEMM and into the one-launch MLP -- the normalised rows never reach memory
        qkv = vit_ops.ln_linear(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, self.attn.qkv.weight, self.attn.qkv.bias)
        if qkv is not None:
            x = self.attn(None, batch, seq, residual=x, segments=segments, qkv=qkv)
        else:
            h, skip = self.norm1.forward_skip(x)
            x = self.attn(h, batch, seq, residual=skip, segments=segments)
        y = vit_ops.ln_mlp_gelu(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, self.mlp.fc1.weight, self.mlp.fc1.bias,
                                self.mlp.fc2.weight, self.mlp.fc2.bias)
        if y is not None:
            return y
        h, skip = self.norm2.forward_skip(x)
        return self.mlp(h, residual=skip)


class PatchEmbed(nn.Module):
    def __init__(self, img_size: int = 224, patch_size: int = 16, in_chans: int = 3, embed_dim: int = 768):
        super().__init__()
        if in_chans != 3:
            raise ValueError("patch embedding is built for 3-channel wafer images")
        self.img_size, self.patch_size = img_size, patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = _PatchConv(embed_dim, patch_size)

    def forward(self, x):
        return self.proj(x)


class _PatchConv(nn.Module):
    """nn.Conv2d(3, D, kernel_size=p, stride=p) parameters (weight [D,3,p,p], bias [D])."""

    def __init__(self, embed_dim: int, patch_size: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(embed_dim, 3, patch_size, patch_size))
        self.bias = nn.Parameter(torch.empty(embed_dim))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(3 * patch_size * patch_size)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return vit_ops.bias_act(vit_ops.patch_embed(x, self.weight), self.bias)


class VisionTransformer(nn.Module):
    def __init__(self, img_size: int = 224, patch_size: int = 16, embed_dim: int = 768, depth: int = 12,
                 num_heads: int = 12, mlp_ratio: float = 4.0, qkv_bias: bool = True, eps: float = 1e-6):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size, patch_size, 3, embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias, eps) for _ in range(depth)])
        self.norm = hnn.LayerNorm(embed_dim, eps=eps)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        for m in self.modules():  # dino's _init_weights
            if isinstance(m, hnn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        self._interp: Dict[Tuple[int, int], torch.Tensor] = {}

    # ---- positional embedding for other crop sizes (dino: interpolate_pos_encoding) -------------
    def _interp_matrix(self, g_new: int) -> torch.Tensor:
        """[1 + g_new^2, 1 + g^2] matrix of dino's interpolate_pos_encoding: row 0 passes the class-token position through,
        the other rows are dino's bicubic resize of the patch position grid (scale factor (g_new + 0.1) / g), built once
        per crop size by resizing the identity basis on the host.  As ONE product with the whole embedding the resize
        needs no slice / cat of the parameter (each a framework kernel forward and backward)."""
        n = self.pos_embed.shape[1] - 1
        g = int(math.sqrt(n))
        key = (g, g_new)
        m = self._interp.get(key)
        if m is None or m.device != self.pos_embed.device:
            basis = torch.eye(n, dtype=torch.float32).reshape(1, g, g, n).permute(0, 3, 1, 2)
            sf = (g_new + 0.1) / g
            out = nn.functional.interpolate(basis, scale_factor=(sf, sf), mode="bicubic")
            if out.shape[-1] != g_new or out.shape[-2] != g_new:
                raise RuntimeError(f"pos-embed resize {g}->{g_new} produced {tuple(out.shape)}")
            full = torch.zeros(g_new * g_new + 1, n + 1, dtype=torch.float32)
            full[0, 0] = 1.0
            full[1:, 1:] = out.permute(0, 2, 3, 1).reshape(g_new * g_new, n)
            m = full.contiguous().to(self.pos_embed.device)
            self._interp[key] = m
        return m

    def pos_for(self, g_new: int) -> torch.Tensor:
        n = self.pos_embed.shape[1] - 1
        if g_new * g_new == n:
            return self.pos_embed
        # [1 + g_new^2, D]: a 37 x 197 x D product, parameter-side
        return vit_ops.const_matmul(self._interp_matrix(g_new), self.pos_embed).unsqueeze(0)

    # ---- forward ----------------------------------------------------------------------------------
    def prepare_tokens(self, x):
        n, _, s, _ = x.shape
        p = self.patch_embed.patch_size
        g = s // p
        patches = self.patch_embed(x)
        return vit_ops.tokens_assemble(patches, self.cls_token, self.pos_for(g), n, g * g), n, g * g + 1

    def forward_multi(self, xs):
        """A list of image batches of DIFFERENT resolutions (DINO's multi-crop student: [2B, 3, 224, 224] and
        [6B, 3, 96, 96]) -> class-token features of all of them, [sum N_g, D] in list order.  dino runs the backbone
        once per resolution; here the token rows of all groups are concatenated, so every per-token layer (LayerNorm,
        the four Linear layers of a block and their weight gradients) is ONE launch over all rows, and only the attention
        itself runs per group (vit_ops.attention_segments).  Per-row arithmetic is unchanged."""
        if len(xs) == 1:
            return self.forward(xs[0])
        groups, segments = [], []
        p = self.patch_embed.patch_size
        for x in xs:
            n, g = x.shape[0], x.shape[-1] // p
            groups.append((self.patch_embed(x), self.pos_for(g), n, g * g))
            segments.append((n, g * g + 1))
        # every group's token rows written into ONE buffer (no concatenation pass)
        tok = vit_ops.tokens_assemble_multi(groups, self.cls_token)
        for blk in self.blocks:
            tok = blk(tok, 0, 0, segments=segments)
        # the class-token rows of all segments in ONE gather over the concatenated rows (one zero-filled gradient buffer
        # and one scatter in the backward pass, instead of a slice + gather per segment)
        rows = tok.shape[0]

        def build():
            parts, off = [], 0
            for n, seq in segments:
                parts.append(off + torch.arange(n, dtype=torch.int64, device=tok.device) * seq)
                off += n * seq
            return torch.cat(parts).reshape(1, -1).contiguous()

        idx = vit_ops.cached_index(("cls_rows", tuple(segments), str(tok.device)), build)
        return self.norm(vit_ops.gather_rows(tok, idx, 1, rows))

    def forward(self, x):
        """images [N, 3, S, S] -> class-token features [N, D] (bf16), as dino's forward()."""
        tok, n, seq = self.prepare_tokens(x)
        for blk in self.blocks:
            tok = blk(tok, n, seq)
        # LayerNorm is per token: normalise only the class tokens
        idx = vit_ops.cached_index(("cls0", n, str(tok.device)), lambda: torch.zeros((n, 1), dtype=torch.int64, device=tok.device))
        cls = vit_ops.gather_rows(tok, idx, n, seq)
        return self.norm(cls)


def vit_small(patch_size: int = 16, **kw) -> VisionTransformer:
    """dino_vits16: dim 384, depth 12, 6 heads, MLP x4, qkv bias, LayerNorm eps 1e-6."""
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True,
                             eps=1e-6, **kw)


def vit_tiny(patch_size: int = 16, **kw) -> VisionTransformer:
    """dino's vit_tiny: dim 192, depth 12, 3 heads (head dim 64), MLP x4 -- the shape the reference notes as the
    lighter DINO backbone (scripts/WM811k_benchmark.py:668-669) and BASELINE.json configs[2] names."""
    return VisionTransformer(patch_size=patch_size, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4, qkv_bias=True,
                             eps=1e-6, **kw)


def vit_base(patch_size: int = 16, **kw) -> VisionTransformer:
    return VisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                             eps=1e-6, **kw)
