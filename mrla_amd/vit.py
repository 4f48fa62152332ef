"""DeiT / ViT + MRLA-light with the reference's API surface (deit/deit_mrla_light.py:42-471): same class and
factory names, constructor keywords and state_dict keys; attention / MLP / patch embedding are stock PyTorch,
the MRLA term of every block (`x + mrla(x, o_prev)`, :234) is one fused HIP op."""
import math
from collections import OrderedDict
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers

try:                                    # optional: expose the factories to timm.create_model when timm exists
    from timm.models.registry import register_model
except Exception:                       # pragma: no cover - timm is not installed in this image
    def register_model(fn):
        return fn

__all__ = ["deit_mrlal_tiny_patch16_224", "deit_mrlal_small_patch16_224", "deit_mrlal_base_patch16_224",
           "deit_mrlab_tiny_patch16_224", "deit_mrlab_small_patch16_224", "deit_mrlab_base_patch16_224"]


def _pair(v):
    return v if isinstance(v, tuple) else (v, v)


class _LinearFn(torch.autograd.Function):
    """F.linear whose bias gradient is a GEMM (ones^T dY) instead of torch's column-sum reduction.

    Why: replayed from a HIP graph, the training step of these DeiT models returned NaN in a random handful of nn.Linear
    BIAS gradients from the second replay on (weights' gradients of the same layers finite; eager launches always right) --
    with bf16 autocast, stochastic depth on and self-attention in the block; none of this package's kernels involved (the
    same blocks without the MRLA module show it; scripts/archive/deit_replay_debug*.py, profiles/r05_notes.md).  Taking the bias
    gradient through matrix products (ones^T dY per image, then over the images; fp32 accumulation inside each) removes it; everything else is the stock linear: forward with the fused bias epilogue, dX = dY W, dW = dY^T X."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        dx = (dy2 @ w).view(x.shape) if ctx.needs_input_grad[0] else None
        dw = dy2.t() @ x2 if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if dy.dim() == 3 and dy.is_contiguous():
                # per image first ([b, 1, n] x [b, n, N] -> b x N partial sums: b independent products fill the chip; ONE
                # product over all b*n rows is a 50 000-long reduction in N/64 workgroups: +2 ms per deit_mrlal_tiny step).
                # The partial sums stay in fp32 where the batched product can return them (no second rounding, no fp16
                # overflow of a partial under a GradScaler's scale); the sum over the images is an fp32 product either way.
                part = _bmm_f32(dy.new_ones((dy.shape[0], 1, dy.shape[1])), dy).view(dy.shape[0], -1)
                db = (part.new_ones((1, part.shape[0])) @ part).view(-1).to(dy.dtype)
            else:
                db = (dy2.new_ones((1, dy2.shape[0])) @ dy2).view(-1)
        return dx, dw, db


_BMM_OUT_F32 = None        # does torch.bmm take out_dtype=float32 for 16-bit operands on this build?  (asked once, on first use)


def _bmm_f32(a, b):
    """a @ b (batched) with the fp32 accumulator as the result where torch.bmm can hand it out, else the product in the
    operands' dtype widened afterwards."""
    global _BMM_OUT_F32
    if a.dtype == torch.float32:
        return torch.bmm(a, b)
    if _BMM_OUT_F32 is None and not torch.cuda.is_current_stream_capturing():
        try:
            torch.bmm(a[:1, :, :1], b[:1, :1, :1], out_dtype=torch.float32)
            _BMM_OUT_F32 = True
        except (TypeError, RuntimeError):
            _BMM_OUT_F32 = False
    if _BMM_OUT_F32:
        return torch.bmm(a, b, out_dtype=torch.float32)
    return torch.bmm(a, b).float()


def _linear(mod, x):
    """`mod(x)` for an nn.Linear on a CUDA tensor through _LinearFn (autocast handled here: the Function runs with autocast
    off on operands already in the autocast dtype, as torch.amp.custom_fwd(cast_inputs=...) would arrange)."""
    if not (type(mod) is nn.Linear and x.is_cuda and torch.is_grad_enabled()):
        return mod(x)
    w, b = mod.weight, mod.bias
    if torch.is_autocast_enabled("cuda"):
        dt = torch.get_autocast_dtype("cuda")
        x, w, b = x.to(dt), w.to(dt), (b.to(dt) if b is not None else None)
        with torch.autocast("cuda", enabled=False):
            return _LinearFn.apply(x, w, b)
    return _LinearFn.apply(x, w, b)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None):
        super().__init__()
        self.img_size, self.patch_size = _pair(img_size), _pair(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features, hidden_features = out_features or in_features, hidden_features or in_features
        d1, d2 = _pair(drop)
        self.fc1, self.act, self.drop1 = nn.Linear(in_features, hidden_features), act_layer(), nn.Dropout(d1)
        self.fc2, self.drop2 = nn.Linear(hidden_features, out_features), nn.Dropout(d2)

    def forward(self, x):
        return self.drop2(_linear(self.fc2, self.drop1(self.act(_linear(self.fc1, x)))))


class Attention(nn.Module):
    """Multi-head self-attention of the reference blocks (deit_mrla_light.py:66-91).  `fused_attn` (class attribute):
    evaluate softmax(q k^T * scale) v through torch's scaled_dot_product_attention (same math, attention dropout
    included; one fused kernel on the GPU instead of two batched GEMMs and a softmax)."""
    fused_attn = True

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        B, N, C = x.shape
        q, k, v = _linear(self.qkv, x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4).unbind(0)
        if self.fused_attn and x.is_cuda:
            y = F.scaled_dot_product_attention(q, k, v, dropout_p=self.attn_drop.p if self.training else 0.0,
                                               scale=self.scale)
        else:
            y = self.attn_drop(((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)) @ v
        return self.proj_drop(_linear(self.proj, y.transpose(1, 2).reshape(B, N, C)))


class Block(nn.Module):
    def __init__(self, dim, num_heads, dim_mrla, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=partial(nn.LayerNorm, eps=1e-6)):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = layers.DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.mrla = layers.mrlal_module(input_dim=dim, dim_perhead=dim_mrla, norm_layer=norm_layer)

    def forward(self, x):
        ot = x
        x = x + self.drop_path(self.attn(self.norm1(x)))
        x = x + self.drop_path(self.mlp(self.norm2(x)))
        return self.mrla(x, ot, fused_residual=True)            # x + mrla(x, ot), residual inside the HIP op


def _init_vit_weights(module, name="", head_bias=0.0):
    if isinstance(module, nn.Linear):
        if name.startswith("head"):
            nn.init.zeros_(module.weight)
            nn.init.constant_(module.bias, head_bias)
        else:
            nn.init.trunc_normal_(module.weight, std=0.02)
            if module.bias is not None:
                nn.init.zeros_(module.bias)
    elif isinstance(module, (nn.LayerNorm, nn.GroupNorm, nn.BatchNorm2d)):
        nn.init.zeros_(module.bias)
        nn.init.ones_(module.weight)


class ViT_mrlal(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 dim_mrla=16, mlp_ratio=4.0, qkv_bias=True, representation_size=None, distilled=False, drop_rate=0.0,
                 attn_drop_rate=0.0, drop_path_rate=0.0, embed_layer=PatchEmbed, norm_layer=None, act_layer=nn.GELU,
                 weight_init=""):
        super().__init__()
        if weight_init not in ("",):
            raise NotImplementedError("only the default ('') weight_init scheme is provided")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 2 if distilled else 1
        if distilled:
            raise NotImplementedError("distillation token: the MRLA token map needs n - 1 to be a perfect square")
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        act_layer = act_layer or nn.GELU
        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = None
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + self.num_tokens, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.blocks = self._make_blocks(depth, drop_path_rate, dim=embed_dim, num_heads=num_heads, dim_mrla=dim_mrla,
                                        mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate, attn_drop=attn_drop_rate,
                                        norm_layer=norm_layer, act_layer=act_layer)
        self.norm = norm_layer(embed_dim)
        if representation_size:
            self.num_features = representation_size
            self.pre_logits = nn.Sequential(OrderedDict([("fc", nn.Linear(embed_dim, representation_size)),
                                                         ("act", nn.Tanh())]))
        else:
            self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.head_dist = None
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.apply(_init_vit_weights)

    @staticmethod
    def _make_blocks(depth, drop_path_rate, **kw):
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth, device="cpu")]
        return nn.Sequential(*[Block(drop_path=dpr[i], **kw) for i in range(depth)])

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "dist_token"}

    def get_classifier(self):
        return self.head

    def reset_classifier(self, num_classes, global_pool=""):
        self.num_classes = num_classes
        self.head = nn.Linear(self.embed_dim, num_classes) if num_classes > 0 else nn.Identity()

    def forward_features(self, x):
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        x = self.norm(self.blocks(self.pos_drop(x + self.pos_embed)))
        return self.pre_logits(x[:, 0])

    def forward(self, x):
        x = self.forward_features(x)
        return _linear(self.head, x) if isinstance(self.head, nn.Linear) else self.head(x)


class Block_base(nn.Module):
    """deit/deit_mrla_base.py:246-277 (`Block` there): the K/V history restarts every `mrlab_size` blocks."""

    def __init__(self, dim, num_heads, dim_mrla, init_cell=False, layer_index=0, mrlab_size=4, mlp_ratio=4.0,
                 qkv_bias=False, drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6)):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = layers.DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.mrla = layers.mrlab_module(input_dim=dim, dim_perhead=dim_mrla, init_cell=(layer_index % mrlab_size == 0))
        self.mrla.mrla.history_hint = mrlab_size

    def forward(self, x, prev_k, prev_v):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        x = x + self.drop_path(self.mlp(self.norm2(x)))
        attn_t, k, v = self.mrla(x, prev_k, prev_v)
        return x + attn_t, k, v


class ViT_mrlab(ViT_mrlal):
    """deit/deit_mrla_base.py:280-413.  As there, every block's stochastic depth rate is 0.1 whatever
    `drop_path_rate` says (:340), the blocks are an nn.ModuleList and K/V thread through them (:398-401)."""
    mrlab_size = 4

    @classmethod
    def _make_blocks(cls, depth, drop_path_rate, **kw):
        return nn.ModuleList([Block_base(layer_index=i, mrlab_size=cls.mrlab_size, drop_path=0.1, **kw)
                              for i in range(depth)])

    def forward_features(self, x):
        x = self.patch_embed(x)
        x = self.pos_drop(torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1) + self.pos_embed)
        k = v = None
        for blk in self.blocks:
            x, k, v = blk(x, k, v)
        return self.pre_logits(self.norm(x)[:, 0])


def _deit_base_variant(embed_dim, num_heads, **kwargs):
    kwargs.pop("pretrained", None)
    model = ViT_mrlab(patch_size=16, embed_dim=embed_dim, depth=12, num_heads=num_heads, dim_mrla=16, mlp_ratio=4,
                      qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = {"input_size": (3, 224, 224), "num_classes": 1000}
    return model


@register_model
def deit_mrlab_tiny_patch16_224(pretrained=False, **kwargs):
    return _deit_base_variant(192, 3, **kwargs)


@register_model
def deit_mrlab_small_patch16_224(pretrained=False, **kwargs):
    return _deit_base_variant(384, 6, **kwargs)


@register_model
def deit_mrlab_base_patch16_224(pretrained=False, **kwargs):
    return _deit_base_variant(768, 12, **kwargs)


def _deit(embed_dim, num_heads, **kwargs):
    kwargs.pop("pretrained", None)
    model = ViT_mrlal(patch_size=16, embed_dim=embed_dim, depth=12, num_heads=num_heads, dim_mrla=16, mlp_ratio=4,
                      qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    model.default_cfg = {"input_size": (3, 224, 224), "num_classes": 1000}
    return model


@register_model
def deit_mrlal_tiny_patch16_224(pretrained=False, **kwargs):
    return _deit(192, 3, **kwargs)


@register_model
def deit_mrlal_small_patch16_224(pretrained=False, **kwargs):
    return _deit(384, 6, **kwargs)


@register_model
def deit_mrlal_base_patch16_224(pretrained=False, **kwargs):
    return _deit(768, 12, **kwargs)
