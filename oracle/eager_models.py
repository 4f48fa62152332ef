"""TEST INFRASTRUCTURE ONLY -- eager PyTorch restatement of the reference models.

Stock ATen ops only (runs on CPU, or on a GPU through MIOpen/rocBLAS); the math per block is the
one the reference executes, so this file serves as
  * the on-box oracle for full-model parity (`/root/reference` does not travel to the GPU box),
  * the "eager PyTorch-ROCm" baseline the north-star's >=4x target is quoted against,
  * the CPU baseline of `bench.py` (`cpu_baseline.kind == "port"`).
It is pinned against logits produced by the reference itself (tests/golden/model_*.npz).

Parameter names equal the reference's `state_dict` keys, so one checkpoint loads into the
reference, into this restatement and into the product modules in `mrla_amd`.

Reference lines restated (relative to /root/reference):
  resnet/models/resnet_mrla_light.py:32-43,47-118,122-250
  resnet/models/resnet_mrla_base.py:32-51,55-131,134-283
  resnet/models/modules/mrla_light_module.py:28-74, mrla_base_module.py:29-89
  resnet/models/utils/drop.py:7-24
  deit/deit_mrla_light.py:42-63,66-115,117-235,238-471
"""
from functools import partial
from math import log, sqrt

import torch
import torch.nn as nn
import torch.nn.functional as F


def k_size_for(c):
    t = int(abs((log(c, 2) + 1) / 2.0))
    return t if t % 2 else t + 1


def stochastic_depth(t, p, training, mask=None):
    """drop.py:7-19.  `mask` ([b] of 0/1) overrides the RNG draw so tests can share it."""
    if p == 0.0 or not training:
        return t
    keep = 1.0 - p
    shape = (t.shape[0],) + (1,) * (t.ndim - 1)
    if mask is None:
        mask = torch.floor(keep + torch.rand(shape, dtype=t.dtype, device=t.device))
    return t.div(keep) * mask.reshape(shape).to(t.dtype)


class _QKV(nn.Module):
    """Holds Wq / Wk (Conv1d 1->1, k taps) and Wv (depthwise 3x3) under the reference's names."""

    def __init__(self, c, d):
        super().__init__()
        assert c % d == 0
        k = k_size_for(c)
        self.c, self.d, self.g = c, d, c // d
        self.Wq = nn.Conv1d(1, 1, k, padding=(k - 1) // 2, bias=False)
        self.Wk = nn.Conv1d(1, 1, k, padding=(k - 1) // 2, bias=False)
        self.Wv = nn.Conv2d(c, c, 3, padding=1, groups=c, bias=False)
        self.scale = 1.0 / sqrt(d)

    def project(self, x):
        y = x.mean(dim=(2, 3)).unsqueeze(1)            # [b,1,c]
        return self.Wq(y).squeeze(1), self.Wk(y).squeeze(1), self.Wv(x)


class EagerLightLayer(_QKV):
    """mrla_light_module.py:52-74 (act=None) / deit_mrla_light.py:157-180 (act=GELU)."""

    def __init__(self, c, d, act=None):
        super().__init__(c, d)
        self.act = act

    def forward(self, x):
        b = x.shape[0]
        q, k, v = self.project(x)
        if self.act is not None:
            v = self.act(v)
        a = torch.sigmoid((q * k).view(b, self.g, self.d).sum(-1) * self.scale)
        return v * a.repeat_interleave(self.d, dim=1)[:, :, None, None]


class EagerBaseLayer(_QKV):
    """mrla_base_module.py:54-89."""

    def __init__(self, c, d, init_cell=False):
        super().__init__(c, d)
        self.init_cell = init_cell

    def forward(self, x, K_prev, V_prev):
        b, c, h, w = x.shape
        q, k, v = self.project(x)
        if self.init_cell:
            K, V = k.unsqueeze(1), v.unsqueeze(1)
        else:
            K = torch.cat([K_prev, k.unsqueeze(1)], 1)
            V = torch.cat([V_prev, v.unsqueeze(1)], 1)
        t = K.shape[1]
        logits = torch.einsum("bgd,btgd->bgt", q.view(b, self.g, self.d), K.view(b, t, self.g, self.d))
        P = torch.softmax(logits * self.scale, dim=-1)
        out = torch.einsum("bgt,btgn->bgn", P, V.reshape(b, t, self.g, self.d * h * w))
        return out.reshape(b, c, h, w), K, V


class EagerLightModule(nn.Module):
    """resnet_mrla_light.py:32-43."""
    dim_perhead = 32

    def __init__(self, c):
        super().__init__()
        self.mrla = EagerLightLayer(c, self.dim_perhead)
        self.lambda_t = nn.Parameter(torch.randn(c, 1, 1))

    def forward(self, xt, o_prev):
        return self.mrla(xt) + self.lambda_t * o_prev


class EagerBaseModule(nn.Module):
    """resnet_mrla_base.py:32-51."""

    def __init__(self, c, init_cell=False, channel_wise=False):
        super().__init__()
        self.init_cell = init_cell
        self.mrla = EagerBaseLayer(c, 1 if channel_wise else 16, init_cell)

    def forward(self, xt, K_prev, V_prev):
        if self.init_cell:
            K_prev = V_prev = None
        return self.mrla(xt, K_prev, V_prev)


def _bottleneck_trunk(blk, inplanes, planes, stride, downsample, groups, base_width, dilation, norm):
    width = int(planes * (base_width / 64.0)) * groups
    blk.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
    blk.bn1 = norm(width)
    blk.conv2 = nn.Conv2d(width, width, 3, stride, dilation, dilation, groups, bias=False)
    blk.bn2 = norm(width)
    blk.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
    blk.bn3 = norm(planes * 4)
    blk.downsample = downsample
    blk.se = None
    blk.eca = None


def _trunk_forward(blk, x):
    """conv-bn-relu x2, conv-bn, shortcut add, relu.  Returns (x_t, identity)."""
    identity = x
    out = F.relu(blk.bn1(blk.conv1(x)))
    out = F.relu(blk.bn2(blk.conv2(out)))
    out = blk.bn3(blk.conv3(out))
    if blk.downsample is not None:
        identity = blk.downsample(x)
    return F.relu(out + identity), identity


class EagerLightBottleneck(nn.Module):
    """resnet_mrla_light.py:47-118."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64,
                 dilation=1, norm_layer=nn.BatchNorm2d, drop_path=0.0):
        super().__init__()
        _bottleneck_trunk(self, inplanes, planes, stride, downsample, groups, base_width, dilation, norm_layer)
        self.mrla = EagerLightModule(planes * 4)
        self.bn_mrla = norm_layer(planes * 4)
        self.p_drop = drop_path
        self.dp_mask = None     # tests may pin the per-sample keep mask

    def forward(self, x):
        xt, identity = _trunk_forward(self, x)
        z = self.bn_mrla(self.mrla(xt, identity))
        return xt + stochastic_depth(z, self.p_drop, self.training, self.dp_mask)


class EagerLightBasicBlock(nn.Module):
    """Build-side extension (mrla_amd.resnet.MRLA_BasicBlock): torchvision's BasicBlock + the light tail of
    resnet_mrla_light.py:113-116.  There is no reference counterpart; this is the restatement the product is compared with."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=nn.BatchNorm2d, drop_path=0.0):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.mrla = EagerLightModule(planes)
        self.bn_mrla = norm_layer(planes)
        self.p_drop = drop_path
        self.dp_mask = None

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.bn2(self.conv2(F.relu(self.bn1(self.conv1(x)))))
        xt = F.relu(out + identity)
        z = self.bn_mrla(self.mrla(xt, identity))
        return xt + stochastic_depth(z, self.p_drop, self.training, self.dp_mask)


class EagerBaseBottleneck(nn.Module):
    """resnet_mrla_base.py:55-131."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64,
                 dilation=1, norm_layer=nn.BatchNorm2d, drop_path=0.0, init_cell=False,
                 channel_wise_mrla=False):
        super().__init__()
        _bottleneck_trunk(self, inplanes, planes, stride, downsample, groups, base_width, dilation, norm_layer)
        self.mrla = EagerBaseModule(planes * 4, init_cell, channel_wise_mrla)
        self.bn_mrla = norm_layer(planes * 4)
        self.p_drop = drop_path
        self.dp_mask = None

    def forward(self, x, K_prev, V_prev):
        xt, _ = _trunk_forward(self, x)
        attn, K, V = self.mrla(xt, K_prev, V_prev)
        z = F.relu(self.bn_mrla(attn))
        return xt + stochastic_depth(z, self.p_drop, self.training, self.dp_mask), K, V


class _EagerResNet(nn.Module):
    def _init_common(self, num_classes, zero_init_last_bn, groups, width_per_group,
                     replace_stride_with_dilation, norm_layer, drop_rate, drop_path):
        self._norm = norm_layer or nn.BatchNorm2d
        self.num_classes, self.drop_rate, self.drop_path = num_classes, drop_rate, drop_path
        self.groups, self.base_width = groups, width_per_group
        self.inplanes, self.dilation = 64, 1
        self._rswd = replace_stride_with_dilation or [False, False, False]
        if len(self._rswd) != 3:
            raise ValueError("replace_stride_with_dilation should be None or a 3-element tuple")

    def _stage(self, block, planes, n, stride=1, dilate=False, **extra_first):
        prev_dil = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        down, ex = None, getattr(block, "expansion", 4)
        if stride != 1 or self.inplanes != planes * ex:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * ex, 1, stride, bias=False),
                                 self._norm(planes * ex))
        common = dict(groups=self.groups, base_width=self.base_width, norm_layer=self._norm,
                      drop_path=self.drop_path)
        blocks = [block(self.inplanes, planes, stride, down, dilation=prev_dil, **common, **extra_first)]
        self.inplanes = planes * ex
        rest = {k: (False if k == "init_cell" else v) for k, v in extra_first.items()}
        blocks += [block(self.inplanes, planes, dilation=self.dilation, **common, **rest) for _ in range(1, n)]
        return blocks

    def _finish_init(self, zero_init_last_bn):
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(self.inplanes, self.num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        if zero_init_last_bn:
            for m in self.modules():
                if isinstance(m, (EagerLightBottleneck, EagerBaseBottleneck)):
                    nn.init.zeros_(m.bn3.weight)
                elif isinstance(m, EagerLightBasicBlock):
                    nn.init.zeros_(m.bn2.weight)

    def forward(self, x):
        x = torch.flatten(self.avgpool(self.forward_features(x)), 1)
        if self.drop_rate:
            x = F.dropout(x, p=float(self.drop_rate), training=self.training)
        return self.fc(x)


class EagerResNetLight(_EagerResNet):
    """resnet_mrla_light.py:122-238."""

    def __init__(self, layers, num_classes=1000, SE=False, ECA=None, zero_init_last_bn=True, groups=1,
                 width_per_group=64, replace_stride_with_dilation=None, norm_layer=nn.BatchNorm2d,
                 drop_rate=0.0, drop_path=0.0, block=None):
        super().__init__()
        assert not SE and ECA is None, "SE/ECA are out of scope for the oracle"
        self._init_common(num_classes, zero_init_last_bn, groups, width_per_group,
                          replace_stride_with_dilation, norm_layer, drop_rate, drop_path)
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = self._norm(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        B = block or EagerLightBottleneck
        self.layer1 = nn.Sequential(*self._stage(B, 64, layers[0]))
        self.layer2 = nn.Sequential(*self._stage(B, 128, layers[1], 2, self._rswd[0]))
        self.layer3 = nn.Sequential(*self._stage(B, 256, layers[2], 2, self._rswd[1]))
        self.layer4 = nn.Sequential(*self._stage(B, 512, layers[3], 2, self._rswd[2]))
        self._finish_init(zero_init_last_bn)

    def forward_features(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


class EagerResNetBase(_EagerResNet):
    """resnet_mrla_base.py:134-272 (deep 3-conv stem, K/V threaded through ModuleLists)."""

    def __init__(self, layers, num_classes=1000, SE=False, ECA=None, zero_init_last_bn=True, groups=1,
                 width_per_group=64, replace_stride_with_dilation=None, norm_layer=nn.BatchNorm2d,
                 drop_rate=0.0, drop_path=0.0, channel_wise_mrla=False):
        super().__init__()
        assert not SE and ECA is None, "SE/ECA are out of scope for the oracle"
        self._init_common(num_classes, zero_init_last_bn, groups, width_per_group,
                          replace_stride_with_dilation, norm_layer, drop_rate, drop_path)
        sw = 32
        self.conv1 = nn.Sequential(
            nn.Conv2d(3, sw, 3, 2, 1, bias=False), self._norm(sw), nn.ReLU(inplace=True),
            nn.Conv2d(sw, sw, 3, 1, 1, bias=False), self._norm(sw), nn.ReLU(inplace=True),
            nn.Conv2d(sw, 64, 3, 1, 1, bias=False))
        self.bn1 = self._norm(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        B = EagerBaseBottleneck
        ex = dict(init_cell=True, channel_wise_mrla=channel_wise_mrla)
        self.stages = nn.ModuleList([
            nn.ModuleList(self._stage(B, 64, layers[0], **ex)),
            nn.ModuleList(self._stage(B, 128, layers[1], 2, self._rswd[0], **ex)),
            nn.ModuleList(self._stage(B, 256, layers[2], 2, self._rswd[1], **ex)),
            nn.ModuleList(self._stage(B, 512, layers[3], 2, self._rswd[2], **ex))])
        self._finish_init(zero_init_last_bn)

    def forward_features(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        K = V = None
        for stage in self.stages:
            for blk in stage:
                x, K, V = blk(x, K, V)
        return x


class EagerDetBackbone(EagerResNetLight):
    """mmdetection/mmdet/models/backbones/resnet_mrlal.py:116-367: the light network without head, four feature maps
    out, no stochastic depth on the MRLA term (:112), frozen stages / norm_eval handled by train() (:333-367)."""

    def __init__(self, layers=(3, 4, 6, 3), frozen_stages=-1, norm_eval=True, zero_init_last_bn=True, **kw):
        kw.pop("drop_path", None)
        super().__init__(list(layers), num_classes=1, zero_init_last_bn=zero_init_last_bn, drop_path=0.0, **kw)
        del self.avgpool, self.fc
        self.frozen_stages, self.norm_eval = frozen_stages, norm_eval

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        outs = []
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = stage(x)
            outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p_ in m.parameters():
                    p_.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f"layer{i}")
            m.eval()
            for p_ in m.parameters():
                p_.requires_grad = False
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self


def eager_resnet18_mrlal(**kw):
    """Build-side extension (no reference counterpart): see EagerLightBasicBlock."""
    return EagerResNetLight([2, 2, 2, 2], block=EagerLightBasicBlock, **kw)


def eager_resnet50_mrlal(**kw):
    return EagerResNetLight([3, 4, 6, 3], **kw)


def eager_resnet101_mrlal(**kw):
    return EagerResNetLight([3, 4, 23, 3], **kw)


def eager_resnet50_mrlab(**kw):
    return EagerResNetBase([3, 4, 6, 3], **kw)


def eager_resnet101_mrlab(**kw):
    return EagerResNetBase([3, 4, 23, 3], **kw)


# ------------------------------------------------------------------------------------------------
# DeiT + MRLA-light (token layout)
# ------------------------------------------------------------------------------------------------
class EagerTokenLightModule(nn.Module):
    """deit_mrla_light.py:183-209."""

    def __init__(self, c, d):
        super().__init__()
        self.mrla = EagerLightLayer(c, d, act=nn.GELU())
        self.lambda_t = nn.Parameter(torch.randn(c))
        self.normx = nn.LayerNorm(c, eps=1e-6)
        self.normo = nn.LayerNorm(c, eps=1e-6)

    def forward(self, xt, o_prev):
        xn, on = self.normx(xt), self.normo(o_prev)
        b, n, c = xn.shape
        side = int(sqrt(n - 1))
        fmap = xn[:, 1:].reshape(b, side, side, c).permute(0, 3, 1, 2)
        tok = self.mrla(fmap).flatten(2).transpose(1, 2) + self.lambda_t * on[:, 1:]
        return torch.cat([xn[:, :1], tok], dim=1)


class _EagerAttention(nn.Module):
    def __init__(self, dim, heads, qkv_bias):
        super().__init__()
        self.h = heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        q, k, v = self.qkv(x).reshape(B, N, 3, self.h, C // self.h).permute(2, 0, 3, 1, 4).unbind(0)
        att = ((q @ k.transpose(-2, -1)) * (C // self.h) ** -0.5).softmax(-1)
        return self.proj((att @ v).transpose(1, 2).reshape(B, N, C))


class _EagerMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.act, self.fc2 = nn.Linear(dim, hidden), nn.GELU(), nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class EagerViTBlock(nn.Module):
    """deit_mrla_light.py:212-235 (drop_path on attn/mlp only; none on the MRLA term)."""

    def __init__(self, dim, heads, dim_mrla, mlp_ratio=4.0, qkv_bias=True, drop_path=0.0):
        super().__init__()
        ln = partial(nn.LayerNorm, eps=1e-6)
        self.norm1, self.norm2 = ln(dim), ln(dim)
        self.attn = _EagerAttention(dim, heads, qkv_bias)
        self.mlp = _EagerMlp(dim, int(dim * mlp_ratio))
        self.mrla = EagerTokenLightModule(dim, dim_mrla)
        self.p_drop = drop_path

    def forward(self, x):
        o_prev = x
        x = x + stochastic_depth(self.attn(self.norm1(x)), self.p_drop, self.training)
        x = x + stochastic_depth(self.mlp(self.norm2(x)), self.p_drop, self.training)
        return x + self.mrla(x, o_prev)


class _EagerPatchEmbed(nn.Module):
    def __init__(self, img, patch, cin, dim):
        super().__init__()
        self.num_patches = (img // patch) ** 2
        self.proj = nn.Conv2d(cin, dim, patch, patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class EagerViTLight(nn.Module):
    """deit_mrla_light.py:238-384 without the distillation / representation options."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, dim_mrla=16, mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.0):
        super().__init__()
        self.patch_embed = _EagerPatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.Sequential(*[EagerViTBlock(embed_dim, num_heads, dim_mrla, mlp_ratio, qkv_bias, dpr[i])
                                      for i in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        return self.head(self.norm(self.blocks(x))[:, 0])


def eager_deit_mrlal_tiny_patch16_224(**kw):
    return EagerViTLight(embed_dim=192, depth=12, num_heads=3, dim_mrla=16, **kw)


# ------------------------------------------------------------------------------------------------
# DeiT + MRLA-base (token layout; history reset every 4 blocks)  -- deit/deit_mrla_base.py:120-277,280-450
# ------------------------------------------------------------------------------------------------
class EagerTokenBaseModule(nn.Module):
    """deit_mrla_base.py:204-243."""

    def __init__(self, c, d, init_cell=False):
        super().__init__()
        self.init_cell = init_cell
        self.normx = nn.LayerNorm(c, eps=1e-6)
        self.mrla = EagerBaseLayer(c, d, init_cell)

    def forward(self, xt, K_prev, V_prev):
        xn = self.normx(xt)
        if self.init_cell:
            K_prev = V_prev = None
        b, n, c = xn.shape
        side = int(sqrt(n - 1))
        fmap = xn[:, 1:].reshape(b, side, side, c).permute(0, 3, 1, 2)
        out, K, V = self.mrla(fmap, K_prev, V_prev)
        return torch.cat([xn[:, :1], out.flatten(2).transpose(1, 2)], dim=1), K, V


class EagerViTBaseBlock(nn.Module):
    """deit_mrla_base.py:246-277."""

    def __init__(self, dim, heads, dim_mrla, layer_index, mrlab_size=4, mlp_ratio=4.0, qkv_bias=True, drop_path=0.0):
        super().__init__()
        ln = partial(nn.LayerNorm, eps=1e-6)
        self.norm1, self.norm2 = ln(dim), ln(dim)
        self.attn = _EagerAttention(dim, heads, qkv_bias)
        self.mlp = _EagerMlp(dim, int(dim * mlp_ratio))
        self.mrla = EagerTokenBaseModule(dim, dim_mrla, init_cell=(layer_index % mrlab_size == 0))
        self.p_drop = drop_path

    def forward(self, x, K, V):
        x = x + stochastic_depth(self.attn(self.norm1(x)), self.p_drop, self.training)
        x = x + stochastic_depth(self.mlp(self.norm2(x)), self.p_drop, self.training)
        a, K, V = self.mrla(x, K, V)
        return x + a, K, V


class EagerViTBase(nn.Module):
    """deit_mrla_base.py:280-450 (drop-path is hard-coded to 0.1 for every block there, :340)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 dim_mrla=16, mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.0):
        super().__init__()
        del drop_path_rate          # ignored by the reference too (:340)
        self.patch_embed = _EagerPatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.blocks = nn.ModuleList([EagerViTBaseBlock(embed_dim, num_heads, dim_mrla, i, 4, mlp_ratio, qkv_bias, 0.1)
                                     for i in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes)

    def forward(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        K = V = None
        for blk in self.blocks:
            x, K, V = blk(x, K, V)
        return self.head(self.norm(x)[:, 0])


def eager_deit_mrlab_tiny_patch16_224(**kw):
    return EagerViTBase(embed_dim=192, depth=12, num_heads=3, dim_mrla=16, **kw)
