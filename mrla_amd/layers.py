"""Host-side mirror of the reference's MRLA operator modules (same names, constructor arguments,
parameter names and error behaviour), executing on libmrla_hip.so.

Reference counterparts (paths relative to the reference repo):
  mrla_light_layer   resnet/models/modules/mrla_light_module.py:9-74
  mrla_module        resnet/models/resnet_mrla_light.py:32-43          (light: + lambda_t * o_{t-1})
  DropPath           resnet/models/utils/drop.py:22-30
  light_block_tail   resnet/models/resnet_mrla_light.py:116            (x + DropPath(bn_mrla(mrla(x, identity))))
"""
from math import sqrt

import math

import torch
import torch.nn as nn

from . import functional as F_
from ._lib import MrlaHipError


def _heads_and_ksize(input_dim, heads, dim_perhead, k_size):
    if heads is None and dim_perhead is None:
        raise ValueError("arguments heads and dim_perhead cannot be None at the same time !")
    if dim_perhead is not None:
        heads = int(input_dim / dim_perhead)
    if k_size is None:
        k_size = F_.k_size_for(input_dim)
    return heads, k_size


def drop_path_scale(batch, drop_prob, training, device):
    """Per-sample multiplier of stochastic depth, floor(keep + U[0,1)) / keep, or None when inactive."""
    if drop_prob == 0.0 or not training:
        return None
    book = F_.current_bookkeeping()
    if book is not None and book.n_dp > 0:          # one table per forward instead of four tiny kernels per block
        return book.drop_path_row(batch, drop_prob, device)
    keep = 1.0 - drop_prob
    return torch.floor(keep + torch.rand((batch,), dtype=torch.float32, device=device)) / keep


class DropPath(nn.Module):
    """Stochastic depth per sample.  Inside the fused block tail only `drop_prob` is read (the mask is
    applied by the HIP kernel); called on its own it behaves like the reference module."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        s = drop_path_scale(x.shape[0], self.drop_prob or 0.0, self.training, x.device)
        if s is None:
            return x
        return x * s.to(x.dtype).view((-1,) + (1,) * (x.ndim - 1))


class _QKVParams(nn.Module):
    """Wq / Wk (Conv1d 1->1, k taps, no bias) and Wv (depthwise 3x3, no bias): parameter containers whose
    names and shapes equal the reference's, so checkpoints interchange.  They are never called."""

    def __init__(self, input_dim, heads, dim_perhead, k_size):
        super().__init__()
        heads, k_size = _heads_and_ksize(input_dim, heads, dim_perhead, k_size)
        if heads <= 0 or input_dim % heads:
            raise ValueError(f"input_dim ({input_dim}) must be divisible by heads ({heads})")
        self.input_dim, self.heads, self.k_size = input_dim, heads, k_size
        self.dim_perhead = input_dim // heads
        self.Wq = nn.Conv1d(1, 1, kernel_size=k_size, padding=(k_size - 1) // 2, bias=False)
        self.Wk = nn.Conv1d(1, 1, kernel_size=k_size, padding=(k_size - 1) // 2, bias=False)
        self.Wv = nn.Conv2d(input_dim, input_dim, kernel_size=3, stride=1, padding=1, groups=input_dim, bias=False)
        self._norm_fact = 1 / sqrt(input_dim / heads)

    def _check(self, x):
        if x.dim() != 4 or x.shape[1] != self.input_dim:
            raise MrlaHipError(f"expected [b, {self.input_dim}, h, w], got {tuple(x.shape)}")


class mrla_light_layer(_QKVParams):
    """MRLA-light layer: sigmoid(per-head <Wq*y, Wk*y> / sqrt(d)) * dwconv3x3(x), y = GAP(x)."""

    def __init__(self, input_dim, heads=None, dim_perhead=None, k_size=None):
        super().__init__(input_dim, heads, dim_perhead, k_size)

    def forward(self, x):
        self._check(x)
        return F_.mrla_light(x, self.Wq.weight, self.Wk.weight, self.Wv.weight, self.dim_perhead)


class mrla_module(nn.Module):
    """Light recurrence o_t = mrla_light(x_t) + lambda_t * o_{t-1} (one fused HIP pass)."""
    dim_perhead = 32

    def __init__(self, input_dim):
        super().__init__()
        self.mrla = mrla_light_layer(input_dim=input_dim, dim_perhead=self.dim_perhead)
        self.lambda_t = nn.Parameter(torch.randn(input_dim, 1, 1))

    def forward(self, xt, ot_1):
        m = self.mrla
        m._check(xt)
        return F_.mrla_light(xt, m.Wq.weight, m.Wk.weight, m.Wv.weight, m.dim_perhead, o_prev=ot_1, lam=self.lambda_t)


def _bn_args(bn):
    """BatchNorm2d module -> argument dict of the fused tail (handles momentum=None and untracked stats)."""
    training = bn.training or bn.running_mean is None
    momentum = bn.momentum
    rm, rv = bn.running_mean, bn.running_var
    if bn.training and bn.track_running_stats:
        momentum = F_.bump_batch_counter(bn)
    if momentum is None:
        momentum = 0.0
    if rm is None:       # statistics not tracked: batch statistics in both modes, nothing to update
        rm = torch.zeros_like(bn.weight, dtype=torch.float32)
        rv = torch.ones_like(bn.weight, dtype=torch.float32)
    return dict(weight=bn.weight, bias=bn.bias, running_mean=rm, running_var=rv, training=training,
                momentum=momentum, eps=bn.eps)


def light_tail_is_fused(bn_mrla):
    """True when light_block_tail runs as the two fused HIP passes (then bn3's affine may be deferred into them)."""
    return type(bn_mrla) is nn.BatchNorm2d and bn_mrla.affine


def light_block_tail(x, identity, mrla, bn_mrla, drop_path, pre_activation=False):
    """x + DropPath(bn_mrla(mrla(x, identity))) -- fused into two HIP passes when bn_mrla is a BatchNorm2d.
    pre_activation=True: `x` is the bottleneck's bn3 output and x_t = relu(x + identity)
    (resnet_mrla_light.py:113-114) is formed inside the first pass as well."""
    p = getattr(drop_path, "drop_prob", 0.0) or 0.0
    if light_tail_is_fused(bn_mrla):
        m = mrla.mrla
        m._check(x)
        dp = drop_path_scale(x.shape[0], p, drop_path.training if isinstance(drop_path, nn.Module) else False, x.device)
        return F_.mrla_light(x, m.Wq.weight, m.Wk.weight, m.Wv.weight, m.dim_perhead, o_prev=identity,
                             lam=mrla.lambda_t, bn=_bn_args(bn_mrla), dp=dp, res=True, pre_activation=pre_activation)
    # any other norm layer the caller injected: MRLA op on the GPU, then the caller's modules as they are
    if pre_activation:
        x = torch.relu(x + identity)
    return x + drop_path(bn_mrla(mrla(x, identity)))


# ======================================================================================================
# MRLA-base (reference: resnet/models/modules/mrla_base_module.py:10-89, resnet_mrla_base.py:32-51,120-129)
# ======================================================================================================
def _stage_for(layer, x, prev_K, prev_V):
    """The BaseStage this call appends to: a new one for an init_cell layer, else the one the incoming
    K/V views belong to."""
    if layer.init_cell:
        b, c, h, w = x.shape
        return F_.BaseStage(b, c, h, w, layer.dim_perhead, x.dtype, x.device, layer.history_hint or 4,
                            F_.BaseStage.layout_for(x, layer.dim_perhead))
    stage = getattr(prev_V, "_mrla_stage", None)
    if stage is None or getattr(prev_K, "_mrla_stage", None) is not stage:
        raise MrlaHipError("prev_K / prev_V must be the tensors returned by the previous MRLA-base layer of the same "
                           "stage (the K/V history lives in a device ring, not in free-standing tensors)")
    return stage


class mrla_base_layer(_QKVParams):
    """MRLA-base layer: softmax over the stage's depth.  forward(x, prev_K, prev_V) -> (out, K[b,t,c], V[b,t,c,h,w]);
    K and V are views of the stage's ring buffers."""

    def __init__(self, input_dim, heads=None, dim_perhead=None, k_size=None, init_cell=False):
        super().__init__(input_dim, heads, dim_perhead, k_size)
        self.init_cell = init_cell
        self.history_hint = None      # stage depth, set by the network constructor (rings grow if it is too small)

    def forward(self, x, prev_K, prev_V):
        self._check(x)
        stage = _stage_for(self, x, prev_K, prev_V)
        out = F_.mrla_base(x, self.Wq.weight, self.Wk.weight, self.Wv.weight, self.dim_perhead, stage)
        K, V = stage.views()
        return out, K, V


class mrla_base_module(nn.Module):
    """resnet_mrla_base.py:32-51 (`mrla_module` there)."""
    dim_perhead = 16

    def __init__(self, input_dim, init_cell=False, channel_wise=False):
        super().__init__()
        if channel_wise:
            self.dim_perhead = 1
        self.mrla = mrla_base_layer(input_dim=input_dim, dim_perhead=self.dim_perhead, init_cell=init_cell)
        self.init_cell = init_cell

    def forward(self, xt, prev_k, prev_v):
        if self.init_cell:
            prev_k = prev_v = None
        return self.mrla(xt, prev_k, prev_v)


def base_tail_defers_bn3(mrla, bn_mrla, prev_v):
    """Predicate on conv3's output: True when base_block_tail will run its fused path on an NHWC stage, so that bn3's
    elementwise pass can be folded into the MRLA pooling pass."""
    def decide(conv_out):
        if not (type(bn_mrla) is nn.BatchNorm2d and bn_mrla.affine):
            return False
        layer = mrla.mrla
        if layer.init_cell:
            return F_.BaseStage.layout_for(conv_out, layer.dim_perhead) == F_.L.NHWC
        stage = getattr(prev_v, "_mrla_stage", None)
        return stage is not None and stage.layout == F_.L.NHWC
    return decide


def base_block_tail(x, prev_k, prev_v, mrla, bn_mrla, drop_path, identity=None):
    """x + DropPath(relu(bn_mrla(attn))) with attn, K, V from the MRLA-base layer (resnet_mrla_base.py:124-127).
    identity given: `x` is the bottleneck's bn3 output and x_t = relu(x + identity) (:120-121) is formed in-kernel."""
    layer = mrla.mrla
    if type(bn_mrla) is nn.BatchNorm2d and bn_mrla.affine:
        layer._check(x)
        stage = _stage_for(layer, x, prev_k, prev_v)
        p = getattr(drop_path, "drop_prob", 0.0) or 0.0
        dp = drop_path_scale(x.shape[0], p, drop_path.training if isinstance(drop_path, nn.Module) else False, x.device)
        out = F_.mrla_base(x, layer.Wq.weight, layer.Wk.weight, layer.Wv.weight, layer.dim_perhead, stage,
                           bn=_bn_args(bn_mrla), dp=dp, identity=identity)
        K, V = stage.views()
        return out, K, V
    if identity is not None:
        x = torch.relu(x + identity)
    attn, K, V = mrla(x, prev_k, prev_v)
    return x + drop_path(torch.relu(bn_mrla(attn))), K, V


# ======================================================================================================
# DeiT token variant (reference: deit/deit_mrla_light.py:117-209)
# ======================================================================================================
class mrlal_layer(_QKVParams):
    """MRLA-light layer with GELU on V, on a [b, c, h, w] map (deit_mrla_light.py:117-180)."""

    def __init__(self, input_dim, heads=None, dim_perhead=None, k_size=None):
        super().__init__(input_dim, heads, dim_perhead, k_size)

    def forward(self, x):
        self._check(x)
        return F_.mrla_light(x, self.Wq.weight, self.Wk.weight, self.Wv.weight, self.dim_perhead, act_gelu=True)


class mrlal_module(nn.Module):
    """Token module: LayerNorm of x_t and o_{t-1}, MRLA-light (GELU) on the map tokens, + lambda_t * LN(o_{t-1}),
    cls token passed through (deit_mrla_light.py:183-209).  `fused_residual=True` folds the block's x + (.) in."""

    def __init__(self, input_dim, dim_perhead, norm_layer=None):
        super().__init__()
        self.dim_perhead = dim_perhead
        self.mrla = mrlal_layer(input_dim=input_dim, dim_perhead=dim_perhead)
        self.lambda_t = nn.Parameter(torch.randn(input_dim))
        norm_layer = norm_layer or (lambda c: nn.LayerNorm(c, eps=1e-6))
        self.normx = norm_layer(input_dim)
        self.normo = norm_layer(input_dim)

    def forward(self, xt, ot_1, fused_residual=False):
        m, nx, no = self.mrla, self.normx, self.normo
        if not (type(nx) is nn.LayerNorm and type(no) is nn.LayerNorm and nx.elementwise_affine and nx.bias is not None
                and no.elementwise_affine and no.bias is not None and nx.eps == no.eps):
            raise MrlaHipError("mrlal_module: the HIP path implements affine nn.LayerNorm for normx / normo")
        return F_.mrla_token_light(xt, ot_1, nx.weight, nx.bias, no.weight, no.bias, m.Wq.weight, m.Wk.weight,
                                   m.Wv.weight, self.lambda_t, m.dim_perhead, eps=nx.eps, res=fused_residual)


class mrlab_layer(mrla_base_layer):
    """MRLA-base layer of the DeiT variant (deit/deit_mrla_base.py:120-201): identical math to the ResNet one."""


class mrlab_module(nn.Module):
    """Token module of DeiT + MRLA-base (deit/deit_mrla_base.py:204-243): LayerNorm(x_t), the 14x14 map tokens
    through the MRLA-base layer (softmax over the stage history on HIP), cls token passed through."""

    def __init__(self, input_dim, dim_perhead, init_cell=False, channel_wise=False, norm_layer=None):
        super().__init__()
        self.dim_perhead = 1 if channel_wise else dim_perhead
        self.init_cell = init_cell
        norm_layer = norm_layer or (lambda c: nn.LayerNorm(c, eps=1e-6))
        self.normx = norm_layer(input_dim)
        self.mrla = mrlab_layer(input_dim=input_dim, dim_perhead=self.dim_perhead, init_cell=init_cell)

    def _token_stage(self, xt, side, prev_k, prev_v):
        layer = self.mrla
        if layer.init_cell:
            b, _, c = xt.shape
            return F_.BaseStage(b, c, side, side, layer.dim_perhead, xt.dtype, xt.device, layer.history_hint or 4,
                                F_.L.NHWC)
        return _stage_for(layer, xt, prev_k, prev_v)

    def forward(self, xt, prev_k, prev_v):
        if self.init_cell:
            prev_k = prev_v = None
        b, n, c = xt.shape
        side = math.isqrt(n - 1)
        if side * side != n - 1:
            raise MrlaHipError(f"mrlab_module: {n - 1} map tokens do not form a square map")
        nx, layer = self.normx, self.mrla
        if (type(nx) is nn.LayerNorm and nx.elementwise_affine and nx.bias is not None and c == layer.input_dim
                and F_.token_base_supported(xt, layer.dim_perhead)):
            # LayerNorm on load, V_t straight into the stage's ring, the map rows and the cls row of the result written in
            # place: no normalised copy of xt, no token <-> map view copies, no cat (deit_mrla_base.py:224-243)
            stage = self._token_stage(xt, side, prev_k, prev_v)
            if stage.layout == F_.L.NHWC:
                out = F_.mrla_token_base(xt, nx.weight, nx.bias, layer.Wq.weight, layer.Wk.weight, layer.Wv.weight,
                                         layer.dim_perhead, stage, eps=nx.eps)
                K, V = stage.views()
                return out, K, V
        xt = self.normx(xt)
        # the map tokens ARE a channels_last image (pixel pitch c): no token <-> NCHW transposes; the stage's history
        # lives in slot-major NHWC rings and the layer's output comes back as a view of map tokens
        fmap = xt[:, 1:].reshape(b, side, side, c).permute(0, 3, 1, 2)
        out, kt, vt = self.mrla(fmap, prev_k, prev_v)
        tokens = out.permute(0, 2, 3, 1).reshape(b, n - 1, c)
        return torch.cat((xt[:, :1], tokens), dim=1), kt, vt
