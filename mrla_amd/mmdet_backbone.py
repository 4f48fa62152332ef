"""Detection backbone: ResNet + MRLA-light with the surface of the reference's mmdetection plug-in
(mmdetection/mmdet/models/backbones/resnet_mrlal.py:116-367): same class name and constructor keywords
(`frozen_stages`, `norm_eval`, `style`, `init_cfg`, ...), same state_dict keys (a classification checkpoint minus
`fc.*` loads), `forward(x) -> (C2, C3, C4, C5)`, and the `train()` override that keeps frozen stages / BatchNorm
statistics fixed.  Images of any size are accepted (the channels_last MRLA kernels march rows of any width).

Differences that follow the reference file and not the classification network: the block tail has NO stochastic depth
(`out = out + self.bn_mrla(self.mrla(out, identity))`, :112 -- `drop_path` is accepted and ignored), and with
`norm_eval=True` every BatchNorm (bn_mrla included) is a fixed per-channel affine during training, which the HIP tail
runs in its BN_EVAL mode, gradients included.

When mmdet is importable the class registers itself in `mmdet.models.builder.BACKBONES` under the reference's name."""
import warnings

import torch
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from . import functional as F_
from . import layers
from .resnet import MRLA_Bottleneck as _ClsBottleneck
from .resnet import _ResNetMRLA

__all__ = ["ResNet_mrlal", "MRLA_Bottleneck"]


class MRLA_Bottleneck(_ClsBottleneck):
    """resnet_mrlal.py:47-113: the classification block without DropPath on the MRLA term."""

    def __init__(self, inplanes, planes, stride=1, downsample=None, SE=False, ECA_size=None, groups=1, base_width=64,
                 dilation=1, norm_layer=nn.BatchNorm2d, drop_path=0.0, init_cfg=None):
        super().__init__(inplanes, planes, stride, downsample, SE, ECA_size, groups, base_width, dilation, norm_layer,
                         drop_path=0.0)
        self.init_cfg = init_cfg


class ResNet_mrlal(_ResNetMRLA):
    channels_last = True

    def __init__(self, block=MRLA_Bottleneck, layers=(3, 4, 6, 3), SE=False, ECA=None, frozen_stages=-1, norm_eval=True,
                 style="pytorch", zero_init_last_bn=True, groups=1, width_per_group=64, replace_stride_with_dilation=None,
                 norm_layer=nn.BatchNorm2d, drop_rate=0.0, drop_path=0.0, pretrained=None, init_cfg=None):
        super().__init__()
        assert not (init_cfg and pretrained), "init_cfg and pretrained cannot be specified at the same time"
        if isinstance(pretrained, str):
            warnings.warn("DeprecationWarning: pretrained is deprecated, please use \"init_cfg\" instead")
            init_cfg = dict(type="Pretrained", checkpoint=pretrained)
        elif pretrained is not None:
            raise TypeError("pretrained must be a str or None")
        self.init_cfg = init_cfg
        self.zero_init_last_bn = zero_init_last_bn
        self.frozen_stages, self.norm_eval, self.style = frozen_stages, norm_eval, style
        self._setup(0, SE, ECA, groups, width_per_group, replace_stride_with_dilation, norm_layer, drop_rate, drop_path)
        self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = self._norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        E, d = self._ECA, self._rswd
        self.layer1 = nn.Sequential(*self._make_layer(block, 64, layers[0], SE, E[0]))
        self.layer2 = nn.Sequential(*self._make_layer(block, 128, layers[1], SE, E[1], stride=2, dilate=d[0]))
        self.layer3 = nn.Sequential(*self._make_layer(block, 256, layers[2], SE, E[2], stride=2, dilate=d[1]))
        self.layer4 = nn.Sequential(*self._make_layer(block, 512, layers[3], SE, E[3], stride=2, dilate=d[2]))
        self.init_weights()
        if self.channels_last:
            self.to(memory_format=torch.channels_last)

    def init_weights(self):
        """What mmcv's BaseModule.init_weights does with the reference's default init_cfg (:163-178): Kaiming
        (fan_out, relu, normal) for Conv2d, 1/0 for norm layers, bn3.weight = 0 when zero_init_last_bn; a 'Pretrained'
        init_cfg loads the checkpoint (state_dict or {'state_dict': ...}, optional 'module.' / 'backbone.' prefixes)."""
        cfg = self.init_cfg
        if isinstance(cfg, dict) and cfg.get("type") == "Pretrained":
            ckpt = torch.load(cfg["checkpoint"], map_location="cpu")
            sd = ckpt.get("state_dict", ckpt)
            own = self.state_dict()
            for prefix in ("module.", "backbone."):
                sd = {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in sd.items()}
            self.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
            return
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (_BatchNorm, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_last_bn and cfg is None:
            for m in self.modules():
                if isinstance(m, _ClsBottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def forward_features(self, x):
        if self.channels_last and x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        x = F_.bn_relu_maxpool(self.conv1(x), self.bn1, self.maxpool)
        outs = []
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = stage(x)
            outs.append(x)
        return tuple(outs)

    def forward(self, x):
        return self.forward_features(x)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f"layer{i}")
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
        return self


try:                                    # optional registration, exactly where the reference puts it (:115)
    from mmdet.models.builder import BACKBONES
    BACKBONES.register_module()(ResNet_mrlal)
except Exception:                       # pragma: no cover - mmdet is not installed in this image
    pass
