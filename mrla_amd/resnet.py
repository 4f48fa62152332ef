"""ResNet + MRLA model classes with the reference's API surface (constructor keywords, attribute and
state_dict names, factory names), whose MRLA block tails run on libmrla_hip.so.  The backbone
convolutions / BatchNorms stay stock PyTorch (MIOpen), as in the reference.

Reference counterparts: resnet/models/resnet_mrla_light.py:47-250, resnet/models/resnet_mrla_base.py:55-283.
SE / ECA options (off in every BASELINE config) keep their constructor arguments and
state_dict keys and run as plain eager PyTorch modules on bn3's output (outside the accelerated path).
"""

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as F_
from . import layers


def _conv3x3(cin, cout, stride=1, groups=1, dilation=1):
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=dilation, groups=groups, bias=False, dilation=dilation)


def _conv1x1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 1, stride=stride, bias=False)


class se_layer(nn.Module):
    """Squeeze-and-excitation gate (resnet/models/modules/se_module.py:9-24), eager PyTorch: not on the hot path."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(channel // reduction, channel, bias=False), nn.Sigmoid())

    def forward(self, x):
        b, c = x.shape[:2]
        return x * self.fc(self.avg_pool(x).view(b, c)).view(b, c, 1, 1)


class eca_layer(nn.Module):
    """ECA gate (resnet/models/modules/eca_module.py:8-35), eager PyTorch: not on the hot path."""

    def __init__(self, channel, k_size=None):
        super().__init__()
        if k_size is None:
            k_size = F_.k_size_for(channel)
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv = nn.Conv1d(1, 1, kernel_size=k_size, padding=(k_size - 1) // 2, bias=False)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        y = self.conv(self.avg_pool(x).squeeze(-1).transpose(-1, -2)).transpose(-1, -2).unsqueeze(-1)
        return x * self.sigmoid(y)


class _BottleneckTrunk(nn.Module):
    """conv1x1-bn-relu, conv3x3-bn-relu, conv1x1-bn, shortcut add, relu: everything in front of the MRLA tail."""
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample, SE, ECA_size, groups, base_width, dilation, norm_layer):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = _conv1x1(inplanes, width)
        self.bn1 = norm_layer(width)
        self.conv2 = _conv3x3(width, width, stride, groups, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = _conv1x1(width, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.se = se_layer(planes * self.expansion, reduction=16) if SE else None
        self.eca = eca_layer(planes * self.expansion, int(ECA_size)) if ECA_size is not None else None
        self._norm = norm_layer

    def trunk_pre(self, x, defer_bn3=False):
        """Everything up to, but not including, the shortcut add + ReLU: (bn3 output, identity).
        defer_bn3 (bool, or a predicate on conv3's output): only bn3's statistics are taken here; its affine is applied
        by the consumer's first pass."""
        # 1x1 convolutions: HIP MFMA GEMM with the BatchNorm statistics in its epilogue when eligible (bf16, channels_last).
        # The block input's second consumer -- the shortcut itself, or the downsample branch -- takes it through conv1's
        # autograd node, so that its gradient is added in the epilogue of conv1's input-gradient GEMM instead of by
        # autograd's accumulation pass.
        out, identity = F_.conv_bn_act(x, self.conv1, self.bn1, relu=True, passthrough=True)
        out = F_.bn_act(self.conv2(out), self.bn2, relu=True)
        if self.se is not None or self.eca is not None:     # channel attention reads bn3's output: nothing to defer
            defer_bn3 = False
        if callable(defer_bn3):                              # the predicate looks at conv3's output shape / layout only
            b_, _, h_, w_ = out.shape
            cl = out.is_contiguous(memory_format=torch.channels_last) and not out.is_contiguous()
            defer_bn3 = defer_bn3(torch.empty((b_, self.conv3.out_channels, h_, w_), dtype=out.dtype, device="meta",
                                              memory_format=torch.channels_last if cl else torch.contiguous_format))
        out = F_.conv_bn_act(out, self.conv3, self.bn3, relu=False, defer=defer_bn3)
        if self.se is not None:
            out = self.se(out)
        if self.eca is not None:
            out = self.eca(out)
        if self.downsample is not None:
            ds = self.downsample
            if isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[0], nn.Conv2d):
                identity = F_.conv_bn_act(identity, ds[0], ds[1], relu=False)
            else:
                identity = ds(identity)
        return out, identity

    def trunk(self, x):
        out, identity = self.trunk_pre(x)
        out += identity
        return self.relu(out), identity


class MRLA_Bottleneck(_BottleneckTrunk):
    """Bottleneck + MRLA-light tail (resnet_mrla_light.py:47-118)."""

    def __init__(self, inplanes, planes, stride=1, downsample=None, SE=False, ECA_size=None, groups=1, base_width=64,
                 dilation=1, norm_layer=nn.BatchNorm2d, drop_path=0.0):
        super().__init__(inplanes, planes, stride, downsample, SE, ECA_size, groups, base_width, dilation, norm_layer)
        self.mrla = layers.mrla_module(input_dim=planes * self.expansion)
        self.bn_mrla = self._norm(planes * self.expansion)
        self.drop_path = layers.DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def forward(self, x):
        # bn3's affine, the shortcut add and the ReLU all run inside the first MRLA pass
        pre, identity = self.trunk_pre(x, defer_bn3=layers.light_tail_is_fused(self.bn_mrla))
        return layers.light_block_tail(pre, identity, self.mrla, self.bn_mrla, self.drop_path, pre_activation=True)


class MRLA_BasicBlock(nn.Module):
    """BUILD-SIDE EXTENSION -- the reference defines no BasicBlock network (BASELINE.json's
    config 1 names a `resnet18_mrlal` that does not exist upstream; `grep resnet18` is empty, the factories are
    resnet_mrla_light.py:242-250).  torchvision's BasicBlock (two 3x3 convolutions, expansion 1) with the reference's own
    light `mrla_module(planes)` tail: out <- relu(bn2(conv2(.)) + identity); out <- out + DropPath(bn_mrla(mrla(out,
    identity))) -- resnet_mrla_light.py:113-116 word for word, c = 64 / 128 / 256 / 512 (k = 3 / 5 / 5 / 5, 2 / 4 / 8 / 16
    heads of 32 channels).  Parity is pinned at MODULE level (rows a1 - a3 at those channel counts, tests/test_light_gpu.py)
    and, for the whole network, against the eager restatement of the same definition (oracle/eager_models.py); there is no
    reference output to compare with, and the tests say so."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, SE=False, ECA_size=None, groups=1, base_width=64,
                 dilation=1, norm_layer=nn.BatchNorm2d, drop_path=0.0):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride
        self.se = se_layer(planes, reduction=16) if SE else None
        self.eca = eca_layer(planes, int(ECA_size)) if ECA_size is not None else None
        self.mrla = layers.mrla_module(input_dim=planes)
        self.bn_mrla = norm_layer(planes)
        self.drop_path = layers.DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def forward(self, x):
        identity = x
        out = F_.bn_act(self.conv1(x), self.bn1, relu=True)
        # bn2's affine, the shortcut add and the ReLU all run inside the first MRLA pass (as bn3's do in the bottleneck)
        defer = layers.light_tail_is_fused(self.bn_mrla) and self.se is None and self.eca is None
        out = F_.bn_act(self.conv2(out), self.bn2, relu=False, defer=defer)
        if self.se is not None:
            out = self.se(out)
        if self.eca is not None:
            out = self.eca(out)
        if self.downsample is not None:
            ds = self.downsample
            if isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[0], nn.Conv2d):
                identity = F_.conv_bn_act(x, ds[0], ds[1], relu=False)
            else:
                identity = ds(x)
        return layers.light_block_tail(out, identity, self.mrla, self.bn_mrla, self.drop_path, pre_activation=True)


class MRLA_Bottleneck_base(_BottleneckTrunk):
    """Bottleneck + MRLA-base tail (resnet_mrla_base.py:55-131; `MRLA_Bottleneck` there)."""

    def __init__(self, inplanes, planes, stride=1, downsample=None, SE=False, ECA_size=None, groups=1, base_width=64,
                 dilation=1, norm_layer=nn.BatchNorm2d, drop_path=0.0, init_cell=False, channel_wise_mrla=False):
        super().__init__(inplanes, planes, stride, downsample, SE, ECA_size, groups, base_width, dilation, norm_layer)
        self.mrla = layers.mrla_base_module(input_dim=planes * self.expansion, init_cell=init_cell,
                                            channel_wise=channel_wise_mrla)
        self.bn_mrla = self._norm(planes * self.expansion)
        self.drop_path = layers.DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def forward(self, x, prev_k, prev_v):
        # shortcut add + ReLU (and, on channels_last stages, bn3's affine) run inside the MRLA pooling pass
        pre, identity = self.trunk_pre(x, defer_bn3=layers.base_tail_defers_bn3(self.mrla, self.bn_mrla, prev_v))
        return layers.base_block_tail(pre, prev_k, prev_v, self.mrla, self.bn_mrla, self.drop_path, identity=identity)


class _ResNetMRLA(nn.Module):
    """Shared construction logic of the light and base networks."""

    def _setup(self, num_classes, SE, ECA, groups, width_per_group, replace_stride_with_dilation, norm_layer,
               drop_rate, drop_path):
        self._norm_layer = norm_layer or nn.BatchNorm2d
        self.num_classes, self.drop_rate, self.drop_path = num_classes, drop_rate, drop_path
        self.inplanes, self.dilation = 64, 1
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        if len(replace_stride_with_dilation) != 3:
            raise ValueError("replace_stride_with_dilation should be None "
                             "or a 3-element tuple, got {}".format(replace_stride_with_dilation))
        if ECA is None:
            ECA = [None] * 4
        elif len(ECA) != 4:
            raise ValueError("argument ECA should be a 4-element tuple, got {}".format(ECA))
        self.groups, self.base_width = groups, width_per_group
        self._SE, self._ECA, self._rswd = SE, ECA, replace_stride_with_dilation

    def _make_layer(self, block, planes, blocks, SE, ECA_size, stride=1, dilate=False, **first_only):
        norm_layer = self._norm_layer
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_conv1x1(self.inplanes, planes * block.expansion, stride),
                                       norm_layer(planes * block.expansion))
        shared = dict(SE=SE, ECA_size=ECA_size, groups=self.groups, base_width=self.base_width, norm_layer=norm_layer,
                      drop_path=self.drop_path)
        rest = {k: (False if k == "init_cell" else v) for k, v in first_only.items()}
        mods = [block(self.inplanes, planes, stride, downsample, dilation=previous_dilation, **shared, **first_only)]
        self.inplanes = planes * block.expansion
        mods += [block(self.inplanes, planes, dilation=self.dilation, **shared, **rest) for _ in range(1, blocks)]
        return mods

    def _head_and_init(self, block, zero_init_last_bn):
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, self.num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_last_bn:
            for m in self.modules():
                if isinstance(m, _BottleneckTrunk):
                    nn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, MRLA_BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _weight_bank(self):
        """The refreshed WeightBank of this network's 1x1 convolutions (one cast launch per step instead of one per
        convolution), or None when nothing of it applies (CPU model, weights not fp32)."""
        bank = self.__dict__.get("_bank")
        if bank is None:
            convs = []
            for m in self.modules():
                if isinstance(m, _BottleneckTrunk):
                    convs += [m.conv1, m.conv3]
                if isinstance(m, (_BottleneckTrunk, MRLA_BasicBlock)):
                    if isinstance(m.downsample, nn.Sequential) and len(m.downsample) == 2:
                        convs.append(m.downsample[0])
            bank = self.__dict__["_bank"] = F_.WeightBank(convs)
        if not torch.is_autocast_enabled("cuda"):
            return None                            # fp32 runs multiply with the master weights themselves
        return bank.refresh()

    def _stochastic_depth_blocks(self):
        """Blocks that will draw a stochastic-depth mask in this forward (0: none, each block draws its own)."""
        if not (self.training and self.drop_path):
            return 0
        n = self.__dict__.get("_n_trunks")
        if n is None:                      # (the block list is fixed after construction; counted once)
            n = sum(1 for m in self.modules() if isinstance(m, (_BottleneckTrunk, MRLA_BasicBlock)))
            self.__dict__["_n_trunks"] = n
        return n

    def forward(self, x):
        x = self.forward_features(x)
        x = torch.flatten(self.avgpool(x), 1)
        if self.drop_rate:
            x = F.dropout(x, p=float(self.drop_rate), training=self.training)
        return self.fc(x)


class ResNet_mrlal(_ResNetMRLA):
    """ResNet with an MRLA-light module after every bottleneck (resnet_mrla_light.py:122-238).

    `channels_last` (class attribute, default True): activations and convolution weights are kept in
    torch.channels_last inside the network -- MIOpen's fast bf16 kernels on gfx950 are NHWC-native (no layout
    transposes around every convolution) and the MRLA / BatchNorm HIP kernels have NHWC variants.  Inputs may be
    NCHW-contiguous as in the reference; logits, parameters' logical shapes and state_dict keys are unaffected."""
    channels_last = True

    def __init__(self, block, layers, num_classes=1000, SE=False, ECA=None, zero_init_last_bn=True, groups=1,
                 width_per_group=64, replace_stride_with_dilation=None, norm_layer=nn.BatchNorm2d, drop_rate=0.0,
                 drop_path=0.0):
        super().__init__()
        self._setup(num_classes, SE, ECA, groups, width_per_group, replace_stride_with_dilation, norm_layer, drop_rate,
                    drop_path)
        self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = self._norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        E, d = self._ECA, self._rswd
        self.layer1 = nn.Sequential(*self._make_layer(block, 64, layers[0], SE, E[0]))
        self.layer2 = nn.Sequential(*self._make_layer(block, 128, layers[1], SE, E[1], stride=2, dilate=d[0]))
        self.layer3 = nn.Sequential(*self._make_layer(block, 256, layers[2], SE, E[2], stride=2, dilate=d[1]))
        self.layer4 = nn.Sequential(*self._make_layer(block, 512, layers[3], SE, E[3], stride=2, dilate=d[2]))
        self._head_and_init(block, zero_init_last_bn)
        if self.channels_last:
            self.to(memory_format=torch.channels_last)

    def forward_features(self, x):
        if self.channels_last and x.is_cuda:
            x = _to_channels_last(x)
        with F_.batched_bookkeeping(self._stochastic_depth_blocks(), self._weight_bank() if x.is_cuda else None):
            x = F_.bn_relu_maxpool(self.conv1(x), self.bn1, self.maxpool)
            return self.layer4(self.layer3(self.layer2(self.layer1(x))))


def _to_channels_last(x):
    """The image batch in channels_last -- and, under autocast, already in the autocast dtype: the stem convolution would cast
    it anyway (same values), and one copy kernel instead of two halves the passes over the largest input of the step (154 MB
    fp32 at b = 256)."""
    if x.dtype == torch.float32 and torch.is_autocast_enabled("cuda"):
        return x.to(dtype=torch.get_autocast_dtype("cuda"), memory_format=torch.channels_last)
    return x.contiguous(memory_format=torch.channels_last)


class ResNet_mrlab(_ResNetMRLA):
    """ResNet (deep 3-conv stem) with an MRLA-base module after every bottleneck; the K/V history is threaded
    through the blocks of a stage (resnet_mrla_base.py:134-272).  `channels_last` as in ResNet_mrlal: the stage's value
    history then lives in slot-major NHWC rings."""
    channels_last = True

    def __init__(self, block, layers, num_classes=1000, SE=False, ECA=None, zero_init_last_bn=True, groups=1,
                 width_per_group=64, replace_stride_with_dilation=None, norm_layer=nn.BatchNorm2d, drop_rate=0.0,
                 drop_path=0.0, channel_wise_mrla=False):
        super().__init__()
        self._setup(num_classes, SE, ECA, groups, width_per_group, replace_stride_with_dilation, norm_layer, drop_rate,
                    drop_path)
        stem_width = 32
        nl = self._norm_layer
        self.conv1 = nn.Sequential(
            nn.Conv2d(3, stem_width, 3, stride=2, padding=1, bias=False), nl(stem_width), nn.ReLU(inplace=True),
            nn.Conv2d(stem_width, stem_width, 3, stride=1, padding=1, bias=False), nl(stem_width), nn.ReLU(inplace=True),
            nn.Conv2d(stem_width, self.inplanes, 3, stride=1, padding=1, bias=False))
        self.bn1 = nl(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        E, d = self._ECA, self._rswd
        extra = dict(init_cell=True, channel_wise_mrla=channel_wise_mrla)
        stages = [self._make_layer(block, 64, layers[0], SE, E[0], **extra),
                  self._make_layer(block, 128, layers[1], SE, E[1], stride=2, dilate=d[0], **extra),
                  self._make_layer(block, 256, layers[2], SE, E[2], stride=2, dilate=d[1], **extra),
                  self._make_layer(block, 512, layers[3], SE, E[3], stride=2, dilate=d[2], **extra)]
        for mods in stages:                       # tell the first block how deep its stage's history gets
            mods[0].mrla.mrla.history_hint = len(mods)
        self.stages = nn.ModuleList([nn.ModuleList(m) for m in stages])
        self._head_and_init(block, zero_init_last_bn)
        if self.channels_last:
            self.to(memory_format=torch.channels_last)

    def forward_features(self, x):
        if self.channels_last and x.is_cuda:
            x = _to_channels_last(x)
        with F_.batched_bookkeeping(self._stochastic_depth_blocks(), self._weight_bank() if x.is_cuda else None):
            x = F_.bn_relu_maxpool(self.conv1(x), self.bn1, self.maxpool)
            k = v = None
            for stage in self.stages:
                for blk in stage:
                    x, k, v = blk(x, k, v)
        return x


def resnet50_mrlab(**kwargs):
    print("Constructing resnet50_mrla-base......")
    return ResNet_mrlab(MRLA_Bottleneck_base, [3, 4, 6, 3], **kwargs)


def resnet101_mrlab(**kwargs):
    print("Constructing resnet101_mrla-base......")
    return ResNet_mrlab(MRLA_Bottleneck_base, [3, 4, 23, 3], **kwargs)


def resnet18_mrlal(**kwargs):
    """Build-side extension (see MRLA_BasicBlock): BASELINE.json's config 1 by its literal name."""
    print("Constructing resnet18_mrla-light (build-side BasicBlock extension)......")
    return ResNet_mrlal(MRLA_BasicBlock, [2, 2, 2, 2], **kwargs)


def resnet34_mrlal(**kwargs):
    """Build-side extension (see MRLA_BasicBlock)."""
    print("Constructing resnet34_mrla-light (build-side BasicBlock extension)......")
    return ResNet_mrlal(MRLA_BasicBlock, [3, 4, 6, 3], **kwargs)


def resnet50_mrlal(**kwargs):
    print("Constructing resnet50_mrla-light......")
    return ResNet_mrlal(MRLA_Bottleneck, [3, 4, 6, 3], **kwargs)


def resnet101_mrlal(**kwargs):
    print("Constructing resnet101_mrla-light......")
    return ResNet_mrlal(MRLA_Bottleneck, [3, 4, 23, 3], **kwargs)
