"""ResNet-50 v1.5 topology restated (TEST INFRASTRUCTURE ONLY — never imported by the product).

The reference builds its backbone from a third-party dependency that is absent from
/root/reference and from this image: ``torchvision.models.resnet50`` wrapped by
``torchvision.models._utils.IntermediateLayerGetter`` (reference call sites:
src/models/backbone.py:69 and :90-92).  torchvision is un-vendored and unpinned by the
reference (no requirements file); the topology restated here is the published
torchvision "ResNet v1.5" (stride on the 3x3 conv of each Bottleneck, torchvision >= 0.4,
unchanged through 0.2x): keys conv1/bn1/layer{1..4}.{i}.conv{1..3}/bn{1..3}/downsample.{0,1}.

Parity for the backbone is therefore "unpinned" by any reference-owned vector (the
reference has no tests); it is pinned in practice by torch.nn.functional.conv2d on CPU
and by the reference's own FrozenBatchNorm2d (src/models/backbone.py:45-55), which the
reference passes in as ``norm_layer``.
"""
from collections import OrderedDict

import torch
from torch import nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample, dilation, norm_layer):
        super().__init__()
        width = planes
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = norm_layer(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=dilation,
                               dilation=dilation, bias=False)
        self.bn2 = norm_layer(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class ResNet50(nn.Module):
    def __init__(self, norm_layer, replace_stride_with_dilation=(False, False, False)):
        super().__init__()
        self.inplanes, self.dilation = 64, 1
        self._norm = norm_layer
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make(64, 3, 1, False)
        self.layer2 = self._make(128, 4, 2, replace_stride_with_dilation[0])
        self.layer3 = self._make(256, 6, 2, replace_stride_with_dilation[1])
        self.layer4 = self._make(512, 3, 2, replace_stride_with_dilation[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make(self, planes, blocks, stride, dilate):
        prev_dil = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 self._norm(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down, prev_dil, self._norm)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, 1, None, self.dilation, self._norm))
        return nn.Sequential(*layers)


def resnet50(pretrained=False, norm_layer=None, replace_stride_with_dilation=None, **kw):
    """Stand-in for torchvision.models.resnet50; `pretrained` is ignored (URL fetch, no network)."""
    return ResNet50(norm_layer or nn.BatchNorm2d, replace_stride_with_dilation or (False,) * 3)


class IntermediateLayerGetter(nn.ModuleDict):
    """Published behaviour of torchvision.models._utils.IntermediateLayerGetter: keep the
    model's children up to the last requested one; forward returns an OrderedDict of the
    requested intermediate outputs under their new names."""

    def __init__(self, model, return_layers):
        orig = dict(return_layers)
        remaining = dict(return_layers)
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = orig

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out
