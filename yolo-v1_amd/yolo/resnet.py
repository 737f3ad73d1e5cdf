"""ResNet-50 trunk (conv1 .. layer4) written against torchvision's published architecture.

The reference wraps ``torchvision.models.resnet50`` (src/yolo/models.py:131-176, torchvision 0.23.0 in
its lock file) and keeps ``children()[:-2]``: conv1, bn1, relu, maxpool, layer1..layer4.  torchvision is an
un-vendored dependency whose code is not part of the reference, so this is a from-the-paper restatement of
the "v1.5" variant torchvision ships (stride on the 3x3 conv of a bottleneck, conv1 7x7/s2 without bias,
BatchNorm eps 1e-5, ReLU, MaxPool 3x3/s2/p1, blocks [3,4,6,3], expansion 4) with the SAME module names, so a
``state_dict`` saved through torchvision loads here and vice versa
(``extractor.0.weight``, ``extractor.1.running_mean``, ``extractor.4.0.conv1.weight``,
``extractor.5.0.downsample.0.weight`` ...).  Numeric parity with torchvision cannot be pinned in this
environment ("parity unpinned"); the architecture is pinned by parameter count and shapes.
"""

from __future__ import annotations

import torch
import torch.nn as nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: nn.Module | None = None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


def _make_layer(inplanes: int, planes: int, blocks: int, stride: int) -> nn.Sequential:
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
    layers = [Bottleneck(inplanes, planes, stride, down)]
    layers += [Bottleneck(planes * 4, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


def resnet50_trunk() -> nn.Sequential:
    """[conv1, bn1, relu, maxpool, layer1, layer2, layer3, layer4] = list(resnet50().children())[:-2]."""
    mods = [nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, stride=2, padding=1),
            _make_layer(64, 64, 3, 1), _make_layer(256, 128, 4, 2), _make_layer(512, 256, 6, 2), _make_layer(1024, 512, 3, 2)]
    trunk = nn.Sequential(*mods)
    for m in trunk.modules():       # torchvision's initialisation
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    return trunk
