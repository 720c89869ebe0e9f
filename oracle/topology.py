"""Third-party topologies the reference relies on, restated from the published
architectures (torchvision is absent from /root/reference and from this image).

PARITY UNPINNED: the reference calls ``torchvision.models.vgg19(pretrained=True).features``
(dream/models.py:587) and ``torchvision.models.resnet101(pretrained=...)``
(dream/models.py:22); torchvision is an un-vendored, unpinned dependency
(requirements.txt:16).  What is restated here is the published layer list of
"VGG-19 configuration E" (Simonyan & Zisserman 2014) and "ResNet-101 v1.5"
(He et al. 2015, stride on the 3x3 conv).  Cross-check available: parameter counts vs
the released checkpoint byte sizes (trained_models/DOWNLOAD.sh:12-38), see
tests/test_oracle_models.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import torch.nn as nn

# VGG-19 "E": numbers = 3x3 conv (pad 1) output channels followed by ReLU, "M" = 2x2 max-pool.
_VGG19_E = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M",
            512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def vgg19_features():
    """Sequential with torchvision's index layout: conv at 0,2,5,7,10,12,14,16,19,...,34."""
    layers, cin = [], 3
    for v in _VGG19_E:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers.append(nn.Conv2d(cin, v, kernel_size=3, padding=1))
            layers.append(nn.ReLU(inplace=True))
            cin = v
    return nn.Sequential(*layers)


class Bottleneck(nn.Module):
    """1x1 -> 3x3(stride) -> 1x1(x4), BN after each, residual add, ReLU."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class ResNet101(nn.Module):
    BLOCKS = (3, 4, 23, 3)

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self._inplanes = 64
        self.layer1 = self._stage(64, self.BLOCKS[0], 1)
        self.layer2 = self._stage(128, self.BLOCKS[1], 2)
        self.layer3 = self._stage(256, self.BLOCKS[2], 2)
        self.layer4 = self._stage(512, self.BLOCKS[3], 2)

    def _stage(self, planes, n, stride):
        ds = None
        if stride != 1 or self._inplanes != planes * 4:
            ds = nn.Sequential(
                nn.Conv2d(self._inplanes, planes * 4, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * 4))
        blocks = [Bottleneck(self._inplanes, planes, stride, ds)]
        self._inplanes = planes * 4
        blocks += [Bottleneck(self._inplanes, planes) for _ in range(n - 1)]
        return nn.Sequential(*blocks)
