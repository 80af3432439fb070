"""Oracle (TEST INFRASTRUCTURE ONLY): ResNet-50 / ResNet-18 trunks in plain torch-CPU fp32.

ResNet-50: timm `resnet50` with features_only=True as pyiqa's CFANet builds it for topiq_nr
(reference call site models/pyiqa_scorer.py:108-111,212; architecture [DEP-KNOWLEDGE]: 7x7/2 stem, 3x3/2
maxpool, bottlenecks 3-4-6-3 with the stride on the 3x3 conv, outputs after stem-ReLU and each layer).
Second opinion for ResNet-50: equal to transformers' ResNetModel on all five outputs (tests/test_oracle_second_opinion.py).
ResNet-18: torchvision `resnet18` children[:-2] exactly as reference models/samp_net.py:652-662 wires it
(nn.Sequential numbering 0=conv1 1=bn1 2=relu 3=maxpool 4..7=layer1..4) -> parity unpinned for the class
itself (torchvision is not installed), pinned end-to-end through SAMPNet's golden vectors.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Bottleneck(nn.Module):
    def __init__(self, inpl, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inpl, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if down:
            self.downsample = nn.Sequential(nn.Conv2d(inpl, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + idt)


class _Basic(nn.Module):
    def __init__(self, inpl, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inpl, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if down:
            self.downsample = nn.Sequential(nn.Conv2d(inpl, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + idt)


def _make_layers(block, exp, blocks):
    layers, inpl = [], 64
    for li, nb in enumerate(blocks):
        planes = 64 * 2 ** li
        seq = []
        for bi in range(nb):
            stride = 2 if (bi == 0 and li > 0) else 1
            down = stride != 1 or inpl != planes * exp
            seq.append(block(inpl, planes, stride, down))
            inpl = planes * exp
        layers.append(nn.Sequential(*seq))
    return layers


class ResNet50Features(nn.Module):
    """timm resnet50 features_only: returns [act1, layer1, layer2, layer3, layer4]."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1, self.layer2, self.layer3, self.layer4 = _make_layers(_Bottleneck, 4, [3, 4, 6, 3])

    def forward(self, x):
        f0 = F.relu(self.bn1(self.conv1(x)))
        y = F.max_pool2d(f0, 3, 2, 1)
        f1 = self.layer1(y)
        f2 = self.layer2(f1)
        f3 = self.layer3(f2)
        f4 = self.layer4(f3)
        return [f0, f1, f2, f3, f4]


def resnet18_trunk():
    """nn.Sequential(*list(resnet18.children())[:-2]) — same child order/keys as torchvision's."""
    l1, l2, l3, l4 = _make_layers(_Basic, 1, [2, 2, 2, 2])
    return nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                         nn.MaxPool2d(3, 2, 1), l1, l2, l3, l4)


class _FakeTorchvisionResNet18(nn.Module):
    """Stand-in for torchvision.models.resnet18 with torchvision's child order
    (conv1,bn1,relu,maxpool,layer1-4,avgpool,fc); used ONLY by tests/golden/make_samp_golden.py so that the
    reference's models/samp_net.py can be imported in a container without torchvision."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1, self.layer2, self.layer3, self.layer4 = _make_layers(_Basic, 1, [2, 2, 2, 2])
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, 1000)
