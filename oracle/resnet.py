"""CPU restatement (TEST INFRASTRUCTURE ONLY) of torchvision.models.resnet50 with fc = Identity, as the reference's dataset classes
use it to compute retrieval features (/root/reference/dataloader_ref_cluster.py:41-44, 241-261; dataloader_CLC.py:65-68, 250-294).

torchvision is not installed here and the pretrained weights are unreachable, so this is the published ResNet-50 v1.5 architecture
(stride on the 3x3 convolution of each bottleneck) in plain torch.nn with torchvision's module / parameter names — **parity unpinned**
against torchvision itself; it pins the product's HIP extractor (clc_amd.features) to plain PyTorch fp32 on seeded weights.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
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
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._layer(64, 3, 1)
        self.layer2 = self._layer(128, 4, 2)
        self.layer3 = self._layer(256, 6, 2)
        self.layer4 = self._layer(512, 3, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Identity()

    def _layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def trunk(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    def forward(self, x):   # == resnet50(x) with fc = Identity: [N, 2048]
        return torch.flatten(self.avgpool(self.trunk(x)), 1)


def spatial_pyramid_pooling(x, levels=(1, 2, 4)):
    """dataloader_CLC.py:250-256"""
    return torch.cat([F.adaptive_max_pool2d(x, output_size=(l, l)).view(x.size(0), -1) for l in levels], dim=1)


def seed_weights(model, seed=0):
    """Seeded stand-in for the pretrained weights: kaiming-normal filters, BatchNorm affine parameters and running statistics drawn
    so that every BN actually does something (a freshly constructed BN is the identity in eval mode)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (1.6 / fan) ** 0.5)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(0.7 + 0.6 * torch.rand(m.weight.shape, generator=g))
                m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.2 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(0.6 + 0.8 * torch.rand(m.running_var.shape, generator=g))
    return model.eval()
