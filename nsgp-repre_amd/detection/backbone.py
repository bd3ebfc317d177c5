"""ResNet (``style='pytorch'``: the stride sits on the 3x3) + FPN with MMDetection's module names, so
parameter names -- and with them the NSGP names wiring, ignore keys and covariance keys -- are the
reference's (faster-rcnn_r50_fpn.py: depth 50, frozen_stages=1, norm_eval=True, BN requires_grad;
FPN in [256,512,1024,2048] -> 256, num_outs=5)."""
import torch.nn as nn
import torch.nn.functional as F

_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


class Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
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


class ResNet(nn.Module):
    def __init__(self, depth=50, frozen_stages=1, norm_eval=True, width=64):
        super().__init__()
        self.frozen_stages, self.norm_eval = frozen_stages, norm_eval
        self.conv1 = nn.Conv2d(3, width, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inplanes = width
        for i, n in enumerate(_BLOCKS[depth]):
            planes, stride = width * 2 ** i, 1 if i == 0 else 2
            blocks = []
            for b in range(n):
                down = None
                if b == 0:
                    down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
                blocks.append(Bottleneck(inplanes, planes, stride if b == 0 else 1, down))
                inplanes = planes * 4
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.out_channels = [width * 4 * 2 ** i for i in range(4)]
        self._freeze()

    def _freeze(self):
        if self.frozen_stages >= 0:
            for m in (self.conv1, self.bn1):
                m.eval()
                m.requires_grad_(False)
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f"layer{i}")
            m.eval()
            m.requires_grad_(False)

    def train(self, mode=True):
        super().train(mode)
        self._freeze()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        outs = []
        for i in range(4):
            x = getattr(self, f"layer{i + 1}")(x)
            outs.append(x)
        return tuple(outs)


class _ConvModule(nn.Module):
    """mmcv ConvModule without norm/activation: the conv sits under ``.conv``."""

    def __init__(self, cin, cout, k, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding)

    def forward(self, x):
        return self.conv(x)


class FPN(nn.Module):
    def __init__(self, in_channels=(256, 512, 1024, 2048), out_channels=256, num_outs=5):
        super().__init__()
        self.num_outs = num_outs
        self.lateral_convs = nn.ModuleList([_ConvModule(c, out_channels, 1) for c in in_channels])
        self.fpn_convs = nn.ModuleList([_ConvModule(out_channels, out_channels, 3, padding=1) for _ in in_channels])

    def forward(self, inputs):
        lat = [l(x) for l, x in zip(self.lateral_convs, inputs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
        outs = [c(x) for c, x in zip(self.fpn_convs, lat)]
        while len(outs) < self.num_outs:
            outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        return tuple(outs)
