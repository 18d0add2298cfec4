"""CSPDarkNet 'cnn' model (the fourth ``MODEL_TYPE``; drop-in for the reference's ``model.py``).

Out of the accelerated scope (not named by any benchmark configuration): it is
kept so that ``trainer.py`` can construct every MODEL_TYPE and checkpoints interchange.  Stock
PyTorch-ROCm ops only, no custom kernels.  Each time frame is treated as a [C, 64, 1] image
(model.py:182-189); P3/P4/P5 are reduced to 256 channels, fused, pooled to the I x J grid and
classified per cell (model.py:146-221).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Conv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = nn.SiLU(inplace=True)

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class Bottleneck(nn.Module):
    def __init__(self, in_channels, out_channels, shortcut=True):
        super().__init__()
        self.cv1 = Conv(in_channels, out_channels, 1, 1, 0)
        self.cv2 = Conv(out_channels, out_channels, 3, 1, 1)
        self.add = shortcut and in_channels == out_channels

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C3(nn.Module):
    def __init__(self, in_channels, out_channels, n_blocks=1, shortcut=True):
        super().__init__()
        hidden = out_channels // 2
        self.cv1 = Conv(in_channels, hidden, 1, 1, 0)
        self.cv2 = Conv(in_channels, hidden, 1, 1, 0)
        self.cv3 = Conv(2 * hidden, out_channels, 1, 1, 0)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut) for _ in range(n_blocks)])

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), dim=1))


class SPPF(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=5):
        super().__init__()
        hidden = in_channels // 2
        self.cv1 = Conv(in_channels, hidden, 1, 1, 0)
        self.cv2 = Conv(hidden * 4, out_channels, 1, 1, 0)
        self.m = nn.MaxPool2d(kernel_size=kernel_size, stride=1, padding=kernel_size // 2)

    def forward(self, x):
        pyramid = [self.cv1(x)]
        for _ in range(3):
            pyramid.append(self.m(pyramid[-1]))
        return self.cv2(torch.cat(pyramid, dim=1))


class CSPDarkNet53(nn.Module):
    def __init__(self, in_channels=4, base_channels=64, depth_multiple=1.0, width_multiple=1.0):
        super().__init__()
        ch = lambda c: max(round(c * width_multiple), 1)     # noqa: E731
        dp = lambda n: max(round(n * depth_multiple), 1)     # noqa: E731
        self.stem = Conv(in_channels, ch(base_channels), 3, 1, 1)
        self.stage1 = nn.Sequential(Conv(ch(64), ch(128), 3, 2, 1), C3(ch(128), ch(128), n_blocks=dp(3)))
        self.stage2 = nn.Sequential(Conv(ch(128), ch(256), 3, 2, 1), C3(ch(256), ch(256), n_blocks=dp(6)))
        self.stage3 = nn.Sequential(Conv(ch(256), ch(512), 3, 2, 1), C3(ch(512), ch(512), n_blocks=dp(9)))
        self.stage4 = nn.Sequential(Conv(ch(512), ch(1024), 3, 2, 1), C3(ch(1024), ch(1024), n_blocks=dp(3)),
                                    SPPF(ch(1024), ch(1024)))
        self.out_channels = [ch(128), ch(256), ch(512), ch(1024)]

    def forward(self, x):
        p2 = self.stage1(self.stem(x))
        p3 = self.stage2(p2)
        p4 = self.stage3(p3)
        return [p2, p3, p4, self.stage4(p4)]


class SMRSELDWithCSPDarkNet(nn.Module):
    def __init__(self, n_channels=4, grid_size=(18, 36), num_classes=14, use_small=True):
        super().__init__()
        self.I, self.J = grid_size
        self.grid_cells = self.I * self.J
        self.num_classes = num_classes
        scale = dict(depth_multiple=0.33, width_multiple=0.5) if use_small else {}
        self.backbone = CSPDarkNet53(in_channels=n_channels, **scale)
        c3, c4, c5 = self.backbone.out_channels[1:]
        self.reduce_p3 = nn.Conv2d(c3, 256, kernel_size=1)
        self.reduce_p4 = nn.Conv2d(c4, 256, kernel_size=1)
        self.reduce_p5 = nn.Conv2d(c5, 256, kernel_size=1)
        self.conv_fuse = nn.Sequential(
            nn.Conv2d(256 * 3, 512, 3, padding=1, bias=False), nn.BatchNorm2d(512), nn.SiLU(),
            nn.Conv2d(512, 256, 1, bias=False), nn.BatchNorm2d(256), nn.SiLU())
        self.grid_pool = nn.AdaptiveAvgPool2d((self.I, self.J))
        self.classifier = nn.Sequential(
            nn.Linear(256, 128), nn.LayerNorm(128), nn.ReLU(inplace=True), nn.Dropout(0.3),
            nn.Linear(128, num_classes))

    def forward(self, x):
        """x [B, T, C, F] -> logits [B, T, G, M]; every frame is an independent [C, F, 1] image."""
        batch, frames, channels, freq = x.shape
        _, p3, p4, p5 = self.backbone(x.reshape(batch * frames, channels, freq, 1))
        p3 = self.reduce_p3(p3)
        size = p3.shape[2:]
        p4 = F.interpolate(self.reduce_p4(p4), size=size, mode="bilinear", align_corners=False)
        p5 = F.interpolate(self.reduce_p5(p5), size=size, mode="bilinear", align_corners=False)
        fused = self.conv_fuse(torch.cat([p3, p4, p5], dim=1))
        cells = self.grid_pool(fused).view(batch * frames, 256, self.grid_cells).transpose(1, 2)
        cells = F.normalize(cells, p=2, dim=-1)
        return self.classifier(cells).view(batch, frames, self.grid_cells, self.num_classes)
