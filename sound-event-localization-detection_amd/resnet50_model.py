"""ResNet-50 encoder + Conformer temporal model for SELD (drop-in for the reference's
``resnet50_model.py``; same names / ``state_dict`` keys).

The encoder is the torchvision ResNet-50 layout ([3, 4, 6, 3] bottlenecks, expansion 4) with a
3x3 stem and every stride applied to FREQUENCY only -- (1, 2) -- so the 250 time frames survive:
64 mel bins -> 2 (resnet50_model.py:50-118).  [B,2048,T,2] -> Linear 4096->512 -> dropout ->
4 Conformer blocks (d=512, 8 heads, FF 2048, k=31) -> head 512->1024->LN->ReLU->Dropout->9072.
"""
import torch
import torch.nn as nn

from seld_layernorm import run_head
from seld_linear import SeldLinear

from model_conformer import ConformerBlock
from model_crnn import conv3x3


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, in_channels, out_channels, stride=1, downsample=None):
        super().__init__()
        wide = out_channels * self.expansion
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.conv3 = nn.Conv2d(out_channels, wide, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(wide)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        if x.is_cuda:
            import seld_convtail as tail           # fused BatchNorm -> [+ shortcut] -> ReLU (csrc/convtail.hip)
            from seld_linear import conv1x1
            y = conv1x1(self.conv1, x)            # 1x1 convolutions: GEMMs on the channels-last rows
            y = tail.bn_relu(self.bn1, y) if tail.bn_applicable(self.bn1, y) else self.relu(self.bn1(y))
            y = conv3x3(self.conv2, y)            # stride-1 3x3: data gradient as a forward convolution
            y = tail.bn_relu(self.bn2, y) if tail.bn_applicable(self.bn2, y) else self.relu(self.bn2(y))
            y = conv1x1(self.conv3, y)
            if tail.bn_applicable(self.bn3, y) and shortcut.shape == y.shape and shortcut.dtype == y.dtype:
                return tail.bn_relu(self.bn3, y, residual=shortcut)
            return self.relu(self.bn3(y) + shortcut)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + shortcut)


class ResNet50Encoder(nn.Module):
    def __init__(self, in_channels=4, layers=[3, 4, 6, 3]):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=3, stride=(1, 2), padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=(1, 2), padding=1)
        self.layer1 = self._make_layer(Bottleneck, 64, layers[0], stride=1)
        self.layer2 = self._make_layer(Bottleneck, 128, layers[1], stride=(1, 2))
        self.layer3 = self._make_layer(Bottleneck, 256, layers[2], stride=(1, 2))
        self.layer4 = self._make_layer(Bottleneck, 512, layers[3], stride=(1, 2))
        self.out_channels = 2048
        self.out_freq = 2

    def _make_layer(self, block, planes, blocks, stride=1):
        wide = planes * block.expansion
        downsample = None
        if stride != 1 or self.inplanes != wide:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, wide, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(wide))
        stack = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = wide
        stack += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*stack)

    def forward(self, x):
        x = self.conv1(x)
        if x.is_cuda:
            import seld_convtail as tail
            x = tail.bn_relu(self.bn1, x) if tail.bn_applicable(self.bn1, x) else self.relu(self.bn1(x))
        else:
            x = self.relu(self.bn1(x))
        x = self.maxpool(x)
        x = self.layer3(self.layer2(self.layer1(x)))
        if x.is_cuda:
            # backward-pass cut points of the data-parallel captured step (identity otherwise, seld_cut.py): the
            # Conformer + head gradients (72 MB bf16) travel under the whole encoder backward, layer4's (30 MB) under
            # layer3..stem
            import seld_cut
            return seld_cut.boundary(self.layer4(seld_cut.boundary(x, level=2)), level=1)
        return self.layer4(x)


class SELD_ResNet50_Conformer(nn.Module):
    def __init__(self, n_channels=4, n_mels=64, grid_size=(18, 36), num_classes=14,
                 conf_d_model=512, conf_n_heads=8, conf_n_layers=4, conf_kernel_size=31, dropout=0.3):
        super().__init__()
        self.I, self.J = grid_size
        self.grid_cells = self.I * self.J
        self.num_classes = num_classes
        self.encoder = ResNet50Encoder(in_channels=n_channels)
        self.enc_feat_dim = self.encoder.out_channels * (n_mels // 32)
        self.proj = SeldLinear(self.enc_feat_dim, conf_d_model)
        self.dropout = nn.Dropout(dropout)
        self.conformer_blocks = nn.ModuleList([
            ConformerBlock(d_model=conf_d_model, n_heads=conf_n_heads, d_ff=conf_d_model * 4,
                           kernel_size=conf_kernel_size, dropout=dropout)
            for _ in range(conf_n_layers)])
        self.head = nn.Sequential(
            SeldLinear(conf_d_model, 1024),
            nn.LayerNorm(1024),
            nn.ReLU(),
            nn.Dropout(dropout),
            SeldLinear(1024, self.grid_cells * num_classes),
        )

    def forward(self, x):
        """x [B, T, C, F] -> logits [B, T, G, M]; time is the conv 'height', frequency the 'width'."""
        batch, frames = x.shape[0], x.shape[1]
        y = x.permute(0, 2, 1, 3)
        if y.is_cuda:
            y = y.contiguous(memory_format=torch.channels_last)
        if y.is_cuda:
            import seld_convtail
            with seld_convtail.batched_counters():                   # one launch for the 53 BatchNorm counters
                y = self.encoder(y)                                  # [B, 2048, T, F/32]
        else:
            y = self.encoder(y)
        proj = None
        if y.is_cuda:
            from seld_linear import linear_on_channels_last_features
            proj = linear_on_channels_last_features(self.proj, y)          # no 65 MB feature copy: the weight's columns move
        if proj is None:
            proj = self.proj(y.permute(0, 2, 1, 3).reshape(batch, frames, -1))
        y = self.dropout(proj)
        for block in self.conformer_blocks:
            y = block(y)
        return run_head(self.head, y).view(batch, frames, self.grid_cells, self.num_classes)
