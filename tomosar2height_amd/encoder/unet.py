"""Plain convolutional U-Net (reference: tomosar2height/encoder/unet.py:112-187), the image encoder of
``use_image=true`` configs, kept with the reference's parameter names so cloud+image checkpoints load.  Pure dense
convolution: in channels_last mode (the model's default) its 3x3, 2x2-transposed
and 1x1 convolutions with 16-aligned channel counts run on the same implicit-GEMM kernels as the ALTO grid side
(grid.py, csrc/conv.hip) -- every layer except the first (3 input channels)."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init

from .. import grid


class _Level(nn.Module):
    channels_last = False

    def _conv_pair(self, x):
        if self.channels_last:
            return grid.conv3x3_chain(x, (self.conv1, self.conv2))
        return F.relu(self.conv2(F.relu(self.conv1(x))))


class DownConv(_Level):
    def __init__(self, in_channels, out_channels, pooling=True):
        super().__init__()
        self.pooling = pooling
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        if pooling:
            self.pool = nn.MaxPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        x = self._conv_pair(x)
        if not self.pooling:
            return x, x
        return (grid.maxpool2x2(x, self.pool) if self.channels_last else self.pool(x)), x


class UpConv(_Level):
    def __init__(self, in_channels, out_channels, merge_mode="concat", up_mode="transpose"):
        super().__init__()
        self.merge_mode = merge_mode
        if up_mode == "transpose":                                # unet.py upconv2x2
            self.upconv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2)
        else:
            self.upconv = nn.Sequential(nn.Upsample(mode="bilinear", scale_factor=2), nn.Conv2d(in_channels, out_channels, 1))
        self.conv1 = nn.Conv2d(2 * out_channels if merge_mode == "concat" else out_channels, out_channels, 3, padding=1)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)

    def forward(self, from_down, from_up):
        if not self.channels_last:
            up = self.upconv(from_up)
        elif isinstance(self.upconv, nn.Sequential):
            up = grid.upsample_conv1x1(from_up, self.upconv)
        else:
            up = grid.upconv2x2(from_up, self.upconv)
        x = torch.cat((up, from_down), 1) if self.merge_mode == "concat" else up + from_down
        return self._conv_pair(x)


class UNet(nn.Module):
    def __init__(self, num_classes, in_channels=3, depth=5, start_filts=64, up_mode="transpose",
                 merge_mode="concat", **kwargs):
        super().__init__()
        if up_mode not in ("transpose", "upsample"):
            raise ValueError(f"Invalid up_mode: {up_mode}")
        if merge_mode not in ("concat", "add"):
            raise ValueError(f"Invalid merge_mode: {merge_mode}")
        if up_mode == "upsample" and merge_mode == "add":
            raise ValueError("up_mode 'upsample' is incompatible with merge_mode 'add'.")
        self.num_classes, self.in_channels, self.start_filts, self.depth = num_classes, in_channels, start_filts, depth
        self.down_convs, self.up_convs = nn.ModuleList(), nn.ModuleList()
        outs = in_channels
        for i in range(depth):
            ins = in_channels if i == 0 else outs
            outs = start_filts * (2 ** i)
            self.down_convs.append(DownConv(ins, outs, pooling=i < depth - 1))
        for _ in range(depth - 1):
            ins, outs = outs, outs // 2
            self.up_convs.append(UpConv(ins, outs, up_mode=up_mode, merge_mode=merge_mode))
        self.conv_final = nn.Conv2d(outs, num_classes, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                init.xavier_normal_(m.weight)
                init.constant_(m.bias, 0)

    def set_channels_last(self, flag: bool):
        for m in self.modules():
            if isinstance(m, _Level):
                m.channels_last = bool(flag)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("tomosar2height_amd: expected device tensors; there is no CPU path")
        skips = []
        for down in self.down_convs:
            x, before_pool = down(x)
            skips.append(before_pool)
        for i, up in enumerate(self.up_convs):
            x = up(skips[-(i + 2)], x)
        if self.down_convs[0].channels_last:
            return grid.conv1x1(x, self.conv_final)
        return self.conv_final(x)
