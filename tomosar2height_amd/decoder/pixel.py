"""Grid-feature decoder (reference: tomosar2height/decoder/pixel.py).

``PixelwiseDecoder.forward`` sums the feature planes, resamples them to the output raster with the HIP
bilinear kernel (``ops.upsample_bilinear`` == ``F.interpolate(..., align_corners=True)``, image-plane add
fused when it is already at output size) and runs the conv / per-pixel FC head.  In channels_last mode the 3x3
convolutions run on the implicit-GEMM kernels of csrc/conv.hip, otherwise on MIOpen.  Parameter names follow the reference: ``conv_decoder.conv{1..4}``,
``conv_decoder_footprint.*``, ``fc_decoder.{blocks,fc_out}``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import grid, mlp, ops
from ..block import ResnetBlockFC


class ConvDecoder(nn.Module):
    """pixel.py:8-32: three 3x3 convs with a dense skip-concat into a 1x1 head (32+64+128+64 = 288)."""

    def __init__(self, in_channels=32, out_channels=1, leaky=False):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(64, 128, kernel_size=3, padding=1)
        self.conv3 = nn.Conv2d(128, 64, kernel_size=3, padding=1)
        self.conv4 = nn.Conv2d(288, out_channels, kernel_size=1)
        self.act = F.leaky_relu if leaky else F.relu

        self.leaky = leaky
        self.channels_last = False

    def forward(self, x):
        if self.channels_last and not self.leaky and x.is_cuda:
            # implicit-GEMM 3x3 convs with fused bias/ReLU and a concat-free 288 -> 1 head as one autograd node (grid.py)
            return grid.conv_decoder(x, self.conv1, self.conv2, self.conv3, self.conv4)
        x1 = self.act(self.conv1(x))
        x2 = self.act(self.conv2(x1))
        x3 = self.act(self.conv3(x2))
        return self.conv4(torch.cat([x, x1, x2, x3], dim=1))


class FCDecoder(nn.Module):
    """pixel.py:35-58."""

    def __init__(self, in_channels=32, out_channels=1, n_blocks=5, leaky=False):
        super().__init__()
        self.blocks = nn.ModuleList([ResnetBlockFC(in_channels) for _ in range(n_blocks)])
        self.fc_out = nn.Linear(in_channels, out_channels)
        self.leaky = leaky

    def forward(self, x):
        for block in self.blocks:
            x = block(x)
        x = F.leaky_relu(x) if self.leaky else F.relu(x)
        return mlp.linear(x, self.fc_out.weight, self.fc_out.bias)


class PixelwiseDecoder(nn.Module):
    """pixel.py:61-125.  As in the reference, ``mode='fc'`` passes ``leaky`` into FCDecoder's ``n_blocks``
    slot (pixel.py:88), so the default fc head has zero ResNet blocks."""

    def __init__(self, hidden_dim=32, out_dim=1, output_size=512, leaky=False, sample_mode="bilinear", mode="conv",
                 use_footprint=False, **kwargs):
        super().__init__()
        if sample_mode != "bilinear":
            raise NotImplementedError("only sample_mode='bilinear' is built")
        self.mode, self.use_footprint = mode, use_footprint
        self.sample_mode, self.output_size = sample_mode, output_size
        if mode == "conv":
            self.conv_decoder = ConvDecoder(hidden_dim, out_dim, leaky)
            if use_footprint:
                self.conv_decoder_footprint = ConvDecoder(hidden_dim, out_dim)
        elif mode == "fc":
            self.fc_decoder = FCDecoder(hidden_dim, out_dim, leaky)
            if use_footprint:
                self.fc_decoder_footprint = FCDecoder(hidden_dim, out_dim)
        else:
            raise ValueError("Invalid mode. Use 'conv' or 'fc'.")

        self.channels_last = False

    def set_channels_last(self, flag: bool):
        self.channels_last = bool(flag)
        for m in self.modules():
            if isinstance(m, ConvDecoder):
                m.channels_last = bool(flag)

    def _resample(self, x, addend=None):
        if self.channels_last and x.shape[1] % 4 == 0:
            return grid.upsample_bilinear_cl(x, self.output_size, addend)
        return ops.upsample_bilinear(x, self.output_size, addend)

    def forward(self, feature_planes):
        xy, image = feature_planes.get("xy"), feature_planes.get("image")
        if xy is None and image is None:
            raise ValueError("PixelwiseDecoder: no 'xy' or 'image' plane given")
        if xy is not None and image is not None:
            if image.shape[-1] == self.output_size and image.shape[-2] == self.output_size:
                c = self._resample(xy, addend=image)                                # pixel.py:107+110 fused
            else:
                c = self._resample(xy) + self._resample(image)
        else:
            c = self._resample(xy if xy is not None else image)
        x_footprint = None
        if self.mode == "conv":
            x = self.conv_decoder(c).permute(0, 2, 3, 1)
            if self.use_footprint:
                x_footprint = self.conv_decoder_footprint(c).permute(0, 2, 3, 1)
        else:
            c = c.permute(0, 2, 3, 1)
            x = self.fc_decoder(c)
            if self.use_footprint:
                x_footprint = self.fc_decoder_footprint(c)
        return x, x_footprint
