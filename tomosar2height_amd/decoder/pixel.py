"""Grid-feature decoder (reference: tomosar2height/decoder/pixel.py).

``PixelwiseDecoder.forward`` sums the feature planes, resamples them to the output raster with the HIP
bilinear kernel (``ops.upsample_bilinear`` == ``F.interpolate(..., align_corners=True)``, image-plane add
fused when it is already at output size) and runs the conv / per-pixel FC head.  In channels_last mode the 3x3
convolutions run on the implicit-GEMM kernels of csrc/conv.hip, otherwise on MIOpen.
Parameter names follow the reference: ``conv_decoder.conv{1..4}``, ``conv_decoder_footprint.*``,
``fc_decoder.{blocks,fc_out}``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import grid, mlp, ops
from ..block import ResnetBlockFC

_HEAD_WIDTHS = (64, 128, 64)        # conv1..conv3 of pixel.py:20-22; conv4 reads the concat of the input and all three


class ConvDecoder(nn.Module):
    """pixel.py:8-32: three 3x3 convs with a dense skip-concat into a 1x1 head (32+64+128+64 = 288)."""

    def __init__(self, in_channels=32, out_channels=1, leaky=False):
        super().__init__()
        widths = (in_channels,) + _HEAD_WIDTHS
        for k in range(len(_HEAD_WIDTHS)):                     # registered in the reference's order: conv1, conv2, conv3
            self.add_module(f"conv{k + 1}", nn.Conv2d(widths[k], widths[k + 1], kernel_size=3, padding=1))
        # the reference hard-codes 288 = 32 + 64 + 128 + 64 input channels (pixel.py:23), i.e. in_channels = 32
        self.conv4 = nn.Conv2d(288, out_channels, kernel_size=1)
        self.leaky = bool(leaky)
        self.channels_last = False

    def _stack(self):
        return self.conv1, self.conv2, self.conv3

    def forward(self, x):
        if self.channels_last and not self.leaky and x.is_cuda:
            # implicit-GEMM 3x3 convs with fused bias/ReLU and a concat-free 288 -> 1 head as one autograd node (grid.py)
            return grid.conv_decoder(x, *self._stack(), self.conv4)
        if self.channels_last and x.is_cuda:
            # leaky=True (pixel.py:25; no shipped config): the same convolution kernels without their fused ReLU (a shape they do
            # not take goes through the library-fallback guard like everywhere else), F.leaky_relu as torch's element-wise op,
            # the concat-free head
            feats = [x]
            for conv in self._stack():
                feats.append(F.leaky_relu(grid.conv_bias_act(feats[-1], conv, relu=False)))
            return grid.head1x1(feats, self.conv4)
        activation = F.leaky_relu if self.leaky else F.relu
        feats = [x]
        for conv in self._stack():
            feats.append(activation(conv(feats[-1])))
        return self.conv4(torch.cat(feats, dim=1))


class FCDecoder(nn.Module):
    """pixel.py:35-58: ``n_blocks`` ResnetBlockFC layers and a linear head, per pixel."""

    def __init__(self, in_channels=32, out_channels=1, n_blocks=5, leaky=False):
        super().__init__()
        self.blocks = nn.ModuleList(ResnetBlockFC(in_channels) for _ in range(int(n_blocks)))
        self.fc_out = nn.Linear(in_channels, out_channels)
        self.leaky = bool(leaky)

    def forward(self, x):
        for layer in self.blocks:
            x = layer(x)
        if self.leaky:
            return mlp.linear(F.leaky_relu(x), self.fc_out.weight, self.fc_out.bias)
        return mlp.linear(x, self.fc_out.weight, self.fc_out.bias, relu_in=True)          # fc_out(act(x)), the ReLU in the loader


class PixelwiseDecoder(nn.Module):
    """pixel.py:61-125.  As in the reference, ``mode='fc'`` passes ``leaky`` into FCDecoder's ``n_blocks``
    slot (pixel.py:88), so the default fc head has zero ResNet blocks."""

    def __init__(self, hidden_dim=32, out_dim=1, output_size=512, leaky=False, sample_mode="bilinear", mode="conv",
                 use_footprint=False, **kwargs):
        super().__init__()
        if sample_mode not in ("bilinear", "bicubic"):
            raise NotImplementedError("sample_mode must be 'bilinear' (every shipped config, tomosar2height.yaml:27) or 'bicubic'.  "
                                      "Of the others torch accepts, 'nearest' cannot run in the reference either -- its decoder "
                                      "calls F.interpolate(..., mode=sample_mode, align_corners=True), pixel.py:107, which "
                                      "raises for 'nearest' (and for 'area'); 'linear' / 'trilinear' are not 4-D modes")
        if mode not in ("conv", "fc"):
            raise ValueError("Invalid mode. Use 'conv' or 'fc'.")
        self.mode, self.use_footprint = mode, bool(use_footprint)
        self.sample_mode, self.output_size = sample_mode, output_size
        head = ConvDecoder if mode == "conv" else FCDecoder
        # height head gets (hidden, out, leaky) positionally -- for FCDecoder that lands `leaky` in n_blocks (pixel.py:88)
        self.add_module(f"{mode}_decoder", head(hidden_dim, out_dim, leaky))
        if self.use_footprint:
            self.add_module(f"{mode}_decoder_footprint", head(hidden_dim, out_dim))
        self.channels_last = False

    def set_channels_last(self, flag: bool):
        self.channels_last = bool(flag)
        for m in self.modules():
            if isinstance(m, ConvDecoder):
                m.channels_last = bool(flag)

    def _resample(self, x, addend=None):
        if self.sample_mode == "bicubic":
            return ops.upsample_bicubic(x, self.output_size, addend)
        if self.channels_last and x.shape[1] % 4 == 0:
            return grid.upsample_bilinear_cl(x, self.output_size, addend)
        return ops.upsample_bilinear(x, self.output_size, addend)

    def _features(self, planes):
        """Sum of the planes at output resolution (pixel.py:104-110)."""
        xy, image = planes.get("xy"), planes.get("image")
        if xy is None and image is None:
            raise ValueError("PixelwiseDecoder: no 'xy' or 'image' plane given")
        if xy is None or image is None:
            return self._resample(image if xy is None else xy)
        if tuple(image.shape[-2:]) == (self.output_size, self.output_size):
            return self._resample(xy, addend=image)                                 # pixel.py:107+110 fused
        return self._resample(xy) + self._resample(image)

    def forward(self, feature_planes):
        c = self._features(feature_planes)
        heads = [getattr(self, f"{self.mode}_decoder")]
        if self.use_footprint:
            heads.append(getattr(self, f"{self.mode}_decoder_footprint"))
        if self.mode == "conv":
            outs = [h(c).permute(0, 2, 3, 1) for h in heads]                        # [B,1,H,W] -> [B,H,W,1]
        else:
            rows = c.permute(0, 2, 3, 1)
            outs = [h(rows) for h in heads]
        return outs[0], (outs[1] if self.use_footprint else None)
