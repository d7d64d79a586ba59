"""ALTO U-Net (reference: tomosar2height/encoder/alto.py) on the MI355X path.

Every level alternates topology: grid convs (implicit-GEMM HIP kernels of csrc/conv.hip in channels_last mode, the
default; MIOpen through PyTorch-ROCm otherwise) -> bilinear sample to the points -> per-point MLP -> mean-rasterise back
to the grid.
The three point<->grid steps run as HIP kernels on the tile's cell-sorted order
(``ops.sample_plane`` / ``mlp.comm_mlp`` / ``ops.rasterise_mean``); the reference recomputes cell indices
and clones the coordinates at every level (alto.py:79-80, 92-94, 189-190), here the ``TileIndex`` built
once per forward is shared by all of them.

Constructor arguments, sub-module names and therefore ``state_dict`` keys are the reference's
(``down_convs.{i}.{conv1,conv2,fc_comm.0,fc_comm.2,fc_c,conv1x1}``, ``up_convs.{i}.{upconv,upconv_noup,
fc_comm.0,fc_comm.2,fc_c,conv1x1,conv1,conv2}``, ``conv_final``).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init

from .. import deferred, grid, mlp, ops
from ..tile import TileIndex


def conv3x3(in_channels, out_channels):
    return nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)


def conv1x1(in_channels, out_channels):
    return nn.Conv2d(in_channels, out_channels, kernel_size=1)


def upconv2x2(in_channels, out_channels, mode="transpose"):
    """alto.py:23-35: a 2x2 stride-2 transposed convolution, or (mode='upsample', no shipped config) bilinear x2 + 1x1 conv
    with the reference's ``nn.Sequential`` parameter names (``.1.weight`` / ``.1.bias``)."""
    if mode == "transpose":
        return nn.ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2)
    return nn.Sequential(nn.Upsample(mode="bilinear", scale_factor=2), conv1x1(in_channels, out_channels))


def _comm_layers(channels):
    return nn.Sequential(nn.Linear(channels, 2 * channels), nn.ReLU(), nn.Linear(2 * channels, channels))


class _PointGridLevel(nn.Module):
    """Shared point<->grid step of DownConv / UpConv: sample -> fc_comm (+ fc_c) -> rasterise."""

    channels_last = False

    def _conv_pair(self, x):
        """``F.relu(conv2(F.relu(conv1(x))))`` (alto.py:98-99, 226-227); in channels_last mode on the implicit-GEMM
        kernels with fused bias / ReLU / ReLU-backward (grid.py, csrc/conv.hip)."""
        if self.channels_last:
            return grid.conv3x3_chain(x, (self.conv1, self.conv2))
        return F.relu(self.conv2(F.relu(self.conv1(x))))

    def _pool(self, x):
        return grid.maxpool2x2(x, self.pool) if self.channels_last else self.pool(x)

    def _conv1x1(self, conv, x, addend=None):
        """``[addend +] conv(x)`` for 1x1 convs and 2x2 transposed convs (UpConv.conv1x1 is either, alto.py:172-175); in
        channels_last mode the residual add rides the kernel's epilogue."""
        if not self.channels_last:
            return conv(x) if addend is None else addend + conv(x)
        if isinstance(conv, nn.ConvTranspose2d):
            return grid.upconv2x2(x, conv, addend)
        if isinstance(conv, nn.Sequential):                       # upconv2x2(mode='upsample')
            return grid.upsample_conv1x1(x, conv, addend)
        return grid.conv1x1(x, conv, addend)

    def _exchange(self, tile: TileIndex, plane: torch.Tensor, c_last, later_res=None):
        """-> (raster, c, plane): the returned ``plane`` is the input plane for its further consumers (its gradient is then
        summed inside the sample backward kernel, ops.sample_plane_thru).  ``later_res``: (plane resolutions, channel counts) of
        this and every later point<->grid exchange of the U-Net; with it the wide levels run in the deferred form
        (deferred.py): ``c`` is then a ``deferred.Deferred`` -- the per-point features as a linear map of the hidden
        activations, never materialised."""
        fa, fb = self.fc_comm[0], self.fc_comm[2]
        r, ch = plane.shape[2], plane.shape[1]
        if self.sample_mode != "bilinear":
            # r06: 'bicubic' / 'nearest' (alto.py:95,204; no shipped config, and the reference's own U-Net never passes the
            # argument on -- only a directly constructed level can carry it): the plain sample -> MLP -> rasterise on the
            # mode's own kernels (csrc/bicubic.hip); the re-associated forms below rest on the bilinear kernels
            if isinstance(c_last, deferred.Deferred):
                raise NotImplementedError("a level with sample_mode != 'bilinear' after a deferred level: construct the whole "
                                          "U-Net's levels with the same mode (UNet.forward_sorted then defers nothing)")
            sampled = ops.sample_plane_mode(tile, plane, self.sample_mode)
            c = mlp.comm_mlp(sampled, fa.weight, fa.bias, fb.weight, fb.bias, c_last, self.fc_c.weight, self.fc_c.bias)
            raster, c = ops.rasterise_mean_thru(tile, c, plane.shape[2], self.channels_last)
            return raster, c, plane
        if isinstance(c_last, deferred.Deferred) or (later_res is not None and torch.is_tensor(c_last)
                                                     and deferred.applicable(tile, r, ch)):
            if isinstance(c_last, deferred.Deferred):
                state = c_last
            else:
                res, chs = later_res[0], later_res[1]
                cache = later_res[2] if len(later_res) > 2 else None
                state = deferred.Deferred(tile, [tile.level(x) for x in res], chs, c_last, cache=cache)
            # fc_comm.0 on the pixels (alto.py:123); the returned plane is the input plane for its further consumers
            q, plane = mlp.linear_plane_thru(plane, fa.weight, fa.bias)
            raster = state.advance(q, r, fb, self.fc_c).reshape(plane.shape[0], r, r, ch).permute(0, 3, 1, 2)
            return (raster if self.channels_last else raster.contiguous()), state, plane
        if mlp.grid_first_applicable(tile, r, ch):
            # coarse levels (many points per pixel): fc_comm.0 on the pixels, its 2C-wide result interpolated straight into
            # the hidden activations -- see mlp._CommMLPGridFirst
            c, plane = mlp.comm_mlp_grid_first(tile, plane, fa.weight, fa.bias, fb.weight, fb.bias, c_last,
                                               self.fc_c.weight, self.fc_c.bias)
        else:
            sampled, plane = ops.sample_plane_thru(tile, plane)                   # alto.py:121-122 / 245-246
            c = mlp.comm_mlp(sampled, fa.weight, fa.bias, fb.weight, fb.bias, c_last,
                             self.fc_c.weight, self.fc_c.bias)                   # alto.py:123-128 / 248-253
        # alto.py:130 / 255; `c` also feeds the next level's fc_c: its two gradients are summed in the rasterisation's backward
        raster, c = ops.rasterise_mean_thru(tile, c, plane.shape[2], self.channels_last)
        return raster, c, plane


class DownConv(_PointGridLevel):
    """alto.py:47-138."""

    def __init__(self, in_channels, out_channels, i, pooling, depth, sample_mode="bilinear"):
        super().__init__()
        if sample_mode not in ("bilinear", "bicubic", "nearest"):         # (the three modes F.grid_sample takes for 4-D input)
            raise ValueError(f"sample_mode={sample_mode!r}: F.grid_sample (alto.py:95,204) accepts 'bilinear', 'nearest', 'bicubic'")
        self.sample_mode = sample_mode
        self.in_channels, self.out_channels = in_channels, out_channels
        self.pooling, self.downsample, self.depth = pooling, i, depth
        self.conv1 = conv3x3(in_channels, out_channels)
        self.conv2 = conv3x3(out_channels, out_channels)
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.fc_comm = _comm_layers(out_channels)
        self.fc_c = nn.Linear(in_channels, out_channels)
        if i > 0:
            self.conv1x1 = conv1x1(in_channels, out_channels)

    def forward(self, tile: TileIndex, grid_in, prev_conv=None, c_last=None, later_res=None):
        g = self._conv_pair(grid_in)
        if prev_conv is not None:
            # alto.py:104-114: levels 2..depth-1 see the pooled previous conv output, level 1 the unpooled one
            res_in = self._pool(prev_conv) if 2 <= self.downsample < self.depth else prev_conv
            g = self._conv1x1(self.conv1x1, res_in, addend=g)
        raster, c, g = self._exchange(tile, g, c_last, later_res)
        if self.pooling and self.channels_last:         # raster is also the skip connection: one fused gradient sum
            pooled, raster = grid.maxpool2x2_thru(raster, self.pool)
        else:
            pooled = self._pool(raster) if self.pooling else raster
        return pooled, raster, g, c


class UpConv(_PointGridLevel):
    """alto.py:141-257."""

    def __init__(self, in_channels, out_channels, i, depth, merge_mode="concat", up_mode="transpose",
                 sample_mode="bilinear"):
        super().__init__()
        if sample_mode not in ("bilinear", "bicubic", "nearest"):         # (the three modes F.grid_sample takes for 4-D input)
            raise ValueError(f"sample_mode={sample_mode!r}: F.grid_sample (alto.py:95,204) accepts 'bilinear', 'nearest', 'bicubic'")
        self.sample_mode = sample_mode
        self.in_channels, self.out_channels = in_channels, out_channels
        self.merge_mode, self.up_mode, self.depth = merge_mode, up_mode, depth
        self.is_last = i == depth - 2
        self.upconv = upconv2x2(in_channels, out_channels, up_mode)
        if self.is_last:
            self.upconv_noup = conv1x1(in_channels, out_channels)
        self.fc_comm = _comm_layers(out_channels)
        self.fc_c = nn.Linear(in_channels, out_channels)
        self.conv1x1 = conv1x1(in_channels, out_channels) if self.is_last else upconv2x2(in_channels, out_channels, up_mode)
        self.conv1 = conv3x3(2 * out_channels if merge_mode == "concat" else out_channels, out_channels)
        self.conv2 = conv3x3(out_channels, out_channels)

    def forward(self, tile: TileIndex, from_down, from_up, prev_conv, c_last, later_res=None):
        up = self._conv1x1(self.upconv_noup if self.is_last else self.upconv, from_up)            # alto.py:215-218
        g = torch.cat((up, from_down), 1) if self.merge_mode == "concat" else up + from_down
        g = self._conv_pair(g)
        if prev_conv is not None:
            g = self._conv1x1(self.conv1x1, prev_conv, addend=g)                    # alto.py:233-236
        if self.is_last:                                                            # alto.py:241-242
            return g, g, c_last
        raster, c, g = self._exchange(tile, g, c_last, later_res)
        return raster, g, c


class UNet(nn.Module):
    """alto.py:260-382.  ``forward(p, x, c)`` keeps the reference signature: ``p [B,N,3]`` points,
    ``x = {'xy': plane}``, ``c [B,N,C]`` point features.  The encoder calls ``forward_sorted`` with the
    tile it already built; the public signature builds one on the fly."""

    def __init__(self, num_classes, in_channels=3, depth=0, start_filts=64, up_mode="transpose",
                 merge_mode="concat", **kwargs):
        super().__init__()
        if up_mode not in ("transpose", "upsample"):
            raise ValueError(f'"{up_mode}" is not a valid mode for upsampling. Only "transpose" and "upsample" are allowed.')
        if merge_mode not in ("concat", "add"):
            raise ValueError(f'"{merge_mode}" is not a valid mode for merging up and down paths. '
                             'Only "concat" and "add" are allowed.')
        if up_mode == "upsample" and merge_mode == "add":
            raise ValueError('up_mode "upsample" is incompatible with merge_mode "add"')
        self.num_classes, self.in_channels = num_classes, in_channels
        self.start_filts, self.depth = start_filts, depth
        self.up_mode, self.merge_mode = up_mode, merge_mode

        downs, ups = [], []
        outs = in_channels
        for i in range(depth):
            ins = in_channels if i == 0 else outs
            outs = start_filts * (2 ** i)
            downs.append(DownConv(ins, outs, i, pooling=0 < i < depth - 1, depth=depth))
        for i in range(depth - 1):
            ins, outs = outs, outs // 2
            ups.append(UpConv(ins, outs, i, depth=depth, up_mode=up_mode, merge_mode=merge_mode))
        self.down_convs = nn.ModuleList(downs)
        self.up_convs = nn.ModuleList(ups)
        self.conv_final = conv1x1(outs, num_classes)
        self.reset_params()

    @staticmethod
    def weight_init(m):
        if isinstance(m, nn.Conv2d):
            init.xavier_normal_(m.weight)
            init.constant_(m.bias, 0)

    def reset_params(self):
        for m in self.modules():
            self.weight_init(m)

    def set_channels_last(self, flag: bool):
        for m in self.modules():
            if isinstance(m, _PointGridLevel):
                m.channels_last = bool(flag)

    def forward_sorted(self, tile: TileIndex, plane: torch.Tensor, c_sorted: torch.Tensor) -> torch.Tensor:
        skips, prev_conv, c = [], None, c_sorted
        # plane resolution of every point<->grid exchange, in order (the deferred form needs to know where the hidden
        # activations of a level will be rasterised later): down levels, then every up level but the last (alto.py:241-242)
        res, chs, r = [], [], plane.shape[2]
        for down in self.down_convs:
            res.append(r)
            chs.append(down.out_channels)
            r = r // 2 if down.pooling else r
        for up in self.up_convs:
            if not up.is_last:
                r *= 2
                res.append(r)
                chs.append(up.out_channels)
        pos = 0
        cache = getattr(self, "compose_cache", None)        # set by the Trainer (deferred.ComposeCache), else plain autograd
        # (levels with another sample_mode -- swapped in by hand, the constructor cannot set it -- switch the deferred form off)
        plain = any(getattr(m, "sample_mode", "bilinear") != "bilinear" for m in list(self.down_convs) + list(self.up_convs))
        later = (lambda: None) if plain else (lambda: (res[pos:], chs[pos:], cache))
        for down in self.down_convs:
            plane, raster, prev_conv, c = down(tile, plane, prev_conv, c, later())
            skips.append(raster)
            pos += 1
        for i, up in enumerate(self.up_convs):
            plane, prev_conv, c = up(tile, skips[-(i + 2)], plane, prev_conv, c, later())
            pos += 1
        if self.down_convs[0].channels_last:
            return grid.conv1x1(plane, self.conv_final)
        return self.conv_final(plane)

    def forward(self, p, x, c):
        tile = TileIndex(p, x["xy"].shape[2])
        return self.forward_sorted(tile, x["xy"], tile.sort_rows(c))
