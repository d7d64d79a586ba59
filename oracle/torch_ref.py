"""TEST INFRASTRUCTURE (oracle) -- torch fp32 restatement of the reference model,
module for module, on the CPU.  It is the checker for the HIP path's
module-level parity tests and the ``cpu_baseline`` ("port") leg of ``bench.py``.
Never imported by the product package.

Pinned against the reference's own Python modules (imported in the build
container with stub IO deps, ``tests/golden/make_golden.py``) through the fixtures in
``tests/golden/``, which ``tests/test_oracle_golden.py`` replays; the only unpinned
piece is the ``scatter_max`` tie-break (see ``oracle/scatter_ref.py``).

Each class keeps the reference's constructor arguments and ``state_dict`` keys
so a reference checkpoint loads with ``load_state_dict(strict=True)``.
File:line citations are relative to /root/reference.
"""
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .scatter_ref import scatter_max, scatter_mean


# --------------------------------------------------------------------------- operators
def coordinate2index(xy: torch.Tensor, reso: int) -> torch.Tensor:
    """utils/coordinate.py:12-28 -- trunc(x*reso) (C cast, not floor), ix + reso*iy, [B,1,N]."""
    cell = (xy * reso).long()
    return (cell[..., 0] + reso * cell[..., 1]).unsqueeze(1)


def rasterise_mean(xy: torch.Tensor, feat: torch.Tensor, reso: int) -> torch.Tensor:
    """pointnet.py:101-111 / alto.py:76-88,187-197: per-cell mean of point features -> [B,C,r,r]."""
    b, _, c = feat.shape
    idx = coordinate2index(xy, reso)
    plane = scatter_mean(feat.permute(0, 2, 1), idx, out=feat.new_zeros(b, c, reso * reso))
    return plane.reshape(b, c, reso, reso)


def sample_bilinear(xy: torch.Tensor, plane: torch.Tensor) -> torch.Tensor:
    """alto.py:90-95,199-205: grid_sample(bilinear, border, align_corners=True) -> [B,N,C]."""
    vgrid = 2.0 * xy[:, :, None].float() - 1.0
    out = F.grid_sample(plane, vgrid, padding_mode="border", align_corners=True, mode="bilinear")
    return out.squeeze(-1).transpose(1, 2)


def pool_local(index: torch.Tensor, feat: torch.Tensor, reso: int, kind: str = "max") -> torch.Tensor:
    """pointnet.py:92-99: scatter_{max,mean} over the cell, gathered back to every point."""
    src = feat.permute(0, 2, 1)
    if kind == "max":
        cellv = scatter_max(src, index, dim_size=reso * reso)[0]
    else:
        cellv = scatter_mean(src, index, dim_size=reso * reso)
    back = cellv.gather(2, index.expand(-1, feat.size(2), -1))
    return back.permute(0, 2, 1)


# --------------------------------------------------------------------------- block/resnet.py
class ResnetBlockFC(nn.Module):
    """block/resnet.py:4-54."""

    def __init__(self, size_in, size_out=None, size_h=None):
        super().__init__()
        size_out = size_in if size_out is None else size_out
        size_h = min(size_in, size_out) if size_h is None else size_h
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        self.shortcut = nn.Linear(size_in, size_out, bias=False) if size_in != size_out else None
        nn.init.zeros_(self.fc_1.weight)

    def forward(self, x):
        h = self.fc_0(F.relu(x))
        dx = self.fc_1(F.relu(h))
        return (x if self.shortcut is None else self.shortcut(x)) + dx


# --------------------------------------------------------------------------- encoder/alto.py
class _AltoDown(nn.Module):
    """alto.py:47-138 (DownConv)."""

    def __init__(self, cin, cout, i, pooling, depth):
        super().__init__()
        self.i, self.depth, self.pooling = i, depth, pooling
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.fc_comm = nn.Sequential(nn.Linear(cout, 2 * cout), nn.ReLU(), nn.Linear(2 * cout, cout))
        self.fc_c = nn.Linear(cin, cout)
        if i > 0:
            self.conv1x1 = nn.Conv2d(cin, cout, 1)

    def forward(self, xy, plane, prev_conv, c_last):
        g = F.relu(self.conv2(F.relu(self.conv1(plane))))
        if prev_conv is not None:
            if 2 <= self.i < self.depth:                       # alto.py:108-110
                g = g + self.conv1x1(F.max_pool2d(prev_conv, 2, 2))
            else:                                              # alto.py:112-114 (i == 1)
                g = g + self.conv1x1(prev_conv)
        after_conv = g
        c = self.fc_comm(sample_bilinear(xy, g))
        if c_last is not None:
            c = c + self.fc_c(c_last)
        raster = rasterise_mean(xy, c, g.shape[2])
        pooled = F.max_pool2d(raster, 2, 2) if self.pooling else raster
        return pooled, raster, after_conv, c


def _upconv2x2(cin, cout, mode="transpose"):
    """alto.py:23-35 / unet.py: transposed convolution, or (mode='upsample') bilinear x2 followed by a 1x1 convolution."""
    if mode == "transpose":
        return nn.ConvTranspose2d(cin, cout, 2, stride=2)
    return nn.Sequential(nn.Upsample(mode="bilinear", scale_factor=2), nn.Conv2d(cin, cout, 1))


class _AltoUp(nn.Module):
    """alto.py:141-257 (UpConv)."""

    def __init__(self, cin, cout, i, depth, merge_mode="concat", up_mode="transpose"):
        super().__init__()
        self.last = i == depth - 2
        self.merge_mode = merge_mode
        self.upconv = _upconv2x2(cin, cout, up_mode)
        if self.last:
            self.upconv_noup = nn.Conv2d(cin, cout, 1)
        self.fc_comm = nn.Sequential(nn.Linear(cout, 2 * cout), nn.ReLU(), nn.Linear(2 * cout, cout))
        self.fc_c = nn.Linear(cin, cout)
        self.conv1x1 = nn.Conv2d(cin, cout, 1) if self.last else _upconv2x2(cin, cout, up_mode)
        self.conv1 = nn.Conv2d(2 * cout if merge_mode == "concat" else cout, cout, 3, padding=1)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)

    def forward(self, xy, from_down, from_up, prev_conv, c_last):
        up = self.upconv_noup(from_up) if self.last else self.upconv(from_up)
        g = torch.cat((up, from_down), 1) if self.merge_mode == "concat" else up + from_down
        g = F.relu(self.conv2(F.relu(self.conv1(g))))
        if prev_conv is not None:
            g = g + self.conv1x1(prev_conv)
        after_conv = g
        if self.last:                                          # alto.py:241-242
            return g, after_conv, c_last
        c = self.fc_comm(sample_bilinear(xy, g))
        if c_last is not None:
            c = c + self.fc_c(c_last)
        return rasterise_mean(xy, c, g.shape[2]), after_conv, c


class AltoUNet(nn.Module):
    """alto.py:260-382 (UNet).  forward(p, {'xy': plane}, c) -> plane."""

    def __init__(self, num_classes, in_channels=3, depth=0, start_filts=64, up_mode="transpose",
                 merge_mode="concat", **kwargs):
        super().__init__()
        if up_mode not in ("transpose", "upsample"):
            raise ValueError(f'"{up_mode}" is not a valid mode for upsampling')
        if merge_mode not in ("concat", "add"):
            raise ValueError(f'"{merge_mode}" is not a valid mode for merging')
        self.depth = depth
        downs, ups = [], []
        outs = in_channels
        for i in range(depth):
            ins = in_channels if i == 0 else outs
            outs = start_filts * (2 ** i)
            downs.append(_AltoDown(ins, outs, i, pooling=not (i == 0 or i == depth - 1), depth=depth))
        for i in range(depth - 1):
            ins, outs = outs, outs // 2
            ups.append(_AltoUp(ins, outs, i, depth, merge_mode=merge_mode, up_mode=up_mode))
        self.down_convs = nn.ModuleList(downs)
        self.up_convs = nn.ModuleList(ups)
        self.conv_final = nn.Conv2d(outs, num_classes, 1)
        for m in self.modules():                               # alto.py:358-366
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight)
                nn.init.constant_(m.bias, 0)

    def forward(self, p, x, c):
        xy = p[..., :2]
        plane = x["xy"]
        skips, prev_conv = [], None
        for down in self.down_convs:
            plane, raster, prev_conv, c = down(xy, plane, prev_conv, c)
            skips.append(raster)
        for i, up in enumerate(self.up_convs):
            plane, prev_conv, c = up(xy, skips[-(i + 2)], plane, prev_conv, c)
        return self.conv_final(plane)


# --------------------------------------------------------------------------- encoder/unet.py
class _PlainDown(nn.Module):
    def __init__(self, cin, cout, pooling=True):
        super().__init__()
        self.pooling = pooling
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)

    def forward(self, x):
        x = F.relu(self.conv2(F.relu(self.conv1(x))))
        return (F.max_pool2d(x, 2, 2) if self.pooling else x), x


class _PlainUp(nn.Module):
    def __init__(self, cin, cout, merge_mode="concat", up_mode="transpose"):
        super().__init__()
        self.merge_mode = merge_mode
        self.upconv = _upconv2x2(cin, cout, up_mode)
        self.conv1 = nn.Conv2d(2 * cout if merge_mode == "concat" else cout, cout, 3, padding=1)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)

    def forward(self, from_down, from_up):
        up = self.upconv(from_up)
        x = torch.cat((up, from_down), 1) if self.merge_mode == "concat" else up + from_down
        return F.relu(self.conv2(F.relu(self.conv1(x))))


class PlainUNet(nn.Module):
    """encoder/unet.py:112-187 (image encoder, ``encoder2: unet``)."""

    def __init__(self, num_classes, in_channels=3, depth=5, start_filts=64, up_mode="transpose",
                 merge_mode="concat", **kwargs):
        super().__init__()
        if up_mode not in ("transpose", "upsample"):
            raise ValueError(f"Invalid up_mode: {up_mode}")
        if up_mode == "upsample" and merge_mode == "add":
            raise ValueError("up_mode 'upsample' is incompatible with merge_mode 'add'.")
        downs, ups = [], []
        outs = in_channels
        for i in range(depth):
            ins = in_channels if i == 0 else outs
            outs = start_filts * (2 ** i)
            downs.append(_PlainDown(ins, outs, pooling=i < depth - 1))
        for _ in range(depth - 1):
            ins, outs = outs, outs // 2
            ups.append(_PlainUp(ins, outs, merge_mode, up_mode))
        self.down_convs = nn.ModuleList(downs)
        self.up_convs = nn.ModuleList(ups)
        self.conv_final = nn.Conv2d(outs, num_classes, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        skips = []
        for down in self.down_convs:
            x, before = down(x)
            skips.append(before)
        for i, up in enumerate(self.up_convs):
            x = up(skips[-(i + 2)], x)
        return self.conv_final(x)


# --------------------------------------------------------------------------- encoder/pointnet.py
class LocalPoolPointnet(nn.Module):
    """encoder/pointnet.py:12-111."""

    def __init__(self, feature_dim=128, dim=3, hidden_dim=128, scatter_type="max", unet_type="alto",
                 unet_kwargs=None, plane_resolution=None, n_blocks=5):
        super().__init__()
        self.c_dim = feature_dim
        self.fc_pos = nn.Linear(dim, 2 * hidden_dim)
        self.blocks = nn.ModuleList([ResnetBlockFC(2 * hidden_dim, hidden_dim) for _ in range(n_blocks)])
        self.fc_c = nn.Linear(hidden_dim, feature_dim)
        self.unet_type = unet_type
        if unet_type == "unet":
            self.unet = PlainUNet(feature_dim, in_channels=feature_dim, **(unet_kwargs or {}))
        elif unet_type == "alto":
            self.unet = AltoUNet(feature_dim, in_channels=feature_dim, **(unet_kwargs or {}))
        else:
            raise ValueError(f"Unknown unet_type: {unet_type}")
        if scatter_type not in ("max", "mean"):
            raise ValueError("Invalid scatter type")
        self.scatter_type = scatter_type
        self.reso_plane = plane_resolution

    def trunk(self, inputs: torch.Tensor):
        """Per-point features before the U-Net (pointnet.py:69-82)."""
        xy = inputs[:, :, :2]
        index = coordinate2index(xy, self.reso_plane)
        net = self.blocks[0](self.fc_pos(inputs))
        for block in self.blocks[1:]:
            pooled = pool_local(index, net, self.reso_plane, self.scatter_type)
            net = block(torch.cat([net, pooled], dim=2))
        return self.fc_c(F.relu(net))

    def forward(self, inputs: torch.Tensor) -> Dict[str, torch.Tensor]:
        net = self.trunk(inputs)
        plane = rasterise_mean(inputs[:, :, :2], net, self.reso_plane)
        if self.unet_type == "unet":
            return {"xy": self.unet(plane)}
        return {"xy": self.unet(inputs, {"xy": plane}, net)}


# --------------------------------------------------------------------------- decoder/pixel.py
class ConvDecoder(nn.Module):
    """decoder/pixel.py:8-32."""

    def __init__(self, in_channels=32, out_channels=1, leaky=False):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, 3, padding=1)
        self.conv2 = nn.Conv2d(64, 128, 3, padding=1)
        self.conv3 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv4 = nn.Conv2d(288, out_channels, 1)
        self.act = F.leaky_relu if leaky else F.relu

    def forward(self, x):
        x1 = self.act(self.conv1(x))
        x2 = self.act(self.conv2(x1))
        x3 = self.act(self.conv3(x2))
        return self.conv4(torch.cat([x, x1, x2, x3], dim=1))


class FCDecoder(nn.Module):
    """decoder/pixel.py:35-58."""

    def __init__(self, in_channels=32, out_channels=1, n_blocks=5, leaky=False):
        super().__init__()
        self.blocks = nn.ModuleList([ResnetBlockFC(in_channels) for _ in range(n_blocks)])
        self.fc_out = nn.Linear(in_channels, out_channels)
        self.act = F.leaky_relu if leaky else F.relu

    def forward(self, x):
        for block in self.blocks:
            x = block(x)
        return self.fc_out(self.act(x))


class PixelwiseDecoder(nn.Module):
    """decoder/pixel.py:61-125.  Note pixel.py:88: ``leaky`` lands in FCDecoder's n_blocks slot."""

    def __init__(self, hidden_dim=32, out_dim=1, output_size=512, leaky=False, sample_mode="bilinear",
                 mode="conv", use_footprint=False, **kwargs):
        super().__init__()
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

    def forward(self, planes: Dict[str, torch.Tensor]):
        c = 0
        if "xy" in planes:
            c = F.interpolate(planes["xy"], size=self.output_size, mode=self.sample_mode, align_corners=True)
        if "image" in planes:
            c = c + F.interpolate(planes["image"], size=self.output_size, mode=self.sample_mode,
                                  align_corners=True)
        foot = None
        if self.mode == "conv":
            x = self.conv_decoder(c).permute(0, 2, 3, 1)
            if self.use_footprint:
                foot = self.conv_decoder_footprint(c).permute(0, 2, 3, 1)
        else:
            c = c.permute(0, 2, 3, 1)
            x = self.fc_decoder(c)
            if self.use_footprint:
                foot = self.fc_decoder_footprint(c)
        return x, foot


# --------------------------------------------------------------------------- model.py
ENCODERS = {"pointnet_local_pool": LocalPoolPointnet, "unet": PlainUNet}


class TomoSAR2Height(nn.Module):
    """tomosar2height/model.py:9-86.  ``cfg`` needs item access and ``.use_cloud/.use_image``."""

    def __init__(self, cfg):
        super().__init__()
        m = cfg["model"]
        self.use_cloud, self.use_image = cfg.use_cloud, cfg.use_image
        if self.use_cloud:
            self.point_encoder = ENCODERS[m["encoder"]](dim=m["data_dim"], **m["encoder_kwargs"])
        if self.use_image:
            self.image_encoder = ENCODERS[m.get("encoder2")](**m.get("encoder2_kwargs", {}))
        self.decoder = PixelwiseDecoder(**m["decoder_pixel_kwargs"])
        zb = cfg["dataset"]["normalize"]["z_bound"]
        self.z_scale = zb[1] - zb[0]
        for mod in self.modules():                             # model.py:46-52
            if isinstance(mod, (nn.Conv2d, nn.Linear)):
                nn.init.xavier_uniform_(mod.weight)
                if mod.bias is not None:
                    nn.init.zeros_(mod.bias)

    def forward(self, input_cloud=None, input_image=None):
        assert self.use_image or self.use_cloud, "At least one input modality must be used."
        planes = {}
        if self.use_cloud:
            planes.update(self.point_encoder(input_cloud))
        if self.use_image:
            planes["image"] = self.image_encoder(input_image)
        pa, pb = self.decoder(planes)
        return pa * self.z_scale, pb


def train_loss(model, cloud, image, dsm, use_footprint=False, weight_ce=10.0):
    """trainer.py:61-69: L1(mean) + weight_ce * BCEWithLogits(mean) on dsm > 1e-4."""
    pa, pb = model(input_cloud=cloud, input_image=image)
    loss = F.l1_loss(pa.squeeze(), dsm.squeeze().float())
    if use_footprint:
        loss = loss + weight_ce * F.binary_cross_entropy_with_logits(
            pb.squeeze(), (dsm.squeeze() > 0.0001).float())
    return loss
