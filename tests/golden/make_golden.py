#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Every array below is produced by calling the reference's own Python functions /
modules (imported where they lie, see ``ref_import.py``); the only stand-in is
``torch_scatter`` (``oracle/scatter_ref.py``, parity unpinned for the
scatter_max tie-break).  Fixtures are data only: inputs, weights and expected
outputs.  SURVEY.md section 8c lists what each fixture pins.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ref_import import import_reference, make_cfg  # noqa: E402
from detinit import det_init_, synth_cloud  # noqa: E402


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


def edge_xy(m, seed):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(1, m, 2, generator=g)
    edge = torch.tensor([2.0 ** -20, 1 - 2.0 ** -24, 0.5, 0.25, 0.75, 1 / 256, 255 / 256, 1 / 16, 15 / 16,
                         0.49999997, 0.99999, 3 / 128, 1e-6])
    k = edge.numel()
    xy[0, :k, 0] = edge
    xy[0, :k, 1] = edge.flip(0)
    xy[0, k:2 * k, 0] = edge
    xy[0, k:2 * k, 1] = 0.3
    return xy.float()


def main():
    ref = import_reference()
    from utils.coordinate import coordinate2index
    from tomosar2height.encoder.pointnet import LocalPoolPointnet
    from tomosar2height.encoder.alto import DownConv
    from tomosar2height.block import ResnetBlockFC
    from tomosar2height.decoder.pixel import PixelwiseDecoder
    import trainer as ref_trainer

    torch.manual_seed(0)

    # (1) coordinate2index -- utils/coordinate.py:12-28
    xy = edge_xy(200, 1)
    arrs = {"xy": xy}
    for reso in (2, 16, 32, 64, 128, 256):
        arrs[f"index_r{reso}"] = coordinate2index(xy, reso)
    save("coordinate2index", **arrs)

    # (3b) the reference's own known-answer vector -- pointnet.py:114-123
    from torch_scatter import scatter_mean
    pxy = torch.tensor([[[0., 0.], [0.3, 0.9], [0.9, 0.3], [0.9, 0.9], [0.1, 0.2]]])
    idx = coordinate2index(pxy, 2)
    plane = scatter_mean(pxy.permute(0, 2, 1), idx, out=pxy.new_zeros(1, 2, 4)).reshape(1, 2, 2, 2)
    save("pointnet_main_vector", xy=pxy, index=idx, plane=plane)

    # (2) pool_local fwd + grad -- pointnet.py:92-99, with ties and an all-equal cell
    enc = LocalPoolPointnet(feature_dim=8, dim=3, hidden_dim=8, scatter_type="max", unet_type="alto",
                            unet_kwargs=dict(depth=2, merge_mode="concat", start_filts=8), plane_resolution=16)
    arrs = {}
    for reso in (4, 16):
        enc.reso_plane = reso
        cloud = synth_cloud(300, seed=10 + reso)
        g = torch.Generator().manual_seed(reso)
        feat = torch.randn(1, 300, 8, generator=g)
        feat = (feat * 4).round() / 4          # quantised -> many exact ties
        idx = coordinate2index(cloud[:, :, :2], reso)
        cell0 = idx[0, 0] == idx[0, 0, 0]
        feat[0, cell0] = 1.25                  # one all-equal cell: argmax = first point
        feat.requires_grad_(True)
        out = enc.pool_local(idx, feat)
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        arrs.update({f"xy_r{reso}": cloud[:, :, :2], f"feat_r{reso}": feat, f"index_r{reso}": idx,
                     f"out_r{reso}": out, f"gout_r{reso}": gout, f"gfeat_r{reso}": feat.grad})
    save("pool_local", **arrs)

    # (3) generate_plane_features fwd + grad -- pointnet.py:101-111, incl. empty cells
    arrs = {}
    for reso, c in ((4, 8), (16, 8), (32, 12)):
        enc.reso_plane, enc.c_dim = reso, c
        cloud = synth_cloud(257, seed=20 + reso)
        g = torch.Generator().manual_seed(100 + reso)
        feat = torch.randn(1, 257, c, generator=g, requires_grad=True)
        idx = coordinate2index(cloud[:, :, :2], reso)
        plane = enc.generate_plane_features({"xy": idx}, feat, "xy")
        gout = torch.randn(plane.shape, generator=g)
        plane.backward(gout)
        arrs.update({f"xy_r{reso}": cloud[:, :, :2], f"feat_r{reso}": feat, f"plane_r{reso}": plane,
                     f"gout_r{reso}": gout, f"gfeat_r{reso}": feat.grad})
    save("scatter_mean_plane", **arrs)

    # (4) sample_plane_feature fwd + plane grad -- alto.py:90-95
    down = DownConv(4, 4, 0, False, depth=2)
    arrs = {}
    for r, c in ((8, 4), (16, 8), (5, 3)):
        g = torch.Generator().manual_seed(200 + r)
        plane = torch.randn(1, c, r, r, generator=g, requires_grad=True)
        p = synth_cloud(150, seed=30 + r)
        corners = torch.tensor([[0., 0.], [1., 0.], [0., 1.], [1., 1.], [.5, .5], [1 - 2.0 ** -24, 2.0 ** -20],
                                [1 / (r - 1), 2 / (r - 1)], [0.999999, 0.5]])
        p[0, :corners.shape[0], :2] = corners
        out = down.sample_plane_feature(p, plane)           # [1, C, N]
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        arrs.update({f"p_r{r}": p, f"plane_r{r}": plane, f"out_r{r}": out, f"gout_r{r}": gout,
                     f"gplane_r{r}": plane.grad})
    save("grid_sample_points", **arrs)

    # (5) ResnetBlockFC 64->32 and 32->32 -- block/resnet.py
    arrs = {}
    for cin, cout in ((64, 32), (32, 32)):
        blk = det_init_(ResnetBlockFC(cin, cout), seed=5)
        g = torch.Generator().manual_seed(cin)
        x = torch.randn(50, cin, generator=g, requires_grad=True)
        y = blk(x)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        tag = f"{cin}_{cout}"
        arrs.update({f"x_{tag}": x, f"y_{tag}": y, f"gy_{tag}": gy, f"gx_{tag}": x.grad})
        for k, v in blk.state_dict().items():
            arrs[f"w_{tag}.{k}"] = v
        for k, v in blk.named_parameters():
            arrs[f"g_{tag}.{k}"] = v.grad
    save("resnet_block_fc", **arrs)

    # (6) LocalPoolPointnet reduced -- pointnet.py:60-90 + alto.py
    enc = LocalPoolPointnet(feature_dim=8, dim=3, hidden_dim=8, scatter_type="max", unet_type="alto",
                            unet_kwargs=dict(depth=3, merge_mode="concat", start_filts=8), plane_resolution=16)
    det_init_(enc, seed=6)
    cloud = synth_cloud(256, seed=40)
    out = enc(cloud)["xy"]
    g = torch.Generator().manual_seed(6)
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)
    arrs = {"cloud": cloud, "out": out, "gout": gout}
    none_grad = []
    for k, v in enc.state_dict().items():
        arrs["w." + k] = v
    for k, v in enc.named_parameters():
        if v.grad is None:
            none_grad.append(k)
        else:
            arrs["g." + k] = v.grad
    arrs["none_grad"] = np.array(none_grad)
    save("local_pool_pointnet_reduced", **arrs)

    # (7) PixelwiseDecoder small -- pixel.py:94-125
    arrs = {}
    for mode in ("conv", "fc"):
        for foot in (False, True):
            for img in (False, True):
                dec = det_init_(PixelwiseDecoder(hidden_dim=32, out_dim=1, output_size=32, mode=mode,
                                                 use_footprint=foot), seed=7)
                g = torch.Generator().manual_seed(7)
                planes = {"xy": torch.randn(1, 32, 16, 16, generator=g, requires_grad=True)}
                if img:
                    planes["image"] = torch.randn(1, 32, 32, 32, generator=g)
                xy_plane = planes["xy"]
                # the reference's `c += planes['xy']` aliases nothing: c starts as int 0
                x, xf = dec(dict(planes))
                loss = x.sum() + (xf.sum() * 0.5 if xf is not None else 0)
                loss.backward()
                tag = f"{mode}_f{int(foot)}_i{int(img)}"
                arrs[f"xy_{tag}"] = xy_plane
                if img:
                    arrs[f"image_{tag}"] = planes["image"]
                arrs[f"x_{tag}"] = x
                if xf is not None:
                    arrs[f"xf_{tag}"] = xf
                arrs[f"gxy_{tag}"] = xy_plane.grad
                arrs[f"keys_{tag}"] = np.array(list(dec.state_dict().keys()))
    save("pixelwise_decoder", **arrs)

    # (8) full-size models, N = 4096, name-keyed deterministic weights (tests/detinit.py)
    for tag, kw in (("berlin", dict(depth=5)),
                    ("munich", dict(depth=6, use_image=True, use_footprint=True, z_bound=(465.5, 599.5)))):
        cfg = make_cfg(**kw)
        model = det_init_(ref.TomoSAR2Height(cfg), seed=8)
        cloud = synth_cloud(4096, seed=50)
        g = torch.Generator().manual_seed(8)
        image = torch.randn(1, 3, 512, 512, generator=g) if kw.get("use_image") else None
        dsm_lo = torch.rand(1, 64, 64, generator=g) * 30       # stored; dsm = 8x8 block replicate
        dsm = dsm_lo.repeat_interleave(8, 1).repeat_interleave(8, 2)
        pa, pb = model(input_cloud=cloud, input_image=image)
        loss = torch.nn.functional.l1_loss(pa.squeeze(), dsm.squeeze())
        if pb is not None:
            loss = loss + 10.0 * torch.nn.functional.binary_cross_entropy_with_logits(
                pb.squeeze(), (dsm.squeeze() > 0.0001).float())
        loss.backward()
        names, gnorm, gsum, none_grad = [], [], [], []
        for k, v in model.named_parameters():
            if v.grad is None:
                none_grad.append(k)
                continue
            names.append(k)
            gnorm.append(v.grad.double().norm().item())
            gsum.append(v.grad.double().sum().item())
        arrs = dict(cloud=cloud, dsm_lo=dsm_lo, height=pa[0, :, :, 0], loss=loss.detach(),
                    grad_names=np.array(names), grad_norm=np.array(gnorm), grad_sum=np.array(gsum),
                    none_grad=np.array(none_grad), n_params=sum(p.numel() for p in model.parameters()),
                    state_keys=np.array(list(model.state_dict().keys())))
        if image is not None:
            arrs["image_seed"] = 8          # regenerate: randn(1,3,512,512) from Generator(8), first draw
        if pb is not None:
            arrs["footprint_logits"] = pb[0, :, :, 0]
        save(f"full_model_{tag}_n4096", **arrs)

    # (9) Trainer.train_step accumulation semantics -- trainer.py:47-89
    cfg = make_cfg(depth=3, reso=16, hidden=32, start_filts=8)
    model = det_init_(ref.TomoSAR2Height(cfg), seed=9)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)       # train.py:97
    tr = ref_trainer.Trainer(model, opt, device=torch.device("cpu"), optimize_every=3, use_cloud=True)
    arrs = {}
    track = ["point_encoder.fc_pos.weight", "point_encoder.unet.down_convs.1.fc_comm.0.weight",
             "decoder.conv_decoder.conv4.weight", "point_encoder.unet.up_convs.0.upconv.bias"]
    before = {k: v.detach().clone() for k, v in model.named_parameters() if k in track}
    for t in range(3):
        cloud = synth_cloud(200, seed=60 + t)
        g = torch.Generator().manual_seed(60 + t)
        dsm_lo = torch.rand(64, 64, generator=g) * 30          # stored; dsm = 8x8 block replicate
        dsm = dsm_lo.repeat_interleave(8, 0).repeat_interleave(8, 1)
        # DataLoader collate adds the batch dim: inputs [1,N,3], dsm [1,512,512]
        tr.train_step({"inputs": cloud, "dsm": dsm[None]})
        arrs[f"cloud_{t}"] = cloud
        arrs[f"dsm_lo_{t}"] = dsm_lo
    for k, v in model.named_parameters():
        if k in track:
            arrs["before." + k] = before[k]
            arrs["after." + k] = v
    arrs["last_avg_loss"] = tr.last_avg_loss
    save("trainer_accumulation", **arrs)




def mean_pool_fixture():
    """scatter_type='mean' (pointnet.py:55-56, 92-99; no shipped config selects it): pool_local fwd + grad at two
    resolutions and the reduced LocalPoolPointnet of fixture (6) with mean pooling."""
    import_reference()
    from utils.coordinate import coordinate2index
    from tomosar2height.encoder.pointnet import LocalPoolPointnet
    enc = LocalPoolPointnet(feature_dim=8, dim=3, hidden_dim=8, scatter_type="mean", unet_type="alto",
                            unet_kwargs=dict(depth=2, merge_mode="concat", start_filts=8), plane_resolution=16)
    arrs = {}
    for reso in (4, 16):
        enc.reso_plane = reso
        cloud = synth_cloud(300, seed=60 + reso)
        g = torch.Generator().manual_seed(60 + reso)
        feat = torch.randn(1, 300, 8, generator=g, requires_grad=True)
        idx = coordinate2index(cloud[:, :, :2], reso)
        out = enc.pool_local(idx, feat)
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        arrs.update({f"xy_r{reso}": cloud[:, :, :2], f"feat_r{reso}": feat, f"index_r{reso}": idx,
                     f"out_r{reso}": out, f"gout_r{reso}": gout, f"gfeat_r{reso}": feat.grad})
    save("pool_local_mean", **arrs)

    enc = LocalPoolPointnet(feature_dim=8, dim=3, hidden_dim=8, scatter_type="mean", unet_type="alto",
                            unet_kwargs=dict(depth=3, merge_mode="concat", start_filts=8), plane_resolution=16)
    det_init_(enc, seed=6)
    cloud = synth_cloud(256, seed=40)
    out = enc(cloud)["xy"]
    g = torch.Generator().manual_seed(6)
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)
    arrs = {"cloud": cloud, "out": out, "gout": gout}
    none_grad = []
    for k, v in enc.state_dict().items():
        arrs["w." + k] = v
    for k, v in enc.named_parameters():
        if v.grad is None:
            none_grad.append(k)
        else:
            arrs["g." + k] = v.grad
    arrs["none_grad"] = np.array(none_grad)
    save("local_pool_pointnet_reduced_mean", **arrs)


def upsample_mode_fixture():
    """up_mode='upsample' (alto.py:23-35, unet.py; no shipped config selects it): the reduced LocalPoolPointnet of fixture
    (6) with the ALTO U-Net's non-parametric up path, and a small plain image U-Net."""
    import_reference()
    from tomosar2height.encoder.pointnet import LocalPoolPointnet
    from tomosar2height.encoder.unet import UNet
    enc = LocalPoolPointnet(feature_dim=8, dim=3, hidden_dim=8, scatter_type="max", unet_type="alto",
                            unet_kwargs=dict(depth=3, merge_mode="concat", start_filts=8, up_mode="upsample"),
                            plane_resolution=16)
    det_init_(enc, seed=6)
    cloud = synth_cloud(256, seed=40)
    out = enc(cloud)["xy"]
    g = torch.Generator().manual_seed(6)
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)
    arrs = {"cloud": cloud, "out": out, "gout": gout}
    none_grad = []
    for k, v in enc.state_dict().items():
        arrs["w." + k] = v
    for k, v in enc.named_parameters():
        if v.grad is None:
            none_grad.append(k)
        else:
            arrs["g." + k] = v.grad
    arrs["none_grad"] = np.array(none_grad)
    save("local_pool_pointnet_reduced_upsample", **arrs)

    net = det_init_(UNet(8, in_channels=4, depth=3, start_filts=8, up_mode="upsample", merge_mode="concat"), seed=9)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 4, 16, 16, generator=g, requires_grad=True)
    y = net(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    arrs = {"x": x, "y": y, "gy": gy, "gx": x.grad}
    for k, v in net.state_dict().items():
        arrs["w." + k] = v
    for k, v in net.named_parameters():
        arrs["g." + k] = v.grad
    save("plain_unet_upsample", **arrs)


def _module_fixture(mod, inputs, name_prefix, arrs, seed):
    """Forward + backward of a reference module on `inputs` (dict of tensors -> positional args in order): outputs, the upstream
    gradient, every parameter (state_dict order) and every parameter gradient / the grad-is-None set, under `name_prefix`."""
    args = [v.clone().requires_grad_(v.dtype.is_floating_point and k.startswith("x")) for k, v in inputs.items()]
    out = mod(*args)
    out0 = out["xy"] if isinstance(out, dict) else (out[0] if isinstance(out, (tuple, list)) else out)
    g = torch.Generator().manual_seed(seed)
    gout = torch.randn(out0.shape, generator=g)
    out0.backward(gout)
    for (k, v), a in zip(inputs.items(), args):
        arrs[f"{name_prefix}.in.{k}"] = v
        if a.grad is not None:
            arrs[f"{name_prefix}.gin.{k}"] = a.grad
    arrs[f"{name_prefix}.out"] = out0
    arrs[f"{name_prefix}.gout"] = gout
    none_grad = []
    for k, v in mod.state_dict().items():
        arrs[f"{name_prefix}.w.{k}"] = v
    for k, v in mod.named_parameters():
        if v.grad is None:
            none_grad.append(k)
        else:
            arrs[f"{name_prefix}.g.{k}"] = v.grad
    arrs[f"{name_prefix}.none_grad"] = np.array(none_grad, dtype=str)


def options_fixture():
    """r06: constructor options of the hot path's modules that no shipped config selects but the reference accepts --
    ``ConvDecoder(leaky=True)`` (pixel.py:8-32), ``LocalPoolPointnet(unet_type='unet')`` (pointnet.py:45-49: the plain U-Net on
    the rasterised plane), ``merge_mode='add'`` in the ALTO U-Net (alto.py:176-179, 221-224) and in the plain U-Net
    (unet.py:92-105).  Each: inputs, weights, output, upstream gradient, every gradient, produced by the reference's modules."""
    import_reference()
    from tomosar2height.encoder.pointnet import LocalPoolPointnet
    from tomosar2height.encoder.unet import UNet
    from tomosar2height.decoder.pixel import PixelwiseDecoder
    arrs = {}
    g = torch.Generator().manual_seed(21)
    dec = det_init_(PixelwiseDecoder(hidden_dim=32, out_dim=1, output_size=32, mode="conv", leaky=True), seed=21)

    class _Dec(torch.nn.Module):                     # (dict in, tuple out -> tensor in, tensor out; parameters under the same names)
        def __init__(self, d):
            super().__init__()
            self.d = d

        def forward(self, x):
            return self.d({"xy": x})[0]
    _module_fixture(_Dec(dec), {"x": torch.randn(1, 32, 16, 16, generator=g)}, "leaky_decoder", arrs, 22)
    enc = det_init_(LocalPoolPointnet(feature_dim=8, dim=3, hidden_dim=8, scatter_type="max", unet_type="unet",
                                      unet_kwargs=dict(depth=3, merge_mode="concat", start_filts=8), plane_resolution=16), seed=23)
    _module_fixture(enc, {"cloud": synth_cloud(256, seed=41)}, "plane_unet", arrs, 24)
    enc = det_init_(LocalPoolPointnet(feature_dim=8, dim=3, hidden_dim=8, scatter_type="max", unet_type="alto",
                                      unet_kwargs=dict(depth=3, merge_mode="add", start_filts=8), plane_resolution=16), seed=25)
    _module_fixture(enc, {"cloud": synth_cloud(256, seed=42)}, "alto_add", arrs, 26)
    net = det_init_(UNet(8, in_channels=4, depth=3, start_filts=8, up_mode="transpose", merge_mode="add"), seed=27)
    _module_fixture(net, {"x": torch.randn(2, 4, 16, 16, generator=g)}, "unet_add", arrs, 28)
    save("reference_options", **arrs)

    # sample_mode = 'bicubic' / 'nearest' (alto.py:51,95; pixel.py:75,107,110): the reference's own sampling method of a level and
    # its decoder with the mode set.  Points include the plane's corners, edges and exact pixel centres.
    from tomosar2height.encoder.alto import DownConv
    arrs = {}
    pts = torch.rand(2, 300, 3, generator=g)
    pts[0, :6, :2] = torch.tensor([[0., 0.], [1., 1.], [1., 0.], [0.5, 0.5], [1 / 15, 14 / 15], [0.999999, 1e-7]])
    for mode in ("bicubic", "nearest"):
        lvl = DownConv(8, 8, 0, False, depth=3, sample_mode=mode)
        c = torch.randn(2, 8, 16, 16, generator=g, requires_grad=True)
        out = lvl.sample_plane_feature(pts, c)                       # [B, C, N]
        gout = torch.randn(out.shape, generator=g)
        out.backward(gout)
        arrs.update({f"{mode}.pts": pts, f"{mode}.plane": c, f"{mode}.out": out, f"{mode}.gout": gout, f"{mode}.gplane": c.grad})
    for size in (40, 32):                                            # 16 -> 40 (scale 15/39) and 16 -> 32 (15/31), with an image plane
        dec = det_init_(PixelwiseDecoder(hidden_dim=32, out_dim=1, output_size=size, mode="conv", sample_mode="bicubic"), seed=33)
        xy = torch.randn(1, 32, 16, 16, generator=g, requires_grad=True)
        img = torch.randn(1, 32, size, size, generator=g)
        x, _ = dec({"xy": xy, "image": img})
        gx = torch.randn(x.shape, generator=g)
        x.backward(gx)
        arrs.update({f"dec{size}.xy": xy, f"dec{size}.image": img, f"dec{size}.x": x, f"dec{size}.gx": gx, f"dec{size}.gxy": xy.grad})
        if size == 40:                                               # (same seed: the two decoders share their weights)
            for k, v in dec.state_dict().items():
                arrs[f"dec.w.{k}"] = v
        # the resampling alone (what ops.interpolate replaces), with its adjoint
        xi = torch.randn(2, 5, 16, 16, generator=g, requires_grad=True)
        yi = torch.nn.functional.interpolate(xi, size=size, mode="bicubic", align_corners=True)
        gi = torch.randn(yi.shape, generator=g)
        yi.backward(gi)
        arrs.update({f"interp{size}.x": xi, f"interp{size}.y": yi, f"interp{size}.gy": gi, f"interp{size}.gx": xi.grad})
    save("sample_modes", **arrs)


def blend_weight_fixture():
    """(10) DSMGenerator._linear_blend_patch_weight (generator.py:85-113).  generator.py cannot be imported as a
    module here (dataset.py needs `transformations`), so the static method is compiled from the reference file where
    it lies and executed; nothing is copied."""
    import ast
    import math
    src = open("/root/reference/generator.py").read()
    tree = ast.parse(src)
    fn = next(n for cls in tree.body if isinstance(cls, ast.ClassDef) and cls.name == "DSMGenerator"
              for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "_linear_blend_patch_weight")
    fn.decorator_list = []
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"torch": torch, "math": math}
    exec(compile(mod, "/root/reference/generator.py", "exec"), ns)
    arrs = {}
    for shape, pct in (((64, 64), (0.5, 0.5)), ((32, 48), (0.25, 0.5)), ((16, 16), (0.0, 0.0)), ((10, 7), (0.3, 0.2))):
        w = ns["_linear_blend_patch_weight"](shape, list(pct))
        assert w.dtype == torch.float64
        arrs[f"w_{shape[0]}x{shape[1]}_{pct[0]}_{pct[1]}"] = w
    w = ns["_linear_blend_patch_weight"]((512, 512), [0.5, 0.5])      # the shipped configuration: store its two ramps
    arrs["w512_row0"], arrs["w512_col0"], arrs["w512_centre"] = w[0, :], w[:, 0], w[255:257, 255:257]
    save("mosaic_blend_weight", **arrs)




def tile_producer_fixture():
    """(11) the point half of TomoSARDataset.__getitem__ (dataset.py:229-278, default config) composed from the
    reference's OWN utility functions (utils/crop_cloud.py:crop_pc_2d, utils/coordinate.py:apply_transform /
    invert_transform); dataset.py itself cannot be imported (needs `transformations`, rasterio data)."""
    import_reference()
    from utils import crop_pc_2d, invert_transform, apply_transform
    g = torch.Generator().manual_seed(11)
    p = 6000
    chunk = torch.stack([386000.0 + torch.rand(p, generator=g, dtype=torch.float64) * 1500.0,
                         5820000.0 + torch.rand(p, generator=g, dtype=torch.float64) * 1500.0,
                         30.0 + torch.rand(p, generator=g, dtype=torch.float64) * 60.0], 1)
    anchors = torch.tensor([[386100.0, 5820200.0], [386700.25, 5820900.5], [385900.0, 5819900.0], [390000.0, 5830000.0]],
                           dtype=torch.float64)
    chunk[0, :2] = anchors[0]                                         # exactly on the window corner: excluded (strict)
    chunk[1, 0], chunk[1, 1] = anchors[0, 0] + 512.0, anchors[0, 1] + 100.0   # on the right edge: excluded
    chunk[2, :2] = anchors[0] + 1e-9                                  # just inside
    patch_size = torch.tensor([512.0, 512.0], dtype=torch.float64)
    z_bound = [-33.7, 156.5]
    scale_mat = torch.diag(torch.tensor([512.0, 512.0, z_bound[1] - z_bound[0], 1], dtype=torch.float64))   # dataset.py:187-190
    shift_norm = torch.cat([torch.eye(4, 3, dtype=torch.float64), torch.tensor([0.5, 0.5, 0, 1]).reshape(-1, 1)], 1)
    arrs = {"chunk": chunk, "anchors": anchors}
    for i, anchor in enumerate(anchors):
        min_bound, max_bound = anchor, anchor + patch_size
        inputs, index = crop_pc_2d(chunk, min_bound, max_bound)                                  # dataset.py:234
        arrs[f"n_{i}"] = len(inputs)
        if len(inputs) == 0:
            continue
        z_shift = torch.min(inputs[:, 2]).double().reshape(1)                                    # dataset.py:246
        transform_mat = scale_mat.clone()
        transform_mat[0:3, 3] = torch.cat([(min_bound + max_bound) / 2., z_shift], 0)            # dataset.py:268
        normalize_mat = shift_norm.double() @ torch.eye(4, dtype=torch.float64) @ torch.eye(4, dtype=torch.float64) \
            @ invert_transform(transform_mat).double()                                           # :269-270, no augmentation
        inputs_norm = apply_transform(inputs, normalize_mat).float()                             # :275-276
        inputs_norm, index2 = crop_pc_2d(inputs_norm, [0.0, 0.0], [1.0, 1.0])                    # :278
        arrs[f"index_{i}"] = index[index2]
        arrs[f"inputs_{i}"] = inputs_norm
        arrs[f"zshift_{i}"] = z_shift
    save("tile_producer", **arrs)


def _rotation_z(angle):
    """`transformations.rotation_matrix(angle, [0, 0, 1])` (dataset.py:30-35; the package is absent here, so its published
    formula is restated: R = cos I + (1 - cos) n n^T + sin [n]x with math.sin / math.cos -- cos(-pi/2) is 6e-17, not 0)."""
    import math
    sina, cosa = math.sin(angle), math.cos(angle)
    m = torch.eye(4, dtype=torch.float64)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = cosa, -sina, sina, cosa
    return m


def _reflection(axis):
    """`transformations.reflection_matrix(origin, axis)` = I - 2 n n^T for a plane through the origin (dataset.py:38-42)."""
    m = torch.eye(4, dtype=torch.float64)
    m[axis, axis] = -1.0
    return m


def tile_producer_aug_fixture():
    """(12) TomoSARDataset.__getitem__ with augmentation (dataset.py:253-328): points through the reference's own
    crop_pc_2d / invert_transform / apply_transform with flip_mat @ rot_mat for every (rot_times, flip_dim); the DSM and
    image patches through the torch indexing expressions of dataset.py:296-328 (slice, rot90(k, [-1, -2]), flip, .float(),
    .flip(-2))."""
    import math
    import_reference()
    from utils import crop_pc_2d, invert_transform, apply_transform
    g = torch.Generator().manual_seed(12)
    p = 1500
    chunk = torch.stack([1000.0 + torch.rand(p, generator=g, dtype=torch.float64) * 900.0,
                         2000.0 + torch.rand(p, generator=g, dtype=torch.float64) * 900.0,
                         10.0 + torch.rand(p, generator=g, dtype=torch.float64) * 50.0], 1)
    anchor = torch.tensor([1200.0, 2150.0], dtype=torch.float64)
    patch_size = torch.tensor([512.0, 512.0], dtype=torch.float64)
    z_bound = [-33.7, 156.5]
    scale_mat = torch.diag(torch.tensor([512.0, 512.0, z_bound[1] - z_bound[0], 1], dtype=torch.float64))
    shift_norm = torch.cat([torch.eye(4, 3, dtype=torch.float64), torch.tensor([0.5, 0.5, 0, 1]).reshape(-1, 1)], 1)
    rot_mat_dic = {k: (torch.eye(4).double() if k == 0 else _rotation_z(-90.0 * k * math.pi / 180.0)) for k in range(4)}
    flip_mat_dic = {-1: torch.eye(4).double(), 0: _reflection(0), 1: _reflection(1)}
    dsm_data = torch.rand(40, 48, generator=g) * 30.0                                   # float32 raster (dataset.py:133)
    image = (torch.randint(0, 2048, (3, 40, 48), generator=g).double() - 500.0) / 300.0   # (int - mean) / std, float64 (:113)
    row, col, shape = 25, 7, (16, 16)
    arrs = {"chunk": chunk, "anchor": anchor, "dsm_data": dsm_data, "image": image, "row_col_shape": np.array([row, col, 16, 16])}
    min_bound, max_bound = anchor, anchor + patch_size
    inputs, index = crop_pc_2d(chunk, min_bound, max_bound)
    z_shift = torch.min(inputs[:, 2]).double().reshape(1)
    for rot_times in range(4):
        for flip_dim in (-1, 0, 1):
            tag = f"r{rot_times}_f{flip_dim + 1}"
            transform_mat = scale_mat.clone()
            transform_mat[0:3, 3] = torch.cat([(min_bound + max_bound) / 2., z_shift], 0)
            normalize_mat = shift_norm.double() @ flip_mat_dic[flip_dim].double() @ rot_mat_dic[rot_times].double() \
                @ invert_transform(transform_mat).double()                                       # dataset.py:268-269
            inputs_norm = apply_transform(inputs, normalize_mat).float()
            inputs_norm, index2 = crop_pc_2d(inputs_norm, [0.0, 0.0], [1.0, 1.0])
            arrs[f"index_{tag}"] = index[index2]
            arrs[f"inputs_{tag}"] = inputs_norm
            for name, src in (("dsm", dsm_data[None]), ("image", image)):
                t = src[:, row - shape[0] + 1:row + 1, col:col + shape[1]]                        # dataset.py:299-300, 316
                if rot_times > 0:
                    t = t.rot90(rot_times, [-1, -2])
                if flip_dim == 0:
                    t = t.flip(-1)
                if flip_dim == 1:
                    t = t.flip(-2)
                arrs[f"{name}_{tag}"] = t.float().flip(-2)                                        # :310, :328
    save("tile_producer_aug", **arrs)


if __name__ == "__main__":
    if "--only-aug" in sys.argv:
        tile_producer_aug_fixture()
        sys.exit(0)
    if "--only-mean" in sys.argv:
        mean_pool_fixture()
        sys.exit(0)
    if "--only-upsample" in sys.argv:
        upsample_mode_fixture()
        sys.exit(0)
    if "--only-options" in sys.argv:
        options_fixture()
        sys.exit(0)
    if "--only-blend" not in sys.argv and "--only-producer" not in sys.argv:
        main()
    if "--only-producer" not in sys.argv:
        blend_weight_fixture()
    if "--only-blend" not in sys.argv:
        tile_producer_fixture()
        tile_producer_aug_fixture()
    if "--only-blend" not in sys.argv and "--only-producer" not in sys.argv:
        mean_pool_fixture()
        upsample_mode_fixture()
        options_fixture()
