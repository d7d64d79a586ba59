"""CPU: the oracle (C restatement + torch restatement) against the fixtures captured from the
reference's own modules (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import c_oracle, torch_ref
from oracle.scatter_ref import scatter_max, scatter_mean
from detinit import det_init_
from ref_import import make_cfg


def test_coordinate2index_bit_exact():
    g = load_golden("coordinate2index")
    xy = g["xy"]
    for reso in (2, 16, 32, 64, 128, 256):
        want = g[f"index_r{reso}"]
        assert np.array_equal(c_oracle.coordinate2index(xy, reso), want)
        assert np.array_equal(torch_ref.coordinate2index(torch.from_numpy(xy), reso).numpy(), want)


def test_reference_main_vector():
    """The reference's only known-answer check (pointnet.py:114-123)."""
    g = load_golden("pointnet_main_vector")
    xy = g["xy"]
    idx = c_oracle.coordinate2index(xy, 2)
    assert np.array_equal(idx, g["index"])
    assert idx.reshape(-1).tolist() == [0, 2, 1, 3, 0]
    plane = c_oracle.scatter_mean_fwd(xy, idx, 2)
    np.testing.assert_array_equal(plane, g["plane"])
    np.testing.assert_allclose(plane[0, 0], [[0.05, 0.9], [0.3, 0.9]], rtol=1e-6)
    np.testing.assert_allclose(plane[0, 1], [[0.1, 0.3], [0.9, 0.9]], rtol=1e-6)


def test_scatter_max_documented_semantics():
    """pytorch-scatter semantics (parity unpinned): first index wins, untouched -> (0, N)."""
    src = np.array([[[1.], [3.], [3.], [-2.], [-5.]]], np.float32)          # [B=1, N=5, C=1]
    idx = np.array([[0, 0, 0, 2, 2]])
    out, arg = c_oracle.scatter_max(src, idx, 4)
    assert out.reshape(-1).tolist() == [3.0, 0.0, -2.0, 0.0]
    assert arg.reshape(-1).tolist() == [1, 5, 3, 5]
    o2, a2 = scatter_max(torch.from_numpy(src).permute(0, 2, 1), torch.from_numpy(idx)[:, None], dim_size=4)
    assert np.array_equal(o2.numpy(), out) and np.array_equal(a2.numpy(), arg)


@pytest.mark.parametrize("reso", [4, 16])
def test_pool_local(reso):
    g = load_golden("pool_local")
    feat, idx = g[f"feat_r{reso}"], g[f"index_r{reso}"]
    assert np.array_equal(c_oracle.coordinate2index(g[f"xy_r{reso}"], reso), idx)
    pooled, arg = c_oracle.pool_local_fwd(feat, idx, reso * reso)
    np.testing.assert_array_equal(pooled, g[f"out_r{reso}"])
    gfeat = c_oracle.pool_local_bwd(g[f"gout_r{reso}"], idx, arg, reso * reso)
    np.testing.assert_allclose(gfeat, g[f"gfeat_r{reso}"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("reso", [4, 16])
def test_pool_local_mean(reso):
    """scatter_type='mean' (pointnet.py:55-56): the oracle's pool_local against the reference's own call."""
    g = load_golden("pool_local_mean")
    feat = torch.from_numpy(g[f"feat_r{reso}"]).requires_grad_(True)
    idx = torch.from_numpy(g[f"index_r{reso}"])
    out = torch_ref.pool_local(idx, feat, reso, "mean")
    np.testing.assert_array_equal(out.detach().numpy(), g[f"out_r{reso}"])
    out.backward(torch.from_numpy(g[f"gout_r{reso}"]))
    np.testing.assert_allclose(feat.grad.numpy(), g[f"gfeat_r{reso}"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("reso", [4, 16, 32])
def test_scatter_mean_plane(reso):
    g = load_golden("scatter_mean_plane")
    feat = g[f"feat_r{reso}"]
    idx = c_oracle.coordinate2index(g[f"xy_r{reso}"], reso)
    plane = c_oracle.scatter_mean_fwd(feat, idx, reso)
    np.testing.assert_array_equal(plane, g[f"plane_r{reso}"])      # same sequential sum order
    assert reso == 4 or (plane == 0).any()                           # finer fixtures contain empty cells
    gfeat = c_oracle.scatter_mean_bwd(g[f"gout_r{reso}"], idx, feat.shape[1])
    np.testing.assert_allclose(gfeat, g[f"gfeat_r{reso}"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("r", [8, 16, 5])
def test_grid_sample_points(r):
    g = load_golden("grid_sample_points")
    p, plane = g[f"p_r{r}"], g[f"plane_r{r}"]
    out = c_oracle.grid_sample_fwd(plane, p)                         # [B,N,C]
    np.testing.assert_allclose(out, g[f"out_r{r}"].transpose(0, 2, 1), rtol=1e-5, atol=1e-6)
    gplane = c_oracle.grid_sample_bwd(g[f"gout_r{r}"].transpose(0, 2, 1), p, r, r)
    np.testing.assert_allclose(gplane, g[f"gplane_r{r}"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag", ["64_32", "32_32"])
def test_resnet_block_fc(tag):
    g = load_golden("resnet_block_fc")
    ws = g[f"w_{tag}.shortcut.weight"] if f"w_{tag}.shortcut.weight" in g.files else None
    y = c_oracle.resblock_fwd(g[f"x_{tag}"], g[f"w_{tag}.fc_0.weight"], g[f"w_{tag}.fc_0.bias"],
                              g[f"w_{tag}.fc_1.weight"], g[f"w_{tag}.fc_1.bias"], ws)
    np.testing.assert_allclose(y, g[f"y_{tag}"], rtol=1e-5, atol=1e-6)
    cin, cout = (int(v) for v in tag.split("_"))
    blk = torch_ref.ResnetBlockFC(cin, cout)
    blk.load_state_dict({k[len(f"w_{tag}."):]: torch.from_numpy(g[k]) for k in g.files
                         if k.startswith(f"w_{tag}.")}, strict=True)
    x = torch.from_numpy(g[f"x_{tag}"]).requires_grad_(True)
    yy = blk(x)
    yy.backward(torch.from_numpy(g[f"gy_{tag}"]))
    np.testing.assert_allclose(yy.detach().numpy(), g[f"y_{tag}"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(x.grad.numpy(), g[f"gx_{tag}"], rtol=1e-5, atol=1e-6)
    for k, v in blk.named_parameters():
        np.testing.assert_allclose(v.grad.numpy(), g[f"g_{tag}.{k}"], rtol=1e-5, atol=1e-5)


def test_upsample_bilinear_matches_torch():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 16, 16, generator=g)
    want = torch.nn.functional.interpolate(x, size=32, mode="bilinear", align_corners=True)
    got = c_oracle.upsample_bilinear_fwd(x.numpy(), 32)
    np.testing.assert_allclose(got, want.numpy(), rtol=1e-5, atol=1e-6)
    xg = x.clone().requires_grad_(True)
    go = torch.randn(2, 3, 32, 32, generator=g)
    torch.nn.functional.interpolate(xg, size=32, mode="bilinear", align_corners=True).backward(go)
    np.testing.assert_allclose(c_oracle.upsample_bilinear_bwd(go.numpy(), 16, 16), xg.grad.numpy(),
                               rtol=1e-5, atol=1e-5)


def test_local_pool_pointnet_reduced():
    g = load_golden("local_pool_pointnet_reduced")
    enc = torch_ref.LocalPoolPointnet(feature_dim=8, dim=3, hidden_dim=8, scatter_type="max", unet_type="alto",
                                      unet_kwargs=dict(depth=3, merge_mode="concat", start_filts=8),
                                      plane_resolution=16)
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    assert list(sd.keys()) == list(enc.state_dict().keys())
    enc.load_state_dict(sd, strict=True)
    out = enc(torch.from_numpy(g["cloud"]))["xy"]
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-6, atol=1e-6)
    out.backward(torch.from_numpy(g["gout"]))
    none_grad = [k for k, v in enc.named_parameters() if v.grad is None]
    assert none_grad == g["none_grad"].tolist()
    assert len(none_grad) == 8          # up_convs[depth-2].{upconv,fc_comm,fc_c}: SURVEY a9
    for k, v in enc.named_parameters():
        if v.grad is not None:
            np.testing.assert_allclose(v.grad.numpy(), g["g." + k], rtol=1e-4, atol=1e-5)


def test_local_pool_pointnet_reduced_mean():
    g = load_golden("local_pool_pointnet_reduced_mean")
    enc = torch_ref.LocalPoolPointnet(feature_dim=8, dim=3, hidden_dim=8, scatter_type="mean", unet_type="alto",
                                      unet_kwargs=dict(depth=3, merge_mode="concat", start_filts=8),
                                      plane_resolution=16)
    enc.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}, strict=True)
    out = enc(torch.from_numpy(g["cloud"]))["xy"]
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-6, atol=1e-6)
    out.backward(torch.from_numpy(g["gout"]))
    assert [k for k, v in enc.named_parameters() if v.grad is None] == g["none_grad"].tolist()
    for k, v in enc.named_parameters():
        if v.grad is not None:
            np.testing.assert_allclose(v.grad.numpy(), g["g." + k], rtol=1e-4, atol=1e-5)


def test_up_mode_upsample():
    """up_mode='upsample' (alto.py:23-35, unet.py: bilinear x2 + 1x1 conv instead of the transposed convolution)."""
    g = load_golden("local_pool_pointnet_reduced_upsample")
    enc = torch_ref.LocalPoolPointnet(feature_dim=8, dim=3, hidden_dim=8, scatter_type="max", unet_type="alto",
                                      unet_kwargs=dict(depth=3, merge_mode="concat", start_filts=8, up_mode="upsample"),
                                      plane_resolution=16)
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    assert list(sd.keys()) == list(enc.state_dict().keys())
    enc.load_state_dict(sd, strict=True)
    out = enc(torch.from_numpy(g["cloud"]))["xy"]
    np.testing.assert_allclose(out.detach().numpy(), g["out"], rtol=1e-6, atol=1e-6)
    out.backward(torch.from_numpy(g["gout"]))
    assert [k for k, v in enc.named_parameters() if v.grad is None] == g["none_grad"].tolist()
    for k, v in enc.named_parameters():
        if v.grad is not None:
            np.testing.assert_allclose(v.grad.numpy(), g["g." + k], rtol=1e-4, atol=1e-5)
    g = load_golden("plain_unet_upsample")
    net = torch_ref.PlainUNet(8, in_channels=4, depth=3, start_filts=8, up_mode="upsample")
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    assert list(sd.keys()) == list(net.state_dict().keys())
    net.load_state_dict(sd, strict=True)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = net(x)
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-6, atol=1e-6)
    y.backward(torch.from_numpy(g["gy"]))
    np.testing.assert_allclose(x.grad.numpy(), g["gx"], rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        torch_ref.PlainUNet(8, up_mode="upsample", merge_mode="add")


@pytest.mark.parametrize("mode", ["conv", "fc"])
@pytest.mark.parametrize("foot", [False, True])
@pytest.mark.parametrize("img", [False, True])
def test_pixelwise_decoder(mode, foot, img):
    g = load_golden("pixelwise_decoder")
    tag = f"{mode}_f{int(foot)}_i{int(img)}"
    dec = det_init_(torch_ref.PixelwiseDecoder(hidden_dim=32, out_dim=1, output_size=32, mode=mode,
                                               use_footprint=foot), seed=7)
    assert list(dec.state_dict().keys()) == g[f"keys_{tag}"].tolist()
    planes = {"xy": torch.from_numpy(g[f"xy_{tag}"]).requires_grad_(True)}
    if img:
        planes["image"] = torch.from_numpy(g[f"image_{tag}"])
    x, xf = dec(planes)
    np.testing.assert_allclose(x.detach().numpy(), g[f"x_{tag}"], rtol=1e-5, atol=1e-6)
    loss = x.sum()
    if foot:
        np.testing.assert_allclose(xf.detach().numpy(), g[f"xf_{tag}"], rtol=1e-5, atol=1e-6)
        loss = loss + 0.5 * xf.sum()
    else:
        assert xf is None
    loss.backward()
    np.testing.assert_allclose(planes["xy"].grad.numpy(), g[f"gxy_{tag}"], rtol=1e-4, atol=1e-5)


def test_full_model_berlin_n4096():
    g = load_golden("full_model_berlin_n4096")
    model = det_init_(torch_ref.TomoSAR2Height(make_cfg(depth=5)), seed=8)
    assert sum(p.numel() for p in model.parameters()) == int(g["n_params"]) == 10930881
    assert list(model.state_dict().keys()) == g["state_keys"].tolist()
    cloud = torch.from_numpy(g["cloud"])
    dsm = torch.from_numpy(g["dsm_lo"]).repeat_interleave(8, 1).repeat_interleave(8, 2)
    loss = torch_ref.train_loss(model, cloud, None, dsm)
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-6)
    with torch.no_grad():
        pa, pb = model(input_cloud=cloud)
    assert pb is None
    np.testing.assert_allclose(pa[0, :, :, 0].numpy(), g["height"], rtol=1e-5, atol=1e-4)
    loss.backward()
    grads = dict(model.named_parameters())
    assert [k for k, v in grads.items() if v.grad is None] == g["none_grad"].tolist()
    for k, n, s in zip(g["grad_names"].tolist(), g["grad_norm"], g["grad_sum"]):
        np.testing.assert_allclose(grads[k].grad.double().norm().item(), n, rtol=1e-4, atol=1e-9)


def test_trainer_accumulation_semantics():
    """trainer.py:47-89: grads of `optimize_every` tiles are SUMMED (no 1/k), then one AdamW step."""
    g = load_golden("trainer_accumulation")
    model = det_init_(torch_ref.TomoSAR2Height(make_cfg(depth=3, reso=16, hidden=32, start_filts=8)), seed=9)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    opt.zero_grad()
    losses = []
    for t in range(3):
        dsm = torch.from_numpy(g[f"dsm_lo_{t}"]).repeat_interleave(8, 0).repeat_interleave(8, 1)
        loss = torch_ref.train_loss(model, torch.from_numpy(g[f"cloud_{t}"]), None, dsm[None])
        loss.backward()
        losses.append(loss.item())
    opt.step()
    np.testing.assert_allclose(np.mean(losses), float(g["last_avg_loss"]), rtol=1e-6)
    params = dict(model.named_parameters())
    for k in g.files:
        if k.startswith("after."):
            name = k[len("after."):]
            np.testing.assert_allclose(params[name].detach().numpy(), g[k], rtol=1e-6, atol=1e-7)
            assert not np.array_equal(g[k], g["before." + name]) or "up_convs.0.upconv" not in name


def _check_module_fixture(g, prefix, mod, inputs, rtol=1e-4, atol=1e-5):
    sd = {k[len(prefix) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix + ".w.")}
    assert list(sd.keys()) == list(mod.state_dict().keys())
    mod.load_state_dict(sd, strict=True)
    args = [torch.from_numpy(g[f"{prefix}.in.{k}"]).requires_grad_(k.startswith("x")) for k in inputs]
    out = mod(*args)
    out = out["xy"] if isinstance(out, dict) else out
    np.testing.assert_allclose(out.detach().numpy(), g[prefix + ".out"], rtol=1e-6, atol=1e-6)
    out.backward(torch.from_numpy(g[prefix + ".gout"]))
    assert [k for k, v in mod.named_parameters() if v.grad is None] == g[prefix + ".none_grad"].tolist()
    for k, v in mod.named_parameters():
        if v.grad is not None:
            np.testing.assert_allclose(v.grad.numpy(), g[f"{prefix}.g.{k}"], rtol=rtol, atol=atol)
    for k, a in zip(inputs, args):
        if a.grad is not None:
            np.testing.assert_allclose(a.grad.numpy(), g[f"{prefix}.gin.{k}"], rtol=rtol, atol=atol)


def test_reference_options():
    """r06: constructor options no shipped config selects -- ConvDecoder(leaky=True), unet_type='unet', merge_mode='add' in
    both U-Nets -- the oracle's restatement against the reference's own outputs and gradients (reference_options.npz)."""
    g = load_golden("reference_options")

    class _Dec(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.d = torch_ref.PixelwiseDecoder(hidden_dim=32, out_dim=1, output_size=32, mode="conv", leaky=True)

        def forward(self, x):
            return self.d({"xy": x})[0]
    _check_module_fixture(g, "leaky_decoder", _Dec(), ["x"])
    _check_module_fixture(g, "plane_unet", torch_ref.LocalPoolPointnet(
        feature_dim=8, dim=3, hidden_dim=8, scatter_type="max", unet_type="unet",
        unet_kwargs=dict(depth=3, merge_mode="concat", start_filts=8), plane_resolution=16), ["cloud"])
    _check_module_fixture(g, "alto_add", torch_ref.LocalPoolPointnet(
        feature_dim=8, dim=3, hidden_dim=8, scatter_type="max", unet_type="alto",
        unet_kwargs=dict(depth=3, merge_mode="add", start_filts=8), plane_resolution=16), ["cloud"])
    _check_module_fixture(g, "unet_add", torch_ref.PlainUNet(8, in_channels=4, depth=3, start_filts=8, merge_mode="add"), ["x"])


def test_sample_modes_fixture_is_what_torch_computes():
    """r06: sample_modes.npz (generated through the reference's DownConv.sample_plane_feature and PixelwiseDecoder with
    sample_mode='bicubic' / 'nearest') against the same torch calls made here, and the oracle's decoder with the mode set."""
    g = load_golden("sample_modes")
    for mode in ("bicubic", "nearest"):
        c = torch.from_numpy(g[f"{mode}.plane"]).requires_grad_(True)
        vgrid = 2.0 * torch.from_numpy(g[f"{mode}.pts"])[..., :2][:, :, None] - 1.0
        out = torch.nn.functional.grid_sample(c, vgrid, padding_mode="border", align_corners=True, mode=mode).squeeze(-1)
        np.testing.assert_allclose(out.detach().numpy(), g[f"{mode}.out"], rtol=1e-6, atol=1e-6)
        out.backward(torch.from_numpy(g[f"{mode}.gout"]))
        np.testing.assert_allclose(c.grad.numpy(), g[f"{mode}.gplane"], rtol=1e-5, atol=1e-5)
    for size in (40, 32):
        dec = torch_ref.PixelwiseDecoder(hidden_dim=32, out_dim=1, output_size=size, mode="conv", sample_mode="bicubic")
        dec.load_state_dict({k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("dec.w.")}, strict=True)
        xy = torch.from_numpy(g[f"dec{size}.xy"]).requires_grad_(True)
        x, _ = dec({"xy": xy, "image": torch.from_numpy(g[f"dec{size}.image"])})
        np.testing.assert_allclose(x.detach().numpy(), g[f"dec{size}.x"], rtol=1e-5, atol=1e-5)
        x.backward(torch.from_numpy(g[f"dec{size}.gx"]))
        np.testing.assert_allclose(xy.grad.numpy(), g[f"dec{size}.gxy"], rtol=1e-4, atol=1e-5)
