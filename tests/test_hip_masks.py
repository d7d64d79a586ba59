"""-m gpu: mask-pinned gradient checks above operator level (the counterpart of
test_hip_trunk.py::test_fused_backward_vs_float64_with_the_same_masks for the ALTO levels and the fused decoder).

The gradient of a ReLU / max-pool network is piecewise constant in its activations, so module-level comparisons with the
oracle have to tolerate mask flips (1e-2 max-norm / 3e-3 L2 in test_hip_model.py).  Here the float64 restatement of one
level uses the HIP path's OWN ReLU masks and max-pool winners (read off the HIP forward, whose launches are bit-reproducible),
so no mask can flip and what remains is the arithmetic of the kernels themselves -- conv pair with fused bias / ReLU /
ReLU-backward epilogues, residual 1x1 / transposed conv with the add in its epilogue, bilinear sample and its "thru"
backward, fc_comm + fc_c (`_CommMLP`), mean rasterisation with the joined gradient, 2x2 max-pool with the joined skip
gradient; the decoder's bilinear upsample, three 3x3 convs and the concat-free 288 -> 1 head -- pinned at 1e-5
(max-normalised per tensor).

Reference semantics restated in float64: alto.py:97-138 (DownConv.forward), alto.py:207-257 (UpConv.forward),
pixel.py:8-32 + 94-125 (ConvDecoder, PixelwiseDecoder.forward)."""
import pytest
import torch
import torch.nn.functional as F

from detinit import det_init_, synth_cloud

pytestmark = pytest.mark.gpu

TOL = 1e-5
D = torch.float64


def _dev():
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(_dev())


def _cl(x):
    return x.contiguous(memory_format=torch.channels_last)


class _Checks:
    """Collects every comparison of a test, prints them all, then asserts -- so one run shows the whole picture."""

    def __init__(self):
        self.rows = []

    def add(self, name, got, want, tol=TOL):
        got, want = got.detach().to(D), want.detach().to(D)
        assert got.shape == want.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
        self.rows.append((((got - want).abs().max() / (want.abs().max() + 1e-300)).item(), tol, name))

    def finish(self, title):
        print(f"[{title}] max-normalised errors vs float64 with the same masks:")
        for err, tol, name in sorted(self.rows, reverse=True):
            print(f"    {err:.2e} (tol {tol:g}) {name}")
        bad = [f"{name}: {err:.2e} > {tol:g}" for err, tol, name in self.rows if not err <= tol]
        assert not bad, "; ".join(bad)


# ------------------------------------------------------------------------------------------------ float64 pieces
def _p64(module):
    """float64 leaf copies of a module's parameters, keyed like named_parameters()."""
    return {k: v.detach().to(D).requires_grad_(True) for k, v in module.named_parameters()}


def _conv64(x, p, name, padding):
    return F.conv2d(x, p[name + ".weight"], p[name + ".bias"], padding=padding)


def _pool64(x64, x32):
    """2x2 max-pool of the float64 plane with the winners of the fp32 plane (ATen's first-maximum tie-break, which
    t2h_maxpool2x2_nhwc_fwd reproduces)."""
    _, idx = F.max_pool2d(x32.detach().contiguous(), 2, 2, return_indices=True)
    b, c, h, w = idx.shape
    return x64.flatten(2).gather(2, idx.flatten(2)).view(b, c, h, w)


def _sample64(plane64, pts):
    """alto.py:90-95: grid_sample(plane, 2 xy - 1, bilinear, border, align_corners=True) -> [N, C]."""
    xy = pts[:, :2].to(D)
    vgrid = (2.0 * xy - 1.0)[None, :, None, :]
    out = F.grid_sample(plane64, vgrid, mode="bilinear", padding_mode="border", align_corners=True)     # [1, C, N, 1]
    return out[0, :, :, 0].t()


def _raster64(c64, pts, reso):
    """alto.py:76-88: scatter_mean of the point rows into the plane (empty cell = 0)."""
    xy = pts[:, :2]
    ix, iy = (xy[:, 0] * reso).long(), (xy[:, 1] * reso).long()                      # coordinate.py:12-28, trunc == floor here
    idx = ix + reso * iy
    n, ch = c64.shape
    sums = torch.zeros(reso * reso, ch, dtype=D, device=c64.device).index_add_(0, idx, c64)
    cnt = torch.zeros(reso * reso, dtype=D, device=c64.device).index_add_(0, idx, torch.ones(n, dtype=D, device=c64.device))
    return (sums / cnt.clamp_min(1.0)[:, None]).t().reshape(1, ch, reso, reso)


def _comm64(sampled, c_last, p, mask_h):
    h = (sampled @ p["fc_comm.0.weight"].t() + p["fc_comm.0.bias"]) * mask_h
    return h @ p["fc_comm.2.weight"].t() + p["fc_comm.2.bias"] + c_last @ p["fc_c.weight"].t() + p["fc_c.bias"]


def _hidden_mask(level, tile, plane):
    """ReLU mask of fc_comm's hidden layer from the same launches the level's forward issues: point-first
    (`_CommMLP.forward`: sample, then fc_comm.0 on the [N, C] rows) or grid-first (`_CommMLPGridFirst.forward`: fc_comm.0 on
    the pixels, then the sample) -- whichever `_exchange` takes for this tile.  The float64 restatement is the reference's
    order (alto.py:121-123) in both cases, so the grid-first levels are checked against the point-first formulation."""
    from tomosar2height_amd import mlp, ops
    fa = level.fc_comm[0]
    r, c = plane.shape[2], plane.shape[1]
    if mlp.grid_first_applicable(tile, r, c):
        rows = ops.to_nhwc(plane).reshape(-1, c)
        h = mlp.hidden_from_plane(tile, rows, r, fa.weight.detach(), fa.bias.detach())
    else:
        sampled = ops.sample_plane(tile, plane)
        h = torch.empty(sampled.shape[0], fa.weight.shape[0], dtype=torch.float32, device=sampled.device)
        mlp.linear_fwd_(sampled.contiguous(), fa.weight.detach(), fa.bias.detach(), h, relu_out=True)
    return (h > 0).to(D)


def _run64(fn):
    """The float64 restatement runs on the device (ATen's native double kernels); a ROCm build without a double
    convolution falls back to the host."""
    try:
        return fn(_dev())
    except RuntimeError as e:                    # pragma: no cover - depends on the ROCm build
        print(f"[test_hip_masks] float64 restatement on the host ({str(e)[:80]})")
        return fn(torch.device("cpu"))


# ------------------------------------------------------------------------------------------------ DownConv
@pytest.mark.parametrize("n_points", [20000, 131072])
def test_down_level_gradients_vs_float64_with_the_same_masks(n_points):
    """DownConv i = 2 of the Berlin ALTO U-Net (64 -> 128 channels at 128^2, pooled residual from the 256^2 level, pooling
    on): all four outputs and every gradient (parameters, the two input planes, the incoming point features).  N = 20000 takes
    the point-first exchange, the benchmarked N = 131072 (8 points per pixel) the grid-first one."""
    from tomosar2height_amd import grid, ops
    from tomosar2height_amd.encoder.alto import DownConv
    from tomosar2height_amd.tile import TileIndex
    dev = _dev()
    level = det_init_(DownConv(64, 128, 2, pooling=True, depth=5), seed=41).to(dev)
    level.channels_last = True
    tile = TileIndex(synth_cloud(n_points, seed=5).to(dev), 256)
    x = _cl(_rand((1, 64, 128, 128), 1)).requires_grad_(True)
    prev = _cl(_rand((1, 64, 256, 256), 2)).requires_grad_(True)
    c_last = _rand((n_points, 64), 3).requires_grad_(True)
    pooled, raster, g, c = level(tile, x, prev, c_last)
    ups = [_rand(tuple(t.shape), 10 + i) for i, t in enumerate((pooled, raster, g, c))]
    sum((o * u).sum() for o, u in zip((pooled, raster, g, c), ups)).backward()

    # masks, from the same kernels on the same inputs (bit-reproducible launches); the asserts prove they are the module's
    with torch.no_grad():
        y1 = grid.conv_bias_act(x.detach(), level.conv1, relu=True)
        y2 = grid.conv_bias_act(y1, level.conv2, relu=True)
        res_in = grid.maxpool2x2(prev.detach(), level.pool)
        g_step = grid.conv1x1(res_in, level.conv1x1, y2)
        assert torch.equal(g_step, g.detach()), "stepwise forward differs from the module's: masks would not be the module's"
        mask_h = _hidden_mask(level, tile, g_step)
        m1, m2 = (y1 > 0), (y2 > 0)

    def ref(dev64):
        p = {k: v.detach().to(dev64).requires_grad_(True) for k, v in _p64(level).items()}
        to = lambda t: t.detach().to(dev64)                                             # noqa: E731
        x64, prev64, cl64 = (to(t).to(D).requires_grad_(True) for t in (x, prev, c_last))
        pts = to(tile.pts)
        a1 = _conv64(x64, p, "conv1", 1) * to(m1).to(D)
        a2 = _conv64(a1, p, "conv2", 1) * to(m2).to(D)
        g64 = a2 + _conv64(_pool64(prev64, to(prev)), p, "conv1x1", 0)                  # alto.py:108-114 (i >= 2: pooled)
        c64 = _comm64(_sample64(g64, pts), cl64, p, to(mask_h))                         # alto.py:121-128
        raster64 = _raster64(c64, pts, 128)                                             # alto.py:130
        pooled64 = _pool64(raster64, to(raster))                                        # alto.py:135-136
        outs = (pooled64, raster64, g64, c64)
        sum((o * to(u).to(D)).sum() for o, u in zip(outs, ups)).backward()
        return outs, p, (x64, prev64, cl64)

    outs64, p, ins64 = _run64(ref)
    ck = _Checks()
    for name, got, want in zip(("pooled", "raster", "g", "c"), (pooled, raster, g, c), outs64):
        ck.add("out " + name, got.cpu(), want.cpu())
    for name, got, want in zip(("d grid_in", "d prev_conv", "d c_last"), (x, prev, c_last), ins64):
        ck.add(name, got.grad.cpu(), want.grad.cpu())
    for k, v in level.named_parameters():
        ck.add("d " + k, v.grad.cpu(), p[k].grad.cpu())
    ck.finish(f"DownConv 64->128 @128^2, N={n_points}")


# ------------------------------------------------------------------------------------------------ UpConv
@pytest.mark.parametrize("n_points", [30000, 131072])
def test_up_level_gradients_vs_float64_with_the_same_masks(n_points):
    """UpConv i = 1 of the Berlin ALTO U-Net (256 -> 128 channels, 64^2 -> 128^2: transposed-conv upsampling, concat with the
    skip, conv pair, transposed-conv residual of the previous level, then the point<->grid exchange -- point-first at
    N = 30000, grid-first at the benchmarked N = 131072)."""
    from tomosar2height_amd import grid, ops
    from tomosar2height_amd.encoder.alto import UpConv
    from tomosar2height_amd.tile import TileIndex
    dev = _dev()
    level = det_init_(UpConv(256, 128, 1, depth=5), seed=43).to(dev)
    level.channels_last = True
    tile = TileIndex(synth_cloud(n_points, seed=6).to(dev), 256)
    from_down = _cl(_rand((1, 128, 128, 128), 1)).requires_grad_(True)
    from_up = _cl(_rand((1, 256, 64, 64), 2)).requires_grad_(True)
    prev = _cl(_rand((1, 256, 64, 64), 3)).requires_grad_(True)
    c_last = _rand((n_points, 256), 4).requires_grad_(True)
    raster, g, c = level(tile, from_down, from_up, prev, c_last)
    ups = [_rand(tuple(t.shape), 20 + i) for i, t in enumerate((raster, g, c))]
    sum((o * u).sum() for o, u in zip((raster, g, c), ups)).backward()

    with torch.no_grad():
        up = grid.upconv2x2(from_up.detach(), level.upconv, None)
        cat = torch.cat((up, from_down.detach()), 1)
        y1 = grid.conv_bias_act(cat, level.conv1, relu=True)
        y2 = grid.conv_bias_act(y1, level.conv2, relu=True)
        g_step = grid.upconv2x2(prev.detach(), level.conv1x1, y2)
        assert torch.equal(g_step, g.detach()), "stepwise forward differs from the module's: masks would not be the module's"
        mask_h = _hidden_mask(level, tile, g_step)
        m1, m2 = (y1 > 0), (y2 > 0)

    def ref(dev64):
        p = {k: v.detach().to(dev64).requires_grad_(True) for k, v in _p64(level).items()}
        to = lambda t: t.detach().to(dev64)                                             # noqa: E731
        fd64, fu64, prev64, cl64 = (to(t).to(D).requires_grad_(True) for t in (from_down, from_up, prev, c_last))
        pts = to(tile.pts)
        up64 = F.conv_transpose2d(fu64, p["upconv.weight"], p["upconv.bias"], stride=2)          # alto.py:217-218
        a1 = _conv64(torch.cat((up64, fd64), 1), p, "conv1", 1) * to(m1).to(D)                   # alto.py:220-227
        a2 = _conv64(a1, p, "conv2", 1) * to(m2).to(D)
        g64 = a2 + F.conv_transpose2d(prev64, p["conv1x1.weight"], p["conv1x1.bias"], stride=2)  # alto.py:233-236
        c64 = _comm64(_sample64(g64, pts), cl64, p, to(mask_h))                                   # alto.py:245-253
        raster64 = _raster64(c64, pts, 128)                                                       # alto.py:255
        outs = (raster64, g64, c64)
        sum((o * to(u).to(D)).sum() for o, u in zip(outs, ups)).backward()
        return outs, p, (fd64, fu64, prev64, cl64)

    outs64, p, ins64 = _run64(ref)
    ck = _Checks()
    for name, got, want in zip(("raster", "g", "c"), (raster, g, c), outs64):
        ck.add("out " + name, got.cpu(), want.cpu())
    for name, got, want in zip(("d from_down", "d from_up", "d prev_conv", "d c_last"), (from_down, from_up, prev, c_last), ins64):
        ck.add(name, got.grad.cpu(), want.grad.cpu())
    for k, v in level.named_parameters():
        ck.add("d " + k, v.grad.cpu(), p[k].grad.cpu())
    ck.finish(f"UpConv 256->128 @64^2->128^2, N={n_points}")


# ------------------------------------------------------------------------------------------------ decoder
@pytest.mark.parametrize("plane,size", [(64, 128), (256, 512)], ids=["64to128", "256to512_bench_size"])
def test_conv_decoder_gradients_vs_float64_with_the_same_masks(plane, size):
    """PixelwiseDecoder(mode='conv') in channels_last mode = bilinear upsample (align_corners=True) + the fused ConvDecoder node
    (three 3x3 convs whose activations each feed the next conv AND the 288 -> 1 head); second parametrisation = the
    benchmarked size (256^2 -> 512^2)."""
    from tomosar2height_amd import grid
    from tomosar2height_amd.decoder.pixel import PixelwiseDecoder
    dev = _dev()
    dec = det_init_(PixelwiseDecoder(hidden_dim=32, out_dim=1, output_size=size, mode="conv"), seed=47).to(dev)
    dec.set_channels_last(True)
    xy = _cl(_rand((1, 32, plane, plane), 1)).requires_grad_(True)
    out, none = dec({"xy": xy})
    assert none is None and out.shape == (1, size, size, 1)
    up = _rand((1, size, size, 1), 2)
    (out * up).sum().backward()
    head = dec.conv_decoder

    with torch.no_grad():
        c0 = grid.upsample_bilinear_cl(xy.detach(), size)
        acts = [c0]
        for conv in (head.conv1, head.conv2, head.conv3):
            acts.append(grid.conv_bias_act(acts[-1], conv, relu=True))
        assert torch.equal(grid.head1x1(acts, head.conv4).permute(0, 2, 3, 1), out.detach()), \
            "stepwise forward differs from the fused decoder node's"
        masks = [a > 0 for a in acts[1:]]

    def ref(dev64):
        p = {k: v.detach().to(dev64).requires_grad_(True) for k, v in _p64(head).items()}
        xy64 = xy.detach().to(dev64).to(D).requires_grad_(True)
        feats = [F.interpolate(xy64, size=size, mode="bilinear", align_corners=True)]          # pixel.py:107
        for i, m in enumerate(masks, start=1):
            feats.append(_conv64(feats[-1], p, f"conv{i}", 1) * m.to(dev64).to(D))                # pixel.py:25-30
        o64 = _conv64(torch.cat(feats, 1), p, "conv4", 0).permute(0, 2, 3, 1)                   # pixel.py:31-32
        (o64 * up.to(dev64).to(D)).sum().backward()
        return o64, p, xy64

    o64, p, xy64 = _run64(ref)
    ck = _Checks()
    # the forward value is a 288-term sum (conv4) over three chained 3x3 convolutions with reductions of 288 .. 1152 fp32
    # products each, whose terms cancel to ~1/10 of their size: its rounding error relative to max|out| is a few 1e-6 typical
    # and grows with the number of pixels the maximum is taken over (262144 at the benchmarked size) -- 3e-5 for the value;
    # the gradients, which this test is about, stay at 1e-5
    ck.add("out", out.cpu(), o64.cpu(), tol=3e-5)
    # gradients: 1e-5; at the benchmarked size the weight gradients are fp32 reductions over 262144 pixels and the input
    # gradient passes through four reductions of up to 1152 terms -- measured 8.4e-6 .. 1.16e-5 there (2.1e-6 .. 3.2e-6 at
    # 128^2), so 2e-5
    gtol = TOL if size <= 128 else 2e-5
    ck.add("d xy", xy.grad.cpu(), xy64.grad.cpu(), tol=gtol)
    for k, v in head.named_parameters():
        ck.add("d " + k, v.grad.cpu(), p[k].grad.cpu(), tol=gtol)
    ck.finish(f"PixelwiseDecoder conv head {plane}^2 -> {size}^2")


# ------------------------------------------------------------------------------------------------ grid-first exchange
def test_bottom_level_grid_first_vs_float64_with_the_same_masks():
    """DownConv i = 4 of the Berlin ALTO U-Net (256 -> 512 channels at 32^2, 128 points per pixel: the level whose fc_comm.0 was
    the largest per-point GEMM): conv pair, pooled 1x1 residual, grid-first exchange with 1024 hidden channels -- against the
    reference's point-first order in float64 with this forward's masks."""
    from tomosar2height_amd import grid, mlp
    from tomosar2height_amd.encoder.alto import DownConv
    from tomosar2height_amd.tile import TileIndex
    dev = _dev()
    n_points = 131072
    level = det_init_(DownConv(256, 512, 4, pooling=False, depth=5), seed=45).to(dev)
    level.channels_last = True
    tile = TileIndex(synth_cloud(n_points, seed=8).to(dev), 256)
    assert mlp.grid_first_applicable(tile, 32, 512)
    x = _cl(_rand((1, 256, 32, 32), 1)).requires_grad_(True)
    prev = _cl(_rand((1, 256, 64, 64), 2)).requires_grad_(True)
    c_last = _rand((n_points, 256), 3).requires_grad_(True)
    pooled, raster, g, c = level(tile, x, prev, c_last)
    assert pooled is raster or torch.equal(pooled, raster)                          # no pooling at the bottom level
    ups = [_rand(tuple(t.shape), 30 + i) for i, t in enumerate((raster, g, c))]
    sum((o * u).sum() for o, u in zip((raster, g, c), ups)).backward()
    with torch.no_grad():
        y1 = grid.conv_bias_act(x.detach(), level.conv1, relu=True)
        y2 = grid.conv_bias_act(y1, level.conv2, relu=True)
        g_step = grid.conv1x1(grid.maxpool2x2(prev.detach(), level.pool), level.conv1x1, y2)
        assert torch.equal(g_step, g.detach())
        mask_h = _hidden_mask(level, tile, g_step)
        m1, m2 = (y1 > 0), (y2 > 0)

    def ref(dev64):
        p = {k: v.detach().to(dev64).requires_grad_(True) for k, v in _p64(level).items()}
        to = lambda t: t.detach().to(dev64)                                             # noqa: E731
        x64, prev64, cl64 = (to(t).to(D).requires_grad_(True) for t in (x, prev, c_last))
        pts = to(tile.pts)
        a1 = _conv64(x64, p, "conv1", 1) * to(m1).to(D)
        a2 = _conv64(a1, p, "conv2", 1) * to(m2).to(D)
        g64 = a2 + _conv64(_pool64(prev64, to(prev)), p, "conv1x1", 0)
        c64 = _comm64(_sample64(g64, pts), cl64, p, to(mask_h))                         # the reference's order: sample first
        raster64 = _raster64(c64, pts, 32)
        outs = (raster64, g64, c64)
        sum((o * to(u).to(D)).sum() for o, u in zip(outs, ups)).backward()
        return outs, p, (x64, prev64, cl64)

    outs64, p, ins64 = _run64(ref)
    ck = _Checks()
    for name, got, want in zip(("raster", "g", "c"), (raster, g, c), outs64):
        ck.add("out " + name, got.cpu(), want.cpu())
    for name, got, want in zip(("d grid_in", "d prev_conv", "d c_last"), (x, prev, c_last), ins64):
        ck.add(name, got.grad.cpu(), want.grad.cpu())
    for k, v in level.named_parameters():
        ck.add("d " + k, v.grad.cpu(), p[k].grad.cpu())
    ck.finish("DownConv 256->512 @32^2 (grid-first), N=131072")


@pytest.mark.parametrize("c,r,n_points", [(512, 32, 131072), (256, 64, 131072), (128, 128, 131072), (64, 256, 262144)])
def test_grid_first_exchange_equals_point_first(c, r, n_points):
    """mlp.comm_mlp_grid_first against sample_plane + mlp.comm_mlp on the same plane, points and weights: the two
    associations of the same function.  Values to 1e-5; gradients to the resolution two fp32 evaluations of one ReLU layer
    have -- a hidden unit within 1e-7 of zero may take either side, and with 128 points per pixel one flipped unit moves a
    plane-gradient entry by a percent of its size (measured: d plane 3.4e-3 max-normalised / 6.4e-4 L2 at C = 512, r = 32) --
    the 1e-2 / 3e-3 every cross-platform gradient comparison of this suite uses.  The tight check of the grid-first path is
    test_bottom_level_grid_first_vs_float64_with_the_same_masks (<= 1.6e-6 with the masks pinned)."""
    from tomosar2height_amd import mlp, ops
    from tomosar2height_amd.tile import TileIndex
    dev = _dev()                          # (r = 256: bench.py --points 262144 runs grid-first there, 4 points per pixel)
    tile = TileIndex(synth_cloud(n_points, seed=9).to(dev), 256)
    lin = lambda o, i, s: det_init_(torch.nn.Linear(i, o), seed=s).to(dev)              # noqa: E731
    fa, fb, fc = lin(2 * c, c, 1), lin(c, 2 * c, 2), lin(c, c // 2, 3)
    plane0 = _cl(_rand((1, c, r, r), 4))
    c_last0 = _rand((n_points, c // 2), 5)
    up = _rand((n_points, c), 6)
    res = {}
    for mode in ("grid", "point"):
        for m in (fa, fb, fc):
            m.zero_grad(set_to_none=True)
        plane, c_last = plane0.clone().requires_grad_(True), c_last0.clone().requires_grad_(True)
        if mode == "grid":
            out, _ = mlp.comm_mlp_grid_first(tile, plane, fa.weight, fa.bias, fb.weight, fb.bias, c_last, fc.weight, fc.bias)
        else:
            out = mlp.comm_mlp(ops.sample_plane(tile, plane), fa.weight, fa.bias, fb.weight, fb.bias, c_last, fc.weight, fc.bias)
        (out * up).sum().backward()
        res[mode] = {"out": out.detach(), "d plane": plane.grad, "d c_last": c_last.grad,
                     **{f"d {n}.{k}": v.grad.clone() for n, m in (("fa", fa), ("fb", fb), ("fc", fc)) for k, v in m.named_parameters()}}
    for k, want in res["point"].items():
        got = res["grid"][k].to(D)
        want = want.to(D)
        scale = want.abs().max().item() + 1e-300
        mx = (got - want).abs().max().item() / scale
        l2 = ((got - want).norm() / (want.norm() + 1e-300)).item()
        lim = (1e-5, 1e-5) if k == "out" else (1e-2, 3e-3)
        assert mx <= lim[0] and l2 <= lim[1], f"{k}: max {mx:.2e}, L2 {l2:.2e}"


# ------------------------------------------------------------------------------------------------ deferred point features
def test_deferred_point_features_equal_the_point_wise_chain():
    """The whole Berlin point encoder at the benchmarked size, deferred form (deferred.py: no per-point c from the first
    256-channel level on) against the point-wise chain (T2H_DEFER_MIN_CHANNELS = 0) with the same weights and tile: the
    output plane to 1e-5, every parameter gradient at the mask-flip resolution (1e-2 max-normalised / 3e-3 L2)."""
    from tomosar2height_amd import deferred
    from tomosar2height_amd.encoder.pointnet import LocalPoolPointnet
    from tomosar2height_amd.synthetic import berlin_tile
    dev = _dev()
    enc = det_init_(LocalPoolPointnet(feature_dim=32, dim=3, hidden_dim=32, scatter_type="max", unet_type="alto",
                                      unet_kwargs=dict(depth=5, merge_mode="concat", start_filts=32), plane_resolution=256),
                    seed=51).to(dev)
    enc.set_channels_last(True)
    cloud = berlin_tile(seed=3)["inputs"].to(dev)
    gout = _rand((1, 32, 256, 256), 7)
    res = {}
    old = deferred.DEFER_MIN_CHANNELS
    try:
        for mode, thr in (("deferred", 256), ("pointwise", 0)):
            deferred.DEFER_MIN_CHANNELS = thr
            enc.zero_grad(set_to_none=True)
            out = enc(cloud)["xy"]
            out.backward(gout)
            res[mode] = (out.detach().clone(), {k: v.grad.clone() for k, v in enc.named_parameters() if v.grad is not None})
    finally:
        deferred.DEFER_MIN_CHANNELS = old
    (o1, g1), (o0, g0) = res["deferred"], res["pointwise"]
    assert g1.keys() == g0.keys()
    err = ((o1 - o0).abs().max() / o0.abs().max()).item()
    assert err <= 1e-5, f"plane: {err:.2e}"
    worst = []
    for k in g0:
        a, b = g1[k].to(D), g0[k].to(D)
        mx = ((a - b).abs().max() / (b.abs().max() + 1e-300)).item()
        l2 = ((a - b).norm() / (b.norm() + 1e-300)).item()
        worst.append((mx, l2, k))
    worst.sort(reverse=True)
    print(f"[deferred vs point-wise] plane {err:.2e}; worst gradients:", [(f"{m:.1e}", f"{l:.1e}", k) for m, l, k in worst[:4]])
    for mx, l2, k in worst:
        assert mx <= 1e-2 and l2 <= 3e-3, f"{k}: max {mx:.2e}, L2 {l2:.2e}"


# ------------------------------------------------------------------------------------------------ image U-Net levels
# (the plain image U-Net -- unet.py:112-187 -- had no mask-pinned level check, and it is where the module-level
# gradient bound is loosest.)  Same method: the float64 restatement takes the HIP forward's own ReLU masks / pool winners.
@pytest.mark.parametrize("cin,cout,hw,pooling", [(64, 128, 128, True), (32, 64, 256, True), (512, 1024, 16, False)],
                         ids=["down2_64_128_at_128", "down1_32_64_at_256", "down5_512_1024_at_16"])
def test_image_unet_down_level_vs_float64_with_the_same_masks(cin, cout, hw, pooling):
    """DownConv of the image encoder (unet.py: conv3x3 -> ReLU -> conv3x3 -> ReLU -> 2x2 max-pool): both outputs and every
    gradient, on the convolution kernels the model takes at that size (matrix cores with the exact bf16 split from 32-wide
    planes, conv.hip's fp32 MFMA below)."""
    from tomosar2height_amd import grid
    from tomosar2height_amd.encoder.unet import DownConv
    dev = _dev()
    level = det_init_(DownConv(cin, cout, pooling=pooling), seed=61).to(dev)
    level.channels_last = True
    for m in (level.conv1, level.conv2):
        m.weight.data = _cl(m.weight.data)
    x = _cl(_rand((1, cin, hw, hw), 1)).requires_grad_(True)
    pooled, before = level(x)
    g_pool, g_before = _rand(tuple(pooled.shape), 2), _rand(tuple(before.shape), 3)
    ((pooled * g_pool).sum() + ((before * g_before).sum() if pooling else 0.0)).backward()
    with torch.no_grad():                                       # the same launches -> the same bits -> the forward's masks
        y1 = grid._empty_cl(1, cout, hw, hw, dev)
        grid.conv3x3_fwd_(x.detach(), grid._w_cl(level.conv1.weight), level.conv1.bias, y1, relu=True)
    m1, m2 = (y1 > 0).to(D), (before.detach() > 0).to(D)

    def restate(d):
        p = {k: v.detach().to(d).requires_grad_(True) for k, v in _p64(level).items()}
        x64 = x.detach().to(d, D).requires_grad_(True)
        a1 = _conv64(x64, p, "conv1", 1) * m1.to(d)
        a2 = _conv64(a1, p, "conv2", 1) * m2.to(d)
        out = _pool64(a2, before.detach().to(d)) if pooling else a2
        ((out * g_pool.to(d, D)).sum() + ((a2 * g_before.to(d, D)).sum() if pooling else 0.0)).backward()
        return out, a2, x64, p
    out64, a264, x64, p = _run64(restate)
    ck = _Checks()
    ck.add("out pooled", pooled.cpu(), out64.cpu())
    ck.add("out before_pool", before.cpu(), a264.cpu())
    ck.add("d x", x.grad.cpu(), x64.grad.cpu())
    for k, v in level.named_parameters():
        ck.add("d " + k, v.grad.cpu(), p[k].grad.cpu())
    ck.finish(f"image U-Net DownConv {cin}->{cout} @{hw}^2")


def test_image_unet_up_level_vs_float64_with_the_same_masks():
    """UpConv 0 of the image encoder (1024 -> 512: transposed 2x2 conv of the 16 x 16 bottom plane, concat with the 32 x 32 skip,
    conv3x3 -> ReLU -> conv3x3 -> ReLU): output and every gradient."""
    from tomosar2height_amd import grid
    from tomosar2height_amd.encoder.unet import UpConv
    dev = _dev()
    level = det_init_(UpConv(1024, 512), seed=62).to(dev)
    level.channels_last = True
    for m in (level.conv1, level.conv2, level.upconv):
        m.weight.data = _cl(m.weight.data)
    from_up = _cl(_rand((1, 1024, 16, 16), 4)).requires_grad_(True)
    from_down = _cl(_rand((1, 512, 32, 32), 5)).requires_grad_(True)
    out = level(from_down, from_up)
    g = _rand(tuple(out.shape), 6)
    (out * g).sum().backward()
    with torch.no_grad():
        up = grid.upconv2x2(from_up.detach(), level.upconv)
        y1 = grid._empty_cl(1, 512, 32, 32, dev)
        grid.conv3x3_fwd_(_cl(torch.cat((up, from_down.detach()), 1)), grid._w_cl(level.conv1.weight), level.conv1.bias, y1, relu=True)
    m1, m2 = (y1 > 0).to(D), (out.detach() > 0).to(D)

    def restate(d):
        p = {k: v.detach().to(d).requires_grad_(True) for k, v in _p64(level).items()}
        fu = from_up.detach().to(d, D).requires_grad_(True)
        fd = from_down.detach().to(d, D).requires_grad_(True)
        u = F.conv_transpose2d(fu, p["upconv.weight"], p["upconv.bias"], stride=2)
        a1 = _conv64(torch.cat((u, fd), 1), p, "conv1", 1) * m1.to(d)
        a2 = _conv64(a1, p, "conv2", 1) * m2.to(d)
        (a2 * g.to(d, D)).sum().backward()
        return a2, fu, fd, p
    a264, fu, fd, p = _run64(restate)
    ck = _Checks()
    ck.add("out", out.cpu(), a264.cpu())
    ck.add("d from_up", from_up.grad.cpu(), fu.grad.cpu())
    ck.add("d from_down", from_down.grad.cpu(), fd.grad.cpu())
    for k, v in level.named_parameters():
        ck.add("d " + k, v.grad.cpu(), p[k].grad.cpu())
    ck.finish("image U-Net UpConv 1024->512 @16^2->32^2")
