"""GPU edge cases and skew through the whole point<->grid operator set, against the C oracle: single point, sizes
around the 64-lane / 2048-key tile boundaries, every point in one cell, points on a line, two tiles of very different
density in one batch, points on cell and plane borders.  Index results bit exact, fp32 as stated per assert."""
import numpy as np
import pytest
import torch

from detinit import synth_cloud

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _clouds():
    g = torch.Generator().manual_seed(123)
    out = {}
    for n in (1, 2, 63, 64, 65, 2047, 2048, 2049):
        out[f"uniform_{n}"] = synth_cloud(n, seed=n, clustered=False)
    one_cell = torch.full((1, 777, 3), 0.3137)
    one_cell[..., 2] = torch.rand(1, 777, generator=g)
    out["one_cell"] = one_cell
    line = torch.rand(1, 3000, 3, generator=g)
    line[..., 1] = 0.5                                     # a vertical facade: one row of cells
    out["line"] = line
    corners = torch.tensor([[[2.0 ** -20, 2.0 ** -20, 0.1], [1 - 2.0 ** -24, 2.0 ** -20, 0.2], [2.0 ** -20, 1 - 2.0 ** -24, 0.3],
                             [1 - 2.0 ** -24, 1 - 2.0 ** -24, 0.4], [0.5, 0.5, 0.5], [0.25, 0.75, 0.0]]])
    out["corners"] = corners.repeat(1, 50, 1)
    dense = synth_cloud(4000, seed=9)
    sparse = torch.full((1, 4000, 3), 0.9)
    sparse[0, :5] = synth_cloud(5, seed=3)[0]
    out["uneven_batch"] = torch.cat([dense, sparse], 0)    # tile 1: 3995 points in one cell
    return out


CLOUDS = _clouds()


@pytest.mark.parametrize("name", sorted(CLOUDS))
@pytest.mark.parametrize("reso,level,c", [(16, 0, 8), (64, 2, 32), (256, 3, 64)])
def test_ops_vs_oracle_on_edge_clouds(name, reso, level, c):
    from oracle import c_oracle
    from tomosar2height_amd import ops
    from tomosar2height_amd.tile import TileIndex
    cloud = CLOUDS[name]
    b, n, _ = cloud.shape
    r = reso >> level
    g = torch.Generator().manual_seed(len(name) + reso)
    t = TileIndex(cloud.to(_dev()), reso)
    assert t.out_of_domain() == 0
    # tile index
    idx0 = c_oracle.coordinate2index(cloud.numpy(), reso)
    counts = np.zeros(b * reso * reso, np.int64)
    # pool (finest level), quantised values -> ties
    feat = (torch.randn(b, n, c, generator=g) * 2).round() / 2
    gout = torch.randn(b, n, c, generator=g)
    f = t.sort_rows(feat.to(_dev())).requires_grad_(True)
    pooled = ops.pool_max(t, f)
    pooled.backward(t.sort_rows(gout.to(_dev())))
    want, arg = c_oracle.pool_local_fwd(feat.numpy(), idx0, reso * reso)
    assert np.array_equal(t.unsort_rows(pooled.detach()).cpu().numpy(), want)
    want_g = c_oracle.pool_local_bwd(gout.numpy(), idx0, arg, reso * reso)
    got_g = t.unsort_rows(f.grad).cpu().numpy()
    scale = np.abs(want_g).max() + 1e-9
    assert np.abs(got_g - want_g).max() <= 2e-5 * scale
    assert np.array_equal(got_g != 0, want_g != 0)
    # rasterise at level
    idx = c_oracle.coordinate2index(cloud.numpy(), r)
    f2 = t.sort_rows(feat.to(_dev())).requires_grad_(True)
    plane = ops.rasterise_mean(t, f2, r)
    wantp = c_oracle.scatter_mean_fwd(feat.numpy(), idx, r)
    np.testing.assert_allclose(plane.detach().cpu().numpy(), wantp, rtol=2e-5, atol=2e-6)
    gp = torch.randn(b, c, r, r, generator=g)
    plane.backward(gp.to(_dev()))
    np.testing.assert_allclose(t.unsort_rows(f2.grad).cpu().numpy(), c_oracle.scatter_mean_bwd(gp.numpy(), idx, n),
                               rtol=1e-6, atol=1e-7)
    # sample + deterministic backward
    pl = torch.randn(b, c, r, r, generator=g)
    p = pl.to(_dev()).requires_grad_(True)
    out = ops.sample_plane(t, p)
    np.testing.assert_allclose(t.unsort_rows(out.detach()).cpu().numpy(), c_oracle.grid_sample_fwd(pl.numpy(), cloud.numpy()),
                               rtol=1e-5, atol=1e-6)
    out.backward(t.sort_rows(gout.to(_dev())))
    wantgp = c_oracle.grid_sample_bwd(gout.numpy(), cloud.numpy(), r, r)
    sc = np.abs(wantgp).max() + 1e-9
    assert np.abs(p.grad.cpu().numpy() - wantgp).max() <= 1e-4 * sc


def test_full_model_on_degenerate_tiles():
    """The whole network on a 1-point tile and on an all-in-one-cell tile: finite heights equal to the oracle's."""
    from detinit import det_init_
    from oracle import torch_ref
    from tomosar2height_amd import TomoSAR2Height
    from tomosar2height_amd.config import berlin_config
    cfg = berlin_config()
    ref = det_init_(torch_ref.TomoSAR2Height(cfg), seed=31)
    model = TomoSAR2Height(cfg)
    model.load_state_dict(ref.state_dict())
    model.to(_dev())
    for name in ("uniform_1", "one_cell", "uniform_65"):
        cloud = CLOUDS[name]
        with torch.no_grad():
            want, _ = ref(input_cloud=cloud)
            got, _ = model(input_cloud=cloud.to(_dev()))
        assert torch.isfinite(got).all()
        scale = want.abs().max().item() + 1e-9
        assert (got.cpu() - want).abs().max().item() <= 1e-4 * scale, name


@pytest.mark.parametrize("channels_last", [False, True])
def test_training_step_on_ragged_point_counts(channels_last):
    """Forward + backward of the whole network at point counts around the tile edges of the GEMM kernels (rows of the
    per-point GEMMs, reduction length of the weight gradients): heights equal to the oracle's, every gradient finite."""
    from detinit import det_init_
    from oracle import torch_ref
    from tomosar2height_amd import TomoSAR2Height
    from tomosar2height_amd.config import berlin_config
    cfg = berlin_config()
    cfg.model.encoder_kwargs.unet_kwargs.depth = 4
    import tomosar2height_amd as t2h
    t2h.allow_library_fallback(not channels_last).set()        # the NCHW grid side is MIOpen's by definition
    ref = det_init_(torch_ref.TomoSAR2Height(cfg), seed=32)
    model = TomoSAR2Height(cfg)
    model.load_state_dict(ref.state_dict())
    model.to(_dev())
    model.set_channels_last(channels_last)
    for n in (1, 7, 129, 4097):
        cloud = synth_cloud(n, seed=40 + n)
        model.zero_grad(set_to_none=True)
        got, _ = model(input_cloud=cloud.to(_dev()))
        got.square().mean().backward()
        with torch.no_grad():
            want, _ = ref(input_cloud=cloud)
        scale = want.abs().max().item() + 1e-9
        assert (got.detach().cpu() - want).abs().max().item() <= 1e-4 * scale, n
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        assert len(grads) > 100 and all(torch.isfinite(g).all() for g in grads), n


@pytest.mark.parametrize("name", ["uniform_1", "uniform_65", "uniform_2049", "one_cell", "line", "corners", "uneven_batch"])
@pytest.mark.parametrize("c", [4, 32, 64])
def test_row_balanced_pooling_equals_cell_parallel_pooling(name, c):
    """t2h_pool_rows_fwd/bwd (a workgroup per 128 sorted rows, cells crossing chunk borders reduced cooperatively) against
    t2h_pool_max_fwd/bwd (a lane group per cell): identical pooled values and winner bits on every edge cloud (dense cells
    spanning many chunks, single points, two tiles), gradients equal to summation-order rounding."""
    from tomosar2height_amd import _lib
    from tomosar2height_amd.tile import TileIndex
    if name not in CLOUDS:
        pytest.skip("cloud not in this build of the edge set")
    cloud = CLOUDS[name].to(_dev())
    tile = TileIndex(cloud, 16)
    n = tile.n_points
    g = torch.Generator().manual_seed(n + c)
    feat = torch.randint(-3, 4, (n, c), generator=g).float().to(_dev())       # many exact ties
    gp = torch.randn(n, c, generator=g).to(_dev())
    ws = _lib.load().t2h_pool_winner_stride(c)
    outs = []
    for rows in (False, True):
        pooled = torch.empty_like(feat)
        winner = torch.zeros(n, ws, dtype=torch.uint8, device=_dev())
        gfeat = torch.full_like(feat, 0.5)
        if rows:
            _lib.call("t2h_pool_rows_fwd", _lib.ptr(feat), c, _lib.ptr(tile.cell), _lib.ptr(tile.off0), n, c, _lib.ptr(pooled), c,
                      _lib.ptr(winner), _lib.stream())
            _lib.call("t2h_pool_rows_bwd", _lib.ptr(gp), c, _lib.ptr(winner), _lib.ptr(tile.cell), _lib.ptr(tile.off0), n, c, 1,
                      _lib.ptr(gfeat), c, _lib.stream())
        else:
            _lib.call("t2h_pool_max_fwd", _lib.ptr(feat), c, _lib.ptr(tile.off0), tile.B, tile.nbits, c, _lib.ptr(pooled), c,
                      _lib.ptr(winner), _lib.stream())
            _lib.call("t2h_pool_max_bwd", _lib.ptr(gp), c, _lib.ptr(winner), _lib.ptr(tile.off0), tile.B, tile.nbits, c, 1,
                      _lib.ptr(gfeat), c, _lib.stream())
        outs.append((pooled, winner, gfeat))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    scale = outs[0][2].abs().max().item() + 1e-12
    assert (outs[0][2] - outs[1][2]).abs().max().item() <= 2e-6 * scale
