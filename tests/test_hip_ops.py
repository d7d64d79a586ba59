"""GPU parity: every HIP operator (through the C ABI) against the C oracle on the same seeded inputs and
against the fixtures captured from the reference.  Integer/index results bit exact; fp32 within the
tolerance written at each assert (north_star: 1e-4 relative; most ops are far tighter)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from detinit import synth_cloud

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _morton(ix, iy):
    def part(v):
        v = v.astype(np.uint32) & 0xFFFF
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return part(ix) | (part(iy) << 1)


def _tile(cloud, reso):
    from tomosar2height_amd.tile import TileIndex
    return TileIndex(cloud.to(_dev()), reso)


@pytest.mark.parametrize("n,reso,batch", [(1, 2, 1), (5, 2, 1), (300, 16, 1), (2049, 64, 2), (70000, 256, 1),
                                          (131072, 256, 1)])
def test_tile_index_bit_exact(n, reso, batch):
    from oracle import c_oracle
    cloud = synth_cloud(n, seed=n + reso, batch=batch)
    t = _tile(cloud, reso)
    assert t.out_of_domain() == 0
    idx = c_oracle.coordinate2index(cloud.numpy(), reso)[:, 0]                     # [B,N] reference cell ids
    m0 = reso * reso
    want_code = _morton(idx % reso, idx // reso).astype(np.int64) + np.arange(batch)[:, None] * m0
    perm = t.perm.cpu().numpy().reshape(batch, n).astype(np.int64)
    cell = t.cell.cpu().numpy().reshape(batch, n).astype(np.int64)
    off0 = t.off0.cpu().numpy().astype(np.int64)
    for b in range(batch):
        assert np.array_equal(np.sort(perm[b]), np.arange(n))                      # a permutation
        assert np.array_equal(cell[b], want_code[b][perm[b]])                      # cell of each sorted point
        assert np.all(np.diff(cell[b]) >= 0)                                       # sorted
        same = np.diff(cell[b]) == 0
        assert np.all(np.diff(perm[b])[same] > 0)                                  # stable inside a cell
    counts = np.bincount(want_code.reshape(-1), minlength=batch * m0)
    assert np.array_equal(off0, np.concatenate([[0], np.cumsum(counts)]))          # CSR
    pts = t.pts.cpu().numpy().reshape(batch, n, 3)
    for b in range(batch):
        assert np.array_equal(pts[b], cloud.numpy()[b][perm[b]])


def test_tile_index_out_of_domain_is_counted():
    cloud = synth_cloud(100, seed=1)
    cloud[0, 3, 0] = 1.0
    cloud[0, 7, 1] = -0.25
    cloud[0, 9, 0] = float("nan")
    t = _tile(cloud, 16)
    assert t.out_of_domain() == 3
    with pytest.raises(ValueError):
        t.check_domain()


def test_coordinate2index_golden():
    from tomosar2height_amd import ops
    g = load_golden("coordinate2index")
    xy = torch.from_numpy(g["xy"]).to(_dev())
    for reso in (2, 16, 32, 64, 128, 256):
        assert np.array_equal(ops.coordinate2index(xy, reso).cpu().numpy(), g[f"index_r{reso}"])


def _pool_case(cloud, feat, reso, gout):
    from tomosar2height_amd import ops
    from oracle import c_oracle
    t = _tile(cloud, reso)
    f = t.sort_rows(feat.to(_dev())).requires_grad_(True)
    pooled = ops.pool_max(t, f)
    pooled.backward(t.sort_rows(gout.to(_dev())))
    idx = c_oracle.coordinate2index(cloud.numpy(), reso)
    want, arg = c_oracle.pool_local_fwd(feat.numpy(), idx, reso * reso)
    want_g = c_oracle.pool_local_bwd(gout.numpy(), idx, arg, reso * reso)
    return t.unsort_rows(pooled.detach()).cpu().numpy(), t.unsort_rows(f.grad).cpu().numpy(), want, want_g


@pytest.mark.parametrize("n,reso,c,batch", [(300, 4, 8, 1), (300, 16, 32, 2), (5000, 64, 32, 1), (777, 16, 12, 1),
                                            (40000, 256, 32, 1), (3000, 32, 6, 1), (1000, 8, 512, 1)])
def test_pool_max_vs_oracle(n, reso, c, batch):
    g = torch.Generator().manual_seed(n + c)
    cloud = synth_cloud(n, seed=n, batch=batch)
    feat = (torch.randn(batch, n, c, generator=g) * 4).round() / 4         # quantised: plenty of exact ties
    gout = torch.randn(batch, n, c, generator=g)
    got, got_g, want, want_g = _pool_case(cloud, feat, reso, gout)
    assert np.array_equal(got, want)                                        # max is exact
    np.testing.assert_allclose(got_g, want_g, rtol=1e-5, atol=1e-5)        # sums over a cell: order-free to 1e-5
    assert np.array_equal(got_g != 0, want_g != 0)                          # same arg-max routing (tie-break)


def test_pool_max_golden_and_single_cell():
    g = load_golden("pool_local")
    for reso in (4, 16):
        cloud = torch.cat([torch.from_numpy(g[f"xy_r{reso}"]), torch.zeros(1, 300, 1)], 2)
        got, got_g, _, _ = _pool_case(cloud, torch.from_numpy(g[f"feat_r{reso}"]), reso,
                                      torch.from_numpy(g[f"gout_r{reso}"]))
        assert np.array_equal(got, g[f"out_r{reso}"])
        np.testing.assert_allclose(got_g, g[f"gfeat_r{reso}"], rtol=1e-5, atol=1e-5)
    # all points in ONE cell, all values equal: first point takes the whole gradient
    n = 1000
    cloud = torch.full((1, n, 3), 0.5)
    feat = torch.ones(1, n, 32)
    gout = torch.ones(1, n, 32)
    got, got_g, want, want_g = _pool_case(cloud, feat, 256, gout)
    assert np.array_equal(got, want)
    assert np.array_equal(got_g, want_g) and got_g[0, 0, 0] == n and got_g[0, 1:].sum() == 0


def _pool_mean_case(cloud, feat, reso, gout):
    from tomosar2height_amd import ops
    t = _tile(cloud, reso)
    f = t.sort_rows(feat.to(_dev())).requires_grad_(True)
    pooled = ops.pool_mean(t, f)
    pooled.backward(t.sort_rows(gout.to(_dev())))
    return t.unsort_rows(pooled.detach()).cpu().numpy(), t.unsort_rows(f.grad).cpu().numpy()


def test_pool_mean_golden():
    """scatter_type='mean' pooling (pointnet.py:55-56, 92-99) against the reference's own pool_local: the stable sort
    keeps the point order inside a cell, so the sums -- and the means -- are the reference's bit for bit."""
    g = load_golden("pool_local_mean")
    for reso in (4, 16):
        cloud = torch.cat([torch.from_numpy(g[f"xy_r{reso}"]), torch.zeros(1, 300, 1)], 2)
        got, got_g = _pool_mean_case(cloud, torch.from_numpy(g[f"feat_r{reso}"]), reso, torch.from_numpy(g[f"gout_r{reso}"]))
        assert np.array_equal(got, g[f"out_r{reso}"])
        assert np.array_equal(got_g, g[f"gfeat_r{reso}"])


@pytest.mark.parametrize("n,reso,c,batch", [(300, 4, 8, 1), (300, 16, 32, 2), (5000, 64, 32, 1), (777, 16, 12, 1),
                                            (40000, 256, 32, 1), (3000, 32, 6, 1), (1000, 8, 512, 1)])
def test_pool_mean_vs_oracle(n, reso, c, batch):
    from oracle import torch_ref, c_oracle
    g = torch.Generator().manual_seed(n + c)
    cloud = synth_cloud(n, seed=n, batch=batch)
    feat = torch.randn(batch, n, c, generator=g)
    gout = torch.randn(batch, n, c, generator=g)
    got, got_g = _pool_mean_case(cloud, feat, reso, gout)
    idx = torch.from_numpy(c_oracle.coordinate2index(cloud.numpy(), reso))
    f = feat.clone().requires_grad_(True)
    want = torch_ref.pool_local(idx, f, reso, "mean")
    want.backward(gout)
    assert np.array_equal(got.reshape(batch, n, c), want.detach().numpy())
    assert np.array_equal(got_g.reshape(batch, n, c), f.grad.numpy())


@pytest.mark.parametrize("n,reso,c,level,batch", [(257, 4, 8, 0, 1), (257, 16, 8, 0, 1), (257, 32, 12, 0, 1),
                                                  (5000, 64, 32, 0, 2), (5000, 64, 64, 1, 1), (5000, 64, 128, 3, 1),
                                                  (40000, 256, 32, 0, 1), (20000, 256, 512, 3, 1), (3000, 256, 256, 8, 1)])
def test_rasterise_mean_vs_oracle(n, reso, c, level, batch):
    from tomosar2height_amd import ops
    from oracle import c_oracle
    g = torch.Generator().manual_seed(n + c + level)
    cloud = synth_cloud(n, seed=n + 1, batch=batch)
    feat = torch.randn(batch, n, c, generator=g)
    r = reso >> level
    t = _tile(cloud, reso)
    f = t.sort_rows(feat.to(_dev())).requires_grad_(True)
    plane = ops.rasterise_mean(t, f, r)
    assert plane.shape == (batch, c, r, r) and plane.is_contiguous()
    gout = torch.randn(batch, c, r, r, generator=g)
    plane.backward(gout.to(_dev()))
    idx = c_oracle.coordinate2index(cloud.numpy(), r)
    want = c_oracle.scatter_mean_fwd(feat.numpy(), idx, r)
    np.testing.assert_allclose(plane.detach().cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    assert np.array_equal(plane.detach().cpu().numpy() == 0, want == 0)      # empty cells are exactly 0
    want_g = c_oracle.scatter_mean_bwd(gout.numpy(), idx, n)
    np.testing.assert_allclose(t.unsort_rows(f.grad).cpu().numpy(), want_g, rtol=1e-6, atol=1e-7)
    # channels_last output is the same numbers in NHWC memory
    plane_cl = ops.rasterise_mean(t, f.detach(), r, channels_last=True)
    assert torch.equal(plane_cl, plane.detach()) and plane_cl.permute(0, 2, 3, 1).is_contiguous()


def test_rasterise_mean_golden():
    from tomosar2height_amd import ops
    g = load_golden("scatter_mean_plane")
    for reso in (4, 16, 32):
        xy = torch.from_numpy(g[f"xy_r{reso}"])
        cloud = torch.cat([xy, torch.zeros(1, xy.shape[1], 1)], 2)
        t = _tile(cloud, reso)
        f = t.sort_rows(torch.from_numpy(g[f"feat_r{reso}"]).to(_dev())).requires_grad_(True)
        plane = ops.rasterise_mean(t, f, reso)
        np.testing.assert_allclose(plane.detach().cpu().numpy(), g[f"plane_r{reso}"], rtol=1e-5, atol=1e-6)
        plane.backward(torch.from_numpy(g[f"gout_r{reso}"]).to(_dev()))
        np.testing.assert_allclose(t.unsort_rows(f.grad).cpu().numpy(), g[f"gfeat_r{reso}"], rtol=1e-6, atol=1e-7)
    # the reference's own known-answer vector (pointnet.py:114-123)
    g = load_golden("pointnet_main_vector")
    xy = torch.from_numpy(g["xy"])
    t = _tile(torch.cat([xy, torch.zeros(1, 5, 1)], 2), 2)
    plane = ops.rasterise_mean(t, t.sort_rows(xy.to(_dev())), 2)
    np.testing.assert_allclose(plane.cpu().numpy(), g["plane"], rtol=1e-6)


@pytest.mark.parametrize("n,reso,c,level,batch,cl", [(150, 8, 4, 0, 1, False), (150, 16, 8, 0, 2, True),
                                                     (5000, 64, 32, 0, 1, False), (5000, 64, 64, 1, 1, True),
                                                     (40000, 256, 32, 0, 1, False), (20000, 256, 512, 3, 1, False),
                                                     (3000, 64, 12, 2, 1, False), (2000, 256, 16, 8, 1, False),
                                                     (30000, 256, 128, 1, 2, True), (9000, 256, 64, 0, 1, True),
                                                     (700, 8, 32, 0, 3, False), (6000, 32, 20, 1, 1, False)])
def test_sample_plane_vs_oracle(n, reso, c, level, batch, cl):
    from tomosar2height_amd import ops
    from oracle import c_oracle
    g = torch.Generator().manual_seed(n + c)
    cloud = synth_cloud(n, seed=n + 2, batch=batch)
    r = reso >> level
    plane = torch.randn(batch, c, r, r, generator=g)
    gout = torch.randn(batch, n, c, generator=g)
    t = _tile(cloud, reso)
    p = plane.to(_dev())
    if cl:
        p = p.contiguous(memory_format=torch.channels_last)
    p.requires_grad_(True)
    out = ops.sample_plane(t, p)
    out.backward(t.sort_rows(gout.to(_dev())))
    want = c_oracle.grid_sample_fwd(plane.numpy(), cloud.numpy())
    np.testing.assert_allclose(t.unsort_rows(out.detach()).cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    want_g = c_oracle.grid_sample_bwd(gout.numpy(), cloud.numpy(), r, r)
    scale = np.abs(want_g).max() + 1e-6
    np.testing.assert_allclose(p.grad.cpu().numpy(), want_g, rtol=1e-4, atol=1e-5 * scale)
    # the backward has no atomics at any level: a second run gives the same bits
    first = p.grad.clone()
    p.grad = None
    ops.sample_plane(t, p).backward(t.sort_rows(gout.to(_dev())))
    assert torch.equal(first, p.grad)


@pytest.mark.parametrize("n,reso,c,level,batch", [(9000, 256, 64, 0, 1), (30000, 256, 128, 1, 2), (40000, 256, 32, 0, 1),
                                                  (3000, 32, 16, 0, 1), (131072, 256, 64, 0, 1), (300, 8, 64, 0, 3),
                                                  (5000, 64, 12, 1, 1), (2000, 256, 20, 8, 1), (60000, 256, 256, 2, 1)])
def test_sample_bwd_through_the_transposed_matrix_is_bit_identical_to_the_gather(n, reso, c, level, batch):
    """t2h_sample_adjoint_build + t2h_sample_bwd_adjoint (the CSR of the transposed sampling matrix, cached per tile and
    level) list a pixel's (row, weight) entries in the order the per-pixel gather visits them: identical bits, with and
    without an addend; every point contributes at most four entries and the weights of a point sum to <= 1."""
    from tomosar2height_amd import _lib
    g = torch.Generator().manual_seed(n + c)
    t = _tile(synth_cloud(n, seed=n + 1, batch=batch), reso)
    r = reso >> level
    gout = torch.randn(batch * n, c, generator=g).to(_dev())
    addend = torch.randn(batch, r, r, c, generator=g).to(_dev())
    offsets, entries = t.sample_adjoint(level)
    assert t.sample_adjoint(level)[0] is offsets                       # cached
    off = offsets[:batch * r * r + 1].cpu().numpy()
    assert off[0] == 0 and (np.diff(off) >= 0).all() and off[-1] <= 4 * batch * n
    ent = entries[:off[-1]].cpu().numpy()
    w = ent[:, 1].copy().view(np.float32)
    assert ent[:, 0].min() >= 0 and ent[:, 0].max() < batch * n and (w >= 0).all() and (w <= 1).all()
    per_point = np.bincount(ent[:, 0], weights=w.astype(np.float64), minlength=batch * n)
    assert per_point.max() <= 1 + 1e-5 and np.bincount(ent[:, 0], minlength=batch * n).max() <= 4
    for add in (None, addend):
        a = torch.full((batch, r, r, c), float("nan"), device=_dev())
        _lib.call("t2h_sample_bwd_adjoint", _lib.ptr(gout), _lib.ptr(offsets), _lib.ptr(entries), batch, t.nbits, level, c,
                  None if add is None else _lib.ptr(add), _lib.ptr(a), _lib.stream())
        b = torch.full((batch, r, r, c), float("nan"), device=_dev())
        # the reference run: float atomics are not bit-stable, so compare with the deterministic kernels -- bit for bit with
        # the gather, to rounding with the per-cell partials the coarse levels use
        lib = _lib.load()
        ws_bytes = lib.t2h_sample_bwd_workspace_bytes(batch, n, t.nbits, level, c)
        ws = _lib.workspace(ws_bytes, _dev())
        _lib.call("t2h_sample_bwd_add", _lib.ptr(gout), _lib.ptr(t.pts), t.dim, _lib.ptr(t.off0), batch, n, t.nbits, level, c,
                  None if add is None else _lib.ptr(add), _lib.ptr(b), _lib.ptr(ws), ws_bytes, _lib.stream())
        if ws_bytes == 0:
            assert torch.equal(a, b)
        else:
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()))     # two orders of summation


def test_sample_plane_golden():
    from tomosar2height_amd import ops
    g = load_golden("grid_sample_points")
    for r in (8, 16):
        cloud = torch.from_numpy(g[f"p_r{r}"]).clone()
        # the fixture probes x == 1.0 / y == 1.0 (grid_sample's border clip); cell binning needs [0,1), so
        # keep the sample coordinates but build the tile on coordinates nudged inside the border cell
        t_cloud = cloud.clone()
        t_cloud[..., :2] = t_cloud[..., :2].clamp(max=1 - 2.0 ** -24)
        same_cell = torch.equal((t_cloud[..., :2] * r).long().clamp(max=r - 1), (cloud[..., :2] * r).long().clamp(max=r - 1))
        assert same_cell
        t = _tile(t_cloud, r)
        t.pts.copy_(t.sort_rows(cloud.to(_dev())))                       # sample at the fixture's exact coordinates
        p = torch.from_numpy(g[f"plane_r{r}"]).to(_dev()).requires_grad_(True)
        out = ops.sample_plane(t, p)
        want = np.transpose(g[f"out_r{r}"], (0, 2, 1))
        np.testing.assert_allclose(t.unsort_rows(out.detach()).cpu().numpy(), want, rtol=1e-5, atol=1e-6)
        out.backward(t.sort_rows(torch.from_numpy(np.transpose(g[f"gout_r{r}"], (0, 2, 1)).copy()).to(_dev())))
        np.testing.assert_allclose(p.grad.cpu().numpy(), g[f"gplane_r{r}"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("b,c,h,size", [(1, 32, 256, 512), (2, 3, 16, 32), (1, 5, 7, 19), (1, 4, 32, 32), (1, 2, 1, 8)])
def test_upsample_bilinear_vs_oracle(b, c, h, size):
    from tomosar2height_amd import ops
    from oracle import c_oracle
    g = torch.Generator().manual_seed(h + size)
    x = torch.randn(b, c, h, h, generator=g)
    add = torch.randn(b, c, size, size, generator=g)
    gout = torch.randn(b, c, size, size, generator=g)
    xd = x.to(_dev()).requires_grad_(True)
    y = ops.upsample_bilinear(xd, size)
    np.testing.assert_allclose(y.detach().cpu().numpy(), c_oracle.upsample_bilinear_fwd(x.numpy(), size),
                               rtol=1e-6, atol=1e-6)
    y.backward(gout.to(_dev()))
    np.testing.assert_allclose(xd.grad.cpu().numpy(), c_oracle.upsample_bilinear_bwd(gout.numpy(), h, h),
                               rtol=1e-5, atol=1e-5)
    ad = add.to(_dev()).requires_grad_(True)
    y2 = ops.upsample_bilinear(xd.detach(), size, ad)
    np.testing.assert_allclose(y2.detach().cpu().numpy(), y.detach().cpu().numpy() + add.numpy(), rtol=1e-6, atol=1e-6)
    y2.backward(gout.to(_dev()))
    assert torch.equal(ad.grad.cpu(), gout)


def test_layout_roundtrip():
    from tomosar2height_amd import ops
    x = torch.randn(2, 37, 19, 19, device=_dev())
    nhwc = ops.to_nhwc(x)
    assert torch.equal(nhwc, x.permute(0, 2, 3, 1))
    assert torch.equal(ops.from_nhwc(nhwc.contiguous(), channels_last=False), x)


def test_cpu_tensor_is_rejected_loudly():
    from tomosar2height_amd.tile import TileIndex
    with pytest.raises(RuntimeError, match="no CPU path"):
        TileIndex(synth_cloud(10), 16)


def test_operator_level_dropins_reference_signatures():
    """scatter_mean / scatter_max / grid_sample with the reference's own call signatures (raw int64 index)."""
    from tomosar2height_amd import ops
    from oracle import c_oracle
    g = load_golden("pointnet_main_vector")                      # the reference's own vector, pointnet.py:114-123
    xy = torch.from_numpy(g["xy"]).to(_dev())
    index = ops.coordinate2index(xy, 2)
    out = ops.scatter_mean(xy.permute(0, 2, 1), index, out=xy.new_zeros(1, 2, 4))
    np.testing.assert_allclose(out.reshape(1, 2, 2, 2).cpu().numpy(), g["plane"], rtol=1e-6)

    gen = torch.Generator().manual_seed(5)
    cloud = synth_cloud(700, seed=9, batch=2)
    feat = (torch.randn(2, 700, 8, generator=gen) * 4).round() / 4
    idx = c_oracle.coordinate2index(cloud.numpy(), 16)
    want_v, want_a = c_oracle.scatter_max(feat.numpy(), idx, 256)
    val, arg = ops.scatter_max(feat.permute(0, 2, 1).to(_dev()), torch.from_numpy(idx).to(_dev()), dim_size=256)
    assert np.array_equal(val.cpu().numpy(), want_v) and np.array_equal(arg.cpu().numpy(), want_a)
    want_m = c_oracle.scatter_mean_fwd(feat.numpy(), idx, 16).reshape(2, 8, 256)
    got_m = ops.scatter_mean(feat.permute(0, 2, 1).to(_dev()), torch.from_numpy(idx).to(_dev()), dim_size=256)
    np.testing.assert_allclose(got_m.cpu().numpy(), want_m, rtol=1e-5, atol=1e-6)
    bad = torch.from_numpy(idx).to(_dev()).clone()
    bad[0, 0, 0] = 256
    ops.scatter_mean(feat.permute(0, 2, 1).to(_dev()), bad, dim_size=256)      # no synchronisation inside the operator:
    with pytest.raises(ValueError):                                            # the range check is reported afterwards
        ops.check_indices()

    gs = load_golden("grid_sample_points")
    for r in (8, 16):
        p = torch.from_numpy(gs[f"plane_r{r}"]).to(_dev()).requires_grad_(True)
        out = ops.grid_sample_points(p, torch.from_numpy(gs[f"p_r{r}"]).to(_dev()))
        np.testing.assert_allclose(out.detach().cpu().numpy(), gs[f"out_r{r}"], rtol=1e-5, atol=1e-6)
        out.backward(torch.from_numpy(gs[f"gout_r{r}"]).to(_dev()))
        np.testing.assert_allclose(p.grad.cpu().numpy(), gs[f"gplane_r{r}"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("c,reso,level", [(32, 256, 0), (128, 256, 1), (12, 64, 2), (512, 256, 3)])
def test_rasterise_mean_thru_sums_the_two_gradients_bit_exactly(c, reso, level):
    """ops.rasterise_mean_thru: the gradient of point features that feed both the rasterisation and another consumer
    (alto.py:123-130) is the same sum autograd would form with an extra add, bit for bit."""
    from tomosar2height_amd import ops
    g = torch.Generator().manual_seed(c)
    cloud = synth_cloud(7000, seed=3)
    t = _tile(cloud, reso)
    r = reso >> level
    feat = torch.randn(7000, c, generator=g).to(_dev())
    gplane = torch.randn(1, c, r, r, generator=g).to(_dev())
    gother = torch.randn(7000, c, generator=g).to(_dev())
    a = feat.clone().requires_grad_(True)
    plane_a = ops.rasterise_mean(t, a, r)
    torch.autograd.backward([plane_a, a * 1.0], [gplane, gother])               # autograd's own sum of the two gradients
    b = feat.clone().requires_grad_(True)
    plane_b, thru = ops.rasterise_mean_thru(t, b, r)
    assert thru.data_ptr() == b.data_ptr() and torch.equal(plane_a, plane_b)
    torch.autograd.backward([plane_b, thru * 1.0], [gplane, gother])
    assert torch.equal(a.grad, b.grad)
    # either consumer alone
    d = feat.clone().requires_grad_(True)
    plane_d, thru_d = ops.rasterise_mean_thru(t, d, r)
    plane_d.backward(gplane)
    e = feat.clone().requires_grad_(True)
    ops.rasterise_mean(t, e, r).backward(gplane)
    assert torch.equal(d.grad, e.grad)
    f = feat.clone().requires_grad_(True)
    (ops.rasterise_mean_thru(t, f, r)[1] * 1.0).backward(gother)
    assert torch.equal(f.grad, gother)


@pytest.mark.parametrize("c,reso,level,n", [(32, 256, 0, 9000), (64, 64, 1, 40000), (256, 256, 2, 60000), (12, 32, 0, 500)])
def test_sample_plane_thru_sums_the_two_gradients_bit_exactly(c, reso, level, n):
    """ops.sample_plane_thru: fine gather and coarse (cells + gather9) backward with the other consumer's gradient added
    in the final store == autograd's own sum."""
    from tomosar2height_amd import ops
    g = torch.Generator().manual_seed(c + n)
    t = _tile(synth_cloud(n, seed=5), reso)
    r = reso >> level
    plane = torch.randn(1, c, r, r, generator=g).to(_dev()).contiguous(memory_format=torch.channels_last)
    gs = torch.randn(n, c, generator=g).to(_dev())
    gother = torch.randn(1, c, r, r, generator=g).to(_dev()).contiguous(memory_format=torch.channels_last)
    a = plane.clone(memory_format=torch.preserve_format).requires_grad_(True)
    torch.autograd.backward([ops.sample_plane(t, a), a * 1.0], [gs, gother])
    b = plane.clone(memory_format=torch.preserve_format).requires_grad_(True)
    sampled, thru = ops.sample_plane_thru(t, b)
    # the alias keeps the strides exactly (B = 1 channels_last): torch.cat must still see a channels_last plane
    assert thru.stride() == b.stride() and torch.cat((thru, thru), 1).is_contiguous(memory_format=torch.channels_last)
    torch.autograd.backward([sampled, thru * 1.0], [gs, gother])
    assert torch.equal(a.grad, b.grad)
    d = plane.clone(memory_format=torch.preserve_format).requires_grad_(True)
    ops.sample_plane_thru(t, d)[0].backward(gs)
    e = plane.clone(memory_format=torch.preserve_format).requires_grad_(True)
    ops.sample_plane(t, e).backward(gs)
    assert torch.equal(d.grad, e.grad)


def test_maxpool_thru_sums_the_two_gradients_bit_exactly():
    from tomosar2height_amd import grid
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 32, 16, 24, generator=g).to(_dev()).contiguous(memory_format=torch.channels_last)
    gp = torch.randn(2, 32, 8, 12, generator=g).to(_dev()).contiguous(memory_format=torch.channels_last)
    gother = torch.randn(2, 32, 16, 24, generator=g).to(_dev()).contiguous(memory_format=torch.channels_last)
    a = x.clone(memory_format=torch.preserve_format).requires_grad_(True)
    torch.autograd.backward([grid.maxpool2x2(a), a * 1.0], [gp, gother])
    b = x.clone(memory_format=torch.preserve_format).requires_grad_(True)
    pooled, thru = grid.maxpool2x2_thru(b)
    assert thru.stride() == b.stride()
    x1 = x[:1].clone(memory_format=torch.preserve_format)
    assert grid.maxpool2x2_thru(x1)[1].stride() == x1.stride()
    torch.autograd.backward([pooled, thru * 1.0], [gp, gother])
    assert torch.equal(a.grad, b.grad) and torch.equal(pooled, grid.maxpool2x2(x))


@pytest.mark.parametrize("c,r", [(32, 256), (64, 256), (128, 128), (256, 64), (512, 32)])
def test_full_size_point_grid_ops_are_adjoint_pairs(c, r):
    """BASELINE.json config 2 sizes (N = 131072, every ALTO level shape): each linear operator and its hand-written
    backward satisfy <A x, y> == <x, A^T y> (float64 dot products) -- bilinear sample / its transposed-matrix or per-cell
    backward, mean rasterisation / its gather, mean pooling (its own adjoint)."""
    from tomosar2height_amd import ops
    from tomosar2height_amd.synthetic import berlin_tile
    g = torch.Generator().manual_seed(c + r)
    t = _tile(berlin_tile(3)["inputs"], 256)
    n = t.n_points

    def dot(a, b):
        return float((a.double() * b.double()).sum())

    def check(fn, x, y_shape, what):
        x = x.to(_dev()).requires_grad_(True)
        out = fn(x)
        y = torch.randn(y_shape, generator=g).to(_dev())
        out.backward(y)
        lhs, rhs = dot(out.detach(), y), dot(x.detach(), x.grad)
        assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0) + 1e-3, (what, lhs, rhs)

    plane = torch.randn(1, c, r, r, generator=g).contiguous(memory_format=torch.channels_last)
    check(lambda p: ops.sample_plane(t, p), plane, (n, c), "sample_plane")
    feat = torch.randn(n, c, generator=g)
    check(lambda f: ops.rasterise_mean(t, f, r, True), feat, (1, c, r, r), "rasterise_mean")
    if r == 256:
        check(lambda f: ops.pool_mean(t, f), feat, (n, c), "pool_mean")
