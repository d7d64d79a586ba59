"""-m gpu: the fused PointNet trunk block kernels (csrc/trunk.hip) -- pool_local in the loader + the three GEMMs of a
ResnetBlockFC in one launch -- against the unfused per-Linear path, the C oracle's scatter_max and the reference fixtures
(pointnet.py:72-82, 92-99; resnet.py:36-54)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from detinit import det_init_, synth_cloud

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _trunk_modules(seed=5):
    from tomosar2height_amd.encoder.pointnet import LocalPoolPointnet
    enc = LocalPoolPointnet(feature_dim=32, dim=3, hidden_dim=32, scatter_type="max", unet_type="alto",
                            unet_kwargs=dict(depth=2, merge_mode="concat", start_filts=8), plane_resolution=256)
    return det_init_(enc, seed=seed).to(_dev())


def _params(enc):
    ps = [enc.fc_pos.weight, enc.fc_pos.bias]
    for b in enc.blocks:
        ps += [b.fc_0.weight, b.fc_0.bias, b.fc_1.weight, b.fc_1.bias, b.shortcut.weight]
    return ps + [enc.fc_c.weight, enc.fc_c.bias]


def _clouds():
    g = torch.Generator().manual_seed(3)
    one_cell = torch.cat([0.5 + 0.003 * torch.rand(1, 700, 2, generator=g), torch.rand(1, 700, 1, generator=g)], 2)   # one cell, 6 tiles
    ties = synth_cloud(1500, seed=9)
    ties[:, :, 2] = (ties[:, :, 2] * 4).round() / 4                   # few distinct z values
    ties[:, 500:1000, :2] = ties[:, :500, :2]                          # duplicated positions -> equal features -> exact ties
    ties[:, 500:1000, 2] = ties[:, :500, 2]
    return {"ragged": synth_cloud(3001, seed=1), "tiny": synth_cloud(5, seed=2), "single": synth_cloud(1, seed=4),
            "one_cell": one_cell.float().contiguous(), "ties": ties, "big": synth_cloud(40000, seed=6)}


@pytest.mark.parametrize("name", ["ragged", "tiny", "single", "one_cell", "ties", "big"])
def test_fused_forward_vs_unfused_and_pool_exact(name):
    from tomosar2height_amd import _lib, mlp
    from tomosar2height_amd.tile import TileIndex
    enc = _trunk_modules()
    cloud = _clouds()[name].to(_dev())
    tile = TileIndex(cloud, 256)
    params = [p.detach() for p in _params(enc)]
    nb = len(enc.blocks)
    blocks = [params[2 + 5 * i: 7 + 5 * i] for i in range(nb)]
    c, nets, pooled_f, hrs, winners, cats = mlp._trunk_forward_fused(tile, tile.pts, params[0], params[1], blocks, params[-2],
                                                                     params[-1], want_x_full=True)
    # (1) the pooling inside the loader == t2h_pool_max_fwd on the same block output: values and arg-max bits exact
    for i in range(1, nb):
        net = nets[i - 1]
        assert torch.equal(net, cats[i][:, :32])
        pooled = torch.empty_like(net)
        win = torch.empty(net.shape[0], 8, dtype=torch.uint8, device=net.device)
        mlp._pool_fwd_(tile, net, pooled, win)
        assert torch.equal(pooled, cats[i][:, 32:]), f"pooled half of block {i}"
        assert torch.equal(pooled, pooled_f[i]), f"kept pooled half of block {i}"
        assert torch.equal(win, winners[i - 1]), f"winner bits of block {i}"
    # (2) the GEMM side against the unfused path (different k order inside the MFMA chain: fp32 rounding only)
    old = mlp.FUSED_TRUNK
    mlp.FUSED_TRUNK = False
    try:
        with torch.no_grad():
            want = mlp.point_trunk(tile, tile.pts, enc.fc_pos, enc.blocks, enc.fc_c)
    finally:
        mlp.FUSED_TRUNK = old
    scale = want.abs().max().item() + 1e-30
    assert (c - want).abs().max().item() <= 2e-5 * scale
    # (3) and against float64 arithmetic on the fused path's own inputs, block by block
    x = cats[0].double()
    w = [p.double() for p in params]
    np.testing.assert_allclose(x.cpu().numpy(), (tile.pts.double() @ w[0].t() + w[1]).cpu().numpy(), rtol=0, atol=1e-6)
    for i in range(nb):
        w0, b0, w1, b1, ws = (t.double() for t in blocks[i])
        xi = cats[i].double()
        hr = torch.relu(torch.relu(xi) @ w0.t() + b0)
        out = xi @ ws.t() + hr @ w1.t() + b1
        np.testing.assert_allclose(hrs[i].cpu().numpy(), hr.cpu().numpy(), rtol=0, atol=2e-6 * (hr.abs().max().item() + 1))
        got = nets[i]
        np.testing.assert_allclose(got.cpu().numpy(), out.cpu().numpy(), rtol=0, atol=2e-6 * (out.abs().max().item() + 1))


@pytest.mark.parametrize("name", ["ragged", "tiny", "single", "one_cell", "ties", "big", "benchmark", "dense_cells", "batch"])
@pytest.mark.parametrize("stride", [-1, 0, 96, 128, 40])
def test_one_launch_trunk_equals_the_per_block_launches_bit_for_bit(name, stride):
    """r06: t2h_trunk_fused_fwd (fc_pos -> 5 blocks with their 4 poolings -> fc_c in ONE launch, whole cells per workgroup,
    activations in LDS) against the five t2h_trunk_block_fwd launches: every tensor the backward reads -- hr and out of every
    block, the pooled halves, the winner bits -- and c, bit for bit.  Clouds: ragged sizes, fewer rows than a tile, one row, ONE
    CELL holding 700 rows and a tile of ~80 points per cell (work units longer than a tile: the block-by-block path inside the
    launch), exact ties, the benchmark tile, a ragged batch of three tiles; units packed greedily (default) or by strides 96, 128, 40."""
    from tomosar2height_amd import mlp
    from tomosar2height_amd.synthetic import berlin_tile
    from tomosar2height_amd.tile import TileIndex
    enc = _trunk_modules(seed=13)
    reso = 256
    if name == "benchmark":
        cloud = berlin_tile(seed=3, n_points=131072)["inputs"].to(_dev())
    elif name == "dense_cells":
        cloud, reso = synth_cloud(20000, seed=8).to(_dev()), 16
    elif name == "batch":
        cloud = [synth_cloud(n, seed=20 + i).to(_dev()) for i, n in enumerate((5000, 777, 12001))]
    else:
        cloud = _clouds()[name].to(_dev())
    tile = TileIndex(cloud, reso)
    params = [p.detach() for p in _params(enc)]
    nb = len(enc.blocks)
    blocks = [params[2 + 5 * i: 7 + 5 * i] for i in range(nb)]
    old = (mlp._TRUNK_FUSED, mlp._TRUNK_FUSED_STRIDE, mlp._TRUNK_UNIT_BOUNDS)
    old_min, mlp._TRUNK_FUSED_MIN_ROWS = mlp._TRUNK_FUSED_MIN_ROWS, 0
    try:
        # stride -1: the default -- units packed greedily once per tile index (t2h_trunk_units_build); >= 0: fixed windows of
        # `stride` rows (0: 96) snapped to cell boundaries and looked up inside the launch
        mlp._TRUNK_FUSED, mlp._TRUNK_FUSED_STRIDE, mlp._TRUNK_UNIT_BOUNDS = True, max(stride, 0), stride < 0
        if stride < 0:
            used = tile.trunk_unit_list().cpu()
            assert bool((used[:, 1] > used[:, 0]).all())
            assert int(used[0, 0]) == 0 and int(used[-1, 1]) == tile.pts.shape[0] and torch.equal(used[1:, 0], used[:-1, 1])
            starts = set(tile.off0.cpu().tolist())
            assert all(int(v) in starts for v in used[:, 0]), "a unit starts inside a cell"
        one = mlp._trunk_forward_fused(tile, tile.pts, params[0], params[1], blocks, params[-2], params[-1])
        mlp._TRUNK_FUSED = False
        per = mlp._trunk_forward_fused(tile, tile.pts, params[0], params[1], blocks, params[-2], params[-1])
    finally:
        mlp._TRUNK_FUSED, mlp._TRUNK_FUSED_STRIDE, mlp._TRUNK_UNIT_BOUNDS = old
        mlp._TRUNK_FUSED_MIN_ROWS = old_min
    names = ("c", "nets", "pooled", "hrs", "winners")
    assert torch.equal(one[0], per[0]), "c"
    for what, a, b in zip(names[1:], one[1:], per[1:]):
        assert len(a) == len(b)
        for i, (x, y) in enumerate(zip(a, b)):
            if x is None or y is None:
                assert x is None and y is None
                continue
            assert torch.equal(x, y), f"{what}[{i}] differs ({int((x != y).sum())} of {x.numel()} elements)"


def test_fused_pool_matches_c_oracle_scatter_max():
    """The in-loader pooling against the oracle's restated torch_scatter semantics (first maximum wins, pointnet.py:92-99)."""
    from oracle import scatter_ref
    from tomosar2height_amd import mlp
    from tomosar2height_amd.tile import TileIndex
    enc = _trunk_modules(seed=11)
    cloud = _clouds()["ties"].to(_dev())
    tile = TileIndex(cloud, 256)
    params = [p.detach() for p in _params(enc)]
    blocks = [params[2 + 5 * i: 7 + 5 * i] for i in range(len(enc.blocks))]
    _, nets, pooled_f, _, winners = mlp._trunk_forward_fused(tile, tile.pts, params[0], params[1], blocks, params[-2], params[-1])
    net = nets[0].cpu()                                           # sorted rows; the sort is stable inside a cell
    idx = tile.cell.cpu().long()
    out, arg = scatter_ref.scatter_max(net.t()[None], idx[None, None], dim=-1, dim_size=256 * 256)
    want = out[0].t()[idx]                                        # gather back to the points (pointnet.py:98)
    assert torch.equal(pooled_f[1].cpu(), want)
    rows = torch.arange(net.shape[0])
    is_win = (arg[0].t()[idx] == rows[:, None])                   # [M, 32] bool
    bits = winners[0].cpu()
    got = torch.stack([(bits[:, c // 4] >> (c % 4)) & 1 for c in range(32)], 1).bool()
    assert torch.equal(got, is_win)


def test_trunk_gradients_fused_vs_unfused():
    from tomosar2height_amd import mlp
    from tomosar2height_amd.tile import TileIndex
    cloud = _clouds()["ragged"].to(_dev())
    gout = torch.randn(cloud.shape[1], 32, generator=torch.Generator().manual_seed(2)).to(_dev())
    grads = {}
    for fused in (True, False):
        enc = _trunk_modules()
        tile = TileIndex(cloud, 256)
        old = mlp.FUSED_TRUNK
        mlp.FUSED_TRUNK = fused
        try:
            out = mlp.point_trunk(tile, tile.pts, enc.fc_pos, enc.blocks, enc.fc_c)
            out.backward(gout)
        finally:
            mlp.FUSED_TRUNK = old
        grads[fused] = {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None and not k.startswith("unet")}
    assert grads[True].keys() == grads[False].keys() and len(grads[True]) == 2 + 5 * 5 + 2
    for k, g in grads[False].items():
        scale = g.abs().max().item() + 1e-30
        # relu / arg-max masks can flip under the fp32 re-association between the two paths: max-norm 1e-3, L2 3e-4
        assert (grads[True][k] - g).abs().max().item() <= 1e-3 * scale, k
        assert ((grads[True][k] - g).norm() / (g.norm() + 1e-30)).item() <= 3e-4, k


@pytest.mark.parametrize("tag", ["64_32", "32_32"])
def test_resblock_golden(tag):
    """mlp.resblock (block/resnet.py:36-54) straight against the reference fixture: shortcut (64 -> 32) and identity
    (32 -> 32) branches, output and every gradient."""
    from tomosar2height_amd import mlp
    g = load_golden("resnet_block_fc")
    dev = _dev()
    x = torch.from_numpy(g[f"x_{tag}"]).to(dev).requires_grad_(True)
    names = ["fc_0.weight", "fc_0.bias", "fc_1.weight", "fc_1.bias"] + (["shortcut.weight"] if f"w_{tag}.shortcut.weight" in g.files else [])
    ws = [torch.from_numpy(g[f"w_{tag}.{n}"]).to(dev).requires_grad_(True) for n in names]
    out = mlp.resblock(x, ws[0], ws[1], ws[2], ws[3], ws[4] if len(ws) == 5 else None)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"y_{tag}"], rtol=1e-5, atol=1e-6)
    out.backward(torch.from_numpy(g[f"gy_{tag}"]).to(dev))
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"gx_{tag}"], rtol=1e-4, atol=1e-6)
    for n, w in zip(names, ws):
        np.testing.assert_allclose(w.grad.cpu().numpy(), g[f"g_{tag}.{n}"], rtol=1e-4, atol=2e-6, err_msg=n)


@pytest.mark.parametrize("name", ["ragged", "one_cell", "ties", "tiny", "big"])
def test_fused_backward_vs_float64_with_the_same_masks(name):
    """Every gradient of the fused backward against float64 arithmetic that uses the SAME ReLU masks and arg-max bits (so
    no mask can flip): isolates the kernel's own arithmetic -- GEMMs, the pooling backward folded into the loader, the
    slab reduction, the fc_pos / fc_c ends -- at 1e-5."""
    from tomosar2height_amd import mlp
    from tomosar2height_amd.tile import TileIndex
    enc = _trunk_modules(seed=21)
    cloud = _clouds()[name].to(_dev())
    tile = TileIndex(cloud, 256)
    params = [p.detach() for p in _params(enc)]
    nb = len(enc.blocks)
    blocks = [params[2 + 5 * i: 7 + 5 * i] for i in range(nb)]
    c, nets, pooled_f, hrs, winners, cats = mlp._trunk_forward_fused(tile, tile.pts, params[0], params[1], blocks, params[-2],
                                                                     params[-1], want_x_full=True)
    g_out = torch.randn(c.shape, generator=torch.Generator().manual_seed(5)).to(_dev())
    got = mlp._trunk_backward_fused(tile, tile.pts, params, nets, pooled_f, hrs, winners, g_out)
    assert all(g is not None for g in got)

    d = torch.float64
    w = [p.to(d) for p in params]
    want = [None] * len(params)
    go = g_out.to(d)
    out_last = nets[-1].to(d)
    want[-2], want[-1] = go.t() @ torch.relu(out_last), go.sum(0)
    g = (go @ w[-2]) * (out_last > 0)
    cell = tile.cell.long()
    uniq, inv = torch.unique_consecutive(cell, return_inverse=True)
    for i in range(nb - 1, -1, -1):
        w0, b0, w1, b1, ws = w[2 + 5 * i: 7 + 5 * i]
        x, hr = cats[i].to(d), hrs[i].to(d)
        dhr = (g @ w1) * (hr > 0)
        want[2 + 5 * i: 7 + 5 * i] = [dhr.t() @ torch.relu(x), dhr.sum(0), g.t() @ hr, g.sum(0), g.t() @ x]
        dx = g @ ws + (dhr @ w0) * (x > 0)
        if i > 0:
            sums = torch.zeros(len(uniq), 32, dtype=d, device=dx.device).index_add_(0, inv, dx[:, 32:])
            bits = winners[i - 1]
            mask = torch.stack([(bits[:, ch // 4] >> (ch % 4)) & 1 for ch in range(32)], 1).to(d)
            g = dx[:, :32] + mask * sums[inv]
        else:
            want[0], want[1] = dx.t() @ tile.pts.to(d), dx.sum(0)
    for k, (a, b) in enumerate(zip(got, want)):
        scale = b.abs().max().item() + 1e-30
        assert (a.to(d) - b).abs().max().item() <= 1e-5 * scale, f"param {k}: {(a.to(d) - b).abs().max().item() / scale:.2e}"


def test_trunk_forwards_behind_poisoned_lds():
    """r06: t2h_debug_poison_lds (NaN patterns into the whole LDS of every free CU) in front of both trunk forwards: the results are
    those of the clean run bit for bit -- neither form reads LDS it has not written.  (What the poison cannot show is a read that
    usually comes after the write anyway: the missing barrier of r06_coresidency.txt section 8 passes this test too; that one is
    pinned by the stalled windows of test_coresidency.py.)"""
    from tomosar2height_amd import _lib, mlp
    from tomosar2height_amd.tile import TileIndex
    enc = _trunk_modules(seed=13)
    tile = TileIndex(synth_cloud(30000, seed=4).to(_dev()), 256)
    params = [p.detach() for p in _params(enc)]
    blocks = [params[2 + 5 * i: 7 + 5 * i] for i in range(len(enc.blocks))]
    lib = _lib.load()
    old = mlp._TRUNK_FUSED, mlp._TRUNK_FUSED_MIN_ROWS
    try:
        for fused in (True, False):
            mlp._TRUNK_FUSED, mlp._TRUNK_FUSED_MIN_ROWS = fused, 0
            clean = mlp._trunk_forward_fused(tile, tile.pts, params[0], params[1], blocks, params[-2], params[-1])
            torch.cuda.synchronize()
            for _ in range(3):
                assert lib.t2h_debug_poison_lds(_lib.stream()) == 0
                got = mlp._trunk_forward_fused(tile, tile.pts, params[0], params[1], blocks, params[-2], params[-1])
                torch.cuda.synchronize()
                for a, b in zip(clean, got):
                    for x, y in zip(a if isinstance(a, (list, tuple)) else [a], b if isinstance(b, (list, tuple)) else [b]):
                        assert (x is None and y is None) or torch.equal(x, y)
    finally:
        mlp._TRUNK_FUSED, mlp._TRUNK_FUSED_MIN_ROWS = old
