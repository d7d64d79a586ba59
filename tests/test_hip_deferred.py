"""-m gpu: the entry points of the deferred ALTO point update (include/t2h.h: t2h_sample_fwd_relu + sign bits, t2h_segsum_fwd,
t2h_plane_sumpool2x2, t2h_segsum_bwd_multi, t2h_sample_bwd_from_sums, t2h_cell_counts, t2h_mean_bias_fwd / _bwd) one by one
against plain torch restatements of what they replace (reference: alto.py:76-95, 121-130: grid_sample, ReLU, scatter_add,
count clamp, division), through the C ABI.  Whole-level and whole-network checks of the same path: test_hip_masks.py,
test_full_size_vs_oracle.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from detinit import synth_cloud

pytestmark = pytest.mark.gpu
D = torch.float64


def _dev():
    return torch.device("cuda:0")


def _tile(n=40000, seed=3, batch=1):
    from tomosar2height_amd.tile import TileIndex
    return TileIndex(synth_cloud(n, seed=seed, batch=batch).to(_dev()), 256)


def _cell_index(tile, level):
    """Row-major cell index of every sorted point at ALTO level ``level`` (coordinate.py:12-28 at resolution R >> level)."""
    r = tile.R >> level
    xy = tile.pts[:, :2]
    ix, iy = (xy[:, 0] * r).long(), (xy[:, 1] * r).long()
    b = torch.arange(tile.B, device=xy.device).repeat_interleave(tile.N)
    return (b * r + iy) * r + ix, r


@pytest.mark.parametrize("c,r", [(256, 64), (1024, 32), (128, 128)])
def test_sample_relu_and_sign_bits(c, r):
    """relu(grid_sample(plane)) and, for C % 256 == 0, the packed sign pattern (bit l of word j of chunk q <=> channel
    256 q + 4 l + j of that row is > 0)."""
    from tomosar2height_amd import _lib, ops
    tile = _tile()
    g = torch.Generator().manual_seed(1)
    plane = torch.randn(1, r, r, c, generator=g).to(_dev())                      # NHWC
    want = torch.relu(ops.sample_plane(tile, plane.permute(0, 3, 1, 2)))          # the plain sample kernel (oracle-tested)
    h = torch.empty(tile.n_points, c, device=_dev())
    bits = torch.zeros(tile.n_points * (c // 256) * 4, dtype=torch.int64, device=_dev()) if c % 256 == 0 else None
    _lib.call("t2h_sample_fwd_relu", _lib.ptr(plane), _lib.ptr(tile.pts), tile.dim, tile.B, tile.N, r, c, _lib.ptr(h),
              None if bits is None else _lib.ptr(bits), _lib.stream())
    assert torch.equal(h, want)
    if bits is not None:
        w = bits.view(c // 256, tile.n_points, 4).cpu().numpy().view(np.uint64)       # chunk-major: [C / 256][rows][4]
        got = np.zeros((tile.n_points, c), bool)
        for j in range(4):
            for lane in range(64):
                got[:, lane * 4 + j::256] = ((w[:, :, j].T >> np.uint64(lane)) & np.uint64(1)).astype(bool)
        assert np.array_equal(got, (h > 0).cpu().numpy())


def test_cell_sums_pooling_and_counts():
    """Per-cell sums into a column block of a wider matrix (row stride), their 2x2 pooling down the levels and the counts."""
    from tomosar2height_amd import deferred
    tile = _tile(batch=2)
    c, ktot, off = 64, 160, 32
    rows = torch.randn(tile.n_points, c, generator=torch.Generator().manual_seed(2)).to(_dev())
    mats = {lv: torch.full((tile.B * (256 >> lv) ** 2, ktot), 7.0, device=_dev()) for lv in range(4)}
    deferred._segsum_into(tile, rows, 0, mats[0][:, off:off + c])
    for lv in range(3):
        deferred._sumpool_into(tile, mats[lv][:, off:off + c], lv, mats[lv + 1][:, off:off + c])
    for lv in range(4):
        idx, r = _cell_index(tile, lv)
        want = torch.zeros(tile.B * r * r, c, dtype=D, device=_dev()).index_add_(0, idx, rows.to(D))
        got = mats[lv][:, off:off + c].to(D)
        assert ((got - want).abs().max() / want.abs().max()).item() <= 2e-6, lv
        assert torch.all(mats[lv][:, :off] == 7.0) and torch.all(mats[lv][:, off + c:] == 7.0), "wrote outside its block"
        cnt = torch.zeros(tile.B * r * r, device=_dev()).index_add_(0, idx, torch.ones(tile.n_points, device=_dev()))
        assert torch.equal(deferred.counts(tile, lv), cnt)


@pytest.mark.parametrize("with_mask", [False, True])
def test_multi_level_gather(with_mask):
    """d rows = (mask > 0) * sum_l plane_l[cell_l(n)] from column blocks with a row stride."""
    from tomosar2height_amd import deferred
    tile = _tile(batch=2)
    c, ktot, off = 64, 96, 16
    g = torch.Generator().manual_seed(4)
    planes = [(torch.randn(tile.B * (256 >> lv) ** 2, ktot, generator=g).to(_dev())[:, off:off + c], lv) for lv in (0, 2, 3)]
    mask = torch.randn(tile.n_points, c, generator=g).to(_dev()) if with_mask else None
    got = deferred._gather(tile, planes, c, mask=mask)
    want = torch.zeros(tile.n_points, c, dtype=D, device=_dev())
    for p, lv in planes:
        want += p.to(D)[_cell_index(tile, lv)[0]]
    if with_mask:
        want = want * (mask > 0)
    assert ((got.to(D) - want).abs().max() / want.abs().max()).item() <= 1e-6


def _dense_tile(n=70000, seed=4):
    """Every point inside ONE cell of the 32 x 32 level (the densest building of a real tile, taken to the extreme), a few in
    the far corner: one wave-quarter walks ~all rows, most cells and children are empty."""
    from tomosar2height_amd.tile import TileIndex
    g = torch.Generator().manual_seed(seed)
    pts = torch.rand(1, n, 3, generator=g)
    pts[..., :2] = (pts[..., :2] + torch.tensor([11.0, 20.0])) / 32.0
    pts[0, :7, :2] = torch.rand(7, 2, generator=g) * 0.01 + 0.985
    return TileIndex(pts.to(_dev()), 256)


@pytest.mark.parametrize("c2,r,use_bits,n,batch", [
    (256, 64, True, 60000, 1), (256, 64, False, 60000, 1), (512, 32, True, 60000, 1), (128, 64, False, 60000, 1),
    # the row walk with the cell's children split over the four waves (>= 64 rows per cell), two tiles, few rows per cell at
    # r = 128 (2 x 2 blocks of cells, one level of children), everything in one cell
    (512, 32, True, 140000, 1), (256, 32, True, 90000, 2), (256, 128, True, 150000, 1), (256, 32, True, -1, 1), (256, 64, True, -1, 1),
    # five gradient planes (a deeper U-Net): the walk holds four, so the matrix-core partials with per-row gathers take it
    (256, 32, True, 80000, -5)])
def test_fused_sample_adjoint_from_sums(c2, r, use_bits, n, batch):
    """t2h_sample_bwd_from_sums == gather + mask, then the plain sample adjoint (both paths of the library), and == the float64
    adjoint of F.grid_sample applied to the masked gather."""
    from tomosar2height_amd import _lib, deferred, ops
    five_planes = batch == -5
    batch = abs(batch) if not five_planes else 1
    tile = _dense_tile() if n < 0 else _tile(n=n, batch=batch)
    g = torch.Generator().manual_seed(5)
    level = tile.level(r)
    levels = [lv for lv in (0, 1, 2, 3) if lv <= level + 1] + ([4] if five_planes else [])
    planes = [(torch.randn(tile.B * (256 >> lv) ** 2, c2, generator=g).to(_dev()), lv) for lv in levels]
    h = torch.relu(torch.randn(tile.n_points, c2, generator=g)).to(_dev())
    ws_bytes = _lib.load().t2h_sample_bwd_workspace_bytes(tile.B, tile.N, tile.nbits, level, c2)
    assert ws_bytes > 0, "this shape should take the per-cell partials"
    mask_arg = h
    if use_bits:
        q = torch.randn(tile.B, r, r, c2, generator=g).to(_dev())
        h = torch.empty(tile.n_points, c2, device=_dev())
        bits = torch.empty(tile.n_points * (c2 // 256) * 4, dtype=torch.int64, device=_dev())
        _lib.call("t2h_sample_fwd_relu", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, tile.B, tile.N, r, c2, _lib.ptr(h),
                  _lib.ptr(bits), _lib.stream())
        mask_arg = bits
    arr, lvs, lds = deferred._plane_args(planes)
    ws = _lib.workspace(ws_bytes, _dev())
    got = torch.empty(tile.B * r * r, c2, device=_dev())
    _lib.call("t2h_sample_bwd_from_sums", arr, lvs, lds, len(planes), _lib.ptr(tile.cell), _lib.ptr(mask_arg), int(use_bits),
              _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits, level, c2, _lib.ptr(got),
              _lib.ptr(ws), ws_bytes, _lib.stream())
    dh = deferred._gather(tile, planes, c2, mask=h)
    two_pass = ops._sample_bwd(tile, dh, r, c2, None).reshape(tile.B * r * r, c2)
    scale = two_pass.abs().max().item()
    # (a dense cell sums thousands of rows per pixel: the two summation orders differ by more fp32 roundings)
    assert (got - two_pass).abs().max().item() <= (2e-6 if n >= 0 else 2e-5) * scale
    # float64 adjoint of the reference's grid_sample (alto.py:90-95)
    plane64 = torch.zeros(tile.B, c2, r, r, dtype=D, device=_dev(), requires_grad=True)
    vgrid = (2.0 * tile.pts[:, :2].to(D) - 1.0).reshape(tile.B, tile.N, 1, 2)
    out = F.grid_sample(plane64, vgrid, mode="bilinear", padding_mode="border", align_corners=True)[..., 0].permute(0, 2, 1)
    out.backward(dh.to(D).reshape(tile.B, tile.N, c2))
    want = plane64.grad.permute(0, 2, 3, 1).reshape(tile.B * r * r, c2)
    assert ((got.to(D) - want).abs().max() / want.abs().max()).item() <= (5e-6 if n >= 0 else 2e-5)


def test_mean_bias_forward_and_backward():
    """raster = acc / max(count, 1) + [count > 0] * const and its gradients (alto.py:76-88: scatter_mean's clamp)."""
    g = torch.Generator().manual_seed(6)
    p, c = 4096, 64
    acc = torch.randn(p, c, generator=g).to(_dev())
    const = torch.randn(1, c, generator=g).to(_dev())
    cnt = torch.randint(0, 4, (p,), generator=g).float().to(_dev())
    from tomosar2height_amd import _lib
    out = torch.empty(p, c, device=_dev())
    _lib.call("t2h_mean_bias_fwd", _lib.ptr(acc.detach()), _lib.ptr(cnt), _lib.ptr(const.detach()), p, c, _lib.ptr(out), _lib.stream())
    up = torch.randn(p, c, generator=g).to(_dev())
    dacc, dconst = torch.empty(p, c, device=_dev()), torch.empty(1, c, device=_dev())
    ws_bytes = _lib.load().t2h_mean_bias_bwd_workspace_bytes(p, c)
    ws = _lib.workspace(ws_bytes, _dev())
    _lib.call("t2h_mean_bias_bwd", _lib.ptr(up), _lib.ptr(cnt), p, c, _lib.ptr(dacc), _lib.ptr(dconst), _lib.ptr(ws), ws_bytes,
              _lib.stream())
    a64, c64 = acc.detach().to(D).requires_grad_(True), const.detach().to(D).requires_grad_(True)
    want = a64 / cnt.to(D).clamp_min(1.0)[:, None] + (cnt > 0).to(D)[:, None] * c64
    assert ((out.to(D) - want).abs().max() / want.abs().max()).item() <= 1e-6
    want.backward(up.to(D))
    assert ((dacc.to(D) - a64.grad).abs().max() / a64.grad.abs().max()).item() <= 1e-6
    assert ((dconst.to(D) - c64.grad).abs().max() / c64.grad.abs().max()).item() <= 1e-5


def test_compose_stack_matches_matmul():
    """A_k = [A_{k-1} Wc^T ; W1^T] and its three gradients against float64 matmuls."""
    from tomosar2height_amd import deferred
    g = torch.Generator().manual_seed(7)
    a_prev = torch.randn(384, 128, generator=g).to(_dev()).requires_grad_(True)
    wc = (torch.randn(256, 128, generator=g) * 0.1).to(_dev()).requires_grad_(True)
    w1 = (torch.randn(256, 512, generator=g) * 0.1).to(_dev()).requires_grad_(True)
    out = deferred._ComposeStack.apply(a_prev, wc, w1)
    up = torch.randn(out.shape, generator=g).to(_dev())
    out.backward(up)
    a64, c64, w64 = (t.detach().to(D).requires_grad_(True) for t in (a_prev, wc, w1))
    want = torch.cat([a64 @ c64.t(), w64.t()], 0)
    want.backward(up.to(D))
    for name, got, ref in (("out", out, want), ("d a_prev", a_prev.grad, a64.grad), ("d wc", wc.grad, c64.grad), ("d w1", w1.grad, w64.grad)):
        assert ((got.to(D) - ref).abs().max() / ref.abs().max()).item() <= 2e-6, name
    first = deferred._ComposeStack.apply(None, wc.detach(), w1.detach())          # the base tensor's map is the identity
    assert torch.equal(first, torch.cat([wc.detach().t(), w1.detach().t()], 0))


@pytest.mark.parametrize("c,r,sum_level,dense", [(256, 64, 0, False), (1024, 32, 0, False), (512, 32, 1, False),
                                                (256, 32, 3, False), (256, 32, 0, True), (256, 128, 0, True)])
def test_on_chip_hidden_activations_equal_the_two_kernel_form(c, r, sum_level, dense):
    """t2h_sample_relu_cellsums (hidden activations never written) == t2h_sample_fwd_relu + t2h_segsum_fwd, bit for bit: the
    per-cell sums at `sum_level` inside a column block, and the packed sign bits.  dense: all points in one 32 x 32 cell."""
    from tomosar2height_amd import _lib, deferred
    tile = _dense_tile() if dense else _tile(n=70000, batch=2)
    g = torch.Generator().manual_seed(8)
    q = torch.randn(tile.B, r, r, c, generator=g).to(_dev())
    h = torch.empty(tile.n_points, c, device=_dev())
    bits_ref = torch.zeros(tile.n_points * (c // 256) * 4, dtype=torch.int64, device=_dev())
    _lib.call("t2h_sample_fwd_relu", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, tile.B, tile.N, r, c, _lib.ptr(h),
              _lib.ptr(bits_ref), _lib.stream())
    rs = 256 >> sum_level
    ktot, off = c + 64, 32
    want = torch.full((tile.B * rs * rs, ktot), 3.0, device=_dev())
    deferred._segsum_into(tile, h, sum_level, want[:, off:off + c])
    got = torch.full((tile.B * rs * rs, ktot), 3.0, device=_dev())
    bits = torch.zeros_like(bits_ref)
    blk = got[:, off:off + c]
    _lib.call("t2h_sample_relu_cellsums", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N,
              tile.nbits, tile.level(r), sum_level, c, blk.data_ptr(), blk.stride(0), _lib.ptr(bits), _lib.stream())
    assert torch.equal(bits, bits_ref)
    if sum_level < tile.level(r):
        # the same call with the pooled sums one level up: == t2h_plane_sumpool2x2 of its own finest sums, bit for bit
        got2 = torch.full((tile.B * rs * rs, ktot), 3.0, device=_dev())
        rs2 = rs // 2
        pooled = torch.full((tile.B * rs2 * rs2, ktot), 5.0, device=_dev())
        b2, p2 = got2[:, off:off + c], pooled[:, off:off + c]
        _lib.call("t2h_sample_relu_cellsums2", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N,
                  tile.nbits, tile.level(r), sum_level, c, b2.data_ptr(), b2.stride(0), p2.data_ptr(), p2.stride(0),
                  _lib.ptr(torch.zeros_like(bits_ref)), _lib.stream())
        assert torch.equal(got2, got)
        want2 = torch.full((tile.B * rs2 * rs2, ktot), 5.0, device=_dev())
        deferred._sumpool_into(tile, got[:, off:off + c], sum_level, want2[:, off:off + c])
        assert torch.equal(pooled, want2)
    if _lib.load().t2h_segmean_workspace_bytes(tile.B, tile.N, tile.nbits, sum_level, c) == 0:
        assert torch.equal(got, want)             # the two-kernel form sums a cell's rows in sequence too: same bits
    else:                                         # ... unless it takes the per-(cell, split) partials there: same sums, re-associated
        assert torch.equal(got[:, :off], want[:, :off]) and torch.equal(got[:, off + c:], want[:, off + c:])
        assert ((got - want).abs().max() / want.abs().max()).item() <= 2e-6


@pytest.mark.parametrize("c,r,sum_level,batch", [(1024, 32, 0, 1), (512, 64, 0, 1), (256, 128, 0, 1), (256, 64, 1, 2), (256, 128, 1, 2)])
def test_cell_order_is_longest_first_and_changes_no_bit(c, r, sum_level, batch):
    """t2h_cell_order_build: the level's cells, then its 2 x 2 blocks of cells, each a permutation by FALLING row count; the
    on-chip walks launched with it (dense cells' workgroups first) give the bits of the launch without it -- forward (per-cell
    sums, pooled sums, sign words), backward (dQ) -- and both forms of the r05 forward (a cell shared by four waves / a
    2 x 2 block of cells per workgroup) give the bits of the r04 kernel."""
    import os
    from tomosar2height_amd import _lib, deferred
    tile = _tile(n=150000 if r == 128 else 90000, batch=batch)
    lv = tile.level(r)
    cells = tile.B * r * r
    order = tile.cell_order(lv)
    assert order.numel() == _lib.load().t2h_cell_order_len(tile.B, tile.nbits, lv) == cells + cells // 4
    for lst, shift in ((order[:cells], lv), (order[cells:], lv + 1)):
        n = lst.numel()
        assert torch.equal(lst.sort().values, torch.arange(n, device=_dev(), dtype=torch.int32))
        bounds = tile.off0[::4 ** shift]
        rows = (bounds[1:] - bounds[:-1])[lst.long()]
        quantum = max(1, (tile.n_points // n + 63) // 64)         # the sort's key: rows / quantum, capped at 2047
        key = (rows // quantum).clamp(max=2047)
        assert bool((key[1:] <= key[:-1]).all()) and int(rows.sum()) == tile.n_points
    g = torch.Generator().manual_seed(9)
    q = torch.randn(tile.B, r, r, c, generator=g).to(_dev())
    rs = 256 >> sum_level

    def forward(order_arg, variant):
        os.environ["T2H_CELLSUMS_V2"] = variant
        try:
            sums = torch.full((tile.B * rs * rs, c), 3.0, device=_dev())
            pooled = torch.full((tile.B * (rs // 2) ** 2, c), 5.0, device=_dev())
            bits = torch.zeros(tile.n_points * (c // 256) * 4, dtype=torch.int64, device=_dev())
            _lib.call("t2h_sample_relu_cellsums_ordered", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B,
                      tile.N, tile.nbits, lv, sum_level, c, _lib.ptr(sums), c, _lib.ptr(pooled) if sum_level < lv else None, c,
                      _lib.ptr(bits), None if order_arg is None else _lib.ptr(order_arg), _lib.stream())
            return sums, pooled, bits
        finally:
            os.environ.pop("T2H_CELLSUMS_V2", None)

    ref = forward(None, "0")                                     # the r04 kernel (one wave per chunk, its own neighbourhood)
    for variant in ("1", "2"):                                   # r05: shared by four waves / 2 x 2 blocks of cells
        for o in (None, order):
            got = forward(o, variant)
            assert all(torch.equal(a, b) for a, b in zip(ref, got)), (variant, o is not None)
    planes = [(torch.randn(tile.B * (256 >> l) ** 2, c, generator=g).to(_dev()), l) for l in range(lv + 1)]
    arr, lvs, lds = deferred._plane_args(planes)
    ws_bytes = _lib.load().t2h_sample_bwd_workspace_bytes(tile.B, tile.N, tile.nbits, lv, c)
    assert ws_bytes > 0
    ws = _lib.workspace(ws_bytes, _dev())
    outs = []
    for o in (None, order, order):
        dq = torch.full((tile.B * r * r, c), float("nan"), device=_dev())
        _lib.call("t2h_sample_bwd_from_sums_ordered", arr, lvs, lds, len(planes), _lib.ptr(tile.cell), _lib.ptr(ref[2]), 1,
                  _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits, lv, c, _lib.ptr(dq), _lib.ptr(ws),
                  ws_bytes, None if o is None else _lib.ptr(o), _lib.stream())
        outs.append(dq)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]) and bool(torch.isfinite(outs[0]).all())
