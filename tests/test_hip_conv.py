"""GPU parity of the implicit-GEMM 3x3 convolutions (csrc/conv.hip, SURVEY 8f-1 second step) and of the 1x1 convolutions
on the per-point GEMM kernels.  The reference's operator here is torch's own nn.Conv2d (alto.py:59-61,157-182;
pixel.py:20-32), so the checker is F.conv2d / its autograd evaluated on the CPU in float64 on the same inputs.
Tolerances: fp32 products with fp32 accumulation over K = 9*C <= 4608 terms (and over up to 2^18 pixels for the
weight gradient) against float64: 2e-5 of the result's max-norm."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _cl(t):
    return t.to(_dev()).contiguous(memory_format=torch.channels_last)


def _close(got, want, tol=2e-5):
    want = want.to(torch.float64)
    scale = want.abs().max().item() + 1e-30
    err = (got.detach().cpu().to(torch.float64) - want).abs().max().item()
    assert got.shape == want.shape and err <= tol * scale, (err / scale, tuple(got.shape))


# (B, H, W, Cin, Cout): decoder-like (many pixels, few channels), ALTO bottom levels (few pixels, many channels: split
# reduction), non-square, batch > 1 (image borders inside a row tile), N tiles of 32 / 64 / 128 and a ragged one (Cout=80)
SHAPES = [(1, 64, 64, 32, 32), (1, 32, 32, 64, 128), (2, 16, 16, 128, 64), (1, 8, 8, 512, 256), (3, 4, 8, 16, 16),
          (1, 16, 64, 48, 80), (1, 128, 128, 32, 64), (2, 2, 2, 32, 32), (1, 1, 1, 16, 16)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("relu", [False, True])
def test_conv3x3_forward(shape, relu):
    from tomosar2height_amd import grid
    b, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x, wt, bias = torch.randn(b, cin, h, w, generator=g), torch.randn(cout, cin, 3, 3, generator=g) * 0.1, torch.randn(cout, generator=g)
    want = F.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    want = F.relu(want) if relu else want
    y = grid._empty_cl(b, cout, h, w, _dev())
    grid.conv3x3_fwd_(_cl(x), _cl(wt), bias.to(_dev()), y, relu=relu)
    _close(y, want)
    # accumulate flag: y += act(conv)
    base = torch.randn(b, cout, h, w, generator=g)
    y2 = _cl(base.clone())
    grid.conv3x3_fwd_(_cl(x), _cl(wt), None, y2, relu=relu, accumulate=True)
    want2 = F.conv2d(x.double(), wt.double(), None, padding=1)
    _close(y2, base.double() + (F.relu(want2) if relu else want2))


@pytest.mark.parametrize("shape", SHAPES)
def test_conv3x3_data_and_weight_gradient(shape):
    from tomosar2height_amd import grid
    b, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(7 + sum(shape))
    x = torch.randn(b, cin, h, w, generator=g).double().requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).double().requires_grad_(True)
    bias = torch.randn(cout, generator=g).double().requires_grad_(True)
    gy = torch.randn(b, cout, h, w, generator=g)
    F.conv2d(x, wt, bias, padding=1).backward(gy.double())
    xd, wd, gyd = _cl(x.detach().float()), _cl(wt.detach().float()), _cl(gy)

    dx = grid._empty_cl(b, cin, h, w, _dev())
    grid.conv3x3_dgrad_(gyd, wd, dx)
    _close(dx, x.grad)
    # mask epilogue (ReLU backward of the producing layer) and accumulation
    mask = torch.randn(b, cin, h, w, generator=g)
    base = torch.randn(b, cin, h, w, generator=g)
    dx2 = _cl(base.clone())
    grid.conv3x3_dgrad_(gyd, wd, dx2, mask=_cl(mask), accumulate=True)
    _close(dx2, base.double() + x.grad * (mask > 0))

    dw = torch.empty(cout, cin, 3, 3, device=_dev()).contiguous(memory_format=torch.channels_last)
    db = torch.empty(cout, device=_dev())
    grid.conv3x3_wgrad_(gyd, xd, dw, db)
    _close(dw, wt.grad)
    _close(db, bias.grad)
    dw0, db0 = dw.clone(), db.clone()
    grid.conv3x3_wgrad_(gyd, xd, dw, db, accumulate=True)
    assert torch.equal(dw, dw0 + dw0) and torch.equal(db, db0 + db0)          # same slabs, same order: exact doubling
    grid.conv3x3_wgrad_(gyd, xd, dw, None)                                    # bias gradient optional
    assert torch.equal(dw, dw0)


# ---- the bf16 matrix-core path with the exact 3-way operand split (csrc/conv_bx3.hip) ------------------------------------------
# (B, H, W, Cin, Cout): one tile, several chunks, batch > 1, non-square, every BN (32 / 64 / 128 / 2 x 128) and every wgrad
# grouping (Cout = 32, 64, 96, 128, 256), H = 4 (tile = plane height), and the ALTO bottom levels, whose reduction is split into
# slabs (512 channels at 32 x 32: 16 splits; 256 -> 512; 512 -> 256 at 64 x 64)
BX3_SHAPES = [(1, 32, 32, 32, 32), (1, 64, 64, 64, 128), (2, 16, 32, 32, 64), (1, 4, 64, 128, 32), (1, 32, 128, 96, 96),
              (3, 8, 32, 64, 256), (1, 128, 128, 32, 64), (1, 32, 32, 512, 512), (1, 32, 32, 256, 512), (1, 64, 64, 512, 256)]


@pytest.fixture(params=["bf16x3", "f16x2"])
def bx3_everywhere(monkeypatch, request):
    """Both split arithmetics of csrc/conv_bx3.hip: the bf16 three-way split (6 MFMAs per product) and the fp16 two-way split with
    block scales (3 MFMAs) -- same kernels, same tests, same tolerances."""
    from tomosar2height_amd import grid
    monkeypatch.setattr(grid, "CONV_PRECISION", request.param)
    monkeypatch.setattr(grid, "BX3_MIN_PIXELS", 0)
    monkeypatch.setattr(grid, "BX3_WGRAD", True)
    return grid


@pytest.mark.parametrize("shape", BX3_SHAPES)
def test_conv3x3_bx3_against_float64(shape, bx3_everywhere):
    """Forward (bias, ReLU, accumulate), data gradient (mask, accumulate) and weight / bias gradient of the split-bf16 kernels
    against float64 F.conv2d -- the SAME tolerance as the fp32 MFMA kernels (2e-5 of the max-norm): the split is exact, the
    accumulation fp32."""
    grid = bx3_everywhere
    b, h, w, cin, cout = shape
    assert grid.bx3_applicable(b, h, w, cin, cout)
    g = torch.Generator().manual_seed(11 + sum(shape))
    x = torch.randn(b, cin, h, w, generator=g).double().requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).double().requires_grad_(True)
    bias = torch.randn(cout, generator=g).double().requires_grad_(True)
    gy = torch.randn(b, cout, h, w, generator=g)
    want = F.conv2d(x, wt, bias, padding=1)
    want.backward(gy.double())
    xd, wd, gyd = _cl(x.detach().float()), _cl(wt.detach().float()), _cl(gy)
    from tomosar2height_amd import _lib
    with _lib.KernelTimeline() as tl:
        y = grid._empty_cl(b, cout, h, w, _dev())
        grid.conv3x3_fwd_(xd, wd, bias.detach().float().to(_dev()), y, relu=True)
        dx = grid._empty_cl(b, cin, h, w, _dev())
        grid.conv3x3_dgrad_(gyd, wd, dx)
        dw = torch.empty(cout, cin, 3, 3, device=_dev()).contiguous(memory_format=torch.channels_last)
        db = torch.empty(cout, device=_dev())
        grid.conv3x3_wgrad_(gyd, xd, dw, db)
    torch.cuda.synchronize()
    assert sorted(r[5].split("<")[0] for r in tl.records if r[5].startswith("bx3")) == ["bx3_rows_kernel", "bx3_rows_kernel", "bx3_wgrad_kernel"]
    _close(y, F.relu(want.detach()))
    _close(dx, x.grad)
    _close(dw, wt.grad)
    _close(db, bias.grad)
    base = torch.randn(b, cout, h, w, generator=g)
    y2 = _cl(base.clone())
    grid.conv3x3_fwd_(xd, wd, None, y2, accumulate=True)
    _close(y2, base.double() + F.conv2d(x.detach(), wt.detach(), None, padding=1))
    mask = torch.randn(b, cin, h, w, generator=g)
    base = torch.randn(b, cin, h, w, generator=g)
    dx2 = _cl(base.clone())
    grid.conv3x3_dgrad_(gyd, wd, dx2, mask=_cl(mask), accumulate=True)
    _close(dx2, base.double() + x.grad * (mask > 0))
    dw0, db0 = dw.clone(), db.clone()
    grid.conv3x3_wgrad_(gyd, xd, dw, db, accumulate=True)
    assert torch.equal(dw, dw0 + dw0) and torch.equal(db, db0 + db0)          # same slabs, same order: exact doubling
    grid.conv3x3_wgrad_(gyd, xd, dw, None)
    assert torch.equal(dw, dw0)


def test_conv3x3_bx3_error_is_fp32_grade(bx3_everywhere):
    """The claim behind routing fp32 convolutions through bf16 MFMAs: against float64 the split path's error, relative to
    sum |a b| of each output, is the fp32 fma chain's (measured 1.5-2.1e-7 for both; bound here 4e-7), for inputs with sign changes
    and three decades of dynamic range; and NaN / Inf propagate."""
    grid = bx3_everywhere
    g = torch.Generator().manual_seed(5)
    b, cin, cout, h, w = 1, 64, 128, 64, 64
    x = torch.randn(b, cin, h, w, generator=g) * torch.logspace(-1.5, 1.5, cin).view(1, -1, 1, 1)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    want = F.conv2d(x.double(), wt.double(), None, padding=1)
    mag = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=1)
    errs = {}
    split = grid.CONV_PRECISION
    for mode in (split, "fp32"):
        grid.CONV_PRECISION = mode
        y = grid._empty_cl(b, cout, h, w, _dev())
        grid.conv3x3_fwd_(_cl(x), _cl(wt), None, y)
        errs[mode] = ((y.cpu().double() - want).abs() / mag).max().item()
    grid.CONV_PRECISION = split
    print(f"[bx3] max |err| / sum|a b| vs float64: {errs}")
    assert errs[split] <= 4e-7 and errs["fp32"] <= 4e-7
    x2 = x.clone()
    x2[0, 3, 10, 10], x2[0, 5, 40, 40] = float("nan"), float("inf")
    y = grid._empty_cl(b, cout, h, w, _dev())
    grid.conv3x3_fwd_(_cl(x2), _cl(wt), None, y)
    yc = y.cpu()
    assert torch.isnan(yc[0, :, 9:12, 9:12]).all() and not torch.isfinite(yc[0, :, 39:42, 39:42]).any()
    assert torch.isfinite(yc[0, :, 20:30, 20:30]).all()


@pytest.mark.parametrize("shape", [(1, 64, 64, 64, 128), (2, 16, 32, 32, 64), (1, 32, 32, 256, 512)])
def test_conv3x3_bf16_mode(shape, bx3_everywhere):
    """BASELINE configs[2]'s arithmetic on the same kernels (T2H_BF16: operands rounded to bf16 once, ONE MFMA per product, fp32
    accumulate): forward, data and weight gradient within 1e-2 of the float64 result's max-norm (bf16: 8 significant bits; measured
    ~2e-3) -- and measurably different from the exact split, i.e. the flag reaches the kernels."""
    grid = bx3_everywhere
    b, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(3 + sum(shape))
    x = torch.randn(b, cin, h, w, generator=g).double().requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).double().requires_grad_(True)
    bias = torch.randn(cout, generator=g).double().requires_grad_(True)
    gy = torch.randn(b, cout, h, w, generator=g)
    want = F.conv2d(x, wt, bias, padding=1)
    want.backward(gy.double())
    xd, wd, gyd = _cl(x.detach().float()), _cl(wt.detach().float()), _cl(gy)
    grid.set_conv_precision("bf16")
    try:
        y = grid._empty_cl(b, cout, h, w, _dev())
        grid.conv3x3_fwd_(xd, wd, bias.detach().float().to(_dev()), y)
        dx = grid._empty_cl(b, cin, h, w, _dev())
        grid.conv3x3_dgrad_(gyd, wd, dx)
        dw = torch.empty(cout, cin, 3, 3, device=_dev()).contiguous(memory_format=torch.channels_last)
        db = torch.empty(cout, device=_dev())
        grid.conv3x3_wgrad_(gyd, xd, dw, db)
    finally:
        grid.set_conv_precision(None)
        grid.CONV_PRECISION = "bf16x3"
    for got, ref in ((y, want.detach()), (dx, x.grad), (dw, wt.grad)):
        _close(got, ref, tol=1e-2)
        err = (got.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        assert err > 2e-5, "bf16 mode gave fp32-grade results: the flag is not reaching the kernels"
    _close(db, bias.grad)                                                 # (the bias gradient is summed in fp32 from the fp32 rows)


def test_split_weight_cache_follows_the_weight(bx3_everywhere):
    """The split weights are recomputed when the weight's version counter or storage moves (optimizer steps -- FlatAdamW bumps
    the counter --, load_state_dict), in place (a captured hipGraph keeps its pointers), and not otherwise."""
    grid = bx3_everywhere
    g = torch.Generator().manual_seed(9)
    x = _cl(torch.randn(1, 32, 32, 32, generator=g))
    w = torch.nn.Parameter(_cl(torch.randn(32, 32, 3, 3, generator=g) * 0.1))
    buf = grid.split_weights.get(w, False)
    ptr = buf.data_ptr()
    snap = buf.clone()
    assert grid.split_weights.get(w, False).data_ptr() == ptr and torch.equal(buf, snap)
    with torch.no_grad():
        w.mul_(2.0)                                                         # in-place update: version bump
    buf2 = grid.split_weights.get(w, False)
    assert buf2.data_ptr() == ptr and not torch.equal(buf2, snap)
    y = grid._empty_cl(1, 32, 32, 32, _dev())
    grid.conv3x3_fwd_(x, w.detach(), None, y)                               # (a detached alias is another tensor object: own entry)
    _close(y, F.conv2d(x.cpu().double(), w.detach().cpu().double(), None, padding=1))
    w.data.zero_()                                                          # raw write without a version bump ...
    grid.split_weights.refresh()                                            # ... is what refresh() is for (hipGraph replay)
    planes = grid.split_weights.get(w, False)
    planes = planes[:-256] if grid.CONV_PRECISION == "f16x2" else planes      # (the fp16 buffers end with the tensor's scale: 2^0 here)
    assert int(planes.abs().sum().item()) == 0


def test_f16x2_block_scales(monkeypatch):
    """The fp16 two-way split keeps ONE power of two per staged block (halo tile x channel chunk; 32-pixel unit of the weight
    gradient), so its error is relative to the BLOCK's largest element, not the tensor's: (a) image regions 2^40 apart in magnitude
    (far beyond fp16's 2^30 range) are each fp32-grade, as are channel chunks 2^30 apart (accumulators rescaled exactly when a larger
    block arrives); (b) tensors scaled by 2^+-60 give the scaled result; (c) the documented bound for elements far below their own
    block's maximum: error <= 2^-21 sum|a b| + 2^-39 (block max |a|) sum|b| -- a tiny pixel next to a huge one."""
    from tomosar2height_amd import grid
    monkeypatch.setattr(grid, "CONV_PRECISION", "f16x2")
    monkeypatch.setattr(grid, "BX3_MIN_PIXELS", 0)
    g = torch.Generator().manual_seed(21)
    b, cin, cout, h, w = 1, 64, 64, 64, 64
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    base = torch.randn(b, cin, h, w, generator=g)
    gy0 = torch.randn(b, cout, h, w, generator=g)

    def run(x, gy):
        y, dx = grid._empty_cl(b, cout, h, w, _dev()), grid._empty_cl(b, cin, h, w, _dev())
        dw = torch.empty(cout, cin, 3, 3, device=_dev()).contiguous(memory_format=torch.channels_last)
        grid.conv3x3_fwd_(_cl(x), _cl(wt), None, y)
        grid.conv3x3_dgrad_(_cl(gy), _cl(wt), dx)
        grid.conv3x3_wgrad_(_cl(gy), _cl(x), dw, None)
        return y.cpu().double(), dx.cpu().double(), dw.cpu().double()

    def rel(got, want, mag):
        return ((got - want).abs() / mag).max().item()

    # (a) rows of the image 2^40 apart (row tiles are 4 rows: rows 0-31 large, 32-63 tiny), channel chunks 2^30 apart
    scale = torch.ones(1, cin, h, 1)
    scale[:, :, 32:, :] = 2.0 ** -40
    scale[:, 32:, :, :] *= 2.0 ** -30
    x = base * scale
    gy = gy0 * scale[:, :1].expand(1, cout, h, 1)
    y, dx, dw = run(x, gy)
    want_y = F.conv2d(x.double(), wt.double(), None, padding=1)
    mag_y = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=1)
    sel = torch.ones(h, dtype=torch.bool)
    sel[28:36] = False                                       # (row tiles whose halo straddles the jump: covered by (c))
    assert rel(y[:, :, sel], want_y[:, :, sel], mag_y[:, :, sel]) <= 5e-7
    want_dx = F.conv_transpose2d(gy.double(), wt.double(), None, padding=1)
    mag_dx = F.conv_transpose2d(gy.double().abs(), wt.double().abs(), None, padding=1)
    assert rel(dx[:, :, sel], want_dx[:, :, sel], mag_dx[:, :, sel]) <= 5e-7
    want_dw = torch.nn.grad.conv2d_weight(x.double(), wt.shape, gy.double(), padding=1)
    _close(dw.float(), want_dw)
    # (b) far outside fp16's exponent range
    for k in (-60, 60):
        y2, _, _ = run(base * 2.0 ** k, gy0)
        y1, _, _ = run(base, gy0)
        assert torch.equal(y2, y1 * 2.0 ** k)
    # (c) one huge pixel: its neighbours inside the same staged block lose relative, not absolute, accuracy
    x = base.clone()
    x[0, :, 20, 20] *= 2.0 ** 30
    y, _, _ = run(x, gy0)
    want_y = F.conv2d(x.double(), wt.double(), None, padding=1)
    bound = (2.0 ** -21 * F.conv2d(x.double().abs(), wt.double().abs(), None, padding=1)
             + 2.0 ** -39 * x.abs().max().item() * wt.double().abs().sum(dim=(1, 2, 3)).view(1, -1, 1, 1))
    assert ((y - want_y).abs() <= bound).all()
    far = torch.ones(h, w, dtype=torch.bool)
    far[8:32, :] = False                                     # tiles (8 rows at most) that hold the huge pixel's halo
    mag_y = F.conv2d(x.double().abs(), wt.double().abs(), None, padding=1)
    assert rel(y[:, :, far], want_y[:, :, far], mag_y[:, :, far]) <= 5e-7


def test_conv3x3_is_deterministic():
    from tomosar2height_amd import grid
    g = torch.Generator().manual_seed(3)
    x, wt, gy = _cl(torch.randn(1, 256, 32, 32, generator=g)), _cl(torch.randn(512, 256, 3, 3, generator=g)), _cl(torch.randn(1, 512, 32, 32, generator=g))
    outs = []
    for _ in range(2):
        y = grid._empty_cl(1, 512, 32, 32, _dev())
        dx = grid._empty_cl(1, 256, 32, 32, _dev())
        dw = torch.empty_like(wt)
        grid.conv3x3_fwd_(x, wt, None, y)
        grid.conv3x3_dgrad_(gy, wt, dx)
        grid.conv3x3_wgrad_(gy, x, dw, None)
        outs.append((y, dx, dw))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _loose_fp64(pairs):
    """Gradients through ReLUs against float64: a pre-activation within fp32 rounding of zero flips its ReLU and moves the
    entries it feeds by O(1) of their size (torch's own fp32 path shows the same, seed-dependent, up to 5e-3 in L2), so this
    only guards against gross errors; exactness is pinned by the op-level tests above plus bit-equality with the unfused
    composition of the same kernels."""
    for got, want in pairs:
        want = want.double()
        diff = got.detach().cpu().double() - want
        assert diff.norm().item() <= 2e-2 * want.norm().item()


@pytest.mark.parametrize("cin,mid,cout,hw", [(32, 32, 32, 64), (64, 128, 128, 32), (512, 256, 256, 16)])
def test_conv_chain_fused_relu_backward(cin, mid, cout, hw):
    """conv1 -> ReLU -> conv2 -> ReLU with the ReLU backward fused into the data-gradient epilogue: bit-identical to the
    unfused composition of the same kernels (separate relu_mask passes); forward against float64."""
    import copy
    from tomosar2height_amd import grid
    torch.manual_seed(cin + cout)
    c1, c2 = torch.nn.Conv2d(cin, mid, 3, padding=1), torch.nn.Conv2d(mid, cout, 3, padding=1)
    with torch.no_grad():
        for c in (c1, c2):
            c.bias.uniform_(-0.1, 0.1)
    x = torch.randn(2, cin, hw, hw)
    gout = torch.randn(2, cout, hw, hw)
    r1, r2 = copy.deepcopy(c1).double(), copy.deepcopy(c2).double()
    xr = x.double().requires_grad_(True)
    yr = F.relu(r2(F.relu(r1(xr))))
    yr.backward(gout.double())

    c1, c2 = (c.to(_dev()).to(memory_format=torch.channels_last) for c in (c1, c2))
    xg = _cl(x).requires_grad_(True)
    y = grid.conv3x3_chain(xg, (c1, c2))
    y.backward(_cl(gout))
    _close(y, yr.detach())
    fused = [t.clone() for t in (y.detach(), xg.grad, c1.weight.grad, c1.bias.grad, c2.weight.grad, c2.bias.grad)]
    _loose_fp64(zip(fused[1:], (xr.grad, r1.weight.grad, r1.bias.grad, r2.weight.grad, r2.bias.grad)))
    xg.grad = None
    c1.zero_grad()
    c2.zero_grad()
    y2 = grid.conv_bias_act(grid.conv_bias_act(xg, c1, relu=True), c2, relu=True)
    y2.backward(_cl(gout))
    for a, b in zip(fused, (y2.detach(), xg.grad, c1.weight.grad, c1.bias.grad, c2.weight.grad, c2.bias.grad)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("need_x", [True, False])
def test_fused_conv_decoder(need_x):
    """ConvDecoder (pixel.py:20-32) as one autograd node -- head gradient written ReLU-masked, conv data gradients
    accumulated onto it -- is bit-identical to the unfused composition of the same kernels (autograd summing the two
    consumers of every activation, separate relu_mask passes); forward against cat + convs in float64."""
    import copy
    from tomosar2height_amd import grid
    from tomosar2height_amd.decoder.pixel import ConvDecoder
    torch.manual_seed(5)
    dec = ConvDecoder(32, 1)
    with torch.no_grad():
        for c in (dec.conv1, dec.conv2, dec.conv3, dec.conv4):
            c.bias.uniform_(-0.2, 0.2)
    ref = copy.deepcopy(dec).double()
    x = torch.randn(2, 32, 32, 32)
    gout = torch.randn(2, 1, 32, 32)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(gout.double())

    dec = dec.to(_dev()).to(memory_format=torch.channels_last)
    dec.channels_last = True
    convs = (dec.conv1, dec.conv2, dec.conv3, dec.conv4)
    xg = _cl(x).requires_grad_(need_x)
    y = dec(xg)
    assert type(y.grad_fn).__name__ == "_ConvDecoderBackward"
    y.backward(gout.to(_dev()))
    _close(y, yr.detach())
    fused = [y.detach().clone()] + [c.weight.grad.clone() for c in convs] + [c.bias.grad.clone() for c in convs]
    want = [getattr(ref, n).weight.grad for n in ("conv1", "conv2", "conv3", "conv4")]
    want += [getattr(ref, n).bias.grad for n in ("conv1", "conv2", "conv3", "conv4")]
    if need_x:
        fused.append(xg.grad.clone())
        want.append(xr.grad)
    else:
        assert xg.grad is None
    _loose_fp64(zip(fused[1:], want))
    xg.grad = None
    dec.zero_grad()
    x1 = grid.conv_bias_act(xg, dec.conv1)
    x2 = grid.conv_bias_act(x1, dec.conv2)
    x3 = grid.conv_bias_act(x2, dec.conv3)
    y2 = grid.head1x1([xg, x1, x2, x3], dec.conv4)
    y2.backward(gout.to(_dev()))
    unfused = [y2.detach()] + [c.weight.grad for c in convs] + [c.bias.grad for c in convs] + ([xg.grad] if need_x else [])
    for a, b in zip(fused, unfused):
        assert torch.equal(a, b)


def test_conv_decoder_rank1_epilogue_is_bit_identical():
    """r06: at 256 x 256 (no split reduction) the decoder's backward forms the 1 x 1 head's share of d x, d x1, d x2 --
    g[pixel] * w4[channel], ReLU-masked -- in the epilogue of the data gradient that produces the rest of that gradient
    (t2h_conv3x3_bx3_dgrad_rank1) instead of having the head write it and the data gradient accumulate onto it: every gradient
    bit for bit the same (T2H_HEAD_RANK1 = 0 / 1), the head's kernel no longer writes those three tensors."""
    from tomosar2height_amd import _lib, grid
    from tomosar2height_amd.decoder.pixel import ConvDecoder
    torch.manual_seed(7)
    dec = ConvDecoder(32, 1)
    with torch.no_grad():
        for c in (dec.conv1, dec.conv2, dec.conv3, dec.conv4):
            c.bias.uniform_(-0.2, 0.2)
    dec = dec.to(_dev()).to(memory_format=torch.channels_last)
    dec.channels_last = True
    convs = (dec.conv1, dec.conv2, dec.conv3, dec.conv4)
    x = torch.randn(1, 32, 256, 256, generator=torch.Generator().manual_seed(8))
    gout = torch.randn(1, 1, 256, 256, generator=torch.Generator().manual_seed(9)).to(_dev())
    res, names = {}, {}
    old = grid.HEAD_RANK1
    try:
        for on in (False, True, False):                     # (the first pass also prepares the transposed split weights)
            grid.HEAD_RANK1 = on
            dec.zero_grad()
            xg = _cl(x).requires_grad_(True)
            y = dec(xg)
            with _lib.KernelTimeline() as tl:
                y.backward(gout)
            torch.cuda.synchronize()
            names[on] = [r[0] for r in tl.records]
            res[on] = [xg.grad.clone()] + [c.weight.grad.clone() for c in convs] + [c.bias.grad.clone() for c in convs]
    finally:
        grid.HEAD_RANK1 = old
    assert grid.dgrad_rank1_ok(torch.empty(1, 64, 256, 256, device=_dev()), dec.conv1.weight)
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
    assert len(names[True]) == len(names[False])          # same launches: the rank-1 form replaces the accumulating one


def test_conv_module_path_uses_hip_and_accumulates_directly():
    """conv_bias_act on a 3x3 conv: same numbers as torch's module, weight.grad keeps the parameter's channels_last
    layout, and with direct accumulation the gradient lands in the existing .grad buffers (the trainer's bucket)."""
    from tomosar2height_amd import grid, mlp
    g = torch.Generator().manual_seed(11)
    torch.manual_seed(11)
    conv = torch.nn.Conv2d(32, 64, 3, padding=1).to(_dev()).to(memory_format=torch.channels_last)
    x = _cl(torch.randn(1, 32, 64, 64, generator=g)).requires_grad_(True)
    gout = _cl(torch.randn(1, 64, 64, 64, generator=g))
    assert grid.conv3x3_supported(x, conv)
    y = grid.conv_bias_act(x, conv, relu=True)
    y.backward(gout)
    got = (y.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
    x.grad = None
    conv.zero_grad()
    yr = F.relu(conv(x))
    yr.backward(gout)
    _close(got[0], yr.detach().cpu(), tol=1e-5)
    _loose_fp64(zip(got[1:], (x.grad.cpu(), conv.weight.grad.cpu(), conv.bias.grad.cpu())))
    # direct accumulation into pre-existing gradient buffers (created by the HIP path: channels_last like the parameter;
    # the layout of a gradient that MIOpen allocates depends on the algorithm it picks)
    x.grad = None
    conv.zero_grad()
    grid.conv_bias_act(x, conv, relu=True).backward(gout)
    wbuf, bbuf = conv.weight.grad, conv.bias.grad
    assert wbuf.permute(0, 2, 3, 1).is_contiguous()
    w0, b0 = wbuf.clone(), bbuf.clone()
    with mlp.direct_grad_accumulation(True):
        grid.conv_bias_act(x, conv, relu=True).backward(gout)
    assert conv.weight.grad is wbuf and conv.bias.grad is bbuf
    assert torch.equal(wbuf, w0 + w0) and torch.equal(bbuf, b0 + b0)


@pytest.mark.parametrize("cin,cout,hw", [(64, 32, 64), (32, 32, 33), (256, 512, 8)])
def test_conv1x1_on_gemm_kernels(cin, cout, hw):
    from tomosar2height_amd import grid
    g = torch.Generator().manual_seed(cin * cout)
    torch.manual_seed(cin * cout)
    conv = torch.nn.Conv2d(cin, cout, 1)
    x = torch.randn(2, cin, hw, hw, generator=g)
    gout = torch.randn(2, cout, hw, hw, generator=g)
    import copy
    ref = copy.deepcopy(conv).double()
    xr = x.double().requires_grad_(True)
    ref(xr).backward(gout.double())
    conv = conv.to(_dev()).to(memory_format=torch.channels_last)
    xg = _cl(x).requires_grad_(True)
    y = grid.conv1x1(xg, conv)
    y.backward(_cl(gout))
    _close(y, ref(xr).detach())
    _close(xg.grad, xr.grad)
    _close(conv.weight.grad, ref.weight.grad)
    _close(conv.bias.grad, ref.bias.grad)


@pytest.mark.parametrize("b,h,w,cin,cout", [(1, 32, 32, 512, 256), (2, 8, 16, 64, 32), (1, 128, 128, 128, 64), (3, 1, 2, 16, 16),
                                            (1, 16, 16, 48, 80)])
def test_upconv2x2_matches_conv_transpose(b, h, w, cin, cout):
    """nn.ConvTranspose2d(k=2, s=2) forward, data / weight / bias gradient against torch in float64; direct accumulation."""
    import copy
    from tomosar2height_amd import grid, mlp
    g = torch.Generator().manual_seed(b + h + cin + cout)
    torch.manual_seed(b + h + cin + cout)
    conv = torch.nn.ConvTranspose2d(cin, cout, 2, stride=2)
    ref = copy.deepcopy(conv).double()
    x = torch.randn(b, cin, h, w, generator=g)
    gout = torch.randn(b, cout, 2 * h, 2 * w, generator=g)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(gout.double())
    conv = conv.to(_dev()).to(memory_format=torch.channels_last)
    xg = _cl(x).requires_grad_(True)
    assert grid.upconv2x2_supported(xg, conv)
    y = grid.upconv2x2(xg, conv)
    assert type(y.grad_fn).__name__ == "_UpConv2x2Backward"
    y.backward(_cl(gout))
    _close(y, yr.detach())
    _close(xg.grad, xr.grad)
    _close(conv.weight.grad, ref.weight.grad)
    _close(conv.bias.grad, ref.bias.grad)
    wbuf, bbuf = conv.weight.grad, conv.bias.grad
    w0, b0 = wbuf.clone(), bbuf.clone()
    with mlp.direct_grad_accumulation(True):
        grid.upconv2x2(xg, conv).backward(_cl(gout))
    assert conv.weight.grad is wbuf and conv.bias.grad is bbuf
    assert torch.equal(wbuf, w0 + w0) and torch.equal(bbuf, b0 + b0)


@pytest.mark.parametrize("b,h,w,cin,cout", [(1, 32, 32, 512, 256), (1, 64, 64, 256, 128), (1, 128, 128, 128, 64), (2, 16, 32, 64, 64),
                                            (4, 32, 32, 192, 64), (1, 8, 16, 1024, 512)])
@pytest.mark.parametrize("mode", ["f16x2", "bf16x3"])
def test_upconv2x2_on_split_bf16_kernels(b, h, w, cin, cout, mode, monkeypatch):
    """The transposed convolutions on csrc/conv_bx3.hip (scattering epilogue / gathering loader on the 1-tap form): forward with bias
    and residual addend, data gradient (split-K slabs at the small planes), against float64, at the error of the fp32-MFMA kernels;
    the kernels actually ran; the fp32 path (T2H_UPCONV_BX3=0 sibling) gives the same values to fp32 rounding."""
    import copy
    from tomosar2height_amd import _lib, grid
    monkeypatch.setattr(grid, "BX3_MIN_PIXELS", 128)
    monkeypatch.setattr(grid, "CONV_PRECISION", mode)
    npl = 2 if mode == "f16x2" else 3
    g = torch.Generator().manual_seed(b * 7 + h + cin + cout)
    torch.manual_seed(b + h + cin + cout)
    conv = torch.nn.ConvTranspose2d(cin, cout, 2, stride=2)
    ref = copy.deepcopy(conv).double()
    x = torch.randn(b, cin, h, w, generator=g)
    add = torch.randn(b, cout, 2 * h, 2 * w, generator=g)
    gout = torch.randn(b, cout, 2 * h, 2 * w, generator=g)
    xr = x.double().requires_grad_(True)
    yr = ref(xr) + add.double()
    yr.backward(gout.double())
    conv = conv.to(_dev()).to(memory_format=torch.channels_last)
    outs = {}
    for on in (True, False):
        monkeypatch.setattr(grid, "UPCONV_BX3", on)
        conv.zero_grad(set_to_none=True)
        xg = _cl(x).requires_grad_(True)
        ag = _cl(add).requires_grad_(True)
        assert grid._up_bx3(xg, conv.weight, grid._w_cl(conv.weight)) == on
        with _lib.KernelTimeline() as tl:
            y = grid.upconv2x2(xg, conv, ag)
            y.backward(_cl(gout))
        torch.cuda.synchronize()
        names = [r[5] for r in tl.records if r[5].startswith("bx3_")]
        if on:
            assert names[0].endswith(f",64,{npl},1,1,false>") and names[1].endswith(f",64,{npl},1,2,false>"), names
            assert names[2:] == ([f"bx3_wgrad_kernel<{4 if cin % 128 == 0 else 2},{npl},true,0>"] if w >= 32 else []), names
        else:
            assert not names, names
        outs[on] = (y.detach().clone(), xg.grad.clone(), ag.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
    for on in (True, False):
        y, dx, da, dw, db = outs[on]
        _close(y, yr.detach())
        _close(dx, xr.grad)
        assert torch.equal(da.cpu(), gout)
        _close(dw, ref.weight.grad)
        _close(db, ref.bias.grad)
    # direct accumulation into .grad: same slabs, same order -> exact doubling
    from tomosar2height_amd import mlp
    monkeypatch.setattr(grid, "UPCONV_BX3", True)
    wbuf, bbuf = conv.weight.grad, conv.bias.grad
    conv.weight.grad.copy_(outs[True][3]); conv.bias.grad.copy_(outs[True][4])
    with mlp.direct_grad_accumulation(True):
        grid.upconv2x2(_cl(x).requires_grad_(True), conv, None).backward(_cl(gout))
    assert conv.weight.grad is wbuf and torch.equal(wbuf, outs[True][3] * 2) and torch.equal(bbuf, outs[True][4] * 2)
    # the split product is exact to fp32 rounding of the accumulation: both paths sit at the same distance from float64
    for i in (0, 1):
        e_on = (outs[True][i].double().cpu() - (yr.detach(), xr.grad)[i]).abs().max().item()
        e_off = (outs[False][i].double().cpu() - (yr.detach(), xr.grad)[i]).abs().max().item()
        assert e_on <= 2.0 * e_off + 1e-6, (i, e_on, e_off)


def test_upconv2x2_bx3_weights_follow_the_optimizer():
    """The split planes of a transposed-convolution weight are cached per weight version: an in-place update re-splits them."""
    from tomosar2height_amd import grid
    torch.manual_seed(3)
    conv = torch.nn.ConvTranspose2d(64, 64, 2, stride=2).to(_dev()).to(memory_format=torch.channels_last)
    x = _cl(torch.randn(1, 64, 32, 32))
    y0 = grid.upconv2x2(x, conv).detach().clone()
    with torch.no_grad():
        conv.weight.mul_(2.0)
        conv.bias.zero_()
    y1 = grid.upconv2x2(x, conv).detach()
    ref = torch.nn.functional.conv_transpose2d(x.double().cpu(), conv.weight.detach().double().cpu(), None, stride=2)
    _close(y1, ref)
    assert not torch.allclose(y0, y1)


def test_conv3x3_argument_errors():
    from tomosar2height_amd import _lib, grid
    x = grid._empty_cl(1, 32, 12, 16, _dev())
    w = torch.empty(32, 32, 3, 3, device=_dev()).contiguous(memory_format=torch.channels_last)
    y = grid._empty_cl(1, 32, 12, 16, _dev())
    with pytest.raises(RuntimeError, match="powers of two"):
        grid.conv3x3_fwd_(x, w, None, y)
    x8 = grid._empty_cl(1, 8, 16, 16, _dev())
    w8 = torch.empty(32, 8, 3, 3, device=_dev()).contiguous(memory_format=torch.channels_last)
    with pytest.raises(RuntimeError, match="multiple of 16"):
        grid.conv3x3_fwd_(x8, w8, None, grid._empty_cl(1, 32, 16, 16, _dev()))
    # the module-level entry falls back to MIOpen for unsupported geometry instead of raising
    conv = torch.nn.Conv2d(8, 32, 3, padding=1).to(_dev())
    assert not grid.conv3x3_supported(x8, conv)
    assert grid.conv_bias_act(x8, conv).shape == (1, 32, 16, 16)
    assert _lib.load().t2h_conv3x3_fwd_workspace_bytes(1, 32, 32, 512, 512) > 0
    assert _lib.load().t2h_conv3x3_fwd_workspace_bytes(1, 512, 512, 32, 64) == 0


@pytest.mark.parametrize("cin,cout,hw", [(64, 128, 512), (128, 64, 512), (32, 64, 512), (512, 512, 32), (512, 256, 64)])
def test_conv3x3_at_bench_sizes(cin, cout, hw):
    """The layer shapes of the Berlin step at full size (decoder at 512 x 512, ALTO bottom levels): forward, data and
    weight gradient against torch's convolution on the same device (fp32, another summation order: 2e-5 of the max-norm;
    the weight gradient sums 2^18 pixels: 1e-4), plus linearity of the forward in its input."""
    from tomosar2height_amd import grid
    g = torch.Generator(device=_dev()).manual_seed(cin + cout + hw)

    def rnd(*shape):
        return torch.randn(*shape, device=_dev(), generator=g)

    x = rnd(1, cin, hw, hw).contiguous(memory_format=torch.channels_last)
    gy = rnd(1, cout, hw, hw).contiguous(memory_format=torch.channels_last)
    w = (rnd(cout, cin, 3, 3) / (9 * cin) ** 0.5).contiguous(memory_format=torch.channels_last)
    bias = rnd(cout)
    y = grid._empty_cl(1, cout, hw, hw, _dev())
    grid.conv3x3_fwd_(x, w, bias, y)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=1)
    yr.backward(gy)

    def close(a, b, tol):
        assert (a - b).abs().max().item() <= tol * b.abs().max().item()

    close(y, yr.detach(), 2e-5)
    dx = grid._empty_cl(1, cin, hw, hw, _dev())
    grid.conv3x3_dgrad_(gy, w, dx)
    close(dx, xr.grad, 2e-5)
    dw, db = torch.empty_like(w), torch.empty(cout, device=_dev())
    grid.conv3x3_wgrad_(gy, x, dw, db)
    close(dw, wr.grad, 1e-4)
    close(db, br.grad, 1e-4)
    # linearity: conv(2 x + x') - bias = 2 (conv(x) - bias) + (conv(x') - bias)
    x2 = rnd(1, cin, hw, hw).contiguous(memory_format=torch.channels_last)
    y2, y3 = torch.empty_like(y), torch.empty_like(y)
    grid.conv3x3_fwd_(x2, w, None, y2)
    grid.conv3x3_fwd_((2 * x + x2).contiguous(memory_format=torch.channels_last), w, None, y3)
    close(y3, 2 * (y - bias.view(1, -1, 1, 1)) + y2, 2e-5)


@pytest.mark.parametrize("kind,cin,cout,hw", [("conv1x1", 64, 128, 32), ("conv1x1", 32, 64, 16), ("conv1x1", 64, 32, 16),
                                              ("upconv", 128, 64, 16), ("upconv", 512, 256, 8)])
def test_residual_addend_rides_the_epilogue(kind, cin, cout, hw):
    """``addend + conv(x)`` (the residual connections of alto.py:110,114,236) fused into the 1x1-conv / transposed-conv
    epilogue: forward and all gradients (including the pass-through gradient of the addend) against torch in float64; the
    addend tensor itself is left untouched."""
    import copy
    from tomosar2height_amd import grid
    torch.manual_seed(cin + cout + hw)
    conv = torch.nn.Conv2d(cin, cout, 1) if kind == "conv1x1" else torch.nn.ConvTranspose2d(cin, cout, 2, stride=2)
    ref = copy.deepcopy(conv).double()
    out_hw = hw if kind == "conv1x1" else 2 * hw
    x, res = torch.randn(2, cin, hw, hw), torch.randn(2, cout, out_hw, out_hw)
    gout = torch.randn(2, cout, out_hw, out_hw)
    xr, rr = x.double().requires_grad_(True), res.double().requires_grad_(True)
    yr = rr + ref(xr)
    yr.backward(gout.double())
    conv = conv.to(_dev()).to(memory_format=torch.channels_last)
    xg, rg = _cl(x).requires_grad_(True), _cl(res).requires_grad_(True)
    keep = rg.detach().clone()
    y = grid.conv1x1(xg, conv, rg) if kind == "conv1x1" else grid.upconv2x2(xg, conv, rg)
    assert "Backward" in type(y.grad_fn).__name__ and "Add" not in type(y.grad_fn).__name__
    y.backward(_cl(gout))
    assert torch.equal(rg.detach(), keep)
    _close(y, yr.detach())
    _close(xg.grad, xr.grad)
    _close(rg.grad, rr.grad)
    _close(conv.weight.grad, ref.weight.grad)
    _close(conv.bias.grad, ref.bias.grad)


# ------------------------------------------------------------------------------------------------ few input channels
@pytest.mark.parametrize("b,h,w,cin,cout,relu", [(1, 64, 64, 3, 32, True), (2, 37, 50, 3, 32, False), (1, 512, 512, 3, 32, True),
                                                 (1, 16, 16, 1, 8, True), (1, 40, 24, 8, 64, False)])
def test_conv3x3_small_cin_vs_torch(b, h, w, cin, cout, relu):
    """csrc/conv_small.hip (the image U-Net's Conv2d(3, 32, 3, padding=1), encoder/unet.py:112-187) against float64
    F.conv2d: forward, input gradient, weight and bias gradient; and bit-reproducible."""
    import torch.nn.functional as F
    from tomosar2height_amd import grid
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(b * h + cin)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).to(dev)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    x = torch.randn(b, cin, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn(b, cout, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    assert grid.conv3x3_small_supported(x, conv)
    y = grid.conv_bias_act(x, conv, relu=relu)
    y.backward(gy)
    xd = x.detach().double().requires_grad_(True)
    wd, bd = conv.weight.detach().double().requires_grad_(True), conv.bias.detach().double().requires_grad_(True)
    yd = F.conv2d(xd, wd, bd, padding=1)
    yd = torch.relu(yd) if relu else yd
    yd.backward(gy.double())
    for got, want, what in ((y, yd, "y"), (x.grad, xd.grad, "dx"), (conv.weight.grad, wd.grad, "dw"), (conv.bias.grad, bd.grad, "db")):
        scale = want.abs().max().item() + 1e-30
        assert (got.double() - want).abs().max().item() <= 2e-5 * scale, what
    first = (conv.weight.grad.clone(), conv.bias.grad.clone())
    conv.weight.grad = None; conv.bias.grad = None; x.grad = None
    grid.conv_bias_act(x, conv, relu=relu).backward(gy)
    assert torch.equal(first[0], conv.weight.grad) and torch.equal(first[1], conv.bias.grad)


def test_no_library_fallback_on_the_reference_configurations():
    """Berlin cloud-only and Munich cloud+image+footprint training steps take NO MIOpen / rocBLAS / ATen fallback (the
    default policy would raise on one): every convolution incl. the image U-Net's 3-channel first layer, every Linear
    and every pooling runs on libt2h_hip.so."""
    import tomosar2height_amd as t2h
    from tomosar2height_amd import TomoSAR2Height
    from tomosar2height_amd.config import berlin_config, munich_config
    from tomosar2height_amd.synthetic import berlin_tile
    from tomosar2height_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    t2h.fallback_counts(reset=True)
    for cfg, image, foot in ((berlin_config(), False, False), (munich_config(use_image=True), True, True)):
        torch.manual_seed(1)
        model = TomoSAR2Height(cfg).to(dev)
        tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=dev, optimize_every=2, use_cloud=True,
                     use_image=image, use_footprint=foot)
        tile = berlin_tile(3, n_points=5000, with_image=image)
        for _ in range(2):
            tr.train_step({k: v.to(dev) for k, v in tile.items() if k != "is_valid"})
        torch.cuda.synchronize()
    assert t2h.fallback_counts() == {}


def test_unsupported_geometry_raises_unless_allowed():
    import tomosar2height_amd as t2h
    from tomosar2height_amd import _lib, grid
    dev = torch.device("cuda:0")
    conv = torch.nn.Conv2d(16, 16, 5, padding=2).to(dev)
    x = torch.randn(1, 16, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
    with pytest.raises(_lib.T2HLibraryError, match="library fallbacks are off"):
        grid.conv_bias_act(x, conv)
    t2h.fallback_counts(reset=True)
    with t2h.allow_library_fallback():
        y = grid.conv_bias_act(x, conv)
    assert y.shape == (1, 16, 32, 32) and sum(t2h.fallback_counts(reset=True).values()) == 1


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3"])
def test_batched_weight_preparation_equals_the_single_calls(mode, monkeypatch):
    """``SplitWeightCache.refresh`` (t2h_split_weights_batch: all buffers in a few launches, alternating max slots for the fp16
    buffers) leaves exactly the bytes the per-buffer preparation leaves -- 3x3 weights in both orientations, a matrix in both
    layouts, a transposed-convolution weight -- run after run, and follows in-place weight updates."""
    from tomosar2height_amd import grid
    monkeypatch.setattr(grid, "CONV_PRECISION", mode)
    cache = grid.SplitWeightCache()
    g = torch.Generator().manual_seed(4)
    w3 = _cl(torch.randn(64, 32, 3, 3, generator=g))
    wm = (torch.randn(128, 192, generator=g) * 3).to(_dev())
    wu = torch.randn(64, 64, 2, 2, generator=g).to(_dev()).contiguous(memory_format=torch.channels_last)
    def single():
        fresh = grid.SplitWeightCache()
        return [fresh.get(w3, False).clone(), fresh.get(w3, True).clone(), fresh.get_gemm(wm, False).clone(),
                fresh.get_gemm(wm, True).clone(), fresh.get_up(wu, True).clone(), fresh.get_up(wu, False).clone()]
    bufs = [cache.get(w3, False), cache.get(w3, True), cache.get_gemm(wm, False), cache.get_gemm(wm, True), cache.get_up(wu, True),
            cache.get_up(wu, False)]
    planes = (lambda t: t[:-256]) if mode == "f16x2" else (lambda t: t)
    for rnd in range(4):
        with torch.no_grad():
            w3.mul_(1.7); wm.add_(0.25); wu.mul_(-0.6)                      # scale changes: the fp16 buffers' exponents move
        cache.refresh(stale_only=bool(rnd & 1))
        want = single()
        for got, ref in zip(bufs, want):
            assert torch.equal(planes(got), planes(ref))
            if mode == "f16x2":                                              # 2^-e_w and e_w of the trailer
                assert torch.equal(got[-256:].view(torch.int32)[1:3], ref[-256:].view(torch.int32)[1:3])
    before = [b.clone() for b in bufs]
    cache.refresh(stale_only=True)                                           # nothing moved: nothing launched, nothing changed
    assert all(torch.equal(a, b) for a, b in zip(before, bufs))


def test_sliced_gradients_are_read_in_place(monkeypatch):
    """``torch.cat``'s backward hands each input a channel SLICE of the concatenation's gradient (alto.py:227).  The transposed
    convolution (data and weight gradient on the split kernels), the 1x1 convolution and the max-pool's skip gradient read such a
    slice in place through a pixel stride -- no dense copy -- and give the gradients of the float64 reference."""
    import copy
    from tomosar2height_amd import grid
    monkeypatch.setattr(grid, "BX3_MIN_PIXELS", 128)
    torch.manual_seed(12)
    g = torch.Generator().manual_seed(12)
    up = torch.nn.ConvTranspose2d(64, 64, 2, stride=2)
    c1 = torch.nn.Conv2d(64, 32, 1)
    x = torch.randn(1, 64, 16, 32, generator=g)
    skip = torch.randn(1, 32, 32, 64, generator=g)
    mix = torch.randn(1, 64 + 32 + 32, 32, 64, generator=g)
    upr, c1r = copy.deepcopy(up).double(), copy.deepcopy(c1).double()
    xr, sr = x.double().requires_grad_(True), skip.double().requires_grad_(True)
    yr = upr(xr)
    pr = F.max_pool2d(sr, 2, 2)
    (torch.cat((yr, c1r(yr), sr), 1) * mix.double()).sum().backward(retain_graph=True)
    (pr * pr).sum().backward()
    up, c1 = up.to(_dev()).to(memory_format=torch.channels_last), c1.to(_dev()).to(memory_format=torch.channels_last)
    xg, sg = _cl(x).requires_grad_(True), _cl(skip).requires_grad_(True)
    copies = []
    real_as_cl = grid._as_cl
    monkeypatch.setattr(grid, "_as_cl", lambda t: (copies.append(tuple(t.shape)) if not grid.is_cl(t) else None, real_as_cl(t))[1])
    y = grid.upconv2x2(xg, up)
    pooled, thru = grid.maxpool2x2_thru(sg)
    loss = (torch.cat((y, grid.conv1x1(y, c1), thru), 1) * _cl(mix)).sum() + (pooled * pooled).sum()
    loss.backward()
    assert not copies, f"dense copies of sliced gradients: {copies}"
    _close(xg.grad, xr.grad)
    _close(sg.grad, sr.grad)
    _close(up.weight.grad, upr.weight.grad)
    _close(up.bias.grad, upr.bias.grad)
    _close(c1.weight.grad, c1r.weight.grad)
    _close(c1.bias.grad, c1r.bias.grad)
