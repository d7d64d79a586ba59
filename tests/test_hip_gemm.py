"""GPU parity of the fp32 MFMA linear-layer kernels (t2h_linear_fwd / dgrad / wgrad) against float64 on the CPU.
Tolerance: fp32 fma-chain rounding, 2e-5 relative to the output scale (K up to 1024)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _rel(got, want):
    want = want.double()
    return ((got.double().cpu() - want).abs().max() / (want.abs().max() + 1e-30)).item()


SHAPES = [(300, 8, 16), (257, 64, 32), (1000, 32, 32), (4096, 64, 128), (777, 128, 64), (3000, 256, 512),
          (2048, 512, 1024), (1500, 1024, 512), (129, 3, 64), (5000, 36, 20),
          # edges of the LDS-DMA staged forward kernel (N > 64, K % 16 == 0): one row, ragged row / column tiles, one slab
          (1, 16, 128), (129, 48, 96), (500, 16, 200),
          # few rows, long reduction: the four waves of a workgroup split the reduction (gemm_kwaves_kernel); ragged last
          # slab, ragged tiles, fewer slabs than waves
          (1000, 300, 100), (1024, 1664, 512), (33, 260, 68), (640, 2432, 128)]


@pytest.mark.parametrize("m,k,n", SHAPES)
@pytest.mark.parametrize("relu_in,relu_out,accum", [(False, False, False), (True, True, True)])
def test_linear_fwd(m, k, n, relu_in, relu_out, accum):
    from tomosar2height_amd import mlp
    g = torch.Generator().manual_seed(m + k + n)
    ldx, ldy = k + (4 if k % 4 == 0 else 0), n + 8
    xbuf = torch.randn(m, ldx, generator=g)
    w = torch.randn(n, k, generator=g) / k ** 0.5
    b = torch.randn(n, generator=g)
    ybuf = torch.randn(m, ldy, generator=g)
    x, y0 = xbuf[:, :k], ybuf[:, :n]
    want = (x.double().clamp(min=0) if relu_in else x.double()) @ w.double().t() + b.double()
    if relu_out:
        want = want.clamp(min=0)
    if accum:
        want = want + y0.double()
    xd, yd = xbuf.to(_dev()), ybuf.to(_dev())
    mlp.linear_fwd_(xd[:, :k], w.to(_dev()), b.to(_dev()), yd[:, :n], relu_in=relu_in, relu_out=relu_out, accumulate=accum)
    assert _rel(yd[:, :n], want) < 2e-5
    assert torch.equal(yd[:, n:].cpu(), ybuf[:, n:])            # columns outside the slice untouched


@pytest.mark.parametrize("m,k,n", [s for s in SHAPES if s[1] % 4 == 0 and s[2] % 4 == 0])
@pytest.mark.parametrize("masked,accum", [(False, False), (True, True)])
def test_linear_dgrad(m, k, n, masked, accum):
    from tomosar2height_amd import mlp
    g = torch.Generator().manual_seed(m + k + n + 1)
    dy = torch.randn(m, n, generator=g)
    w = torch.randn(n, k, generator=g) / n ** 0.5
    mask = torch.randn(m, k, generator=g)
    dx0 = torch.randn(m, k, generator=g)
    want = dy.double() @ w.double()
    if masked:
        want = want * (mask > 0)
    if accum:
        want = want + dx0.double()
    dxd = dx0.to(_dev())
    mlp.linear_dgrad_(dy.to(_dev()), w.to(_dev()), dxd, mask=mask.to(_dev()) if masked else None, accumulate=accum)
    assert _rel(dxd, want) < 2e-5


@pytest.mark.parametrize("m,k,n", SHAPES + [(131072, 32, 64), (20000, 512, 256)])
@pytest.mark.parametrize("relu_in,accum", [(False, False), (True, True)])
def test_linear_wgrad(m, k, n, relu_in, accum):
    from tomosar2height_amd import mlp
    if n % 4:
        pytest.skip("N_out is a multiple of 4 in every layer of the network")
    g = torch.Generator().manual_seed(m + k + n + 2)
    dy = torch.randn(m, n, generator=g)
    x = torch.randn(m, k, generator=g)
    dw0, db0 = torch.randn(n, k, generator=g), torch.randn(n, generator=g)
    xa = x.double().clamp(min=0) if relu_in else x.double()
    want_w, want_b = dy.double().t() @ xa, dy.double().sum(0)
    if accum:
        want_w, want_b = want_w + dw0.double(), want_b + db0.double()
    dwd, dbd = dw0.to(_dev()), db0.to(_dev())
    mlp.linear_wgrad_(dy.to(_dev()), x.to(_dev()), dwd, dbd, relu_in=relu_in, accumulate=accum)
    assert _rel(dwd, want_w) < 3e-5
    assert _rel(dbd, want_b) < 3e-5
    # deterministic: a second run gives bit-identical sums
    dw2, db2 = dw0.to(_dev()), db0.to(_dev())
    mlp.linear_wgrad_(dy.to(_dev()), x.to(_dev()), dw2, db2, relu_in=relu_in, accumulate=accum)
    assert torch.equal(dw2, dwd) and torch.equal(db2, dbd)


@pytest.mark.parametrize("m,k,n", [(1024, 512, 1664), (4096, 256, 512), (1000, 300, 100), (16384, 128, 2432), (300, 68, 36)])
@pytest.mark.parametrize("relu_in,accum", [(False, False), (True, True)])
def test_linear_wgrad_without_bias_gradient(m, k, n, relu_in, accum):
    """The grid-side products of the deferred point update ask for dW only (few rows: one launch whose waves split the rows)."""
    from tomosar2height_amd import mlp
    g = torch.Generator().manual_seed(m + k + n + 3)
    dy, x, dw0 = torch.randn(m, n, generator=g), torch.randn(m, k, generator=g), torch.randn(n, k, generator=g)
    want = dy.double().t() @ (x.double().clamp(min=0) if relu_in else x.double())
    if accum:
        want = want + dw0.double()
    dwd = dw0.to(_dev())
    mlp.linear_wgrad_(dy.to(_dev()), x.to(_dev()), dwd, None, relu_in=relu_in, accumulate=accum)
    assert _rel(dwd, want) < 3e-5
    dw2 = dw0.to(_dev())
    mlp.linear_wgrad_(dy.to(_dev()), x.to(_dev()), dw2, None, relu_in=relu_in, accumulate=accum)
    assert torch.equal(dw2, dwd)


def test_a_equals_identity_asymmetric_b():
    """Layout check the MFMA guide asks for: A = I with an asymmetric B catches a transposed C write."""
    from tomosar2height_amd import mlp
    k = 64
    x = torch.eye(k)
    w = torch.arange(k * 96, dtype=torch.float32).reshape(96, k)          # asymmetric integers
    y = torch.empty(k, 96, device=_dev())
    mlp.linear_fwd_(x.to(_dev()), w.to(_dev()), None, y)
    assert torch.equal(y.cpu(), w.t().contiguous())


@pytest.mark.parametrize("m,k,n", [(1000, 32, 32), (4096, 64, 128), (3000, 256, 512), (2048, 512, 1024), (777, 128, 64)])
def test_bf16_mode_is_exact_bf16_rounding(m, k, n):
    """T2H_BF16: operands rounded to bf16 (RNE), products exact, fp32 accumulation -- so the result must equal a
    float64 matmul of the bf16-rounded operands to fp32 accumulation error (2e-5), for all three GEMMs."""
    from tomosar2height_amd import mlp
    g = torch.Generator().manual_seed(m + k + n + 7)
    x, w, b = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    dy, mask = torch.randn(m, n, generator=g), torch.randn(m, k, generator=g)
    r = lambda t: t.bfloat16().double()
    mlp.set_precision("bf16")
    try:
        y = torch.empty(m, n, device=_dev())
        mlp.linear_fwd_(x.to(_dev()), w.to(_dev()), b.to(_dev()), y, relu_in=True, relu_out=True)
        want = (r(x.clamp(min=0)) @ r(w).t() + b.double()).clamp(min=0)
        assert _rel(y, want) < 2e-5
        fp32_err = _rel(y, (x.double().clamp(min=0) @ w.double().t() + b.double()).clamp(min=0))
        assert 1e-4 < fp32_err < 3e-2          # really is bf16 arithmetic, and of the expected size
        dx = torch.empty(m, k, device=_dev())
        mlp.linear_dgrad_(dy.to(_dev()), w.to(_dev()), dx, mask=mask.to(_dev()))
        assert _rel(dx, (r(dy) @ r(w)) * (mask > 0)) < 2e-5
        dw, db = torch.empty(n, k, device=_dev()), torch.empty(n, device=_dev())
        mlp.linear_wgrad_(dy.to(_dev()), x.to(_dev()), dw, db, relu_in=True)
        assert _rel(dw, r(dy).t() @ r(x.clamp(min=0))) < 3e-5
        assert _rel(db, dy.double().sum(0)) < 3e-5            # the bias gradient stays an fp32 sum
    finally:
        mlp.set_precision("fp32")


@pytest.mark.parametrize("m,k,n", [(1000, 32, 32), (4096, 64, 128), (3000, 256, 512), (2048, 512, 1024), (1500, 1024, 512)])
def test_bf16x3_mode_is_fp32_grade(m, k, n):
    """T2H_BF16X3 (opt-in): products from an exact 3-way bf16 split.  Against float64 its error must be of the same
    order as the native fp32 MFMA path's (within 4x), i.e. ~1e-7 .. 1e-6, nowhere near bf16's 1e-3."""
    from tomosar2height_amd import mlp
    g = torch.Generator().manual_seed(m + k + n + 9)
    x, w, b = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    dy, mask = torch.randn(m, n, generator=g), torch.randn(m, k, generator=g)
    want_y = x.double() @ w.double().t() + b.double()
    want_dx = (dy.double() @ w.double()) * (mask > 0)
    want_dw = dy.double().t() @ x.double()
    errs = {}
    for mode in ("fp32", "bf16x3"):
        mlp.set_precision(mode)
        try:
            y, dx = torch.empty(m, n, device=_dev()), torch.empty(m, k, device=_dev())
            dw, db = torch.empty(n, k, device=_dev()), torch.empty(n, device=_dev())
            mlp.linear_fwd_(x.to(_dev()), w.to(_dev()), b.to(_dev()), y)
            mlp.linear_dgrad_(dy.to(_dev()), w.to(_dev()), dx, mask=mask.to(_dev()))
            mlp.linear_wgrad_(dy.to(_dev()), x.to(_dev()), dw, db)
            errs[mode] = (_rel(y, want_y), _rel(dx, want_dx), _rel(dw, want_dw))
        finally:
            mlp.set_precision("fp32")
    for e32, e3 in zip(errs["fp32"], errs["bf16x3"]):
        assert e3 < 2e-5 and e3 <= 4 * e32 + 2e-7, (errs)


@pytest.mark.gpu
@pytest.mark.parametrize("k,n", [(512, 1024), (1024, 512), (256, 512), (128, 256), (64, 128), (32, 32)])
def test_full_size_linear_kernels_satisfy_the_bilinear_identities(k, n):
    """BASELINE.json config 2 rows (M = 131072): with y = x W^T, <y, g> == <x, dgrad(g)> == <W, wgrad(g, x)> in float64 --
    forward, data-gradient and weight-gradient kernels of one layer agree with each other at the full problem size (the
    small-size tests pin each of them against torch)."""
    from tomosar2height_amd import mlp
    m = 131072
    gen = torch.Generator().manual_seed(k + n)
    x = torch.randn(m, k, generator=gen).cuda()
    w = (torch.randn(n, k, generator=gen) / k ** 0.5).cuda()
    g = torch.randn(m, n, generator=gen).cuda()
    y = torch.empty(m, n, device="cuda")
    mlp.linear_fwd_(x, w, None, y)
    dx = mlp.linear_dgrad_(g, w, torch.empty_like(x))
    dw = torch.empty_like(w)
    mlp.linear_wgrad_(g, x, dw, None)

    def dot(a, b):
        return float((a.double() * b.double()).sum())

    a, b, c = dot(y, g), dot(x, dx), dot(w, dw)
    scale = max(abs(a), 1.0)
    assert abs(a - b) <= 2e-5 * scale + 1e-2 and abs(a - c) <= 2e-5 * scale + 1e-2, (a, b, c)


# ---- the 1-tap form of the split-bf16 matrix-core kernels as a GEMM on rows (csrc/conv_bx3.hip, t2h_gemm_bx3) -------------------
@pytest.mark.parametrize("m,k,n", [(65536, 320, 64), (16384, 832, 128), (1024, 1856, 512), (4096, 64, 832), (128, 64, 32),
                                   (16384, 128, 2624), (8192, 64, 2752)])
def test_gemm_bx3_rows_vs_float64(m, k, n, monkeypatch):
    """The grid-side products of the deferred point update (deferred.py): y = x W^T and y = x W (k-major weight) on column SLICES
    of wider matrices, with mask / accumulate, and a few-row case whose reduction is split into slabs -- against float64 at the
    tolerance of the fp32 MFMA GEMMs (2e-5 of the max-norm).  k = 64 with >= 8 column tiles runs the persistent form (r05: rows staged
    and split once per workgroup, reused for all its column tiles): (4096, 64, 832) and the stacked per-pixel product's (., 64, 2752)."""
    from tomosar2height_amd import mlp
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(m + k + n)
    wide = torch.randn(m, k + 64, generator=g).to(dev)
    x = wide[:, 32:32 + k]                                                   # row stride k + 64, 128-byte offset
    w_nk = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
    w_kn = w_nk.t().contiguous()
    want = x.double().cpu() @ w_nk.double().cpu().t()
    scale = want.abs().max().item()
    assert mlp._bx3_gemm_ok(m, k, n, x, force=True)
    monkeypatch.setattr(mlp, "_bx3_gemm_ok", lambda *a, **kw: True)      # every shape through the split form here

    out_wide = torch.zeros(m, n + 32, device=dev)
    y = out_wide[:, 16:16 + n]
    from tomosar2height_amd import _lib
    with _lib.KernelTimeline() as tl:
        mlp.linear_fwd_(x, w_nk, None, y, bx3=True)
    torch.cuda.synchronize()
    assert any(r[5].startswith("bx3_rows_kernel") for r in tl.records), [r[5] for r in tl.records]
    assert (y.double().cpu() - want).abs().max().item() <= 2e-5 * scale
    assert float(out_wide[:, :16].abs().max()) == 0.0 and float(out_wide[:, 16 + n:].abs().max()) == 0.0
    # k-major weight ("dx = dy w"), ReLU mask of the consumer, accumulate
    mask = torch.randn(m, n, generator=g).to(dev)
    base = torch.randn(m, n, generator=g).to(dev)
    dx = base.clone()
    mlp.linear_dgrad_(x, w_kn, dx, mask=mask, accumulate=True, bx3=True)
    want2 = base.double().cpu() + want * (mask.cpu() > 0)
    assert (dx.double().cpu() - want2).abs().max().item() <= 2e-5 * max(scale, want2.abs().max().item())
    # determinism (slabs are summed in a fixed order)
    y2 = torch.empty(m, n, device=dev)
    mlp.linear_fwd_(x, w_nk, None, y2, bx3=True)
    assert torch.equal(y2, y.contiguous())


@pytest.mark.parametrize("m,k,n", [(65536, 64, 2752), (16384, 128, 2624), (4096, 256, 2368), (262144, 64, 2752), (4096, 64, 1024),
                                   (32, 64, 64)])
def test_gemm_bx3_wgrad_vs_float64(m, k, n, monkeypatch):
    """r06: the weight gradients of the wide grid-side products (dW = dY^T X over the pixel rows of a level: N = 2368 .. 2752
    columns, K = 64 .. 256, alto.py:123-130 re-associated) on the split kernels, t2h_gemm_bx3_wgrad = bx3_wgrad_kernel<2, 2, false,
    K / 32> -- against float64 at the tolerance of the fp32 MFMA GEMMs (2e-5 of the max-norm of dW, 2e-5 for db), on strided row
    slices, with accumulate, with row blocks whose magnitudes are 2^20 apart (one power-of-two scale per 32-row unit and operand),
    at one and at four tiles' rows, and bit-reproducible (slabs summed in a fixed order)."""
    from tomosar2height_amd import _lib, mlp
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(m + k + n)
    wide_y = torch.randn(m, n + 64, generator=g)
    wide_x = torch.randn(m, k + 32, generator=g)
    if m >= 4096:                                        # row blocks of very different size: every unit has its own scale
        wide_y[m // 4: m // 2] *= 2.0 ** 10
        wide_x[m // 2: 3 * m // 4] *= 2.0 ** -10
    dy, x = wide_y.to(dev)[:, 32:32 + n], wide_x.to(dev)[:, 16:16 + k]
    want_w = dy.double().cpu().t() @ x.double().cpu()
    want_b = dy.double().cpu().sum(0)
    monkeypatch.setattr(mlp, "_GEMM_BX3_WGRAD", True)
    monkeypatch.setattr(mlp, "_BX3_WGRAD_MIN_N", 0)
    monkeypatch.setattr(mlp, "_BX3_WGRAD_MIN_M", 0)
    dw, db = torch.full((n, k), 7.0, device=dev), torch.full((n,), 7.0, device=dev)
    with _lib.KernelTimeline() as tl:
        mlp.linear_wgrad_(dy, x, dw, db)
    torch.cuda.synchronize()
    assert any(r[5] == f"bx3_wgrad_kernel<2,2,false,{k // 32}>" for r in tl.records), [r[5] for r in tl.records]
    sw, sb = want_w.abs().max().item(), want_b.abs().max().item()
    assert (dw.double().cpu() - want_w).abs().max().item() <= 2e-5 * sw
    assert (db.double().cpu() - want_b).abs().max().item() <= 2e-5 * sb
    # accumulate on top of what is there; the same bits every time
    dw2, db2 = dw.clone(), db.clone()
    mlp.linear_wgrad_(dy, x, dw2, db2, accumulate=True)
    assert (dw2.double().cpu() - 2 * want_w).abs().max().item() <= 4e-5 * sw
    assert (db2.double().cpu() - 2 * want_b).abs().max().item() <= 4e-5 * sb
    dw3, db3 = torch.empty_like(dw), torch.empty_like(db)
    mlp.linear_wgrad_(dy, x, dw3, db3)
    assert torch.equal(dw3, dw) and torch.equal(db3, db)
    # against the fp32 MFMA kernel it replaces (same tolerance class)
    monkeypatch.setattr(mlp, "_GEMM_BX3_WGRAD", False)
    dw4, db4 = torch.empty_like(dw), torch.empty_like(db)
    mlp.linear_wgrad_(dy, x, dw4, db4)
    assert (dw4.double().cpu() - want_w).abs().max().item() <= 2e-5 * sw
    assert (dw4 - dw).abs().max().item() <= 4e-5 * sw


def test_batched_slab_reductions_are_bit_identical():
    """_lib.reduce_capture: weight-gradient calls that opt in record their slab reduction and the block's exit runs them in one
    launch -- the same summation tree per output, so dw / db equal the immediately reduced ones bit for bit; two calls into the
    SAME output inside one capture are never batched together (the first is flushed)."""
    from tomosar2height_amd import _lib, grid, mlp
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    shapes = [(131072, 64, 128), (65536, 320, 64), (4096, 256, 512)]
    data = [(torch.randn(m, n, generator=g).to(dev), torch.randn(m, k, generator=g).to(dev)) for m, k, n in shapes]
    x = torch.randn(1, 64, 128, 128, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(1, 128, 128, 128, generator=g).to(dev).contiguous(memory_format=torch.channels_last)

    def run(capture):
        outs = [(torch.full((n, k), 0.5, device=dev), torch.full((n,), 0.25, device=dev)) for m, k, n in shapes]
        cw = torch.ones(128, 64, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
        cb = torch.ones(128, device=dev)
        with _lib.reduce_capture(capture):
            for (dy, xx), (dw, db) in zip(data, outs):
                mlp.linear_wgrad_(dy, xx, dw, db, accumulate=True, defer=True)
            grid.conv3x3_wgrad_(gy, x, cw, cb, accumulate=True, defer=True)
            if capture:
                assert _lib.load().t2h_reduce_capture_pending() >= 3
            mlp.linear_wgrad_(data[0][0], data[0][1], outs[0][0], outs[0][1], accumulate=True, defer=True)   # same output again
        assert _lib.load().t2h_reduce_capture_pending() == -1
        torch.cuda.synchronize()
        return [t for pair in outs for t in pair] + [cw, cb]

    for a, b in zip(run(False), run(True)):
        assert torch.equal(a, b)


def test_split_weights_filled_on_one_stream_are_waited_for_on_another():
    """r06: a SplitWeightCache entry is filled lazily by the stream that first needs it; another stream's first use waits for the
    event behind that fill (_lib.Ready).  Here the fill is queued behind a 0.1 s spin on stream a and the product runs on stream b
    at once: without the wait it multiplies by whatever the fresh buffer held."""
    from tomosar2height_amd import grid, mlp
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4096, 1664, generator=g).to(dev)
    w = torch.randn(512, 1664, generator=g).to(dev)
    ref = (x.double() @ w.double().t())
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    assert mlp._bx3_gemm_ok(4096, 1664, 512, x)
    with torch.cuda.stream(a):
        torch.cuda._sleep(int(3e8))
        grid.split_weights.get_gemm(w, False)
    with torch.cuda.stream(b):
        y = torch.empty(4096, 512, device=dev)
        mlp.linear_fwd_(x, w, None, y, bx3=True)
    torch.cuda.synchronize()
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err
    # the conv form and the transposed-convolution form go through the same guard
    wc = torch.randn(64, 64, 3, 3, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    xi = torch.randn(1, 64, 64, 64, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(int(3e8))
        grid.split_weights.get(wc, False)
    with torch.cuda.stream(b):
        yc = torch.empty(1, 64, 64, 64, device=dev).contiguous(memory_format=torch.channels_last)
        grid.conv3x3_fwd_(xi, wc, None, yc)
    torch.cuda.synchronize()
    refc = torch.nn.functional.conv2d(xi.double().cpu(), wc.double().cpu(), padding=1)
    errc = float((yc.double().cpu() - refc).abs().max() / refc.abs().max())
    assert errc < 1e-5, errc


def test_ready_orders_a_fill_before_another_streams_first_use():
    """_lib.Ready by itself: a buffer filled behind a 0.1 s spin on stream a, marked, read on stream b after wait() -- the reader sees
    the fill; a second wait() on b and any wait() on a are look-ups only (no event is waited for twice)."""
    from tomosar2height_amd import _lib
    dev = torch.device("cuda:0")
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    buf = torch.zeros(1 << 20, device=dev)
    torch.cuda.synchronize()
    r = _lib.Ready()
    r.wait()                                                   # nothing marked yet: a no-op
    with torch.cuda.stream(a):
        torch.cuda._sleep(int(3e8))
        buf.fill_(3.0)
        r.mark()
        r.wait()
        assert r.seen == {_lib.stream()}
    with torch.cuda.stream(b):
        r.wait()
        got = buf.sum()
        assert _lib.stream() in r.seen and len(r.seen) == 2
        ev = r.event
        r.wait()
        assert r.event is ev
    torch.cuda.synchronize()
    assert float(got) == 3.0 * (1 << 20)
