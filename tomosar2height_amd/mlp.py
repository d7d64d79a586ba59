"""Per-point MLP arithmetic on the MI355X: the dense side of the hot path (reference: pointnet.py:36-40,72-82,
block/resnet.py:26-54, alto.py:63-69,121-128,164-170,245-253) as explicit forward/backward chains over the
fp32 MFMA kernels of include/t2h.h (t2h_linear_fwd / dgrad / wgrad) and the strided pool kernels.

Rows are points in the tile's cell-sorted order.  Every elementwise neighbour of a Linear layer is folded into a
kernel prologue/epilogue (bias, ReLU, ReLU-mask of the backward, residual/shortcut accumulation, bias gradient),
and the reference's ``torch.cat([net, pooled], dim=2)`` (pointnet.py:78) never materialises: each ResNet block
writes its output into the left half of the next block's [M, 2h] input buffer and ``pool_max`` writes the right
half in place.  There is no CPU path here: all tensors must live on the device.
"""
import os
from typing import Optional

import torch

from . import _lib


# ------------------------------------------------------------------------------------------------ precision
_MODE = "fp32"


def set_precision(name: str):
    """'fp32' (default; exact fp32 MFMA, BASELINE.json configs[1]) or 'bf16' (operands rounded to bf16 while staging,
    fp32 accumulate, fp32 tensors in HBM: configs[2] "bf16 MLP GEMMs on MFMA") or 'bf16x3' (opt-in: fp32-grade products
    from an exact 3-way bf16 split of both operands, six bf16 MFMAs per product).  Process-wide; applies to the forward
    and backward GEMMs of the ALTO per-point layers (alto.py:63-69) and of the block-by-block trunk.  The FUSED PointNet
    trunk (csrc/trunk.hip, the default for the shipped widths; 1.3 % of the per-point flops) and fc_pos (K = 3) always
    compute in exact fp32 -- more accurate than the mode asks for, never less; ``trunk_precision()`` reports what a
    forward will use, and bench.py prints it."""
    global _MODE
    if name not in ("fp32", "bf16", "bf16x3"):
        raise ValueError("precision must be 'fp32', 'bf16' or 'bf16x3'")
    _MODE = name


def get_precision() -> str:
    return _MODE


def trunk_precision() -> str:
    """Arithmetic of the PointNet trunk GEMMs under the current mode: the fused block kernel has one (exact fp32) form."""
    return "fp32 (fused trunk block kernel)" if (FUSED_TRUNK or _MODE == "fp32") else _MODE


def _pflag() -> int:
    return {"fp32": 0, "bf16": _lib.BF16, "bf16x3": _lib.BF16X3}[_MODE]


# ------------------------------------------------------------------------------------------------ kernel calls
def _rows(t: torch.Tensor, what: str):
    """A 2-D fp32 device view whose rows are contiguous (row stride >= width): returns (ptr, ld)."""
    if t.dim() != 2 or t.dtype != torch.float32 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError(f"{what}: expected a 2-D float32 tensor with unit inner stride, got {tuple(t.shape)} "
                         f"strides {t.stride()}")
    if not t.is_cuda:
        raise RuntimeError(f"{what}: expected device tensors; tomosar2height_amd has no CPU path "
                           "(the CPU restatement lives in oracle/ for tests only)")
    return t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


# Grid-side products of the deferred point update (deferred.py) on the bf16 matrix cores with the exact 3-way split
# (csrc/conv_bx3.hip, 1-tap form; fp32-grade like the convolutions): T2H_GEMM_BX3=0 keeps them on the fp32 MFMA kernels (A/B)
GEMM_BX3 = os.environ.get("T2H_GEMM_BX3", "1") != "0"


_BX3_MIN_N = int(os.environ.get("T2H_GEMM_BX3_MIN_N", "64"))
_BX3_MIN_K = int(os.environ.get("T2H_GEMM_BX3_MIN_K", "128"))
# (r06: the transposed product dW = dY^T X of those matrices on the split kernels too, t2h_gemm_bx3_wgrad.  OFF by default: 215 against
# 232 us per tile on the r = 256 level's [N = 2752, K = 64] at four tiles per launch, 248 against 280 at one -- its 32-row units of
# 256-byte row segments keep it at 3.4 TB/s, not at the HBM floor the arithmetic would allow; profiles/r06_level256_probe.txt)
_GEMM_BX3_WGRAD = os.environ.get("T2H_GEMM_BX3_WGRAD", "0") == "1"
_BX3_WGRAD_MIN_N = int(os.environ.get("T2H_GEMM_BX3_WGRAD_MIN_N", "1024"))
_BX3_WGRAD_MIN_M = int(os.environ.get("T2H_GEMM_BX3_WGRAD_MIN_M", "4096"))
_PERSIST_N = os.environ.get("T2H_BX3_PERSIST_N", "1") != "0"


def _bx3_gemm_ok(m, k, n, *rows, force=False) -> bool:
    """Where the split form (csrc/conv_bx3.hip, 1-tap) wins, measured.  Every staged A element has to be split and is then used for
    n outputs, and the kernel streams A with one chunk in flight per workgroup: with the bf16 three-way split narrow outputs lost
    (the r = 256 level product 2752 -> 64: 250 us on fp32 MFMA, 278 us split); with the fp16 two-way split and the weights fetched
    straight into registers (r04d) it is 197 us (profiles/gemm_layout_probe.py), so 64 outputs are in (r05: the threshold itself had
    stayed at 128; at 64 the data gradient of the stacked per-pixel product, 2752 -> 64 over 65 536 rows, goes from 245 to 208 us).
    One-chunk reductions that
    only WRITE a wide matrix (64 -> 2752: 274 us split against 247 us) stay on the fp32 kernels."""
    if not (GEMM_BX3 and _MODE == "fp32" and bool(_lib.ws_bytes("t2h_gemm_bx3_supported", m, k, n))
            and all(t is None or (t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0) for t in rows)):
        return False
    min_n, min_k = _BX3_MIN_N, _BX3_MIN_K
    # r05: a ONE-chunk reduction (k = 64) into a wide matrix goes to the split kernel's persistent form (conv_bx3.hip, PERSIST: the
    # rows are staged and split once per workgroup and reused for all its column tiles); T2H_BX3_PERSIST_N=0: A/B
    wide_short = _PERSIST_N and k == 64 and n >= 512 and m >= 4096
    return force or wide_short or (n >= min_n and k >= min_k and m >= 4096)


def _gemm_bx3(x, w, w_is_kn, bias, mask, y, relu_out, accumulate, tag):
    from . import grid
    (xp, ldx), (yp, ldy) = _rows(x, "gemm_bx3 x"), _rows(y, "gemm_bx3 y")
    m, k = x.shape
    n = y.shape[1]
    mp, ldm = (None, 0) if mask is None else _rows(mask, "gemm_bx3 mask")
    nws = _lib.ws_bytes("t2h_gemm_bx3_workspace_bytes", m, k, n)
    ws = _lib.workspace(nws, x.device)
    flags = (_lib.RELU_OUT if relu_out else 0) | (_lib.ACCUM if accumulate else 0) | (_lib.F16X2 if grid._h2() else 0)
    _lib.call("t2h_gemm_bx3", xp, ldx, _lib.ptr(grid.split_weights.get_gemm(w, w_is_kn)), bias.data_ptr() if bias is not None else None,
              mp, ldm, yp, ldy, m, k, n, flags, _lib.ptr(ws), nws, _lib.stream(),
              nbytes=4 * (m * k + m * n + n * k + (m * n if mask is not None else 0)), flops=2 * m * k * n, tag=_lib.timing() and tag)
    return y


def linear_fwd_(x, w, bias, y, relu_in=False, relu_out=False, accumulate=False, bx3=False):
    """y = [y +] act_out(act_in(x) w^T + bias), written in place into ``y`` (may be a column slice).  ``bx3``: the caller's
    product is a grid-side one with a weight that lives across calls (split once): take the split-bf16 matrix-core kernel."""
    (xp, ldx), (yp, ldy) = _rows(x, "linear_fwd x"), _rows(y, "linear_fwd y")
    m, k = x.shape
    n = w.shape[0]
    if bx3 and not relu_in and w.is_contiguous() and _bx3_gemm_ok(m, k, n, x, y):
        return _gemm_bx3(x, w, False, bias, None, y, relu_out, accumulate, f"t2h_linear_fwd[K={k},N={n}]")
    w = w.contiguous()
    flags = (_lib.RELU_IN if relu_in else 0) | (_lib.RELU_OUT if relu_out else 0) | (_lib.ACCUM if accumulate else 0) | _pflag()
    _lib.call("t2h_linear_fwd", xp, ldx, w.data_ptr(), bias.data_ptr() if bias is not None else None, yp, ldy, m, k, n,
              flags, _lib.stream(), nbytes=4 * (m * k + m * n + n * k), flops=2 * m * k * n,
              tag=_lib.timing() and f"t2h_linear_fwd[K={k},N={n}]")
    return y


def linear_dgrad_(dy, w, dx, mask=None, accumulate=False, bx3=False):
    """dx = [dx +] (dy w) * (mask > 0), in place into ``dx``.  ``bx3``: see ``linear_fwd_``."""
    (gp, ldg), (dp, ldd) = _rows(dy, "linear_dgrad dy"), _rows(dx, "linear_dgrad dx")
    m, n = dy.shape
    k = w.shape[1]
    if bx3 and w.is_contiguous() and _bx3_gemm_ok(m, n, k, dy, dx, mask):
        return _gemm_bx3(dy, w, True, None, mask, dx, False, accumulate, f"t2h_linear_dgrad[N={n},K={k}]")
    w = w.contiguous()
    mp, ldm = (None, 0) if mask is None else _rows(mask, "linear_dgrad mask")
    _lib.call("t2h_linear_dgrad", gp, ldg, w.data_ptr(), dp, ldd, m, k, n, mp, ldm,
              (_lib.ACCUM if accumulate else 0) | _pflag(),
              _lib.stream(), nbytes=4 * (m * k + m * n + n * k + (m * k if mask is not None else 0)),
              flops=2 * m * k * n, tag=_lib.timing() and f"t2h_linear_dgrad[N={n},K={k}]")
    return dx


def _grid_h2() -> bool:
    from . import grid
    return grid._h2()


def linear_wgrad_(dy, x, dw, db, relu_in=False, accumulate=False, defer=False):
    """dw = [dw +] dy^T act_in(x); db = [db +] colsum(dy) (db may be None); deterministic split reduction.  ``defer``: dw / db are
    buffers nobody reads before the backward pass ends: the reduction may join the pass's batched one (``_lib.reduce_capture``)."""
    (gp, ldg), (xp, ldx) = _rows(dy, "linear_wgrad dy"), _rows(x, "linear_wgrad x")
    m, n = dy.shape
    k = x.shape[1]
    if not dw.is_contiguous() or (db is not None and not db.is_contiguous()):
        raise ValueError("linear_wgrad: dw / db must be contiguous")
    if m == 0:
        if not accumulate:
            dw.zero_()
            if db is not None:
                db.zero_()
        return
    if (_GEMM_BX3_WGRAD and GEMM_BX3 and _MODE == "fp32" and not relu_in and n >= _BX3_WGRAD_MIN_N and m >= _BX3_WGRAD_MIN_M
            and _grid_h2() and bool(_lib.ws_bytes("t2h_gemm_bx3_wgrad_supported", m, k, n))):
        # r06: the wide grid-side products' weight gradients (the per-resolution sum matrices of the deferred update: N = 2368 ..
        # 2752 columns, K = 64 .. 256) on the split kernels -- fp32-grade products from three fp16 MFMAs like their forward and data
        # gradient (bx3_wgrad_kernel<.., GT = K / 32>; T2H_GEMM_BX3_WGRAD=1)
        ws_bytes = _lib.ws_bytes("t2h_gemm_bx3_wgrad_workspace_bytes", m, k, n)
        ws = _lib.workspace(ws_bytes, dy.device)
        flags = (_lib.ACCUM if accumulate else 0) | _lib.F16X2 | (_lib.defer_reduce(ws, dw) if defer else 0)
        _lib.call("t2h_gemm_bx3_wgrad", gp, ldg, xp, ldx, m, k, n, dw.data_ptr(), db.data_ptr() if db is not None else None, flags,
                  ws.data_ptr(), ws_bytes, _lib.stream(), nbytes=4 * (m * k + m * n + n * k), flops=2 * m * k * n,
                  tag=_lib.timing() and f"t2h_linear_wgrad[N={n},K={k}]")
        return
    ws_bytes = _lib.ws_bytes("t2h_linear_wgrad_workspace_bytes", m, k, n)
    ws = _lib.workspace(ws_bytes, dy.device)
    flags = (_lib.RELU_IN if relu_in else 0) | (_lib.ACCUM if accumulate else 0) | _pflag() | (_lib.defer_reduce(ws, dw) if defer else 0)
    _lib.call("t2h_linear_wgrad", gp, ldg, xp, ldx, m, k, n, flags, dw.data_ptr(), db.data_ptr() if db is not None else None,
              ws.data_ptr(), ws_bytes, _lib.stream(), nbytes=4 * (m * k + m * n + n * k), flops=2 * m * k * n,
              tag=_lib.timing() and f"t2h_linear_wgrad[N={n},K={k}]")


_DIRECT_ACCUM = False
_WGRAD_STREAM = None
_CONV_WGRAD_STREAM = None        # side stream for the convolution weight gradients (grid._conv3x3_param_grads)
_MAIN_STREAM = None              # torch's current stream when the block was entered (what the side streams fork from)


_HELD = []                       # tensors read by side-stream launches of the current pass: released after the streams have joined


def hold(*tensors):
    """Keep ``tensors`` (inputs of a launch on a side stream) alive until the pass's streams have joined
    (``direct_grad_accumulation.__exit__``), instead of ``record_stream``: the caching allocator then sees every free in the main
    stream's order -- no per-block event bookkeeping, no blocks parked behind unfinished side-stream work (with 8 tiles per
    micro-batch the parked blocks made the allocator grow mid-run: 5.7 vs 13 ms per tile, run to run)."""
    _HELD.extend(tensors)


def fork_to(side):
    """Order ``side`` behind everything issued so far on the stream the backward pass runs on, then launch there:
    ``with mlp.fork_to(side): ...`` (``_lib.on_stream``: the cheap form of ``torch.cuda.stream``)."""
    main = _MAIN_STREAM if _MAIN_STREAM is not None else torch.cuda.current_stream()
    side.wait_stream(main)
    return _lib.on_stream(side, main)


class direct_grad_accumulation:
    """While active, weight/bias gradients of the per-point layers are accumulated by the wgrad kernel straight into
    an existing contiguous ``param.grad`` (e.g. the Trainer's flat bucket views) and the autograd Function returns
    ``None`` for them -- this removes one elementwise add launch per parameter per tile.  Off by default so that
    ``torch.autograd.grad`` and first-touch (``grad is None``) semantics stay the standard ones."""

    def __init__(self, enabled: bool = True, side_stream=None, conv_side_stream=None):
        """``side_stream``: issue the accumulating weight-gradient GEMMs there.  They are off the backward's critical
        path (nothing reads the bucket before the optimizer step), so they fill the GPU while small-grid conv kernels
        run on the main stream.  ``conv_side_stream``: the same for the weight gradients of convolutions on small planes.
        The caller joins the streams before touching the gradients."""
        self.enabled = enabled
        self.side_stream = side_stream if enabled else None
        self.conv_side_stream = conv_side_stream if enabled else None

    def __enter__(self):
        global _DIRECT_ACCUM, _WGRAD_STREAM, _CONV_WGRAD_STREAM, _MAIN_STREAM
        self.prev, _DIRECT_ACCUM = _DIRECT_ACCUM, self.enabled
        self.prev_stream, _WGRAD_STREAM = _WGRAD_STREAM, self.side_stream
        self.prev_conv_stream, _CONV_WGRAD_STREAM = _CONV_WGRAD_STREAM, self.conv_side_stream
        self.prev_main = _MAIN_STREAM
        if self.side_stream is not None or self.conv_side_stream is not None:
            _MAIN_STREAM = torch.cuda.current_stream()          # (once per pass: the autograd engine runs the backward on it)
        return self

    def __exit__(self, *exc):
        global _DIRECT_ACCUM, _WGRAD_STREAM, _CONV_WGRAD_STREAM, _MAIN_STREAM
        if _MAIN_STREAM is not None and self.prev_main is None:
            # join: afterwards the gradients are visible in stream order on the main stream like any other result, and what the
            # side-stream launches read may be recycled
            for st in {id(s): s for s in (self.side_stream, self.conv_side_stream) if s is not None}.values():
                _MAIN_STREAM.wait_stream(st)
            _HELD.clear()
        _DIRECT_ACCUM = self.prev
        _WGRAD_STREAM = self.prev_stream
        _CONV_WGRAD_STREAM = self.prev_conv_stream
        _MAIN_STREAM = self.prev_main


def _wgrad(dy, x, w, bias, relu_in=False):
    if (_DIRECT_ACCUM and w.shape[0] % 4 == 0 and w.grad is not None and w.grad.is_contiguous()
            and (bias is None or (bias.grad is not None and bias.grad.is_contiguous()))):
        side = _WGRAD_STREAM
        if side is None:
            linear_wgrad_(dy, x, w.grad, None if bias is None else bias.grad, relu_in=relu_in, accumulate=True, defer=True)
        else:
            with fork_to(side):                                     # dy / x are produced on the main stream
                linear_wgrad_(dy, x, w.grad, None if bias is None else bias.grad, relu_in=relu_in, accumulate=True, defer=True)
            hold(dy, x)                                             # keep the allocator from recycling them early
        return None, None
    if w.shape[0] % 4 != 0:
        # odd output widths (the 1-channel head of the non-default per-pixel FC decoder, pixel.py:51) are off the
        # per-point hot path: library GEMM on the device
        _lib.library_fallback(f"linear wgrad with {w.shape[0]} output column(s) (rocBLAS)")
        xa = torch.relu(x) if relu_in else x
        return dy.t() @ xa, (dy.sum(0) if bias is not None else None)
    dw = torch.empty_like(w, memory_format=torch.contiguous_format)
    db = torch.empty_like(bias) if bias is not None else None
    linear_wgrad_(dy, x, dw, db, relu_in=relu_in)
    return dw, db


def _pool_rows_ok(c, *lds):
    """The row-balanced pooling kernels (t2h_pool_rows_*) serve 16-byte rows of up to 64 channels."""
    return c % 4 == 0 and c <= 64 and all(ld % 4 == 0 for ld in lds)


def _pool_fwd_(tile, feat, pooled, winner):
    (fp, ldf), (pp, ldp) = _rows(feat, "pool feat"), _rows(pooled, "pool out")
    c = feat.shape[1]
    # forward: the cell-parallel kernel is the faster one (13 vs 24 us at the bench shape); backward: the row-balanced one
    _lib.call("t2h_pool_max_fwd", fp, ldf, _lib.ptr(tile.off0), tile.B, tile.nbits, c, pp, ldp, _lib.ptr(winner),
              _lib.stream(), nbytes=8 * c * tile.n_points + 4 * tile.n_points)


def _pool_bwd_(tile, gpooled, winner, gfeat, accumulate):
    (gp, ldg), (op, ldo) = _rows(gpooled, "pool gpooled"), _rows(gfeat, "pool gfeat")
    c = gpooled.shape[1]
    nbytes = 8 * c * tile.n_points + 4 * tile.n_points
    if _pool_rows_ok(c, ldg, ldo):
        _lib.call("t2h_pool_rows_bwd", gp, ldg, _lib.ptr(winner), _lib.ptr(tile.cell), _lib.ptr(tile.off0), tile.n_points, c,
                  1 if accumulate else 0, op, ldo, _lib.stream(), nbytes=nbytes, tag="t2h_pool_max_bwd")
    else:
        _lib.call("t2h_pool_max_bwd", gp, ldg, _lib.ptr(winner), _lib.ptr(tile.off0), tile.B, tile.nbits, c,
                  1 if accumulate else 0, op, ldo, _lib.stream(), nbytes=nbytes)


def _pool_mean_(tile, src, dst, accumulate):
    """dst (=|+=) per-cell mean of src, scatter_type='mean' (pointnet.py:55-56); its own adjoint, so also the backward."""
    (sp, lds), (dp, ldd) = _rows(src, "pool src"), _rows(dst, "pool dst")
    c = src.shape[1]
    _lib.call("t2h_pool_mean", sp, lds, _lib.ptr(tile.off0), tile.B, tile.nbits, c, 1 if accumulate else 0, dp, ldd,
              _lib.stream(), nbytes=(8 + (4 if accumulate else 0)) * c * tile.n_points + 4 * tile.n_points)


def _empty(rows, cols, like):
    return torch.empty(rows, cols, dtype=torch.float32, device=like.device)


# ------------------------------------------------------------------------------------------------ nn.Linear
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, relu_in: bool):
        y = _empty(x.shape[0], w.shape[0], x)
        linear_fwd_(x, w, bias, y, relu_in=relu_in)
        ctx.save_for_backward(x, w, bias)
        ctx.relu_in = relu_in
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, bias = ctx.saved_tensors
        gy = gy.contiguous()
        if w.shape[0] == 1 and w.shape[1] % 4 == 0 and x.is_contiguous():
            # one output column (fc_out of the per-pixel FC decoder, pixel.py:51,56: Linear(32, 1) over 512 x 512 pixel rows): a
            # rank-1 product per row -- the concat-free head's kernels with ONE input (t2h_head1x1_bwd: dx = g w masked by the
            # ReLU, dw = sum_p g[p] a[p, :], db = sum_p g[p], deterministic two-stage sums) instead of a library GEMM
            return _linear1_backward(ctx, x, w, bias, gy)
        dx = None
        if ctx.needs_input_grad[0]:
            if w.shape[1] % 4 == 0 and w.shape[0] % 4 == 0:
                dx = linear_dgrad_(gy, w, torch.empty_like(x), mask=x if ctx.relu_in else None)
            else:   # odd widths never occur on the network's hot path; keep the generic seam correct
                _lib.library_fallback(f"linear dgrad {w.shape[0]}->{w.shape[1]} (rocBLAS)")
                dx = gy @ w
                if ctx.relu_in:
                    dx = dx * (x > 0)
        dw, db = _wgrad(gy, x, w, bias, relu_in=ctx.relu_in)
        return dx, dw, db, None


def _linear1_backward(ctx, x, w, bias, gy):
    import ctypes
    m, k = x.shape
    a = x
    if ctx.relu_in:                                    # the head kernels take the ReLU OUTPUT: a = x * (x > 0), one pass
        a = torch.empty_like(x)
        _lib.call("t2h_relu_mask", _lib.ptr(x), _lib.ptr(x), _lib.ptr(a), x.numel(), _lib.stream(), nbytes=12 * x.numel())
    dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
    dw = torch.empty(k, dtype=torch.float32, device=x.device)
    db = torch.empty(1, dtype=torch.float32, device=x.device) if bias is not None else None
    ws_bytes = _lib.ws_bytes("t2h_head1x1_bwd_workspace_bytes", m, k)
    ws = _lib.workspace(ws_bytes, x.device)
    xarr = (ctypes.c_void_p * 1)(a.data_ptr())
    dxarr = (ctypes.c_void_p * 1)(dx.data_ptr() if dx is not None else None)
    carr = (ctypes.c_int * 1)(k)
    _lib.call("t2h_head1x1_bwd", ctypes.cast(xarr, ctypes.c_void_p), ctypes.cast(dxarr, ctypes.c_void_p),
              ctypes.cast(carr, ctypes.c_void_p), 1, _lib.ptr(w.reshape(-1).contiguous()), _lib.ptr(gy), m,
              (1 << 8) if ctx.relu_in else 0, _lib.ptr(dw), None if db is None else _lib.ptr(db), _lib.ptr(ws), ws_bytes,
              _lib.stream(), nbytes=4 * (3 * k + 1) * m)
    return dx, dw.reshape(w.shape), db, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], relu_in: bool = False) -> torch.Tensor:
    """``F.linear(relu(x) if relu_in else x, weight, bias)`` over the last dim (any leading dims)."""
    lead = x.shape[:-1]
    y = _Linear.apply(x.reshape(-1, x.shape[-1]).contiguous(), weight, bias, relu_in)
    return y.reshape(*lead, weight.shape[0])


def sole_owner(g) -> bool:
    """True if the gradient tensor ``g`` a backward() received is referenced by nobody but the engine's hand-off to THIS node, so
    that accumulating into it in place cannot be seen elsewhere.  Not so when a consumer of the forward value returned its own
    incoming gradient unchanged to several inputs (``z = y + other``: torch hands the SAME tensor to both producers) -- the
    buffer is then still queued for another node.  The C++ reference count tells: 2 = the engine's input list + the Python
    argument; the same for the base of a view (a channels_last gradient is a permuted view of its NHWC buffer)."""
    if g._use_count() > 2:
        return False
    base = g._base
    return base is None or base._use_count() <= 2


def _join_plane_grad(dy, w, gthru, shape_nhwc, was_cl):
    """dplane = dy w (rows -> the NHWC plane) + gthru: the gradient of the plane's other consumers joins inside the data-gradient
    kernel (accumulating epilogue) where it arrives as a dense NHWC tensor that nobody else still reads (``sole_owner``), else
    by one add into a fresh tensor."""
    from . import ops
    b, r1, r2, c = shape_nhwc
    if (gthru is not None and gthru.dtype == torch.float32 and gthru.permute(0, 2, 3, 1).is_contiguous() and was_cl
            and sole_owner(gthru)):
        linear_dgrad_(dy, w, gthru.permute(0, 2, 3, 1).reshape(b * r1 * r2, c), accumulate=True)
        return gthru
    drows = linear_dgrad_(dy, w, torch.empty(b * r1 * r2, c, dtype=torch.float32, device=dy.device))
    dplane = drows.reshape(b, r1, r2, c)
    if gthru is not None:
        dplane = dplane + ops.to_nhwc(gthru)
    return ops.from_nhwc(dplane, was_cl)


class _LinearPlaneThru(torch.autograd.Function):
    """``(q, plane)``: q [B H W, N] = the pixels of ``plane`` [B, C, H, W] as rows through an nn.Linear (fc_comm.0 applied on the
    grid, deferred.py); the second output is the plane itself for its other consumers, whose gradient joins this node's
    inside its data-gradient kernel instead of by an autograd add over the plane."""

    @staticmethod
    def forward(ctx, plane, w, bias):
        from . import ops
        ctx.was_cl = ops._is_channels_last(plane)
        p = ops.to_nhwc(plane)
        _lib.require_device(p, what="linear_plane_thru")
        b, h, wd, c = p.shape
        rows = p.reshape(b * h * wd, c)
        q = _empty(rows.shape[0], w.shape[0], rows)
        linear_fwd_(rows, w, bias, q)
        ctx.shape = (b, h, wd, c)
        ctx.save_for_backward(rows, w, bias)
        return q, ops._alias(plane)

    @staticmethod
    def backward(ctx, gq, gthru):
        rows, w, bias = ctx.saved_tensors
        if gq is None:
            return gthru, None, None
        gq = gq.contiguous()
        dplane = gthru
        if ctx.needs_input_grad[0]:
            dplane = _join_plane_grad(gq, w, gthru, ctx.shape, ctx.was_cl)
        dw, db = _wgrad(gq, rows, w, bias)
        return dplane, dw, db


def linear_plane_thru(plane, weight, bias):
    """-> (q rows [B H W, N], plane): use the returned plane for every further consumer of the input plane."""
    return _LinearPlaneThru.apply(plane, weight, bias)


# ------------------------------------------------------------------------------------------------ ResnetBlockFC
def _resblock_fwd(x, w0, b0, w1, b1, ws, out):
    """out (a [M, Cout] view) = shortcut(x) + fc_1(relu(fc_0(relu(x)))); returns Hr = relu(fc_0(relu(x)))."""
    hr = _empty(x.shape[0], w0.shape[0], x)
    linear_fwd_(x, w0, b0, hr, relu_in=True, relu_out=True)
    if ws is not None:
        linear_fwd_(x, ws, None, out)
    else:
        out.copy_(x)
    linear_fwd_(hr, w1, b1, out, accumulate=True)
    return hr


def _resblock_bwd(gy, x, hr, w0, b0, w1, b1, ws, need_dx=True):
    """Gradients of one block; returns (dx [M,Cin] or None, dw0, db0, dw1, db1, dws)."""
    dw1, db1 = _wgrad(gy, hr, w1, b1)
    dhr = linear_dgrad_(gy, w1, torch.empty_like(hr), mask=hr)        # through fc_1 and the ReLU in front of it
    dw0, db0 = _wgrad(dhr, x, w0, b0, relu_in=True)
    dws = _wgrad(gy, x, ws, None)[0] if ws is not None else None
    dx = None
    if need_dx:
        dx = _empty(x.shape[0], x.shape[1], x)
        if ws is not None:
            linear_dgrad_(gy, ws, dx)
        else:
            dx.copy_(gy)
        linear_dgrad_(dhr, w0, dx, mask=x, accumulate=True)           # through fc_0 and relu(x)
    return dx, dw0, db0, dw1, db1, dws


class _ResBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, ws):
        out = _empty(x.shape[0], w1.shape[0], x)
        hr = _resblock_fwd(x, w0, b0, w1, b1, ws, out)
        ctx.save_for_backward(x, hr, w0, b0, w1, b1, ws)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, hr, w0, b0, w1, b1, ws = ctx.saved_tensors
        dx, dw0, db0, dw1, db1, dws = _resblock_bwd(gy.contiguous(), x, hr, w0, b0, w1, b1, ws, ctx.needs_input_grad[0])
        return dx, dw0, db0, dw1, db1, dws


def resblock(x: torch.Tensor, w0, b0, w1, b1, ws) -> torch.Tensor:
    """block/resnet.py:36-54 over the last dim."""
    lead = x.shape[:-1]
    y = _ResBlock.apply(x.reshape(-1, x.shape[-1]).contiguous(), w0, b0, w1, b1, ws)
    return y.reshape(*lead, w1.shape[0])


# ------------------------------------------------------------------------------------------------ ALTO point update
class _CommMLP(torch.autograd.Function):
    """c = fc_comm.2(relu(fc_comm.0(sampled))) + fc_c(c_last)   (alto.py:121-128, 245-253)."""

    @staticmethod
    def forward(ctx, sampled, c_last, wa, ba, wb, bb, wc, bc):
        m = sampled.shape[0]
        h = _empty(m, wa.shape[0], sampled)
        linear_fwd_(sampled, wa, ba, h, relu_out=True)
        out = _empty(m, wb.shape[0], sampled)
        linear_fwd_(h, wb, bb, out)
        if c_last is not None:
            linear_fwd_(c_last, wc, bc, out, accumulate=True)
        ctx.has_last = c_last is not None
        ctx.save_for_backward(sampled, c_last, h, wa, ba, wb, bb, wc, bc)
        return out

    @staticmethod
    def backward(ctx, g):
        sampled, c_last, h, wa, ba, wb, bb, wc, bc = ctx.saved_tensors
        g = g.contiguous()
        dwb, dbb = _wgrad(g, h, wb, bb)
        dh = linear_dgrad_(g, wb, torch.empty_like(h), mask=h)
        dwa, dba = _wgrad(dh, sampled, wa, ba)
        ds = linear_dgrad_(dh, wa, torch.empty_like(sampled)) if ctx.needs_input_grad[0] else None
        dlast = dwc = dbc = None
        if ctx.has_last:
            dwc, dbc = _wgrad(g, c_last, wc, bc)
            if ctx.needs_input_grad[1]:
                dlast = linear_dgrad_(g, wc, torch.empty_like(c_last))
        return ds, dlast, dwa, dba, dwb, dbb, dwc, dbc


def comm_mlp(sampled, w_a, b_a, w_b, b_b, c_last, w_c, b_c) -> torch.Tensor:
    return _CommMLP.apply(sampled.contiguous(), None if c_last is None else c_last.contiguous(),
                          w_a, b_a, w_b, b_b, w_c, b_c)


# ------------------------------------------------------------------------------------------------ grid-first point update
# relu(fc_comm.0(sample(P))) == relu(sample(fc_comm.0(P))): an nn.Linear commutes with the bilinear interpolation of
# alto.py:90-95 because the interpolation is a linear map over pixels whose four tap weights sum to 1 (padding_mode='border',
# align_corners=True) -- so the bias passes through as well.  Where a level has far fewer pixels than the tile has points
# (Berlin, N = 131072: r = 32 -> 128 points per pixel, r = 64 -> 32, r = 128 -> 8) the first Linear of fc_comm therefore runs
# on the [r^2, C] pixel rows instead of the [N, C] point rows, the sample kernel interpolates the 2C-wide result straight into
# the hidden activations (ReLU in its store), and `sampled` [N, C] never exists.  In the backward the sample's adjoint brings
# the masked hidden gradient back to the pixels, where fc_comm.0's weight / bias / data gradients are again [r^2]-row
# products.  Same function, different association of the fp32 sums (1e-6 relative); per level it removes one of the three
# per-point GEMM triples (forward, data gradient, weight gradient).  GRID_FIRST_MIN_RATIO = points per pixel from which the
# path is taken (0 switches it off: A/B).
GRID_FIRST_MIN_RATIO = float(os.environ.get("T2H_GRID_FIRST_MIN_RATIO", "4"))


def grid_first_applicable(tile, r: int, c: int) -> bool:
    return (GRID_FIRST_MIN_RATIO > 0 and c % 4 == 0 and tile.n_points >= GRID_FIRST_MIN_RATIO * tile.B * r * r
            and tile.R % r == 0)


def hidden_from_plane(tile, plane_rows, r, w_a, b_a):
    """h = relu(sample(P W_a^T + b_a)) -> [N, 2C]; ``plane_rows`` = the [B r r, C] pixel rows of the NHWC plane."""
    q = _empty(plane_rows.shape[0], w_a.shape[0], plane_rows)
    linear_fwd_(plane_rows, w_a, b_a, q)                                       # fc_comm.0 on the pixels
    h = _empty(tile.n_points, w_a.shape[0], plane_rows)
    c2 = w_a.shape[0]
    _lib.call("t2h_sample_fwd_relu", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, tile.B, tile.N, r, c2, _lib.ptr(h), None,
              _lib.stream(), nbytes=4 * c2 * tile.n_points + 8 * tile.n_points + 4 * q.numel(),
              tag=_lib.timing() and f"t2h_sample_fwd_relu[C={c2},r={r}]")
    return h


class _CommMLPGridFirst(torch.autograd.Function):
    """``(c, plane)`` with c = fc_comm.2(relu(fc_comm.0(sample(plane)))) + fc_c(c_last) (alto.py:121-128, 245-253), fc_comm.0
    applied on the grid (see above).  The second output is the plane itself for its other consumers (the next level's residual
    convolution): their gradient ``gthru`` is added to this node's plane gradient here instead of by an autograd pass."""

    @staticmethod
    def forward(ctx, plane, c_last, wa, ba, wb, bb, wc, bc, tile):
        from . import ops
        ctx.was_cl = ops._is_channels_last(plane)
        p = ops.to_nhwc(plane)
        _lib.require_device(p, what="comm_mlp_grid_first")
        b, r, _, c = p.shape
        rows = p.reshape(b * r * r, c)
        h = hidden_from_plane(tile, rows, r, wa, ba)
        out = _empty(tile.n_points, wb.shape[0], rows)
        linear_fwd_(h, wb, bb, out)
        if c_last is not None:
            linear_fwd_(c_last, wc, bc, out, accumulate=True)
        ctx.has_last = c_last is not None
        ctx.tile, ctx.r = tile, r
        ctx.save_for_backward(rows, c_last, h, wa, ba, wb, bb, wc, bc)
        return out, ops._alias(plane)

    @staticmethod
    def backward(ctx, g, gthru):
        from . import ops
        rows, c_last, h, wa, ba, wb, bb, wc, bc = ctx.saved_tensors
        tile, r = ctx.tile, ctx.r
        if g is None:
            return gthru, None, None, None, None, None, None, None, None
        g = g.contiguous()
        dwb, dbb = _wgrad(g, h, wb, bb)
        dh = linear_dgrad_(g, wb, torch.empty_like(h), mask=h)                  # through fc_comm.2 and the ReLU
        dq = ops._sample_bwd(tile, dh, r, h.shape[1], None).reshape(rows.shape[0], h.shape[1])    # S^T dh: [B r r, 2C]
        dwa, dba = _wgrad(dq, rows, wa, ba)                                     # column sums of dq == of dh (taps sum to 1)
        dplane = None
        if ctx.needs_input_grad[0]:
            dplane = _join_plane_grad(dq, wa, gthru, (tile.B, r, r, rows.shape[1]), ctx.was_cl)
        elif gthru is not None:
            dplane = gthru
        dlast = dwc = dbc = None
        if ctx.has_last:
            dwc, dbc = _wgrad(g, c_last, wc, bc)
            if ctx.needs_input_grad[1]:
                dlast = linear_dgrad_(g, wc, torch.empty_like(c_last))
        return dplane, dlast, dwa, dba, dwb, dbb, dwc, dbc, None


def comm_mlp_grid_first(tile, plane, w_a, b_a, w_b, b_b, c_last, w_c, b_c):
    """-> (c [N, C], plane): use the returned plane for every further consumer of the input plane."""
    return _CommMLPGridFirst.apply(plane, None if c_last is None else c_last.contiguous(), w_a, b_a, w_b, b_b, w_c, b_c, tile)


# ------------------------------------------------------------------------------------------------ PointNet trunk
FUSED_TRUNK = os.environ.get("T2H_FUSED_TRUNK", "1") != "0"      # A/B switch: 0 = one GEMM launch per Linear + pool kernels


def _fused_trunk_applicable(pts, params, n_blocks) -> bool:
    """The fused block kernel (csrc/trunk.hip) is built for the reference's trunk widths: hidden_dim = 32, i.e.
    fc_pos 3 -> 64, blocks 64 -> 32 with a shortcut, fc_c 32 -> 32 (tomosar2height.yaml:7-8, pointnet.py:36-40)."""
    if not FUSED_TRUNK or n_blocks < 2 or pts.shape[1] < 3 or tuple(params[0].shape) != (64, 3):
        return False
    for i in range(n_blocks):
        w0, b0, w1, b1, ws = params[2 + 5 * i: 7 + 5 * i]
        if ws is None or tuple(w0.shape) != (32, 64) or tuple(w1.shape) != (32, 32) or tuple(ws.shape) != (32, 64):
            return False
    return tuple(params[-2].shape) == (32, 32)


# r06: the whole trunk forward in one launch (t2h_trunk_fused_fwd, bit-identical to one launch per block).  With one scan per cell in
# its pooling (fused_pool_cells) it is ahead at every size measured: 453 against 726 us at four tiles per launch, 143 against 193 at
# one, 40 against 58 at 7 k rows, the 9 us that build a tile's work units included once (profiles/r06_trunk_fused.txt).
# T2H_TRUNK_FUSED: "auto" (default) = from T2H_TRUNK_FUSED_MIN_ROWS rows on (0: always), "1" always, "0" never (one launch per block)
_TRUNK_FUSED_MODE = os.environ.get("T2H_TRUNK_FUSED", "auto")
_TRUNK_FUSED = _TRUNK_FUSED_MODE != "0"
_TRUNK_FUSED_MIN_ROWS = 0 if _TRUNK_FUSED_MODE == "1" else int(os.environ.get("T2H_TRUNK_FUSED_MIN_ROWS", "0"))
_TRUNK_FUSED_STRIDE = int(os.environ.get("T2H_TRUNK_FUSED_STRIDE", "0"))
_TRUNK_UNIT_BOUNDS = os.environ.get("T2H_TRUNK_UNIT_BOUNDS", "1") != "0"     # greedy units built once per tile index (0: fixed-stride windows looked up in the kernel)


def _trunk_forward_one_launch(tile, pts, w_pos, b_pos, blocks, w_c, b_c):
    """pointnet.py:72-82 as ONE launch: same return value as the per-block form."""
    import ctypes
    m, nb, dev = pts.shape[0], len(blocks), pts.device
    hrs = [_empty(m, 32, pts) for _ in range(nb)]
    nets = [_empty(m, 32, pts) for _ in range(nb)]
    pooled = [None] + [_empty(m, 32, pts) for _ in range(nb - 1)]
    winners = [torch.empty(m, 8, dtype=torch.uint8, device=dev) for _ in range(nb - 1)]
    c_out = _empty(m, 32, pts)
    keep = [t.contiguous() for blk in blocks for t in blk] + [w_pos.contiguous(), w_c.contiguous()]
    arr = ctypes.c_void_p * nb
    params = (ctypes.c_void_p * (5 * nb))(*[t.data_ptr() for t in keep[:5 * nb]])
    a_hr, a_out = arr(*[t.data_ptr() for t in hrs]), arr(*[t.data_ptr() for t in nets])
    a_pool = arr(None, *[t.data_ptr() for t in pooled[1:]])
    a_win = arr(None, *[t.data_ptr() for t in winners])
    # algorithmic bytes: points in; hr, out of every block, pooled + winner bits of every pooling, c out (what the backward reads)
    nbytes = m * (4 * pts.shape[1] + nb * 256 + (nb - 1) * (128 + 8 + 4) + 128)
    flops = 2 * m * (nb * (2 * 64 * 32 + 32 * 32) + 32 * 32 + 3 * 64)
    bounds = tile.trunk_units() if _TRUNK_UNIT_BOUNDS else None
    _lib.call("t2h_trunk_fused_fwd", _lib.ptr(pts), pts.shape[1], _lib.ptr(keep[-2]), _lib.ptr(b_pos), params, nb,
              _lib.ptr(keep[-1]), _lib.ptr(b_c), _lib.ptr(tile.cell), _lib.ptr(tile.off0), m, a_hr, a_out, a_pool, a_win,
              _lib.ptr(c_out), _TRUNK_FUSED_STRIDE, None if bounds is None else _lib.ptr(bounds), _lib.stream(),
              nbytes=nbytes, flops=flops, tag="t2h_trunk_fused_fwd")
    return c_out, nets, pooled, hrs, winners


def _trunk_forward_fused(tile, pts, w_pos, b_pos, blocks, w_c, b_c, want_x_full=False):
    """pointnet.py:72-82 as one launch per block.  Returns (c, nets, pooled, hrs, winners[, x_full]): nets[i] = block i's
    [M, 32] output, pooled[i] (i >= 1) = the pooled half of block i's input (= pool_local(nets[i-1])), winners[i-1] its
    arg-max bits, hrs[i] the hidden activations; ``want_x_full`` also materialises every block's [M, 64] input (tests)."""
    m, nb, dev = pts.shape[0], len(blocks), pts.device
    if not want_x_full and _TRUNK_FUSED and 2 <= nb <= 8 and m > 0 and m >= _TRUNK_FUSED_MIN_ROWS:
        return _trunk_forward_one_launch(tile, pts, w_pos, b_pos, blocks, w_c, b_c)
    nets, pooled, hrs, winners, x_fulls = [], [None], [], [], []
    net_prev, c_out = None, None
    for i, (w0, b0, w1, b1, ws) in enumerate(blocks):
        first, last = i == 0, i == nb - 1
        x_full = _empty(m, 64, pts) if want_x_full else None
        pool = None if first else _empty(m, 32, pts)
        hr, out = _empty(m, 32, pts), _empty(m, 32, pts)
        win = None if first else torch.empty(m, 8, dtype=torch.uint8, device=dev)
        if last:
            c_out = _empty(m, 32, pts)
        wts = [t.contiguous() for t in (w0, b0, w1, b1, ws)]
        nbytes = m * ((4 * pts.shape[1] if first else 4 * 32 + 4) + 2 * 128 + (0 if first else 8 + 128) + (128 if last else 0)
                      + (256 if x_full is not None else 0))
        flops = 2 * m * (2 * 64 * 32 + 32 * 32 + (32 * 32 if last else 0) + (3 * 64 if first else 0))
        _lib.call("t2h_trunk_block_fwd",
                  _lib.ptr(pts) if first else None, pts.shape[1], _lib.ptr(w_pos.contiguous()) if first else None,
                  _lib.ptr(b_pos) if first else None,
                  None if first else _lib.ptr(net_prev), 32, None if first else _lib.ptr(tile.cell),
                  None if first else _lib.ptr(tile.off0),
                  *[_lib.ptr(t) for t in wts], _lib.ptr(w_c.contiguous()) if last else None, _lib.ptr(b_c) if last else None,
                  m, None if x_full is None else _lib.ptr(x_full), None if pool is None else _lib.ptr(pool), _lib.ptr(hr),
                  _lib.ptr(out), 32, None if win is None else _lib.ptr(win), None if c_out is None else _lib.ptr(c_out),
                  _lib.stream(), nbytes=nbytes, flops=flops,
                  tag="t2h_trunk_block_fwd[%s]" % ("first" if first else ("last" if last else "mid")))
        nets.append(out)
        hrs.append(hr)
        x_fulls.append(x_full)
        if not first:
            pooled.append(pool)
            winners.append(win)
        net_prev = out
    if want_x_full:
        return c_out, nets, pooled, hrs, winners, x_fulls
    return c_out, nets, pooled, hrs, winners


def _trunk_backward_fused(tile, pts, params, nets, pooled, hrs, winners, g_out):
    """Backward of ``_trunk_forward_fused``: one t2h_trunk_block_bwd + one slab reduction per block, last block first.
    Block i's kernel folds the backward of the pooling that consumed its output (gather -> scatter-add over the cell,
    scatter_max -> its arg-max row, pointnet.py:95-98) into its loader, the last block's the backward of
    ``fc_c(relu(.))``, the first block's the fc_pos weight gradient.  Returns the gradients in ``params`` order
    (``None`` where they were accumulated straight into ``param.grad``)."""
    lib = _lib.load()
    nb = (len(params) - 4) // 5
    m = pts.shape[0]
    w_pos, b_pos, w_c, b_c = params[0], params[1], params[-2], params[-1]
    direct = _DIRECT_ACCUM and all(p.grad is not None and p.grad.is_contiguous() for p in params)
    grads = [None] * len(params)
    ws_bytes = _lib.ws_bytes("t2h_trunk_block_bwd_workspace_bytes", m)
    ws = _lib.workspace(ws_bytes, pts.device)            # reused by every block: launches are stream ordered
    dx_next = None
    for i in range(nb - 1, -1, -1):
        first, last = i == 0, i == nb - 1
        w0, b0, w1, b1, wsc = params[2 + 5 * i: 7 + 5 * i]
        dx = None if first else _empty(m, 64, pts)
        nbytes = m * ((128 if last else 256 + 8 + 4) + 128 + (4 * pts.shape[1] if first else 256) + (128 if last else 0)
                      + (0 if first else 256))
        flops = 2 * m * (2 * 32 * 32 + 4 * 64 * 32 + (2 * 32 * 32 if last else 0))
        _lib.call("t2h_trunk_block_bwd",
                  None if last else _lib.ptr(dx_next), 64, None if last else _lib.ptr(dx_next) + 128, 64,
                  None if last else _lib.ptr(winners[i]), _lib.ptr(tile.cell), _lib.ptr(tile.off0),
                  _lib.ptr(g_out) if last else None, _lib.ptr(w_c.contiguous()) if last else None,
                  _lib.ptr(nets[-1]) if last else None, _lib.ptr(hrs[i]),
                  None if first else _lib.ptr(nets[i - 1]), 32, None if first else _lib.ptr(pooled[i]), 32,
                  _lib.ptr(pts) if first else None, pts.shape[1], _lib.ptr(w_pos.contiguous()) if first else None,
                  _lib.ptr(b_pos) if first else None, _lib.ptr(w0.contiguous()), _lib.ptr(w1.contiguous()),
                  _lib.ptr(wsc.contiguous()), m, None if dx is None else _lib.ptr(dx), _lib.ptr(ws), ws_bytes, _lib.stream(),
                  nbytes=nbytes, flops=flops,
                  tag="t2h_trunk_block_bwd[%s]" % ("first" if first else ("last" if last else "mid")))
        extra = (w_pos, b_pos) if first else ((w_c, b_c) if last else None)
        if direct:
            dst = [w0.grad, b0.grad, w1.grad, b1.grad, wsc.grad] + ([extra[0].grad, extra[1].grad] if extra else [None, None])
        else:
            dst = [torch.empty_like(t, memory_format=torch.contiguous_format) for t in (w0, b0, w1, b1, wsc)]
            dst += [torch.empty_like(t, memory_format=torch.contiguous_format) for t in extra] if extra else [None, None]
            grads[2 + 5 * i: 7 + 5 * i] = dst[:5]
            if first:
                grads[0], grads[1] = dst[5], dst[6]
            if last:
                grads[-2], grads[-1] = dst[5], dst[6]
        _lib.call("t2h_trunk_block_reduce", _lib.ptr(ws), m, int(first), int(last), _lib.ptr(dst[0]), _lib.ptr(dst[1]),
                  _lib.ptr(dst[2]), _lib.ptr(dst[3]), _lib.ptr(dst[4]), None if dst[5] is None else _lib.ptr(dst[5]),
                  None if dst[6] is None else _lib.ptr(dst[6]), 1 if direct else 0, _lib.stream(),
                  nbytes=ws_bytes, tag="t2h_trunk_block_reduce")
        dx_next = dx
    return grads


class _PointTrunk(torch.autograd.Function):
    """pointnet.py:72-82 on sorted rows: fc_pos -> block0 -> 4 x {pool_max, block} -> relu -> fc_c.

    Buffers: ``cat[i]`` is the [M, 2h] input of block i (cat[0] = fc_pos output); block i writes its [M, h]
    output into cat[i+1][:, :h] and the pool writes cat[i+1][:, h:].  Weights arrive flattened:
    (w_pos, b_pos, [w0, b0, w1, b1, ws] * n_blocks, w_c, b_c)."""

    @staticmethod
    def forward(ctx, tile, pts, pool, *params):
        n_blocks = (len(params) - 4) // 5
        ctx.pool = pool
        w_pos, b_pos = params[0], params[1]
        blocks = [params[2 + 5 * i: 7 + 5 * i] for i in range(n_blocks)]
        w_c, b_c = params[-2], params[-1]
        m = pts.shape[0]
        h = blocks[0][2].shape[0]
        ctx.tile, ctx.n_blocks, ctx.h = tile, n_blocks, h
        if pool == "max" and _fused_trunk_applicable(pts, params, n_blocks):
            out, nets, pooled, hrs, winners = _trunk_forward_fused(tile, pts, w_pos, b_pos, blocks, w_c, b_c)
            ctx.fused = True                               # (block 0's input is recomputed from the points in the backward)
            ctx.save_for_backward(pts, *params, *nets, *pooled[1:], *hrs, *winners)
            return out
        ctx.fused = False
        if pts.shape[1] != w_pos.shape[1]:                 # (a ragged TileIndex keeps the tile index in one more column)
            pts = pts[:, :w_pos.shape[1]].contiguous()
        cats, hrs, winners = [], [], []
        cat0 = _empty(m, w_pos.shape[0], pts)
        linear_fwd_(pts, w_pos, b_pos, cat0)                                   # pointnet.py:72
        cats.append(cat0)
        for i, (w0, b0, w1, b1, ws) in enumerate(blocks):
            last = i == n_blocks - 1
            nxt = _empty(m, h, pts) if last else _empty(m, 2 * h, pts)
            hrs.append(_resblock_fwd(cats[i], w0, b0, w1, b1, ws, nxt[:, :h]))     # pointnet.py:73,79
            if not last and pool == "mean":
                _pool_mean_(tile, nxt[:, :h], nxt[:, h:], accumulate=False)        # pointnet.py:77-78, scatter_mean
            elif not last:
                win = torch.empty(m, _lib.load().t2h_pool_winner_stride(h), dtype=torch.uint8, device=pts.device)
                _pool_fwd_(tile, nxt[:, :h], nxt[:, h:], win)                      # pointnet.py:77-78
                winners.append(win)
            cats.append(nxt)
        out = _empty(m, w_c.shape[0], pts)
        linear_fwd_(cats[-1], w_c, b_c, out, relu_in=True)                          # pointnet.py:81-82
        ctx.save_for_backward(pts, *params, *cats, *hrs, *winners)
        return out

    @staticmethod
    def backward(ctx, g_out):
        nb, h, tile = ctx.n_blocks, ctx.h, ctx.tile
        saved = ctx.saved_tensors
        pts = saved[0]
        n_params = 4 + 5 * nb
        params = saved[1:1 + n_params]
        cats = saved[1 + n_params: 2 + n_params + nb]          # nb + 1 buffers
        hrs = saved[2 + n_params + nb: 2 + n_params + 2 * nb]
        winners = saved[2 + n_params + 2 * nb:]
        w_pos, b_pos = params[0], params[1]
        blocks = [params[2 + 5 * i: 7 + 5 * i] for i in range(nb)]
        w_c, b_c = params[-2], params[-1]

        g_out = g_out.contiguous()
        if ctx.fused:
            rest = saved[1 + n_params:]
            nets, pooled, hrs, winners = rest[:nb], [None, *rest[nb:2 * nb - 1]], rest[2 * nb - 1:3 * nb - 1], rest[3 * nb - 1:]
            return (None, None, None, *_trunk_backward_fused(tile, pts, params, nets, pooled, hrs, winners, g_out.contiguous()))
        dw_c, db_c = _wgrad(g_out, cats[-1], w_c, b_c, relu_in=True)
        g = linear_dgrad_(g_out, w_c, torch.empty_like(cats[-1]), mask=cats[-1])     # grad of the last block output
        grads = [None] * n_params
        grads[-2], grads[-1] = dw_c, db_c
        for i in range(nb - 1, -1, -1):
            w0, b0, w1, b1, ws = blocks[i]
            dx, dw0, db0, dw1, db1, dws = _resblock_bwd(g, cats[i], hrs[i], w0, b0, w1, b1, ws)
            grads[2 + 5 * i: 7 + 5 * i] = [dw0, db0, dw1, db1, dws]
            if i > 0:
                # dx = [d net | d pooled]: fold the pool's gradient into the left half, which is then d(net_i)
                if ctx.pool == "mean":
                    _pool_mean_(tile, dx[:, h:], dx[:, :h], accumulate=True)
                else:
                    _pool_bwd_(tile, dx[:, h:], winners[i - 1], dx[:, :h], accumulate=True)
                g = dx[:, :h]
            else:
                g = dx
        grads[0], grads[1] = _wgrad(g, pts, w_pos, b_pos)
        return (None, None, None, *grads)


def point_trunk(tile, pts, fc_pos, blocks, fc_c, pool="max") -> torch.Tensor:
    """``pool``: 'max' (scatter_max, every shipped config) or 'mean' (scatter_mean; block-by-block kernels only -- the
    fused trunk block of csrc/trunk.hip carries the max pooling in its loader)."""
    params = [fc_pos.weight, fc_pos.bias]
    for b in blocks:
        if b.shortcut is None:
            raise NotImplementedError("trunk blocks always change width (2h -> h) and so carry a shortcut")
        params += [b.fc_0.weight, b.fc_0.bias, b.fc_1.weight, b.fc_1.bias, b.shortcut.weight]
    params += [fc_c.weight, fc_c.bias]
    return _PointTrunk.apply(tile, pts, pool, *params)
