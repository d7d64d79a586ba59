"""Grid side of the hot path on channels_last (NHWC) tensors: the convolutions of the ALTO U-Net, the image
U-Net and the pixel decoder on the implicit-GEMM HIP kernels, and the elementwise work around them fused away.

    conv_bias_act(x, conv, relu)   3x3 stride-1 convs: implicit-GEMM MFMA kernels of csrc/conv.hip (forward with fused
                                   bias/ReLU, data gradient, deterministic weight + bias gradient); any other conv:
                                   MIOpen (bias-free call) + fused bias/ReLU passes around it
    conv1x1(x, conv)               1x1 convs as the per-point GEMM kernels on the [pixels, C] rows of the NHWC plane
    head1x1(xs, conv4)             torch.cat(xs, 1) -> 1x1 conv to one channel without the concat (pixel.py:31)
    upsample_bilinear_cl(x, size)  F.interpolate(bilinear, align_corners=True) on channels_last planes (pixel.py:107)

Every function requires channels_last device tensors; callers (alto.py / pixel.py here) use them only when the model
runs in channels_last mode and otherwise keep the plain torch composition.
"""
import ctypes
import contextlib
import os

import torch
import torch.nn.functional as F

from . import _lib, mlp

# A/B switch: T2H_HIP_CONV=0 keeps every convolution on MIOpen (an explicit request, so it also allows the fallbacks)
USE_HIP_CONV = os.environ.get("T2H_HIP_CONV", "1") != "0"
if not USE_HIP_CONV:
    _lib.allow_library_fallback(True).set()


def fallback_count() -> int:
    """Vendor-library / ATen fallbacks taken so far in this process (0 on the reference's default configurations)."""
    return sum(_lib.fallback_counts().values())


def _alias(x: torch.Tensor) -> torch.Tensor:
    """Same memory, exactly the same strides (see ops._alias)."""
    return x.as_strided(x.size(), x.stride(), x.storage_offset())


def is_cl(x: torch.Tensor) -> bool:
    """[B,C,H,W] tensor whose memory is dense NHWC (a C == 1 or H*W == 1 tensor counts when its NHWC view is contiguous)."""
    if x.dim() != 4 or not x.is_cuda or x.dtype != torch.float32:
        return False
    # (fast path: one C call instead of building the permuted view -- ~200 of these per tile-step; torch's own channels_last test
    # is stricter about size-1 dimensions, the view test below decides those)
    return x.is_contiguous(memory_format=torch.channels_last) or x.permute(0, 2, 3, 1).is_contiguous()


def _as_cl(x: torch.Tensor) -> torch.Tensor:
    return x if is_cl(x) else x.contiguous(memory_format=torch.channels_last)


def _cl_ld(x: torch.Tensor):
    """(tensor, pixel stride in floats): ``x`` itself when it is NHWC with a pixel stride >= C -- dense, or a channel slice of a
    wider NHWC tensor, which is how ``torch.cat``'s backward hands each input its gradient (alto.py:227) -- read in place by the
    kernels that take a stride; a dense NHWC copy otherwise."""
    if x.dim() == 4 and x.is_cuda and x.dtype == torch.float32:
        b, c, h, w = x.shape
        sb, sc, sh, sw = x.stride()
        if (sc == 1 or c == 1) and sw >= c and sw % 4 == 0 and sh == w * sw and (sb == h * sh or b == 1) and x.data_ptr() % 16 == 0:
            return x, sw
    x = _as_cl(x)
    return x, x.shape[1]


def _empty_cl(b, c, h, w, device) -> torch.Tensor:
    return torch.empty((b, c, h, w), dtype=torch.float32, device=device, memory_format=torch.channels_last)


# ------------------------------------------------------------------------------------------------ conv + bias + relu
class _ConvBiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, relu: bool):
        y = F.conv2d(x, weight, None, stride, padding)
        y = _as_cl(y)
        b, c, h, w = y.shape
        _lib.call("t2h_bias_relu_fwd", _lib.ptr(y), _lib.ptr(bias), b * h * w, c, 1 if relu else 0, _lib.stream(),
                  nbytes=8 * y.numel())
        ctx.save_for_backward(x, weight, y)
        ctx.conf = (stride, padding, relu)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        stride, padding, relu = ctx.conf
        g = _as_cl(g)
        b, c, h, w = g.shape
        gm = torch.empty_like(g, memory_format=torch.channels_last) if relu else g     # masked gradient, out of place
        dbias = torch.empty(c, dtype=torch.float32, device=g.device)
        ws_bytes = _lib.ws_bytes("t2h_bias_relu_bwd_workspace_bytes", b * h * w, c)
        ws = _lib.workspace(ws_bytes, g.device)
        _lib.call("t2h_bias_relu_bwd", _lib.ptr(g), _lib.ptr(y), _lib.ptr(gm) if relu else None, b * h * w, c,
                  1 if relu else 0, 0, _lib.ptr(dbias), _lib.ptr(ws), ws_bytes, _lib.stream(),
                  nbytes=(12 if relu else 4) * g.numel())
        dx, dw, _ = torch.ops.aten.convolution_backward(
            gm, x, weight, None, list(stride), list(padding), [1, 1], False, [0, 0], 1,
            [ctx.needs_input_grad[0], True, False])
        return dx, dw, dbias, None, None, None


# ------------------------------------------------------------------------------------------------ conv3x3 (HIP)
def _pow2(v: int) -> bool:
    return v > 0 and (v & (v - 1)) == 0


def _w_cl(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3] weight whose memory is [Cout][3][3][Cin] (what the kernels read)."""
    if w.is_contiguous(memory_format=torch.channels_last) or w.permute(0, 2, 3, 1).is_contiguous():
        return w
    return w.contiguous(memory_format=torch.channels_last)


def conv3x3_supported(x: torch.Tensor, conv: torch.nn.Conv2d) -> bool:
    return (USE_HIP_CONV and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"
            and conv.in_channels % 16 == 0 and conv.out_channels % 16 == 0 and x.dim() == 4 and x.is_cuda
            and x.dtype == torch.float32 and _pow2(x.shape[2]) and _pow2(x.shape[3]))


# ---- the same convolutions on the 16-bit matrix cores with split operands (csrc/conv_bx3.hip) ---------------------------------
# T2H_CONV_PRECISION selects the arithmetic of the 3x3 / transposed convolutions and of the wide grid-side products on planes of at
# least T2H_BX3_MIN_PIXELS pixels (default 32 x 32; smaller ones stay on conv.hip):
#   'f16x2' (default)  v_mfma_f32_32x32x16_f16 with a TWO-way fp16 split of both operands and one power-of-two scale per staged
#                      block -- three MFMAs per product, fp32-grade unless an element lies more than 2^18 below the largest element
#                      of its own staged block (csrc/conv_bx3.hip, "the fp16 two-way split"; include/t2h.h T2H_F16X2)
#   'bf16x3'           the exact three-way bf16 split: six MFMAs per product (the r04b form)
#   'fp32'             conv.hip's fp32 MFMA kernels (A/B; same tolerance: tests/test_hip_conv.py measures all three against float64)
#   'bf16'             (set by TomoSAR2Height.set_mlp_precision('bf16'), BASELINE configs[2]) the leading bf16 part of both operands
#                      only: one MFMA per product, tolerance of that MODE 2e-2 of the height scale (tests/test_full_size_vs_oracle.py)
CONV_PRECISION = os.environ.get("T2H_CONV_PRECISION", "f16x2")
BX3_MIN_PIXELS = int(os.environ.get("T2H_BX3_MIN_PIXELS", str(32 * 32)))
_CONV_PRECISIONS = ("fp32", "bf16x3", "f16x2", "bf16")
if CONV_PRECISION not in _CONV_PRECISIONS:
    raise ValueError(f"T2H_CONV_PRECISION={CONV_PRECISION!r}: expected one of {_CONV_PRECISIONS}")
_DEFAULT_CONV_PRECISION = CONV_PRECISION


def set_conv_precision(name: str = None):
    """'f16x2', 'bf16x3', 'fp32' or 'bf16' (see above); None restores the process default (T2H_CONV_PRECISION)."""
    global CONV_PRECISION
    name = _DEFAULT_CONV_PRECISION if name is None else name
    if name not in _CONV_PRECISIONS:
        raise ValueError(f"conv precision must be one of {_CONV_PRECISIONS}")
    CONV_PRECISION = name


def bx3_applicable(b: int, h: int, wd: int, cin: int, cout: int) -> bool:
    return (CONV_PRECISION in ("bf16x3", "f16x2", "bf16") and h * wd >= BX3_MIN_PIXELS
            and bool(_lib.ws_bytes("t2h_conv3x3_bx3_supported", b, h, wd, cin, cout)))      # (memoised like the workspace queries)


def _bx3_flag() -> int:
    return _lib.BF16 if CONV_PRECISION == "bf16" else (_lib.F16X2 if CONV_PRECISION == "f16x2" else 0)


def _h2() -> bool:
    """The prepared weights are the fp16 two-plane buffers (T2H_F16X2)."""
    return CONV_PRECISION == "f16x2"


class SplitWeightCache:
    """The three bf16 planes of a convolution weight in MFMA operand order (``t2h_conv3x3_bx3_prepare``): a function of the
    weight alone, so computed when the weight changes (its version counter / storage moves: every optimizer, FlatAdamW included,
    bumps the counter) instead of per tile.  The buffers are updated IN PLACE, so a captured hipGraph keeps reading current
    values after ``refresh()`` (called by ``Trainer.optimizer_boundary``; a replayed graph runs no Python per tile)."""

    def __init__(self):
        self.entries = {}          # (id(weight), transposed) -> [weakref, version, data_ptr, buffer, max slot, _lib.Ready]

    def get(self, w: torch.Tensor, transposed: bool) -> torch.Tensor:
        import weakref
        key = (id(w), ("h2t" if transposed else "h2f") if _h2() else bool(transposed))
        e = self.entries.get(key)
        cout, cin = w.shape[0], w.shape[1]
        if e is None or e[0]() is not w:
            lib = _lib.load()
            nbytes = int((lib.t2h_conv3x3_f16x2_weights_bytes if _h2() else lib.t2h_conv3x3_bx3_weights_bytes)(cin, cout))
            buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
            e = self.entries[key] = [weakref.ref(w, lambda _r, k=key: self.entries.pop(k, None)), None, None, buf, 3, _lib.Ready()]
        if e[1] != w._version or e[2] != w.data_ptr():
            self._prepare(w, key[1], e[3])
            e[1], e[2], e[4] = w._version, w.data_ptr(), 3
            e[5].mark()
        else:
            e[5].wait()                                       # (filled on another stream: that fill first)
        return e[3]

    @staticmethod
    def _prepare(w, kind, buf):
        cout, cin = w.shape[0], w.shape[1]
        h2 = kind in ("h2t", "h2f")
        transposed = kind in (True, "h2t")
        _lib.call("t2h_conv3x3_f16x2_prepare" if h2 else "t2h_conv3x3_bx3_prepare", _lib.ptr(w), cin, cout, 1 if transposed else 0,
                  _lib.ptr(buf), _lib.stream(), nbytes=(12 if h2 else 10) * w.numel())

    def get_gemm(self, w: torch.Tensor, w_is_kn: bool) -> torch.Tensor:
        """The same for a plain weight matrix of the 1-tap (GEMM) form: ``w`` [N, K] (nn.Linear) or, ``w_is_kn``, [K, N]."""
        import weakref
        key = (id(w), ("kn" if w_is_kn else "nk") + ("_h2" if _h2() else ""))
        e = self.entries.get(key)
        k, n = (w.shape[0], w.shape[1]) if w_is_kn else (w.shape[1], w.shape[0])
        if e is None or e[0]() is not w:
            lib = _lib.load()
            buf = torch.empty(int((lib.t2h_gemm_f16x2_weights_bytes if _h2() else lib.t2h_gemm_bx3_weights_bytes)(k, n)),
                              dtype=torch.uint8, device=w.device)
            e = self.entries[key] = [weakref.ref(w, lambda _r, kk=key: self.entries.pop(kk, None)), None, None, buf, 3, _lib.Ready()]
        if e[1] != w._version or e[2] != w.data_ptr():
            self._prepare_gemm(w, key[1], e[3])
            e[1], e[2], e[4] = w._version, w.data_ptr(), 3
            e[5].mark()
        else:
            e[5].wait()
        return e[3]

    @staticmethod
    def _prepare_gemm(w, kind, buf):
        w_is_kn, h2 = kind.startswith("kn"), kind.endswith("_h2")
        k, n = (w.shape[0], w.shape[1]) if w_is_kn else (w.shape[1], w.shape[0])
        _lib.call("t2h_gemm_f16x2_prepare" if h2 else "t2h_gemm_bx3_prepare", _lib.ptr(w), w.stride(0), k, n, 1 if w_is_kn else 0,
                  _lib.ptr(buf), _lib.stream(), nbytes=(12 if h2 else 10) * w.numel())

    def get_up(self, w: torch.Tensor, w_is_kn: bool) -> torch.Tensor:
        """The same for a ConvTranspose2d(2, stride 2) weight [Cin, Cout, 2, 2] whose memory is [Cin][2][2][Cout]: the matrix
        [Cin, (tap, co)] as the forward's [K, N] operand (``w_is_kn``) or as the data gradient's [N, K] operand."""
        import weakref
        key = (id(w), ("up_kn" if w_is_kn else "up_nk") + ("_h2" if _h2() else ""))
        e = self.entries.get(key)
        cin, n4 = w.shape[0], 4 * w.shape[1]
        if e is None or e[0]() is not w:
            k, n = (cin, n4) if w_is_kn else (n4, cin)
            lib = _lib.load()
            buf = torch.empty(int((lib.t2h_gemm_f16x2_weights_bytes if _h2() else lib.t2h_gemm_bx3_weights_bytes)(k, n)),
                              dtype=torch.uint8, device=w.device)
            e = self.entries[key] = [weakref.ref(w, lambda _r, kk=key: self.entries.pop(kk, None)), None, None, buf, 3, _lib.Ready()]
        if e[1] != w._version or e[2] != w.data_ptr():
            self._prepare_up(w, key[1], e[3])
            e[1], e[2], e[4] = w._version, w.data_ptr(), 3
            e[5].mark()
        else:
            e[5].wait()
        return e[3]

    @staticmethod
    def _prepare_up(w, kind, buf):
        w_is_kn, h2 = kind.startswith("up_kn"), kind.endswith("_h2")
        cin, n4 = w.shape[0], 4 * w.shape[1]
        k, n = (cin, n4) if w_is_kn else (n4, cin)
        _lib.call("t2h_gemm_f16x2_prepare" if h2 else "t2h_gemm_bx3_prepare", _lib.ptr(w), n4, k, n, 1 if w_is_kn else 0,
                  _lib.ptr(buf), _lib.stream(), nbytes=(12 if h2 else 10) * w.numel())

    def refresh(self, stale_only: bool = False):
        """Re-split every live weight (``stale_only``: those whose version counter or storage moved) into its existing buffer, all
        of them in a few launches (``t2h_split_weights_batch``: 24 buffers per launch instead of 1-3 launches per buffer --
        ``Trainer.optimizer_boundary`` calls this right after the optimizer step, so the tiles that follow find every buffer
        current and launch nothing)."""
        todo = []
        for (_, kind), e in list(self.entries.items()):
            w = e[0]()
            if w is None or not w.is_cuda or (stale_only and e[1] == w._version and e[2] == w.data_ptr()):
                continue
            d = _lib.PrepDesc()
            d.w, d.wf = w.data_ptr(), e[3].data_ptr()
            if isinstance(kind, str) and kind.startswith("up_"):           # [Cin, Cout, 2, 2] as the matrix [Cin][4 Cout]
                kn, h2 = kind.startswith("up_kn"), kind.endswith("_h2")
                cin, n4 = w.shape[0], 4 * w.shape[1]
                d.kind, d.a, d.b, d.ldw = (2, cin, n4, n4) if kn else (3, n4, cin, n4)
            elif isinstance(kind, str) and kind[:2] in ("kn", "nk"):
                kn, h2 = kind.startswith("kn"), kind.endswith("_h2")
                if w.dim() != 2 or w.stride(1) != 1:
                    continue
                d.kind, d.a, d.b, d.ldw = (2, w.shape[0], w.shape[1], w.stride(0)) if kn else (3, w.shape[1], w.shape[0], w.stride(0))
            else:
                h2 = kind in ("h2t", "h2f")
                if not w.permute(0, 2, 3, 1).is_contiguous():
                    continue
                d.kind, d.a, d.b, d.ldw = (1 if kind in (True, "h2t") else 0), w.shape[1], w.shape[0], 0
            d.h2, d.maxslot, d.trailer_word = int(h2), e[4], (e[3].numel() - 256) // 4 if h2 else 0
            todo.append((d, e, w))
        if not todo:
            return
        arr = (_lib.PrepDesc * len(todo))(*[t[0] for t in todo])
        _lib.call("t2h_split_weights_batch", ctypes.addressof(arr), len(todo), _lib.stream())
        done = _lib.Ready()
        done.mark()                                           # one event behind the batch, shared by its entries
        for d, e, w in todo:
            e[1], e[2], e[4] = w._version, w.data_ptr(), (e[4] ^ 3) if d.h2 else e[4]
            e[5] = done


split_weights = SplitWeightCache()
BX3_WGRAD = os.environ.get("T2H_BX3_WGRAD", "1") != "0"        # A/B: 0 = weight gradients stay on conv.hip


def conv3x3_fwd_(x, w, bias, y, relu=False, accumulate=False):
    """y [B,Cout,H,W] (channels_last) = [y +] act(conv3x3(x, w) + bias); raw kernel call on NHWC-dense tensors."""
    b, cin, h, wd = x.shape
    cout = w.shape[0]
    lib = _lib.load()
    if bx3_applicable(b, h, wd, cin, cout):
        flags = (_lib.RELU_OUT if relu else 0) | (_lib.ACCUM if accumulate else 0) | _bx3_flag()
        nws = _lib.ws_bytes("t2h_conv3x3_bx3_fwd_workspace_bytes", b, h, wd, cin, cout)
        ws = _lib.workspace(nws, x.device)
        _lib.call("t2h_conv3x3_bx3_fwd", _lib.ptr(x), _lib.ptr(split_weights.get(w, False)),
                  _lib.ptr(bias) if bias is not None else None, _lib.ptr(y), b, h, wd, cin, cout, flags, _lib.ptr(ws), nws, _lib.stream(),
                  nbytes=4 * (x.numel() + y.numel() + w.numel()), flops=2 * 9 * cin * cout * b * h * wd,
                  tag=_lib.timing() and f"t2h_conv3x3_fwd[{cin}->{cout},{h}x{wd}]")
        return y
    nws = _lib.ws_bytes("t2h_conv3x3_fwd_workspace_bytes", b, h, wd, cin, cout)
    ws = _lib.workspace(nws, x.device)
    flags = (_lib.RELU_OUT if relu else 0) | (_lib.ACCUM if accumulate else 0)
    _lib.call("t2h_conv3x3_fwd", _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias) if bias is not None else None, _lib.ptr(y),
              b, h, wd, cin, cout, flags, _lib.ptr(ws), nws, _lib.stream(),
              nbytes=4 * (x.numel() + y.numel() + w.numel()), flops=2 * 9 * cin * cout * b * h * wd,
              tag=_lib.timing() and f"t2h_conv3x3_fwd[{cin}->{cout},{h}x{wd}]")
    return y


# r06: the 1 x 1 head's rank-1 share of a decoder activation's gradient formed in the data gradient's epilogue (T2H_HEAD_RANK1=0: the
# head writes it, the data gradient accumulates onto it -- bit-identical)
HEAD_RANK1 = os.environ.get("T2H_HEAD_RANK1", "1") != "0"


def dgrad_rank1_ok(gy, w) -> bool:
    b, cout, h, wd = gy.shape
    cin = w.shape[1]
    return (HEAD_RANK1 and bx3_applicable(b, h, wd, cin, cout)
            and bool(_lib.load().t2h_conv3x3_bx3_dgrad_rank1_supported(b, h, wd, cin, cout)))


def conv3x3_dgrad_rank1_(gy, w, dx, mask, g, w1):
    """dx = mask(conv3x3 data gradient of gy) + mask(g[pixel] * w1[channel]), written (``dgrad_rank1_ok`` shapes only)."""
    b, cout, h, wd = gy.shape
    cin = w.shape[1]
    _lib.call("t2h_conv3x3_bx3_dgrad_rank1", _lib.ptr(gy), _lib.ptr(split_weights.get(w, True)), _lib.ptr(dx),
              _lib.ptr(mask) if mask is not None else None, _lib.ptr(g), _lib.ptr(w1), b, h, wd, cin, cout, _bx3_flag(), _lib.stream(),
              nbytes=4 * (gy.numel() + dx.numel() * (2 if mask is not None else 1) + w.numel() + g.numel()),
              flops=2 * 9 * cin * cout * b * h * wd, tag=_lib.timing() and f"t2h_conv3x3_dgrad[{cout}->{cin},{h}x{wd}]")
    return dx


def conv3x3_dgrad_(gy, w, dx, mask=None, accumulate=False):
    b, cout, h, wd = gy.shape
    cin = w.shape[1]
    lib = _lib.load()
    if bx3_applicable(b, h, wd, cin, cout):
        nws = _lib.ws_bytes("t2h_conv3x3_bx3_dgrad_workspace_bytes", b, h, wd, cin, cout)
        ws = _lib.workspace(nws, gy.device)
        _lib.call("t2h_conv3x3_bx3_dgrad", _lib.ptr(gy), _lib.ptr(split_weights.get(w, True)), _lib.ptr(dx),
                  _lib.ptr(mask) if mask is not None else None, b, h, wd, cin, cout, (_lib.ACCUM if accumulate else 0) | _bx3_flag(),
                  _lib.ptr(ws), nws, _lib.stream(),
                  nbytes=4 * (gy.numel() + dx.numel() * (2 if mask is not None else 1) + w.numel()),
                  flops=2 * 9 * cin * cout * b * h * wd, tag=_lib.timing() and f"t2h_conv3x3_dgrad[{cout}->{cin},{h}x{wd}]")
        return dx
    nws = _lib.ws_bytes("t2h_conv3x3_dgrad_workspace_bytes", b, h, wd, cin, cout)
    ws = _lib.workspace(nws, gy.device)
    _lib.call("t2h_conv3x3_dgrad", _lib.ptr(gy), _lib.ptr(w), _lib.ptr(dx), _lib.ptr(mask) if mask is not None else None,
              b, h, wd, cin, cout, _lib.ACCUM if accumulate else 0, _lib.ptr(ws), nws, _lib.stream(),
              nbytes=4 * (gy.numel() + dx.numel() * (2 if mask is not None else 1) + w.numel()),
              flops=2 * 9 * cin * cout * b * h * wd, tag=_lib.timing() and f"t2h_conv3x3_dgrad[{cout}->{cin},{h}x{wd}]")
    return dx


def conv3x3_wgrad_(gy, x, dw, db, accumulate=False, defer=False):
    """``defer``: dw / db are final gradient buffers nobody reads before the backward pass ends -- the slab reduction may join the
    pass's batched one (``_lib.reduce_capture``)."""
    b, cout, h, wd = gy.shape
    cin = x.shape[1]
    lib = _lib.load()
    entry = "t2h_conv3x3_bx3_wgrad" if (BX3_WGRAD and bx3_applicable(b, h, wd, cin, cout)) else "t2h_conv3x3_wgrad"
    if entry == "t2h_conv3x3_bx3_wgrad":
        nws = _lib.ws_bytes("t2h_conv3x3_bx3_wgrad_workspace_bytes", b, h, wd, cin, cout)
        ws = _lib.workspace(nws, gy.device)
        _lib.call(entry, _lib.ptr(gy), _lib.ptr(x), _lib.ptr(dw), _lib.ptr(db) if db is not None else None,
                  b, h, wd, cin, cout, (_lib.ACCUM if accumulate else 0) | _bx3_flag() | (_lib.defer_reduce(ws, dw) if defer else 0),
                  _lib.ptr(ws), nws, _lib.stream(),
                  nbytes=4 * (gy.numel() + x.numel() + dw.numel()), flops=2 * 9 * cin * cout * b * h * wd,
                  tag=_lib.timing() and f"t2h_conv3x3_wgrad[{cin}->{cout},{h}x{wd}]")
        return
    nws = _lib.ws_bytes("t2h_conv3x3_wgrad_workspace_bytes", b, h, wd, cin, cout)
    ws = _lib.workspace(nws, gy.device)
    _lib.call("t2h_conv3x3_wgrad", _lib.ptr(gy), _lib.ptr(x), _lib.ptr(dw), _lib.ptr(db) if db is not None else None,
              b, h, wd, cin, cout, (_lib.ACCUM if accumulate else 0) | (_lib.defer_reduce(ws, dw) if defer else 0), _lib.ptr(ws), nws,
              _lib.stream(),
              nbytes=4 * (gy.numel() + x.numel() + dw.numel()), flops=2 * 9 * cin * cout * b * h * wd,
              tag=_lib.timing() and f"t2h_conv3x3_wgrad[{cin}->{cout},{h}x{wd}]")


class _Conv3x3(torch.autograd.Function):
    """relu?(conv3x3(x) + bias) on csrc/conv.hip.  ``mask_input``: x is a ReLU output consumed by this conv only, so the
    data gradient is returned already multiplied by (x > 0); the producer is then built with ``grad_premasked`` and
    skips its own ReLU-backward pass (the pair is set up by ``conv3x3_chain``)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu: bool, mask_input: bool, grad_premasked: bool):
        x = _as_cl(x)
        w = _w_cl(weight)
        b, _, h, wd = x.shape
        y = _empty_cl(b, w.shape[0], h, wd, x.device)
        conv3x3_fwd_(x, w, bias, y, relu=relu)
        need_y = relu and not grad_premasked
        ctx.save_for_backward(x, weight, bias, y if need_y else None)
        ctx.conf = (relu, mask_input, grad_premasked)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, bias, y = ctx.saved_tensors
        relu, mask_input, grad_premasked = ctx.conf
        g = _as_cl(g)
        w = _w_cl(weight)
        if relu and not grad_premasked:
            gm = torch.empty_like(g, memory_format=torch.channels_last)
            _lib.call("t2h_relu_mask", _lib.ptr(g), _lib.ptr(y), _lib.ptr(gm), g.numel(), _lib.stream(),
                      nbytes=12 * g.numel())
        else:
            gm = g
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x, memory_format=torch.channels_last)
            conv3x3_dgrad_(gm, w, dx, mask=x if mask_input else None)
        dw, db = _conv3x3_param_grads(gm, x, weight, bias)
        return dx, dw, db, None, None, None


# planes whose weight gradients go to the trainer's side stream (r04d: all of them -- 8.29 -> 8.14 ms per step against 128 x 128)
OVERLAP_MAX_PIXELS = int(os.environ.get("T2H_OVERLAP_CONV_MAX_PIXELS", str(512 * 512)))


def _conv3x3_param_grads(gm, x, weight, bias):
    """Weight / bias gradient of one conv3x3: straight into the existing ``.grad`` buffers (the trainer's bucket) under
    ``mlp.direct_grad_accumulation`` -- returns (None, None) then -- else as fresh tensors."""
    wg, bg = weight.grad, (bias.grad if bias is not None else None)
    if (mlp._DIRECT_ACCUM and wg is not None and wg.permute(0, 2, 3, 1).is_contiguous()
            and (bias is None or (bg is not None and bg.is_contiguous()))):
        side = mlp._CONV_WGRAD_STREAM
        if side is not None and x.shape[2] * x.shape[3] <= OVERLAP_MAX_PIXELS:
            # small planes: the weight gradient (off the backward's critical path: nothing reads the bucket before the
            # optimizer step) runs on a side stream beside the next layers' data gradients -- these launches are
            # latency-bound and leave most CUs idle; the trainer joins the stream at the end of the tile
            with mlp.fork_to(side):
                conv3x3_wgrad_(gm, x, wg, bg, accumulate=True, defer=True)
            mlp.hold(gm, x)
        else:
            conv3x3_wgrad_(gm, x, wg, bg, accumulate=True, defer=True)
        return None, None
    dw = torch.empty_like(weight, memory_format=torch.channels_last)
    db = torch.empty_like(bias) if bias is not None else None
    conv3x3_wgrad_(gm, x, dw, db)
    return dw, db


# ------------------------------------------------------------------------------------------------ conv3x3, few input channels
def conv3x3_small_supported(x: torch.Tensor, conv) -> bool:
    """The image U-Net's first layer (encoder/unet.py:112-187: Conv2d(3, 32, 3, padding=1)): csrc/conv_small.hip."""
    return (USE_HIP_CONV and isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"
            and conv.in_channels <= 8 and conv.out_channels % 4 == 0 and conv.out_channels <= 64 and conv.bias is not None
            and x.dim() == 4 and x.is_cuda and x.dtype == torch.float32)


class _Conv3x3Small(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu: bool, grad_premasked: bool):
        x = _as_cl(x)
        w = _w_cl(weight)
        b, cin, h, wd = x.shape
        cout = w.shape[0]
        y = _empty_cl(b, cout, h, wd, x.device)
        _lib.call("t2h_conv3x3_smallcin_fwd", _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y), b, h, wd, cin, cout,
                  _lib.RELU_OUT if relu else 0, _lib.stream(), nbytes=4 * (x.numel() + y.numel()),
                  flops=2 * 9 * cin * cout * b * h * wd, tag=_lib.timing() and f"t2h_conv3x3_smallcin_fwd[{cin}->{cout},{h}x{wd}]")
        need_y = relu and not grad_premasked
        ctx.save_for_backward(x, weight, bias, y if need_y else None)
        ctx.conf = (relu, grad_premasked)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, bias, y = ctx.saved_tensors
        relu, grad_premasked = ctx.conf
        g = _as_cl(g)
        w = _w_cl(weight)
        b, cin, h, wd = x.shape
        cout = w.shape[0]
        if relu and not grad_premasked:
            gm = torch.empty_like(g, memory_format=torch.channels_last)
            _lib.call("t2h_relu_mask", _lib.ptr(g), _lib.ptr(y), _lib.ptr(gm), g.numel(), _lib.stream(), nbytes=12 * g.numel())
        else:
            gm = g
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x, memory_format=torch.channels_last)
            _lib.call("t2h_conv3x3_smallcin_dgrad", _lib.ptr(gm), _lib.ptr(w), _lib.ptr(dx), b, h, wd, cin, cout, 0,
                      _lib.stream(), nbytes=4 * (gm.numel() + dx.numel()), flops=2 * 9 * cin * cout * b * h * wd,
                      tag=_lib.timing() and f"t2h_conv3x3_smallcin_dgrad[{cout}->{cin},{h}x{wd}]")
        wg, bg = weight.grad, bias.grad
        direct = (mlp._DIRECT_ACCUM and wg is not None and wg.permute(0, 2, 3, 1).is_contiguous() and bg is not None
                  and bg.is_contiguous())
        dw = wg if direct else torch.empty_like(weight, memory_format=torch.channels_last)
        db = bg if direct else torch.empty_like(bias)
        nws = _lib.ws_bytes("t2h_conv3x3_smallcin_wgrad_workspace_bytes", cin, cout)
        ws = _lib.workspace(nws, g.device)
        _lib.call("t2h_conv3x3_smallcin_wgrad", _lib.ptr(gm), _lib.ptr(x), _lib.ptr(dw), _lib.ptr(db), b, h, wd, cin, cout,
                  _lib.ACCUM if direct else 0, _lib.ptr(ws), nws, _lib.stream(), nbytes=4 * (gm.numel() + x.numel()),
                  flops=2 * 9 * cin * cout * b * h * wd, tag=_lib.timing() and f"t2h_conv3x3_smallcin_wgrad[{cin}->{cout},{h}x{wd}]")
        return (dx, None, None, None, None) if direct else (dx, dw, db, None, None)


def conv_bias_act(x: torch.Tensor, conv: torch.nn.Conv2d, relu: bool = True, mask_input: bool = False,
                  grad_premasked: bool = False) -> torch.Tensor:
    """``relu(conv(x))`` (or ``conv(x)``).  3x3 / stride 1 / padding 1 convs with 16-aligned channel counts run on the
    implicit-GEMM kernels of csrc/conv.hip; other convs stay on MIOpen with the bias add, the ReLU and their backward
    fused into one pass each."""
    if conv3x3_supported(x, conv):
        return _Conv3x3.apply(x, conv.weight, conv.bias, relu, mask_input, grad_premasked)
    if conv3x3_small_supported(x, conv) and not mask_input:
        return _Conv3x3Small.apply(x, conv.weight, conv.bias, relu, grad_premasked)
    if mask_input or grad_premasked:
        raise ValueError("conv_bias_act: mask_input / grad_premasked need the HIP conv3x3 path")
    _lib.library_fallback(f"conv2d {conv.in_channels}->{conv.out_channels} k{tuple(conv.kernel_size)} s{tuple(conv.stride)} (MIOpen)")
    if conv.bias is None or conv.out_channels % 4 or conv.groups != 1 or conv.dilation != (1, 1) or not x.is_cuda:
        y = conv(x)
        return F.relu(y) if relu else y
    return _ConvBiasAct.apply(x, conv.weight, conv.bias, conv.stride, conv.padding, relu)


def conv3x3_chain(x: torch.Tensor, convs, relu_last: bool = True) -> torch.Tensor:
    """``relu(conv_n(... relu(conv_1(x))))`` where every intermediate activation feeds the next conv only (the
    conv1 -> conv2 pairs of alto.py:98-99,226-227): each data gradient applies the previous ReLU's mask in its
    epilogue, so only the last ReLU needs a backward pass of its own."""
    convs = list(convs)
    # conv i applies the ReLU mask of its input in its data gradient (mask_input) iff it runs on the implicit-GEMM kernels;
    # its producer (also a small-Cin first layer) then skips its own ReLU backward (grad_premasked)
    # (the predicate reads only x's rank / device / dtype / plane size, which a 3x3-s1-p1 chain preserves -- no probe tensor)
    masks = [conv3x3_supported(x, c) for c in convs]
    own = [masks[i] or conv3x3_small_supported(x, c) for i, c in enumerate(convs)]
    for i, conv in enumerate(convs):
        last = i == len(convs) - 1
        mask_in = i > 0 and masks[i] and own[i - 1]
        premasked = (not last) and own[i] and masks[i + 1]
        x = conv_bias_act(x, conv, relu=relu_last or not last, mask_input=mask_in, grad_premasked=premasked)
    return x


class _Conv1x1(torch.autograd.Function):
    """1x1 convolution = the per-point GEMM kernels (always exact fp32 here) on the [pixels, C] rows of the NHWC plane."""

    @staticmethod
    def forward(ctx, x, weight, bias, addend):
        x = _as_cl(x)
        b, cin, h, wd = x.shape
        cout, m = weight.shape[0], b * h * wd
        w2 = weight.reshape(cout, cin).contiguous()
        y = _empty_cl(b, cout, h, wd, x.device)
        if addend is not None:
            addend = _as_cl(addend)
            if addend.shape != y.shape:
                raise ValueError("conv1x1: addend must have the output's shape")
        _lib.call("t2h_linear_fwd_add", _lib.ptr(x), cin, _lib.ptr(w2), _lib.ptr(bias) if bias is not None else None,
                  _lib.ptr(addend) if addend is not None else None, cout, _lib.ptr(y), cout, m, cin, cout, 0, _lib.stream(),
                  nbytes=4 * (m * cin + m * cout * (2 if addend is not None else 1) + cin * cout),
                  flops=2 * m * cin * cout, tag=_lib.timing() and f"t2h_linear_fwd[K={cin},N={cout}]")
        ctx.has_addend = addend is not None
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, bias = ctx.saved_tensors
        g, ldg = _cl_ld(g)                                  # (a channel slice of a concatenation's gradient is read in place)
        b, cin, h, wd = x.shape
        cout, m = weight.shape[0], b * h * wd
        w2 = weight.reshape(cout, cin).contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _empty_cl(b, cin, h, wd, x.device)
            _lib.call("t2h_linear_dgrad", _lib.ptr(g), ldg, _lib.ptr(w2), _lib.ptr(dx), cin, m, cin, cout, None, 0, 0,
                      _lib.stream(), nbytes=4 * (m * cin + m * cout + cin * cout), flops=2 * m * cin * cout,
                      tag=_lib.timing() and f"t2h_linear_dgrad[N={cout},K={cin}]")
        wg, bg = weight.grad, (bias.grad if bias is not None else None)
        direct = (mlp._DIRECT_ACCUM and wg is not None and wg.permute(0, 2, 3, 1).is_contiguous()
                  and (bias is None or (bg is not None and bg.is_contiguous())))
        dw = wg if direct else torch.empty_like(weight, memory_format=torch.channels_last)
        db = bg if direct else (torch.empty_like(bias) if bias is not None else None)
        lib = _lib.load()
        nws = _lib.ws_bytes("t2h_linear_wgrad_workspace_bytes", m, cin, cout)
        ws = _lib.workspace(nws, g.device)
        _lib.call("t2h_linear_wgrad", _lib.ptr(g), ldg, _lib.ptr(x), cin, m, cin, cout, _lib.ACCUM if direct else 0,
                  _lib.ptr(dw), _lib.ptr(db) if db is not None else None, _lib.ptr(ws), nws, _lib.stream(),
                  nbytes=4 * (m * cin + m * cout + cin * cout), flops=2 * m * cin * cout,
                  tag=_lib.timing() and f"t2h_linear_wgrad[N={cout},K={cin}]")
        ga = g if ctx.has_addend else None
        return (dx, None, None, ga) if direct else (dx, dw, db, ga)


def conv1x1(x: torch.Tensor, conv: torch.nn.Conv2d, addend: torch.Tensor = None) -> torch.Tensor:
    """``conv(x)`` or ``addend + conv(x)`` for a 1x1 convolution (alto.py conv1x1 residuals, upconv_noup, conv_final) on the
    per-point GEMM kernels; the addend rides the GEMM epilogue."""
    ok = (USE_HIP_CONV and type(conv) is torch.nn.Conv2d and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
          and conv.groups == 1 and conv.in_channels % 4 == 0 and conv.out_channels % 4 == 0 and x.is_cuda
          and x.dtype == torch.float32)
    if not ok:
        _lib.library_fallback(f"conv1x1 {conv.in_channels}->{conv.out_channels} (MIOpen)")
        return conv(x) if addend is None else addend + conv(x)
    return _Conv1x1.apply(x, conv.weight, conv.bias, addend)


# ------------------------------------------------------------------------------------------------ upconv2x2 (HIP)
def upconv2x2_supported(x: torch.Tensor, conv) -> bool:
    return (USE_HIP_CONV and type(conv) is torch.nn.ConvTranspose2d and conv.kernel_size == (2, 2) and conv.stride == (2, 2)
            and conv.padding == (0, 0) and conv.output_padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.in_channels % 16 == 0 and conv.out_channels % 16 == 0 and x.dim() == 4 and x.is_cuda
            and x.dtype == torch.float32 and _pow2(x.shape[2]) and _pow2(x.shape[3]))


UPCONV_BX3 = os.environ.get("T2H_UPCONV_BX3", "1") != "0"      # A/B: 0 = transposed convolutions stay on conv.hip (fp32 MFMA)


def _up_bx3(x, weight, w) -> bool:
    """The transposed convolution runs on the split-bf16 kernels (csrc/conv_bx3.hip, 1-tap form with a scattering epilogue /
    gathering loader) when the precision mode asks for them, the shape fits and the weight already lies [Cin][2][2][Cout]."""
    b, cin, h, wd = x.shape
    return bool(UPCONV_BX3 and CONV_PRECISION in ("bf16x3", "f16x2", "bf16") and w is weight and b * h * wd >= BX3_MIN_PIXELS
                and _lib.ws_bytes("t2h_upconv2x2_bx3_supported", b, h, wd, cin, weight.shape[1]))


class _UpConv2x2(torch.autograd.Function):
    """nn.ConvTranspose2d(kernel_size=2, stride=2) (upconv2x2, alto.py:175,215-218,236) on csrc/conv.hip."""

    @staticmethod
    def forward(ctx, x, weight, bias, addend):
        x = _as_cl(x)
        w = _w_cl(weight)                                   # [Cin][2][2][Cout] in memory
        b, cin, h, wd = x.shape
        cout = weight.shape[1]
        y = _empty_cl(b, cout, 2 * h, 2 * wd, x.device)
        if addend is not None:
            addend = _as_cl(addend)
            if addend.shape != y.shape:
                raise ValueError("upconv2x2: addend must have the output's shape")
        ctx.has_addend = addend is not None
        ctx.bx3 = _up_bx3(x, weight, w)
        if ctx.bx3:
            _lib.call("t2h_upconv2x2_bx3_fwd", _lib.ptr(x), _lib.ptr(split_weights.get_up(w, True)),
                      _lib.ptr(bias) if bias is not None else None, _lib.ptr(addend) if addend is not None else None, _lib.ptr(y),
                      b, h, wd, cin, cout, _lib.F16X2 if _h2() else 0, _lib.stream(),
                      nbytes=4 * (x.numel() + y.numel() * (2 if addend is not None else 1)) + 6 * w.numel(),
                      flops=2 * 4 * cin * cout * b * h * wd, tag=_lib.timing() and f"t2h_upconv2x2_bx3_fwd[{cin}->{cout},{h}x{wd}]")
            ctx.save_for_backward(x, weight, bias)
            return y
        _lib.call("t2h_upconv2x2_fwd_add", _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias) if bias is not None else None,
                  _lib.ptr(addend) if addend is not None else None, _lib.ptr(y),
                  b, h, wd, cin, cout, 0, _lib.stream(),
                  nbytes=4 * (x.numel() + y.numel() * (2 if addend is not None else 1) + w.numel()),
                  flops=2 * 4 * cin * cout * b * h * wd, tag=_lib.timing() and f"t2h_upconv2x2_fwd[{cin}->{cout},{h}x{wd}]")
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, bias = ctx.saved_tensors
        b, cin, h, wd = x.shape
        cout = weight.shape[1]
        up = "t2h_upconv2x2_bx3_wgrad" if (ctx.bx3 and BX3_WGRAD and wd >= 32) else "t2h_upconv2x2_wgrad_bias"
        g0 = g
        # the split kernels read a channel slice of a concatenation's gradient in place (pixel stride ldg); the fp32 ones want it dense
        g, ldg = _cl_ld(g) if (ctx.bx3 and up == "t2h_upconv2x2_bx3_wgrad") else (_as_cl(g), cout)
        w = _w_cl(weight)
        lib = _lib.load()
        flops = 2 * 4 * cin * cout * b * h * wd
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x, memory_format=torch.channels_last)
        if dx is not None and ctx.bx3:
            nws = _lib.ws_bytes("t2h_upconv2x2_bx3_dgrad_workspace_bytes", b, h, wd, cin, cout)
            ws = _lib.workspace(nws, g.device)
            _lib.call("t2h_upconv2x2_bx3_dgrad", _lib.ptr(g), ldg, _lib.ptr(split_weights.get_up(w, False)), _lib.ptr(dx),
                      b, h, wd, cin, cout, _lib.F16X2 if _h2() else 0, _lib.ptr(ws), nws, _lib.stream(),
                      nbytes=4 * (g.numel() + dx.numel()) + 6 * w.numel(), flops=flops,
                      tag=_lib.timing() and f"t2h_upconv2x2_bx3_dgrad[{cout}->{cin},{h}x{wd}]")
        elif dx is not None:
            nws = _lib.ws_bytes("t2h_upconv2x2_dgrad_workspace_bytes", b, h, wd, cin, cout)
            ws = _lib.workspace(nws, g.device)
            _lib.call("t2h_upconv2x2_dgrad", _lib.ptr(g), _lib.ptr(w), _lib.ptr(dx), b, h, wd, cin, cout, 0, _lib.ptr(ws), nws,
                      _lib.stream(), nbytes=4 * (g.numel() + dx.numel() + w.numel()), flops=flops,
                      tag=_lib.timing() and f"t2h_upconv2x2_dgrad[{cout}->{cin},{h}x{wd}]")
        wg, bg = weight.grad, (bias.grad if bias is not None else None)
        direct = (mlp._DIRECT_ACCUM and wg is not None and wg.permute(0, 2, 3, 1).is_contiguous()
                  and (bias is None or (bg is not None and bg.is_contiguous())))
        dw = wg if direct else torch.empty_like(weight, memory_format=torch.channels_last)
        db = bg if direct else (torch.empty_like(bias) if bias is not None else None)
        nws = _lib.ws_bytes(up.replace("_bias", "") + "_workspace_bytes", b, h, wd, cin, cout)
        # weight AND bias gradient from one kernel (the bias gradient is the column sum of the dY tiles it stages anyway); like the
        # 3x3 weight gradients on the trainer's side stream when there is one (_conv3x3_param_grads) -- unless g is also handed on
        # as the addend's gradient: its consumer may accumulate into it in place on the main stream (mlp.sole_owner)
        side = mlp._CONV_WGRAD_STREAM if (direct and not ctx.has_addend) else None
        with mlp.fork_to(side) if side is not None else contextlib.nullcontext():
            ws = _lib.workspace(nws, g.device)
            _lib.call(up, _lib.ptr(g), *((ldg,) if up == "t2h_upconv2x2_bx3_wgrad" else ()), _lib.ptr(x), _lib.ptr(dw),
                      None if db is None else _lib.ptr(db), b, h, wd, cin, cout,
                      ((_lib.ACCUM | _lib.defer_reduce(ws, dw)) if direct else 0) | (_lib.F16X2 if (_h2() and up.endswith("bx3_wgrad")) else 0),
                      _lib.ptr(ws), nws, _lib.stream(), nbytes=4 * (g.numel() + x.numel() + dw.numel()), flops=flops,
                      tag=_lib.timing() and f"{up}[{cin}->{cout},{h}x{wd}]")
        if side is not None:
            mlp.hold(g, x)
        ga = g0 if ctx.has_addend else None
        return (dx, None, None, ga) if direct else (dx, dw, db, ga)


def upconv2x2(x: torch.Tensor, conv, addend: torch.Tensor = None) -> torch.Tensor:
    """``conv(x)`` or ``addend + conv(x)`` for the 2x2 stride-2 transposed convolutions of the ALTO up path."""
    if not upconv2x2_supported(x, conv):
        _lib.library_fallback(f"conv_transpose2d {conv.in_channels}->{conv.out_channels} (MIOpen)")
        return conv(x) if addend is None else addend + conv(x)
    return _UpConv2x2.apply(x, conv.weight, conv.bias, addend)


# ------------------------------------------------------------------------------------------------ concat-free 1x1 head
def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])
    return arr


def _head_fwd(xs, weight, bias):
    b, _, h, w = xs[0].shape
    chans = [x.shape[1] for x in xs]
    wflat = weight.reshape(-1).contiguous()
    out = torch.empty(b, 1, h, w, dtype=torch.float32, device=xs[0].device)
    xarr = _ptr_array(xs)
    carr = (ctypes.c_int * len(xs))(*chans)
    _lib.call("t2h_head1x1_fwd", ctypes.cast(xarr, ctypes.c_void_p), ctypes.cast(carr, ctypes.c_void_p), len(xs),
              _lib.ptr(wflat), _lib.ptr(bias) if bias is not None else None, b * h * w, _lib.ptr(out), _lib.stream(),
              nbytes=4 * (sum(chans) + 1) * b * h * w)
    return out


def _head_bwd(xs, dxs, weight, bias, g, relu_inputs=()):
    """dxs[i] (may be None) = g * w_i, times (x_i > 0) for i in ``relu_inputs``; returns (dw, db)."""
    g = g.contiguous()
    b, _, h, w = xs[0].shape
    chans = [x.shape[1] for x in xs]
    ctot = sum(chans)
    wflat = weight.reshape(-1).contiguous()
    # under mlp.direct_grad_accumulation: straight into the existing .grad buffers (the trainer's bucket), (None, None) returned --
    # r05: this head was the last layer whose gradients still went through autograd's AccumulateGrad (two adds per tile, and in the
    # tile pipeline a hop to the stream the accumulator node was created on)
    wg, bg = weight.grad, (bias.grad if bias is not None else None)
    direct = (mlp._DIRECT_ACCUM and wg is not None and wg.is_contiguous() and wg.numel() == ctot
              and (bias is None or (bg is not None and bg.is_contiguous())))
    if direct:
        dw, db = wg, bg
    else:
        dw = torch.empty(ctot, dtype=torch.float32, device=g.device)
        db = torch.empty(1, dtype=torch.float32, device=g.device) if bias is not None else None
    ws_bytes = _lib.ws_bytes("t2h_head1x1_bwd_workspace_bytes", b * h * w, ctot)
    ws = _lib.workspace(ws_bytes, g.device)
    xarr, dxarr = _ptr_array(xs), _ptr_array(dxs)
    carr = (ctypes.c_int * len(xs))(*chans)
    flags = sum(1 << (8 + i) for i in relu_inputs) | (2 if direct else 0)
    nmask = sum(chans[i] for i in relu_inputs)
    _lib.call("t2h_head1x1_bwd", ctypes.cast(xarr, ctypes.c_void_p), ctypes.cast(dxarr, ctypes.c_void_p),
              ctypes.cast(carr, ctypes.c_void_p), len(xs), _lib.ptr(wflat), _lib.ptr(g), b * h * w, flags, _lib.ptr(dw),
              _lib.ptr(db) if db is not None else None, _lib.ptr(ws), ws_bytes, _lib.stream(),
              nbytes=4 * (2 * ctot + nmask + 1) * b * h * w)
    if direct:
        return None, None
    return dw.reshape(weight.shape), db


class _Head1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, bias, *xs):
        xs = [_as_cl(x) for x in xs]
        out = _head_fwd(xs, weight, bias)
        ctx.save_for_backward(weight, bias, *xs)
        return out

    @staticmethod
    def backward(ctx, g):
        weight, bias, *xs = ctx.saved_tensors
        dxs = [torch.empty_like(x, memory_format=torch.channels_last) if ctx.needs_input_grad[2 + i] else None
               for i, x in enumerate(xs)]
        dw, db = _head_bwd(xs, dxs, weight, bias, g)
        return (dw, db, *dxs)


class _ConvDecoder(torch.autograd.Function):
    """ConvDecoder of pixel.py:20-32 as one autograd node: x1 = relu(conv1(x)), x2 = relu(conv2(x1)), x3 = relu(conv3(x2)),
    out = conv4(cat[x, x1, x2, x3]).  Every activation has two consumers (the next conv and the head); in the backward
    the head writes its rank-1 share of each gradient already ReLU-masked, and every conv data gradient accumulates onto
    it with the same mask in its epilogue -- no gradient-sum and no ReLU-backward passes over the 512 x 512 planes."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, w4, b4):
        acts = [_as_cl(x)]
        b, _, h, wd = acts[0].shape
        for w, bias in ((w1, b1), (w2, b2), (w3, b3)):
            y = _empty_cl(b, w.shape[0], h, wd, x.device)
            conv3x3_fwd_(acts[-1], _w_cl(w), bias, y, relu=True)
            acts.append(y)
        out = _head_fwd(acts, w4, b4)
        ctx.save_for_backward(w1, b1, w2, b2, w3, b3, w4, b4, *acts)
        return out

    @staticmethod
    def backward(ctx, g):
        w1, b1, w2, b2, w3, b3, w4, b4, *acts = ctx.saved_tensors
        need_x = ctx.needs_input_grad[0]
        dxs = [torch.empty_like(a, memory_format=torch.channels_last) if (i > 0 or need_x) else None
               for i, a in enumerate(acts)]
        g = g.contiguous()
        # r06: where the data gradient can form the head's rank-1 share itself (g[pixel] * w4[channel], same mask) the head writes
        # no gradient for that activation and the data gradient reads no old values: 0.65 GB less traffic per tile at 512 x 512
        convs = ((w1, b1), (w2, b2), (w3, b3))
        fold = [dxs[i] is not None and dgrad_rank1_ok(dxs[i + 1], _w_cl(convs[i][0])) for i in range(3)]      # input i of the head
        w4flat = w4.reshape(-1).contiguous()
        starts = [0]
        for a in acts:
            starts.append(starts[-1] + a.shape[1])
        dw4, db4 = _head_bwd(acts, [None if (i < 3 and fold[i]) else d for i, d in enumerate(dxs)], w4, b4, g, relu_inputs=(1, 2, 3))
        grads = []
        for level, (w, bias) in reversed(list(enumerate(convs, start=1))):
            gm, xin = dxs[level], acts[level - 1]            # gm: complete and already ReLU-masked
            grads.append(_conv3x3_param_grads(gm, xin, w, bias))
            if dxs[level - 1] is None:
                continue
            if fold[level - 1]:
                conv3x3_dgrad_rank1_(gm, _w_cl(w), dxs[level - 1], xin if level > 1 else None, g,
                                     w4flat[starts[level - 1]:starts[level]])
            else:
                conv3x3_dgrad_(gm, _w_cl(w), dxs[level - 1], mask=xin if level > 1 else None, accumulate=True)
        (dw3, db3), (dw2, db2), (dw1, db1) = grads
        return dxs[0], dw1, db1, dw2, db2, dw3, db3, dw4, db4


def conv_decoder(x: torch.Tensor, conv1, conv2, conv3, conv4) -> torch.Tensor:
    """The whole ConvDecoder head (pixel.py:20-32) on the HIP kernels; falls back to the unfused pieces when a layer is
    outside the conv3x3 kernels' geometry."""
    if (all(conv3x3_supported(x, c) and c.bias is not None for c in (conv1, conv2, conv3)) and conv4.out_channels == 1
            and conv4.kernel_size == (1, 1) and conv4.in_channels == x.shape[1] + conv1.out_channels + conv2.out_channels
            + conv3.out_channels and conv2.in_channels == conv1.out_channels and conv3.in_channels == conv2.out_channels):
        return _ConvDecoder.apply(x, conv1.weight, conv1.bias, conv2.weight, conv2.bias, conv3.weight, conv3.bias,
                                  conv4.weight, conv4.bias)
    x1 = conv_bias_act(x, conv1)
    x2 = conv_bias_act(x1, conv2)
    x3 = conv_bias_act(x2, conv3)
    return head1x1([x, x1, x2, x3], conv4)


def head1x1(xs, conv: torch.nn.Conv2d) -> torch.Tensor:
    """``conv(torch.cat(xs, dim=1))`` for a 1x1 conv with ONE output channel, without materialising the concat."""
    ok = (conv.out_channels == 1 and conv.kernel_size == (1, 1) and 1 <= len(xs) <= 4 and all(x.shape[1] % 4 == 0 for x in xs)
          and sum(x.shape[1] for x in xs) == conv.in_channels and all(x.is_cuda for x in xs))
    if not ok:
        return conv(torch.cat(list(xs), dim=1))
    return _Head1x1.apply(conv.weight, conv.bias, *xs)


# ------------------------------------------------------------------------------------------------ 2x2 max-pool
class _MaxPool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _as_cl(x)
        b, c, h, w = x.shape
        y = _empty_cl(b, c, h // 2, w // 2, x.device)
        which = torch.empty(b * (h // 2) * (w // 2) * c, dtype=torch.uint8, device=x.device)
        _lib.call("t2h_maxpool2x2_nhwc_fwd", _lib.ptr(x), b, h, w, c, _lib.ptr(y), _lib.ptr(which), _lib.stream(),
                  nbytes=4 * x.numel() + 5 * y.numel(), tag="t2h_maxpool2x2_fwd")
        ctx.save_for_backward(which)
        ctx.shape = (b, c, h, w)
        return y

    @staticmethod
    def backward(ctx, g):
        (which,) = ctx.saved_tensors
        b, c, h, w = ctx.shape
        g = _as_cl(g)
        gin = _empty_cl(b, c, h, w, g.device)
        _lib.call("t2h_maxpool2x2_nhwc_bwd", _lib.ptr(g), _lib.ptr(which), b, h, w, c, _lib.ptr(gin), _lib.stream(),
                  nbytes=5 * g.numel() + 4 * gin.numel(), tag="t2h_maxpool2x2_bwd")
        return gin


class _MaxPool2x2Thru(torch.autograd.Function):
    """``(maxpool2x2(x), x)``: ``x`` is also a U-Net skip connection (alto.py:135-138, 373-378); the pooling backward adds the
    skip's gradient while it scatters its own (``t2h_maxpool2x2_nhwc_bwd_add``)."""

    @staticmethod
    def forward(ctx, x):
        y = _MaxPool2x2.forward(ctx, x)
        return y, _alias(x)

    @staticmethod
    def backward(ctx, g, gthru):
        if g is None:
            return gthru
        (which,) = ctx.saved_tensors
        b, c, h, w = ctx.shape
        g = _as_cl(g)
        addend, ld_add = (None, 0) if gthru is None else _cl_ld(gthru)   # (the skip's gradient: a slice of the concatenation's)
        gin = _empty_cl(b, c, h, w, g.device)
        _lib.call("t2h_maxpool2x2_nhwc_bwd_add", _lib.ptr(g), _lib.ptr(which), b, h, w, c,
                  None if addend is None else _lib.ptr(addend), ld_add, _lib.ptr(gin), _lib.stream(),
                  nbytes=5 * g.numel() + 4 * gin.numel() * (2 if addend is not None else 1), tag="t2h_maxpool2x2_bwd")
        return gin


def maxpool2x2_thru(x: torch.Tensor, pool: torch.nn.MaxPool2d = None):
    """``(maxpool2x2(x), x)`` -- use the returned ``x`` for the skip connection (see ``_MaxPool2x2Thru``)."""
    if not _maxpool_ok(x, pool):
        _lib.library_fallback(f"max_pool2d on {tuple(x.shape)} (ATen)")
        return (pool(x) if pool is not None else F.max_pool2d(x, 2, 2)), x
    return _MaxPool2x2Thru.apply(x)


def _maxpool_ok(x: torch.Tensor, pool) -> bool:
    ok = (USE_HIP_CONV and x.dim() == 4 and x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0
          and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and x.shape[2] >= 2 and x.shape[3] >= 2)
    if pool is not None:
        k, st = pool.kernel_size, pool.stride
        ok = ok and (k in (2, (2, 2))) and (st in (2, (2, 2))) and pool.padding in (0, (0, 0)) and not pool.ceil_mode \
            and pool.dilation in (1, (1, 1)) and not pool.return_indices
    return ok


def maxpool2x2(x: torch.Tensor, pool: torch.nn.MaxPool2d = None) -> torch.Tensor:
    """``nn.MaxPool2d(kernel_size=2, stride=2)(x)`` on channels_last planes (same winners on ties as ATen)."""
    ok = (USE_HIP_CONV and x.dim() == 4 and x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0
          and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and x.shape[2] >= 2 and x.shape[3] >= 2)
    if pool is not None:
        k, st = pool.kernel_size, pool.stride
        ok = ok and (k in (2, (2, 2))) and (st in (2, (2, 2))) and pool.padding in (0, (0, 0)) and not pool.ceil_mode \
            and pool.dilation in (1, (1, 1)) and not pool.return_indices
    if not ok:
        _lib.library_fallback(f"max_pool2d on {tuple(x.shape)} (ATen)")
        return pool(x) if pool is not None else F.max_pool2d(x, 2, 2)
    return _MaxPool2x2.apply(x)


# ------------------------------------------------------------------------------------------------ NHWC upsample
class _UpsampleCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size: int, addend):
        x = _as_cl(x)
        b, c, h, w = x.shape
        out = _empty_cl(b, c, size, size, x.device)
        if addend is not None:
            addend = _as_cl(addend)
            if addend.shape != out.shape:
                raise ValueError("upsample_bilinear_cl: addend must already have the output size")
        _lib.call("t2h_upsample_bilinear_nhwc_fwd", _lib.ptr(x), _lib.ptr(addend) if addend is not None else None, b, c, h, w,
                  size, size, _lib.ptr(out), _lib.stream(),
                  nbytes=4 * (x.numel() + out.numel() * (2 if addend is not None else 1)), tag="t2h_upsample_bilinear_fwd")
        ctx.shape = (b, c, h, w, size)
        ctx.has_addend = addend is not None
        return out

    @staticmethod
    def backward(ctx, g):
        b, c, h, w, size = ctx.shape
        g = _as_cl(g)
        gin = _empty_cl(b, c, h, w, g.device)
        _lib.call("t2h_upsample_bilinear_nhwc_bwd", _lib.ptr(g), b, c, h, w, size, size, _lib.ptr(gin), _lib.stream(),
                  nbytes=4 * (g.numel() + gin.numel()), tag="t2h_upsample_bilinear_bwd")
        return gin, None, (g if ctx.has_addend else None)


class _Upsample2x(torch.autograd.Function):
    """nn.Upsample(mode='bilinear', scale_factor=2) (align_corners=False) on channels_last planes: upconv2x2(mode='upsample'),
    alto.py:23-35 / unet.py."""

    @staticmethod
    def forward(ctx, x):
        x = _as_cl(x)
        b, c, h, w = x.shape
        out = _empty_cl(b, c, 2 * h, 2 * w, x.device)
        _lib.call("t2h_upsample2x_nhwc_fwd", _lib.ptr(x), b, c, h, w, _lib.ptr(out), _lib.stream(),
                  nbytes=4 * (x.numel() + out.numel()))
        ctx.shape = (b, c, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        b, c, h, w = ctx.shape
        g = _as_cl(g)
        gin = _empty_cl(b, c, h, w, g.device)
        _lib.call("t2h_upsample2x_nhwc_bwd", _lib.ptr(g), b, c, h, w, _lib.ptr(gin), _lib.stream(),
                  nbytes=4 * (g.numel() + gin.numel()))
        return gin


def upsample2x(x: torch.Tensor) -> torch.Tensor:
    if x.dim() != 4 or x.shape[1] % 4 or not x.is_cuda or x.dtype != torch.float32:
        _lib.library_fallback("nn.Upsample(scale_factor=2) on a plane with C % 4 != 0 (ATen)")
        return torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear")
    return _Upsample2x.apply(x)


def upsample_conv1x1(x: torch.Tensor, seq, addend: torch.Tensor = None) -> torch.Tensor:
    """``seq(x)`` or ``addend + seq(x)`` for ``nn.Sequential(nn.Upsample(bilinear, x2), conv1x1)`` -- upconv2x2 with
    mode='upsample' (alto.py:31-35)."""
    return conv1x1(upsample2x(x), seq[1], addend)


def upsample_bilinear_cl(x: torch.Tensor, size: int, addend: torch.Tensor = None) -> torch.Tensor:
    if x.shape[1] % 4 or not x.is_cuda:
        raise ValueError("upsample_bilinear_cl needs a device tensor with C % 4 == 0")
    return _UpsampleCL.apply(x, int(size), addend)
