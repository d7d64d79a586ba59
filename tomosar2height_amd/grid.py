"""Grid-side fused operators (first step): the elementwise passes around the MIOpen convolutions of the
ALTO U-Net and the pixel decoder as HIP kernels on channels_last (NHWC) tensors.

    conv_bias_act(x, conv, relu)   conv (MIOpen, bias-free call) + fused bias/ReLU; backward = fused ReLU-mask +
                                   bias gradient, then MIOpen's data / weight gradients
    head1x1(xs, conv4)             torch.cat(xs, 1) -> 1x1 conv to one channel without the concat (pixel.py:31)
    upsample_bilinear_cl(x, size)  F.interpolate(bilinear, align_corners=True) on channels_last planes (pixel.py:107)

Every function requires channels_last device tensors; callers (alto.py / pixel.py here) use them only when the model
runs in channels_last mode and otherwise keep the plain torch composition.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib


def is_cl(x: torch.Tensor) -> bool:
    """[B,C,H,W] tensor whose memory is dense NHWC (a C == 1 or H*W == 1 tensor counts when its NHWC view is contiguous)."""
    return x.dim() == 4 and x.is_cuda and x.dtype == torch.float32 and x.permute(0, 2, 3, 1).is_contiguous()


def _as_cl(x: torch.Tensor) -> torch.Tensor:
    return x if is_cl(x) else x.contiguous(memory_format=torch.channels_last)


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
        ws_bytes = _lib.load().t2h_bias_relu_bwd_workspace_bytes(b * h * w, c)
        ws = _lib.workspace(ws_bytes, g.device)
        _lib.call("t2h_bias_relu_bwd", _lib.ptr(g), _lib.ptr(y), _lib.ptr(gm) if relu else None, b * h * w, c,
                  1 if relu else 0, 0, _lib.ptr(dbias), _lib.ptr(ws), ws_bytes, _lib.stream(),
                  nbytes=(12 if relu else 4) * g.numel())
        dx, dw, _ = torch.ops.aten.convolution_backward(
            gm, x, weight, None, list(stride), list(padding), [1, 1], False, [0, 0], 1,
            [ctx.needs_input_grad[0], True, False])
        return dx, dw, dbias, None, None, None


def conv_bias_act(x: torch.Tensor, conv: torch.nn.Conv2d, relu: bool = True) -> torch.Tensor:
    """``relu(conv(x))`` (or ``conv(x)``) with the bias add, the ReLU and their backward fused into one pass each."""
    if conv.bias is None or conv.out_channels % 4 or conv.groups != 1 or conv.dilation != (1, 1) or not x.is_cuda:
        y = conv(x)
        return F.relu(y) if relu else y
    return _ConvBiasAct.apply(x, conv.weight, conv.bias, conv.stride, conv.padding, relu)


# ------------------------------------------------------------------------------------------------ concat-free 1x1 head
def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])
    return arr


class _Head1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, bias, *xs):
        xs = [_as_cl(x) for x in xs]
        b, _, h, w = xs[0].shape
        chans = [x.shape[1] for x in xs]
        wflat = weight.reshape(-1).contiguous()
        out = torch.empty(b, 1, h, w, dtype=torch.float32, device=xs[0].device)
        xarr = _ptr_array(xs)
        carr = (ctypes.c_int * len(xs))(*chans)
        _lib.call("t2h_head1x1_fwd", ctypes.cast(xarr, ctypes.c_void_p), ctypes.cast(carr, ctypes.c_void_p), len(xs),
                  _lib.ptr(wflat), _lib.ptr(bias) if bias is not None else None, b * h * w, _lib.ptr(out), _lib.stream(),
                  nbytes=4 * (sum(chans) + 1) * b * h * w)
        ctx.save_for_backward(weight, bias, *xs)
        return out

    @staticmethod
    def backward(ctx, g):
        weight, bias, *xs = ctx.saved_tensors
        g = g.contiguous()
        b, _, h, w = xs[0].shape
        chans = [x.shape[1] for x in xs]
        ctot = sum(chans)
        wflat = weight.reshape(-1).contiguous()
        dxs = [torch.empty_like(x, memory_format=torch.channels_last) if ctx.needs_input_grad[2 + i] else None
               for i, x in enumerate(xs)]
        dw = torch.empty(ctot, dtype=torch.float32, device=g.device)
        db = torch.empty(1, dtype=torch.float32, device=g.device) if bias is not None else None
        ws_bytes = _lib.load().t2h_head1x1_bwd_workspace_bytes(b * h * w, ctot)
        ws = _lib.workspace(ws_bytes, g.device)
        xarr, dxarr = _ptr_array(xs), _ptr_array(dxs)
        carr = (ctypes.c_int * len(xs))(*chans)
        _lib.call("t2h_head1x1_bwd", ctypes.cast(xarr, ctypes.c_void_p), ctypes.cast(dxarr, ctypes.c_void_p),
                  ctypes.cast(carr, ctypes.c_void_p), len(xs), _lib.ptr(wflat), _lib.ptr(g), b * h * w, 0, _lib.ptr(dw),
                  _lib.ptr(db) if db is not None else None, _lib.ptr(ws), ws_bytes, _lib.stream(),
                  nbytes=4 * (2 * ctot + 1) * b * h * w)
        return (dw.reshape(weight.shape), db, *dxs)


def head1x1(xs, conv: torch.nn.Conv2d) -> torch.Tensor:
    """``conv(torch.cat(xs, dim=1))`` for a 1x1 conv with ONE output channel, without materialising the concat."""
    ok = (conv.out_channels == 1 and conv.kernel_size == (1, 1) and 1 <= len(xs) <= 4 and all(x.shape[1] % 4 == 0 for x in xs)
          and sum(x.shape[1] for x in xs) == conv.in_channels and all(x.is_cuda for x in xs))
    if not ok:
        return conv(torch.cat(list(xs), dim=1))
    return _Head1x1.apply(conv.weight, conv.bias, *xs)


# ------------------------------------------------------------------------------------------------ NHWC upsample
class _UpsampleCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size: int, addend):
        x = _as_cl(x)
        b, c, h, w = x.shape
        out = torch.empty(b, c, size, size, dtype=torch.float32, device=x.device).contiguous(memory_format=torch.channels_last)
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
        gin = torch.empty(b, c, h, w, dtype=torch.float32, device=g.device).contiguous(memory_format=torch.channels_last)
        _lib.call("t2h_upsample_bilinear_nhwc_bwd", _lib.ptr(g), b, c, h, w, size, size, _lib.ptr(gin), _lib.stream(),
                  nbytes=4 * (g.numel() + gin.numel()), tag="t2h_upsample_bilinear_bwd")
        return gin, None, (g if ctx.has_addend else None)


def upsample_bilinear_cl(x: torch.Tensor, size: int, addend: torch.Tensor = None) -> torch.Tensor:
    if x.shape[1] % 4 or not x.is_cuda:
        raise ValueError("upsample_bilinear_cl needs a device tensor with C % 4 == 0")
    return _UpsampleCL.apply(x, int(size), addend)
