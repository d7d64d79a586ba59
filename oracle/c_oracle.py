"""TEST INFRASTRUCTURE -- ctypes/numpy front end of the C oracle (oracle/t2h_oracle.c)."""
import ctypes
import os

import numpy as np

from . import build as _build

_F = ctypes.POINTER(ctypes.c_float)
_L = ctypes.POINTER(ctypes.c_int64)
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.OUT
        if not os.path.exists(path):
            path = _build.build()
        _lib = ctypes.CDLL(path)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_F)


def _l(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_L)


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"t2h_oracle_{what} failed with code {rc}")


def coordinate2index(pts, reso):
    """pts [B,N,D>=2] -> int64 [B,1,N] (utils/coordinate.py:12-28)."""
    pts, pp = _f(pts)
    b, n, d = pts.shape
    out = np.empty((b, n), np.int64)
    _chk(lib().t2h_oracle_coordinate2index(pp, d, b, n, int(reso), out.ctypes.data_as(_L)), "coordinate2index")
    return out[:, None, :]


def scatter_max(feat, index, cells):
    """feat [B,N,C], index [B,1,N] -> (out [B,C,cells], arg [B,C,cells])."""
    feat, fp = _f(feat)
    index, ip = _l(np.asarray(index).reshape(feat.shape[0], -1))
    b, n, c = feat.shape
    out = np.empty((b, c, cells), np.float32)
    arg = np.empty((b, c, cells), np.int64)
    _chk(lib().t2h_oracle_scatter_max(fp, ip, b, n, c, int(cells), out.ctypes.data_as(_F),
                                      arg.ctypes.data_as(_L)), "scatter_max")
    return out, arg


def pool_local_fwd(feat, index, cells):
    feat, fp = _f(feat)
    index, ip = _l(np.asarray(index).reshape(feat.shape[0], -1))
    b, n, c = feat.shape
    pooled = np.empty((b, n, c), np.float32)
    arg = np.empty((b, c, cells), np.int64)
    _chk(lib().t2h_oracle_pool_local_fwd(fp, ip, b, n, c, int(cells), pooled.ctypes.data_as(_F),
                                         arg.ctypes.data_as(_L)), "pool_local_fwd")
    return pooled, arg


def pool_local_bwd(gpooled, index, arg, cells):
    gpooled, gp = _f(gpooled)
    b, n, c = gpooled.shape
    index, ip = _l(np.asarray(index).reshape(b, -1))
    arg, ap = _l(arg)
    gfeat = np.empty((b, n, c), np.float32)
    _chk(lib().t2h_oracle_pool_local_bwd(gp, ip, ap, b, n, c, int(cells), gfeat.ctypes.data_as(_F)),
         "pool_local_bwd")
    return gfeat


def scatter_mean_fwd(feat, index, reso):
    """feat [B,N,C] -> plane [B,C,reso,reso]."""
    feat, fp = _f(feat)
    b, n, c = feat.shape
    index, ip = _l(np.asarray(index).reshape(b, -1))
    plane = np.empty((b, c, reso * reso), np.float32)
    _chk(lib().t2h_oracle_scatter_mean_fwd(fp, ip, b, n, c, reso * reso, plane.ctypes.data_as(_F)),
         "scatter_mean_fwd")
    return plane.reshape(b, c, reso, reso)


def scatter_mean_bwd(gplane, index, n):
    gplane, gp = _f(gplane)
    b, c, r, _ = gplane.shape
    index, ip = _l(np.asarray(index).reshape(b, -1))
    gfeat = np.empty((b, n, c), np.float32)
    _chk(lib().t2h_oracle_scatter_mean_bwd(gp, ip, b, n, c, r * r, gfeat.ctypes.data_as(_F)),
         "scatter_mean_bwd")
    return gfeat


def grid_sample_fwd(plane, pts):
    """plane [B,C,H,W], pts [B,N,D>=2] -> [B,N,C] (alto.py:90-95 followed by the transpose at :122)."""
    plane, pp = _f(plane)
    pts, xp = _f(pts)
    b, c, h, w = plane.shape
    n, d = pts.shape[1], pts.shape[2]
    out = np.empty((b, n, c), np.float32)
    _chk(lib().t2h_oracle_grid_sample_fwd(pp, xp, d, b, c, h, w, n, out.ctypes.data_as(_F)), "grid_sample_fwd")
    return out


def grid_sample_bwd(gout, pts, h, w):
    gout, gp = _f(gout)
    pts, xp = _f(pts)
    b, n, c = gout.shape
    d = pts.shape[2]
    gplane = np.empty((b, c, h, w), np.float32)
    _chk(lib().t2h_oracle_grid_sample_bwd(gp, xp, d, b, c, h, w, n, gplane.ctypes.data_as(_F)),
         "grid_sample_bwd")
    return gplane


def upsample_bilinear_fwd(x, size):
    x, xp = _f(x)
    b, c, h, w = x.shape
    out = np.empty((b, c, size, size), np.float32)
    _chk(lib().t2h_oracle_upsample_bilinear_fwd(xp, b, c, h, w, size, size, out.ctypes.data_as(_F)),
         "upsample_bilinear_fwd")
    return out


def upsample_bilinear_bwd(gout, h, w):
    gout, gp = _f(gout)
    b, c, hh, ww = gout.shape
    gin = np.empty((b, c, h, w), np.float32)
    _chk(lib().t2h_oracle_upsample_bilinear_bwd(gp, b, c, h, w, hh, ww, gin.ctypes.data_as(_F)),
         "upsample_bilinear_bwd")
    return gin


def linear_fwd(x, w, bias=None, relu_in=False):
    x, xp = _f(x)
    w, wp = _f(w)
    m, k = x.shape
    nout = w.shape[0]
    bp = None
    if bias is not None:
        bias, bp = _f(bias)
    y = np.empty((m, nout), np.float32)
    _chk(lib().t2h_oracle_linear_fwd(xp, wp, bp, m, k, nout, int(relu_in), y.ctypes.data_as(_F)), "linear_fwd")
    return y


def resblock_fwd(x, w0, b0, w1, b1, ws=None):
    x, xp = _f(x)
    w0, w0p = _f(w0)
    b0, b0p = _f(b0)
    w1, w1p = _f(w1)
    b1, b1p = _f(b1)
    wsp = None
    if ws is not None:
        ws, wsp = _f(ws)
    m, cin = x.shape
    ch, cout = w0.shape[0], w1.shape[0]
    y = np.empty((m, cout), np.float32)
    _chk(lib().t2h_oracle_resblock_fwd(xp, w0p, b0p, w1p, b1p, wsp, m, cin, ch, cout, y.ctypes.data_as(_F)),
         "resblock_fwd")
    return y
