"""Deferred point features: the ALTO point update (reference alto.py:121-130, 245-255) without per-point features.

The reference keeps a per-point feature vector c_k through the U-Net levels,

    h_k = relu(fc_comm.0(sample(plane_k)))            [N, 2 C_k]      (alto.py:121-123)
    c_k = fc_comm.2(h_k) + fc_c(c_{k-1})              [N, C_k]        (alto.py:123-128)
    raster_k = scatter_mean(c_k)                      [r_k^2, C_k]    (alto.py:130)

and nothing but Linear layers ever touches c: it is consumed by the next level's fc_c and by scatter_mean, both linear.  So

    c_k      = sum_{j <= k} h_j A_{k,j}^T + c_base A_{k,base}^T + const_k,      A_{k,j} = Wc_k ... Wc_{j+1} W1_j
    raster_k = ( sum_j cellsum_{r_k}(h_j) A_{k,j}^T + cellsum_{r_k}(c_base) A_{k,base}^T ) / count + [count > 0] const_k

where cellsum_r is the per-cell SUM at resolution r.  The products run on the r_k^2 pixels instead of the N points, the
composed matrices A (at most 1024 x 512) are themselves tiny products, and c_k is never formed.  What stays per point is
what is non-linear: the hidden activations h_j (written once by ``t2h_sample_fwd_relu``, read once by ``t2h_segsum_fwd``;
their sums at the coarser resolutions come from 2x2 pooling of the finest one) and, in the backward, their masked gradient
(one ``t2h_segsum_bwd_multi`` gather from the <= 4 resolutions, read once by the sample adjoint).

Same function as the reference's chain, re-associated (fp32: 1e-6 relative; pinned by the full-size oracle comparison and
the float64 / same-mask checks of tests/test_hip_masks.py).  ``base`` is the last per-point feature tensor of the levels
that run point-wise (narrow levels, where the per-point products are cheap): deferral starts at the first level with at
least ``DEFER_MIN_CHANNELS`` channels on which the grid-first exchange applies (T2H_DEFER_MIN_CHANNELS, 0 = off).
"""
import ctypes
import os

import torch

from . import _lib, mlp, ops

DEFER_MIN_CHANNELS = int(os.environ.get("T2H_DEFER_MIN_CHANNELS", "256"))
FUSED_SAMPLE_BWD = os.environ.get("T2H_FUSED_SAMPLE_BWD", "1") != "0"      # A/B: 0 = separate gather + sample adjoint


# ------------------------------------------------------------------------------------------------ per-cell sums
def _segsum(tile, rows, level):
    """[N, C] rows -> [B r r, C] per-cell sums at ALTO level ``level`` (r = R >> level)."""
    n, c = rows.shape
    r = tile.R >> level
    plane = torch.empty(tile.B * r * r, c, dtype=torch.float32, device=rows.device)
    ws_bytes = _lib.load().t2h_segmean_workspace_bytes(tile.B, tile.N, tile.nbits, level, c)
    ws = _lib.workspace(ws_bytes, rows.device)
    _lib.call("t2h_segsum_fwd", _lib.ptr(rows), _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits, level, c, _lib.ptr(plane),
              _lib.ptr(ws), ws_bytes, _lib.stream(), nbytes=4 * c * n + 4 * n + 4 * plane.numel(),
              tag=f"t2h_segsum_fwd[C={c},r={r}]")
    return plane


def _sumpool(tile, fine, level_fine):
    """per-cell sums at level ``level_fine`` -> level_fine + 1 (half the resolution)."""
    c = fine.shape[1]
    r = tile.R >> level_fine
    coarse = torch.empty(tile.B * (r // 2) * (r // 2), c, dtype=torch.float32, device=fine.device)
    _lib.call("t2h_plane_sumpool2x2", _lib.ptr(fine), tile.B, r, c, _lib.ptr(coarse), _lib.stream(),
              nbytes=4 * (fine.numel() + coarse.numel()), tag=f"t2h_plane_sumpool2x2[C={c}]")
    return coarse


def cell_sums(tile, rows, levels):
    """Per-cell sums of ``rows`` at every ALTO level in ``levels`` (ascending = finest first): the finest from the rows, the
    others by 2x2 pooling (a cell is the union of its four children) -- the rows are read once."""
    out = {}
    cur, cur_level = _segsum(tile, rows, levels[0]), levels[0]
    out[cur_level] = cur
    for lv in levels[1:]:
        while cur_level < lv:
            cur, cur_level = _sumpool(tile, cur, cur_level), cur_level + 1
        out[lv] = cur
    return [out[lv] for lv in levels]


def _gather(tile, grads, levels, c, mask=None):
    """d rows [N, C] = (mask > 0 ?) sum over the given levels of gplane_l[cell_l(n)]  (adjoint of ``cell_sums``)."""
    planes = [(g.contiguous(), lv) for g, lv in zip(grads, levels) if g is not None]
    out = torch.empty(tile.n_points, c, dtype=torch.float32, device=tile.device)
    if not planes:
        return out.zero_()
    arr = (ctypes.c_void_p * len(planes))(*[p.data_ptr() for p, _ in planes])
    lvs = (ctypes.c_int * len(planes))(*[lv for _, lv in planes])
    _lib.call("t2h_segsum_bwd_multi", ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(lvs, ctypes.c_void_p), len(planes),
              _lib.ptr(tile.cell), tile.B, tile.N, tile.nbits, c, None if mask is None else _lib.ptr(mask), None,
              _lib.ptr(out), _lib.stream(),
              nbytes=4 * c * tile.n_points * (2 if mask is not None else 1) + 4 * tile.n_points + sum(4 * p.numel() for p, _ in planes),
              tag=f"t2h_segsum_bwd_multi[C={c},n={len(planes)}]")
    return out


class _HiddenSums(torch.autograd.Function):
    """Q = fc_comm.0 applied to the plane's pixels ([B r r, 2C] rows) -> per-cell sums of h = relu(sample(Q)) at ``levels``.
    h [N, 2C] lives only inside this node (saved for the mask); backward: gather + mask in one pass, then the sample adjoint."""

    @staticmethod
    def forward(ctx, q_rows, tile, r, levels):
        q_rows = q_rows.contiguous()
        c2 = q_rows.shape[1]
        h = torch.empty(tile.n_points, c2, dtype=torch.float32, device=q_rows.device)
        ctx.tile, ctx.r, ctx.levels, ctx.c2 = tile, r, tuple(levels), c2
        ctx.save_for_backward(h)
        # (interpolation + ReLU + the finest level's per-cell sums in ONE pass over the cells was built and measured in r03:
        # 414 us against 162 + 210 us at C = 1024 -- a cell's rows in sequence expose the tap loads' latency -- and dropped)
        _lib.call("t2h_sample_fwd_relu", _lib.ptr(q_rows), _lib.ptr(tile.pts), tile.dim, tile.B, tile.N, r, c2, _lib.ptr(h),
                  _lib.stream(), nbytes=4 * c2 * tile.n_points + 8 * tile.n_points + 4 * q_rows.numel(),
                  tag=f"t2h_sample_fwd_relu[C={c2},r={r}]")
        return tuple(cell_sums(tile, h, list(levels)))

    @staticmethod
    def backward(ctx, *grads):
        (h,) = ctx.saved_tensors
        tile, r, c2 = ctx.tile, ctx.r, ctx.c2
        level = tile.level(r)
        ws_bytes = _lib.load().t2h_sample_bwd_workspace_bytes(tile.B, tile.N, tile.nbits, level, c2)
        planes = [(g.contiguous(), lv) for g, lv in zip(grads, ctx.levels) if g is not None]
        if FUSED_SAMPLE_BWD and ws_bytes > 0 and planes and c2 % 4 == 0:
            # coarse level: gather + mask inside the sample adjoint's row load, dh [N, 2C] is never written
            arr = (ctypes.c_void_p * len(planes))(*[p.data_ptr() for p, _ in planes])
            lvs = (ctypes.c_int * len(planes))(*[lv for _, lv in planes])
            ws = _lib.workspace(ws_bytes, h.device)
            dq = torch.empty(tile.B * r * r, c2, dtype=torch.float32, device=h.device)
            _lib.call("t2h_sample_bwd_from_sums", ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(lvs, ctypes.c_void_p), len(planes),
                      _lib.ptr(tile.cell), _lib.ptr(h), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N,
                      tile.nbits, level, c2, _lib.ptr(dq), _lib.ptr(ws), ws_bytes, _lib.stream(),
                      nbytes=4 * c2 * tile.n_points + 12 * tile.n_points + 4 * dq.numel() + sum(4 * p.numel() for p, _ in planes),
                      tag=f"t2h_sample_bwd_from_sums[C={c2},r={r}]")
            return dq, None, None, None
        dh = _gather(tile, grads, ctx.levels, c2, mask=h)
        dq = ops._sample_bwd(tile, dh, r, c2, None).reshape(tile.B * r * r, c2)
        return dq, None, None, None


class _PointSums(torch.autograd.Function):
    """Per-cell sums of a per-point feature tensor [N, C] at ``levels`` (the last point-wise level's c, or the trunk's)."""

    @staticmethod
    def forward(ctx, rows, tile, levels):
        rows = rows.contiguous()
        ctx.tile, ctx.levels, ctx.c = tile, tuple(levels), rows.shape[1]
        return tuple(cell_sums(tile, rows, list(levels)))

    @staticmethod
    def backward(ctx, *grads):
        return _gather(ctx.tile, grads, ctx.levels, ctx.c), None, None


# ------------------------------------------------------------------------------------------------ grid-side products
class _SumMatmulNN(torch.autograd.Function):
    """sum_j x_j @ A[off_j : off_j + K_j] for row blocks x_j [m, K_j] and ONE stacked matrix A [sum K_j, n] (the composed maps
    of all sources, stacked by rows): accumulated by the GEMM epilogues -- no concat of the x_j, no elementwise adds, and the
    gradient of A is written block by block into one buffer.  args = (A, x_0, x_1, ...)."""

    @staticmethod
    def forward(ctx, a_all, *xs):
        a_all = a_all.contiguous()
        xs = [x.contiguous() for x in xs]
        out = torch.empty(xs[0].shape[0], a_all.shape[1], dtype=torch.float32, device=a_all.device)
        off = 0
        for j, x in enumerate(xs):
            mlp.linear_dgrad_(x, a_all[off:off + x.shape[1]], out, accumulate=j > 0)        # "dx = dy w" is x @ A_j
            off += x.shape[1]
        if off != a_all.shape[0]:
            raise ValueError(f"_SumMatmulNN: the blocks cover {off} of A's {a_all.shape[0]} rows")
        ctx.save_for_backward(a_all, *xs)
        return out

    @staticmethod
    def backward(ctx, g):
        a_all, *xs = ctx.saved_tensors
        g = g.contiguous()
        da = torch.empty_like(a_all) if ctx.needs_input_grad[0] else None
        dxs, off = [], 0
        for j, x in enumerate(xs):
            k = x.shape[1]
            blk = a_all[off:off + k]
            dxs.append(mlp.linear_fwd_(g, blk, None, torch.empty_like(x)) if ctx.needs_input_grad[1 + j] else None)   # g @ A_j^T
            if da is not None:
                mlp.linear_wgrad_(x, g, da[off:off + k], None)                               # x_j^T g
            off += k
        return (da, *dxs)


class _ComposeStack(torch.autograd.Function):
    """A_k = [A_{k-1} Wc_k^T ; W1_k^T]: the maps of all earlier sources composed with this level's fc_c in ONE product, and the
    new source's map (fc_comm.2, transposed) appended -- written straight into the stacked matrix (no cat, no transposed copy
    outside).  ``a_prev`` None: the base tensor's map before its first fc_c is the identity, so its block is Wc_k^T itself."""

    @staticmethod
    def forward(ctx, a_prev, wc, w1):
        ck, cprev = wc.shape
        k2 = w1.shape[1]
        rows_prev = cprev if a_prev is None else a_prev.shape[0]
        out = torch.empty(rows_prev + k2, ck, dtype=torch.float32, device=wc.device)
        wc_c, w1_c = wc.contiguous(), w1.contiguous()
        if a_prev is None:
            _transpose(wc_c, out[:rows_prev])                                  # [C_k, C_prev] -> [C_prev, C_k]
        else:
            a_prev = a_prev.contiguous()
            mlp.linear_fwd_(a_prev, wc_c, None, out[:rows_prev])               # A_{k-1} Wc_k^T
        _transpose(w1_c, out[rows_prev:])                                      # [C_k, 2 C_k] -> [2 C_k, C_k]
        ctx.has_prev = a_prev is not None
        ctx.rows_prev = rows_prev
        ctx.save_for_backward(a_prev if a_prev is not None else wc_c, wc_c)
        return out

    @staticmethod
    def backward(ctx, g):
        a_prev, wc = ctx.saved_tensors
        g = g.contiguous()
        rp = ctx.rows_prev
        gp, gn = g[:rp], g[rp:]
        da = dwc = None
        if ctx.has_prev:
            if ctx.needs_input_grad[0]:
                da = mlp.linear_dgrad_(gp, wc, torch.empty_like(a_prev))       # gp Wc
            dwc = torch.empty_like(wc)
            mlp.linear_wgrad_(gp, a_prev, dwc, None)                           # gp^T A_{k-1}
        else:
            dwc = torch.empty_like(wc)
            _transpose(gp, dwc)
        dw1 = torch.empty(gn.shape[1], gn.shape[0], dtype=torch.float32, device=g.device)
        _transpose(gn, dw1)
        return da, dwc, dw1


def _transpose(src, dst):
    """dst [n, m] = src [m, n]^T, both contiguous row-major (the [B, C, P] -> [B, P, C] layout kernel with B = 1)."""
    m, n = src.shape
    _lib.call("t2h_nchw_to_nhwc", _lib.ptr(src), 1, m, n, _lib.ptr(dst), _lib.stream(), nbytes=8 * m * n)


class _MeanBias(torch.autograd.Function):
    """raster = acc / max(count, 1) + [count > 0] * const: scatter_mean's division and empty-cell rule on the product of the
    per-cell sums, plus the composed bias -- one launch (t2h_mean_bias_fwd) instead of three elementwise ones."""

    @staticmethod
    def forward(ctx, acc, const, cnt):
        acc, const = acc.contiguous(), const.contiguous()
        out = torch.empty_like(acc)
        p, c = acc.shape
        _lib.call("t2h_mean_bias_fwd", _lib.ptr(acc), _lib.ptr(cnt), _lib.ptr(const), p, c, _lib.ptr(out), _lib.stream(),
                  nbytes=8 * p * c + 4 * p)
        ctx.save_for_backward(cnt)
        ctx.const_shape = const.shape
        return out

    @staticmethod
    def backward(ctx, g):
        (cnt,) = ctx.saved_tensors
        g = g.contiguous()
        p, c = g.shape
        dacc = torch.empty_like(g) if ctx.needs_input_grad[0] else None
        dconst = torch.empty(ctx.const_shape, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[1] else None
        ws_bytes = _lib.load().t2h_mean_bias_bwd_workspace_bytes(p, c)
        ws = _lib.workspace(ws_bytes, g.device)
        _lib.call("t2h_mean_bias_bwd", _lib.ptr(g), _lib.ptr(cnt), p, c, None if dacc is None else _lib.ptr(dacc),
                  None if dconst is None else _lib.ptr(dconst), _lib.ptr(ws), ws_bytes, _lib.stream(), nbytes=8 * p * c + 4 * p)
        return dacc, dconst, None


# ------------------------------------------------------------------------------------------------ state
def counts(tile, level):
    """Points per cell of ALTO level ``level`` as a [B r r] float column in plane (row-major) order, cached on the tile."""
    cache = tile.__dict__.setdefault("_cell_counts", {})
    if level not in cache:
        r = tile.R >> level
        cnt = torch.empty(tile.B * r * r, dtype=torch.float32, device=tile.device)
        _lib.call("t2h_cell_counts", _lib.ptr(tile.off0), tile.B, tile.nbits, level, _lib.ptr(cnt), _lib.stream(),
                  nbytes=12 * cnt.numel())
        cache[level] = cnt
    return cache[level]


class Deferred:
    """c_k = sum_j src_j A_j + const, never materialised.  ``sums``: per source {level: [B r r, K_j] per-cell sums};
    ``a_all`` [sum K_j, C_k]: the maps A_j = (Wc_k ... Wc_{j+1} W1_j)^T of all sources stacked by rows, so that one product
    per level composes them all with the level's fc_c (None until the first level: the base tensor's map is the identity)."""

    def __init__(self, tile, later_levels, base_rows):
        self.tile = tile
        levels = tuple(sorted(set(later_levels)))
        self.sums = [dict(zip(levels, _PointSums.apply(base_rows, tile, levels)))]
        self.a_all = None
        self.const = None                                 # [1, C_k] row or None (zero)

    def advance(self, q_rows, r, later_levels, fc_b, fc_c):
        """One level: compose the maps with this level's fc_c, add the level's hidden activations as a source (their fc_comm.2
        is the new source's map), return the level's raster rows [B r r, C_k].  ``later_levels`` = ALTO levels of this and
        every later exchange (what the new source will be rasterised at)."""
        tile = self.tile
        wc, bc, w1, b1 = fc_c.weight, fc_c.bias, fc_b.weight, fc_b.bias
        self.a_all = _ComposeStack.apply(self.a_all, wc, w1)
        if self.const is None:
            self.const = (bc + b1).reshape(1, -1)
        else:
            self.const = mlp.linear(self.const, wc, bc) + b1.reshape(1, -1)   # const_k = const_{k-1} Wc_k^T + bc_k + b1_k
        levels = tuple(sorted(set(later_levels)))
        self.sums.append(dict(zip(levels, _HiddenSums.apply(q_rows, tile, r, levels))))
        lv = tile.level(r)
        acc = _SumMatmulNN.apply(self.a_all, *[s[lv] for s in self.sums])
        return _MeanBias.apply(acc, self.const, counts(tile, lv))             # scatter_mean: mean per cell, 0 if empty


def applicable(tile, r: int, c: int) -> bool:
    return DEFER_MIN_CHANNELS > 0 and c >= DEFER_MIN_CHANNELS and mlp.grid_first_applicable(tile, r, c)
