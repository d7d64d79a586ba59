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

DEFER_MIN_CHANNELS = int(os.environ.get("T2H_DEFER_MIN_CHANNELS", "128"))
FUSED_SAMPLE_BWD = os.environ.get("T2H_FUSED_SAMPLE_BWD", "1") != "0"      # A/B: 0 = separate gather + sample adjoint
CELLS_MFMA = os.environ.get("T2H_CELLS_MFMA", "1") != "0"                  # (mirrors the library's switch: the bit mask needs it)
SIGN_BITS = os.environ.get("T2H_SIGN_BITS", "1") != "0"                    # A/B: 0 = keep the hidden activations for the mask
ON_CHIP_HIDDEN = os.environ.get("T2H_ON_CHIP_HIDDEN", "1") != "0"          # A/B: 0 = sample kernel + per-cell sum kernel
CELL_ORDER = os.environ.get("T2H_CELL_ORDER", "1") != "0"                  # A/B: 0 = the on-chip walks start their cells in Morton order
ON_CHIP_MIN_PTS_PER_CELL = float(os.environ.get("T2H_ON_CHIP_MIN_PTS", "8"))   # the walk is sequential inside a cell


# ------------------------------------------------------------------------------------------------ per-cell sums
# Data layout: per resolution ONE matrix S_lv [B r r, K_total] whose column blocks are the per-cell sums of the sources in the
# order they appear (base features, h of the first deferred level, h of the second, ...).  A level's scatter_mean is then a single
# product over the leading K_k columns (all sources that exist at that level), its backward a single product into the gradient
# matrix dS_lv of the same layout, and a source's kernels address "their" block through a row stride.
def _segsum_into(tile, rows, level, dst):
    """Per-cell sums of ``rows`` [N, C] at ALTO level ``level`` into the [B r r, C] column block ``dst`` (row stride dst.stride(0))."""
    n, c = rows.shape
    r = tile.R >> level
    ws_bytes = _lib.ws_bytes("t2h_segmean_workspace_bytes", tile.B, tile.N, tile.nbits, level, c)
    ws = _lib.workspace(ws_bytes, rows.device)
    _lib.call("t2h_segsum_fwd", _lib.ptr(rows), _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits, level, c, dst.data_ptr(),
              dst.stride(0), _lib.ptr(ws), ws_bytes, _lib.stream(), nbytes=4 * c * n + 4 * n + 4 * c * tile.B * r * r,
              tag=_lib.timing() and f"t2h_segsum_fwd[C={c},r={r}]")


def _sumpool_into(tile, fine, level_fine, coarse):
    """Sums at ``level_fine`` (column block ``fine``) -> level_fine + 1 (column block ``coarse``): a cell is the union of its four
    children."""
    c = fine.shape[1]
    r = tile.R >> level_fine
    _lib.call("t2h_plane_sumpool2x2", fine.data_ptr(), fine.stride(0), tile.B, r, c, coarse.data_ptr(), coarse.stride(0),
              _lib.stream(), nbytes=5 * c * tile.B * r * r, tag=_lib.timing() and f"t2h_plane_sumpool2x2[C={c}]")


def _plane_args(planes):
    """ctypes arrays (pointers, levels, row strides) of a list of (column block, level)."""
    arr = (ctypes.c_void_p * len(planes))(*[p.data_ptr() for p, _ in planes])
    lvs = (ctypes.c_int * len(planes))(*[lv for _, lv in planes])
    lds = (ctypes.c_int * len(planes))(*[p.stride(0) for p, _ in planes])
    return ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(lvs, ctypes.c_void_p), ctypes.cast(lds, ctypes.c_void_p)


def _gather(tile, planes, c, mask=None):
    """d rows [N, C] = (mask > 0 ?) sum over ``planes`` [(column block, level)] of block[cell_level(n)]: the adjoint of the sums."""
    out = torch.empty(tile.n_points, c, dtype=torch.float32, device=tile.device)
    arr, lvs, lds = _plane_args(planes)
    _lib.call("t2h_segsum_bwd_multi", arr, lvs, lds, len(planes), _lib.ptr(tile.cell), tile.B, tile.N, tile.nbits, c,
              None if mask is None else _lib.ptr(mask), None, _lib.ptr(out), _lib.stream(),
              nbytes=4 * c * tile.n_points * (2 if mask is not None else 1) + 4 * tile.n_points
              + sum(4 * c * p.shape[0] for p, _ in planes), tag=_lib.timing() and f"t2h_segsum_bwd_multi[C={c},n={len(planes)}]")
    return out


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


# ------------------------------------------------------------------------------------------------ grid-side products
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
        ctx.wc_param = wc                                  # (the parameter itself: mlp._wgrad looks at its .grad)
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
            dwc, _ = mlp._wgrad(gp, a_prev, ctx.wc_param, None)                # gp^T A_{k-1} (straight into wc.grad where allowed)
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


class _DeferredLevel(torch.autograd.Function):
    """One exchange level of the deferred form.  Forward: h = relu(sample(Q)) [N, 2C] (kept inside), its per-cell sums into the
    sum matrices of every resolution it will be needed at, then raster = (S_lv[:, :K] @ A) / count + [count > 0] const -- ONE
    product over all sources.  Backward: the product's two gradients (A; the sums, accumulated into the gradient matrices dS),
    then -- every later level has already added its share, autograd runs them first -- this level's own source: gather + mask
    inside the sample adjoint -> dQ.  The first deferred level also owns the base features' sums and returns their gradient."""

    @staticmethod
    def forward(ctx, q_rows, a_all, const, base_rows, state, idx, r):
        tile = state.tile
        q_rows, a_all, const = q_rows.contiguous(), a_all.contiguous(), const.contiguous()
        c2 = q_rows.shape[1]
        if idx == 0:
            state.write_sums(0, base_rows.contiguous())
        # forward only (model.eval() under no_grad, generator.py:142-147): nothing will read the sign pattern -- do not write it
        need_bwd = any(ctx.needs_input_grad[:4])
        bits_ok = (SIGN_BITS and c2 % 256 == 0 and FUSED_SAMPLE_BWD and CELLS_MFMA and tile.n_points > 0 and
                   _lib.ws_bytes("t2h_sample_bwd_workspace_bytes", tile.B, tile.N, tile.nbits, tile.level(r), c2) > 0)
        on_chip = (bits_ok if need_bwd else (c2 % 256 == 0 and tile.n_points > 0)) and ON_CHIP_HIDDEN
        if on_chip and tile.n_points >= ON_CHIP_MIN_PTS_PER_CELL * tile.B * r * r:
            # coarse sampling level: interpolation, ReLU, sign bits and the per-cell sums of the finest needed resolution in one
            # pass over the cells -- the hidden activations are never written to memory (t2h_sample_relu_cellsums)
            bits = None
            if need_bwd:
                bits = torch.empty(tile.n_points * (c2 // 256) * 4, dtype=torch.int64, device=q_rows.device)
            lo, hi = state.off[idx + 1], state.off[idx + 2]
            levels = state.needed(idx + 1)
            finest = state._matrix(state.S, levels[0])[:, lo:hi]
            # the next coarser needed resolution comes out of the same pass (the four children of a cell are in registers)
            second = None
            if len(levels) > 1 and levels[1] == levels[0] + 1 and levels[0] < tile.level(r):
                second = state._matrix(state.S, levels[1])[:, lo:hi]
            order = tile.cell_order(tile.level(r)) if CELL_ORDER else None
            _lib.call("t2h_sample_relu_cellsums_ordered", _lib.ptr(q_rows), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B,
                      tile.N, tile.nbits, tile.level(r), levels[0], c2, finest.data_ptr(), finest.stride(0),
                      None if second is None else second.data_ptr(), 0 if second is None else second.stride(0),
                      None if bits is None else _lib.ptr(bits), None if order is None else _lib.ptr(order), _lib.stream(),
                      nbytes=4 * q_rows.numel() + 12 * tile.n_points + 4 * c2 * finest.shape[0]
                      + (0 if bits is None else (c2 // 8) * tile.n_points) + (0 if second is None else 4 * c2 * second.shape[0]),
                      tag=_lib.timing() and f"t2h_sample_relu_cellsums[C={c2},r={r}]")
            state.pool_down(idx + 1, start=1 if second is not None else 0)
            return _DeferredLevel._finish(ctx, state, idx, r, a_all, const, bits, True)
        h = torch.empty(tile.n_points, c2, dtype=torch.float32, device=q_rows.device)
        # the backward needs only the sign pattern of h: where its fused form will run, keep 1 bit per element (written by the
        # sample kernel's ballots) and let h go after the per-cell sums -- 1/32 of the bytes to keep and to re-read
        bits = None
        if bits_ok and need_bwd:
            bits = torch.empty(tile.n_points * (c2 // 256) * 4, dtype=torch.int64, device=q_rows.device)
        _lib.call("t2h_sample_fwd_relu", _lib.ptr(q_rows), _lib.ptr(tile.pts), tile.dim, tile.B, tile.N, r, c2, _lib.ptr(h),
                  None if bits is None else _lib.ptr(bits), _lib.stream(),
                  nbytes=4 * c2 * tile.n_points + 8 * tile.n_points + 4 * q_rows.numel() + (c2 // 8) * tile.n_points * (bits is not None),
                  tag=_lib.timing() and f"t2h_sample_fwd_relu[C={c2},r={r}]")
        state.write_sums(idx + 1, h)
        return _DeferredLevel._finish(ctx, state, idx, r, a_all, const, h if bits is None else bits, bits is not None)

    @staticmethod
    def _finish(ctx, state, idx, r, a_all, const, saved_mask, mask_is_bits):
        """The level's product over all sources' sums + the mean / empty-cell / bias epilogue; saves what the backward needs."""
        tile = state.tile
        lv = state.levels_seq[idx]
        k = state.off[idx + 2]                                             # columns of all sources that exist at this level
        if a_all.shape[0] != k:
            raise RuntimeError(f"deferred level {idx}: {a_all.shape[0]} composed rows for {k} source columns")
        x = state.S[lv][:, :k]
        acc = torch.empty(x.shape[0], a_all.shape[1], dtype=torch.float32, device=x.device)
        mlp.linear_dgrad_(x, a_all, acc, bx3=True)                         # "dx = dy w" is exactly x @ A
        cnt = counts(tile, lv)
        out = torch.empty_like(acc)
        _lib.call("t2h_mean_bias_fwd", _lib.ptr(acc), _lib.ptr(cnt), _lib.ptr(const), acc.shape[0], acc.shape[1], _lib.ptr(out),
                  _lib.stream(), nbytes=8 * acc.numel() + 4 * acc.shape[0])
        ctx.state, ctx.idx, ctx.r, ctx.has_base = state, idx, r, idx == 0
        ctx.const_shape = const.shape
        ctx.mask_is_bits = mask_is_bits
        ctx.save_for_backward(saved_mask, a_all)
        # second output: the composed maps themselves for the next level's compose product -- its gradient then arrives HERE and
        # this level's own share is accumulated into it by the weight-gradient kernel (no autograd add over [K, C])
        return out, a_all.as_strided(a_all.size(), a_all.stride(), a_all.storage_offset())

    @staticmethod
    def backward(ctx, g, ga_thru):
        h, a_all = ctx.saved_tensors
        state, idx, r = ctx.state, ctx.idx, ctx.r
        tile = state.tile
        lv = state.levels_seq[idx]
        k = state.off[idx + 2]
        g = g.contiguous()
        p, c = g.shape
        cnt = counts(tile, lv)
        # through the mean / bias epilogue
        dacc = torch.empty_like(g)
        cached = state.cache is not None and not ctx.needs_input_grad[1]
        dconst = torch.empty(ctx.const_shape, dtype=torch.float32, device=g.device) if (ctx.needs_input_grad[2] or cached) else None
        ws_bytes = _lib.ws_bytes("t2h_mean_bias_bwd_workspace_bytes", p, c)
        ws = _lib.workspace(ws_bytes, g.device)
        _lib.call("t2h_mean_bias_bwd", _lib.ptr(g), _lib.ptr(cnt), p, c, _lib.ptr(dacc), None if dconst is None else _lib.ptr(dconst),
                  _lib.ptr(ws), ws_bytes, _lib.stream(), nbytes=8 * p * c + 4 * p)
        x = state.S[lv][:, :k]
        da = ga_thru
        if state.cache is not None and not ctx.needs_input_grad[1]:
            # maps from the trainer's ComposeCache: this tile's share goes onto the persistent sums (zeroed by flush())
            e = state.cache.levels[idx]
            e["ready"].wait()
            mlp.linear_wgrad_(x, dacc, e["ga"], None, accumulate=True, defer=True)   # += x^T dacc (read at flush(): after the pass)
            e["gconst"].add_(dconst)
            dconst = None
            state.cache.pending = True
        elif ctx.needs_input_grad[1]:
            if ga_thru is not None and ga_thru.is_contiguous() and ga_thru.dtype == torch.float32 and mlp.sole_owner(ga_thru):
                mlp.linear_wgrad_(x, dacc, ga_thru, None, accumulate=True)  # x^T dacc, onto the next level's share
            else:
                da = torch.empty_like(a_all)
                mlp.linear_wgrad_(x, dacc, da, None)                       # x^T dacc
                if ga_thru is not None:
                    da = da + ga_thru
        # into the gradient matrix of this resolution: the leading k columns; a later level of the same resolution (run before,
        # more columns) has initialised them already -> accumulate
        dS, seen = state.grad_matrix(lv)
        if seen and seen < k:
            raise RuntimeError("deferred backward out of order")
        mlp.linear_fwd_(dacc, a_all, None, dS[:, :k], accumulate=bool(seen), bx3=True)   # dacc @ A^T
        state.dS_cols[lv] = max(seen, k)
        # this level's own source: every level that used its sums has contributed by now
        lo, hi = state.off[idx + 1], state.off[idx + 2]
        planes = [(state.dS[l][:, lo:hi], l) for l in state.needed(idx + 1)]
        dq = None
        if ctx.needs_input_grad[0]:
            dq = state.hidden_grad(planes, h, r, hi - lo, ctx.mask_is_bits)
        dbase = None
        if ctx.has_base and ctx.needs_input_grad[3]:
            lo0, hi0 = state.off[0], state.off[1]
            dbase = _gather(tile, [(state.dS[l][:, lo0:hi0], l) for l in state.needed(0)], hi0 - lo0)
        return dq, da, dconst, dbase, None, None, None


class ComposeCache:
    """The composed maps ``a_all`` / ``const`` of the deferred levels depend on the WEIGHTS only, and their backward is linear in
    the gradient they receive -- so between two optimizer steps (the reference accumulates 64 tiles per step, trainer.py:72-89)
    they are computed once, every tile's backward adds its ``d a_all`` / ``d const`` into persistent buffers (in the weight-
    gradient kernel: no extra pass), and ``flush()`` runs the compose chain's backward ONCE on the sums, adding into the
    parameters' ``.grad`` like any other backward.  Per tile this removes the compose products, their two gradients each, the
    bias chain, the weight transposes and the parameter-gradient adds of every deferred level (~0.5 ms of a 12 ms step).

    Owner: ``Trainer`` (``optimizer_boundary`` flushes before the gradient all-reduce and calls ``refresh()`` after the optimizer
    step); a model used without it never sees a cache and keeps plain autograd semantics.  Buffers are updated in place, so a
    captured hipGraph keeps reading current values."""

    def __init__(self):
        self.levels = []            # per deferred level: dict(params, versions, a_all, const, ga, gconst)
        self.pending = False

    @staticmethod
    def _versions(params):
        return tuple(p._version for p in params)

    def _compute(self, idx, params):
        wc, bc, w1, b1 = params
        prev = self.levels[idx - 1] if idx > 0 else None
        with torch.no_grad():
            a_all = _ComposeStack.apply(None if prev is None else prev["a_all"], wc, w1)
            if prev is None:
                const = (bc + b1).reshape(1, -1)
            else:
                const = mlp.linear(prev["const"], wc, bc) + b1.reshape(1, -1)
        return a_all, const

    def get(self, idx, params):
        """(a_all, const) of deferred level ``idx`` for the current weights (levels are asked for in order within a forward)."""
        params = tuple(params)
        if idx > len(self.levels):
            raise RuntimeError("ComposeCache: levels must be requested in order")
        ver = self._versions(params)
        if idx == len(self.levels):
            a_all, const = self._compute(idx, params)
            self.levels.append({"params": params, "versions": ver, "a_all": a_all, "const": const,
                                "ga": torch.zeros_like(a_all), "gconst": torch.zeros_like(const), "ready": _lib.Ready()})
            self.levels[idx]["ready"].mark()
        else:
            e = self.levels[idx]
            if any(p is not q for p, q in zip(e["params"], params)):
                raise RuntimeError("ComposeCache: a different network uses this cache")
            if e["versions"] != ver:
                if self.pending:
                    raise RuntimeError("ComposeCache: weights changed with unflushed gradients (call flush() before the optimizer step)")
                a_all, const = self._compute(idx, params)
                e["a_all"].copy_(a_all)
                e["const"].copy_(const)
                e["versions"] = ver
                e["ready"].mark()
            else:
                e["ready"].wait()                             # (computed on another stream, e.g. the other tile stream: that first)
        e = self.levels[idx]
        return e["a_all"], e["const"]

    def snapshot(self):
        """The accumulated gradients (for a caller that runs passes whose gradients must not count: hipGraph warm-up / capture)."""
        return [(e["ga"].clone(), e["gconst"].clone()) for e in self.levels], self.pending

    def restore(self, snap):
        saved, self.pending = snap
        for i, e in enumerate(self.levels):          # levels created after the snapshot start from zero
            if i < len(saved):
                e["ga"].copy_(saved[i][0])
                e["gconst"].copy_(saved[i][1])
            else:
                e["ga"].zero_()
                e["gconst"].zero_()

    def refresh(self):
        """Recompute every level's maps in place (after an optimizer step; needed explicitly only under hipGraph replay, where
        no Python runs per tile)."""
        for idx, e in enumerate(self.levels):
            e["versions"] = None
            self.get(idx, e["params"])

    def flush(self):
        """The compose chain's backward, once, on the gradients accumulated since the last flush."""
        if not self.levels:
            return
        outs, grads, a_prev, const = [], [], None, None
        with torch.enable_grad():
            for e in self.levels:
                wc, bc, w1, b1 = e["params"]
                a_all = _ComposeStack.apply(a_prev, wc, w1)
                const = (bc + b1).reshape(1, -1) if const is None else mlp.linear(const, wc, bc) + b1.reshape(1, -1)
                outs += [a_all, const]
                grads += [e["ga"], e["gconst"]]
                a_prev = a_all
            torch.autograd.backward(outs, [g.clone() for g in grads])
        for e in self.levels:
            e["ga"].zero_()
            e["gconst"].zero_()
        self.pending = False


class Deferred:
    """c_k = sum_j src_j A_j + const, never materialised.  Sources: 0 = the base per-point features (the last point-wise level's
    c), j >= 1 = the hidden activations of deferred level j - 1.  ``S[lv]`` / ``dS[lv]``: [B r r, K_total] matrices of the per-cell
    sums / their gradients, one column block per source.  ``a_all`` [K_k, C_k]: the maps A_j = (Wc_k ... Wc_{j+1} W1_j)^T of all
    sources stacked by rows in the same order, composed with each level's fc_c by one product (``_ComposeStack``)."""

    def __init__(self, tile, levels_seq, channels_seq, base_rows, cache=None):
        self.tile = tile
        self.cache = cache                                 # ComposeCache of the owning trainer, or None: plain autograd
        self.levels_seq = list(levels_seq)                 # ALTO level (resolution) of this and every later exchange
        ks = [base_rows.shape[1]] + [2 * c for c in channels_seq]
        self.off = [0]
        for kk in ks:
            self.off.append(self.off[-1] + kk)
        self.base = base_rows
        self.S, self.dS, self.dS_cols = {}, {}, {}
        self.a_all, self.const = None, None
        self.n_done = 0

    def needed(self, src):
        """ALTO levels (finest first) at which source ``src`` is rasterised: the resolutions of its own level and all later ones."""
        first = 0 if src == 0 else src - 1
        return sorted(set(self.levels_seq[first:]))

    def _matrix(self, store, lv):
        if lv not in store:
            r = self.tile.R >> lv
            store[lv] = torch.empty(self.tile.B * r * r, self.off[-1], dtype=torch.float32, device=self.tile.device)
        return store[lv]

    def grad_matrix(self, lv):
        return self._matrix(self.dS, lv), self.dS_cols.get(lv, 0)

    def write_sums(self, src, rows):
        """Per-cell sums of ``rows`` [N, K_src] into the source's column block at every needed resolution: the finest from the
        rows (read once), the coarser ones by 2x2 pooling."""
        lo, hi = self.off[src], self.off[src + 1]
        levels = self.needed(src)
        _segsum_into(self.tile, rows, levels[0], self._matrix(self.S, levels[0])[:, lo:hi])
        self.pool_down(src)

    def pool_down(self, src, start=0):
        """The source's sums at its coarser needed resolutions from the (already written) finest one, by 2x2 pooling;
        ``start``: index of the coarsest needed resolution that is written already."""
        tile, (lo, hi) = self.tile, (self.off[src], self.off[src + 1])
        levels = self.needed(src)
        cur_level = levels[start]
        cur = self._matrix(self.S, cur_level)[:, lo:hi]
        for lv in levels[start + 1:]:
            while cur_level < lv:
                nxt_level = cur_level + 1
                if nxt_level in levels:
                    nxt = self._matrix(self.S, nxt_level)[:, lo:hi]
                else:                                       # a resolution no level uses: a compact temporary on the way down
                    rn = tile.R >> nxt_level
                    nxt = torch.empty(tile.B * rn * rn, hi - lo, dtype=torch.float32, device=tile.device)
                _sumpool_into(tile, cur, cur_level, nxt)
                cur, cur_level = nxt, nxt_level

    def hidden_grad(self, planes, h, r, c2, mask_is_bits=False):
        """dQ [B r r, 2C] = S^T ((h > 0) * sum_l dS_l[cell_l(.)]): gather + ReLU mask inside the sample adjoint where the level
        takes the per-cell partials, else as two passes.  ``h``: the hidden activations, or their packed sign bits."""
        tile = self.tile
        level = tile.level(r)
        ws_bytes = _lib.ws_bytes("t2h_sample_bwd_workspace_bytes", tile.B, tile.N, tile.nbits, level, c2)
        if mask_is_bits or (FUSED_SAMPLE_BWD and ws_bytes > 0 and c2 % 4 == 0):
            arr, lvs, lds = _plane_args(planes)
            ws = _lib.workspace(ws_bytes, h.device)
            dq = torch.empty(tile.B * r * r, c2, dtype=torch.float32, device=h.device)
            order = tile.cell_order(level) if (CELL_ORDER and mask_is_bits) else None
            _lib.call("t2h_sample_bwd_from_sums_ordered", arr, lvs, lds, len(planes), _lib.ptr(tile.cell), _lib.ptr(h),
                      1 if mask_is_bits else 0, _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits,
                      level, c2, _lib.ptr(dq), _lib.ptr(ws), ws_bytes, None if order is None else _lib.ptr(order), _lib.stream(),
                      nbytes=(c2 // 8 if mask_is_bits else 4 * c2) * tile.n_points + 12 * tile.n_points + 4 * dq.numel()
                      + sum(4 * c2 * p.shape[0] for p, _ in planes), tag=_lib.timing() and f"t2h_sample_bwd_from_sums[C={c2},r={r}]")
            return dq
        dh = _gather(tile, planes, c2, mask=h)
        return ops._sample_bwd(tile, dh, r, c2, None).reshape(tile.B * r * r, c2)

    def advance(self, q_rows, r, fc_b, fc_c):
        """One level: compose the maps with this level's fc_c and append fc_comm.2's as the new source's, then the level's raster
        rows [B r r, C_k] (``_DeferredLevel``)."""
        idx = self.n_done
        if self.tile.level(r) != self.levels_seq[idx]:
            raise RuntimeError("deferred levels out of order")
        wc, bc, w1, b1 = fc_c.weight, fc_c.bias, fc_b.weight, fc_b.bias
        if self.cache is not None:                          # weights-only work shared by all tiles of an optimizer step
            a_all, const = self.cache.get(idx, (wc, bc, w1, b1))
            raster, _ = _DeferredLevel.apply(q_rows, a_all, const, self.base if idx == 0 else None, self, idx, r)
            self.n_done += 1
            return raster
        self.a_all = _ComposeStack.apply(self.a_all, wc, w1)
        if self.const is None:
            self.const = (bc + b1).reshape(1, -1)
        else:
            self.const = mlp.linear(self.const, wc, bc) + b1.reshape(1, -1)   # const_k = const_{k-1} Wc_k^T + bc_k + b1_k
        raster, self.a_all = _DeferredLevel.apply(q_rows, self.a_all, self.const, self.base if idx == 0 else None, self, idx, r)
        self.n_done += 1
        if self.n_done == len(self.levels_seq):
            # last level: nobody composes further.  (Also breaks the cycle state -> a_all -> its autograd node -> state, which
            # would leave the sum matrices of every step to the cyclic garbage collector.)
            self.a_all = self.const = None
        return raster


def applicable(tile, r: int, c: int) -> bool:
    return DEFER_MIN_CHANNELS > 0 and c >= DEFER_MIN_CHANNELS and mlp.grid_first_applicable(tile, r, c)
