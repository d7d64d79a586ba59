"""Autograd operators over the C ABI (include/t2h.h).  Point features are ``[B*N, C]`` rows in the tile's
cell-sorted order; planes cross this boundary as ``[B, C, r, r]`` tensors (NCHW or channels_last).

Reference seam: ``pool_local`` (pointnet.py:92-99), ``generate_plane_features``
(pointnet.py:101-111, alto.py:76-88), ``sample_plane_feature`` (alto.py:90-95), ``F.interpolate``
(pixel.py:107).  No op here has a torch/CPU fallback.
"""
import atexit
import os

import torch

from . import _lib
from .tile import TileIndex


def _f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: float32 expected, got {t.dtype}")
    return t


# --------------------------------------------------------------------------------------- layout glue
def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W] (any strides) -> contiguous [B,H,W,C]; free when x is already channels_last."""
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v
    x = x.contiguous()
    _lib.require_device(x, what="to_nhwc")
    b, c, h, w = x.shape
    out = torch.empty(b, h, w, c, dtype=x.dtype, device=x.device)
    _lib.call("t2h_nchw_to_nhwc", _lib.ptr(x), b, c, h * w, _lib.ptr(out), _lib.stream(), nbytes=8 * x.numel())
    return out


def from_nhwc(x_nhwc: torch.Tensor, channels_last: bool) -> torch.Tensor:
    """contiguous [B,H,W,C] -> [B,C,H,W]; a view if channels_last, else an NCHW-contiguous copy."""
    if channels_last:
        return x_nhwc.permute(0, 3, 1, 2)
    _lib.require_device(x_nhwc, what="from_nhwc")
    b, h, w, c = x_nhwc.shape
    out = torch.empty(b, c, h, w, dtype=x_nhwc.dtype, device=x_nhwc.device)
    _lib.call("t2h_nhwc_to_nchw", _lib.ptr(x_nhwc), b, c, h * w, _lib.ptr(out), _lib.stream(),
              nbytes=8 * x_nhwc.numel())
    return out


def _alias(x: torch.Tensor) -> torch.Tensor:
    """A second tensor object on the same memory with EXACTLY the same strides (``view_as`` rewrites the stride of size-1
    dims, after which ``torch.cat`` no longer recognises a B = 1 channels_last plane and emits NCHW)."""
    return x.as_strided(x.size(), x.stride(), x.storage_offset())


def _is_channels_last(x: torch.Tensor) -> bool:
    return x.permute(0, 2, 3, 1).is_contiguous() and not x.is_contiguous()


# --------------------------------------------------------------------------------------- pool_local
class _PoolMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, tile: TileIndex):
        feat = _f32(feat, "pool_max").contiguous()
        _lib.require_device(feat, what="pool_max")
        n, c = feat.shape
        if n != tile.n_points:
            raise ValueError(f"pool_max: {n} feature rows for a tile of {tile.n_points} points")
        pooled = torch.empty_like(feat)
        winner = torch.empty(n, _lib.load().t2h_pool_winner_stride(c), dtype=torch.uint8, device=feat.device)
        _lib.call("t2h_pool_max_fwd", _lib.ptr(feat), c, _lib.ptr(tile.off0), tile.B, tile.nbits, c, _lib.ptr(pooled), c,
                  _lib.ptr(winner), _lib.stream(), nbytes=8 * c * n + 4 * n)
        ctx.tile, ctx.c = tile, c
        ctx.save_for_backward(winner)
        return pooled

    @staticmethod
    def backward(ctx, gpooled):
        (winner,) = ctx.saved_tensors
        tile = ctx.tile
        gpooled = gpooled.contiguous()
        gfeat = torch.empty_like(gpooled)
        if ctx.c % 4 == 0 and ctx.c <= 64:          # row-balanced backward (14 vs 26 us at the bench shape)
            _lib.call("t2h_pool_rows_bwd", _lib.ptr(gpooled), ctx.c, _lib.ptr(winner), _lib.ptr(tile.cell), _lib.ptr(tile.off0),
                      tile.n_points, ctx.c, 0, _lib.ptr(gfeat), ctx.c, _lib.stream(),
                      nbytes=8 * ctx.c * tile.n_points + 4 * tile.n_points, tag="t2h_pool_max_bwd")
        else:
            _lib.call("t2h_pool_max_bwd", _lib.ptr(gpooled), ctx.c, _lib.ptr(winner), _lib.ptr(tile.off0), tile.B,
                      tile.nbits, ctx.c, 0, _lib.ptr(gfeat), ctx.c, _lib.stream(),
                      nbytes=8 * ctx.c * tile.n_points + 4 * tile.n_points)
        return gfeat, None


def pool_max(tile: TileIndex, feat: torch.Tensor) -> torch.Tensor:
    """Per-cell max at the finest level, broadcast back to every point (pointnet.py:92-99)."""
    return _PoolMax.apply(feat, tile)


class _PoolMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, tile: TileIndex):
        feat = _f32(feat, "pool_mean").contiguous()
        _lib.require_device(feat, what="pool_mean")
        n, c = feat.shape
        if n != tile.n_points:
            raise ValueError(f"pool_mean: {n} feature rows for a tile of {tile.n_points} points")
        pooled = torch.empty_like(feat)
        _lib.call("t2h_pool_mean", _lib.ptr(feat), c, _lib.ptr(tile.off0), tile.B, tile.nbits, c, 0, _lib.ptr(pooled), c,
                  _lib.stream(), nbytes=8 * c * n + 4 * n)
        ctx.tile, ctx.c = tile, c
        return pooled

    @staticmethod
    def backward(ctx, gpooled):
        gpooled = gpooled.contiguous()
        tile, c = ctx.tile, ctx.c
        gfeat = torch.empty_like(gpooled)
        _lib.call("t2h_pool_mean", _lib.ptr(gpooled), c, _lib.ptr(tile.off0), tile.B, tile.nbits, c, 0, _lib.ptr(gfeat), c,
                  _lib.stream(), nbytes=8 * c * tile.n_points + 4 * tile.n_points)
        return gfeat, None


def pool_mean(tile: TileIndex, feat: torch.Tensor) -> torch.Tensor:
    """Per-cell mean at the finest level, broadcast back to every point (pointnet.py:92-99 with scatter_type='mean')."""
    return _PoolMean.apply(feat, tile)


# --------------------------------------------------------------------------------------- scatter_mean -> plane
class _RasteriseMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, tile: TileIndex, level: int, channels_last: bool):
        feat = _f32(feat, "rasterise_mean").contiguous()
        _lib.require_device(feat, what="rasterise_mean")
        n, c = feat.shape
        if n != tile.n_points:
            raise ValueError(f"rasterise_mean: {n} feature rows for a tile of {tile.n_points} points")
        r = tile.R >> level
        plane = torch.empty(tile.B, r, r, c, dtype=torch.float32, device=feat.device)
        ws_bytes = _lib.ws_bytes("t2h_segmean_workspace_bytes", tile.B, tile.N, tile.nbits, level, c)
        ws = _lib.workspace(ws_bytes, feat.device)
        _lib.call("t2h_segmean_fwd", _lib.ptr(feat), _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits, level, c,
                  _lib.ptr(plane), _lib.ptr(ws), ws_bytes, _lib.stream(), nbytes=4 * c * n + 4 * n + 4 * plane.numel(), tag=_lib.timing() and f"t2h_segmean_fwd[C={c},r={r}]")
        ctx.tile, ctx.level, ctx.c = tile, level, c
        return from_nhwc(plane, channels_last)

    @staticmethod
    def backward(ctx, gplane):
        tile = ctx.tile
        g = to_nhwc(gplane)
        gfeat = torch.empty(tile.n_points, ctx.c, dtype=torch.float32, device=g.device)
        _lib.call("t2h_segmean_bwd", _lib.ptr(g), _lib.ptr(tile.cell), _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits,
                  ctx.level, ctx.c, _lib.ptr(gfeat), _lib.stream(),
                  nbytes=4 * g.numel() + 4 * tile.n_points + 4 * g.numel() // ctx.c + 4 * ctx.c * tile.n_points,
                  tag=_lib.timing() and f"t2h_segmean_bwd[C={ctx.c},r={g.shape[1]}]")
        return gfeat, None, None, None


class _RasteriseMeanThru(torch.autograd.Function):
    """``(rasterise_mean(feat), feat)``: the second output is ``feat`` itself for its OTHER consumer (the next level's
    ``fc_c``, alto.py:126-128 / 251-253).  Autograd would sum the two gradients of ``feat`` with an elementwise pass over
    an [N, C] tensor (0.4 ms per step over all levels); here the rasterisation's backward adds the other gradient while it
    writes its own (``t2h_segmean_bwd_add``) -- same sum, bit for bit."""

    @staticmethod
    def forward(ctx, feat, tile: TileIndex, level: int, channels_last: bool):
        plane = _RasteriseMean.forward(ctx, feat, tile, level, channels_last)
        return plane, _alias(feat)

    @staticmethod
    def backward(ctx, gplane, gthru):
        tile = ctx.tile
        n, c = tile.n_points, ctx.c
        if gplane is None:
            return gthru, None, None, None
        g = to_nhwc(gplane)
        addend = None
        if gthru is not None:
            addend = gthru.contiguous()
            _lib.require_device(addend, what="rasterise_mean_thru")
        gfeat = torch.empty(n, c, dtype=torch.float32, device=g.device)
        _lib.call("t2h_segmean_bwd_add", _lib.ptr(g), _lib.ptr(tile.cell), _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits,
                  ctx.level, c, None if addend is None else _lib.ptr(addend), _lib.ptr(gfeat), _lib.stream(),
                  nbytes=4 * g.numel() + 4 * n + 4 * g.numel() // c + 4 * c * n * (2 if addend is not None else 1),
                  tag=_lib.timing() and f"t2h_segmean_bwd[C={c},r={g.shape[1]}]")
        return gfeat, None, None, None


def rasterise_mean_thru(tile: TileIndex, feat: torch.Tensor, reso: int, channels_last: bool = False):
    """``(plane, feat)`` -- use the returned ``feat`` for every further consumer of the point features so that their
    gradient is summed inside the rasterisation's backward kernel."""
    return _RasteriseMeanThru.apply(feat, tile, tile.level(reso), channels_last)


def rasterise_mean(tile: TileIndex, feat: torch.Tensor, reso: int, channels_last: bool = False) -> torch.Tensor:
    """Per-cell mean of point features -> ``[B, C, reso, reso]``; empty cells are 0
    (generate_plane_features: pointnet.py:101-111; alto.py:76-88,187-197)."""
    return _RasteriseMean.apply(feat, tile, tile.level(reso), channels_last)


# --------------------------------------------------------------------------------------- grid_sample at points
# Sample backward through the cached transposed matrix (TileIndex.sample_adjoint) where a level holds few rows per pixel
# (N = 131072: r = 256 -- product 16-21 us against 60-80 us for the gather, build 47 us once for the level's three calls;
# at r = 128 the product is 57 us against 73 us and no longer pays for its build); above SAMPLE_ADJOINT_MAX_ROWS rows per
# pixel t2h_sample_bwd's gather / per-cell partials stay.  T2H_SAMPLE_ADJOINT=0 switches the path off (A/B).
SAMPLE_ADJOINT = os.environ.get("T2H_SAMPLE_ADJOINT", "1") != "0"
SAMPLE_ADJOINT_MAX_ROWS = float(os.environ.get("T2H_SAMPLE_ADJOINT_MAX_ROWS", "4"))


def _sample_bwd(tile, gout, r, c, addend):
    """gplane [B, r, r, C] = [addend +] (d sample / d plane)^T gout"""
    level = tile.level(r)
    gplane = torch.empty(tile.B, r, r, c, dtype=torch.float32, device=gout.device)
    nbytes = 4 * c * tile.n_points + 8 * tile.n_points + 4 * gplane.numel() * (2 if addend is not None else 1)
    if SAMPLE_ADJOINT and tile.n_points <= SAMPLE_ADJOINT_MAX_ROWS * tile.B * r * r and tile.n_points > 0:
        offsets, entries = tile.sample_adjoint(level)
        _lib.call("t2h_sample_bwd_adjoint", _lib.ptr(gout), _lib.ptr(offsets), _lib.ptr(entries), tile.B, tile.nbits, level, c,
                  None if addend is None else _lib.ptr(addend), _lib.ptr(gplane), _lib.stream(), nbytes=nbytes,
                  tag=_lib.timing() and f"t2h_sample_bwd[C={c},r={r}]")
        return gplane
    ws_bytes = _lib.ws_bytes("t2h_sample_bwd_workspace_bytes", tile.B, tile.N, tile.nbits, level, c)
    ws = _lib.workspace(ws_bytes, gout.device)
    _lib.call("t2h_sample_bwd_add", _lib.ptr(gout), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N,
              tile.nbits, level, c, None if addend is None else _lib.ptr(addend), _lib.ptr(gplane), _lib.ptr(ws),
              ws_bytes, _lib.stream(), nbytes=nbytes, tag=_lib.timing() and f"t2h_sample_bwd[C={c},r={r}]")
    return gplane


class _SamplePlane(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plane, tile: TileIndex):
        _f32(plane, "sample_plane")
        if plane.dim() != 4 or plane.shape[2] != plane.shape[3] or plane.shape[0] != tile.B:
            raise ValueError(f"sample_plane: expected [B={tile.B}, C, r, r], got {tuple(plane.shape)}")
        ctx.was_cl = _is_channels_last(plane)
        p = to_nhwc(plane)
        _lib.require_device(p, what="sample_plane")
        b, r, _, c = p.shape
        out = torch.empty(tile.n_points, c, dtype=torch.float32, device=p.device)
        _lib.call("t2h_sample_fwd", _lib.ptr(p), _lib.ptr(tile.pts), tile.dim, tile.B, tile.N, r, c, _lib.ptr(out),
                  _lib.stream(), nbytes=4 * c * tile.n_points + 8 * tile.n_points + 4 * p.numel(),
                  tag=_lib.timing() and f"t2h_sample_fwd[C={c},r={r}]")
        ctx.tile, ctx.r, ctx.c = tile, r, c
        return out

    @staticmethod
    def backward(ctx, gout):
        tile, r, c = ctx.tile, ctx.r, ctx.c
        gplane = _sample_bwd(tile, gout.contiguous(), r, c, None)
        return from_nhwc(gplane, ctx.was_cl), None


class _SamplePlaneThru(torch.autograd.Function):
    """``(sample_plane(plane), plane)``: the second output is the plane itself for its OTHER consumer (the next level's
    residual 1x1 / transposed convolution, alto.py:104-114, 233-236); the sample backward adds that consumer's gradient in
    its final store (``t2h_sample_bwd_add``) instead of autograd summing the two with an extra pass."""

    @staticmethod
    def forward(ctx, plane, tile: TileIndex):
        out = _SamplePlane.forward(ctx, plane, tile)
        return out, _alias(plane)

    @staticmethod
    def backward(ctx, gout, gthru):
        tile, r, c = ctx.tile, ctx.r, ctx.c
        if gout is None:
            return gthru, None
        gplane = _sample_bwd(tile, gout.contiguous(), r, c, None if gthru is None else to_nhwc(gthru))
        return from_nhwc(gplane, ctx.was_cl), None


def sample_plane_thru(tile: TileIndex, plane: torch.Tensor):
    """``(sampled, plane)`` -- hand the returned ``plane`` to every further consumer so that its gradient is summed inside
    the sample backward kernel."""
    return _SamplePlaneThru.apply(plane, tile)


def sample_plane(tile: TileIndex, plane: torch.Tensor) -> torch.Tensor:
    """Bilinear/border/align_corners sample of ``plane [B,C,r,r]`` at every point -> ``[B*N, C]``
    (sample_plane_feature + transpose: alto.py:90-95,122)."""
    return _SamplePlane.apply(plane, tile)


# --------------------------------------------------------------------------------------- F.interpolate
class _UpsampleBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size: int, addend):
        x = _f32(x, "upsample_bilinear").contiguous()
        if addend is not None:
            addend = _f32(addend, "upsample_bilinear").contiguous()
        _lib.require_device(x, addend, what="upsample_bilinear")
        b, c, h, w = x.shape
        out = torch.empty(b, c, size, size, dtype=torch.float32, device=x.device)
        if addend is not None and addend.shape != out.shape:
            raise ValueError("upsample_bilinear: addend must already have the output size")
        _lib.call("t2h_upsample_bilinear_fwd", _lib.ptr(x), _lib.ptr(addend) if addend is not None else None, b, c, h, w,
                  size, size, _lib.ptr(out), _lib.stream(),
                  nbytes=4 * (x.numel() + out.numel() * (2 if addend is not None else 1)))
        ctx.shape = (b, c, h, w, size)
        ctx.has_addend = addend is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        b, c, h, w, size = ctx.shape
        gout = gout.contiguous()
        gin = torch.empty(b, c, h, w, dtype=torch.float32, device=gout.device)
        _lib.call("t2h_upsample_bilinear_bwd", _lib.ptr(gout), b, c, h, w, size, size, _lib.ptr(gin), _lib.stream(),
                  nbytes=4 * (gout.numel() + gin.numel()))
        return gin, None, (gout if ctx.has_addend else None)


def upsample_bilinear(x: torch.Tensor, size: int, addend: torch.Tensor = None) -> torch.Tensor:
    """``F.interpolate(x, size, mode='bilinear', align_corners=True)`` (+ ``addend``) -- pixel.py:107,110."""
    return _UpsampleBilinear.apply(x, int(size), addend)


# --------------------------------------------------------------------------------------- sample_mode = 'bicubic' (r06)
def _plane_layout(x: torch.Tensor):
    """-> (tensor whose memory is dense NCHW or dense NHWC, channels_last flag)."""
    if x.dim() != 4:
        raise ValueError(f"expected a [B, C, H, W] plane, got {tuple(x.shape)}")
    if not x.is_cuda:
        raise RuntimeError(f"expected a tensor on the MI355X (cuda device), got {x.device}: tomosar2height_amd has no CPU path")
    if x.permute(0, 2, 3, 1).is_contiguous() and not x.is_contiguous():
        return x, 1
    return x.contiguous(), 0


def _empty_like_layout(b, c, h, w, cl, device):
    if cl:
        return torch.empty(b, h, w, c, dtype=torch.float32, device=device).permute(0, 3, 1, 2)
    return torch.empty(b, c, h, w, dtype=torch.float32, device=device)


class _UpsampleBicubic(torch.autograd.Function):
    """``F.interpolate(x, size, mode='bicubic', align_corners=True)`` (+ addend), pixel.py:107,110 with sample_mode='bicubic'."""

    @staticmethod
    def forward(ctx, x, size: int, addend):
        x, cl = _plane_layout(_f32(x, "upsample_bicubic"))
        b, c, h, w = x.shape
        out = _empty_like_layout(b, c, size, size, cl, x.device)
        if addend is not None:
            if tuple(addend.shape) != (b, c, size, size):
                raise ValueError("upsample_bicubic: addend must already have the output size")
            addend = addend.contiguous(memory_format=torch.channels_last) if cl else addend.contiguous()
        _lib.call("t2h_upsample_bicubic_fwd", _lib.ptr(x), _lib.ptr(addend) if addend is not None else None, b, c, h, w, size, size,
                  cl, out.data_ptr(), _lib.stream(), nbytes=4 * (x.numel() + out.numel() * (2 if addend is not None else 1)))
        ctx.meta = (b, c, h, w, size, cl)
        ctx.has_addend = addend is not None
        return out

    @staticmethod
    def backward(ctx, g):
        b, c, h, w, size, cl = ctx.meta
        g = g.contiguous(memory_format=torch.channels_last) if cl else g.contiguous()
        gin = _empty_like_layout(b, c, h, w, cl, g.device)
        _lib.call("t2h_upsample_bicubic_bwd", _lib.ptr(g), b, c, h, w, size, size, cl, gin.data_ptr(), _lib.stream(),
                  nbytes=4 * (g.numel() + gin.numel()))
        return gin, None, (g if ctx.has_addend else None)


def upsample_bicubic(x: torch.Tensor, size: int, addend: torch.Tensor = None) -> torch.Tensor:
    return _UpsampleBicubic.apply(x, int(size), addend)


class _SampleBicubic(torch.autograd.Function):
    """Bicubic / border / align_corners sample of plane [B, C, r, r] at pts [B, N, 2+] (x, y in [0, 1]) -> [B, N, C] point-major.
    The backward adds into the plane gradient with atomics (an off-default mode: alto.py:95 with sample_mode='bicubic')."""

    @staticmethod
    def forward(ctx, plane, pts, mode="bicubic"):
        plane, cl = _plane_layout(_f32(plane, "sample_bicubic"))
        b, c, r, r2 = plane.shape
        if r != r2 or pts.shape[0] != b:
            raise ValueError(f"sample_bicubic: plane {tuple(plane.shape)} / points {tuple(pts.shape)}")
        pts = _f32(pts, "sample_bicubic points").contiguous()
        n = pts.shape[1]
        out = torch.empty(b, n, c, dtype=torch.float32, device=plane.device)
        _lib.call(f"t2h_sample_{mode}_fwd", _lib.ptr(plane), _lib.ptr(pts), pts.shape[2], b, n, r, c, cl, _lib.ptr(out), _lib.stream(),
                  nbytes=4 * c * b * n + 4 * pts.numel() + 4 * plane.numel())
        ctx.save_for_backward(pts)
        ctx.meta = (b, c, r, n, cl, mode)
        return out

    @staticmethod
    def backward(ctx, gout):
        (pts,) = ctx.saved_tensors
        b, c, r, n, cl, mode = ctx.meta
        gout = gout.contiguous()
        gplane = _empty_like_layout(b, c, r, r, cl, gout.device)
        _lib.call(f"t2h_sample_{mode}_bwd", _lib.ptr(gout), _lib.ptr(pts), pts.shape[2], b, n, r, c, cl, gplane.data_ptr(), _lib.stream(),
                  nbytes=4 * c * b * n + 4 * pts.numel() + 4 * gplane.numel())
        return gplane, None, None


def sample_plane_mode(tile: TileIndex, plane: torch.Tensor, mode: str) -> torch.Tensor:
    """alto.py:90-95 with sample_mode='bicubic' / 'nearest' on the tile's sorted rows -> [rows, C] (regular batches only)."""
    if tile.pts.shape[0] != tile.B * tile.N:
        raise NotImplementedError(f"sample_mode={mode!r} is built for regular [B, N, 3] batches only")
    pts = tile.pts.view(tile.B, tile.N, tile.pts.shape[1])
    return _SampleBicubic.apply(plane, pts, mode).reshape(-1, plane.shape[1])


# --------------------------------------------------------------------------------------- operator-level drop-ins
def coordinate2index(x: torch.Tensor, reso: int) -> torch.Tensor:
    """utils/coordinate.py:12-28: ``x [B,N,2+] -> int64 [B,1,N]``, bit exact."""
    x = _f32(x, "coordinate2index").contiguous()
    _lib.require_device(x, what="coordinate2index")
    b, n, d = x.shape
    out = torch.empty(b, n, dtype=torch.int64, device=x.device)
    _lib.call("t2h_coordinate2index", _lib.ptr(x), d, b * n, int(reso), _lib.ptr(out), _lib.stream(),
              nbytes=16 * b * n)
    return out[:, None, :]


# The rest mirrors the reference's OPERATOR calls with their original signatures (original point order, raw int64 cell
# index, channel-major [B, C, N] features): SURVEY.md 8b "operator-level seam".  It is what a caller gets who keeps the
# reference's own module graph and swaps only the operators; every one of them is differentiable exactly where the
# reference's is.  The tile index is built from the index tensor (cell centres stand in for the coordinates) ONCE per index
# tensor -- the reference computes an index with coordinate2index and hands it to several scatter calls (pointnet.py:70,
# 76-77, 88) -- and the same HIP kernels run.  No call here synchronises with the device: an index outside
# [0, dim_size) is counted on the device, clamped into a border cell, and reported by a LATER call of this seam (or by
# ``check_indices()``) as ValueError -- the asynchronous form of the index error torch_scatter raises.  The packaged
# modules never take this route (they keep features in sorted order).
_index_tiles = {}          # id(index tensor) -> (weakref, (version, data_ptr, shape, device), cells, TileIndex, ready event, stream)
_pending_index_checks = []  # (event, pinned status copy, description)
# T2H_SEAM_SYNC_CHECK=1 (debugging): every seam call waits for its own index check and raises at the call, like torch_scatter
_SYNC_CHECK = os.environ.get("T2H_SEAM_SYNC_CHECK", "0") == "1"


def forget_index(index: torch.Tensor = None):
    """Drop the cached tile index of ``index`` (or all of them).  The cache follows a tensor by object, version counter, storage
    address, shape and device; a write that bumps none of these -- ``index.data[...] = ...``, a raw kernel, a DLPack alias -- is
    invisible to it: call this after such a write."""
    if index is None:
        _index_tiles.clear()
    else:
        _index_tiles.pop(id(index), None)


def _report_at_exit():
    # a program whose LAST seam call carried a bad index makes no later call that would report it: say so on the way out
    try:
        _poll_index_checks(wait=True)
    except ValueError as e:
        import sys
        print(f"tomosar2height_amd.ops: unreported index error at exit: {e}", file=sys.stderr)
    except Exception:
        pass


atexit.register(_report_at_exit)


def _poll_index_checks(wait: bool = False):
    keep = []
    bad = None
    for ev, host, what in _pending_index_checks:
        if wait:
            ev.synchronize()
        if ev.query():
            if int(host[0]) and bad is None:
                bad = (int(host[0]), what)
        else:
            keep.append((ev, host, what))
    _pending_index_checks[:] = keep
    if bad is not None:
        raise ValueError(f"{bad[1]}: {bad[0]} index value(s) outside [0, dim_size) (clamped into border cells on the device; "
                         "torch_scatter raises an index error for these)")


def check_indices():
    """Wait for the index-range checks of all operator-seam calls issued so far and raise ValueError if one failed."""
    _poll_index_checks(wait=True)


def _tile_from_index(index: torch.Tensor, dim_size: int) -> TileIndex:
    import weakref
    _poll_index_checks()
    if index.dim() != 3 or index.shape[1] != 1:
        raise ValueError("index must be [B, 1, N] (what coordinate2index returns)")
    if index.dtype != torch.int64:
        raise TypeError(f"index must be int64, got {index.dtype}")
    _lib.require_device(index, what="scatter index")
    hit = _index_tiles.get(id(index))
    sig = (index._version, index.data_ptr(), tuple(index.shape), index.device)
    if hit is not None and hit[0]() is index and hit[1] == sig and hit[2] == int(dim_size):
        cur = torch.cuda.current_stream(index.device)
        if hit[5] != cur:
            cur.wait_event(hit[4])           # built on another stream: its kernels must have finished before this stream reads it
        return hit[3]
    reso = int(round(dim_size ** 0.5))
    if reso * reso != dim_size:
        raise ValueError(f"dim_size={dim_size} is not a square plane")
    idx = index[:, 0, :]
    ix, iy = (idx % reso).float(), torch.div(idx, reso, rounding_mode="floor").float()
    centres = torch.stack([(ix + 0.5) / reso, (iy + 0.5) / reso, torch.zeros_like(ix)], dim=2)
    tile = TileIndex(centres.contiguous(), reso)
    # index outside [0, dim_size): status[0] counts them; copied to pinned memory in stream order and looked at later
    host = torch.empty(2, dtype=torch.int32, pin_memory=True)
    host.copy_(tile.status, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _pending_index_checks.append((ev, host, f"scatter index [B={index.shape[0]}, N={index.shape[2]}] into {dim_size} cells"))
    key = id(index)
    _index_tiles[key] = (weakref.ref(index, lambda _r, k=key: _index_tiles.pop(k, None)), sig, int(dim_size), tile, ev,
                         torch.cuda.current_stream(index.device))
    if _SYNC_CHECK:
        _poll_index_checks(wait=True)
    return tile


def _point_major(src: torch.Tensor, what: str) -> torch.Tensor:
    """src [B, C, N] -> contiguous [B, N, C] fp32 on the device; free when src is the permuted view of a point-major tensor
    (what the reference passes: pointnet.py:95,109)."""
    if src.dim() != 3:
        raise ValueError(f"{what}: src must be [B, C, N], got {tuple(src.shape)}")
    pm = _f32(src, what).permute(0, 2, 1)
    pm = pm if pm.is_contiguous() else pm.contiguous()
    _lib.require_device(pm, what=what)
    return pm


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out: torch.Tensor = None,
                 dim_size: int = None) -> torch.Tensor:
    """``torch_scatter.scatter_mean(src[B,C,N], index[B,1,N], out=zeros[B,C,R*R])`` (pointnet.py:109; alto.py:85,194).
    ``out`` must be all zero as at every reference call site; it is filled and returned.  Differentiable w.r.t. ``src``
    (grad[b,c,n] = grad_out[b,c,cell(n)] / count(cell(n)))."""
    if dim not in (-1, 2):
        raise NotImplementedError("only the last-dim form the reference uses is built")
    cells = out.shape[-1] if out is not None else int(dim_size)
    tile = _tile_from_index(index, cells)
    feat = tile.sort_rows(_point_major(src, "scatter_mean"))
    plane = rasterise_mean(tile, feat, tile.R)                       # [B, C, R, R]
    flat = plane.reshape(plane.shape[0], plane.shape[1], cells)
    if out is not None:
        out.copy_(flat)            # in place like torch_scatter (autograd follows the copy: `out` becomes differentiable)
        return out
    return flat


class _ScatterMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, index, cells: int):
        pm = _point_major(src, "scatter_max")
        b, n, c = pm.shape
        tile = _tile_from_index(index, cells)
        if tile.B != b or tile.N != n:
            raise ValueError(f"scatter_max: src {tuple(src.shape)} does not match index {tuple(index.shape)}")
        val = torch.empty(b, c, cells, dtype=torch.float32, device=pm.device)
        arg = torch.empty(b, c, cells, dtype=torch.int64, device=pm.device)
        _lib.call("t2h_scatter_max_fwd", _lib.ptr(pm), c, _lib.ptr(tile.perm), _lib.ptr(tile.off0), b, n, tile.nbits, c,
                  _lib.ptr(val), _lib.ptr(arg), _lib.stream(), nbytes=4 * c * b * n + 8 * b * n + 12 * c * b * cells)
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable(arg)
        ctx.n = n
        return val, arg

    @staticmethod
    def backward(ctx, gval, _garg):
        (arg,) = ctx.saved_tensors
        b, c, cells = arg.shape
        _poll_index_checks()                  # (a forward's bad index is reported by its own backward at the latest)
        gval = gval.contiguous()
        _lib.require_device(gval, what="scatter_max backward")
        gsrc = torch.empty(b, ctx.n, c, dtype=torch.float32, device=gval.device)
        _lib.call("t2h_scatter_max_bwd", _lib.ptr(gval), _lib.ptr(arg), b, c, ctx.n, cells, _lib.ptr(gsrc), _lib.stream(),
                  nbytes=12 * c * b * cells + 4 * c * b * ctx.n)
        return gsrc.permute(0, 2, 1), None, None


def scatter_max(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out=None, dim_size: int = None):
    """``torch_scatter.scatter_max(src[B,C,N], index[B,1,N], dim_size=R*R) -> (out[B,C,R*R], arg[B,C,R*R])``
    (pointnet.py:95).  Untouched cells: value 0, arg = N.  Ties: first point wins.  Differentiable w.r.t. ``src``: the
    gradient of ``out`` goes to the arg-max point only (pytorch-scatter's backward); ``arg`` is not differentiable."""
    if dim not in (-1, 2) or out is not None:
        raise NotImplementedError("only scatter_max(src, index, dim_size=...) over the last dim is built")
    if dim_size is None:
        raise NotImplementedError("scatter_max: pass dim_size (the reference always does, pointnet.py:95): inferring it "
                                  "from index.max() would synchronise with the device")
    return _ScatterMax.apply(src, index, int(dim_size))


class _GridSamplePoints(torch.autograd.Function):
    """Bilinear / border / align_corners sample at N points given in the caller's order.  The forward is a plain gather
    (no sort); only the backward -- ATen's is 4 C atomics per point -- sorts the points once to run the atomic-free
    segmented adjoint."""

    @staticmethod
    def forward(ctx, plane, xy):
        _f32(plane, "grid_sample")
        b, c, r, r2 = plane.shape
        if r != r2 or xy.shape[0] != b:
            raise ValueError(f"grid_sample: expected plane [B, C, r, r] and points [B, N, 2], got {tuple(plane.shape)}, "
                             f"{tuple(xy.shape)}")
        ctx.was_cl = _is_channels_last(plane)
        p = to_nhwc(plane)
        _lib.require_device(p, xy, what="grid_sample")
        n = xy.shape[1]
        pts = torch.cat([xy[..., :2], torch.zeros_like(xy[..., :1])], dim=2).contiguous()
        out = torch.empty(b * n, c, dtype=torch.float32, device=p.device)
        _lib.call("t2h_sample_fwd", _lib.ptr(p), _lib.ptr(pts), 3, b, n, r, c, _lib.ptr(out), _lib.stream(),
                  nbytes=4 * c * b * n + 8 * b * n + 4 * p.numel())
        ctx.save_for_backward(pts)
        ctx.shape = (b, c, r, n)
        return out.view(b, n, c).permute(0, 2, 1)

    @staticmethod
    def backward(ctx, gout):
        (pts,) = ctx.saved_tensors
        b, c, r, n = ctx.shape
        tile = TileIndex(pts.clamp(0.0, 1.0 - 2.0 ** -24), r)        # binning needs [0,1); sampling uses the raw xy
        tile.pts.copy_(tile.sort_rows(pts))
        g = tile.sort_rows(gout.permute(0, 2, 1))
        gplane = _sample_bwd(tile, g, r, c, None)
        return from_nhwc(gplane, ctx.was_cl), None


def grid_sample_points(plane: torch.Tensor, xy: torch.Tensor) -> torch.Tensor:
    """``F.grid_sample(plane, 2*xy[:, :, None]-1, padding_mode='border', align_corners=True).squeeze(-1)``
    (alto.py:90-95): plane [B,C,r,r], xy [B,N,2+] -> [B,C,N].  Differentiable w.r.t. the plane."""
    return _GridSamplePoints.apply(plane, _f32(xy, "grid_sample_points"))


def grid_sample(input: torch.Tensor, grid: torch.Tensor, mode: str = "bilinear", padding_mode: str = "border",
                align_corners: bool = True) -> torch.Tensor:
    """``F.grid_sample(c[B,C,r,r], vgrid[B,N,1,2], padding_mode='border', align_corners=True, mode='bilinear') ->
    [B,C,N,1]`` with the reference's own signature (alto.py:95,204).  ``vgrid = 2*xy - 1``: the kernels take xy, which is
    recovered as (vgrid + 1) / 2 in float64 -- exactly the coordinate whose ``2*xy - 1`` rounds to the given vgrid whenever
    vgrid was produced that way (every reference call), so taps and weights are ATen's bit for bit on such grids; an
    arbitrary grid is evaluated within 2^-25 of its coordinate.  Differentiable w.r.t. ``input`` only (the reference's
    points do not require grad)."""
    if mode not in ("bilinear", "bicubic", "nearest") or padding_mode != "border" or not align_corners:
        raise NotImplementedError("grid_sample: mode='bilinear' / 'bicubic' / 'nearest' with padding_mode='border', "
                                  "align_corners=True (the reference's call, alto.py:95) is built")
    if grid.dim() != 4 or grid.shape[2] != 1 or grid.shape[3] != 2:
        raise NotImplementedError("grid_sample: the grid must be [B, N, 1, 2] (one sampling location per point)")
    if grid.requires_grad:
        raise NotImplementedError("grid_sample: no gradient w.r.t. the grid is built (the reference's points are data)")
    xy = ((grid[:, :, 0, :].double() + 1.0) * 0.5).float()
    if mode != "bilinear":
        return _SampleBicubic.apply(input, xy, mode).permute(0, 2, 1).unsqueeze(-1)
    return _GridSamplePoints.apply(input, xy).unsqueeze(-1)


def interpolate(input: torch.Tensor, size=None, scale_factor=None, mode: str = "bilinear",
                align_corners: bool = True) -> torch.Tensor:
    """``F.interpolate(x, size=S, mode='bilinear' | 'bicubic', align_corners=True)`` with the reference's signature (pixel.py:107)."""
    if mode not in ("bilinear", "bicubic") or not align_corners or scale_factor is not None or size is None:
        raise NotImplementedError("interpolate: only size=S, mode='bilinear' / 'bicubic', align_corners=True (pixel.py:107) is built")
    if isinstance(size, (tuple, list)):
        if len(size) != 2 or size[0] != size[1]:
            raise NotImplementedError("interpolate: square outputs only")
        size = size[0]
    if mode == "bicubic":
        return upsample_bicubic(input, int(size))
    return upsample_bilinear(input, int(size))
