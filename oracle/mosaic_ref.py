"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of the DSM mosaic step of the reference's test path
(generator.py:85-113 blend weights; :149-157 accumulate / normalise / clamp).  float64 throughout, as the reference.

Pin status: the blend weights are pinned by tests/golden/mosaic_blend_weight.npz (produced by executing the
reference's own static method); the accumulate / normalise / clamp lines are three plain tensor statements restated
verbatim in meaning (`dsm[t:b+1, l:r+1] += h * w; weight[...] += w; dsm /= weight; dsm = maximum(dsm, 0)`)."""
import math

import numpy as np

MIN_WEIGHT = 1e-3


def linear_blend_patch_weight(shape, half_blend_percent):
    """generator.py:85-113."""
    rows, cols = shape
    wx = np.ones((rows, cols), np.float64)
    wy = np.ones((rows, cols), np.float64)
    ix = math.floor(rows * half_blend_percent[0])      # the reference sizes BOTH ramps from grid_shape_2d[0] / [1]
    iy = math.floor(cols * half_blend_percent[1])
    if ix > 0:
        wx[:, :ix] = np.linspace(MIN_WEIGHT, 1, ix)[None, :]
        wx[:, -ix:] = np.linspace(1, MIN_WEIGHT, ix)[None, :]
    if iy > 0:
        wy[:iy, :] = np.linspace(MIN_WEIGHT, 1, iy)[:, None]
        wy[-iy:, :] = np.linspace(1, MIN_WEIGHT, iy)[:, None]
    return wx * wy


def col_row(x, y, left, top, pixel_size):
    """RasterData.query_col_row with T = Affine(px, 0, left, 0, -py, top) (io_raster.py:57-66,134-142)."""
    return int(math.floor((x - left) / pixel_size[0])), int(math.floor((top - y) / pixel_size[1]))


def mosaic(tiles, dsm_shape, patch_weight):
    """tiles: iterable of (height [H,W] float32 as returned by model(...)[0].squeeze(), t_row, l_col).
    generator.py:147-157: flip(1) of the [1,H,W,1] output == vertical flip of the [H,W] grid."""
    dsm = np.zeros(dsm_shape, np.float64)
    weight = np.zeros(dsm_shape, np.float64)
    h, w = patch_weight.shape
    for height, t_row, l_col in tiles:
        grid = np.asarray(height, np.float32)[::-1, :].astype(np.float64)
        dsm[t_row:t_row + h, l_col:l_col + w] += grid * patch_weight
        weight[t_row:t_row + h, l_col:l_col + w] += patch_weight
    with np.errstate(invalid="ignore", divide="ignore"):
        dsm = dsm / weight
    return np.where(np.isnan(dsm), np.nan, np.maximum(dsm, 0.0))
