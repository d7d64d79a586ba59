"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of the point half of TomoSARDataset.__getitem__ for the default
configuration (dataset.py:229-278: strict crop, z_shift = local min, float64 normalise to [0,1], float32 cast, strict
re-crop).  The reference builds a 4x4 matrix and inverts it; with no augmentation that is the closed form below
(float64 rounding differences of 1e-16 can move a float32 result by one ulp in rare ties).  Pinned by
tests/golden/tile_producer.npz, produced with the reference's own crop_pc_2d / invert_transform / apply_transform."""
import numpy as np


def produce_tile(chunk, anchor, patch_size=(512.0, 512.0), z_span=190.2, rot_times=0, flip_dim=-1):
    chunk = np.asarray(chunk, np.float64)
    lo = np.asarray(anchor, np.float64)
    hi = lo + np.asarray(patch_size, np.float64)
    first = (chunk[:, 0] > lo[0]) & (chunk[:, 0] < hi[0]) & (chunk[:, 1] > lo[1]) & (chunk[:, 1] < hi[1])
    idx = np.nonzero(first)[0]
    if idx.size == 0:
        return idx, np.zeros((0, 3), np.float32), np.nan
    pts = chunk[idx]
    z_shift = pts[:, 2].min()
    centre = (lo + hi) / 2.0
    u, v = (pts[:, 0] - centre[0]) / patch_size[0], (pts[:, 1] - centre[1]) / patch_size[1]
    for _ in range(rot_times):            # flip_mat @ rot_mat of dataset.py:253-269: -90 deg about z per step, then the flip
        u, v = v, -u
    if flip_dim == 0:
        u = -u
    if flip_dim == 1:
        v = -v
    norm = np.stack([u + 0.5, v + 0.5, (pts[:, 2] - z_shift) / z_span], 1).astype(np.float32)
    keep = (norm[:, 0] > 0) & (norm[:, 0] < 1) & (norm[:, 1] > 0) & (norm[:, 1] < 1)
    return idx[keep], norm[keep], z_shift


def raster_patch(raster, row, col, shape, rot_times=0, flip_dim=-1):
    """dataset.py:296-328: raster [C, H, W]; (col, row) = pixel of the window's bottom-left corner; the patch's rows end at
    `row` (rasters are north-up); quarter turns in the (W, H) plane, the augmentation flip, float32, south row first."""
    raster = np.asarray(raster)
    t = raster[:, row - shape[0] + 1:row + 1, col:col + shape[1]]
    if rot_times > 0:
        t = np.rot90(t, rot_times, axes=(-1, -2))
    if flip_dim == 0:
        t = t[:, :, ::-1]
    if flip_dim == 1:
        t = t[:, ::-1, :]
    return np.ascontiguousarray(t.astype(np.float32)[:, ::-1, :])
