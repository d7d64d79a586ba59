"""DSM generation on the device (reference: generator.py:15-165, the per-tile forward + mosaic blend of `test.py`).

Per tile: ``model.eval(); no_grad`` forward through the HIP path, then the float64 weighted accumulate into the
mosaic with ``t2h_mosaic_accumulate`` (the reference's ``.flip(1)``, ``h * patch_weight`` and the two ``+=`` in one
kernel, no per-tile host round trip), and one ``t2h_mosaic_finalize`` (``/=``, ``maximum(., 0)``).  The reference's
constructor also builds an unused 512 x 512 x 600 x 3 query grid (~1.9 GB, generator.py:51-53,74-83); that is not
reproduced.  GeoTIFF writing needs rasterio and stays out of scope: the mosaic is returned as a tensor.

``tiles`` is any iterable of the dicts ``TomoSARDataset.__getitem__`` yields (after collate): ``inputs [1,N,3]``,
optional ``image [1,3,512,512]``, ``min_bound``/``max_bound`` (world x, y[, z]) and ``is_valid``.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import _lib

MIN_WEIGHT = 1e-3


class DSMGenerator:
    NODATA_VALUE = np.nan

    def __init__(self, model, device, tiles, bounds, dsm_pixel_size=(1.0, 1.0), patch_size=(512.0, 512.0),
                 half_blend_percent=None, use_cloud=True, use_image=False, process_group=None):
        """``process_group`` (an addition): with W ranks, rank r runs the tiles ``i % W == r`` and the
        float64 ``dsm / weight`` pair is summed over ranks with ONE all-reduce before the normalisation
        (generator.py:149-157 -- the blend is a sum over tiles, so it shards without any other exchange)."""
        self.model, self.device, self.tiles = model, device, tiles
        self.group = process_group
        self.world = dist.get_world_size(process_group) if process_group is not None else 1
        self.rank = dist.get_rank(process_group) if process_group is not None else 0
        self.pixel_size = [float(dsm_pixel_size[0]), float(dsm_pixel_size[1])]
        self.half_blend_percent = half_blend_percent or [0.5, 0.5]
        self.use_cloud, self.use_image = use_cloud, use_image
        self.l_bound, self.b_bound, self.r_bound, self.t_bound = (float(v) for v in bounds)
        self.dsm_shape = self.cal_dsm_shape((self.l_bound, self.b_bound), (self.r_bound, self.t_bound), self.pixel_size)
        grid = (int(round(patch_size[0] / self.pixel_size[0])), int(round(patch_size[1] / self.pixel_size[1])))
        self.patch_weight = self._linear_blend_patch_weight(grid, self.half_blend_percent).to(device)

    @staticmethod
    def cal_dsm_shape(bl_bound, tr_bound, pixel_size):
        """utils/io_raster.py:78-95."""
        return (math.floor((tr_bound[1] - bl_bound[1]) / pixel_size[1]),
                math.floor((tr_bound[0] - bl_bound[0]) / pixel_size[0]))

    @staticmethod
    def _linear_blend_patch_weight(grid_shape_2d, half_blend_percent):
        """generator.py:85-113 (float64; both ramp lengths derive from the grid shape as in the reference)."""
        assert 0 <= half_blend_percent[0] <= 0.5 and 0 <= half_blend_percent[1] <= 0.5
        rows, cols = grid_shape_2d
        wx = torch.ones(rows, cols, dtype=torch.float64)
        wy = torch.ones(rows, cols, dtype=torch.float64)
        ix, iy = math.floor(rows * half_blend_percent[0]), math.floor(cols * half_blend_percent[1])
        if ix > 0:
            wx[:, :ix] = torch.linspace(MIN_WEIGHT, 1, ix, dtype=torch.float64)[None, :]
            wx[:, -ix:] = torch.linspace(1, MIN_WEIGHT, ix, dtype=torch.float64)[None, :]
        if iy > 0:
            wy[:iy, :] = torch.linspace(MIN_WEIGHT, 1, iy, dtype=torch.float64)[:, None]
            wy[-iy:, :] = torch.linspace(1, MIN_WEIGHT, iy, dtype=torch.float64)[:, None]
        return wx * wy

    def query_col_row(self, x, y):
        """RasterData.query_col_row for T = Affine(px, 0, left, 0, -py, top) (utils/io_raster.py:57-66,134-142)."""
        return (int(math.floor((x - self.l_bound) / self.pixel_size[0])),
                int(math.floor((self.t_bound - y) / self.pixel_size[1])))

    def accumulate(self, dsm, weight, height, t_row, l_col):
        """One tile into the mosaic; ``height`` is the model's [1,H,W,1] (or [H,W]) fp32 output, un-flipped."""
        h = height.reshape(height.shape[-3], height.shape[-2]) if height.dim() == 4 else height
        h = h.contiguous()
        _lib.require_device(h, dsm, weight, what="mosaic_accumulate")
        if tuple(h.shape) != tuple(self.patch_weight.shape):
            raise ValueError(f"height map {tuple(h.shape)} does not have the blend-weight shape "
                             f"{tuple(self.patch_weight.shape)} (patch_size / dsm_pixel_size)")
        if t_row < 0 or l_col < 0 or t_row + h.shape[0] > dsm.shape[0] or l_col + h.shape[1] > dsm.shape[1]:
            # the reference's slice `+=` (generator.py:152-154) raises a shape error here
            raise ValueError(f"tile rows [{t_row}, {t_row + h.shape[0]}) x cols [{l_col}, {l_col + h.shape[1]}) is not "
                             f"inside the {tuple(dsm.shape)} mosaic")
        _lib.call("t2h_mosaic_accumulate", _lib.ptr(h), h.shape[0], h.shape[1], _lib.ptr(self.patch_weight), _lib.ptr(dsm),
                  _lib.ptr(weight), dsm.shape[0], dsm.shape[1], int(t_row), int(l_col), 1, _lib.stream(),
                  nbytes=h.numel() * (4 + 8 + 4 * 8))

    def generate_dsm(self) -> torch.Tensor:
        dev = self.device
        dsm = torch.zeros(self.dsm_shape, dtype=torch.float64, device=dev)
        weight = torch.zeros(self.dsm_shape, dtype=torch.float64, device=dev)
        self.model.eval()
        for i, data in enumerate(self.tiles):
            if i % self.world != self.rank:
                continue
            if not bool(data["is_valid"][0]):
                continue
            min_b = data["min_bound"].squeeze().double()
            max_b = data["max_bound"].squeeze().double()
            l_col, _ = self.query_col_row(min_b[0].item() + self.pixel_size[0] / 2, min_b[1].item() + self.pixel_size[1] / 2)
            _, t_row = self.query_col_row(max_b[0].item() - self.pixel_size[0] / 2, max_b[1].item() - self.pixel_size[1] / 2)
            with torch.no_grad():
                cloud = data.get("inputs").to(dev) if self.use_cloud else None
                image = data.get("image").to(dev) if self.use_image else None
                height = self.model(input_cloud=cloud, input_image=image)[0]
            self.accumulate(dsm, weight, height, t_row, l_col)
        if self.world > 1:
            pair = torch.stack((dsm, weight))
            dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=self.group)      # one collective per mosaic
            dsm, weight = pair[0].contiguous(), pair[1].contiguous()
        if hasattr(self.model, "out_of_domain_total"):
            bad = self.model.out_of_domain_total()
            if bad:
                raise ValueError(f"{bad} input point(s) had x or y outside [0, 1) (or NaN): un-normalised tile")
        _lib.call("t2h_mosaic_finalize", _lib.ptr(dsm), _lib.ptr(weight), dsm.numel(), _lib.stream(), nbytes=24 * dsm.numel())
        return dsm
