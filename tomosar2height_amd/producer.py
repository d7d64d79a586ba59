"""Device-side tile producer: the point-cloud half of ``TomoSARDataset.__getitem__``
(reference dataset.py:201-289, default config: no flip / rotate augmentation, ``z_shift: local_min``).

A chunk cloud stays resident in HBM as float64 world coordinates; ``crop(anchor)`` returns the normalised float32 tile
``inputs [1, N, 3]`` (same point order as ``torch.where`` gives the reference), ready for ``TomoSAR2Height`` -- no
DataLoader pickling hop.  One device->host read of N per tile is inherent (tiles have data-dependent size)."""
import torch

from . import _lib


class TileProducer:
    def __init__(self, chunk_points: torch.Tensor, patch_size=(512.0, 512.0), z_bound=(-33.7, 156.5),
                 x_range=(0.0, 1.0), y_range=(0.0, 1.0)):
        if chunk_points.dim() != 2 or chunk_points.shape[1] != 3 or chunk_points.dtype != torch.float64:
            raise ValueError("chunk_points must be a [P, 3] float64 tensor (geo-coordinates need float64, dataset.py:232)")
        self.points = chunk_points.contiguous()
        _lib.require_device(self.points, what="TileProducer")
        self.patch_size = (float(patch_size[0]), float(patch_size[1]))
        # scale_mat diagonal, dataset.py:187-190
        self.scale = (self.patch_size[0] / (x_range[1] - x_range[0]), self.patch_size[1] / (y_range[1] - y_range[0]),
                      float(z_bound[1] - z_bound[0]))
        if tuple(x_range) != (0.0, 1.0) or tuple(y_range) != (0.0, 1.0):
            raise NotImplementedError("only the shipped normalisation ranges [0, 1] are built (conf/dataset/base.yaml:19-21)")
        p = self.points.shape[0]
        dev = self.points.device
        self._out = torch.empty(max(p, 1), 3, dtype=torch.float32, device=dev)
        self._src = torch.empty(max(p, 1), dtype=torch.int32, device=dev)
        self._count = torch.zeros(1, dtype=torch.int32, device=dev)
        self._zshift = torch.zeros(1, dtype=torch.float64, device=dev)
        self._ws_bytes = _lib.load().t2h_tile_crop_workspace_bytes(p)
        self._ws = _lib.workspace(self._ws_bytes, dev)

    def crop(self, anchor, with_index: bool = False):
        """anchor = (x, y) of the window's bottom-left corner in world coordinates (dataset.py:229-230)."""
        ax, ay = float(anchor[0]), float(anchor[1])
        mx, my = ax + self.patch_size[0], ay + self.patch_size[1]
        p = self.points.shape[0]
        _lib.call("t2h_tile_crop_normalise", _lib.ptr(self.points), p, ax, ay, mx, my, self.scale[0], self.scale[1],
                  self.scale[2], _lib.ptr(self._out), _lib.ptr(self._src), _lib.ptr(self._count), _lib.ptr(self._zshift),
                  _lib.ptr(self._ws), self._ws_bytes, _lib.stream(), nbytes=2 * 24 * p)
        _lib.call("t2h_tile_crop_finish", _lib.ptr(self._zshift), _lib.stream())
        n = int(self._count.item())                                  # the one inherent sync per tile
        out = {"min_bound": torch.tensor([ax, ay], dtype=torch.float64), "max_bound": torch.tensor([mx, my], dtype=torch.float64),
               "is_valid": torch.tensor([n > 0])}
        if n > 0:
            out["inputs"] = self._out[:n].clone()[None]
            out["z_shift"] = self._zshift.clone()
            if with_index:
                out["index"] = self._src[:n].long().clone()
        return out
