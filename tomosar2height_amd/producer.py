"""Device-side tile producer: ``TomoSARDataset.__getitem__`` (reference dataset.py:201-330, ``z_shift:
local_min``) on data resident in HBM -- ``TileProducer`` the point-cloud half (strict crop, float64 normalise, optional
flip / rotate augmentation), ``RasterPatcher`` the raster half (DSM target and satellite-image patch: slice, quarter
turns, flips, float32, south row first), ``TileSource`` both together as the dict the Trainer consumes.

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
        self._ws_bytes = _lib.ws_bytes("t2h_tile_crop_workspace_bytes", p)
        self._ws = _lib.workspace(self._ws_bytes, dev)

    def crop(self, anchor, with_index: bool = False, rot_times: int = 0, flip_dim: int = -1):
        """anchor = (x, y) of the window's bottom-left corner in world coordinates (dataset.py:229-230); ``rot_times`` /
        ``flip_dim``: the training augmentation of dataset.py:253-269 (keys of rot_mat_dic / flip_mat_dic)."""
        ax, ay = float(anchor[0]), float(anchor[1])
        mx, my = ax + self.patch_size[0], ay + self.patch_size[1]
        p = self.points.shape[0]
        _lib.call("t2h_tile_crop_normalise_aug", _lib.ptr(self.points), p, ax, ay, mx, my, self.scale[0], self.scale[1],
                  self.scale[2], int(rot_times), int(flip_dim), _lib.ptr(self._out), _lib.ptr(self._src), _lib.ptr(self._count),
                  _lib.ptr(self._zshift), _lib.ptr(self._ws), self._ws_bytes, _lib.stream(), nbytes=2 * 24 * p)
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


class RasterPatcher:
    """The DSM / satellite-image patch of a tile (dataset.py:291-328) from a north-up raster resident in HBM.

    ``raster``: ``[H, W]`` or ``[C, H, W]`` device tensor, float32 (the DSM, dataset.py:133) or float64 (the image after
    ``(int - mean) / std``, dataset.py:108-113); ``left`` / ``top``: world coordinates of the raster's upper-left corner,
    ``pixel_size``: (x, y) metres per pixel -- the affine transform Affine(px, 0, left, 0, -py, top) of io_raster.py."""

    def __init__(self, raster: torch.Tensor, left: float, top: float, pixel_size=(1.0, 1.0), patch_size=(512.0, 512.0)):
        if raster.dim() == 2:
            raster = raster[None]
        if raster.dim() != 3 or raster.dtype not in (torch.float32, torch.float64):
            raise ValueError("raster must be a [H, W] or [C, H, W] float32 / float64 tensor")
        self.raster = raster.contiguous()
        _lib.require_device(self.raster, what="RasterPatcher")
        self.left, self.top = float(left), float(top)
        self.pixel_size = (float(pixel_size[0]), float(pixel_size[1]))
        shape = (patch_size[1] / self.pixel_size[1], patch_size[0] / self.pixel_size[0])
        if any(v != int(v) for v in shape):
            raise ValueError("Patch size should be integer multiple of the raster pixel size")      # dataset.py:121-122,136-137
        self.patch_shape = (int(shape[0]), int(shape[1]))

    def query_col_row(self, x: float, y: float):
        """io_raster.py:134-142 for a north-up raster."""
        import math
        return (int(math.floor((x - self.left) / self.pixel_size[0])), int(math.floor((self.top - y) / self.pixel_size[1])))

    def patch(self, anchor, rot_times: int = 0, flip_dim: int = -1) -> torch.Tensor:
        """``[C, ph, pw]`` float32: the raster under the window whose bottom-left corner is ``anchor`` (dataset.py:296-328)."""
        col, row = self.query_col_row(float(anchor[0]) + self.pixel_size[0] / 2.0, float(anchor[1]) + self.pixel_size[1] / 2.0)
        ph, pw = self.patch_shape
        c, h, w = self.raster.shape
        out = torch.empty(c, ph, pw, dtype=torch.float32, device=self.raster.device)
        _lib.call("t2h_raster_patch", _lib.ptr(self.raster), 1 if self.raster.dtype == torch.float64 else 0, c, h, w,
                  row - ph + 1, col, ph, pw, int(rot_times), int(flip_dim), _lib.ptr(out), _lib.stream(),
                  nbytes=(self.raster.element_size() + 4) * out.numel())
        return out


class TileSource:
    """``TomoSARDataset.__getitem__`` on the device: points + DSM target (+ image), one augmentation draw for all three
    (dataset.py:253-263: ``np.random.choice`` over the keys of rot_mat_dic / flip_mat_dic).  Returns the collated dict
    ``Trainer.train_step`` takes: ``inputs [1, N, 3]``, ``dsm [1, ph, pw]``, ``image [1, 3, ph, pw]``, ``is_valid``."""

    def __init__(self, points: TileProducer, dsm: RasterPatcher, image: RasterPatcher = None, flip_augm=False,
                 rotate_augm=False, rng=None, stream=None):
        """``stream``: a side ``torch.cuda.Stream`` to produce tiles on.  The one host read per tile (its point count) then
        waits for the crop kernels only, not for the training step still running on the main stream, so tile t + 1 can be
        produced while step t executes (the reference gets the same overlap from its DataLoader workers, train.py:80-85);
        the main stream is made to wait for the produced tensors before ``get`` returns."""
        import numpy as np
        self.points, self.dsm, self.image = points, dsm, image
        self.flip_augm, self.rotate_augm = flip_augm, rotate_augm
        self.rng = rng if rng is not None else np.random
        self.stream = stream
        self._joined = False

    def get(self, anchor, defer_wait: bool = False):
        """``defer_wait`` (side stream only): do not make the caller's stream wait now; the tile carries a ``ready`` event and
        the consumer calls ``TileSource.wait(tile)`` when it USES the tile -- so that a step issued between production and
        use does not wait for the production."""
        if self.stream is None:
            return self._produce(anchor)
        main = torch.cuda.current_stream()
        if not self._joined:
            # the chunk upload and the producers' persistent buffers (_count, _out, _zshift) were initialised on the caller's
            # stream: the side stream must see them before its first crop (afterwards only the side stream touches them)
            self.stream.wait_stream(main)
            self._joined = True
        with torch.cuda.stream(self.stream):
            out = self._produce(anchor)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        for v in out.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(main)                             # allocated on the side stream, consumed on the main one
        if defer_wait:
            out["ready"] = ready
        else:
            main.wait_event(ready)
        return out

    @staticmethod
    def wait(tile):
        """Make the current stream wait for a tile produced with ``defer_wait=True`` (no-op otherwise); returns the tile."""
        ev = tile.pop("ready", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        return tile

    def _produce(self, anchor):
        rot = int(self.rng.choice(4)) if self.rotate_augm else 0
        flip = (-1, 0, 1)[int(self.rng.choice(3))] if self.flip_augm else -1
        out = self.points.crop(anchor, rot_times=rot, flip_dim=flip)
        if not bool(out["is_valid"][0]):
            return out                                            # dataset.py:235-241: skipped by the training loop
        out["flip"], out["rotate"] = flip, rot
        if self.image is not None:
            out["image"] = self.image.patch(anchor, rot, flip)[None]
        out["dsm"] = self.dsm.patch(anchor, rot, flip)
        return out
