"""LocalPoolPointnet (reference: tomosar2height/encoder/pointnet.py:12-111) on the MI355X path.

One ``TileIndex`` (bin + stable Morton sort + CSR) is built per forward; all per-point tensors then live
in cell-sorted order, where the reference's ``scatter_max``/``gather`` neighbourhood pooling and its
``scatter_mean`` rasterisation are contiguous-segment HIP kernels (``ops.pool_max``,
``ops.rasterise_mean``).  Parameters and ``state_dict`` keys are the reference's: ``fc_pos``,
``blocks.{k}.{fc_0,fc_1,shortcut}``, ``fc_c``, ``unet.*``.
"""
from typing import Dict

import torch
import torch.nn as nn

from .. import _lib, mlp, ops
from ..block import ResnetBlockFC
from ..tile import TileIndex
from .alto import UNet as Alto
from .unet import UNet


class LocalPoolPointnet(nn.Module):
    def __init__(self, feature_dim=128, dim=3, hidden_dim=128, scatter_type="max", unet_type="alto",
                 unet_kwargs=None, plane_resolution=None, n_blocks=5):
        super().__init__()
        self.c_dim = feature_dim
        self.fc_pos = nn.Linear(dim, 2 * hidden_dim)
        self.blocks = nn.ModuleList([ResnetBlockFC(2 * hidden_dim, hidden_dim) for _ in range(n_blocks)])
        self.fc_c = nn.Linear(hidden_dim, feature_dim)
        self.actvn = nn.ReLU()
        self.unet_type = unet_type
        if unet_type == "unet":
            self.unet = UNet(feature_dim, in_channels=feature_dim, **(unet_kwargs or {}))
        elif unet_type == "alto":
            self.unet = Alto(feature_dim, in_channels=feature_dim, **(unet_kwargs or {}))
        else:
            raise ValueError(f"Unknown unet_type: {unet_type}")
        self.reso_plane = plane_resolution
        if scatter_type not in ("max", "mean"):                                   # pointnet.py:53-58
            raise ValueError("Invalid scatter type")
        self.scatter_type = scatter_type
        self.channels_last = False
        # check_domain=True: synchronise and raise on every forward.  Default: count on the device only; the running
        # total (out_of_domain_total) is read by Trainer at optimizer-step boundaries and by DSMGenerator per mosaic
        self.check_domain = False
        self.register_buffer("domain_status", torch.zeros(2, dtype=torch.int32), persistent=False)
        # one status pair per STREAM that builds tile indices: t2h_tile_build zeroes status[0] in stream order and counts into it
        # with atomics, so two tiles building concurrently (the trainer's tile pipeline: forward i + 1 beside backward i, each on
        # its own stream) must not share the pair -- one tile's reset would race the other's atomics.  ``domain_status`` serves
        # the first stream seen; the running totals (status[1]) of all of them are summed by out_of_domain_total()
        self._status_by_stream = {}
        # snapshot_domain (set by the Trainer): right after a tile's index is built its status pair's running total is copied to
        # pinned host memory in stream order (8 bytes + an event).  out_of_domain_total() then waits for the LATEST snapshot of
        # every pair -- i.e. for the last tile's tile_build, the first kernels of its forward -- instead of reading the device
        # counters with .item(), which waits for everything queued: at an optimizer boundary that drained the tile pipeline and
        # left the chip idle while the host issued the next forward (~3 ms per boundary)
        self.snapshot_domain = False
        self._domain_snaps = {}          # (id(status tensor), raw stream) -> [event, pinned int32[2], status tensor, valid]

    def set_channels_last(self, flag: bool):
        """Keep the grid side in channels_last memory so planes need no NCHW<->NHWC copies."""
        self.channels_last = bool(flag)
        if hasattr(self.unet, "set_channels_last"):
            self.unet.set_channels_last(flag)

    def out_of_domain_total(self, reset: bool = True) -> int:
        """Points with x or y outside [0, 1) (or NaN) over all forwards since the last reset (synchronises).  Such
        points were clamped into the border cells; the reference would index out of range (coordinate.py:12-28)."""
        extra = [st for st in self._status_by_stream.values() if st is not self.domain_status
                 and st.device == self.domain_status.device]
        pairs = [self.domain_status] + extra
        used = self.__dict__.setdefault("_used_pairs", set())
        snaps = [[sn for (sid, _), sn in self._domain_snaps.items() if sid == id(st) and sn[3]] for st in pairs]
        if self.snapshot_domain and all(sn or id(st) not in used for sn, st in zip(snaps, pairs)):
            n = 0
            for sns, st in zip(snaps, pairs):               # (a pair no tile has used since the reset holds 0)
                if id(st) in used:
                    for sn in sns:
                        sn[0].synchronize()
                    n += max(int(sn[1][1]) for sn in sns)   # (the running total is monotone: the newest snapshot is the largest)
        elif extra:                                         # (the caller's stream has joined the tile streams: Trainer.flush_pipeline)
            n = int(torch.stack(pairs)[:, 1].sum().item())
        else:
            n = int(self.domain_status[1].item())
        if reset:
            used.clear()
            for sn in self._domain_snaps.values():
                sn[3] = False
        if reset and n:
            self.domain_status.zero_()
            for st in extra:
                st.zero_()
        return n

    def _snapshot(self, status):
        """Called by every forward right after its tile index exists (see ``snapshot_domain``)."""
        self.__dict__.setdefault("_used_pairs", set()).add(id(status))
        if not self.snapshot_domain:
            return
        # one pinned buffer + event per (pair, snapshotting stream): a pair whose index was prebuilt on a producer stream is
        # snapshotted alternately from the two tile streams, and nothing orders their copies against each other -- a shared buffer
        # could receive the OLDER total last.  status[1] only grows between resets, so the reader takes the maximum per pair
        key = (id(status), _lib.stream())
        ent = self._domain_snaps.get(key)
        if ent is None:
            ent = self._domain_snaps[key] = [torch.cuda.Event(), torch.empty(2, dtype=torch.int32, pin_memory=True), status, False]
        ent[1].copy_(status, non_blocking=True)
        ent[0].record()
        ent[3] = True

    def _status_for(self, raw_stream: int) -> torch.Tensor:
        hit = self._status_by_stream.get(raw_stream)
        if hit is None or hit.device != self.domain_status.device:
            taken = any(st is self.domain_status for st in self._status_by_stream.values())
            hit = torch.zeros_like(self.domain_status) if taken else self.domain_status
            self._status_by_stream[raw_stream] = hit
        return hit

    def point_features(self, tile: TileIndex) -> torch.Tensor:
        """pointnet.py:72-82 on sorted rows: fc_pos, 5 ResNet blocks with 4 local max- (or mean-) pools, fc_c."""
        return mlp.point_trunk(tile, tile.pts, self.fc_pos, self.blocks, self.fc_c, self.scatter_type)

    def prepare(self, inputs: torch.Tensor, stream=None) -> TileIndex:
        """Build the index of a tile ahead of its step, on ``stream`` (see ``TileIndex.prebuild``)."""
        status = self._status_for(stream.cuda_stream if stream is not None else _lib.stream())
        return TileIndex.prebuild(inputs, self.reso_plane, status=status, stream=stream)

    def forward(self, inputs: torch.Tensor) -> Dict[str, torch.Tensor]:
        """inputs ``[B, N, 3]`` in [0,1) -> ``{'xy': [B, feature_dim, R, R]}``.  ``inputs`` may also be the tile's index built
        ahead of the step (``TileIndex.prebuild`` / ``prepare``)."""
        if isinstance(inputs, TileIndex):
            if inputs.R != self.reso_plane:
                raise ValueError(f"prebuilt TileIndex has resolution {inputs.R}, the encoder {self.reso_plane}")
            tile = inputs.wait_ready()
        else:
            tile = TileIndex(inputs, self.reso_plane, status=self._status_for(_lib.stream()))
        if self.check_domain:
            tile.check_domain()
        self._snapshot(tile.status)
        net = self.point_features(tile)
        if self.unet_type == "alto":      # net also feeds ALTO's first fc_c: one fused gradient sum (ops.rasterise_mean_thru)
            plane, net = ops.rasterise_mean_thru(tile, net, self.reso_plane, self.channels_last)      # pointnet.py:83
        else:
            plane = ops.rasterise_mean(tile, net, self.reso_plane, self.channels_last)
        if self.unet_type == "unet":
            return {"xy": self.unet(plane)}
        return {"xy": self.unet.forward_sorted(tile, plane, net)}                       # pointnet.py:88
