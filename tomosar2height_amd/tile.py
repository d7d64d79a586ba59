"""Per-tile cell index: the once-per-forward replacement of the reference's repeated
``coordinate2index`` calls (utils/coordinate.py:12-28; called at pointnet.py:70 and alto.py:80,190)."""
import torch

from . import _lib


def _log2_exact(v: int) -> int:
    if v < 1 or v & (v - 1):
        raise ValueError(f"plane resolution must be a power of two, got {v}")
    return v.bit_length() - 1


class TileIndex:
    """Points of a batch of tiles in cell-sorted (Morton) order + CSR offsets of the finest level.

    ``cloud`` is ``[B, N, dim]`` fp32 on the GPU with x, y in [0, 1) (the reference's tiles are strictly
    inside (0, 1): dataset.py:278).  Everything downstream (per-point features, their gradients) lives in
    this sorted order; the network is permutation-equivariant over points, so only the summation order
    inside a cell differs from the reference (stable sort: original order is kept inside a finest cell).
    """

    def __init__(self, cloud, plane_resolution: int, status: torch.Tensor = None):
        """``status``: optional persistent int32 ``[2]`` device tensor; ``status[1]`` keeps a running total of
        out-of-domain points over all tiles built with it (read it with one sync whenever convenient).

        ``cloud``: ``[B, N, dim]`` -- or a list / tuple of ``[1, N_i, dim]`` (``[N_i, dim]``) clouds with DIFFERENT point
        counts: a ragged batch (``t2h_tile_build_ragged``).  The reference runs the tiles of its 64-tile accumulation window
        one at a time because N varies per tile (tomosar2height.yaml:40); here they share one index and one set of launches.
        A ragged index reports ``N = -total_rows`` (the library's convention for such batches) and ``n_points`` rows."""
        if isinstance(cloud, (list, tuple)):
            self._init_ragged(cloud, plane_resolution, status)
            return
        if cloud.dim() != 3 or cloud.shape[-1] < 2:
            raise ValueError(f"expected a [B, N, dim>=2] point tensor, got {tuple(cloud.shape)}")
        if cloud.dtype != torch.float32:
            raise TypeError("points must be float32")
        self.R = int(plane_resolution)
        self.nbits = _log2_exact(self.R)
        if not 1 <= self.nbits <= 10:
            raise ValueError("plane resolution must be in [2, 1024]")
        cloud = cloud.contiguous()
        _lib.require_device(cloud, what="TileIndex")
        self.B, self.N, self.dim = cloud.shape
        lib = _lib.load()
        dev = cloud.device
        bn = self.B * self.N
        cells = self.B * (1 << (2 * self.nbits))
        self.pts = torch.empty(bn, self.dim, dtype=torch.float32, device=dev)
        self.perm = torch.empty(bn, dtype=torch.int32, device=dev)
        self.cell = torch.empty(bn, dtype=torch.int32, device=dev)
        self.off0 = torch.empty(cells + 1, dtype=torch.int32, device=dev)
        if status is None:
            status = torch.zeros(2, dtype=torch.int32, device=dev)
        elif status.dtype != torch.int32 or status.numel() != 2 or status.device != dev:
            raise ValueError("status must be an int32 [2] tensor on the points' device")
        self.status = status
        ws_bytes = _lib.ws_bytes("t2h_tile_workspace_bytes", self.B, self.N, self.nbits)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        _lib.call("t2h_tile_build", _lib.ptr(cloud), self.dim, self.B, self.N, self.nbits, _lib.ptr(self.pts),
                  _lib.ptr(self.perm), _lib.ptr(self.cell), _lib.ptr(self.off0), _lib.ptr(self.status), _lib.ptr(ws),
                  ws_bytes, _lib.stream(), nbytes=16 * bn + 4 * cells)
        self.device = dev
        self._adjoint = {}

    def _init_ragged(self, clouds, plane_resolution, status):
        clouds = [c.reshape(-1, c.shape[-1]) for c in clouds]
        if not clouds or any(c.dtype != torch.float32 for c in clouds) or len({c.shape[1] for c in clouds}) != 1:
            raise TypeError("a ragged batch is a non-empty list of float32 [1, N_i, dim] clouds of one dim")
        if any(c.shape[0] < 1 for c in clouds):
            raise ValueError("every tile of a ragged batch must hold at least one point (skip invalid tiles, train.py:150-151)")
        self.R = int(plane_resolution)
        self.nbits = _log2_exact(self.R)
        if not 1 <= self.nbits <= 10:
            raise ValueError("plane resolution must be in [2, 1024]")
        flat = torch.cat(clouds, 0).contiguous()
        _lib.require_device(flat, what="TileIndex")
        import ctypes
        self.B = len(clouds)
        counts = [int(c.shape[0]) for c in clouds]
        starts = [0]
        for n in counts:
            starts.append(starts[-1] + n)
        total, in_dim = starts[-1], flat.shape[1]
        self.counts, self.starts = counts, starts
        self.N = -total                                       # the library's "ragged batch, -N rows in all" (include/t2h.h)
        self.dim = in_dim + 1                                 # sorted rows carry their tile index in one more float
        dev = flat.device
        cells = self.B * (1 << (2 * self.nbits))
        self.pts = torch.empty(total, self.dim, dtype=torch.float32, device=dev)
        self.perm = torch.empty(total, dtype=torch.int32, device=dev)
        self.cell = torch.empty(total, dtype=torch.int32, device=dev)
        self.off0 = torch.empty(cells + 1, dtype=torch.int32, device=dev)
        if status is None:
            status = torch.zeros(2, dtype=torch.int32, device=dev)
        elif status.dtype != torch.int32 or status.numel() != 2 or status.device != dev:
            raise ValueError("status must be an int32 [2] tensor on the points' device")
        self.status = status
        ws_bytes = int(_lib.load().t2h_tile_ragged_workspace_bytes(self.B, total, max(counts), self.nbits))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        host_starts = (ctypes.c_int32 * (self.B + 1))(*starts)
        _lib.call("t2h_tile_build_ragged", _lib.ptr(flat), in_dim, self.B, ctypes.cast(host_starts, ctypes.c_void_p), self.nbits,
                  _lib.ptr(self.pts), self.dim, _lib.ptr(self.perm), _lib.ptr(self.cell), _lib.ptr(self.off0),
                  _lib.ptr(self.status), _lib.ptr(ws), ws_bytes, _lib.stream(), nbytes=16 * total + 4 * cells)
        self.device = dev
        self._adjoint = {}

    @property
    def ragged(self) -> bool:
        return self.N < 0

    def sample_adjoint(self, level: int):
        """``(offsets, entries)`` of the transposed bilinear-sampling matrix at ALTO level ``level`` (CSR over the pixels,
        ``t2h_sample_adjoint_build``), built on first use and kept for the tile's lifetime: every further sample backward
        at this level is one pass over ~4 N (row, weight) entries instead of a nine-cell scan per pixel."""
        hit = self._adjoint.get(level)
        if hit is None:
            npix = self.B << (2 * (self.nbits - level))
            offsets = torch.empty(_lib.load().t2h_sample_adjoint_offsets_len(self.B, self.nbits, level), dtype=torch.int32,
                                  device=self.device)
            entries = torch.empty(max(4 * self.n_points, 1), 2, dtype=torch.int32, device=self.device)
            _lib.call("t2h_sample_adjoint_build", _lib.ptr(self.pts), self.dim, _lib.ptr(self.off0), self.B, self.N, self.nbits,
                      level, _lib.ptr(offsets), _lib.ptr(entries), _lib.stream(),
                      nbytes=2 * 9 * 8 * self.n_points + 8 * npix + 32 * self.n_points)
            hit = self._adjoint[level] = (offsets, entries)
        return hit

    def trunk_units(self):
        """Work units of the one-launch trunk forward (``t2h_trunk_units_build``): a dense list of (first row, end row) pairs of whole
        finest-level cells packed into at most 128 rows, then their number (``trunk_unit_list`` decodes it).  They depend on the
        index only: built on first use, kept for the tile's lifetime."""
        hit = self._adjoint.get("trunk_units")
        if hit is None:
            lib = _lib.load()
            m = int(self.pts.shape[0])
            n = int(lib.t2h_trunk_units_words(m))
            hit = torch.empty(n + (n & 1), dtype=torch.int32, device=self.device)
            _lib.call("t2h_trunk_units_build", _lib.ptr(self.cell), _lib.ptr(self.off0), m, _lib.ptr(hit), _lib.stream(), nbytes=4 * n)
            self._adjoint["trunk_units"] = hit
        return hit

    def trunk_unit_list(self) -> torch.Tensor:
        """The work units as an [n, 2] int32 tensor on the host side of a synchronisation (tests, probes)."""
        buf = self.trunk_units()
        m = int(self.pts.shape[0])
        cap = (m + 1023) // 1024 * 24
        n = int(buf[2 * cap].item())
        return buf[:2 * n].view(n, 2)

    def cell_order(self, level: int):
        """Dispatch order of the on-chip walks at ALTO level ``level`` (``t2h_cell_order_build``): the level's cells, then its
        2 x 2 blocks of cells, each by falling row count -- the dense cells' workgroups start first.  Built on first use and kept
        for the tile's lifetime (the forward and the backward walk of the level share it).  Scheduling only: results do not depend
        on it.  ``None`` where the level has no such list (the coarsest possible level, a 1 x 1 plane)."""
        hit = self._adjoint.get(("order", level), False)
        if hit is False:
            # all levels that can take an on-chip walk (1 .. 3) in ONE launch, the first time any of them is asked for
            lib = _lib.load()
            lo, hi = 1, min(3, self.nbits - 1)
            if not (lo <= level <= hi):
                lo = hi = level
            lens = [int(lib.t2h_cell_order_len(self.B, self.nbits, lv)) for lv in range(lo, hi + 1)]
            if min(lens) > 0 and self.n_points > 0:
                buf = torch.empty(sum(lens), dtype=torch.int32, device=self.device)
                _lib.call("t2h_cell_order_build_range", _lib.ptr(self.off0), self.B, self.nbits, lo, hi, _lib.ptr(buf),
                          _lib.stream(), nbytes=16 * sum(lens))
                at = 0
                for lv, n in zip(range(lo, hi + 1), lens):
                    self._adjoint[("order", lv)] = buf[at:at + n]
                    at += n
            else:
                for lv in range(lo, hi + 1):
                    self._adjoint[("order", lv)] = None
            hit = self._adjoint[("order", level)]
        return hit

    @property
    def n_points(self) -> int:
        return -self.N if self.N < 0 else self.B * self.N

    # -------------------------------------------------------------------------------------- built ahead of its step
    @classmethod
    def prebuild(cls, cloud: torch.Tensor, plane_resolution: int, status: torch.Tensor = None, stream=None):
        """The index of the NEXT tile, built on ``stream`` (a side stream) while the current tile's step runs: the sort, the
        sampling adjoint of the finest level (when the sample backward will take it) and the per-cell point counts are a
        handful of small, latency-bound kernels (~150 us at N = 131072) that leave the chip idle when they run alone at the
        head of a step.  Pass the result where the ``[B, N, 3]`` cloud would go (``LocalPoolPointnet.forward``); the consumer
        waits for ``ready`` on its own stream.  Same kernels, same results as building inside the step."""
        if stream is None:
            return cls(cloud, plane_resolution, status=status)
        main = torch.cuda.current_stream(cloud.device)
        stream.wait_stream(main)                                   # the cloud was produced on the caller's stream
        with torch.cuda.stream(stream):
            tile = cls(cloud, plane_resolution, status=status)
            from . import ops, deferred
            if ops.SAMPLE_ADJOINT and 0 < tile.n_points <= ops.SAMPLE_ADJOINT_MAX_ROWS * tile.B * tile.R * tile.R:
                tile.sample_adjoint(0)
            for lv in range(min(4, tile.nbits)):
                deferred.counts(tile, lv)
                if lv >= 1 and deferred.CELL_ORDER:
                    tile.cell_order(lv)
            tile.ready = torch.cuda.Event()
            tile.ready.record(stream)
        cloud.record_stream(stream)
        for t in tile._tensors():
            t.record_stream(main)                                  # allocated on the side stream, read on the main one
        return tile

    def _tensors(self):
        out = [self.pts, self.perm, self.cell, self.off0]
        for key, val in self._adjoint.items():
            if isinstance(key, tuple):                       # ("order", level): one tensor or None
                out += [val] if val is not None else []
            else:
                out += list(val)
        out += list(self.__dict__.get("_cell_counts", {}).values())
        return out

    def wait_ready(self):
        """Make the current stream wait for a prebuilt index (no-op for one built in stream order)."""
        ev = self.__dict__.pop("ready", None)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        return self

    def to(self, device=None, *args, **kwargs):
        """Where a cloud tensor would be moved to the model's device: the index already lives there."""
        if device is not None and torch.device(device).type != self.device.type:
            raise ValueError("a TileIndex cannot be moved between devices")
        return self

    def level(self, reso: int) -> int:
        """ALTO level k whose plane resolution is ``reso`` (= R >> k)."""
        k = self.nbits - _log2_exact(int(reso))
        if k < 0:
            raise ValueError(f"plane resolution {reso} is finer than the tile's {self.R}")
        return k

    def out_of_domain(self) -> int:
        """Number of points with x or y outside [0,1) (synchronises).  The reference would raise an index
        error inside torch_scatter for these; here they were clamped into the border cells."""
        return int(self.status[0].item())

    def check_domain(self):
        bad = self.out_of_domain()
        if bad:
            raise ValueError(f"{bad} point(s) have x or y outside [0, 1): normalise the tile first")

    def sort_rows(self, feat: torch.Tensor) -> torch.Tensor:
        """[B, N, C] original order -> [B*N, C] sorted order (test/interop helper, not on the hot path)."""
        if self.N < 0:
            raise NotImplementedError("sort_rows / unsort_rows: equal-N batches only (test / interop helpers)")
        b, n, c = feat.shape
        idx = (self.perm.long() + torch.arange(b, device=feat.device).repeat_interleave(n) * n)
        return feat.reshape(b * n, c)[idx].contiguous()

    def unsort_rows(self, feat_sorted: torch.Tensor) -> torch.Tensor:
        """[B*N, C] sorted order -> [B, N, C] original order (test/interop helper)."""
        c = feat_sorted.shape[1]
        idx = (self.perm.long() + torch.arange(self.B, device=feat_sorted.device).repeat_interleave(self.N) * self.N)
        out = torch.empty_like(feat_sorted)
        out[idx] = feat_sorted
        return out.reshape(self.B, self.N, c)
