#!/usr/bin/env python3
"""A/B probe of the on-chip hidden-activation kernels (t2h_sample_relu_cellsums2 / t2h_sample_bwd_from_sums) at the three
shapes of the benchmarked step (N = 131072 clustered points): the r04 kernels against the r05 ones (environment switches read
per call), outputs compared bit for bit, durations by HIP events (median of REPS launches, kernels alone).

    python3 profiles/walk_probe.py            # prints one line per (kernel, shape, variant)
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomosar2height_amd import _lib, deferred                 # noqa: E402
from tomosar2height_amd.synthetic import berlin_tile          # noqa: E402
from tomosar2height_amd.tile import TileIndex                 # noqa: E402

REPS = int(os.environ.get("REPS", "30"))
FWD = [v for v in os.environ.get("FWD_VARIANTS", "0,1,1o").split(",") if v]     # 0 = r04 kernel, 1 = r05, o = + dispatch order
BWD = [v for v in os.environ.get("BWD_VARIANTS", "1,1o").split(",") if v]      # (r05 kernel without / with the dispatch order)
dev = torch.device("cuda:0")
tile = TileIndex(berlin_tile(1000, clustered=os.environ.get("UNIFORM", "0") != "1")["inputs"].to(dev), 256)
M = tile.n_points


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return statistics.median(ts), min(ts)


for c2, r in ((1024, 32), (512, 64), (256, 128)):
    lv = tile.level(r)
    q = torch.randn(r * r, c2, device=dev)
    planes = {l: torch.empty((256 >> l) ** 2, c2, device=dev) for l in range(lv + 1)}
    bits = torch.empty(M * (c2 // 256) * 4, dtype=torch.int64, device=dev)
    fwd_bytes = 4 * q.numel() + 12 * M + 4 * c2 * (planes[0].shape[0] + planes[1].shape[0]) + (c2 // 8) * M

    order = torch.empty(_lib.load().t2h_cell_order_len(tile.B, tile.nbits, lv), dtype=torch.int32, device=dev)
    _lib.call("t2h_cell_order_build", _lib.ptr(tile.off0), tile.B, tile.nbits, lv, _lib.ptr(order), _lib.stream())
    med, best = timed(lambda: _lib.call("t2h_cell_order_build", _lib.ptr(tile.off0), tile.B, tile.nbits, lv, _lib.ptr(order), _lib.stream()))
    cells = r * r
    rows = (tile.off0[::4 ** lv][1:] - tile.off0[::4 ** lv][:-1])[order[:cells].long()]
    key = (rows // max(1, (M // cells + 63) // 64)).clamp(max=2047)         # the sort's key: rows / quantum, capped
    print(f"order r={r}: build {med:.1f} us; is a permutation: {bool((order[:cells].sort().values == torch.arange(cells, device=dev)).all())}, "
          f"{bool((order[cells:].sort().values == torch.arange(cells // 4, device=dev)).all())}; keys falling: {bool((key[1:] <= key[:-1]).all())} "
          f"(rows {int(rows[0])} .. {int(rows[-1])})", flush=True)
    use_order = [False]

    def fwd():
        _lib.call("t2h_sample_relu_cellsums_ordered", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N,
                  tile.nbits, lv, 0, c2, planes[0].data_ptr(), planes[0].stride(0), planes[1].data_ptr(), planes[1].stride(0),
                  _lib.ptr(bits), _lib.ptr(order) if use_order[0] else None, _lib.stream())

    ref = None
    for v in FWD:
        os.environ["T2H_CELLSUMS_V2"] = v.rstrip("o")
        use_order[0] = v.endswith("o")
        planes[0].fill_(float("nan")); planes[1].fill_(float("nan")); bits.fill_(-1)
        fwd()
        torch.cuda.synchronize()
        out = (planes[0].clone(), planes[1].clone(), bits.clone())
        if ref is None:
            ref = out
        same = all(torch.equal(a, b) for a, b in zip(ref, out))
        med, best = timed(fwd)
        print(f"fwd  C={c2:5d} r={r:4d} variant={v}: {med:7.1f} us (min {best:6.1f})  {fwd_bytes / med / 1e6:6.2f} TB/s  "
              f"frac {fwd_bytes / med / 1e6 / 8:.3f}  identical={same}", flush=True)
    os.environ.pop("T2H_CELLSUMS_V2", None)

    grads = [(torch.randn_like(p), l) for l, p in planes.items()]
    arr, lvs, lds = deferred._plane_args(grads)
    ws_bytes = _lib.ws_bytes("t2h_sample_bwd_workspace_bytes", tile.B, tile.N, tile.nbits, lv, c2)
    ws = _lib.workspace(ws_bytes, dev)
    dq = torch.empty(r * r, c2, device=dev)
    bwd_bytes = (c2 // 8) * M + 12 * M + 4 * dq.numel() + sum(4 * c2 * p.shape[0] for p, _ in grads)

    def bwd():
        _lib.call("t2h_sample_bwd_from_sums_ordered", arr, lvs, lds, len(grads), _lib.ptr(tile.cell), _lib.ptr(ref[2]), 1, _lib.ptr(tile.pts),
                  tile.dim, _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits, lv, c2, _lib.ptr(dq), _lib.ptr(ws), ws_bytes,
                  _lib.ptr(order) if use_order[0] else None, _lib.stream())

    bref = None
    for v in BWD:
        os.environ["T2H_WALK_V2"] = v.rstrip("o")
        use_order[0] = v.endswith("o")
        dq.fill_(float("nan"))
        bwd()
        torch.cuda.synchronize()
        out = dq.clone()
        if bref is None:
            bref = out
        same = torch.equal(bref, out)
        med, best = timed(bwd)
        print(f"bwd  C={c2:5d} r={r:4d} variant={v}: {med:7.1f} us (min {best:6.1f})  {bwd_bytes / med / 1e6:6.2f} TB/s  "
              f"frac {bwd_bytes / med / 1e6 / 8:.3f}  identical={same}", flush=True)
    os.environ.pop("T2H_WALK_V2", None)
print("walk_probe done")
