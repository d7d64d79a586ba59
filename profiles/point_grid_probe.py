#!/usr/bin/env python3
"""Time the point <-> grid kernels of one Berlin tile-step in isolation (N = 131072 clustered points, the ALTO level
shapes): sample fwd / bwd, rasterise fwd / bwd, with HIP events; prints microseconds and achieved GB/s over the
algorithmic bytes.

    python profiles/point_grid_probe.py [--reps 20] [--only sample_bwd]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomosar2height_amd import _lib, ops                    # noqa: E402
from tomosar2height_amd.synthetic import berlin_tile        # noqa: E402
from tomosar2height_amd.tile import TileIndex               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--points", type=int, default=131072)
ap.add_argument("--only", default="")
ap.add_argument("--uniform", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
tile = TileIndex(berlin_tile(0, n_points=args.points, clustered=not args.uniform)["inputs"].to(dev), 256)
N = args.points


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / args.reps


for C, r in ((32, 256), (64, 256), (128, 128), (256, 64), (512, 32)):
    plane = torch.randn(1, r, r, C, device=dev).permute(0, 3, 1, 2)          # channels_last storage
    feat = torch.randn(N, C, device=dev)
    rows = {}
    if args.only in ("", "sample_fwd"):
        rows["sample_fwd"] = (timed(lambda: ops.sample_plane(tile, plane)), 4 * C * N + 8 * N + 4 * C * r * r)
    if args.only in ("", "sample_bwd"):
        gplane = torch.empty(1, r, r, C, device=dev)
        lvl = tile.level(r)
        ws_bytes = _lib.load().t2h_sample_bwd_workspace_bytes(1, N, tile.nbits, lvl, C)
        ws = _lib.workspace(ws_bytes, dev)

        def bwd():
            _lib.call("t2h_sample_bwd", _lib.ptr(feat), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), 1, N, tile.nbits, lvl,
                      C, _lib.ptr(gplane), _lib.ptr(ws), ws_bytes, _lib.stream())
        rows["sample_bwd"] = (timed(bwd), 4 * C * N + 8 * N + 4 * C * r * r)

        # the same product through the transposed matrix (built once per tile and level, TileIndex.sample_adjoint)
        def build():
            tile._adjoint.clear()
            tile.sample_adjoint(lvl)
        rows["adjoint build"] = (timed(build), 2 * 9 * 8 * N + 8 * r * r + 32 * N)
        offsets, entries = tile.sample_adjoint(lvl)
        gplane2 = torch.empty(1, r, r, C, device=dev)

        def bwd_adj():
            _lib.call("t2h_sample_bwd_adjoint", _lib.ptr(feat), _lib.ptr(offsets), _lib.ptr(entries), 1, tile.nbits, lvl, C, None,
                      _lib.ptr(gplane2), _lib.stream())
        rows["sample_bwd adjoint"] = (timed(bwd_adj), 4 * C * N + 32 * N + 4 * C * r * r)
        bwd(); bwd_adj(); torch.cuda.synchronize()
        print("   adjoint == gather/partials bitwise:", bool(torch.equal(gplane, gplane2)),
              " max |diff| %.3g" % float((gplane - gplane2).abs().max()))
    if args.only in ("", "segmean_fwd"):
        rows["segmean_fwd"] = (timed(lambda: ops.rasterise_mean(tile, feat, r, True)), 4 * C * N + 4 * N + 4 * C * r * r)
    for k, (us, nbytes) in rows.items():
        print(f"{k:18s} C={C:4d} r={r:4d} {us:8.1f} us {nbytes / us / 1e3:8.1f} GB/s  ({nbytes / us / 8e6 * 100:5.1f} % of 8 TB/s)")
