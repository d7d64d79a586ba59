#!/usr/bin/env python3
"""Launch the scatter-reduce / sample kernels of the coarsest ALTO level (C = 512, r = 32, N = 131072 clustered points) in
isolation, so that a rocprofv3 --pmc pass attributes HBM traffic to ops whose kernel symbol is shared across levels:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_fetch -- python3 profiles/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_write -- python3 profiles/pmc_probe.py
    python profiles/collect_pmc.py --bench F W --probe out_fetch out_write      # -> profiles/pmc_traffic.json
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomosar2height_amd import ops                           # noqa: E402
from tomosar2height_amd.synthetic import berlin_tile          # noqa: E402
from tomosar2height_amd.tile import TileIndex                 # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from probe_manifest import REPS                          # noqa: E402
dev = torch.device("cuda:0")
# r06: as many tiles per launch as the Trainer coalesces by default (bench.py's launches are then the probe's launches)
B = int(os.environ.get("T2H_PROBE_TILES", os.environ.get("T2H_COALESCE_TILES", "4")))
tile = TileIndex(torch.cat([berlin_tile(i)["inputs"] for i in range(B)], 0).to(dev), 256)
M = tile.n_points
x512 = torch.randn(M, 512, device=dev)
for _ in range(REPS):
    ops.rasterise_mean(tile, x512, 32, channels_last=True)    # segmean_cells_kernel + segmean_finalize_kernel
plane = torch.randn(B, 512, 32, 32, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
for _ in range(REPS):
    out = ops.sample_plane(tile, plane)                       # sample_fwd_kernel<4>
    out.backward(x512)                                        # sample_bwd_cells_kernel + sample_bwd_gather9_kernel
# r03: the deferred point update's scatter-reduce pair on the widest hidden tensor (N x 1024), finest resolution
from tomosar2height_amd import deferred                      # noqa: E402
x1024 = torch.randn(M, 1024, device=dev)
planes = {lv: torch.empty(B * (256 >> lv) ** 2, 1024, device=dev) for lv in range(4)}
for _ in range(REPS):
    deferred._segsum_into(tile, x1024, 0, planes[0])          # segmean_fwd_kernel<4, false>
    for lv in range(3):
        deferred._sumpool_into(tile, planes[lv], lv, planes[lv + 1])      # plane_sumpool2x2_kernel
grads = [(torch.randn_like(p), lv) for lv, p in planes.items()]
for _ in range(REPS):
    deferred._gather(tile, grads, 1024, mask=x1024)          # segsum_bwd_multi_kernel<4>
# r03: the hidden activations that stay on chip -- interpolation + ReLU + per-cell sums + sign bits in one pass over the cells
# (t2h_sample_relu_cellsums), and its backward twin (t2h_sample_bwd_from_sums, walk form), widest level: 1024 x 32^2
from tomosar2height_amd import _lib                           # noqa: E402
import ctypes                                                 # noqa: E402
q = torch.randn(B * 32 * 32, 1024, device=dev)
sums = torch.empty(B * 256 * 256, 1024, device=dev)
bits = torch.empty(M * 4 * 4, dtype=torch.int64, device=dev)
lv32 = tile.level(32)
order = torch.empty(_lib.load().t2h_cell_order_len(tile.B, tile.nbits, lv32), dtype=torch.int32, device=dev)
for _ in range(REPS):                                         # r05: the level's longest-first dispatch order (cell_order_kernel)
    _lib.call("t2h_cell_order_build", _lib.ptr(tile.off0), tile.B, tile.nbits, lv32, _lib.ptr(order), _lib.stream())
pooled = torch.empty(B * 128 * 128, 1024, device=dev)
for _ in range(REPS):                                         # as deferred.py calls it: finest sums + the pooled ones, ordered
    _lib.call("t2h_sample_relu_cellsums_ordered", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N,
              tile.nbits, lv32, 0, 1024, sums.data_ptr(), sums.stride(0), pooled.data_ptr(), pooled.stride(0), _lib.ptr(bits),
              _lib.ptr(order), _lib.stream())
arr, lvs, lds = deferred._plane_args(grads)
ws_bytes = _lib.ws_bytes("t2h_sample_bwd_workspace_bytes", tile.B, tile.N, tile.nbits, lv32, 1024)
ws = _lib.workspace(ws_bytes, dev)
dq = torch.empty(B * 32 * 32, 1024, device=dev)
for _ in range(REPS):
    _lib.call("t2h_sample_bwd_from_sums_ordered", arr, lvs, lds, len(grads), _lib.ptr(tile.cell), _lib.ptr(bits), 1, _lib.ptr(tile.pts),
              tile.dim, _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits, lv32, 1024, _lib.ptr(dq), _lib.ptr(ws), ws_bytes,
              _lib.ptr(order), _lib.stream())
torch.cuda.synchronize()
print("pmc_probe done")
