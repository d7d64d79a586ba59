#!/usr/bin/env python3
"""The two on-chip hidden-activation kernels of the deferred ALTO point update at the widest level (C = 1024 hidden channels,
sampling resolution r = 32; N = 131072 clustered points) in isolation, for rocprofv3 --pmc passes over the SQ counters
(is `t2h_sample_relu_cellsums` / `t2h_sample_bwd_from_sums` issue-bound or memory-bound?):

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
              SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d out -- python3 profiles/issue_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomosar2height_amd import _lib, deferred                 # noqa: E402
from tomosar2height_amd.synthetic import berlin_tile          # noqa: E402
from tomosar2height_amd.tile import TileIndex                 # noqa: E402

REPS = int(os.environ.get("REPS", "3"))
dev = torch.device("cuda:0")
tile = TileIndex(berlin_tile(1000)["inputs"].to(dev), 256)
M = tile.n_points
for c2, r in ((1024, 32), (512, 64), (256, 128)):
    lv = tile.level(r)
    q = torch.randn(r * r, c2, device=dev)
    planes = {l: torch.empty((256 >> l) ** 2, c2, device=dev) for l in range(lv + 1)}
    grads = [(torch.randn_like(p), l) for l, p in planes.items()]
    bits = torch.empty(M * (c2 // 256) * 4, dtype=torch.int64, device=dev)
    for _ in range(REPS):
        _lib.call("t2h_sample_relu_cellsums2", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N,
                  tile.nbits, lv, 0, c2, planes[0].data_ptr(), planes[0].stride(0), planes[1].data_ptr(), planes[1].stride(0),
                  _lib.ptr(bits), _lib.stream())
    arr, lvs, lds = deferred._plane_args(grads)
    ws_bytes = _lib.ws_bytes("t2h_sample_bwd_workspace_bytes", tile.B, tile.N, tile.nbits, lv, c2)
    ws = _lib.workspace(ws_bytes, dev)
    dq = torch.empty(r * r, c2, device=dev)
    for _ in range(REPS):
        _lib.call("t2h_sample_bwd_from_sums", arr, lvs, lds, len(grads), _lib.ptr(tile.cell), _lib.ptr(bits), 1, _lib.ptr(tile.pts),
                  tile.dim, _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits, lv, c2, _lib.ptr(dq), _lib.ptr(ws), ws_bytes,
                  _lib.stream())
torch.cuda.synchronize()
print("issue_probe done")
