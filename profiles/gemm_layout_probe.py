#!/usr/bin/env python3
"""Is the r = 256 level product (65536 rows x 2752 stacked columns, 250 us = 3 TB/s in either direction) bound by the LAYOUT of the
sum matrix?  Times t2h_gemm_bx3 on column blocks of the same height read / written densely (row stride = block width) and as slices of
the 2752-wide matrix.     python profiles/gemm_layout_probe.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomosar2height_amd import _lib, grid, mlp  # noqa: E402

dev = torch.device("cuda:0")
M = 65536
big = torch.randn(M, 2752, device=dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for k in (2752, 1024, 512, 256):
    w = torch.randn(64, k, device=dev) / k ** 0.5
    y = torch.empty(M, 64, device=dev)
    for name, x in (("slice of 2752", big[:, :k]), ("dense", big[:, :k].contiguous())):
        us = timed(lambda: mlp._gemm_bx3(x, w, False, None, None, y, False, False, None))
        print(f"read  K={k:5d} -> N=64   {name:14s} {us:8.1f} us  {4 * M * k / us / 1e3:7.0f} GB/s")
for n in (2752, 1024, 512, 256):
    w = torch.randn(n, 64, device=dev) / 8
    x = torch.randn(M, 64, device=dev)
    for name, y in (("slice of 2752", big[:, :n]), ("dense", torch.empty(M, n, device=dev))):
        us = timed(lambda: mlp._gemm_bx3(x, w, False, None, None, y, False, False, None))
        print(f"write K=64 -> N={n:5d}   {name:14s} {us:8.1f} us  {4 * M * n / us / 1e3:7.0f} GB/s")
