#!/usr/bin/env python3
"""r06: the three products around the r = 256 level's sum matrix -- forward Y [M, 2752] = X [M, 64] W^T, data
gradient dX = dY W, weight gradient dW = dY^T X -- timed alone at M = B x 65 536 pixel rows for B = 1 (tile by tile) and B = 4 (the
coalesced default), as the deferred point update calls them (mlp.linear_fwd_ / linear_dgrad_ / linear_wgrad_ with bx3=True routing).

    [T2H_BX3_PERSIST_WGS=8192 ...] python profiles/level256_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomosar2height_amd import mlp                      # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, reps=12):
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


K, N = 64, 2752
print("env:", {k: v for k, v in os.environ.items() if k.startswith("T2H_")})
for B in (1, 4):
    M = B * 65536
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / 8
    b = torch.randn(N, device=dev, generator=g)
    y = torch.empty(M, N, device=dev)
    dy = torch.randn(M, N, device=dev, generator=g)
    dx = torch.empty(M, K, device=dev)
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    nbytes = 4 * (M * K + M * N + N * K)
    def wgrad(split):
        def go():
            mlp._GEMM_BX3_WGRAD = split
            mlp.linear_wgrad_(dy, x, dw, db)
        return go
    for name, fn in (("forward  Y = X W^T", lambda: mlp.linear_fwd_(x, w, b, y, bx3=True)),
                     ("dgrad    dX = dY W", lambda: mlp.linear_dgrad_(dy, w, dx, bx3=True)),
                     ("wgrad    fp32 MFMA", wgrad(False)), ("wgrad    split TN (r06)", wgrad(True))):
        us = timed(fn)
        print(f"B = {B}  {name:22s} {us:9.1f} us = {us / B:8.1f} us per tile   {nbytes / us / 1e6:6.2f} TB/s   "
              f"{2.0 * M * N * K / us / 1e6:7.1f} TF")
