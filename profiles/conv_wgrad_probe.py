#!/usr/bin/env python3
"""The weight gradient of the decoder's widest 3x3 layer (64 -> 128 channels on the 512 x 512 plane, pixel.py:20-32) in
isolation, for the traffic-against-time A/B of its split count:

    T2H_CONV_WGRAD_WGS=<workgroups> rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 profiles/conv_wgrad_probe.py
    (the same with WRITE_SIZE, and with --kernel-trace --stats for the durations)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomosar2height_amd import grid                          # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
cl = torch.channels_last
x = torch.randn(1, 64, 512, 512, generator=g).to(dev).contiguous(memory_format=cl)
gy = torch.randn(1, 128, 512, 512, generator=g).to(dev).contiguous(memory_format=cl)
dw = torch.empty(128, 64, 3, 3, device=dev).contiguous(memory_format=cl)
db = torch.empty(128, device=dev)
for _ in range(5):
    grid.conv3x3_wgrad_(gy, x, dw, db)
torch.cuda.synchronize()
print("conv_wgrad_probe done")
