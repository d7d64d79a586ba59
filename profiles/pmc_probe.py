#!/usr/bin/env python3
"""Launch the roofline-relevant t2h kernels in isolation (BASELINE.json configs[1] shapes, N = 131072) so that a
rocprofv3 --pmc pass can attribute HBM traffic per launch:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_fetch -- python3 profiles/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_write -- python3 profiles/pmc_probe.py
    python profiles/collect_pmc.py out_fetch out_write      # -> profiles/pmc_traffic.json

Each op runs REPS times; inputs are > 256 MiB apart in time?  No: inputs stay resident, so re-reads that hit the
256 MiB Infinity Cache are still counted by the fabric-side counters (MI355X_MICROARCH.md, HBM section)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomosar2height_amd import grid, mlp, ops                # noqa: E402
from tomosar2height_amd.synthetic import berlin_tile          # noqa: E402
from tomosar2height_amd.tile import TileIndex                 # noqa: E402

REPS = 3
dev = torch.device("cuda:0")
tile = TileIndex(berlin_tile(0)["inputs"].to(dev), 256)
M = tile.n_points
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*shape):
    return torch.randn(*shape, device=dev, generator=g)


x512, x1024, x32 = rnd(M, 512), rnd(M, 1024), rnd(M, 32)
w = rnd(1024, 512) / 22.0
b = rnd(1024)
y = torch.empty(M, 1024, device=dev)
dw, db = torch.empty(1024, 512, device=dev), torch.empty(1024, device=dev)
torch.cuda.synchronize()
w2, b2, y2 = rnd(512, 1024) / 32.0, rnd(512), torch.empty(M, 512, device=dev)
for _ in range(REPS):
    mlp.linear_fwd_(x512, w, b, y, relu_out=True)             # gemm_dma_kernel (LDS-DMA staged NT), K=512 -> N=1024
for _ in range(REPS):
    mlp.linear_fwd_(x1024, w2, b2, y2)                        # gemm_dma_kernel, K=1024 -> N=512
for _ in range(REPS):
    mlp.linear_dgrad_(x1024, w, x512.clone(), mask=x512)      # gemm_dma_nn_kernel (LDS-DMA staged NN), dX[M,512]
for _ in range(REPS):
    mlp.linear_dgrad_(x512, w2, x1024.clone(), mask=x1024)    # same kernel, dX[M,1024]
for _ in range(REPS):
    mlp.linear_wgrad_(x1024, x512, dw, db)                    # gemm_kernel<128,128,2,2,false,false> + reduce_slabs
dw2, db2 = torch.empty(512, 1024, device=dev), torch.empty(512, device=dev)
for _ in range(REPS):
    mlp.linear_wgrad_(x512, x1024, dw2, db2)                  # the mirrored shape dW[512,1024] (same kernel, 2nd in order)
for _ in range(REPS):
    ops.rasterise_mean(tile, x512, 32, channels_last=True)    # segmean_cells_kernel + segmean_finalize_kernel
for _ in range(REPS):
    ops.pool_max(tile, x32)                                   # pool_max_fwd_kernel<4>
plane = rnd(1, 512, 32, 32).contiguous(memory_format=torch.channels_last).requires_grad_(True)
for _ in range(REPS):
    out = ops.sample_plane(tile, plane)                       # sample_fwd_kernel<4>
    out.backward(x512)                                        # sample_bwd_cells_kernel + sample_bwd_gather9_kernel
# decoder-sized 3x3 convolutions (pixel.py:20-32 at 512 x 512): conv_rows_kernel<128,..,0> / <64,..,1> / conv_wgrad_kernel
cx, cg = rnd(1, 64, 512, 512).contiguous(memory_format=torch.channels_last), rnd(1, 128, 512, 512).contiguous(memory_format=torch.channels_last)
cw = (rnd(128, 64, 3, 3) / 24.0).contiguous(memory_format=torch.channels_last)
cy, cdx = grid._empty_cl(1, 128, 512, 512, dev), grid._empty_cl(1, 64, 512, 512, dev)
cdw, cdb = torch.empty_like(cw), torch.empty(128, device=dev)
for _ in range(REPS):
    grid.conv3x3_fwd_(cx, cw, cdb, cy, relu=True)
for _ in range(REPS):
    grid.conv3x3_dgrad_(cg, cw, cdx, mask=cx)
for _ in range(REPS):
    grid.conv3x3_wgrad_(cg, cx, cdw, cdb)
torch.cuda.synchronize()
print("pmc_probe done")
