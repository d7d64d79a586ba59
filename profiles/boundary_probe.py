#!/usr/bin/env python3
"""r06: what one optimizer boundary costs (trainer.optimizer_boundary, once per 64 tiles; 1 of the driver's 20 timed steps ends with
one).  Each part between device synchronisations: host wall time (issue + GPU) and the GPU time by events.

    python profiles/boundary_probe.py
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tomosar2height_amd import TomoSAR2Height, grid                # noqa: E402
from tomosar2height_amd.config import berlin_config               # noqa: E402
from tomosar2height_amd.optim import FlatAdamW                    # noqa: E402
from tomosar2height_amd.synthetic import berlin_tile              # noqa: E402
from tomosar2height_amd.trainer import Trainer                    # noqa: E402

dev = torch.device("cuda:0")
cfg = berlin_config()
torch.manual_seed(0)
model = TomoSAR2Height(cfg).to(dev)
model.set_channels_last(True)
opt = FlatAdamW(model.parameters(), lr=1e-4)
tr = Trainer(model, opt, device=dev, optimize_every=100000, use_cloud=True)
tiles = [{k: berlin_tile(seed=i)[k].to(dev) for k in ("inputs", "dsm")} for i in range(4)]


def window(n=8):
    for i in range(n):
        tr.train_step(tiles[i % 4])
    tr.flush_pipeline()
    torch.cuda.synchronize()


def part(name, fn):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    print(f"  {name:58s} wall {1e3 * (time.perf_counter() - t0):7.3f} ms   gpu {a.elapsed_time(b):7.3f} ms")


window(9)
tr.optimizer_boundary()                                            # every lazy buffer of the boundary exists
for rep in range(2):
    window()
    print(f"window {rep}: the boundary's parts, in its order (trainer.optimizer_boundary)")
    part("compose_cache.flush (back-propagate the composed maps)", lambda: tr.compose_cache.flush() if tr.compose_cache is not None else None)
    part("out_of_domain_total", lambda: model.out_of_domain_total())
    part("optimizer.step (FlatAdamW, one kernel)", lambda: opt.step())
    part("compose_cache.refresh", lambda: tr.compose_cache.refresh() if tr.compose_cache is not None else None)
    part("split_weights.refresh", lambda: grid.split_weights.refresh(stale_only=True))
    part("bucket.zero_ + accumulators", lambda: tr._reset_accumulators())
window()
print("the whole call after a window:")
part("optimizer_boundary()", tr.optimizer_boundary)
window()
part("optimizer_boundary() again", tr.optimizer_boundary)
