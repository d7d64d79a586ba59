"""r05: which C-ABI call of a tile's backward, replayed on a second stream, changes the result of the on-chip walk (fixed inputs)
running beside it?  python profiles/coresidency_replay.py <trials> bwd [nobits,nopool,noorder]   (T2H_LIBRARY=<old .so> for a control).
Findings: profiles/r05_coresidency.txt; the permanent form is tests/test_coresidency.py."""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import torch
from detinit import det_init_, synth_cloud
from tomosar2height_amd import TomoSAR2Height, _lib
from tomosar2height_amd.config import berlin_config
from tomosar2height_amd.trainer import Trainer
from tomosar2height_amd.tile import TileIndex
dev = torch.device("cuda:0")
reps = int(sys.argv[1]); MODE = sys.argv[3] if len(sys.argv) > 3 else ""
cfg = berlin_config()
tiles = [{"inputs": synth_cloud(40000, seed=700 + i).to(dev),
          "dsm": (torch.rand(1, 512, 512, generator=torch.Generator().manual_seed(i)) * 30).to(dev)} for i in range(3)]
# the walk's fixed inputs and every output buffer it will ever need, allocated FIRST
tile = TileIndex(synth_cloud(40000, seed=703).to(dev), 128)
level, C = 3, 1024
r = 128 >> level
q = torch.randn(r * r, C, device=dev)
rows = tile.B << (2 * tile.nbits)
def outs():
    return (torch.zeros(rows, C, device=dev), torch.zeros(rows // 4, C, device=dev),
            torch.zeros(tile.n_points * (C // 256) * 4, dtype=torch.int64, device=dev))
order = tile.cell_order(level)
def walk(o):
    _lib.call("t2h_sample_relu_cellsums_ordered", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits,
              level, 0, C, o[0].data_ptr(), C, (o[1].data_ptr() if 'nopool' not in MODE else None), (C if 'nopool' not in MODE else 0), (o[2].data_ptr() if 'nobits' not in MODE else None), (_lib.ptr(order) if 'noorder' not in MODE else None), _lib.stream())
ref = outs(); walk(ref)
res = [outs() for _ in range(4)]
torch.cuda.synchronize()
A, Bs = torch.cuda.Stream(), torch.cuda.Stream()
model = det_init_(TomoSAR2Height(cfg), seed=15).to(dev)
model.set_channels_last(True)
tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=dev, optimize_every=100, use_cloud=True)
tr.pipeline_tiles = False
tr.overlap_wgrad = tr.overlap_conv_wgrad = False
tr.train_step(tiles[0]); tr.train_step(tiles[1]); torch.cuda.synchronize()
rec = []
orig = _lib.call
def recording(name, *a, **k):
    rec.append((name, a)); return orig(name, *a)
phase = sys.argv[2] if len(sys.argv) > 2 else "bwd"
with torch.cuda.stream(Bs):
    if phase == "fwd": _lib.call = recording
    with tr._own_cache():
        l1, ce = tr._losses(tiles[2], 0.0001)
    loss = l1 + ce
    _lib.call = recording if phase in ("bwd", "fwd") else orig
    if phase == "fwd": _lib.call = orig
    if phase == "bwd": _lib.call = recording
    tr._backward(loss)
    _lib.call = orig
torch.cuda.synchronize()
names = []
for n, a in rec:
    if n not in names: names.append(n)
print(len(rec), "recorded calls,", len(names), "distinct")
def trial(calls, n_iter):
    bad = 0
    for it in range(n_iter):
        A.wait_stream(torch.cuda.current_stream()); Bs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(Bs):
            for _ in range(3):
                for n, a in calls: orig(n, *a)
        with torch.cuda.stream(A):
            for o in res: walk(o)
        torch.cuda.synchronize()
        for o in res:
            if not (torch.equal(ref[0], o[0]) and torch.equal(ref[1], o[1]) and torch.equal(ref[2], o[2])): bad += 1
    return bad
print("all calls in order:", trial(rec, reps), "of", 4 * reps)
for n in names:
    if "bx3" not in n: continue
    calls = [c for c in rec if c[0] == n]
    mult = max(1, 40 // len(calls))
    b = trial(calls * mult, reps)
    print(f"{n:44s} x{len(calls):3d}: mismatching walks {b} of {4 * reps}" + ("   <<<<" if b else ""))
