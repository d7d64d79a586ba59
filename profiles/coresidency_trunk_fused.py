"""r06: the one-launch trunk forward beside the split convolutions, launch by launch (the stalled windows of coresidency_trace.py name
`t2h_trunk_fused_fwd` as the first call that differs, 3-9 windows in 200-400 at N = 131 072; this says WHAT differs).

Victim: mlp._trunk_forward_one_launch on one tile of N points (fresh outputs per launch), K launches per trial on stream A.
Competitor on stream B: the recorded `bx3` calls of a whole model backward (what shares the chip in the tile pipeline).
Every output tensor of every launch is compared with the launch alone; for a launch that differs: which tensors, which rows, where
those rows sit inside their work unit and inside a wave.

    [T2H_LIBRARY=...] python profiles/coresidency_trunk_fused.py [launches=2000] [points=131072] [passes=3]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
from detinit import det_init_, synth_cloud
from tomosar2height_amd import TomoSAR2Height, _lib, mlp
from tomosar2height_amd.config import berlin_config
from tomosar2height_amd.tile import TileIndex
from tomosar2height_amd.trainer import Trainer

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
points = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
K = 8
dev = torch.device("cuda:0")
tiles = [{"inputs": synth_cloud(points, seed=700 + i).to(dev),
          "dsm": (torch.rand(1, 512, 512, generator=torch.Generator().manual_seed(i)) * 30).to(dev)} for i in range(3)]
model = det_init_(TomoSAR2Height(berlin_config()), seed=15).to(dev)
model.set_channels_last(True)
tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=dev, optimize_every=100, use_cloud=True)
tr.pipeline_tiles = False
tr.overlap_wgrad = tr.overlap_conv_wgrad = False
tr.coalesce_tiles = 1
tr.train_step(tiles[0])
tr.train_step(tiles[1])
torch.cuda.synchronize()
A, B = torch.cuda.Stream(), torch.cuda.Stream()
orig = _lib.call
rec = []


def recording(name, *a, **k):
    rec.append((name, a))
    return orig(name, *a)


with torch.cuda.stream(B):
    _lib.call = recording
    try:
        with tr._own_cache():
            l1, ce = tr._losses(tiles[2], 0.0001)
        n_fwd = len(rec)
        tr._backward(l1 + ce)
    finally:
        _lib.call = orig
torch.cuda.synchronize()
comp = [c for c in rec[n_fwd:] if "bx3" in c[0]]

enc = model.point_encoder
ps = [enc.fc_pos.weight, enc.fc_pos.bias]
for b in enc.blocks:
    ps += [b.fc_0.weight, b.fc_0.bias, b.fc_1.weight, b.fc_1.bias, b.shortcut.weight]
ps = [p.detach() for p in ps + [enc.fc_c.weight, enc.fc_c.bias]]
blocks = [ps[2 + 5 * i: 7 + 5 * i] for i in range(len(enc.blocks))]
tile = TileIndex(synth_cloud(points, seed=733).to(dev), enc.reso_plane)           # default stream: never one of B's recycled blocks
units = tile.trunk_unit_list().cpu()
starts = units[:, 0].contiguous()
mlp._TRUNK_FUSED, mlp._TRUNK_FUSED_MIN_ROWS = True, 0


def flat(res):
    out, nets, pooled, hrs, winners = res
    names = ["c"] + [f"net{i}" for i in range(len(nets))] + [f"pooled{i}" for i in range(len(pooled))] + \
            [f"hr{i}" for i in range(len(hrs))] + [f"winner{i}" for i in range(len(winners))]
    return names, [out] + list(nets) + list(pooled) + list(hrs) + list(winners)


def launch():
    return flat(mlp._trunk_forward_one_launch(tile, tile.pts, ps[0], ps[1], blocks, ps[-2], ps[-1]))[1]


main = torch.cuda.current_stream()
A.wait_stream(main)
with torch.cuda.stream(A):
    names = flat(mlp._trunk_forward_one_launch(tile, tile.pts, ps[0], ps[1], blocks, ps[-2], ps[-1]))[0]
    ref = launch()
    again = launch()
torch.cuda.synchronize()
assert all(a is None or torch.equal(a, b) for a, b in zip(ref, again)), "the launch alone is not reproducible"
print(f"library {os.environ.get('T2H_LIBRARY', 'shipped')}; one tile of {points} points = {tile.pts.shape[0]} rows in {len(units)} units; competitor: "
      f"{len(comp)} bx3 calls x {passes} passes per trial; {launches} launches, {K} per trial")
bad, shown = 0, 0
for t in range(-(-launches // K)):
    main = torch.cuda.current_stream()
    A.wait_stream(main)
    B.wait_stream(main)
    with torch.cuda.stream(B):
        for _ in range(passes):
            for cn, ca in comp:
                orig(cn, *ca)
    with torch.cuda.stream(A):
        res = [launch() for _ in range(K)]
    torch.cuda.synchronize()
    for got in res:
        diff = [(n, g, r) for n, g, r in zip(names, got, ref) if g is not None and not torch.equal(g, r)]
        if not diff:
            continue
        bad += 1
        if shown >= 6:
            continue
        shown += 1
        print(f"  launch differs: tensors {[n for n, _, _ in diff]}", flush=True)
        for n, g, r in diff[:4]:
            g2, r2 = g.reshape(g.shape[0], -1), r.reshape(r.shape[0], -1)
            d = g2 != r2
            rows = torch.nonzero(d.any(1)).flatten().cpu()
            u = torch.searchsorted(starts, rows, right=True) - 1
            pos = rows - starts[u]
            cols = torch.nonzero(d.any(0)).flatten().tolist()
            print(f"    {n}: {len(rows)} rows, units {sorted(set(u.tolist()))[:6]}, rows-in-unit {pos.tolist()[:24]}, unit lengths "
                  f"{[int(units[i, 1] - units[i, 0]) for i in sorted(set(u.tolist()))[:6]]}, columns {cols[:8]}..{cols[-4:]} ({len(cols)})", flush=True)
            i = int(rows[0])
            print(f"      row {i}: here {g2[i, cols[:6]].tolist()} alone {r2[i, cols[:6]].tolist()}", flush=True)
print(f"t2h_trunk_fused_fwd: {bad} of {-(-launches // K) * K} launches differ from the launch alone")
