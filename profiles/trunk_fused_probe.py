"""r06: where the one-launch trunk forward (t2h_trunk_fused_fwd) spends its time -- parts switched off one at a time in a lab build
(-DT2H_TRUNK_ABLATE, profiles/coresidency_lab_build.py's build_variant; results are then wrong by design, only the time counts).

    T2H_LIBRARY=profiles/_lab/libt2h_trunk_ablate.so python profiles/trunk_fused_probe.py [tiles per batch = 4] [points per tile = 131072]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
from detinit import det_init_
from tomosar2height_amd import mlp
from tomosar2height_amd.encoder.pointnet import LocalPoolPointnet
from tomosar2height_amd.synthetic import berlin_tile
from tomosar2height_amd.tile import TileIndex

nb_tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_points = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
dev = torch.device("cuda:0")
enc = det_init_(LocalPoolPointnet(feature_dim=32, dim=3, hidden_dim=32, scatter_type="max", unet_type="alto",
                                  unet_kwargs=dict(depth=2, merge_mode="concat", start_filts=8), plane_resolution=256), seed=5).to(dev)
ps = [enc.fc_pos.weight, enc.fc_pos.bias]
for b in enc.blocks:
    ps += [b.fc_0.weight, b.fc_0.bias, b.fc_1.weight, b.fc_1.bias, b.shortcut.weight]
ps = [p.detach() for p in ps + [enc.fc_c.weight, enc.fc_c.bias]]
blocks = [ps[2 + 5 * i: 7 + 5 * i] for i in range(5)]
rag = (-0.10, 0.06, -0.04, 0.08)
clouds = [berlin_tile(seed=60 + i, n_points=int(n_points * (1 + rag[i % 4])))["inputs"].to(dev) for i in range(nb_tiles)]
tile = TileIndex(clouds if nb_tiles > 1 else clouds[0], 256)
m = tile.pts.shape[0]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


def run(fused, stride=0, greedy=False):
    def go():
        mlp._TRUNK_FUSED, mlp._TRUNK_FUSED_STRIDE, mlp._TRUNK_UNIT_BOUNDS, mlp._TRUNK_FUSED_MIN_ROWS = fused, stride, greedy, 0
        return mlp._trunk_forward_fused(tile, tile.pts, ps[0], ps[1], blocks, ps[-2], ps[-1])
    return go


print(f"{nb_tiles} tile(s), {m} rows; library {os.environ.get('T2H_LIBRARY', 'shipped')}")
print(f"  five per-block launches                      {timed(run(False)):8.1f} us")
used = tile.trunk_unit_list().cpu()
rows_per = (used[:, 1] - used[:, 0]).float()
print(f"  greedy units: {len(used)}, mean {rows_per.mean():.1f} rows, {int((rows_per > 128).sum())} longer than a tile")
print(f"  one launch, greedy units (default)             {timed(run(True, 0, True)):8.1f} us")


def build_units():
    tile._adjoint.pop("trunk_units", None)
    return tile.trunk_units()


if True:
    if os.environ.get("T2H_PROBE_DEFAULT_ONLY") != "1":
        print(f"  building the unit list (once per tile index)   {timed(build_units):8.1f} us")
for stride in (() if os.environ.get("T2H_PROBE_DEFAULT_ONLY") == "1" else (112, 96, 128)):      # (1: the PMC passes of pmc_trunk.sh)
    print(f"  one launch, stride {stride:3d}                        {timed(run(True, stride)):8.1f} us")
if "ablate" in os.environ.get("T2H_LIBRARY", ""):
    names = {1: "no pooling", 2: "no global stores", 4: "weights staged once", 8: "no MFMAs", 16: "unit bounds without the cell lookups",
             3: "no pooling, no stores", 7: "no pooling, no stores, weights once", 15: "nothing but the skeleton",
             12: "weights once, no MFMAs", 6: "no stores, weights once", 32: "pooling per ROW (the first form)"}
    for abl, what in names.items():
        print(f"  one launch, greedy, {what:30s} {timed(run(True, (abl << 8), True)):8.1f} us")
