import sys, time, cProfile, pstats, io, torch
sys.path.insert(0, '.')
from tomosar2height_amd import TomoSAR2Height, _lib
from tomosar2height_amd.config import berlin_config
from tomosar2height_amd.synthetic import berlin_tile
from tomosar2height_amd.trainer import Trainer
from tomosar2height_amd.optim import FlatAdamW
dev = torch.device("cuda:0")
cfg = berlin_config(use_image=False)
torch.manual_seed(0)
model = TomoSAR2Height(cfg).to(dev)
model.set_channels_last(True)
opt = FlatAdamW(model.parameters(), lr=1e-4)
tr = Trainer(model, opt, device=dev, optimize_every=64, use_cloud=True, use_image=False)
tiles = []
for i in range(4):
    t = berlin_tile(seed=i, n_points=131072)
    tiles.append({k: t[k].to(dev) for k in ("inputs", "dsm")})
for i in range(6): tr.train_step(tiles[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10): tr.train_step(tiles[i % 4])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("issue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile(); pr.enable()
for i in range(10): tr.train_step(tiles[i % 4])
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
