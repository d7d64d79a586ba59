"""r05: how often does a pipelined window differ from the step-synchronised one?  (tests/test_coresidency.py, many times over)
    python profiles/coresidency_soak.py <windows> [image] [prepared] [coalesced] [fresh] [stall[=k]] [stallb=k] [points=N]
r06: `coalesced` = the Trainer's default (nine tiles: micro-batches of 1 + 4 + 4 whose forwards and backwards overlap); without it
every tile is issued by its own call (four tiles), as in r05.  `fresh` = every window on a model with its own weights, its
step-synchronised twin run AFTER it: a buffer that is read before it is filled then cannot happen to hold the previous window's copy
of the same values (which hid the unordered read of r06_coresidency.txt section 7 in 149 windows of 150)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests", os.path.join("tests", "golden")):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), p))
import torch
from detinit import det_init_, synth_cloud
from tomosar2height_amd import TomoSAR2Height
from tomosar2height_amd.config import berlin_config
from tomosar2height_amd.trainer import Trainer

windows = int(sys.argv[1])
image, ahead, coalesced = "image" in sys.argv[2:], "prepared" in sys.argv[2:], "coalesced" in sys.argv[2:]
fresh = "fresh" in sys.argv[2:]
# stall / stall=k: forward number k (default 1: the second) of every window starts 0.1 s late on its stream; stallb=k: backward k does
stall = next((int(a.split("=")[1]) if "=" in a else 1 for a in sys.argv[2:] if a == "stall" or a.startswith("stall=")), None)
stallb = next((int(a.split("=")[1]) for a in sys.argv[2:] if a.startswith("stallb=")), None)
points = next((int(a.split("=")[1]) for a in sys.argv[2:] if a.startswith("points=")), 40000)
dev = torch.device("cuda:0")
tiles = [{"inputs": synth_cloud(points, seed=700 + i).to(dev),
          "dsm": (torch.rand(1, 512, 512, generator=torch.Generator().manual_seed(i)) * 30).to(dev)} for i in range(9 if coalesced else 4)]
if image:
    for i, t in enumerate(tiles):
        t["image"] = torch.randn(1, 3, 512, 512, generator=torch.Generator().manual_seed(40 + i)).to(dev)
cfg = berlin_config(use_image=image)


def run(ahead, stepsync, seed=15):
    model = det_init_(TomoSAR2Height(cfg), seed=seed).to(dev)
    model.set_channels_last(True)
    tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=dev, optimize_every=100, use_cloud=True, use_image=image)
    tr.coalesce_tiles = 4 if coalesced else 1
    side = torch.cuda.Stream() if ahead else None
    prep = (lambda t: tr.prepare(t, side)) if ahead else (lambda t: t)
    losses, inner = [], tr._losses

    def rec(data, thr):
        if stall is not None and not stepsync and len(losses) == stall:
            torch.cuda._sleep(int(3e8))
        l1, ce = inner(data, thr)
        losses.append(l1.detach())
        return l1, ce
    tr._losses = rec
    inner_bwd, nb = tr._backward, [0]

    def bwd(loss):
        if stallb is not None and not stepsync and nb[0] == stallb:
            torch.cuda._sleep(int(3e8))
        nb[0] += 1
        return inner_bwd(loss)
    tr._backward = bwd
    nxt = prep(tiles[0])
    for i in range(len(tiles)):
        cur = nxt
        if i + 1 < len(tiles):
            nxt = prep(tiles[i + 1])
        tr.train_step(cur)
        if stepsync:
            torch.cuda.synchronize()
    tr.flush_gradients()
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, [float(x) for x in losses]


gold, gl = run(ahead, True)
bad = 0
for it in range(windows):
    got, ls = run(ahead, False, 15 + (it + 1) * fresh)
    if fresh:
        gold, gl = run(ahead, True, 16 + it)
    d = [k for k in gold if not torch.equal(got[k], gold[k])]
    if d or ls != gl:
        bad += 1
        tops = {}
        for k in d:
            tops[".".join(k.split(".")[:3])] = tops.get(".".join(k.split(".")[:3]), 0) + 1
        worst = max(d, key=lambda k: float((got[k] - gold[k]).abs().max() / (gold[k].abs().max() + 1e-30))) if d else None
        print(f"window {it}: losses equal {[a == b for a, b in zip(ls, gl)]}, {len(d)} of {len(gold)} gradients differ {tops}; worst {worst} "
              f"{float((got[worst] - gold[worst]).abs().max() / (gold[worst].abs().max() + 1e-30)) if worst else 0:.2e}", flush=True)
print(f"{'cloud+image' if image else 'cloud-only'}{', prepared' if ahead else ''}{', coalesced' if coalesced else ''}{', fresh weights per window' if fresh else ''}{f', forward {stall} stalled' if stall is not None else ''}{f', backward {stallb} stalled' if stallb is not None else ''}, N = {points}: "
      f"{bad} of {windows} windows differ", flush=True)
