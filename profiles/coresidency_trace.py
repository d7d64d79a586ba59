"""r06: WHERE does a coalesced, pipelined window first differ from the step-synchronised one?

coresidency_soak.py counts the windows that differ (1 of 150 cloud-only coalesced windows: the loss of the third micro-batch);
coresidency_hunt.py replays one forward beside one backward and found nothing in 5 000 trials -- so the difference needs the
LIVE window.  This runs the live window with an integer checksum of every tensor a forward C-ABI call was handed, taken right
after the call on the call's stream (Tensor.data_ptr is wrapped to learn pointer -> extent; results land in a preallocated table,
no allocation, no host sync), and names the first (call, tensor) whose checksum differs from the step-synchronised window's.
Entries that differ between two step-synchronised windows (scratch whose order is free, e.g. the cell-order lists) are masked.

    python profiles/coresidency_trace.py <windows> [image] [prepared] [backward]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
from detinit import det_init_, synth_cloud
from tomosar2height_amd import TomoSAR2Height, _lib
from tomosar2height_amd.config import berlin_config
from tomosar2height_amd.trainer import Trainer

windows = int(sys.argv[1])
image, ahead, with_bwd = "image" in sys.argv[2:], "prepared" in sys.argv[2:], "backward" in sys.argv[2:]
points = int(os.environ.get("T2H_TRACE_POINTS", "40000"))
dev = torch.device("cuda:0")
ag = ctypes.CDLL(os.path.join(ROOT, "profiles", "_lab", "libaggr.so"))
ag.aggr_checksum.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
tiles = [{"inputs": synth_cloud(points, seed=700 + i).to(dev),
          "dsm": (torch.rand(1, 512, 512, generator=torch.Generator().manual_seed(i)) * 30).to(dev)} for i in range(9)]
if image:
    for i, t in enumerate(tiles):
        t["image"] = torch.randn(1, 3, 512, 512, generator=torch.Generator().manual_seed(40 + i)).to(dev)
cfg = berlin_config(use_image=image)

CALLS, ARGS = 4096, 24
table = torch.zeros(CALLS, ARGS, dtype=torch.int64, device=dev)
table_ptr = table.data_ptr()
seen, names, state = [], [], {"on": False, "k": 0}
raw_data_ptr = torch.Tensor.data_ptr
raw_call = _lib.call


def extent_bytes(t):
    if t.numel() == 0:
        return 0
    return (sum((s - 1) * st for s, st in zip(t.shape, t.stride())) + 1) * t.element_size()


def data_ptr(t):
    p = raw_data_ptr(t)
    if state["on"] and t.is_cuda:
        seen.append((p, extent_bytes(t)))
    return p


def call(name, *a, **k):
    raw_call(name, *a, **k)
    if not state["on"]:
        return
    i = state["k"]
    state["k"] += 1
    uniq = list(dict.fromkeys(seen))                 # in the order the call's code asked for them (the same in every window)
    del seen[:]
    if len(names) <= i:
        names.append((k.get("tag") or name, name, [n for _, n in uniq]))
    assert i < CALLS and len(uniq) <= ARGS, (i, len(uniq))
    s = _lib.stream()
    for j, (p, n) in enumerate(uniq):
        if n >= 4:
            ag.aggr_checksum(p, n, table_ptr + 8 * (i * ARGS + j), s)


torch.Tensor.data_ptr = data_ptr
_lib.call = call


def run(stepsync):
    model = det_init_(TomoSAR2Height(cfg), seed=15).to(dev)
    model.set_channels_last(True)
    tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=dev, optimize_every=100, use_cloud=True, use_image=image)
    tr.coalesce_tiles = 4
    side = torch.cuda.Stream() if ahead else None
    prep = (lambda t: tr.prepare(t, side)) if ahead else (lambda t: t)
    losses, inner, inner_bwd = [], tr._losses, tr._backward
    torch.cuda.synchronize()
    table.zero_()
    state["k"] = 0
    del seen[:]

    def rec(data, thr):
        state["on"] = True
        try:
            l1, ce = inner(data, thr)
        finally:
            state["on"] = False
        losses.append(l1.detach())
        return l1, ce

    def bwd(loss):
        state["on"] = with_bwd
        try:
            return inner_bwd(loss)
        finally:
            state["on"] = False
    tr._losses, tr._backward = rec, bwd
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
    return table[:state["k"]].clone(), [float(x) for x in losses]


gold, gl = run(True)
gold2, _ = run(True)
free = gold != gold2
print(f"{gold.shape[0]} traced calls per window ({'forward + backward' if with_bwd else 'forward'}), {int((gold != 0).sum())} tensors; "
      f"{int(free.sum())} entries differ between two step-synchronised windows (masked):", flush=True)
for k in sorted(set(torch.nonzero(free)[:, 0].tolist())):
    print(f"    call {k}: {names[k][0]} ({names[k][1]}) tensors {torch.nonzero(free[k]).flatten().tolist()}")
bad, firsts = 0, {}
for it in range(windows):
    got, ls = run(False)
    d = torch.nonzero((got != gold) & ~free)
    if len(d) or ls != gl:
        bad += 1
        rows = sorted(set(d[:, 0].tolist()))
        print(f"window {it}: losses equal {[a == b for a, b in zip(ls, gl)]}; {len(rows)} calls differ", flush=True)
        for k in rows[:6]:
            js = d[d[:, 0] == k][:, 1].tolist()
            print(f"    call {k}: {names[k][0]} ({names[k][1]}): tensors {js} of {len(names[k][2])} (bytes {[names[k][2][j] for j in js]})", flush=True)
        if rows:
            key = f"{rows[0]}:{names[rows[0]][0]}"
            firsts[key] = firsts.get(key, 0) + 1
print(f"{'cloud+image' if image else 'cloud-only'}{', prepared' if ahead else ''}, coalesced, N = {points}: {bad} of {windows} windows differ; "
      f"first differing call -> windows: {firsts or 'none'}", flush=True)
