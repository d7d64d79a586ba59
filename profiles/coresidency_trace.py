"""r06: WHERE does a coalesced, pipelined window first differ from the step-synchronised one?

coresidency_soak.py counts the windows that differ (1 of 150 cloud-only coalesced windows: the loss of the third micro-batch);
coresidency_hunt.py replays one forward beside one backward and found nothing in 5 000 trials -- so the difference needs the
LIVE window.  This runs the live window with an integer checksum of every tensor a forward C-ABI call was handed, taken right
after the call on the call's stream (Tensor.data_ptr is wrapped to learn pointer -> extent; results land in a preallocated table,
no allocation, no host sync), and names the first (call, tensor) whose checksum differs from the step-synchronised window's.
Entries that differ between two step-synchronised windows (scratch whose order is free, e.g. the cell-order lists) are masked.

    python profiles/coresidency_trace.py <windows> [image] [prepared] [backward]
    (T2H_TRACE_POINTS=N, T2H_TRACE_COALESCE=1 for the tile-by-tile window, T2H_TRACE_STALL=1, T2H_TRACE_SAVE=call:tensor, T2H_TRACE_LIST=lo,hi)
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
coalesce = int(os.environ.get("T2H_TRACE_COALESCE", "4"))            # 1: tile by tile (four tiles)
stall = os.environ.get("T2H_TRACE_STALL") == "1"                      # the second forward of a pipelined window starts 0.1 s late
dev = torch.device("cuda:0")
ag = ctypes.CDLL(os.path.join(ROOT, "profiles", "_lab", "libaggr.so"))
ag.aggr_checksum.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
tiles = [{"inputs": synth_cloud(points, seed=700 + i).to(dev),
          "dsm": (torch.rand(1, 512, 512, generator=torch.Generator().manual_seed(i)) * 30).to(dev)} for i in range(9 if coalesce > 1 else 4)]
if image:
    for i, t in enumerate(tiles):
        t["image"] = torch.randn(1, 3, 512, 512, generator=torch.Generator().manual_seed(40 + i)).to(dev)
cfg = berlin_config(use_image=image)

CALLS, ARGS = 4096, 64
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
    assert i < CALLS, i
    uniq = uniq[:ARGS]
    s = _lib.stream()
    for j, (p, n) in enumerate(uniq):
        if n >= 4:
            ag.aggr_checksum(p, n, table_ptr + 8 * (i * ARGS + j), s)
        if SAVE == (i, j):
            saved_n[0] = min(n, saved.numel() * 4) // 4
            hip.hipMemcpyAsync(raw_data_ptr(saved), p, saved_n[0] * 4, 3, s)


torch.Tensor.data_ptr = data_ptr
_lib.call = call
# T2H_TRACE_SAVE=call:tensor -- a copy of that tensor's bytes right after that call, compared word by word where the losses differ
SAVE = tuple(int(v) for v in os.environ["T2H_TRACE_SAVE"].split(":")) if os.environ.get("T2H_TRACE_SAVE") else None
saved = torch.zeros(1 << 26, dtype=torch.int32, device=dev) if SAVE else None          # 256 MB
saved_n = [0]
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]


def run(stepsync):
    model = det_init_(TomoSAR2Height(cfg), seed=15).to(dev)
    model.set_channels_last(True)
    tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=dev, optimize_every=100, use_cloud=True, use_image=image)
    tr.coalesce_tiles = coalesce
    side = torch.cuda.Stream() if ahead else None
    prep = (lambda t: tr.prepare(t, side)) if ahead else (lambda t: t)
    losses, inner, inner_bwd = [], tr._losses, tr._backward
    torch.cuda.synchronize()
    table.zero_()
    state["k"] = 0
    del seen[:]

    def rec(data, thr):
        if stall and not stepsync and len(losses) == 1:
            torch.cuda._sleep(int(3e8))
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
saved_gold = saved[:saved_n[0]].clone() if SAVE else None
gold2, _ = run(True)
if os.environ.get("T2H_TRACE_LIST"):
    lo, hi = (int(v) for v in os.environ["T2H_TRACE_LIST"].split(","))
    for k in range(lo, min(hi, len(names))):
        print(f"    [{k}] {names[k][0]} ({names[k][1]}) {names[k][2]}")
free = gold != gold2
print(f"{gold.shape[0]} traced calls per window ({'forward + backward' if with_bwd else 'forward'}), {int((gold != 0).sum())} tensors; "
      f"{int(free.sum())} entries differ between two step-synchronised windows (masked):", flush=True)
for k in sorted(set(torch.nonzero(free)[:, 0].tolist())):
    print(f"    call {k}: {names[k][0]} ({names[k][1]}) tensors {torch.nonzero(free[k]).flatten().tolist()}")
bad, record = 0, []
for it in range(windows):
    got, ls = run(False)
    d = torch.nonzero((got != gold) & ~free)
    record.append((it, ls == gl, {(int(k), int(j)) for k, j in d.tolist()}, [a == b for a, b in zip(ls, gl)]))
    if SAVE and ls != gl:
        now = saved[:saved_n[0]]
        w = torch.nonzero(now != saved_gold).flatten()
        print(f"window {it}: saved tensor {SAVE}: {len(w)} of {saved_n[0]} words differ"
              + (f", word offsets {w[:6].tolist()} .. {w[-3:].tolist()}; here {[hex(v & 0xffffffff) for v in now[w[:6]].tolist()]} "
                 f"gold {[hex(v & 0xffffffff) for v in saved_gold[w[:6]].tolist()]}; last 64 words here "
                 f"{[hex(v & 0xffffffff) for v in now[-64:].tolist() if v]} gold {[hex(v & 0xffffffff) for v in saved_gold[-64:].tolist() if v]}" if len(w) else ""),
              flush=True)
    bad += ls != gl
# entries that also differ in windows whose losses are all equal are not the fault: a tensor handed over as a slice of a wider
# buffer is summed over its whole extent (the r = 256 concatenation: 2.7 GB), columns another call has not written yet included
noise = set()
for _, same, d, _ in record:
    if same:
        noise |= d
print(f"{len(noise)} (call, tensor) entries differ in windows whose losses are equal (unwritten parts of wider buffers): ignored")
firsts = {}
for it, same, d, eq in record:
    if same:
        continue
    real = sorted(d - noise)
    rows = sorted({k for k, _ in real})
    print(f"window {it}: losses equal {eq}; {len(rows)} calls differ beyond the ignored entries", flush=True)
    for k in rows[:8]:
        js = [j for kk, j in real if kk == k]
        print(f"    call {k}: {names[k][0]} ({names[k][1]}): tensors {js} of {len(names[k][2])} (bytes {[names[k][2][j] for j in js]}; all {names[k][2]})", flush=True)
    key = f"{rows[0]}:{names[rows[0]][0]}" if rows else "none beyond the ignored entries"
    firsts[key] = firsts.get(key, 0) + 1
print(f"{'cloud+image' if image else 'cloud-only'}{', prepared' if ahead else ''}, {'coalesced' if coalesce > 1 else 'tile by tile'}{', stalled' if stall else ''}, N = {points}: the losses of {bad} of {windows} windows differ; "
      f"first differing call -> windows: {firsts or 'none'}", flush=True)
