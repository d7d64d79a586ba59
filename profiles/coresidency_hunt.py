"""r06: WHICH call of a micro-batch's forward gives a different result beside another micro-batch's backward?

The soak (profiles/coresidency_soak.py) showed one coalesced window in 150 whose third micro-batch's LOSS differed from the
step-synchronised run: some forward kernel.  This replays the recorded C-ABI calls of one four-tile forward on stream A, with an
integer checksum of every allocator block a call's pointer arguments reach taken right after the call (a kernel of
coresidency_aggressor.hip, results in a preallocated table: nothing is allocated during a replay), beside the recorded backward
calls of ANOTHER micro-batch replayed on stream B -- and names the FIRST call whose checksums differ from the replay alone.

    python profiles/coresidency_hunt.py [trials=200] [competitor: bx3 | all] [passes=0: cover the replay] [tiles=4] [points=40000]
"""
import ctypes
import os
import sys

# (the replay reads the recorded micro-batch's index buffers after the allocator has recycled them: harmless for kernels that only
#  index rows with them, not for the one-launch trunk, whose unit list and count live there -- the hunt runs the per-block trunk)
os.environ.setdefault("T2H_TRUNK_FUSED", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np
import torch
from detinit import det_init_, synth_cloud
from tomosar2height_amd import TomoSAR2Height, _lib
from tomosar2height_amd.config import berlin_config
from tomosar2height_amd.trainer import Trainer

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
which = sys.argv[2] if len(sys.argv) > 2 else "bx3"
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 4
points = int(sys.argv[5]) if len(sys.argv) > 5 else 40000
IMAGE = os.environ.get("T2H_AUDIT_IMAGE") == "1"
dev = torch.device("cuda:0")
ag = ctypes.CDLL(os.path.join(ROOT, "profiles", "_lab", "libaggr.so"))
ag.aggr_checksum.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
MAX_BLOCK = 1 << 31


def tile(i):
    t = {"inputs": synth_cloud(points + 500 * (i % 4), seed=700 + i).to(dev),
         "dsm": (torch.rand(1, 512, 512, generator=torch.Generator().manual_seed(i)) * 30).to(dev)}
    if IMAGE:
        t["image"] = torch.randn(1, 3, 512, 512, generator=torch.Generator().manual_seed(40 + i)).to(dev)
    return t


tiles = [tile(i) for i in range(1 + 3 * nb)]
model = det_init_(TomoSAR2Height(berlin_config(use_image=IMAGE)), seed=15).to(dev)
model.set_channels_last(True)
tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=dev, optimize_every=10000, use_cloud=True, use_image=IMAGE)
tr.pipeline_tiles = False
tr.overlap_wgrad = tr.overlap_conv_wgrad = False
tr.coalesce_tiles = 1
tr.train_step(tiles[0])
tr.train_step(tiles[1:1 + nb])
torch.cuda.synchronize()
SAVE_BYTES = int(os.environ.get("T2H_HUNT_SAVE_GB", "48")) << 30
save = torch.empty(SAVE_BYTES, dtype=torch.uint8, device=dev)    # images of the replayed blocks (allocated before the recording:
hip = ctypes.CDLL("libamdhip64.so")                                #  never one of the blocks a recorded call points into)
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
A, B = torch.cuda.Stream(), torch.cuda.Stream()
orig = _lib.call


def record(stream, batch):
    rec = []

    def recording(name, *a, **k):
        rec.append((name, a, k.get("tag") or name))
        return orig(name, *a)
    with torch.cuda.stream(stream):
        _lib.call = recording
        try:
            with tr._own_cache():
                l1, ce = tr._losses(batch, 0.0001)
            n_fwd = len(rec)
            tr._backward(l1 + ce)
        finally:
            _lib.call = orig
    torch.cuda.synchronize()
    return rec, n_fwd


victims, n_fwd = record(A, tiles[1 + nb:1 + 2 * nb])
comp_all, comp_fwd = record(B, tiles[1 + 2 * nb:1 + 3 * nb])
victims = victims[:n_fwd]
comp = [c for c in comp_all[comp_fwd:] if (which == "all" or "bx3" in c[0])]
print(f"{len(victims)} forward calls of a {nb}-tile micro-batch (N ~ {points}), competitor: {len(comp)} backward calls ({which}); "
      f"{trials} trials{', cloud+image' if IMAGE else ''}")

blocks = []
for seg in torch.cuda.memory_snapshot():
    addr = seg["address"]
    for b in seg["blocks"]:
        blocks.append((b.get("address", addr), b["size"]))
        addr += b["size"]
blocks.sort()
starts = np.array([b[0] for b in blocks], dtype=np.uint64)


def block_of(p):
    i = int(np.searchsorted(starts, np.uint64(p), side="right")) - 1
    if i >= 0 and blocks[i][0] <= p < blocks[i][0] + blocks[i][1]:
        return blocks[i]
    return None


def pointer_blocks(args):
    out = {}
    for a in args:
        v = a.value if isinstance(a, ctypes.c_void_p) else a
        if isinstance(v, int) and v > (1 << 32):
            b = block_of(v)
            if b is not None and b[1] <= MAX_BLOCK:
                out[b[0]] = b[1]
    return sorted(out.items())


per_call = [pointer_blocks(a) for _, a, _ in victims]
width = max(len(b) for b in per_call)
table = torch.zeros(len(victims), width, dtype=torch.int64, device=dev)      # allocated BEFORE any replay, on the default stream


images, used = [], 0                                 # (address, size, offset into `save`) of every block a recorded call reaches
for addr, size in sorted({b for blks in per_call for b in blks}):
    if used + size > SAVE_BYTES:
        raise SystemExit(f"T2H_HUNT_SAVE_GB too small: {used + size} bytes needed so far")
    images.append((addr, size, used))
    used += (size + 255) // 256 * 256


def copy_blocks(to_save, stream):
    for addr, size, off in images:
        src, dst = (addr, save.data_ptr() + off) if to_save else (save.data_ptr() + off, addr)
        assert hip.hipMemcpyAsync(dst, src, size, 3, stream) == 0


def replay_forward():
    """Every replay starts from the same image of the blocks: calls that accumulate onto buffers torch zeroed in the recorded run
    (not C-ABI calls, so not replayed) then give the same bits every time."""
    table.zero_()
    sA = A.cuda_stream
    copy_blocks(False, sA)
    for k, (name, args, _tag) in enumerate(victims):
        orig(name, *args)
        for j, (addr, size) in enumerate(per_call[k]):
            assert ag.aggr_checksum(addr, size, table[k, j].data_ptr(), sA) == 0


main = torch.cuda.current_stream()
A.wait_stream(main)
copy_blocks(True, A.cuda_stream)
torch.cuda.synchronize()
gold, gold2 = torch.empty_like(table), torch.empty_like(table)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for g in (gold, gold2, gold2):
    with torch.cuda.stream(A):
        ev[0].record()
        replay_forward()
        ev[1].record()
    torch.cuda.synchronize()
    g.copy_(table)
stable = (gold2 == gold).all(1)                      # calls whose blocks hold the same bits in two solo replays
print(f"{int(stable.sum())} of {len(victims)} calls reproducible alone ({used / 2**30:.1f} GB of blocks restored before a replay)")
for k in torch.nonzero(~stable).flatten().tolist():
    print(f"  not reproducible alone: call {k} = {victims[k][2]} ({victims[k][0]})")
with torch.cuda.stream(B):
    ev[2].record()
    for cn, ca, _tag in comp:
        orig(cn, *ca)
    ev[3].record()
torch.cuda.synchronize()
t_fwd, t_comp = ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])
if passes <= 0:                                      # as many passes of the competitor as the replay (with its checksums) lasts
    passes = int(t_fwd / max(t_comp, 1e-3)) + 2
print(f"replay with checksums {t_fwd:.1f} ms, one pass of the competitor {t_comp:.1f} ms: {passes} passes per trial")
first_bad = {}
for t in range(trials):
    main = torch.cuda.current_stream()
    A.wait_stream(main)
    B.wait_stream(main)
    with torch.cuda.stream(B):
        for _ in range(passes):
            for cn, ca, _tag in comp:
                orig(cn, *ca)
    with torch.cuda.stream(A):
        replay_forward()
    torch.cuda.synchronize()
    bad = torch.nonzero(~(table == gold).all(1) & stable).flatten().tolist()
    if bad:
        k = bad[0]
        first_bad[k] = first_bad.get(k, 0) + 1
        print(f"  trial {t}: first differing call {k} = {victims[k][2]} ({victims[k][0]}); {len(bad)} calls differ after it", flush=True)
print("summary (first differing call -> trials):", {f"{k}:{victims[k][2]}": n for k, n in sorted(first_bad.items())} or "no trial differs")
