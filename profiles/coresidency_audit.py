"""r05: does ANY C-ABI call of a training tile-step give a different result when other kernels run beside it?

Every call of one tile's forward and backward (the victim, recorded on stream A) is replayed
  (1) alone, twice  -- a call whose two solo runs differ accumulates into its output (or was given recycled inputs): skipped;
  (2) beside a competitor set replayed on stream B (calls recorded from ANOTHER tile's backward / forward on stream B: the two
      streams' allocator pools are disjoint, so the competitor never writes what the victim reads),
and after each run every allocator block that one of the victim's pointer arguments points into is copied to the host and hashed.

    python profiles/coresidency_audit.py [trials] [competitor: bx3 | walks | all | mfma] [competitor passes per trial]

Output: one line per victim call that differs, and a summary per entry point.  Findings: profiles/r05_coresidency.txt."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import torch
import xxhash

from detinit import det_init_, synth_cloud
from tomosar2height_amd import TomoSAR2Height, _lib
from tomosar2height_amd.config import berlin_config
from tomosar2height_amd.trainer import Trainer

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 3
which = sys.argv[2] if len(sys.argv) > 2 else "bx3"
dense = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
MAX_BLOCK = 300 << 20

IMAGE = os.environ.get("T2H_AUDIT_IMAGE") == "1"        # BASELINE configs[2]: with the image U-Net
tiles = [{"inputs": synth_cloud(40000, seed=700 + i).to(dev),
          "dsm": (torch.rand(1, 512, 512, generator=torch.Generator().manual_seed(i)) * 30).to(dev)} for i in range(4)]
if IMAGE:
    for i, t in enumerate(tiles):
        t["image"] = torch.randn(1, 3, 512, 512, generator=torch.Generator().manual_seed(40 + i)).to(dev)
model = det_init_(TomoSAR2Height(berlin_config(use_image=IMAGE)), seed=15).to(dev)
model.set_channels_last(True)
tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=dev, optimize_every=100, use_cloud=True, use_image=IMAGE)
tr.pipeline_tiles = False
tr.overlap_wgrad = tr.overlap_conv_wgrad = False
tr.train_step(tiles[0])
tr.train_step(tiles[1])
torch.cuda.synchronize()
A, B = torch.cuda.Stream(), torch.cuda.Stream()
orig = _lib.call


def record(stream, tile, keep):
    rec = []

    def recording(name, *a, **k):
        rec.append((name, a))
        return orig(name, *a)
    with torch.cuda.stream(stream):
        _lib.call = recording
        try:
            with tr._own_cache():
                l1, ce = tr._losses(tile, 0.0001)
            loss = l1 + ce
            n_fwd = len(rec)
            tr._backward(loss)
        finally:
            _lib.call = orig
    torch.cuda.synchronize()
    return rec, n_fwd


victims, n_fwd = record(A, tiles[2], None)
comp_all, comp_fwd = record(B, tiles[3], None)
if which == "bx3":
    comp = [c for c in comp_all if "bx3" in c[0]]
elif which == "walks":
    comp = [c for c in comp_all if "sample_" in c[0] or "segsum" in c[0] or "trunk" in c[0] or "pool" in c[0]]
elif which == "mfma":        # the synthetic aggressor: a pure v_mfma_f32_16x16x32_f16 loop (profiles/coresidency_aggressor.hip)
    ag = ctypes.CDLL(os.path.abspath(os.environ.get("T2H_AGGR_LIB", "scratch/libaggr.so")))
    ag.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    dummy = torch.zeros(16, device=dev)

    def _aggr(name, *a):
        assert ag.aggr_launch(3, dummy.data_ptr(), 512, 2000, torch.cuda.current_stream().cuda_stream) == 0
    comp = [("aggr", ())] * 2
else:
    comp = comp_all
print(f"{len(victims)} victim calls ({n_fwd} forward), {len(comp)} competitor calls ({which})")

# allocator blocks: address -> size
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


host = np.empty(MAX_BLOCK, dtype=np.uint8)


def digest(bl, keep=None):
    h = xxhash.xxh64()
    for addr, size in bl:
        rc = hip.hipMemcpy(host.ctypes.data, addr, size, 2)
        assert rc == 0, rc
        h.update(host[:size].tobytes() if size < (1 << 16) else memoryview(host[:size]))
        if keep is not None:
            keep[addr] = host[:size].copy()
    return h.hexdigest()


def explain(bl, solo, args):
    """Which block differs from the solo run, where, and which argument points into it."""
    for addr, size in bl:
        rc = hip.hipMemcpy(host.ctypes.data, addr, size, 2)
        assert rc == 0, rc
        d = np.nonzero(host[:size] != solo[addr])[0]
        if len(d):
            who = [i for i, a in enumerate(args) if isinstance(a if not isinstance(a, ctypes.c_void_p) else a.value, int)
                   and addr <= (a if not isinstance(a, ctypes.c_void_p) else a.value) < addr + size]
            w = d // 4
            got = host[:size].view(np.float32)[w[:4]]
            want = solo[addr].view(np.float32)[w[:4]]
            print(f"      block of {size} bytes (arguments {who}): {len(np.unique(w))} words differ, byte offsets {int(d[0])}..{int(d[-1])}; "
                  f"first words beside {got.tolist()} alone {want.tolist()}")


comp_blocks = set()
for n, a in comp:
    comp_blocks.update(b for b, _ in pointer_blocks(a))

summary = {}
for idx, (name, args) in enumerate(victims):
    bl = pointer_blocks(args)
    s = summary.setdefault(name, {"calls": 0, "skipped": 0, "shared": 0, "differ": 0})
    s["calls"] += 1
    if not bl:
        s["skipped"] += 1
        continue
    with torch.cuda.stream(A):
        orig(name, *args)
    torch.cuda.synchronize()
    solo = {}
    h1 = digest(bl, solo)
    with torch.cuda.stream(A):
        orig(name, *args)
    torch.cuda.synchronize()
    if digest(bl) != h1:                       # accumulates into its output, or runs in place
        s["skipped"] += 1
        continue
    if any(b in comp_blocks for b, _ in bl):   # weights / bucket: shared with the competitor (it may write the bucket)
        s["shared"] += 1
    bad = 0
    for t in range(trials):
        main = torch.cuda.current_stream()
        A.wait_stream(main)
        B.wait_stream(main)
        with torch.cuda.stream(B):
            for _ in range(dense):
                for cn, ca in comp:
                    (_aggr if cn == "aggr" else orig)(cn, *ca)
        with torch.cuda.stream(A):
            for _ in range(3 * dense):
                orig(name, *args)
        torch.cuda.synchronize()
        if digest(bl) != h1:
            bad += 1
            if bad == 1:
                explain(bl, solo, args)
            # is the difference lasting (a shared block the competitor wrote) or transient?
            with torch.cuda.stream(A):
                orig(name, *args)
            torch.cuda.synchronize()
            if digest(bl) != h1:
                bad -= 1
                s["shared"] += 1
                break
    if bad:
        s["differ"] += 1
        print(f"  DIFFERS: call {idx} ({'fwd' if idx < n_fwd else 'bwd'}) {name}: {bad} of {trials} runs beside the competitor")
print(f"{'entry point':46s} calls  checked  differ")
for name, s in summary.items():
    print(f"{name:46s} {s['calls']:5d}  {s['calls'] - s['skipped']:7d}  {s['differ']:6d}" + ("   <<<<" if s["differ"] else ""))
