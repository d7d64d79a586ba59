"""r05 forensics: which ROWS does the walk evaluate differently beside the replayed convolution calls (from the per-row sign bits), and
where do they lie in their wave (every run starts at lane 48 of the wave's first 64-row batch)?  Needs the pre-fix library
(T2H_LIBRARY=...); the t2h_debug_set dump it can call existed only in the instrumented build described in profiles/r05_coresidency.txt."""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import torch
from detinit import det_init_, synth_cloud
from tomosar2height_amd import TomoSAR2Height, _lib
from tomosar2height_amd.config import berlin_config
from tomosar2height_amd.trainer import Trainer
from tomosar2height_amd.tile import TileIndex
dev = torch.device("cuda:0")
reps = int(sys.argv[1])
cfg = berlin_config()
tiles = [{"inputs": synth_cloud(40000, seed=700 + i).to(dev),
          "dsm": (torch.rand(1, 512, 512, generator=torch.Generator().manual_seed(i)) * 30).to(dev)} for i in range(3)]
tile = TileIndex(synth_cloud(40000, seed=703).to(dev), 128)
level, C = 3, 1024
r = 128 >> level
q = torch.randn(r * r, C, device=dev)
rows = tile.B << (2 * tile.nbits)
npts = tile.n_points
def outs():
    return (torch.zeros(rows, C, device=dev), torch.zeros(rows // 4, C, device=dev),
            torch.zeros(npts * (C // 256) * 4, dtype=torch.int64, device=dev))
order = tile.cell_order(level) if os.environ.get("NO_ORDER") != "1" else None
def walk(o):
    _lib.call("t2h_sample_relu_cellsums_ordered", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits,
              level, 0, C, o[0].data_ptr(), C, o[1].data_ptr(), C, o[2].data_ptr(), None if order is None else _lib.ptr(order), _lib.stream())
import ctypes
lib = _lib.load()
lib.t2h_debug_set.argtypes = [ctypes.c_void_p]
NW = 1 << 16
dbg_ref = torch.zeros(NW * 64 * 16, device=dev)
dbg = [torch.zeros(NW * 64 * 16, device=dev) for _ in range(4)]
def walk_dbg(o, d):
    torch.cuda.current_stream().synchronize()
    assert lib.t2h_debug_set(d.data_ptr()) == 0
    walk(o)
ref = outs(); walk_dbg(ref, dbg_ref); torch.cuda.synchronize(); assert lib.t2h_debug_set(None) == 0
res = [outs() for _ in range(4)]
torch.cuda.synchronize()
off0 = tile.off0.cpu()
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
with torch.cuda.stream(Bs):
    with tr._own_cache():
        l1, ce = tr._losses(tiles[2], 0.0001)
    loss = l1 + ce
    _lib.call = recording
    tr._backward(loss)
    _lib.call = orig
torch.cuda.synchronize()
calls = [c for c in rec if c[0] == os.environ.get("CULPRIT", "t2h_conv3x3_bx3_wgrad")]
shown = 0
for it in range(reps):
    A.wait_stream(torch.cuda.current_stream()); Bs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(Bs):
        for _ in range(3):
            for n, a in calls: orig(n, *a)
    with torch.cuda.stream(A):
        pass
    for k, o in enumerate(res):
        assert lib.t2h_debug_set(dbg[k].data_ptr()) == 0      # (hipMemcpyToSymbol: synchronous w.r.t. the null stream only)
        with torch.cuda.stream(A): walk(o)
        A.synchronize()
    torch.cuda.synchronize()
    assert lib.t2h_debug_set(None) == 0
    for k, o in enumerate(res):
        if torch.equal(ref[0], o[0]) and torch.equal(ref[2], o[2]): continue
        if shown >= 4: continue
        shown += 1
        dd = (dbg[k] != dbg_ref).view(2, NW, 64, 8)[0]
        wv = dd.any(2).any(1).nonzero().flatten().tolist()
        print(f"  DEBUG launch {k}: waves whose first-batch taps differ: {len(wv)}")
        for w in wv[:6]:
            lanes = dd[w].any(1).nonzero().flatten().tolist()
            fields = dd[w].any(0).nonzero().flatten().tolist()
            l0 = lanes[0]
            print(f"     wave {w}: lanes {lanes[0]}..{lanes[-1]} ({len(lanes)}), fields {fields}; lane {l0}")
        du = (dbg[k] != dbg_ref).view(2, NW, 64, 8)[1]
        wv = du.any(2).any(1).nonzero().flatten().tolist()
        print(f"  DEBUG launch {k}: waves whose per-row values AT USE differ: {len(wv)}")
        G, R = dbg[k].view(2, NW, 64, 8)[1], dbg_ref.view(2, NW, 64, 8)[1]
        for w in wv[:5]:
            rows_i = du[w].any(1).nonzero().flatten().tolist()
            fields = du[w].any(0).nonzero().flatten().tolist()
            i0 = rows_i[0]
            print(f"     wave {w} (lane 0-7 of chunk {(w // 4) % 4}): rows i {rows_i[0]}..{rows_i[-1]} ({len(rows_i)}), fields {fields}")
            print(f"         i={i0}: got {[round(v, 5) for v in G[w, i0].tolist()]}")
            print(f"         i={i0}: ref {[round(v, 5) for v in R[w, i0].tolist()]}")
        b_ref = ref[2].view(C // 256, npts, 4); b_got = o[2].view(C // 256, npts, 4)
        drow = (b_ref != b_got).any(2)                       # [chunk, row]
        dsum = (ref[0] != o[0]).view(rows, C // 256, 256).any(2)   # [cell, chunk]
        print(f"iter {it} launch {k}: rows with different sign bits {int(drow.sum())}, (cell, chunk) sums different {int(dsum.sum())}")
        for ch in range(C // 256):
            rr = drow[ch].nonzero().flatten().tolist()
            if not rr: continue
            # group into runs
            runs, start, prev = [], rr[0], rr[0]
            for x in rr[1:]:
                if x > prev + 3: runs.append((start, prev)); start = x
                prev = x
            runs.append((start, prev))
            for (a, b) in runs:
                n_in = sum(1 for x in rr if a <= x <= b)
                fc = int(torch.searchsorted(off0, torch.tensor(a), right=True)) - 1      # finest cell of the first row
                sc = fc >> (2 * level)                                                     # sampling cell (Morton) at `level`
                s0, s1 = int(off0[sc << (2 * level)]), int(off0[(sc + 1) << (2 * level)])
                # the wave's quarter of the children (v2, K = 0, gz = 1): 16 children each
                nchild = 1 << (2 * level); q4 = nchild // 4
                wq = (fc - (sc << (2 * level))) // q4
                w0, w1 = int(off0[(sc << (2 * level)) + wq * q4]), int(off0[(sc << (2 * level)) + (wq + 1) * q4])
                al = []
                for gz in (1, 2, 4, 8, 16):
                    pw = max(nchild // (gz * 4), 1)
                    ws = int(off0[(sc << (2 * level)) + ((fc - (sc << (2 * level))) // pw) * pw])
                    al.append((a - ws) % 64)
                print(f"    run {a}..{b} len {b - a + 1}: start lane for gz=1,2,4,8,16: {al}")
                continue
                print(f"    chunk {ch}: rows {a}..{b} ({n_in} differ) | sampling cell {sc}: rows {s0}..{s1 - 1} | wave quarter {wq}: rows {w0}..{w1 - 1} | offset in wave {a - w0}..{b - w0}")
print("done")
