"""r06: the one residue of the r05 co-residency audit, counted launch by launch and with its magnitude.

Victim: `t2h_trunk_block_bwd`, FIRST variant (block 0 + fc_pos; trunk.hip, trunk_block_bwd_kernel<true,false>), recorded from one tile's
backward with its real inputs and replayed into K separate slab workspaces per trial, so that EVERY launch is compared with the
launch alone (the r05 audit compared the last of 3 x dense launches only).  Competitor on a second stream: the split-convolution
calls of another tile's backward (`bx3`: what shares the chip in the tile pipeline) or the synthetic v_mfma_f32_16x16x32_f16 loop
(`mfma`).  Also replays the other four trunk backward calls as a control.

    [T2H_LIBRARY=profiles/_lab/libt2h_trunk_pad4.so] python profiles/coresidency_trunk_first.py [launches=2000] [bx3|mfma] [passes=3]

Prints per victim: launches that differ, the slab words that differ and the worst |diff| / max |reference| of the REDUCED gradient
(what the optimizer would see: the slabs summed in the fixed order of t2h_trunk_block_reduce)."""
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

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
which = sys.argv[2] if len(sys.argv) > 2 else "bx3"
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
points = int(os.environ.get("T2H_LAB_POINTS", "40000"))
K = 16
dev = torch.device("cuda:0")
tiles = [{"inputs": synth_cloud(points, seed=700 + i).to(dev),
          "dsm": (torch.rand(1, 512, 512, generator=torch.Generator().manual_seed(i)) * 30).to(dev)} for i in range(4)]
model = det_init_(TomoSAR2Height(berlin_config()), seed=15).to(dev)
model.set_channels_last(True)
tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=dev, optimize_every=100, use_cloud=True)
tr.pipeline_tiles = False
tr.overlap_wgrad = tr.overlap_conv_wgrad = False
tr.train_step(tiles[0])
tr.train_step(tiles[1])
torch.cuda.synchronize()
A, B = torch.cuda.Stream(), torch.cuda.Stream()
orig = _lib.call


def record(stream, tile):
    rec, keep = [], []

    def recording(name, *a, **k):
        rec.append((name, a))
        return orig(name, *a)
    with torch.cuda.stream(stream):
        _lib.call = recording
        try:
            with tr._own_cache():
                l1, ce = tr._losses(tile, 0.0001)
            n_fwd = len(rec)
            # keep every tensor of the forward alive: the replayed backward calls read the saved activations
            loss = l1 + ce
            tr._backward(loss)
        finally:
            _lib.call = orig
    torch.cuda.synchronize()
    return rec, n_fwd


# NOTE the trunk backward's inputs (saved activations, the upstream gradient) live in stream A's allocator pool and are free blocks
# after the recording; nothing else allocates from that pool below (every new tensor is made on the default stream), so they stay intact
victims, _ = record(A, tiles[2])
comp_all, _ = record(B, tiles[3])
trunk = [(i, c) for i, c in enumerate(victims) if c[0] == "t2h_trunk_block_bwd"]
assert len(trunk) == 5, len(trunk)
WS_ARG, WSB_ARG, M_ARG = 24, 25, 22
if which == "bx3":
    comp = [c for c in comp_all if "bx3" in c[0]]
else:
    ag = ctypes.CDLL(os.path.join(ROOT, "profiles", "_lab", "libaggr.so"))
    ag.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    dummy = torch.zeros(16, device=dev)
    comp = [("aggr", ())] * 2


def compete():
    for _ in range(passes):
        for cn, ca in comp:
            if cn == "aggr":
                assert ag.aggr_launch(3, dummy.data_ptr(), 512, 2000, torch.cuda.current_stream().cuda_stream) == 0
            else:
                orig(cn, *ca)


def val(a):
    return a.value if isinstance(a, ctypes.c_void_p) else a


print(f"library: {os.environ.get('T2H_LIBRARY', 'shipped')}; competitor {which}: {len(comp)} calls x {passes} passes per trial; "
      f"{launches} launches per victim, {K} per trial; N = {points}")
SL = {"dW0": (0, 2048), "dWs": (2048, 2048), "dW1": (4096, 1024), "db0": (5120, 32), "db1": (5152, 32), "dWpos/dWc": (5184, 1056)}
exit_code = 0
for j, (idx, (name, args)) in enumerate(trunk):
    first = val(args[15]) is not None and val(args[15]) != 0          # pts != NULL
    last = val(args[7]) is not None and val(args[7]) != 0             # gc != NULL
    kind = "FIRST" if first else ("LAST" if last else "mid")
    n = launches if first else max(launches // 8, K)
    wsb = int(val(args[WSB_ARG]))
    m = int(val(args[M_ARG]))
    slabs = wsb // (6240 * 4)
    has_dx = val(args[23]) is not None and val(args[23]) != 0
    dx_ref = torch.empty(m, 64, device=dev) if has_dx else None
    dxs = [torch.empty(m, 64, device=dev) for _ in range(K)] if has_dx else [None] * K

    def call(ws, dx):
        a = list(args)
        a[WS_ARG] = ws.data_ptr()
        if dx is not None:
            a[23] = dx.data_ptr()
        orig(name, *a)
    ref = torch.zeros(wsb // 4, device=dev)
    with torch.cuda.stream(A):
        call(ref, dx_ref)
    torch.cuda.synchronize()
    again = torch.zeros(wsb // 4, device=dev)
    with torch.cuda.stream(A):
        call(again, dxs[0])
    torch.cuda.synchronize()
    assert torch.equal(ref, again), "the call alone is not reproducible: its inputs were overwritten"
    red_ref = ref.view(slabs, 6240).double().sum(0)
    wss = [torch.zeros(wsb // 4, device=dev) for _ in range(K)]
    bad, words, worst, where, dx_bad = 0, 0, 0.0, {}, 0
    for t in range(-(-n // K)):
        main = torch.cuda.current_stream()
        A.wait_stream(main)
        B.wait_stream(main)
        with torch.cuda.stream(B):
            compete()
        with torch.cuda.stream(A):
            for w, d in zip(wss, dxs):
                call(w, d)
        torch.cuda.synchronize()
        for w, d in zip(wss, dxs):
            if d is not None and not torch.equal(d, dx_ref):
                dx_bad += 1
            if torch.equal(w, ref):
                continue
            bad += 1
            diff = (w != ref).view(slabs, 6240)
            words += int(diff.sum())
            red = w.view(slabs, 6240).double().sum(0)
            for seg, (o, ln) in SL.items():
                e = float((red[o:o + ln] - red_ref[o:o + ln]).abs().max() / red_ref[o:o + ln].abs().max().clamp_min(1e-30))
                if diff[:, o:o + ln].any():
                    where[seg] = max(where.get(seg, 0.0), e)
                    worst = max(worst, e)
            wg = torch.nonzero(diff.any(1)).flatten().tolist()
            if bad <= 3:
                print(f"    launch differs: workgroup slab(s) {wg[:8]}, {int(diff.sum())} words, segments "
                      f"{[s for s, (o, ln) in SL.items() if diff[:, o:o + ln].any()]}", flush=True)
    total = -(-n // K) * K
    print(f"  trunk_block_bwd[{kind}] (call {idx}): {bad} of {total} launches differ in their slabs ({words} words; worst reduced-gradient "
          f"deviation {worst:.3e} of the segment's max; per segment {where}); dx differs in {dx_bad}", flush=True)
    if bad or dx_bad:
        exit_code = 1
sys.exit(exit_code)
