"""-m gpu: kernels of two tiles share the chip in the trainer's tile pipeline (forward i + 1 beside backward i, each on its own stream).
A kernel's result must not depend on what runs beside it.  r05 found one that did: the on-chip scatter-reduce walk (both its r04 and
r05 form) sampled 16 rows of a wave without their east tap whenever the split-convolution kernels (conv_bx3.hip) were resident on
the same CU -- alone, or beside any other kernel, never (point_grid.hip, sample_relu_cellsums_v2_kernel; profiles/r05_coresidency.txt).

The method that found it is kept as the test: record the C-ABI calls of one tile's backward, replay the convolution calls on a second
stream, run the kernel under test beside them on fixed inputs, compare bit for bit with its result alone; and, for everything else,
the pipelined window against the step-synchronised one many times over."""
import os

import pytest
import torch

gpu = pytest.mark.gpu


def _setup():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from detinit import det_init_, synth_cloud
    from tomosar2height_amd import TomoSAR2Height
    from tomosar2height_amd.config import berlin_config
    from tomosar2height_amd.trainer import Trainer
    return det_init_, synth_cloud, TomoSAR2Height, berlin_config, Trainer


def _tiles(synth_cloud, dev, n, points=40000):
    return [{"inputs": synth_cloud(points, seed=700 + i).to(dev),
             "dsm": (torch.rand(1, 512, 512, generator=torch.Generator().manual_seed(i)) * 30).to(dev)} for i in range(n)]


@gpu
def test_walks_beside_the_split_convolutions_are_bit_identical():
    from tomosar2height_amd import _lib, deferred
    from tomosar2height_amd.tile import TileIndex
    det_init_, synth_cloud, TomoSAR2Height, berlin_config, Trainer = _setup()
    dev = torch.device("cuda:0")
    # the forward walk's fixed inputs and every buffer it writes, allocated BEFORE the calls to be replayed are recorded (the replay
    # writes into blocks the allocator considers free)
    tile = TileIndex(synth_cloud(40000, seed=703).to(dev), 128)
    cases = []
    for level, c in ((3, 1024), (2, 512)):
        r = 128 >> level
        q = torch.randn(r * r, c, device=dev)
        rows = tile.B << (2 * tile.nbits)
        order = tile.cell_order(level)

        def outs(rows=rows, c=c):
            return (torch.zeros(rows, c, device=dev), torch.zeros(rows // 4, c, device=dev),
                    torch.zeros(tile.n_points * (c // 256) * 4, dtype=torch.int64, device=dev))

        def walk(o, level=level, c=c, q=q, order=order):
            _lib.call("t2h_sample_relu_cellsums_ordered", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N,
                      tile.nbits, level, 0, c, o[0].data_ptr(), c, o[1].data_ptr(), c, o[2].data_ptr(), _lib.ptr(order), _lib.stream())
        ref = outs()
        walk(ref)
        cases.append((f"forward walk level {level}", walk, ref, [outs() for _ in range(4)]))
    torch.cuda.synchronize()

    model = det_init_(TomoSAR2Height(berlin_config()), seed=15).to(dev)
    model.set_channels_last(True)
    tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=dev, optimize_every=100, use_cloud=True)
    tr.pipeline_tiles = False
    tr.overlap_wgrad = tr.overlap_conv_wgrad = False
    tiles = _tiles(synth_cloud, dev, 3)
    tr.train_step(tiles[0])
    tr.train_step(tiles[1])
    torch.cuda.synchronize()

    side_a, side_b = torch.cuda.Stream(), torch.cuda.Stream()
    recorded, grads = [], []
    orig_call, orig_hidden = _lib.call, deferred.Deferred.hidden_grad

    def recording(name, *a, **k):
        recorded.append((name, a))
        return orig_call(name, *a)

    def keep(t):
        # a copy in ANOTHER stream's pool: a block of this stream's pool may be the (since freed) output of an earlier recorded call,
        # which the replay would then write into
        cur = torch.cuda.current_stream()
        side_a.wait_stream(cur)
        with torch.cuda.stream(side_a):
            c = t.clone()
        t.record_stream(side_a)
        return c

    def hidden(self, planes, h, r, c2, mask_is_bits=False):
        if mask_is_bits:
            grads.append((self, [(keep(p), lv) for p, lv in planes], keep(h), r, c2, mask_is_bits))
        return orig_hidden(self, planes, h, r, c2, mask_is_bits)
    try:
        with torch.cuda.stream(side_b):                        # (the replayed calls carry this stream in their arguments)
            with tr._own_cache():
                l1, ce = tr._losses(tiles[2], 0.0001)
            _lib.call, deferred.Deferred.hidden_grad = recording, hidden
            tr._backward(l1 + ce)
    finally:
        _lib.call, deferred.Deferred.hidden_grad = orig_call, orig_hidden
    torch.cuda.synchronize()
    conv = [c for c in recorded if "bx3" in c[0]]
    assert len(conv) >= 40, sorted({c[0] for c in recorded})
    # the backward walks of that pass, on copies of their inputs
    for state, planes, h, r, c2, is_bits in grads:
        if not is_bits:
            continue

        def bwd(o, state=state, planes=planes, h=h, r=r, c2=c2):
            o[0] = orig_hidden(state, planes, h, r, c2, True)
        ref = [None]
        bwd(ref)
        cases.append((f"backward walk r {r}", bwd, ref, [[None] for _ in range(4)]))
    torch.cuda.synchronize()
    assert len(cases) >= 3

    report = {}
    for what, run, ref, res in cases:
        bad = 0
        for _ in range(12):
            main = torch.cuda.current_stream()
            side_a.wait_stream(main)
            side_b.wait_stream(main)
            with torch.cuda.stream(side_b):
                for _ in range(3):
                    for name, a in conv:
                        orig_call(name, *a)
            with torch.cuda.stream(side_a):
                for o in res:
                    run(o)
            torch.cuda.synchronize()
            bad += sum(not all(torch.equal(x, y) for x, y in zip(ref, o)) for o in res)
        report[what] = bad
    assert not any(report.values()), f"launches (of 48) whose result beside the convolutions differs from the result alone: {report}"


@gpu
@pytest.mark.parametrize("ahead,points,image,windows,coalesce",
                         [(False, 40000, False, 8, 1), (True, 40000, False, 8, 1), (False, 131072, False, 4, 1), (False, 40000, True, 8, 1),
                          (False, 40000, False, 8, 4), (False, 131072, False, 4, 4), (False, 40000, True, 8, 4)],
                         ids=["lazy", "prepared", "benchmark-size", "cloud+image",
                              "coalesced", "coalesced-benchmark-size", "coalesced-cloud+image"])
def test_pipelined_window_equals_the_step_synchronised_one_every_time(ahead, points, image, windows, coalesce):
    """Windows through the tile pipeline (tile indices built ahead on a side stream or inside the step; the benchmarked
    N = 131 072; BASELINE configs[2] with the image U-Net; tile by tile -- four tiles -- and, r06, with the Trainer's default
    coalescing: nine tiles = micro-batches of 1, 4 and 4 whose forwards and backwards overlap), each against the same window with a
    device synchronise after every step: losses and every gradient bit for bit, every time -- before the r05 fix 4-50 % of the
    windows differed.  r06: the cloud+image case is a hard assertion like the others (it was reported as an expected failure)."""
    det_init_, synth_cloud, TomoSAR2Height, berlin_config, Trainer = _setup()
    dev = torch.device("cuda:0")
    tiles = _tiles(synth_cloud, dev, 4 if coalesce == 1 else 9, points)
    if image:
        for i, t in enumerate(tiles):
            t["image"] = torch.randn(1, 3, 512, 512, generator=torch.Generator().manual_seed(40 + i)).to(dev)
    cfg = berlin_config(use_image=image)

    def run(ahead, stepsync, stall=False, seed=15):
        model = det_init_(TomoSAR2Height(cfg), seed=seed).to(dev)
        model.set_channels_last(True)
        tr = Trainer(model, torch.optim.SGD(model.parameters(), lr=0.0), device=dev, optimize_every=100, use_cloud=True, use_image=image)
        tr.coalesce_tiles = coalesce
        side = torch.cuda.Stream() if ahead else None
        prep = (lambda t: tr.prepare(t, side)) if ahead else (lambda t: t)
        losses, inner = [], tr._losses

        def rec(data, thr):
            if stall and len(losses) == 1:
                torch.cuda._sleep(int(3e8))                      # the second forward starts ~0.1 s late on its stream
            l1, ce = inner(data, thr)
            losses.append(l1.detach())
            return l1, ce
        tr._losses = rec
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

    gold, gold_losses = run(ahead, True)
    assert len(gold_losses) == (4 if coalesce == 1 else 3)           # (tile by tile; or micro-batches of 1 + 4 + 4 tiles)
    differing = []
    for it in range(windows):
        got, losses = run(ahead, False)
        n = sum(not torch.equal(got[k], gold[k]) for k in gold)
        if n or losses != gold_losses:
            differing.append((it, n, [a == b for a, b in zip(losses, gold_losses)]))
    assert not differing, f"windows that differ from the synchronised one (iteration, gradients, per-tile loss equal): {differing}"
    # r06: one more window with the second forward held back on its stream.  With coalescing it is the first four-tile forward of
    # the trainer's life and fills the cache entries a single tile does not use (the split weights of the r = 32 level product);
    # the third forward, on the other tile stream, reaches them first and has to wait for the fill (_lib.Ready) -- without that it
    # read an unfilled buffer (1 window in 150 unprovoked; T2H_CACHE_READY=0 shows this assertion catching it).  Other weights
    # than every window before, and the synchronised window AFTER the stalled one: what the fresh buffer holds must not happen to be
    # an earlier window's copy of the same split weights (which is what hid the race 149 times in 150).
    got, losses = run(ahead, False, stall=True, seed=16)
    gold, gold_losses = run(ahead, True, seed=16)
    n = sum(not torch.equal(got[k], gold[k]) for k in gold)
    assert n == 0 and losses == gold_losses, (n, [a == b for a, b in zip(losses, gold_losses)])
