#!/usr/bin/env python3
"""Time the dense layers of one Berlin tile-step (BASELINE.json configs[1]: N = 131072 points, ALTO depth 5) one by one,
in isolation, with HIP events: per-point GEMMs (forward / data gradient / weight gradient) and the grid convolutions.
Prints microseconds and TFLOP/s per entry point call (weight gradients include their slab reduction).

    python profiles/layer_probe.py [--reps 20] [--only linear|conv]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomosar2height_amd import grid, mlp                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--points", type=int, default=131072)
ap.add_argument("--only", default="")
args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*shape):
    return torch.randn(*shape, device=dev, generator=g)


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / args.reps


def report(name, us, flops):
    print(f"{name:44s} {us:9.1f} us {flops / us / 1e6:7.1f} TF")


total = 0.0
if args.only in ("", "linear"):
    M = args.points
    # (K, N, calls per step) of the ALTO fc_comm / fc_c layers and the PointNet trunk
    layers = [(512, 1024, 1), (1024, 512, 1), (256, 512, 3), (512, 256, 3), (128, 256, 3), (256, 128, 3), (64, 128, 3),
              (128, 64, 3), (32, 64, 2), (64, 32, 11), (32, 32, 7)]
    for k, n, calls in layers:
        x, dy, w, b = rnd(M, k), rnd(M, n), rnd(n, k) / k ** 0.5, rnd(n)
        y, dx = torch.empty(M, n, device=dev), torch.empty(M, k, device=dev)
        dw, db = torch.empty(n, k, device=dev), torch.empty(n, device=dev)
        fl = 2.0 * M * k * n
        for name, fn in (("fwd", lambda: mlp.linear_fwd_(x, w, b, y, relu_out=True)),
                         ("dgrad", lambda: mlp.linear_dgrad_(dy, w, dx, mask=x)),
                         ("wgrad", lambda: mlp.linear_wgrad_(dy, x, dw, db))):
            us = timed(fn)
            total += us * calls
            report(f"linear_{name}[K={k},N={n}] x{calls}", us, fl)
if args.only in ("", "conv"):
    # (Cin, Cout, H, calls per step) of the 3x3 convolutions
    convs = [(32, 32, 256, 3), (32, 64, 256, 1), (64, 64, 256, 2), (64, 128, 128, 1), (128, 128, 128, 2), (128, 256, 64, 1),
             (256, 256, 64, 2), (256, 512, 32, 1), (512, 512, 32, 1), (512, 256, 64, 1), (256, 128, 128, 1), (128, 64, 256, 1),
             (64, 32, 256, 1), (32, 64, 512, 1), (64, 128, 512, 1), (128, 64, 512, 1)]
    for cin, cout, h, calls in convs:
        x, gy = cl(rnd(1, cin, h, h)), cl(rnd(1, cout, h, h))
        w, b = cl(rnd(cout, cin, 3, 3) / (9 * cin) ** 0.5), rnd(cout)
        y, dx = grid._empty_cl(1, cout, h, h, dev), grid._empty_cl(1, cin, h, h, dev)
        dw, db = torch.empty_like(w), torch.empty(cout, device=dev)
        fl = 2.0 * 9 * cin * cout * h * h
        for name, fn in (("fwd", lambda: grid.conv3x3_fwd_(x, w, b, y, relu=True)),
                         ("dgrad", lambda: grid.conv3x3_dgrad_(gy, w, dx, mask=x)),
                         ("wgrad", lambda: grid.conv3x3_wgrad_(gy, x, dw, db))):
            us = timed(fn)
            total += us * calls
            report(f"conv3x3_{name}[{cin}->{cout},{h}x{h}] x{calls}", us, fl)
if args.only in ("", "upconv"):
    # (Cin, Cout, H of the INPUT plane, calls per step) of the 2x2 stride-2 transposed convolutions (alto.py:175,215-218)
    from tomosar2height_amd import _lib
    for cin, cout, h, calls in [(512, 256, 32, 2), (256, 128, 64, 2), (128, 64, 128, 2)]:
        x, gy = cl(rnd(1, cin, h, h)), cl(rnd(1, cout, 2 * h, 2 * h))
        w = (rnd(cin, cout, 2, 2) / (4 * cin) ** 0.5).contiguous(memory_format=torch.channels_last)
        b = rnd(cout)
        y, dx, dw = grid._empty_cl(1, cout, 2 * h, 2 * h, dev), grid._empty_cl(1, cin, h, h, dev), torch.empty_like(w)
        lib = _lib.load()
        ws_d = _lib.workspace(lib.t2h_upconv2x2_dgrad_workspace_bytes(1, h, h, cin, cout), dev)
        ws_w = _lib.workspace(lib.t2h_upconv2x2_wgrad_workspace_bytes(1, h, h, cin, cout), dev)
        fl = 2.0 * 4 * cin * cout * h * h
        wf, wft = grid.split_weights.get_up(w, True), grid.split_weights.get_up(w, False)
        ws_b = _lib.workspace(lib.t2h_upconv2x2_bx3_dgrad_workspace_bytes(1, h, h, cin, cout), dev)
        ws_bw = _lib.workspace(lib.t2h_upconv2x2_bx3_wgrad_workspace_bytes(1, h, h, cin, cout), dev)
        db = torch.empty(cout, device=dev)
        for name, fn in (
                ("bx3_wgrad", lambda: _lib.call("t2h_upconv2x2_bx3_wgrad", _lib.ptr(gy), cout, _lib.ptr(x), _lib.ptr(dw), _lib.ptr(db), 1, h, h, cin, cout, 0, _lib.ptr(ws_bw), ws_bw.numel(), _lib.stream())),
                ("bx3_fwd", lambda: _lib.call("t2h_upconv2x2_bx3_fwd", _lib.ptr(x), _lib.ptr(wf), _lib.ptr(b), None, _lib.ptr(y), 1, h, h, cin, cout, 0, _lib.stream())),
                ("bx3_dgrad", lambda: _lib.call("t2h_upconv2x2_bx3_dgrad", _lib.ptr(gy), cout, _lib.ptr(wft), _lib.ptr(dx), 1, h, h, cin, cout, 0, _lib.ptr(ws_b), ws_b.numel(), _lib.stream())),
                ("fwd", lambda: _lib.call("t2h_upconv2x2_fwd_add", _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(y), 1, h, h, cin, cout, 0, _lib.stream())),
                ("dgrad", lambda: _lib.call("t2h_upconv2x2_dgrad", _lib.ptr(gy), _lib.ptr(w), _lib.ptr(dx), 1, h, h, cin, cout, 0, _lib.ptr(ws_d), ws_d.numel(), _lib.stream())),
                ("wgrad", lambda: _lib.call("t2h_upconv2x2_wgrad", _lib.ptr(gy), _lib.ptr(x), _lib.ptr(dw), 1, h, h, cin, cout, 0, _lib.ptr(ws_w), ws_w.numel(), _lib.stream()))):
            us = timed(fn)
            total += us * calls
            report(f"upconv2x2_{name}[{cin}->{cout},{h}x{h}] x{calls}", us, fl)
print(f"sum over one step's calls: {total / 1e3:.2f} ms")
if args.only in ("", "small"):
    # the image U-Net's first layer (encoder/unet.py:112-187): Conv2d(3, 32, 3, padding=1) at 512 x 512
    from tomosar2height_amd import _lib
    cin, cout, h = 3, 32, 512
    x, gy = cl(rnd(1, cin, h, h)), cl(rnd(1, cout, h, h))
    w, b = cl(rnd(cout, cin, 3, 3) / 5.2), rnd(cout)
    y, dw, db = grid._empty_cl(1, cout, h, h, dev), torch.empty(cout, cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last), torch.empty(cout, device=dev)
    ws = _lib.workspace(_lib.load().t2h_conv3x3_smallcin_wgrad_workspace_bytes(cin, cout), dev)
    fl = 2.0 * 9 * cin * cout * h * h
    for name, fn in (("fwd", lambda: _lib.call("t2h_conv3x3_smallcin_fwd", _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), 1, h, h, cin, cout, 2, _lib.stream())),
                     ("wgrad", lambda: _lib.call("t2h_conv3x3_smallcin_wgrad", _lib.ptr(gy), _lib.ptr(x), _lib.ptr(dw), _lib.ptr(db), 1, h, h, cin, cout, 0, _lib.ptr(ws), ws.numel(), _lib.stream()))):
        report(f"conv3x3_smallcin_{name}[{cin}->{cout},{h}x{h}]", timed(fn), fl)
