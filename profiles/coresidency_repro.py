"""r06 stand-alone reproducer of the r05 co-residency fault (profiles/r05_coresidency.txt), and the A/B that decides what kind of fault it is.

Victim: the forward walk (point_grid.hip, sample_relu_cellsums_v2_kernel) built with -DT2H_TAPS_BY_SELECT, i.e. the pre-fix form
    ne = x1ok ? wx1 * wy0 : 0     ->   v_cmp_gt_i32_e64 s[0:1] .. ; s_and_b64 vcc, s[0:1], vcc ; .. 4 instructions .. ; v_cndmask_b32_e64 v20, 0, v20, s[0:1]
Aggressor: a loop of v_mfma_f32_16x16x32_f16 on eight independent accumulators on a second stream (coresidency_aggressor.hip, case 3),
touching no memory.  Count: walks whose output differs from the walk alone, bit for bit.

    python -m tomosar2height_amd.csrc.build ; python profiles/coresidency_lab_build.py       (build container: writes profiles/_lab/*.so)
    T2H_LIBRARY=profiles/_lab/libt2h_select.so       python profiles/coresidency_repro.py    -> differs (r05: 13-23 of 100)
    T2H_LIBRARY=profiles/_lab/libt2h_select_pad4.so  python profiles/coresidency_repro.py    -> the same kernel with `s_nop 4` behind EVERY VALU
                                                         write of an SGPR / VCC (csrc/isa_pass.pad): if a missing wait state were the cause, 0
    (no T2H_LIBRARY)                                  python profiles/coresidency_repro.py    -> the shipped kernel (factors by arithmetic): 0
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
from detinit import synth_cloud
from tomosar2height_amd import _lib
from tomosar2height_amd.tile import TileIndex

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
ag = ctypes.CDLL(os.path.join(ROOT, "profiles", "_lab", "libaggr.so"))
ag.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
tile = TileIndex(synth_cloud(40000, seed=703).to(dev), 128)
level, C = 3, 1024
r = 128 >> level
q = torch.randn(r * r, C, device=dev)
rows = tile.B << (2 * tile.nbits)
order = tile.cell_order(level)


def outs():
    return (torch.zeros(rows, C, device=dev), torch.zeros(rows // 4, C, device=dev),
            torch.zeros(tile.n_points * (C // 256) * 4, dtype=torch.int64, device=dev))


def walk(o):
    _lib.call("t2h_sample_relu_cellsums_ordered", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N,
              tile.nbits, level, 0, C, o[0].data_ptr(), C, o[1].data_ptr(), C, o[2].data_ptr(), _lib.ptr(order), _lib.stream())


ref = outs()
walk(ref)
res = [outs() for _ in range(4)]
dummy = torch.zeros(16, device=dev)
torch.cuda.synchronize()
A, B = torch.cuda.Stream(), torch.cuda.Stream()
total = 0
for which, name in ((3, "v_mfma_f32_16x16x32_f16"), (0, "v_mfma_f32_32x32x16_f16")):
    for blocks, iters in ((512, 2000), (2048, 500)):
        bad, worst = 0, 0.0
        for _ in range(rounds):
            main = torch.cuda.current_stream()
            A.wait_stream(main)
            B.wait_stream(main)
            with torch.cuda.stream(B):
                for _ in range(4):
                    assert ag.aggr_launch(which, dummy.data_ptr(), blocks, iters, torch.cuda.current_stream().cuda_stream) == 0
            with torch.cuda.stream(A):
                for o in res:
                    walk(o)
            torch.cuda.synchronize()
            for o in res:
                if not all(torch.equal(x, y) for x, y in zip(ref, o)):
                    bad += 1
                    worst = max(worst, float((o[0] - ref[0]).abs().max() / ref[0].abs().max()))
        total += bad
        print(f"{os.environ.get('T2H_LIBRARY', 'shipped library'):44s} beside {name:26s} grid {blocks:5d} x {iters:5d}: "
              f"{bad} of {4 * rounds} walks differ (worst |diff| / max |ref| = {worst:.2e})", flush=True)
sys.exit(1 if total else 0)
