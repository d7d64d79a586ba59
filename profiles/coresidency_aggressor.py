"""r05: synthetic single-instruction aggressors (profiles/coresidency_aggressor.hip) on a second stream beside the PRE-FIX forward walk.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC profiles/coresidency_aggressor.hip -o scratch/libaggr.so
    T2H_LIBRARY=<library built before the fix> python profiles/coresidency_aggressor.py
Result (profiles/r05_coresidency.txt): a pure loop of v_mfma_f32_16x16x32_f16 changes 13-23 % of the walks; 32x32x16 f16 / bf16, fp32 MFMA,
VALU compare loops and LDS traffic 0 of 100 each."""
import sys, os, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import torch
from detinit import synth_cloud
from tomosar2height_amd import _lib
from tomosar2height_amd.tile import TileIndex
dev = torch.device("cuda:0")
ag = ctypes.CDLL(os.path.abspath(os.environ.get("T2H_AGGR_LIB", "scratch/libaggr.so")))
ag.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
tile = TileIndex(synth_cloud(40000, seed=703).to(dev), 128)
level, C = 3, 1024
r = 128 >> level
q = torch.randn(r * r, C, device=dev)
rows = tile.B << (2 * tile.nbits)
def outs():
    return (torch.zeros(rows, C, device=dev), torch.zeros(rows // 4, C, device=dev), torch.zeros(tile.n_points * (C // 256) * 4, dtype=torch.int64, device=dev))
has_order = hasattr(tile, "cell_order")
order = tile.cell_order(level)
def walk(o):
    _lib.call("t2h_sample_relu_cellsums_ordered", _lib.ptr(q), _lib.ptr(tile.pts), tile.dim, _lib.ptr(tile.off0), tile.B, tile.N, tile.nbits,
              level, 0, C, o[0].data_ptr(), C, o[1].data_ptr(), C, o[2].data_ptr(), _lib.ptr(order), _lib.stream())
ref = outs(); walk(ref)
res = [outs() for _ in range(4)]
dummy = torch.zeros(16, device=dev)
torch.cuda.synchronize()
A, B = torch.cuda.Stream(), torch.cuda.Stream()
names = ["mfma f32 <- f16 32x32x16", "mfma f32 <- bf16 32x32x16", "mfma f32 32x32x2 (fp32)", "mfma f32 <- f16 16x16x32", "VALU compares/selects", "LDS traffic"]
for which, nm in enumerate(names):
    for blocks, iters in ((512, 2000), (2048, 500)):
        bad = 0
        for it in range(25):
            main = torch.cuda.current_stream(); A.wait_stream(main); B.wait_stream(main)
            with torch.cuda.stream(B):
                for _ in range(4):
                    assert ag.aggr_launch(which, dummy.data_ptr(), blocks, iters, torch.cuda.current_stream().cuda_stream) == 0
            with torch.cuda.stream(A):
                for o in res: walk(o)
            torch.cuda.synchronize()
            bad += sum(not all(torch.equal(x, y) for x, y in zip(ref, o)) for o in res)
        print(f"{nm:28s} grid {blocks:5d} x {iters:5d} iterations: walks that differ {bad} of 100")
