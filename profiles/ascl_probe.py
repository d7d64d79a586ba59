import os, sys, collections, traceback, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tomosar2height_amd import grid
orig = grid._as_cl
seen = collections.Counter()
def spy(x):
    if not x.is_contiguous(memory_format=torch.channels_last):
        fr = traceback.extract_stack(limit=4)[:-1]
        seen[(tuple(x.shape), tuple(x.stride()), " <- ".join(f"{f.name}:{f.lineno}" for f in fr))] += 1
    return orig(x)
grid._as_cl = spy
sys.argv = ["bench.py", "--steps", "10", "--warmup", "2", "--skip-cpu-baseline", "--profile-steps", "0", "--sustain-s", "0"]
import bench
bench.main()
for k, v in seen.most_common(20): print(v, k, file=sys.stderr)
