import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
_extra = sys.argv[1:]
sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--skip-cpu-baseline", "--profile-steps", "0", "--sustain-s", "0"]
import bench
# monkeypatch: time each step of a fresh trainer
import tomosar2height_amd.trainer as T
orig = T.Trainer.train_step
times = []
def timed(self, data):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(self, data)
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    return r
T.Trainer.train_step = timed
sys.argv = ["bench.py", "--steps", "60", "--warmup", "0", "--skip-cpu-baseline", "--profile-steps", "0", "--sustain-s", "0"] + _extra
bench.main()
print("per-step ms (synchronised):", " ".join(f"{t:.1f}" for t in times), file=sys.stderr)
print("reserved MB", torch.cuda.memory_reserved() / 1e6, "alloc retries", torch.cuda.memory_stats().get("num_alloc_retries"), "segments", torch.cuda.memory_stats().get("segment.all.allocated"), file=sys.stderr)
