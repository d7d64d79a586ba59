#!/usr/bin/env python3
"""Where the HOST time of a tile-step goes: cProfile over bench.py's training loop with the autograd engine kept on the calling
thread (torch.autograd.set_multithreading_enabled(False)), so that the Python side of the backward is in the profile.
    python profiles/host_profile.py [bench.py flags]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.autograd.set_multithreading_enabled(False)
STEPS = 60
sys.argv = ["bench.py", "--steps", str(STEPS), "--warmup", "5", "--skip-cpu-baseline", "--profile-steps", "0", "--sustain-s", "0"] + sys.argv[1:]
import bench  # noqa: E402

pr = cProfile.Profile()
orig_run = None
# profile only the timed region: bench.main() calls run(); wrap Trainer.train_step
import tomosar2height_amd.trainer as T  # noqa: E402
inner = T.Trainer.train_step
calls = [0]


def wrapped(self, data):
    calls[0] += 1
    if calls[0] == 8:
        pr.enable()
    return inner(self, data)


T.Trainer.train_step = wrapped
bench.main()
pr.disable()
n = max(1, calls[0] - 7)
st = pstats.Stats(pr, stream=sys.stderr)
print(f"profiled {n} tile-steps", file=sys.stderr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumtime").print_stats("tomosar2height_amd", 40)
