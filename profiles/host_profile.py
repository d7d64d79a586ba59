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
sys.argv = ["bench.py", "--steps", "40", "--warmup", "5", "--skip-cpu-baseline", "--profile-steps", "0", "--sustain-s", "0"] + sys.argv[1:]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr, stream=sys.stderr)
st.sort_stats("tottime").print_stats(40)
for name in ("method 'size'", "method 'contiguous'", "_cuda_getDeviceCount", "method 'to'", "torch.empty"):
    st.print_callers(name)
