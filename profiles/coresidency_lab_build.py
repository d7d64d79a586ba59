"""Build the A/B libraries of profiles/coresidency_repro.py and coresidency_trunk_first.py into profiles/_lab/ (git-ignored; they
travel to the GPU box with the snapshot).  Run in the build container after `python -m tomosar2height_amd.csrc.build`."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tomosar2height_amd.csrc import build as b

lab = os.path.join(ROOT, "profiles", "_lab")
os.makedirs(lab, exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                os.path.join(ROOT, "profiles", "coresidency_aggressor.hip"), "-o", os.path.join(lab, "libaggr.so")], check=True)
pad = int(os.environ.get("T2H_LAB_PAD", "4"))
b.build_variant(os.path.join(lab, "libt2h_select.so"), {"point_grid": {"defines": ["T2H_TAPS_BY_SELECT"]}})
b.build_variant(os.path.join(lab, f"libt2h_select_pad{pad}.so"), {"point_grid": {"defines": ["T2H_TAPS_BY_SELECT"], "pad": pad}})
b.build_variant(os.path.join(lab, f"libt2h_trunk_pad{pad}.so"), {"trunk": {"pad": pad}})
b.build_variant(os.path.join(lab, "libt2h_trunk_ablate.so"), {"trunk": {"defines": ["T2H_TRUNK_ABLATE"]}})      # profiles/trunk_fused_probe.py
b.build_variant(os.path.join(lab, "libt2h_trunk_nobarrier.so"), {"trunk": {"defines": ["T2H_LAB_NO_PROLOGUE_BARRIER"]}})   # r06_coresidency.txt section 8
print(sorted(os.listdir(lab)))
