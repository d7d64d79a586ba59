"""Compile libt2h_hip.so (the C-ABI library of include/t2h.h) for gfx950 with hipcc, in-tree.

    python -m tomosar2height_amd.csrc.build [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "libt2h_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(HERE, "*.h")) + \
        [os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "t2h.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in sources():
        obj = src[:-4] + ".o"
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
               "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-c", src, "-o", obj]
        if verbose:
            cmd.insert(-4, "-Rpass-analysis=kernel-resource-usage")
        subprocess.run(cmd, check=True)
        objs.append(obj)
    subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT] + objs, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
