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


def _compile(args):
    src, verbose, hipcc = args
    obj = src[:-4] + ".o"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
           "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-c", src, "-o", obj]
    if verbose:
        cmd.insert(-4, "-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    """Objects are rebuilt only when their source (or any shared header) is newer; sources compile in parallel."""
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = glob.glob(os.path.join(HERE, "*.h")) + \
        [os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "t2h.h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    todo, objs = [], []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            todo.append((src, verbose, hipcc))
    if todo:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(todo), int(os.environ.get("T2H_BUILD_JOBS", "6")))) as pool:
            list(pool.map(_compile, todo))
    subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT] + objs, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
