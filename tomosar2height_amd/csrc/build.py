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


LLVM = os.environ.get("T2H_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-inline-asm"]


def compile_through_assembly(src, obj, workdir, transform, defines=(), hipcc=None):
    """One source compiled with a pass over its gfx950 ASSEMBLY in between (`transform(path_in, path_out)`): device code to
    text, the pass, assembler, lld, offload bundle, then the host half with that bundle embedded -- the steps `hipcc -c` runs
    itself, taken apart.  Used by the A/B builds of profiles/coresidency_lab.py (isa_pass.pad) and available for any check that
    needs the instruction stream the chip will run."""
    hipcc = hipcc or os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(workdir, exist_ok=True)
    stem = os.path.join(workdir, os.path.basename(src)[:-4])
    defs = [f"-D{d}" for d in defines]
    run = lambda cmd: subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL if "-S" in cmd else None)
    run([hipcc] + FLAGS + defs + ["--cuda-device-only", "-S", src, "-o", stem + ".s"])
    transform(stem + ".s", stem + ".pass.s")
    run([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={ARCH}", "-c", stem + ".pass.s",
         "-o", stem + ".dev.o"])
    run([os.path.join(LLVM, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", stem + ".dev.o", "-o",
         stem + ".hsaco"])
    run([os.path.join(LLVM, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
         f"-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{ARCH}", "-input=/dev/null", f"-input={stem}.hsaco",
         f"-output={stem}.hipfb"])
    run([hipcc] + FLAGS + defs + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", stem + ".hipfb", "-c", src,
                                   "-o", obj])
    return obj


def build_variant(out, overrides):
    """A second library of the same ABI (for `T2H_LIBRARY=`): the objects of the main build, except the sources named in
    `overrides` = {"point_grid": {"defines": [...], "pad": N or None, "only": regex}} which are recompiled with extra defines and /
    or with `s_nop N` behind every VALU write of an SGPR (isa_pass.pad)."""
    from . import isa_pass
    build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    work = os.path.join(os.path.dirname(os.path.abspath(out)), "_obj_" + os.path.basename(out).replace(".so", ""))
    os.makedirs(work, exist_ok=True)
    objs = []
    for src in sources():
        name = os.path.basename(src)[:-4]
        o = overrides.get(name)
        if o is None:
            objs.append(src[:-4] + ".o")
            continue
        obj = os.path.join(work, name + ".o")
        if o.get("pad") is not None:
            compile_through_assembly(src, obj, work, lambda a, b, o=o: isa_pass.pad(a, b, o["pad"], o.get("only")),
                                     o.get("defines", ()), hipcc)
        else:
            subprocess.run([hipcc] + FLAGS + [f"-D{d}" for d in o.get("defines", ())] + ["-c", src, "-o", obj], check=True)
        objs.append(obj)
    subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs, check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
