"""Builds starframe_amd/csrc/libsfx.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libsfx.so")
SOURCES = ["sfx_host.hip"]
HEADERS = ["sfx_device.h", "sfx_kernels.h", os.path.join("..", "..", "include", "sfx.h")]

# -ffp-contract=off: the CPU oracle is compared bit-for-bit on AABBs, so no FMA contraction.
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
]


RCCL_OUT = os.path.join(CSRC, "libsfx_rccl.so")
RCCL_SRC = os.path.join(CSRC, "sfx_rccl.cpp")
RCCL_HEADERS = [os.path.join("..", "..", "include", "sfx_rccl.h"), os.path.join("..", "..", "include", "sfx.h")]


def build_rccl(force=False, verbose=False):
    """libsfx_rccl.so: the slab halo exchange over plain rccl.h (host code; links libsfx, librccl, libamdhip64)."""
    if not force and os.path.exists(RCCL_OUT):
        t = os.path.getmtime(RCCL_OUT)
        if all(os.path.getmtime(os.path.join(CSRC, f)) <= t for f in ["sfx_rccl.cpp"] + RCCL_HEADERS) and os.path.getmtime(OUT) <= t:
            return RCCL_OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", RCCL_OUT, RCCL_SRC, "-L" + CSRC, "-lsfx",
           "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return RCCL_OUT


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        build_rccl(verbose=verbose)
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra) + ["-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    build_rccl(force=True, verbose=verbose)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, extra=[a for a in sys.argv[1:] if a != "--force"])
