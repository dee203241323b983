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


def source_hash():
    """First 16 hex digits of the SHA-256 over the library's sources (contents, in a fixed order): compiled into the
    library as SFX_BUILD_ID, returned by sfx_build_id()."""
    import hashlib

    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def built_id(path=None):
    """The build id of a built library -- what its sfx_build_id() returns -- read from the file (None when it is missing or
    older than the build id).  Not by loading it: a process that asks, rebuilds and then loads the library would get the
    copy it loaded first."""
    path = path or OUT
    if not os.path.exists(path):
        return None
    marker = b"sfx-build-id:"
    with open(path, "rb") as fh:
        data = fh.read()
    at = data.find(marker)
    if at < 0:
        return None
    end = data.find(b"\0", at)
    return data[at + len(marker):end].decode(errors="replace")


def needs_build():
    """The shipped library must BE the tree's: not "newer than its sources" (timestamps do not survive a copy) but built
    from exactly these sources."""
    return built_id() != source_hash()


RACES_OUT = os.path.join(CSRC, "libsfx_races.so")


def build_races(force=False, verbose=False):
    """libsfx_races.so: the same library with -DSFX_DEBUG_RACES (sfx_kernels.h "race detector"): every write of a body's
    record inside a colour step is checked on the device against a second writer of that colour.  Loaded by
    tests/test_gpu_races.py only (SFX_LIBRARY); never by the product path."""
    if not force and built_id(RACES_OUT) == source_hash():
        return RACES_OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ['-DSFX_BUILD_ID="%s"' % source_hash(), "-DSFX_DEBUG_RACES", "-o", RACES_OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return RACES_OUT


def build(force=False, verbose=False, extra=(), out=None):
    """Compiles libsfx.so (+ libsfx_rccl.so next to it when out is the in-tree path).  force: compile whatever is there."""
    target = out or OUT
    if not force and out is None and not needs_build():
        build_rccl(verbose=verbose)
        build_races(verbose=verbose)
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ['-DSFX_BUILD_ID="%s"' % source_hash()] + list(extra) + ["-o", target] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if out is None:
        build_rccl(force=True, verbose=verbose)
        build_races(force=force, verbose=verbose)
    return target


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, extra=[a for a in sys.argv[1:] if a != "--force"])
