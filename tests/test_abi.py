"""CPU tests of the drop-in boundary: the built libsfx.so exports every symbol include/sfx.h
declares, the ctypes layouts match the C header, and without a GPU the library fails loudly
(no fallback).  No compute call is made here."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

from starframe_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sfx.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sfx_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_all_exported(libsfx):
    syms = declared_symbols()
    assert len(syms) >= 24
    assert sorted(_abi.SFX_SYMBOLS) == syms, "starframe_amd/_abi.py symbol list out of date"
    for s in syms:
        assert hasattr(libsfx, s), f"libsfx.so does not export {s}"
    assert libsfx.sfx_abi_version() == 12


def test_rccl_library_exports_its_header():
    """libsfx_rccl.so (the halo exchange over plain rccl.h) exports every symbol include/sfx_rccl.h declares.  Loading it
    pulls in librccl and libamdhip64; no call is made without a GPU."""
    src = open(os.path.join(ROOT, "include", "sfx_rccl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    syms = sorted(set(re.findall(r"\b(sfx_rccl_[a-z_0-9]+)\s*\(", src)))
    assert sorted(_abi.SFX_RCCL_SYMBOLS) == syms
    lib = _abi.load_libsfx_rccl()
    for s in syms:
        assert hasattr(lib, s), s


def test_rust_mirror_declares_every_symbol_and_size():
    """bindings/rust/src/lib.rs cannot be compiled here (no rustc): its text is held against the headers instead.
    Every function of sfx.h / sfx_rccl.h is declared, with as many parameters as the header gives it; every struct
    size the header asserts is asserted there with the same number; every field of every struct appears in order."""
    rs = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    rs_nc = re.sub(r"//[^\n]*", "", rs)
    for header in ("sfx.h", "sfx_rccl.h"):
        src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
        for name, args in re.findall(r"\b(sfx_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", src):
            m = re.search(r"pub fn %s\(([^)]*)\)" % name, rs_nc)
            assert m, f"{name} is not declared in the Rust mirror"
            n_c = 0 if args.strip() in ("", "void") else args.count(",") + 1
            n_rs = 0 if not m.group(1).strip() else m.group(1).count(",") + 1
            assert n_c == n_rs, f"{name}: {n_c} parameters in {header}, {n_rs} in lib.rs"
    src = open(HEADER).read()
    abi = int(re.search(r"SFX_ABI_VERSION (\d+)u", src).group(1))
    assert f"SFX_ABI_VERSION: u32 = {abi};" in rs
    sizes = re.findall(r"SFX_STATIC_ASSERT\(sizeof\((\w+)\) == ([^,]+),", src)
    assert len(sizes) >= 9
    for name, expr in sizes:
        expr = expr.strip()
        assert f"size_of::<{name}>() == {expr})" in rs, f"size of {name} ({expr}) is not asserted in lib.rs"
    body = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for name, fields in re.findall(r"typedef struct (\w+) \{(.*?)\} \1;", body, flags=re.S):
        m = re.search(r"pub struct %s \{(.*?)\n\}" % name, rs_nc, flags=re.S)
        assert m, name
        c_fields = [f for decl in fields.split(";") for f in re.findall(r"(\w+)(?:\[[^\]]*\])*\s*(?:,|$)", decl.strip())]
        rs_fields = [f.rstrip("_") for f in re.findall(r"pub (\w+):", m.group(1))]
        assert rs_fields == c_fields, f"{name}: {c_fields} in sfx.h, {rs_fields} in lib.rs"


def test_struct_layouts_match_header():
    prog = r"""
#include <stdio.h>
#include <stddef.h>
#include "sfx.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(sfx_config), sizeof(sfx_collider),
         sizeof(sfx_material), sizeof(sfx_distance_constraint), sizeof(sfx_contact), sizeof(sfx_frame_stats),
         sizeof(sfx_kernel_time), offsetof(sfx_config, flags), offsetof(sfx_frame_stats, color_count),
         (size_t)SFX_K_CLASSES, sizeof(sfx_ray_hit), sizeof(sfx_angle_constraint),
         offsetof(sfx_distance_constraint, damping));
  return 0;
}
"""
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = list(map(int, out))
    assert sizes[0] == C.sizeof(_abi.SfxConfig)
    assert sizes[1] == _abi.collider_dtype.itemsize
    assert sizes[2] == _abi.material_dtype.itemsize
    assert sizes[3] == _abi.constraint_dtype.itemsize
    assert sizes[4] == _abi.contact_dtype.itemsize
    assert sizes[5] == C.sizeof(_abi.SfxFrameStats)
    assert sizes[6] == C.sizeof(_abi.SfxKernelTime)
    assert sizes[7] == _abi.SfxConfig.flags.offset
    assert sizes[8] == _abi.SfxFrameStats.color_count.offset
    assert sizes[9] == _abi.K_CLASSES
    assert sizes[10] == _abi.ray_hit_dtype.itemsize
    assert sizes[11] == _abi.angle_constraint_dtype.itemsize
    assert sizes[12] == _abi.constraint_dtype.fields["damping"][1]


def test_default_config(libsfx):
    cfg = _abi.SfxConfig()
    libsfx.sfx_default_config(C.byref(cfg))
    assert cfg.struct_size == C.sizeof(_abi.SfxConfig)
    assert cfg.substeps == 4 and abs(cfg.gravity[1] + 9.81) < 1e-6
    assert cfg.color_rounds == 12 and cfg.tile_cells == 0   # 0: automatic (one tile for small worlds, else 2 cells)


def test_create_fails_loudly_without_gpu(libsfx):
    """On a box without a GPU there is no CPU fallback behind the ABI."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cfg = _abi.make_config()
    h = C.c_void_p()
    rc = libsfx.sfx_create(C.byref(cfg), C.byref(h))
    assert rc == _abi.SFX_ERR_DEVICE and not h.value
    assert b"no CPU fallback" in libsfx.sfx_last_error(None)
    from starframe_amd.world import SfxError, World

    with pytest.raises(SfxError):
        World()


def test_bad_config_rejected(libsfx):
    cfg = _abi.make_config()
    cfg.struct_size = 12
    h = C.c_void_p()
    assert libsfx.sfx_create(C.byref(cfg), C.byref(h)) == _abi.SFX_ERR_INVALID
    assert libsfx.sfx_create(None, C.byref(h)) == _abi.SFX_ERR_INVALID
    libsfx.sfx_destroy(None)  # no-op


def test_product_package_does_not_touch_the_oracle():
    """Nothing under starframe_amd/ may import, link or execute the oracle."""
    pkg = os.path.join(ROOT, "starframe_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', text), f
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert "libsfx_oracle" not in text and "dlopen" not in text, f
    out = subprocess.check_output(["ldd", _abi.LIBSFX_PATH]).decode()
    assert "oracle" not in out


def test_shipped_library_is_built_from_the_tree_s_sources(libsfx):
    """what was compiled must be unambiguous.  The library carries the hash of the sources it was
    compiled from (sfx_build_id); it equals the hash of the tree's sources, so build() has nothing to do."""
    from starframe_amd import build

    assert build.built_id() == build.source_hash(), "starframe_amd/csrc/libsfx.so was not built from this tree: run __graft_entry__.build()"
    assert not build.needs_build()


def test_forced_rebuild_compiles_the_tree(tmp_path):
    """build(force=True) compiles whatever is there (hipcc cross-compiles gfx950 without a GPU): the fresh library carries
    this tree's build id and exports the whole header."""
    from starframe_amd import build

    out = str(tmp_path / "libsfx_forced.so")
    assert build.build(force=True, out=out) == out
    assert build.built_id(out) == build.source_hash()
    lib = C.CDLL(out)
    for s in declared_symbols():
        assert hasattr(lib, s), s
