"""Shared comparison helpers for the parity tests."""
import numpy as np


def rel_err(a, b, floor=1e-3):
    """max |a-b| / max(|b|, floor): relative error with an absolute floor for values near 0."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def ulp_diff(a, b):
    """max distance in units in the last place between two f32 arrays."""
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return int(np.max(np.abs(a - b))) if a.size else 0


def colors_by_key(keys, colors, tiles=None):
    """(keys, colours[, owner tiles]) sorted by pair key."""
    order = np.argsort(keys, kind="stable")
    if tiles is None:
        return keys[order], colors[order]
    return keys[order], colors[order], tiles[order]


PRUNED_COLOR = 0xFE


def coloring_valid(keys, colors, dynamic, tiles=None):
    """No dynamic body appears twice within one colour of one batch (tile / fallback list);
    every pair within reach coloured (PRUNED_COLOR marks the pairs the reach test dropped)."""
    live = colors != PRUNED_COLOR
    keys, colors = keys[live], colors[live]
    if tiles is not None:
        tiles = tiles[live]
    a = (keys >> np.uint64(32)).astype(np.int64)
    b = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
    if np.any(colors >= 64):
        return False
    t = np.zeros(len(keys), np.int64) if tiles is None else tiles.astype(np.int64)
    ends = np.concatenate([a, b])
    cols = np.concatenate([colors, colors]).astype(np.int64)
    tt = np.concatenate([t, t])
    dyn = dynamic[ends]
    code = np.stack([tt[dyn], ends[dyn], cols[dyn]], axis=1)
    return len(np.unique(code, axis=0)) == len(code)


def unit_class(tiles):
    """Solver units are reported as class << 28 | owner tile (0 interior, 1 x seam, 2 y seam, 3 corner)."""
    return (tiles >> np.uint32(28)).astype(np.int64)


def tile_phases_disjoint(keys, tiles, dynamic, grid_log2=None, tile_cells=None):
    """The units (tile, class) of one class -- one phase of a sweep -- never share a dynamic body."""
    a = (keys >> np.uint64(32)).astype(np.int64)
    b = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
    ok = tiles != 0xFFFFFFFF
    t = tiles[ok].astype(np.int64)
    ph = unit_class(tiles[ok])
    e2 = np.concatenate([a[ok], b[ok]])
    t2 = np.concatenate([t, t])
    p2 = np.concatenate([ph, ph])
    d = dynamic[e2]
    code = np.unique(np.stack([p2[d], e2[d], t2[d]], axis=1), axis=0)
    # a (phase, body) may appear with one unit only
    return len(np.unique(code[:, :2], axis=0)) == len(code)
