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


def colors_by_key(keys, colors):
    order = np.argsort(keys, kind="stable")
    return keys[order], colors[order]


def coloring_valid(keys, colors, dynamic):
    """No dynamic body appears twice within one colour; every pair coloured."""
    a = (keys >> np.uint64(32)).astype(np.int64)
    b = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
    if np.any(colors >= 64):
        return False
    for end in (a, b):
        pass
    ends = np.concatenate([a, b])
    cols = np.concatenate([colors, colors]).astype(np.int64)
    dyn = dynamic[ends]
    code = ends[dyn] * 64 + cols[dyn]
    return len(np.unique(code)) == len(code)
