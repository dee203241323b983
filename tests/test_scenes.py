"""Synthetic scene generators: PRNG contract and shapes of the five BASELINE.json configs."""
import numpy as np

from starframe_amd import scenes
from starframe_amd._abi import SHAPE_CAPSULE, SHAPE_CIRCLE, SHAPE_RECT, WORLD_BODY


def splitmix64_ref(seed, count):
    M = (1 << 64) - 1
    out = []
    x = seed
    for _ in range(count):
        x = (x + 0x9E3779B97F4A7C15) & M
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        out.append((z >> 40) / float(1 << 24))
    return np.array(out, np.float32)


def test_splitmix64_contract():
    u = scenes.splitmix64_uniform(0x5EED0003, 1000)
    assert np.array_equal(u, splitmix64_ref(0x5EED0003, 1000))
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.03


def test_pyramid_config1():
    s = scenes.pyramid()
    assert s.n == 497 and s.n_dynamic() == 496 and s.substeps == 10
    assert np.all(s.colliders["type"] == SHAPE_RECT)


def test_circles_config2_and_determinism():
    a = scenes.circles_container()
    b = scenes.circles_container()
    assert a.n == 10_003 and a.n_dynamic() == 10_000
    assert np.array_equal(a.pos, b.pos) and np.array_equal(a.colliders, b.colliders)
    r = a.colliders["p"][:10_000, 2]
    assert 0.05 <= r.min() and r.max() <= 0.10


def test_mixed_pile_config3():
    s = scenes.mixed_pile(n=20_000)
    t = s.colliders["type"][:20_000]
    frac = [np.mean(t == k) for k in (SHAPE_CIRCLE, SHAPE_RECT, SHAPE_CAPSULE)]
    assert abs(frac[0] - 0.4) < 0.02 and abs(frac[1] - 0.3) < 0.02 and abs(frac[2] - 0.3) < 0.02
    assert s.substeps == 4 and s.n_dynamic() == 20_000
    assert np.allclose(np.hypot(s.rot[:, 0], s.rot[:, 1]), 1.0, atol=1e-6)
    assert np.all(s.inv_mass[:20_000] > 0) and np.all(s.inv_inertia[:20_000] > 0)


def test_chains_config4():
    s = scenes.chains(n_chains=10, links=100)
    assert s.n == 1000 and len(s.constraints) == 1000
    assert np.count_nonzero(s.constraints["body_a"] == WORLD_BODY) == 10
    assert np.allclose(s.constraints["rest_length"], 0.12)
    assert len(np.unique(s.colliders["group"])) == 10


def test_granular_config5_small():
    s = scenes.granular_slab(n=50_000, width=5000)
    assert s.n_dynamic() == 50_000
    assert s.pos[:50_000, 0].max() - s.pos[:50_000, 0].min() > 1000
