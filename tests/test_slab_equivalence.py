"""The interior step's one-float form of the slab test (trace_math.h slab_entry_key: entry distance or
+inf, then `key < raytMax`) against the oracle's restatement of Bounds3::IntersectP
(util/vecmath.h:1573-1608, pinned to the compiled reference by tests/golden/leaf_slab.npz) on
adversarial inputs: zero / infinite / NaN ray components, origins exactly on slab planes, flat boxes,
denormals, negative and infinite tMax.  float32 numpy restates the device arithmetic operation for
operation (np.fmax / np.fmin ignore a NaN operand as v_max_f32 / v_min_f32 do)."""
import numpy as np
import pytest

from oracle_binding import leaf_batch

F = np.float32
GAMMA3 = F(3) * F(2.0 ** -24) / (F(1) - F(3) * F(2.0 ** -24))
WIDEN = F(1) + F(2) * GAMMA3


def entry_key(box, o, d):
    """slab_entry_key on [n] boxes (min xyz, max xyz) and rays; float32 throughout."""
    with np.errstate(all="ignore"):
        inv = F(1) / d
        neg = inv < 0
        lo, hi = box[:, 0:3], box[:, 3:6]
        a = (np.where(neg, hi, lo) - o) * inv
        b = (np.where(neg, lo, hi) - o) * inv
        amax = np.fmax(np.fmax(a[:, 0], a[:, 1]), a[:, 2])
        bmin = np.fmin(np.fmin(b[:, 0], b[:, 1]), b[:, 2]) * WIDEN  # widened once: the product is monotone
        ok = ~(np.isnan(a[:, 0]) | np.isnan(b[:, 0])) & (amax <= bmin) & (bmin > 0)
        return np.where(ok, amax, F(np.inf)).astype(F)


SPECIAL = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3.4e38, -3.4e38,
                    0.5, 2.0, -2.0, 1e-20, 1e20], np.float32)


def adversarial(rng, n):
    """Boxes with min <= max per axis (what scene creation guarantees) and rays drawn to sit on the
    test's edge cases."""
    lo = rng.integers(-3, 4, (n, 3)).astype(F)
    ext = rng.choice(np.array([0, 1, 2, 4, 8, 0.5, 1e-3], F), (n, 3))
    hi = lo + ext
    box = np.concatenate([lo, hi], 1).astype(F)
    # origins: on a slab plane, inside, outside, special values
    o = rng.integers(-4, 5, (n, 3)).astype(F)
    inside = rng.random((n, 1)) < 0.3
    o = np.where(inside, lo + ext * rng.choice(np.array([0, 0.5, 1], F), (n, 3)), o)
    on_plane = rng.random((n, 3)) < 0.35
    o = np.where(on_plane, np.where(rng.random((n, 3)) < 0.5, lo, hi), o)
    o = np.where(rng.random((n, 3)) < 0.05, rng.choice(SPECIAL, (n, 3)), o).astype(F)
    d = rng.choice(np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.25, 3.0], F), (n, 3))
    d = np.where(rng.random((n, 3)) < 0.08, rng.choice(SPECIAL, (n, 3)), d).astype(F)
    tmax = rng.choice(np.array([np.inf, np.inf, np.inf, 1.0, 2.0, 0.0, -1.0, 0.5, 1e-45, 30.0, np.nan], F), n).astype(F)
    return box, o, d, tmax


def check(box, o, d, tmax):
    rec = np.concatenate([o, d, tmax[:, None], box], 1).astype(F)
    want, _ = leaf_batch("slab", rec)
    with np.errstate(all="ignore"):
        got = entry_key(box, o, d) < tmax
    bad = np.nonzero(got != (want != 0))[0]
    assert bad.size == 0, f"{bad.size} of {len(rec)} differ; first: box={box[bad[0]]} o={o[bad[0]]} d={d[bad[0]]} tmax={tmax[bad[0]]}"
    return int(want.sum())


def test_key_form_equals_reference_slab_on_adversarial_inputs():
    rng = np.random.default_rng(11)
    hits = 0
    for _ in range(8):
        hits += check(*adversarial(rng, 250_000))
    assert hits > 50_000  # the set is not all misses


def test_key_form_equals_reference_slab_on_random_inputs():
    rng = np.random.default_rng(12)
    n = 1_000_000
    c = rng.uniform(-10, 10, (n, 3)).astype(F)
    e = np.abs(rng.normal(0, 2, (n, 3))).astype(F)
    box = np.concatenate([c - e, c + e], 1).astype(F)
    o = rng.uniform(-20, 20, (n, 3)).astype(F)
    d = (c - o + rng.normal(0, 1.5, (n, 3)) * e).astype(F)  # aimed near the box: about half hit
    d[rng.random(n) < 0.2, rng.integers(0, 3)] = 0
    tmax = np.where(rng.random(n) < 0.5, np.inf, rng.uniform(0, 40, n)).astype(F)
    assert check(box, o, d, tmax) > 200_000


def test_golden_slab_vectors():
    """The committed vectors the oracle's slab test is pinned with (compiled-reference verdicts)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "leaf_slab.npz"))
    rec = g["inputs"].astype(F)
    box = rec[:, 7:13]
    valid = np.all(box[:, 0:3] <= box[:, 3:6], 1)
    if valid.sum() == 0:
        pytest.skip("no vectors with min <= max")
    with np.errstate(all="ignore"):
        got = entry_key(box[valid], rec[valid, 0:3], rec[valid, 3:6]) < rec[valid, 6]
    assert np.array_equal(got, g["hit"][valid] != 0)
