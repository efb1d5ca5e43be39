"""Live cross-check of the oracle's leaf functions against the REFERENCE's own compiled code
(oracle/_ref/ref_leaf, built by oracle/Makefile from /root/reference sources) on fresh random
inputs every run-seed — beyond the fixed golden vectors.  Skipped where the binary is absent
(it is git-ignored; it exists in the build container and travels to the GPU box)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_leaf")

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/ref_leaf not built")


def run_ref(mode, recs, nout):
    recs = np.ascontiguousarray(recs, np.float32)
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "i.bin"), os.path.join(td, "o.bin")
        with open(fi, "wb") as f:
            f.write(np.int32(len(recs)).tobytes())
            f.write(recs.tobytes())
        subprocess.run([REF, mode, fi, fo], check=True)
        raw = np.fromfile(fo, np.uint32).reshape(len(recs), 1 + nout)
    return raw[:, 0].astype(np.uint8), raw[:, 1:]


def specials(rng, shape):
    """float32 values biased towards the awkward: exact zeros, +-0, tiny, huge, equal pairs."""
    v = rng.normal(size=shape) * 10.0 ** rng.integers(-6, 6, size=shape)
    pick = rng.random(shape)
    v = np.where(pick < 0.05, 0.0, v)
    v = np.where((pick >= 0.05) & (pick < 0.07), -0.0, v)
    v = np.where((pick >= 0.07) & (pick < 0.09), np.round(v), v)
    v = np.where((pick >= 0.09) & (pick < 0.10), 1e-40, v)
    return v.astype(np.float32)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_triangle_oracle_equals_reference_on_random_inputs(seed):
    rng = np.random.default_rng(seed)
    n = 20000
    recs = specials(rng, (n, 16))
    # half of the cases: aim the ray at a point of the triangle so that hits are common
    p = recs[:, 7:].reshape(n, 3, 3).astype(np.float64)
    b = rng.dirichlet([1, 1, 1], n)
    b[: n // 8, 0] = 0  # on an edge
    b /= b.sum(1, keepdims=True)
    tgt = np.einsum("nk,nkc->nc", b, p)
    aim = rng.random(n) < 0.6
    recs[aim, 3:6] = (tgt[aim] - recs[aim, 0:3].astype(np.float64)).astype(np.float32)
    recs[:, 6] = np.where(rng.random(n) < 0.7, np.inf, np.abs(recs[:, 6]))
    eh, eb = run_ref("tri", recs, 4)
    gh, go = ob.leaf_batch("tri", recs)
    assert (gh == eh).all()
    m = eh.astype(bool)
    assert (go.view(np.uint32)[m] == eb[m]).all()
    assert m.sum() > 1000


@pytest.mark.parametrize("seed", [21, 22])
def test_patch_oracle_equals_reference_on_random_inputs(seed):
    rng = np.random.default_rng(seed)
    n = 20000
    recs = specials(rng, (n, 19))
    p = recs[:, 7:].reshape(n, 4, 3).astype(np.float64)
    u, v = rng.random((n, 1)), rng.random((n, 1))
    tgt = (1 - u) * (1 - v) * p[:, 0] + u * (1 - v) * p[:, 1] + (1 - u) * v * p[:, 2] + u * v * p[:, 3]
    aim = rng.random(n) < 0.7
    recs[aim, 3:6] = (tgt[aim] - recs[aim, 0:3].astype(np.float64)).astype(np.float32)
    recs[:, 6] = np.where(rng.random(n) < 0.7, np.inf, np.abs(recs[:, 6]))
    eh, eb = run_ref("blp", recs, 3)
    gh, go = ob.leaf_batch("blp", recs)
    assert (gh == eh).all()
    m = eh.astype(bool)
    assert (go.view(np.uint32)[m] == eb[m]).all()
    assert m.sum() > 1000


@pytest.mark.parametrize("seed", [31, 32])
def test_slab_oracle_equals_reference_on_random_inputs(seed):
    rng = np.random.default_rng(seed)
    n = 40000
    recs = specials(rng, (n, 13))
    lo = np.minimum(recs[:, 7:10], recs[:, 10:13])
    hi = np.maximum(recs[:, 7:10], recs[:, 10:13])
    recs[:, 7:10], recs[:, 10:13] = lo, hi
    inside = rng.random(n) < 0.3
    recs[inside, 0:3] = (lo[inside] + (hi[inside] - lo[inside]) * rng.random((inside.sum(), 3))).astype(np.float32)
    onface = rng.random(n) < 0.1   # origin exactly on a face: the 0 * inf = NaN cases when d is 0 there
    recs[onface, 0] = lo[onface, 0]
    recs[:, 6] = np.where(rng.random(n) < 0.6, np.inf, np.abs(recs[:, 6]))
    eh, _ = run_ref("slab", recs, 0)
    gh, _ = ob.leaf_batch("slab", recs)
    assert (gh == eh).all()
    assert 0.05 < eh.mean() < 0.95


def random_affine(rng, n):
    """Rotation * non-uniform scale (some negative) + translation in float64; both the matrix
    and its float64 inverse are then rounded to float32, like a pbrt Transform's m / mInv pair."""
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)
    S = 10.0 ** rng.uniform(-2, 2, size=(n, 3)) * rng.choice([-1, 1], size=(n, 3), p=[0.1, 0.9])
    kind = rng.integers(0, 5, n)
    R[kind == 0] = np.eye(3)            # pure scale + translate
    S[kind == 1] = 1.0                  # rigid
    M = np.zeros((n, 4, 4))
    M[:, :3, :3] = R * S[:, None, :]
    M[:, :3, 3] = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-1, 3, size=(n, 1))
    M[kind == 2, :3, 3] = 0             # no translation
    M[:, 3, 3] = 1
    Mi = np.linalg.inv(M)
    return M.astype(np.float32), Mi.astype(np.float32)


@pytest.mark.parametrize("seed", [41, 42])
def test_ray_inverse_transform_equals_reference(seed):
    """Transform::ApplyInverse(Ray, tMax) with its interval arithmetic (transform.h:416-429)."""
    rng = np.random.default_rng(seed)
    n = 20000
    M, Mi = random_affine(rng, n)
    o = specials(rng, (n, 3))
    d = specials(rng, (n, 3))
    t = np.where(rng.random(n) < 0.5, np.inf, np.abs(specials(rng, n))).astype(np.float32)
    recs = np.concatenate([o, d, t[:, None], M.reshape(n, 16), Mi.reshape(n, 16)], 1)
    eh, eb = run_ref("xfray", recs, 7)
    got = ob.apply_inverse_ray(Mi.reshape(n, 16)[:, :12], o, d, t)
    assert (got.view(np.uint32) == eb).all()
