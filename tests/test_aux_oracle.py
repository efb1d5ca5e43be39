"""The oracle's restatements of the small reference functions the wavefront entry points and the
alpha test lean on — Bounds3::IntersectP(o, d, tMax, &t0, &t1) (util/vecmath.h:1547-1571), Hash /
HashFloat (util/hash.h), OffsetRayOrigin / SpawnRayTo (ray.h:75-101), WeightedReservoirSampler on
the PCG32 RNG (util/sampling.h:524-596, util/rng.h) — against golden vectors produced by the
REFERENCE's own compiled code (tests/golden/make_aux_golden.py via oracle/_ref/ref_leaf), bit for
bit, plus live random cases where the binary is present."""
import os

import numpy as np
import pytest

import oracle_binding as ob

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "ref_leaf")


def load(name):
    g = np.load(os.path.join(GOLD, f"aux_{name}.npz"))
    return g["inputs"], g["first"], g["out_bits"]


def test_root_interval_golden():
    x, first, bits = load("slab2")
    hit, tt = ob.bounds_t0t1(x[:, 7:13], x[:, 0:3], x[:, 3:6], x[:, 6])
    assert np.array_equal(hit, first.astype(np.uint8))
    m = hit.astype(bool)
    assert 0.2 < m.mean() < 0.9
    assert np.array_equal(tt[m].view(np.uint32), bits[m])


def test_hash_golden():
    x, first, bits = load("hash")
    lo, hi, hf = ob.hash_batch(x)
    assert np.array_equal(lo, first) and np.array_equal(hi, bits[:, 1])
    assert np.array_equal(hf.view(np.uint32), bits[:, 0])
    assert len(np.unique(lo)) > 4000 and 0.4 < hf.mean() < 0.6


def test_offset_ray_origin_and_spawn_ray_to_golden():
    x, _, bits = load("offset")
    out = ob.offset_batch(x)
    assert np.array_equal(out.view(np.uint32), bits)
    moved = (out[:, :3] != (x[:, 0:3] + x[:, 3:6]) / 2).any(1)
    assert 0.5 < moved.mean() < 1.0


def test_weighted_reservoir_sampler_golden():
    x, first, bits = load("wrs")
    sel, out = ob.wrs_batch(x)
    assert np.array_equal(sel, first.view(np.int32))
    assert np.array_equal(out.view(np.uint32), bits)
    k = x[:, 6].astype(int)
    assert (sel[k == 0] == -1).all() and (sel[k > 0] >= 0).all() and (sel < np.maximum(k, 1)).all()
    # every candidate gets picked sometimes
    assert len(np.unique(sel[k == 8])) == 8


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/ref_leaf not built")
@pytest.mark.parametrize("seed", [1, 2])
def test_live_against_the_reference_binary(seed):
    from test_oracle_vs_reference_live import specials
    rng = np.random.default_rng(seed)
    n = 20000
    x = specials(rng, (n, 6))
    first, bits = run_ref_raw("hash", x, 2)
    lo, hi, hf = ob.hash_batch(x)
    assert np.array_equal(lo, first) and np.array_equal(hi, bits[:, 1]) and np.array_equal(hf.view(np.uint32), bits[:, 0])
    p = specials(rng, (n, 3))
    err = np.abs(specials(rng, (n, 3))) * np.float32(1e-6)
    nn = rng.normal(size=(n, 3)).astype(np.float32)
    nn[rng.random(n) < 0.1] = 0
    lo = (p - err).astype(np.float32)
    y = np.concatenate([lo, np.where(err == 0, lo, p + err), nn, specials(rng, (n, 3))], 1).astype(np.float32)
    y = y[np.isfinite(y).all(1)]
    _, bits = run_ref_raw("offset", y, 9)
    assert np.array_equal(ob.offset_batch(y).view(np.uint32), bits)
    z = np.concatenate([specials(rng, (n, 6)), rng.integers(0, 30, (n, 1))], 1).astype(np.float32)
    first, bits = run_ref_raw("wrs", z, 2)
    sel, out = ob.wrs_batch(z)
    assert np.array_equal(sel, first.view(np.int32)) and np.array_equal(out.view(np.uint32), bits)
    b = np.sort(rng.uniform(-5, 5, (n, 2, 3)), axis=1).reshape(n, 6)
    o = rng.uniform(-8, 8, (n, 3))
    d = (b[:, :3] + rng.uniform(-2, 6, (n, 3)) - o)
    d[::7, 1] = 0
    w = np.concatenate([o, d, np.where(rng.random(n) < 0.5, np.inf, rng.uniform(0, 4, n))[:, None], b], 1).astype(np.float32)
    first, bits = run_ref_raw("slab2", w, 2)
    hit, tt = ob.bounds_t0t1(w[:, 7:13], w[:, 0:3], w[:, 3:6], w[:, 6])
    assert np.array_equal(hit, first.astype(np.uint8))
    assert np.array_equal(tt[hit.astype(bool)].view(np.uint32), bits[hit.astype(bool)])


def run_ref_raw(mode, recs, nout):
    import subprocess
    import tempfile
    recs = np.ascontiguousarray(recs, np.float32)
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "i.bin"), os.path.join(td, "o.bin")
        with open(fi, "wb") as f:
            f.write(np.int32(len(recs)).tobytes())
            f.write(recs.tobytes())
        subprocess.run([REF, mode, fi, fo], check=True)
        raw = np.fromfile(fo, np.uint32).reshape(len(recs), 1 + nout)
    return raw[:, 0].copy(), raw[:, 1:].copy()
