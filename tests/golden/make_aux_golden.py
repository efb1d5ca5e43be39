#!/usr/bin/env python3
"""Generate tests/golden/aux_{slab2,hash,offset,wrs}.npz from the REFERENCE's own compiled inline
functions (oracle/_ref/ref_leaf): Bounds3::IntersectP(o, d, tMax, &t0, &t1), Hash / HashFloat,
OffsetRayOrigin / SpawnRayTo and WeightedReservoirSampler.  Build container only; the committed
files hold seeded inputs and the reference's raw output bits."""
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_leaf_golden import OUT, REF  # noqa: E402

N = 4096


def specials(rng, shape):
    v = rng.normal(size=shape) * 10.0 ** rng.integers(-4, 4, size=shape)
    pick = rng.random(shape)
    v = np.where(pick < 0.05, 0.0, v)
    v = np.where((pick >= 0.05) & (pick < 0.07), -0.0, v)
    v = np.where((pick >= 0.07) & (pick < 0.09), np.round(v), v)
    return v.astype(np.float32)


def slab2_cases(rng):
    lo = rng.uniform(-5, 5, (N, 3))
    ext = rng.uniform(0, 4, (N, 3))
    ext[rng.random(N) < 0.1, rng.integers(0, 3)] = 0
    o = rng.uniform(-8, 8, (N, 3))
    inside = rng.random(N) < 0.2
    o[inside] = (lo + ext * rng.random((N, 3)))[inside]
    tgt = lo + ext * rng.uniform(-0.3, 1.3, (N, 3))
    d = tgt - o
    d[rng.random(N) < 0.1, rng.integers(0, 3)] = 0.0
    d[rng.random(N) < 0.03, rng.integers(0, 3)] = -0.0
    tmax = np.where(rng.random(N) < 0.4, np.inf, rng.uniform(0, 3, N))
    return np.concatenate([o, d, tmax[:, None], lo, lo + ext], 1).astype(np.float32)


def offset_cases(rng):
    p = specials(rng, (N, 3))
    err = np.abs(specials(rng, (N, 3))) * np.float32(1e-6)
    err[rng.random(N) < 0.15] = 0  # exact points (Interaction(Point3f p, ...))
    n = rng.normal(size=(N, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n[rng.random(N) < 0.1] = 0  # n = (0,0,0): the base interaction of IntersectOneRandom
    n[rng.random(N) < 0.1, rng.integers(0, 3)] = 0
    w = specials(rng, (N, 3))
    # Interval::FromValueAndError (math.h:829-838): an exact value is [v, v] (the sign of a zero included)
    lo = (p - err).astype(np.float32)
    hi = np.where(err == 0, lo, (p + err).astype(np.float32)).astype(np.float32)
    return np.concatenate([lo, hi, n, w], 1).astype(np.float32)


def wrs_cases(rng):
    pts = specials(rng, (N, 6))
    k = rng.integers(0, 9, N)
    k[:64] = np.arange(64) % 40
    return np.concatenate([pts, k[:, None]], 1).astype(np.float32)


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/ref_leaf missing: run `make -C oracle ref` in the build container")
    rng = np.random.default_rng(20241009)
    for mode, gen, nout in (("slab2", slab2_cases, 2), ("hash", lambda r: specials(r, (N, 6)), 2),
                            ("offset", offset_cases, 9), ("wrs", wrs_cases, 2)):
        recs = gen(rng)
        with tempfile.TemporaryDirectory() as td:  # first word is a full int32 here, not a flag
            fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
            with open(fi, "wb") as f:
                f.write(np.int32(len(recs)).tobytes())
                f.write(recs.tobytes())
            subprocess.run([REF, mode, fi, fo], check=True)
            raw = np.fromfile(fo, dtype=np.uint32).reshape(len(recs), 1 + nout)
        np.savez_compressed(os.path.join(OUT, f"aux_{mode}.npz"), inputs=recs, first=raw[:, 0].copy(),
                            out_bits=raw[:, 1:].copy())
        print(f"{mode}: {len(recs)} cases")


if __name__ == "__main__":
    main()
