#!/usr/bin/env python3
"""Generate tests/golden/leaf_{tri,blp,slab}.npz from the REFERENCE's own compiled leaf
functions (oracle/_ref/ref_leaf, built by oracle/Makefile from /root/reference sources).

Runs only in the build container (needs oracle/_ref/ref_leaf).  The committed .npz files
hold data only: seeded inputs + the reference's outputs (hit flags and raw float32 bits).

Case mix (per kind), chosen to reach every branch of the reference functions:
  tri : rays aimed at interior points, at edges and at vertices (exact-zero edge functions
        -> the double-precision fallback, shapes.cpp:215-225), axis-aligned directions,
        un-normalised shadow-style directions with tMax = 1 - 1e-4, tMax just below / above
        the hit, degenerate triangles, coordinates scaled 10^U(-3,3), rays leaving from the
        surface (t <= deltaT rejection), and the Triangle.BadCases known-answer case
        (shapes_test.cpp:435-449, must miss).
  blp : planar and twisted patches, rays at interior (u,v) points, a == 0 (parallelogram)
        quadratics, two-root cases, misses, tMax cuts.
  slab: random boxes, origins inside/outside, zero direction components (invDir = +-inf and
        the NaN comparisons that follow), negative zero, tMax cuts, degenerate (flat) boxes.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_leaf")
OUT = os.path.join(ROOT, "tests", "golden")
N = 8192


def run_ref(mode, recs, nout):
    recs = np.ascontiguousarray(recs, dtype=np.float32)
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fi, "wb") as f:
            f.write(np.int32(len(recs)).tobytes())
            f.write(recs.tobytes())
        subprocess.run([REF, mode, fi, fo], check=True)
        raw = np.fromfile(fo, dtype=np.uint32).reshape(len(recs), 1 + nout)
    hit = raw[:, 0].astype(np.uint8)
    vals = raw[:, 1:].copy()  # float32 bit patterns
    return hit, vals


def unit(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def tri_cases(rng):
    n = N
    scale = (10.0 ** rng.uniform(-3, 3, size=(n, 1))).astype(np.float32)
    p = (rng.uniform(-1, 1, size=(n, 3, 3)) * scale[:, None, :]).astype(np.float32)
    # target point: interior / edge / vertex
    bary = rng.dirichlet([1, 1, 1], size=n)
    kind = rng.integers(0, 10, size=n)
    e = kind == 1  # on an edge: one barycentric exactly 0
    bary[e, rng.integers(0, 3)] = 0
    bary[e] /= bary[e].sum(axis=1, keepdims=True)
    v = kind == 2  # through a vertex
    bary[v] = np.eye(3)[rng.integers(0, 3, size=v.sum())]
    tgt = np.einsum("nk,nkc->nc", bary, p.astype(np.float64))
    d = unit(rng, n)
    dist = (10.0 ** rng.uniform(-2, 2, size=(n, 1))) * scale
    o = tgt - d * dist
    d = d * (10.0 ** rng.uniform(-1, 1, size=(n, 1)))  # un-normalised
    tmax = np.full(n, np.inf)
    # misses: aim away from the triangle plane point
    m = kind == 3
    o[m] += unit(rng, m.sum()) * scale[m] * 2
    # axis-aligned directions (two zero components)
    a = kind == 4
    ax = rng.integers(0, 3, size=a.sum())
    da = np.zeros((a.sum(), 3))
    da[np.arange(a.sum()), ax] = rng.choice([-1.0, 1.0], size=a.sum())
    d[a] = da
    o[a] = tgt[a] - da * dist[a]
    # shadow-ray style: d = target - o exactly in float32, tMax = 1 - 1e-4 (hit lies beyond)
    s = kind == 5
    o32 = o.astype(np.float32)
    d[s] = (tgt[s].astype(np.float32) - o32[s]).astype(np.float64)
    tmax[s] = np.float32(1 - 1e-4)
    # tMax cuts around the true distance
    c = kind == 6
    tmax[c] = (dist[c, 0] / np.linalg.norm(d[c], axis=1)) * rng.choice(
        [0.5, 0.999999, 1.0, 1.000001, 2.0], size=c.sum())
    # rays starting on the surface (t ~ 0): exercise the deltaT rejection
    z = kind == 7
    o[z] = tgt[z]
    # degenerate triangles
    g = kind == 8
    p[g, 2] = p[g, 1]
    recs = np.concatenate([o, d, tmax[:, None], p.reshape(n, 9)], axis=1).astype(np.float32)
    # exact vertex/edge hits built in float32: ray from integer-ish coords so e_i == 0 exactly
    k = 256
    q = rng.integers(-8, 9, size=(k, 3, 3)).astype(np.float32)
    oo = rng.integers(-20, 21, size=(k, 3)).astype(np.float32)
    w = rng.integers(0, 3, size=k)
    tt = q[np.arange(k), w]  # aim exactly at a vertex
    half = k // 2
    tt[half:] = (q[half:, 0] + q[half:, 1]) * np.float32(0.5)  # edge midpoint (exact)
    dd = tt - oo
    recs[:k] = np.concatenate([oo, dd, np.full((k, 1), np.inf, np.float32), q.reshape(k, 9)], 1)
    # Triangle.BadCases KAT (shapes_test.cpp:435-449): must miss
    recs[k] = np.array([-1081.47925, 99.9999542, 87.7701111, -32.1072998, -183.355865,
                        -144.607635, np.inf, -1113.45459, -79.049614, -56.2431908, -1113.45459,
                        -87.0922699, -56.2431908, -1113.45459, -79.2090149, -56.2431908],
                       np.float32)
    return recs


def blp_cases(rng):
    n = N
    scale = (10.0 ** rng.uniform(-2, 2, size=(n, 1))).astype(np.float32)
    base = rng.uniform(-1, 1, size=(n, 3))
    eu = rng.uniform(-1, 1, size=(n, 3))
    ev = rng.uniform(-1, 1, size=(n, 3))
    tw = rng.uniform(-0.5, 0.5, size=(n, 3))
    kind = rng.integers(0, 8, size=n)
    tw[kind == 1] = 0  # parallelogram: planar, a == 0 up to rounding
    p00 = base
    p10 = base + eu
    p01 = base + ev
    p11 = base + eu + ev + tw
    uv = rng.uniform(0, 1, size=(n, 2))
    u, v = uv[:, :1], uv[:, 1:]
    tgt = (1 - u) * (1 - v) * p00 + u * (1 - v) * p10 + (1 - u) * v * p01 + u * v * p11
    d = unit(rng, n)
    dist = 10.0 ** rng.uniform(-1, 1, size=(n, 1))
    o = tgt - d * dist
    d = d * (10.0 ** rng.uniform(-1, 1, size=(n, 1)))
    tmax = np.full(n, np.inf)
    m = kind == 2
    o[m] += unit(rng, m.sum()) * 3
    c = kind == 3
    tmax[c] = (dist[c, 0] / np.linalg.norm(d[c], axis=1)) * rng.choice(
        [0.5, 0.999999, 1.000001, 2.0], size=c.sum())
    a = kind == 4  # exact integer parallelograms, axis-aligned rays: a == 0 exactly
    k = a.sum()
    p00[a] = rng.integers(-4, 5, size=(k, 3))
    euq = rng.integers(-4, 5, size=(k, 3))
    evq = rng.integers(-4, 5, size=(k, 3))
    p10[a] = p00[a] + euq
    p01[a] = p00[a] + evq
    p11[a] = p00[a] + euq + evq
    ax = rng.integers(0, 3, size=k)
    da = np.zeros((k, 3))
    da[np.arange(k), ax] = 1.0
    ctr = p00[a] + 0.5 * euq + 0.25 * evq
    d[a] = da
    o[a] = ctr - 7.0 * da
    s = scale.astype(np.float64)
    s[a] = 1.0
    recs = np.concatenate([o * s, d, tmax[:, None] * np.where(np.isinf(tmax[:, None]), 1, s),
                           p00 * s, p10 * s, p01 * s, p11 * s], axis=1).astype(np.float32)
    return recs


def slab_cases(rng):
    n = N
    c = rng.uniform(-10, 10, size=(n, 3))
    h = 10.0 ** rng.uniform(-3, 1, size=(n, 3))
    kind = rng.integers(0, 8, size=n)
    h[kind == 1, rng.integers(0, 3)] = 0  # flat box
    pmin, pmax = c - h, c + h
    tgt = c + rng.uniform(-1.3, 1.3, size=(n, 3)) * h
    d = unit(rng, n)
    dist = 10.0 ** rng.uniform(-2, 1.5, size=(n, 1))
    o = tgt - d * dist
    ins = kind == 2
    o[ins] = c[ins] + rng.uniform(-0.9, 0.9, size=(ins.sum(), 3)) * h[ins]
    z = kind == 3  # one or two exactly-zero direction components
    zi = np.where(z)[0]
    d[zi, rng.integers(0, 3, size=len(zi))] = 0.0
    d[zi[::2], rng.integers(0, 3, size=len(zi[::2]))] = 0.0
    nz = kind == 4  # negative zero
    d[np.where(nz)[0], rng.integers(0, 3, size=nz.sum())] = -0.0
    onb = kind == 5  # origin exactly on a face with a zero component -> 0 * inf = NaN
    oi = np.where(onb)[0]
    axx = rng.integers(0, 3, size=len(oi))
    d[oi, axx] = 0.0
    o[oi, axx] = np.where(rng.random(len(oi)) < 0.5, pmin[oi, axx], pmax[oi, axx])
    tmax = np.full(n, np.inf)
    ct = kind == 6
    tmax[ct] = dist[ct, 0] * rng.choice([0.25, 0.9, 1.0, 1.1, 4.0], size=ct.sum())
    d = d * (10.0 ** rng.uniform(-1, 1, size=(n, 1)))
    recs = np.concatenate([o, d, tmax[:, None], pmin, pmax], axis=1).astype(np.float32)
    # re-impose exact face placement after the float32 rounding
    recs[oi, axx] = np.where(recs[oi, axx] < c[oi, axx], recs[oi, 7 + axx], recs[oi, 10 + axx])
    return recs


def xfray_cases(rng):
    """Transform::ApplyInverse(Ray, tMax) inputs: o[3] d[3] tmax m[16] mInv[16] — rigid, scaled
    (also mirrored), translation-free and pure-scale transforms; origins near and far."""
    n = N
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)
    S = 10.0 ** rng.uniform(-2, 2, size=(n, 3)) * rng.choice([-1, 1], size=(n, 3), p=[0.1, 0.9])
    kind = rng.integers(0, 5, n)
    R[kind == 0] = np.eye(3)
    S[kind == 1] = 1.0
    M = np.zeros((n, 4, 4))
    M[:, :3, :3] = R * S[:, None, :]
    M[:, :3, 3] = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-1, 3, size=(n, 1))
    M[kind == 2, :3, 3] = 0
    M[:, 3, 3] = 1
    Mi = np.linalg.inv(M)
    o = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 4, size=(n, 1))
    d = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-2, 2, size=(n, 1))
    d[kind == 3, rng.integers(0, 3)] = 0.0
    o[kind == 4] = 0.0
    tmax = np.where(rng.random(n) < 0.5, np.inf, 10.0 ** rng.uniform(-3, 3, n))
    return np.concatenate([o, d, tmax[:, None], M.reshape(n, 16), Mi.reshape(n, 16)], 1).astype(np.float32)


def xfbounds_cases(rng):
    """Transform::operator()(Bounds3f) inputs: m[16] pmin[3] pmax[3] — the instance transforms of
    xfray_cases applied to boxes of very different extents, incl. flat and point-like ones."""
    n = 2048
    m = xfray_cases(rng)[:n, 7:23]
    lo = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-2, 3, size=(n, 1))
    ext = 10.0 ** rng.uniform(-3, 3, size=(n, 3))
    kind = rng.integers(0, 6, n)
    ext[kind == 0, rng.integers(0, 3)] = 0.0  # flat box
    ext[kind == 1] = 0.0                      # a point
    return np.concatenate([m, lo, lo + ext], 1).astype(np.float32)


def morton_scene(rng):
    """A triangle soup (partly clustered, partly coincident) + what buildHLBVH feeds EncodeMorton3:
    the bounds of the primitives' centroids and each centroid, in float32 as the builder computes them
    (BVHPrimitive::Centroid() = .5f * pMin + .5f * pMax, cpu/aggregates.h)."""
    n = 6000
    c = rng.uniform(-10, 10, size=(n, 1, 3))
    c[4000:5000] = rng.uniform(-0.01, 0.01, size=(1000, 1, 3)) + 3.0   # a tight cluster
    c[5000:5200] = c[5000]                                            # coincident centroids
    v = (c + rng.uniform(-0.6, 0.6, size=(n, 3, 3))).astype(np.float32)
    v[5000:5200] = v[5000]
    mn, mx = v.min(1), v.max(1)
    cen = (np.float32(0.5) * mn + np.float32(0.5) * mx).astype(np.float32)
    cb = np.concatenate([cen.min(0), cen.max(0)]).astype(np.float32)
    rec = np.concatenate([np.tile(cb, (n, 1)), cen], 1).astype(np.float32)
    return v.reshape(-1, 3), rec


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/ref_leaf missing: run `make -C oracle ref` in the build container")
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20241008)
    for mode, gen, nout in (("tri", tri_cases, 4), ("blp", blp_cases, 3), ("slab", slab_cases, 0),
                            ("xfray", xfray_cases, 7), ("xfbounds", xfbounds_cases, 6)):
        recs = gen(rng)
        hit, bits = run_ref(mode, recs, nout)
        np.savez_compressed(os.path.join(OUT, f"leaf_{mode}.npz"), inputs=recs, hit=hit,
                            out_bits=bits)
        print(f"{mode}: {len(recs)} cases, {int(hit.sum())} hits")
    verts, rec = morton_scene(rng)
    codes, _ = run_ref("morton", rec, 0)
    raw = None
    with tempfile.TemporaryDirectory() as td:  # codes are full int32 values, not flags: re-read them as such
        fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fi, "wb") as f:
            f.write(np.int32(len(rec)).tobytes())
            f.write(rec.tobytes())
        subprocess.run([REF, "morton", fi, fo], check=True)
        raw = np.fromfile(fo, dtype=np.uint32)
    np.savez_compressed(os.path.join(OUT, "hlbvh_morton.npz"), verts=verts, inputs=rec, codes=raw)
    print(f"morton: {len(rec)} codes, {len(np.unique(raw))} distinct")


if __name__ == "__main__":
    main()
