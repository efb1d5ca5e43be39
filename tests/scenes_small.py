"""Small seeded scenes shared by the parity tests (inputs only)."""
import numpy as np

from nn_bvh_amd import make_prims


def random_soup(n_tris=2000, n_patches=0, seed=0, extent=10.0, size=0.6):
    """Random triangle (+ bilinear patch) soup in a box."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, size=(n_tris, 1, 3))
    tv = (c + rng.uniform(-size, size, size=(n_tris, 3, 3))).reshape(-1, 3)
    tri = np.arange(3 * n_tris, dtype=np.int32).reshape(n_tris, 3)
    verts = [tv]
    patch = None
    if n_patches:
        c = rng.uniform(-extent, extent, size=(n_patches, 1, 3))
        eu = rng.uniform(-size, size, size=(n_patches, 1, 3))
        ev = rng.uniform(-size, size, size=(n_patches, 1, 3))
        tw = rng.uniform(-0.3 * size, 0.3 * size, size=(n_patches, 1, 3))
        pv = np.concatenate([c, c + eu, c + ev, c + eu + ev + tw], 1).reshape(-1, 3)
        patch = (np.arange(4 * n_patches, dtype=np.int32) + 3 * n_tris).reshape(n_patches, 4)
        verts.append(pv)
    verts = np.concatenate(verts).astype(np.float32)
    return verts, make_prims(tri, patch)


def grid_mesh(n=48, seed=0, bump=0.4):
    """Connected height-field mesh: shared vertices/edges, so rays graze edges and hit
    shared vertices (the watertightness / tie-breaking cases)."""
    rng = np.random.default_rng(seed)
    x, z = np.meshgrid(np.linspace(-5, 5, n + 1), np.linspace(-5, 5, n + 1), indexing="ij")
    y = bump * rng.standard_normal(x.shape)
    verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a = (i * (n + 1) + j).ravel()
    tri = np.stack([a, a + n + 1, a + n + 2, a, a + n + 2, a + 1], 1).reshape(-1, 3)
    return verts, make_prims(tri.astype(np.int32))


def coincident_centroids(n=200, seed=0):
    """Many primitives with the same centroid -> one big leaf (> maxnodeprims), the
    coffee_maker 64-prim-leaf situation (aggregates.cpp:225-233)."""
    rng = np.random.default_rng(seed)
    # the builder's centroid is the BOUNDS centroid (aggregates.cpp:92): give every triangle the
    # box [c - h, c + h] (h a multiple of 1/64, so .5*min + .5*max == c exactly)
    h = (rng.integers(1, 64, size=(n, 3)) / 64.0).astype(np.float32)
    w = (rng.integers(-63, 64, size=(n, 3)) / 64.0).astype(np.float32) * h
    c = np.array([1.0, 2.0, 3.0], np.float32)
    verts = np.stack([c - h, c + h, c + w], 1).reshape(-1, 3).astype(np.float32)
    tri = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    return verts, make_prims(tri)


def edge_case_rays(verts, prims, seed=0, n=2048):
    """Rays built to sit on the comparison boundaries: axis-aligned (two zero components),
    one zero component, aimed exactly at shared vertices / edge midpoints, origins inside the
    scene box, shadow-style un-normalised d with tMax = 1 - 1e-4, tiny and huge tMax."""
    from nn_bvh_amd import make_rays
    rng = np.random.default_rng(seed)
    lo, hi = verts.min(0), verts.max(0)
    ext = hi - lo
    k = n // 8
    parts = []
    # axis-aligned through random vertices
    tgt = verts[rng.integers(0, len(verts), k)]
    ax = rng.integers(0, 3, k)
    d = np.zeros((k, 3), np.float32)
    d[np.arange(k), ax] = rng.choice([-1.0, 1.0], k)
    parts.append(make_rays(tgt - d * (ext.max() * 2), d))
    # one zero component
    o = lo + rng.random((k, 3)) * ext + ext * np.array([0, 2, 0])
    d = rng.normal(size=(k, 3)).astype(np.float32)
    d[np.arange(k), rng.integers(0, 3, k)] = 0
    d[:, 1] = -np.abs(d[:, 1]) - 0.1
    parts.append(make_rays(o, d))
    # exactly at vertices from outside
    o = (lo + rng.random((k, 3)) * ext + ext * 1.5).astype(np.float32)
    tgt = verts[rng.integers(0, len(verts), k)]
    parts.append(make_rays(o, tgt - o))
    # at edge midpoints of primitives
    p = prims[rng.integers(0, len(prims), k)]
    mid = (verts[p["v"][:, 0]] + verts[p["v"][:, 1]]) * np.float32(0.5)
    parts.append(make_rays(o, mid - o))
    # origins inside the box, random directions
    o2 = (lo + rng.random((k, 3)) * ext).astype(np.float32)
    parts.append(make_rays(o2, rng.normal(size=(k, 3))))
    # shadow style
    a = (lo + rng.random((k, 3)) * ext).astype(np.float32)
    b = (lo + rng.random((k, 3)) * ext).astype(np.float32)
    parts.append(make_rays(a, b - a, tmax=np.float32(1 - 1e-4)))
    # short tMax
    parts.append(make_rays(o2, rng.normal(size=(k, 3)), tmax=rng.uniform(0.01, 2.0, k)))
    # negative zero and denormal direction components
    d = rng.normal(size=(k, 3)).astype(np.float32)
    d[::2, 0] = -0.0
    d[1::2, 2] = 1e-42
    parts.append(make_rays(o2, d))
    return np.concatenate(parts)
