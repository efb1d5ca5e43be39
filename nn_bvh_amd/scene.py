"""Scene inputs for tests and bench: geometry blobs, a procedural stand-in, and seeded
ray-batch generators (primary / diffuse bounce / shadow) shaped like the batches pbrt's
integrators hand to BVHAggregate.  numpy only; no traversal arithmetic.

Geometry sources
  * data/<scene>.npz  — flat float32 vertices + int32 triangle indices extracted from the
    reference's scene PLY files by tools/make_scene_blobs.py in the build container
    (git-ignored; travels to the GPU box with the snapshot).  World space, identity object
    transform (crown/bathroom/coffee_maker carry only a camera transform).
  * procedural_scene() — seeded fallback of the same scale when a blob is absent.
"""
import os

import numpy as np

from ._lib import RAY_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "data")

# world-space cameras of the reference scenes:
# eye, look-at, up, fov (degrees, on the shorter image axis), xres, yres
CAMERAS = {
    "crown": ((0, 5.5, 24), (0, 11, -10), (0, 1, 0), 47.0, 1000, 1400),
    "bathroom": ((0.00724, 0.91241, -0.22758), (-0.74464, 0.92788, -0.88670), (0, 1, 0), 55.0,
                 1024, 1024),
    "coffee_maker": ((-0.00296, 0.19830, 0.82815), (0.00431, 0.15898, -0.17105), (0, 1, 0), 25.0,
                     800, 1000),
    "killeroos": ((400, 20, 30), (0, 63, -110), (0, 0, 1), 39.0, 700, 700),
}


# The six area-light quads of the crown scene (scene description data: the "point3 P" lists of
# /root/reference/scenes/crown/crown.pbrt:26-102; corners in the file's order).  Shadow rays of
# the crown workload aim at uniformly sampled points on them.
CROWN_LIGHT_QUADS = np.array([
    [[5.702774, -13.539273, -76.185936], [-14.945709, -13.791711, -75.26613],
     [-14.886804, 13.04458, -66.57877], [5.761684, 13.297014, -67.49859]],
    [[41.6316, 14.548275, -16.682684], [45.396217, 14.548271, -2.696177],
     [40.693626, 28.926506, -1.430422], [36.92901, 28.926506, -15.416933]],
    [[-45.1155, 14.650443, -2.989656], [-41.24645, 14.65044, -16.947636],
     [-36.506706, 29.012154, -15.633818], [-40.37576, 29.012156, -1.675834]],
    [[20.38967, -4.332172, 17.202255], [13.206003, -4.332175, 23.926588],
     [15.053431, 5.620016, 25.900215], [22.237099, 5.620017, 19.17588]],
    [[-18.349361, -3.757668, 23.42476], [-24.791203, -3.75767, 15.977483],
     [-26.581875, 6.287214, 17.5264], [-20.140032, 6.287215, 24.973679]],
    [[9.260806, 19.171276, 24.593668], [-9.12361, 19.171274, 25.294205],
     [-9.666298, 32.1588, 11.052294], [8.718122, 32.1588, 10.351759]],
], np.float64)


# ---------------------------------------------------------------------------------------
def read_ply(path):
    """Binary-little-endian PLY with float x,y,z(+extras) vertices and uint8-counted int
    face lists (what the reference's scenes use; util/mesh.cpp:322 reads them via rply).
    Quads are split (0,1,2),(0,2,3) like TriQuadMesh::ConvertToOnlyTriangles."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    header = data[:end].decode("ascii").split("\n")
    if "format binary_little_endian 1.0" not in header:
        raise ValueError(f"{path}: only binary_little_endian PLY is supported")
    elems, cur = [], None
    for line in header:
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "element":
            cur = {"name": tok[1], "count": int(tok[2]), "props": []}
            elems.append(cur)
        elif tok[0] == "property":
            cur["props"].append(tok[1:])
    tmap = {"float": "<f4", "float32": "<f4", "double": "<f8", "int": "<i4", "int32": "<i4",
            "uint": "<u4", "uint32": "<u4", "uchar": "u1", "uint8": "u1", "char": "i1",
            "short": "<i2", "ushort": "<u2"}
    off = end
    verts = faces = None
    for e in elems:
        if e["name"] == "vertex":
            dt = np.dtype([(p[1], tmap[p[0]]) for p in e["props"]])
            arr = np.frombuffer(data, dt, e["count"], off)
            off += dt.itemsize * e["count"]
            verts = np.stack([arr["x"], arr["y"], arr["z"]], 1).astype(np.float32)
        elif e["name"] == "face":
            lists = [p for p in e["props"] if p[0] == "list"]
            if len(lists) != 1 or len(e["props"]) != 1:
                # general (slow) path: face_indices etc. alongside the list
                tris = []
                for _ in range(e["count"]):
                    idx = None
                    for p in e["props"]:
                        if p[0] == "list":
                            cdt, idt = np.dtype(tmap[p[1]]), np.dtype(tmap[p[2]])
                            c = int(np.frombuffer(data, cdt, 1, off)[0])
                            off += cdt.itemsize
                            vals = np.frombuffer(data, idt, c, off)
                            off += idt.itemsize * c
                            if p[3] == "vertex_indices":
                                idx = vals
                        else:
                            off += np.dtype(tmap[p[0]]).itemsize
                    tris.append(idx[[0, 1, 2]])
                    if len(idx) == 4:
                        tris.append(idx[[0, 2, 3]])
                faces = np.asarray(tris, np.int32).reshape(-1, 3)
            else:
                p = lists[0]
                cdt, idt = np.dtype(tmap[p[1]]), np.dtype(tmap[p[2]])
                n = e["count"]
                rec3 = np.dtype([("c", cdt), ("v", idt, 3)])
                fast = False
                if off + rec3.itemsize * n <= len(data):
                    arr = np.frombuffer(data, rec3, n, off)
                    fast = bool((arr["c"] == 3).all())
                if fast:
                    faces = arr["v"].astype(np.int32)
                    off += rec3.itemsize * n
                else:
                    tris = []
                    for _ in range(n):
                        c = int(np.frombuffer(data, cdt, 1, off)[0])
                        off += cdt.itemsize
                        idx = np.frombuffer(data, idt, c, off)
                        off += idt.itemsize * c
                        tris.append(idx[[0, 1, 2]])
                        if c == 4:
                            tris.append(idx[[0, 2, 3]])
                    faces = np.asarray(tris, np.int32).reshape(-1, 3)
        else:
            raise ValueError(f"{path}: unexpected element {e['name']}")
    return verts, faces


def blob_path(name):
    return os.path.join(DATA, f"{name}.npz")


def save_blob(name, verts, tris):
    os.makedirs(DATA, exist_ok=True)
    np.savez_compressed(blob_path(name), verts=np.asarray(verts, np.float32),
                        tris=np.asarray(tris, np.int32))


def load_blob(name):
    with np.load(blob_path(name)) as z:
        return z["verts"], z["tris"]


def procedural_scene(n_tris=3_540_000, seed=7):
    """Seeded stand-in of crown scale: a field of tessellated, bumpy ellipsoids ("gems") on a
    ground grid, inside the crown camera's view.  Used only when data/crown.npz is absent."""
    rng = np.random.default_rng(seed)
    verts, tris = [], []
    base = 0
    n_obj = max(8, n_tris // 20000)
    per = n_tris // n_obj
    res_v = max(4, int(np.sqrt(per / 2)))
    res_u = max(4, per // (2 * res_v))
    u = np.linspace(0, 2 * np.pi, res_u + 1)[:-1]
    v = np.linspace(0.02, np.pi - 0.02, res_v + 1)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    iu, iv = np.meshgrid(np.arange(res_u), np.arange(res_v), indexing="ij")
    a = (iu * (res_v + 1) + iv).ravel()
    b = (((iu + 1) % res_u) * (res_v + 1) + iv).ravel()
    quad = np.stack([a, b, b + 1, a, b + 1, a + 1], 1).reshape(-1, 3)
    for _ in range(n_obj):
        c = np.array([rng.uniform(-9, 9), rng.uniform(0.5, 16), rng.uniform(-9, 9)])
        rad = rng.uniform(0.15, 1.6, size=3)
        bump = 1 + 0.08 * np.sin(rng.uniform(3, 17) * uu + rng.uniform(0, 6)) * np.sin(
            rng.uniform(3, 17) * vv)
        p = np.stack([rad[0] * bump * np.cos(uu) * np.sin(vv), rad[1] * bump * np.cos(vv),
                      rad[2] * bump * np.sin(uu) * np.sin(vv)], -1).reshape(-1, 3) + c
        verts.append(p)
        tris.append(quad + base)
        base += len(p)
    g = 64
    gx, gz = np.meshgrid(np.linspace(-30, 30, g + 1), np.linspace(-30, 30, g + 1), indexing="ij")
    verts.append(np.stack([gx, np.zeros_like(gx), gz], -1).reshape(-1, 3))
    gi, gj = np.meshgrid(np.arange(g), np.arange(g), indexing="ij")
    a = (gi * (g + 1) + gj).ravel() + base
    tris.append(np.stack([a, a + g + 1, a + g + 2, a, a + g + 2, a + 1], 1).reshape(-1, 3))
    return np.concatenate(verts).astype(np.float32), np.concatenate(tris).astype(np.int32)


def load_scene(name):
    """(verts, tris, source) — blob if present, else the procedural stand-in."""
    if os.path.exists(blob_path(name)):
        v, t = load_blob(name)
        return v, t, f"{name} (reference PLY geometry, data/{name}.npz)"
    sizes = {"crown": 3_540_000, "bathroom": 592_000, "coffee_maker": 235_000, "killeroos": 66_500}
    v, t = procedural_scene(sizes.get(name, 100_000), seed=7)
    return v, t, f"procedural stand-in for {name} ({len(t)} tris; data/{name}.npz absent)"


# ---------------------------------------------------------------------------------------
_MORTON_CACHE = {}


def _morton_order(xres, yres):
    if (xres, yres) not in _MORTON_CACHE:
        _MORTON_CACHE[(xres, yres)] = _morton_order_uncached(xres, yres)
    return _MORTON_CACHE[(xres, yres)]


def _morton_order_uncached(xres, yres):
    x, y = np.meshgrid(np.arange(xres, dtype=np.uint32), np.arange(yres, dtype=np.uint32),
                       indexing="xy")
    x, y = x.ravel(), y.ravel()

    def part(v):
        v = v.astype(np.uint64)
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v

    order = np.argsort(part(x) | (part(y) << 1), kind="stable")
    return x[order].astype(np.float32), y[order].astype(np.float32)


def camera_rays(name_or_cam, seed=1, sample=0, jitter=True, subsample=1, return_pixels=False,
                subset=None):
    """Pinhole primary rays, one per pixel, in Morton (tile-coherent) pixel order.
    sample selects the jitter stream; subsample>1 keeps every k-th pixel in each axis.
    With return_pixels also returns the integer pixel coordinates (px, py) of every ray.
    subset: indices into the Morton-ordered pixel list; only those rays are built (the jitter
    of a pixel does not depend on the subset)."""
    cam = CAMERAS[name_or_cam] if isinstance(name_or_cam, str) else name_or_cam
    eye, look, up, fov, xres, yres = cam
    eye, look, up = (np.asarray(a, np.float64) for a in (eye, look, up))
    px, py = _morton_order(xres // subsample, yres // subsample)
    px, py = px * subsample, py * subsample
    rng = np.random.default_rng([seed, sample])
    if jitter:
        jx, jy = rng.random(len(px)), rng.random(len(px))
    else:
        jx = jy = np.full(len(px), 0.5)
    if subset is not None:
        px, py, jx, jy = px[subset], py[subset], jx[subset], jy[subset]
    w = look - eye
    w /= np.linalg.norm(w)
    right = np.cross(w, up)  # pbrt LookAt is left-handed: right = up x dir, mirrored images are fine here
    right /= np.linalg.norm(right)
    upv = np.cross(right, w)
    half = np.tan(np.radians(fov) / 2)
    s = min(xres, yres)
    sx = ((px + jx) - xres / 2) / (s / 2) * half
    sy = (yres / 2 - (py + jy)) / (s / 2) * half
    d = w[None] + sx[:, None] * right[None] + sy[:, None] * upv[None]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros(len(px), RAY_DTYPE)
    rays["o"] = eye.astype(np.float32)
    rays["d"] = d.astype(np.float32)
    rays["tmax"] = np.inf
    if return_pixels:
        return rays, px.astype(np.int32), py.astype(np.int32)
    return rays


def hit_points(rays, hits, verts, tris):
    """World hit point and unit geometric normal (facing the ray origin) for hit rays."""
    m = hits["prim"] >= 0
    r, h = rays[m], hits[m]
    tri = tris[h["prim"]]
    p0, p1, p2 = verts[tri[:, 0]], verts[tri[:, 1]], verts[tri[:, 2]]
    p = (h["b0"][:, None] * p0 + h["b1"][:, None] * p1 + h["b2"][:, None] * p2).astype(np.float64)
    n = np.cross(p1 - p0, p2 - p0).astype(np.float64)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = n / np.where(ln > 0, ln, 1)
    flip = (n * r["d"]).sum(1) > 0
    n[flip] *= -1
    return p, n, m


def bounce_rays(rays, hits, verts, tris, seed=2, eps_scale=1e-4):
    """One cosine-weighted diffuse bounce per hit ray (closest-hit class 'bounce')."""
    p, n, _ = hit_points(rays, hits, verts, tris)
    rng = np.random.default_rng(seed)
    u1, u2 = rng.random(len(p)), rng.random(len(p))
    rr, phi = np.sqrt(u1), 2 * np.pi * u2
    lx, ly, lz = rr * np.cos(phi), rr * np.sin(phi), np.sqrt(np.maximum(0, 1 - u1))
    a = np.where(np.abs(n[:, :1]) > 0.9, np.array([[0, 1, 0]]), np.array([[1, 0, 0]]))
    t = np.cross(n, a)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(n, t)
    d = lx[:, None] * t + ly[:, None] * b + lz[:, None] * n
    scale = float(np.abs(verts).max())
    out = np.zeros(len(p), RAY_DTYPE)
    out["o"] = (p + n * eps_scale * scale).astype(np.float32)
    out["d"] = d.astype(np.float32)
    out["tmax"] = np.inf
    return out


def shadow_rays(rays, hits, verts, tris, light_lo, light_hi, seed=3, eps_scale=1e-4):
    """Shadow rays from hit points to uniformly sampled points in an axis-aligned light box
    region: un-normalised d = pLight - p and tMax = 1 - ShadowEpsilon, exactly the shape
    Integrator::Unoccluded passes to IntersectP (cpu/integrators.h:52-54, util/math.h:42)."""
    p, n, _ = hit_points(rays, hits, verts, tris)
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(light_lo, np.float64), np.asarray(light_hi, np.float64)
    pl = lo + rng.random((len(p), 3)) * (hi - lo)
    scale = float(np.abs(verts).max())
    o = (p + n * eps_scale * scale).astype(np.float32)
    out = np.zeros(len(p), RAY_DTYPE)
    out["o"] = o
    out["d"] = pl.astype(np.float32) - o
    out["tmax"] = np.float32(1 - 1e-4)
    return out


def shadow_rays_to_quads(rays, hits, verts, tris, quads, seed=3, eps_scale=1e-4):
    """Shadow rays from hit points to uniformly sampled points on area-light quads (corners
    p0 p1 p2 p3 in order; the light is chosen uniformly, the point bilinearly — the quads are
    parallelograms): un-normalised d = pLight - p, tMax = 1 - ShadowEpsilon."""
    p, n, _ = hit_points(rays, hits, verts, tris)
    rng = np.random.default_rng(seed)
    q = np.asarray(quads, np.float64)[rng.integers(0, len(quads), len(p))]
    u, v = rng.random((len(p), 1)), rng.random((len(p), 1))
    pl = q[:, 0] + u * (q[:, 1] - q[:, 0]) + v * (q[:, 3] - q[:, 0])
    scale = float(np.abs(verts).max())
    o = (p + n * eps_scale * scale).astype(np.float32)
    out = np.zeros(len(p), RAY_DTYPE)
    out["o"] = o
    out["d"] = pl.astype(np.float32) - o
    out["tmax"] = np.float32(1 - 1e-4)
    return out


def random_rays(n, lo, hi, seed=5, tmax=np.inf):
    """Incoherent rays between random points of a box (robustness / property tests)."""
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    a = lo + rng.random((n, 3)) * (hi - lo)
    b = lo + rng.random((n, 3)) * (hi - lo)
    rays = np.zeros(n, RAY_DTYPE)
    rays["o"] = a.astype(np.float32)
    rays["d"] = (b - a).astype(np.float32)
    rays["tmax"] = tmax
    return rays


# ---- text ingestion: pbrt `Shape "trianglemesh"` blocks and Wavefront .obj ----------------------
def read_pbrt_trianglemeshes(path):
    """Triangle meshes of a pbrt scene text file: every `"integer indices" [ ... ]` list with its
    `"point3 P" [ ... ]` list (also the v3 spelling `"point P"`), concatenated into one vertex / index
    array with the indices rebased — the geometry machine_learning/nn_parser.py:6-98
    (parse_pbrt_file_with_meshes) extracts, without its one-statement-per-line restrictions.
    Comments (# ...) are dropped.  Returns (verts float32 [n, 3], tris int32 [m, 3], mesh_of_tri
    int32 [m]); transforms (CTM) are NOT applied: like the reference parser this reads object-space
    coordinates (the shipped benchmark scenes carry only camera transforms)."""
    import re
    text = open(path, "r").read()
    text = re.sub(r"#[^\n]*", " ", text)
    lists = re.findall(r'"(integer\s+indices|point3\s+P|point\s+P)"\s*\[([^\]]*)\]', text)
    verts, tris, mesh_of = [], [], []
    pending_idx, pending_p, base, mesh = None, None, 0, 0
    for name, body in lists:
        if name.startswith("integer"):
            pending_idx = np.array(body.split(), np.int64)
        else:
            pending_p = np.array(body.split(), np.float64).astype(np.float32)
        if pending_idx is not None and pending_p is not None:
            if len(pending_idx) % 3 or len(pending_p) % 3:
                raise ValueError(f"{path}: trianglemesh {mesh}: index / coordinate count not a multiple of 3")
            p = pending_p.reshape(-1, 3)
            t = pending_idx.reshape(-1, 3)
            if len(t) and (t.min() < 0 or t.max() >= len(p)):
                raise ValueError(f"{path}: trianglemesh {mesh}: vertex index out of range")
            verts.append(p)
            tris.append((t + base).astype(np.int32))
            mesh_of.append(np.full(len(t), mesh, np.int32))
            base += len(p)
            mesh += 1
            pending_idx = pending_p = None
    if not verts:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), np.zeros(0, np.int32)
    return np.concatenate(verts), np.concatenate(tris), np.concatenate(mesh_of)


def read_obj(path):
    """Wavefront .obj as machine_learning/nn_parser.py:130-166 (parse_obj_file_with_meshes) reads it:
    `v x y z` vertices, `f a/.. b/.. c/..` faces (first three corners; 1-based), `g` starts a new mesh.
    Returns (verts float32 [n, 3], tris int32 [m, 3], mesh_of_tri int32 [m])."""
    verts, tris, mesh_of, mesh = [], [], [], 0
    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                verts.append([float(tok[1]), float(tok[2]), float(tok[3])])
            elif tok[0] == "f":
                tris.append([int(tok[k].split("/")[0]) - 1 for k in (1, 2, 3)])
                mesh_of.append(mesh)
            elif tok[0] == "g":
                mesh += 1
    verts = np.array(verts, np.float32).reshape(-1, 3)
    tris = np.array(tris, np.int32).reshape(-1, 3)
    if len(tris) and (tris.min() < 0 or tris.max() >= len(verts)):
        raise ValueError(f"{path}: face index out of range")
    return verts, tris, np.array(mesh_of, np.int32)
