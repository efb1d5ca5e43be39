"""Host-side mirror of the KdTreeAggregate entry points (include/nnbvh.h, nnbvh_kd_*): pbrt's
kd-tree accelerator (/root/reference/src/pbrt/cpu/aggregates.cpp:746-1161) and an importer for the
plane arrays the nss learned kd-trees are exported as (machine_learning/nss_kd_tree.py:204-240)."""
import collections
import ctypes

import numpy as np

from . import _lib
from ._lib import HIT_DTYPE, KD_NODE_DTYPE, PRIM_DTYPE, RAY_DTYPE, check, ptr

KdBuilt = collections.namedtuple("KdBuilt", "nodes prim_indices bounds depth build_ms", defaults=((0.0, 0.0),))


def build_kd_tree(prims, verts, prim_bounds=None, isect_cost=5, traversal_cost=1, empty_bonus=0.5, max_prims=1,
                  max_depth=-1, where="host", device=0):
    """KdTreeAggregate's constructor (aggregates.cpp:798-971); defaults are those of KdTreeAggregate::Create
    (aggregates.cpp:1152-1161).  where: "host" (libstdc++'s std::sort order inside multi-primitive leaves),
    "host_stable" (std::stable_sort order) or "gpu" (the device builder: the same arrays as "host_stable")."""
    L = _lib.lib()
    prims = np.ascontiguousarray(prims, PRIM_DTYPE)
    verts = np.ascontiguousarray(verts, np.float32)
    pb = None if prim_bounds is None else np.ascontiguousarray(prim_bounds, np.float32)
    args = (ptr(prims), len(prims), ptr(verts), len(verts), None if pb is None else ptr(pb),
            isect_cost, traversal_cost, ctypes.c_float(empty_bonus), max_prims, max_depth)
    if where == "gpu":
        h = L.nnbvh_kd_build_create_gpu(*args, device)
    elif where == "host_stable":
        h = L.nnbvh_kd_build_create_stable(*args)
    else:
        h = L.nnbvh_kd_build_create(*args)
    if not h:
        raise _lib.NNBVHError(f"nnbvh_kd_build_create ({where}) failed: {_lib.last_error()}")
    try:
        n = ctypes.c_int()
        p = L.nnbvh_kd_build_nodes(h, ctypes.byref(n))
        nodes = np.frombuffer((ctypes.c_char * (n.value * 8)).from_address(p), KD_NODE_DTYPE).copy()
        p = L.nnbvh_kd_build_prim_indices(h, ctypes.byref(n))
        idx = (np.frombuffer((ctypes.c_char * (n.value * 4)).from_address(p), np.int32).copy()
               if n.value else np.zeros(0, np.int32))
        bounds = np.zeros(6, np.float32)
        check(L.nnbvh_kd_build_bounds(h, ptr(bounds)), "nnbvh_kd_build_bounds")
        depth = L.nnbvh_kd_build_depth(h)
        ms = np.zeros(2, np.float64)
        check(L.nnbvh_kd_build_timing(h, ptr(ms)), "nnbvh_kd_build_timing")
    finally:
        L.nnbvh_kd_build_destroy(h)
    return KdBuilt(nodes, idx, bounds, depth, (float(ms[0]), float(ms[1])))


class KdTreeAggregate:
    """KdTreeAggregate (cpu/aggregates.h:75-105) resident on the device."""

    def __init__(self, handle, bounds):
        self._h = handle
        self.bounds = bounds

    @classmethod
    def from_tree(cls, nodes, prim_indices, prims, verts, bounds, device=0, normals=None, uvs=None, prim_alpha=None):
        """normals / uvs (per vertex) and prim_alpha (per entry of prims): what the alpha-tested kinds of smooth meshes
        and the alpha-tested patches read; without them such primitives are the host's (void records)."""
        nodes = np.ascontiguousarray(nodes, KD_NODE_DTYPE)
        idx = np.ascontiguousarray(prim_indices, np.int32)
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        bounds = np.ascontiguousarray(bounds, np.float32)
        nrm = None if normals is None else np.ascontiguousarray(normals, np.float32).reshape(len(verts), 3)
        uv = None if uvs is None else np.ascontiguousarray(uvs, np.float32).reshape(len(verts), 2)
        pa = None if prim_alpha is None else np.ascontiguousarray(prim_alpha, np.float32).reshape(len(prims))
        opt = lambda a: ptr(a) if a is not None else None  # noqa: E731
        h = _lib.lib().nnbvh_kd_scene_create_with_attributes(
            ptr(nodes), len(nodes), ptr(idx) if len(idx) else None, len(idx), ptr(prims), len(prims), ptr(verts),
            len(verts), ptr(bounds), opt(nrm), opt(uv), opt(pa), device)
        if not h:
            raise _lib.NNBVHError(f"nnbvh_kd_scene_create failed: {_lib.last_error()}")
        return cls(h, bounds)

    @classmethod
    def build(cls, prims, verts, device=0, normals=None, uvs=None, prim_alpha=None, **kw):
        t = build_kd_tree(prims, verts, **kw)
        return cls.from_tree(t.nodes, t.prim_indices, prims, verts, t.bounds, device, normals, uvs, prim_alpha)

    def close(self):
        if self._h:
            _lib.lib().nnbvh_kd_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Bounds(self):
        return self.bounds[:3].copy(), self.bounds[3:].copy()

    def Intersect(self, rays):
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        hits = np.zeros(len(rays), HIT_DTYPE)
        check(_lib.lib().nnbvh_kd_intersect_closest(self._h, ptr(rays), len(rays), ptr(hits)),
              "nnbvh_kd_intersect_closest")
        return hits

    def IntersectP(self, rays, counts=False):
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        occ = np.zeros(len(rays), np.uint8)
        vis = np.zeros(len(rays), np.int32) if counts else None
        tst = np.zeros(len(rays), np.int32) if counts else None
        check(_lib.lib().nnbvh_kd_intersect_any(self._h, ptr(rays), len(rays), ptr(occ),
                                                ptr(vis) if counts else None, ptr(tst) if counts else None),
              "nnbvh_kd_intersect_any")
        return (occ, vis, tst) if counts else occ

    def intersect_device(self, d_rays, d_hits, n, stream=0):
        check(_lib.lib().nnbvh_kd_intersect_closest_device(self._h, ctypes.c_void_p(d_rays), n,
                                                           ctypes.c_void_p(d_hits), ctypes.c_void_p(stream)),
              "nnbvh_kd_intersect_closest_device")

    def intersect_p_device(self, d_rays, d_occ, n, d_visited=None, d_tests=None, stream=0):
        check(_lib.lib().nnbvh_kd_intersect_any_device(self._h, ctypes.c_void_p(d_rays), n, ctypes.c_void_p(d_occ),
                                                       ctypes.c_void_p(d_visited) if d_visited else None,
                                                       ctypes.c_void_p(d_tests) if d_tests else None,
                                                       ctypes.c_void_p(stream)),
              "nnbvh_kd_intersect_any_device")


# ---- nss learned kd-trees ---------------------------------------------------------------------
def prim_bounds_of(prims, verts):
    """Per-primitive bounds (Triangle::Bounds / BilinearPatch::Bounds: min / max over the vertices)."""
    prims = np.asarray(prims)
    v = np.asarray(verts, np.float32)[prims["v"]]
    tri = np.isin(prims["kind"], (0, 4, 5))[:, None, None]  # triangles and alpha-tested triangles
    lo = np.where(tri, v[:, :3].min(1, keepdims=True), v.min(1, keepdims=True))[:, 0]
    hi = np.where(tri, v[:, :3].max(1, keepdims=True), v.max(1, keepdims=True))[:, 0]
    return lo.astype(np.float32), hi.astype(np.float32)


def kd_from_planes(planes, prims, verts, scale=None, translate=0.0):
    """KdTreeNode array from an nss plane array.

    `planes` is what kdTree.exportTree_structure / getPlaneArray write (machine_learning/
    nss_kd_tree.py:204-216, 239-240; np.savez key 'b'): float32 [n, 5] in LEVEL order, level l holding
    2**l rows, row = (one-hot split axis x y z, unused, offset - pc_translation) in the NORMALISED
    point-cloud frame.  scale / translate map a normalised offset back to scene space: split =
    offset * scale[axis] + translate[axis] (None: the scene's bounding box, i.e. the
    getAABBox / applyNormalization frame of nss_common).  A row whose axis is all-zero ends the tree on
    that branch (leaf).  Primitives are distributed as KdTreeAggregate::buildTree does (a primitive
    overlapping the plane goes to both sides: below if its min <= split... see classify below), and the
    nodes are laid out in the reference's order (below child at index + 1), so the result feeds
    nnbvh_kd_scene_create.  The nss exporter has no counterpart for leaves; the last plane level's
    children are leaves holding the primitives that reach them.

    Parity: unpinned — the reference ships no exported tree; tests/test_kdtree.py holds a
    synthetic known answer."""
    planes = np.asarray(planes, np.float32).reshape(-1, 5)
    prims = np.ascontiguousarray(prims, PRIM_DTYPE)
    lo, hi = prim_bounds_of(prims, verts)
    bmin, bmax = lo.min(0), hi.max(0)
    if scale is None:
        scale = (bmax - bmin).astype(np.float32)
        translate = bmin.astype(np.float32)
    scale = np.broadcast_to(np.asarray(scale, np.float32), (3,))
    translate = np.broadcast_to(np.asarray(translate, np.float32), (3,))
    n_levels = int(np.log2(len(planes) + 1))
    if 2 ** n_levels - 1 != len(planes):
        raise ValueError("plane array must hold a complete level-order tree (2**levels - 1 rows)")
    nodes, indices = [], []

    def emit(level, slot, ids):
        me = len(nodes)
        nodes.append([0, 0])
        row = planes[2 ** level - 1 + slot] if level < n_levels else None
        axis = -1 if row is None or not row[:3].any() else int(np.argmax(row[:3]))
        if axis < 0 or len(ids) == 0:
            n = len(ids)
            if n == 1:
                nodes[me] = [int(ids[0]), 3 | (1 << 2)]
            else:
                nodes[me] = [len(indices) if n else 0, 3 | (n << 2)]
                indices.extend(int(i) for i in ids)
            return
        split = np.float32(row[4] * scale[axis] + translate[axis])
        # buildTree's classification at an edge t = split (aggregates.cpp:954-960): below gets every
        # primitive that starts before the split plane, above every primitive that ends after it;
        # a primitive lying IN the plane (min == max == split) goes below
        below = ids[(lo[ids, axis] < split) | ((lo[ids, axis] == split) & (hi[ids, axis] == split))]
        above = ids[hi[ids, axis] > split]
        emit(level + 1, 2 * slot, below)
        nodes[me] = [int(split.view(np.uint32)), axis | (len(nodes) << 2)]
        emit(level + 1, 2 * slot + 1, above)

    emit(0, 0, np.arange(len(prims)))
    out = np.zeros(len(nodes), KD_NODE_DTYPE)
    out["split_or_index"] = np.array([n[0] for n in nodes], np.int64).astype(np.uint32)
    out["flags"] = np.array([n[1] for n in nodes], np.uint32)
    bounds = np.concatenate([bmin, bmax]).astype(np.float32)
    depth = n_levels
    return KdBuilt(out, np.array(indices, np.int32), bounds, depth)
