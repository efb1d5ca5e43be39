"""Host-side mirror of the reference's aggregate interface for the accelerated path.

`BVHAggregate` here has the reference's method names and argument meaning
(/root/reference/src/pbrt/cpu/aggregates.h:28-70: Create / Bounds / Intersect / IntersectP),
batched the way the wavefront caller batches them (wavefront/aggregate.cpp:34-68).  All
arithmetic happens in libnnbvh_hip.so on the GPU; numpy/torch only carry buffers.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import HIT_DTYPE, NODE_DTYPE, PRIM_DTYPE, RAY_DTYPE, NNBVHError, check, ptr

SPLIT_METHODS = {"sah": 0, "hlbvh": 1, "middle": 2, "equal": 3}


def make_prims(tri_indices=None, patch_indices=None):
    """Primitive table in creation order: triangles first, then bilinear patches.
    ids are the running primitive index (the reference's position in `prims`)."""
    nt = 0 if tri_indices is None else len(tri_indices)
    npch = 0 if patch_indices is None else len(patch_indices)
    prims = np.zeros(nt + npch, PRIM_DTYPE)
    prims["id"] = np.arange(nt + npch, dtype=np.int32)
    if nt:
        prims["kind"][:nt] = 0
        prims["v"][:nt, :3] = np.asarray(tri_indices, np.int32).reshape(nt, 3)
    if npch:
        prims["kind"][nt:] = 1
        prims["v"][nt:] = np.asarray(patch_indices, np.int32).reshape(npch, 4)
    return prims


def make_rays(o, d, tmax=np.inf, time=0.0):
    o = np.asarray(o, np.float32).reshape(-1, 3)
    rays = np.zeros(len(o), RAY_DTYPE)
    rays["o"] = o
    rays["d"] = np.asarray(d, np.float32).reshape(-1, 3)
    rays["tmax"] = tmax
    rays["time"] = time
    return rays


class BuiltTree:
    """Output of the host builder (BVHAggregate ctor + flattenBVH, aggregates.cpp:140-522)."""

    def __init__(self, nodes, ordered_prims, depth):
        self.nodes = nodes
        self.ordered_prims = ordered_prims
        self.depth = depth


def _take_build(L, h, what):
    if not h:
        raise NNBVHError(what + ": " + _lib.last_error())
    try:
        n = ctypes.c_int(0)
        pn = L.nnbvh_build_nodes(h, ctypes.byref(n))
        nodes = np.frombuffer((ctypes.c_char * (n.value * 32)).from_address(pn),
                              NODE_DTYPE).copy()
        pp = L.nnbvh_build_ordered_prims(h, ctypes.byref(n))
        ordered = np.frombuffer((ctypes.c_char * (n.value * 24)).from_address(pp),
                                PRIM_DTYPE).copy()
        tree = BuiltTree(nodes, ordered, L.nnbvh_build_depth(h))
        ms = np.zeros(5, np.float64)
        L.nnbvh_build_gpu_timing(h, ptr(ms))
        tree.gpu_ms = [float(x) for x in ms]  # phases: see nnbvh_build_gpu_timing
    finally:
        L.nnbvh_build_destroy(h)
    return tree


def build_tree(prims, verts, max_prims_in_node=4, split_method="sah", prim_bounds=None):
    """SAH / HLBVH / middle / equal-counts build on the host (no GPU needed).  prim_bounds
    ([n, 6] float32) is required when the list contains instance primitives (kind 2)."""
    L = _lib.lib()
    prims = np.ascontiguousarray(prims, PRIM_DTYPE)
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
    if split_method not in SPLIT_METHODS:
        # aggregates.cpp:735-738 warns and falls back to sah; we report instead
        raise NNBVHError(f'BVH split method "{split_method}" unknown')
    if prim_bounds is not None:
        prim_bounds = np.ascontiguousarray(prim_bounds, np.float32).reshape(len(prims), 6)
        h = L.nnbvh_build_create_with_bounds(ptr(prims), len(prims), ptr(verts), len(verts),
                                             ptr(prim_bounds), int(max_prims_in_node),
                                             SPLIT_METHODS[split_method])
    else:
        h = L.nnbvh_build_create(ptr(prims), len(prims), ptr(verts), len(verts),
                                 int(max_prims_in_node), SPLIT_METHODS[split_method])
    return _take_build(L, h, "nnbvh_build_create")


def build_tree_gpu(prims, verts, max_prims_in_node=4, prim_bounds=None, device=0, split_method="hlbvh"):
    """SAH or HLBVH tree built on the GPU: byte-identical to build_tree(..., split_method)."""
    if split_method not in ("sah", "hlbvh"):
        raise NNBVHError(f'GPU build supports "sah" and "hlbvh", not "{split_method}"')
    L = _lib.lib()
    prims = np.ascontiguousarray(prims, PRIM_DTYPE)
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
    pb = None
    if prim_bounds is not None:
        pb = np.ascontiguousarray(prim_bounds, np.float32).reshape(len(prims), 6)
    h = L.nnbvh_build_create_gpu(ptr(prims), len(prims), ptr(verts), len(verts),
                                 ptr(pb) if pb is not None else None, int(max_prims_in_node),
                                 SPLIT_METHODS[split_method], int(device))
    return _take_build(L, h, "nnbvh_build_create_gpu")


class BVHAggregate:
    """GPU-resident aggregate.  Construct from primitives (builds the tree on the host like
    BVHAggregate::Create, aggregates.cpp:725-744: splitmethod "sah", maxnodeprims 4) or from
    an already flattened tree via `from_tree`."""

    def __init__(self, prims, verts, max_prims_in_node=4, split_method="sah", device=0):
        tree = build_tree(prims, verts, max_prims_in_node, split_method)
        self._init(tree.nodes, tree.ordered_prims, verts, device, tree.depth)

    @classmethod
    def from_tree(cls, nodes, ordered_prims, verts, device=0, instances=None, n_top_nodes=None, animated=None,
                  normals=None, prim_alpha=None, uvs=None):
        """instances (INSTANCE_DTYPE) + n_top_nodes make a two-level scene: nodes[:n_top_nodes] is
        the top-level tree, the child trees follow (see nn_bvh_amd.instancing).  animated
        (ANIMATED_DTYPE, one per instance) turns instances into AnimatedPrimitives.  normals: per-vertex
        shading normals, needed by alpha-tested triangles / patches of smooth meshes (prim kinds 6 / 7, 10 / 11);
        prim_alpha: one constant alpha per entry of ordered_prims, read for alpha-tested patches (kinds 8 .. 15);
        uvs: per-vertex (u, v), read for the alpha-tested patches of meshes with uv (kinds 12 .. 15)."""
        self = cls.__new__(cls)
        self._init(nodes, ordered_prims, verts, device, None, instances, n_top_nodes, animated, normals, prim_alpha, uvs)
        return self

    @classmethod
    def build_on_device(cls, prims, verts, max_prims_in_node=4, split_method="sah", prim_bounds=None,
                        device=0, normals=None, prim_alpha=None, uvs=None):
        """Tree built AND baked on the GPU (nnbvh_scene_create_gpu_build): the tree never visits the
        host.  Same traversal results as the host-built aggregate; `nodes` / `ordered_prims` are
        not available on this object.  normals (per vertex) / prim_alpha (per entry of `prims`, the caller's order):
        what the smooth alpha-tested kinds and the alpha-tested patches read."""
        if split_method not in ("sah", "hlbvh"):
            raise NNBVHError(f'GPU build supports "sah" and "hlbvh", not "{split_method}"')
        self = cls.__new__(cls)
        L = _lib.lib()
        prims = np.ascontiguousarray(prims, PRIM_DTYPE)
        verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        pb = None
        if prim_bounds is not None:
            pb = np.ascontiguousarray(prim_bounds, np.float32).reshape(len(prims), 6)
        self.nodes = self.ordered_prims = None
        self.verts = verts
        self.device = int(device)
        nrm = None if normals is None else np.ascontiguousarray(normals, np.float32).reshape(len(verts), 3)
        pa = None if prim_alpha is None else np.ascontiguousarray(prim_alpha, np.float32).reshape(len(prims))
        uv = None if uvs is None else np.ascontiguousarray(uvs, np.float32).reshape(len(verts), 2)
        self._h = L.nnbvh_scene_create_gpu_build_with_attributes(
            ptr(prims), len(prims), ptr(verts), len(verts), ptr(pb) if pb is not None else None,
            ptr(nrm) if nrm is not None else None, ptr(uv) if uv is not None else None,
            ptr(pa) if pa is not None else None,
            int(max_prims_in_node), SPLIT_METHODS[split_method], self.device)
        if not self._h:
            raise NNBVHError("nnbvh_scene_create_gpu_build: " + _lib.last_error())
        self._read_info()
        return self

    def _read_info(self):
        info = np.zeros(6, np.int64)
        check(_lib.lib().nnbvh_scene_info(self._h, ptr(info)), "nnbvh_scene_info")
        self.info = {"interior_records": int(info[0]), "prim_slots": int(info[1]),
                     "depth": int(info[2]), "device_bytes": int(info[3]),
                     "grid_blocks": int(info[4]), "stack_window": int(info[5])}

    def _init(self, nodes, ordered_prims, verts, device, depth, instances=None, n_top_nodes=None, animated=None,
              normals=None, prim_alpha=None, uvs=None):
        L = _lib.lib()
        self.nodes = np.ascontiguousarray(nodes, NODE_DTYPE)
        self.ordered_prims = np.ascontiguousarray(ordered_prims, PRIM_DTYPE)
        self.verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        self.device = int(device)
        if instances is not None and len(instances):
            self.instances = np.ascontiguousarray(instances, _lib.INSTANCE_DTYPE)
            if normals is not None or prim_alpha is not None or uvs is not None:
                f32 = lambda a, w: None if a is None else np.ascontiguousarray(a, np.float32).reshape(-1, w)  # noqa: E731
                self.normals, self.uvs = f32(normals, 3), f32(uvs, 2)
                self.prim_alpha = None if prim_alpha is None else np.ascontiguousarray(prim_alpha, np.float32).reshape(-1)
                self.animated = None if animated is None else np.ascontiguousarray(animated, _lib.ANIMATED_DTYPE)
                opt = lambda a: ptr(a) if a is not None else None  # noqa: E731
                self._h = L.nnbvh_scene_create_instanced_with_attributes(
                    ptr(self.nodes), len(self.nodes), int(n_top_nodes), ptr(self.ordered_prims),
                    len(self.ordered_prims), ptr(self.verts), len(self.verts), ptr(self.instances),
                    len(self.instances), opt(self.animated), opt(self.normals), opt(self.uvs), opt(self.prim_alpha),
                    self.device)
            elif animated is not None:
                self.animated = np.ascontiguousarray(animated, _lib.ANIMATED_DTYPE)
                assert len(self.animated) == len(self.instances)
                self._h = L.nnbvh_scene_create_instanced_animated(
                    ptr(self.nodes), len(self.nodes), int(n_top_nodes), ptr(self.ordered_prims),
                    len(self.ordered_prims), ptr(self.verts), len(self.verts), ptr(self.instances),
                    len(self.instances), ptr(self.animated), self.device)
            else:
                self._h = L.nnbvh_scene_create_instanced(
                    ptr(self.nodes), len(self.nodes), int(n_top_nodes), ptr(self.ordered_prims),
                    len(self.ordered_prims), ptr(self.verts), len(self.verts), ptr(self.instances),
                    len(self.instances), self.device)
        elif prim_alpha is not None:
            self.normals = None if normals is None else np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
            self.prim_alpha = np.ascontiguousarray(prim_alpha, np.float32).reshape(-1)
            self.uvs = None if uvs is None else np.ascontiguousarray(uvs, np.float32).reshape(-1, 2)
            assert len(self.prim_alpha) == len(self.ordered_prims)
            assert self.normals is None or len(self.normals) == len(self.verts)
            assert self.uvs is None or len(self.uvs) == len(self.verts)
            self._h = L.nnbvh_scene_create_with_attributes(
                ptr(self.nodes), len(self.nodes), ptr(self.ordered_prims), len(self.ordered_prims), ptr(self.verts),
                ptr(self.normals) if self.normals is not None else None,
                ptr(self.uvs) if self.uvs is not None else None, ptr(self.prim_alpha), len(self.verts), self.device)
        elif normals is not None:
            self.normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
            assert len(self.normals) == len(self.verts)
            self._h = L.nnbvh_scene_create_with_normals(ptr(self.nodes), len(self.nodes), ptr(self.ordered_prims),
                                                        len(self.ordered_prims), ptr(self.verts), ptr(self.normals),
                                                        len(self.verts), self.device)
        else:
            self._h = L.nnbvh_scene_create(ptr(self.nodes), len(self.nodes), ptr(self.ordered_prims),
                                           len(self.ordered_prims), ptr(self.verts), len(self.verts),
                                           self.device)
        if not self._h:
            raise NNBVHError("nnbvh_scene_create: " + _lib.last_error())
        self._read_info()

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().nnbvh_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        check(_lib.lib().nnbvh_scene_set_option(self._h, key.encode(), int(value)),
              f"set_option({key})")

    def sched_stats(self, reset=True):
        """Diagnostics (NNBVH_STATS builds): trips and lanes per step kind."""
        out = np.zeros(16, np.uint64)
        check(_lib.lib().nnbvh_scene_sched_stats(self._h, ptr(out), int(reset)), "sched_stats")
        return dict(zip(("int_trips", "int_lanes", "prim_trips", "prim_lanes", "refill_trips",
                         "refill_lanes", "i_nint", "i_nprim", "i_nidle", "spare0", "int_cycles",
                         "prim_cycles", "refill_cycles", "spare1", "int_steps", "int_step_lanes"),
                        (int(x) for x in out)))

    # -- reference interface ----------------------------------------------------------
    def Bounds(self):
        out = np.zeros(6, np.float32)
        check(_lib.lib().nnbvh_scene_bounds(self._h, ptr(out)), "nnbvh_scene_bounds")
        return out[:3].copy(), out[3:].copy()

    def Intersect(self, rays, out=None):
        """Closest hit for a host ray batch (RAY_DTYPE) -> HIT_DTYPE array (`out`: write into this array,
        e.g. one the caller has pinned)."""
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        hits = np.zeros(len(rays), HIT_DTYPE) if out is None else out
        assert hits.dtype == HIT_DTYPE and len(hits) == len(rays) and hits.flags.c_contiguous
        check(_lib.lib().nnbvh_intersect_closest(self._h, ptr(rays), len(rays), ptr(hits)),
              "nnbvh_intersect_closest")
        return hits

    def IntersectP(self, rays, counts=False):
        """Any hit -> uint8 occluded[, nodes_visited, prim_tests]."""
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        occ = np.zeros(len(rays), np.uint8)
        if counts:
            vis = np.zeros(len(rays), np.int32)
            tst = np.zeros(len(rays), np.int32)
            check(_lib.lib().nnbvh_intersect_any(self._h, ptr(rays), len(rays), ptr(occ),
                                                 ptr(vis), ptr(tst)), "nnbvh_intersect_any")
            return occ, vis, tst
        check(_lib.lib().nnbvh_intersect_any(self._h, ptr(rays), len(rays), ptr(occ), None, None),
              "nnbvh_intersect_any")
        return occ

    # -- device-resident batches (torch tensors are only buffers + the current stream) ----
    def intersect_device(self, d_rays, d_hits, n, stream=0):
        check(_lib.lib().nnbvh_intersect_closest_device(self._h, d_rays, n, d_hits, stream),
              "nnbvh_intersect_closest_device")

    def trace_batches_device(self, batches, stream=0):
        """batches: iterable of (kind, d_rays, n, d_out[, d_nodes_visited, d_prim_tests]) with kind
        "closest" or "any"; all traced concurrently, ordered as one operation on `stream`."""
        arr = np.zeros(len(batches), _lib.BATCH_DTYPE)
        for i, b in enumerate(batches):
            arr[i]["kind"] = {"closest": 0, "any": 1}[b[0]]
            arr[i]["d_rays"], arr[i]["n"], arr[i]["d_out"] = b[1], b[2], b[3]
            if len(b) > 4:
                arr[i]["d_nodes_visited"], arr[i]["d_prim_tests"] = b[4] or 0, b[5] or 0
        check(_lib.lib().nnbvh_trace_batches_device(self._h, ptr(arr), len(arr), stream),
              "nnbvh_trace_batches_device")

    def intersect_p_device(self, d_rays, d_occ, n, d_visited=None, d_tests=None, stream=0):
        check(_lib.lib().nnbvh_intersect_any_device(self._h, d_rays, n, d_occ, d_visited, d_tests,
                                                    stream), "nnbvh_intersect_any_device")
