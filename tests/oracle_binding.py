"""ctypes binding of oracle/libnnbvh_oracle.so — TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package (nn_bvh_amd/)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# NNBVH_ORACLE_LIB: a diagnostic build of the same source (tools/asan_cpu.sh)
ORACLE_LIB = os.environ.get("NNBVH_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libnnbvh_oracle.so")

_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
    return ORACLE_LIB


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "nnbvh_oracle.c")
        if not os.path.exists(ORACLE_LIB) or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(src):
            build()
        _lib = ctypes.CDLL(ORACLE_LIB)
    return _lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


def closest(nodes, prims, verts, rays, nthreads=1):
    from nn_bvh_amd._lib import HIT_DTYPE
    nodes, prims = np.ascontiguousarray(nodes), np.ascontiguousarray(prims)
    verts = np.ascontiguousarray(verts, np.float32)
    rays = np.ascontiguousarray(rays)
    hits = np.zeros(len(rays), HIT_DTYPE)
    lib().orc_intersect_closest(_p(nodes), ctypes.c_int(len(nodes)), _p(prims), _p(verts), _p(rays),
                                ctypes.c_int64(len(rays)), _p(hits), ctypes.c_int(nthreads))
    return hits


def any_hit(nodes, prims, verts, rays, nthreads=1):
    nodes, prims = np.ascontiguousarray(nodes), np.ascontiguousarray(prims)
    verts = np.ascontiguousarray(verts, np.float32)
    rays = np.ascontiguousarray(rays)
    occ = np.zeros(len(rays), np.uint8)
    vis = np.zeros(len(rays), np.int32)
    tst = np.zeros(len(rays), np.int32)
    lib().orc_intersect_any(_p(nodes), ctypes.c_int(len(nodes)), _p(prims), _p(verts), _p(rays),
                            ctypes.c_int64(len(rays)), _p(occ), _p(vis), _p(tst),
                            ctypes.c_int(nthreads))
    return occ, vis, tst


def brute_closest(prims, verts, rays):
    from nn_bvh_amd._lib import HIT_DTYPE
    prims = np.ascontiguousarray(prims)
    verts = np.ascontiguousarray(verts, np.float32)
    rays = np.ascontiguousarray(rays)
    hits = np.zeros(len(rays), HIT_DTYPE)
    lib().orc_brute_closest(_p(prims), ctypes.c_int(len(prims)), _p(verts), _p(rays),
                            ctypes.c_int64(len(rays)), _p(hits))
    return hits


def leaf_batch(mode, inputs):
    """mode in {'tri','blp','slab'}; inputs = golden record array (see tests/golden/make_leaf_golden.py).
    Returns (hit uint8[n], out float32[n,k])."""
    r = np.ascontiguousarray(inputs, np.float32)
    n = len(r)
    o = np.ascontiguousarray(r[:, 0:3])
    d = np.ascontiguousarray(r[:, 3:6])
    t = np.ascontiguousarray(r[:, 6])
    p = np.ascontiguousarray(r[:, 7:])
    hit = np.zeros(n, np.uint8)
    if mode == "tri":
        out = np.zeros((n, 4), np.float32)
        lib().orc_triangle_batch(_p(o), _p(d), _p(t), _p(p), n, _p(hit), _p(out))
    elif mode == "blp":
        out = np.zeros((n, 3), np.float32)
        lib().orc_bilinear_patch_batch(_p(o), _p(d), _p(t), _p(p), n, _p(hit), _p(out))
    else:
        out = np.zeros((n, 0), np.float32)
        lib().orc_slab_batch(_p(p), _p(o), _p(d), _p(t), n, _p(hit))
    return hit, out


INSTANCE_DTYPE = np.dtype([("m", "<f4", 12), ("m_inv", "<f4", 12), ("root", "<i4"), ("n_nodes", "<i4")])


def apply_inverse_ray(m_inv12, o, d, tmax):
    """orc_apply_inverse_ray on a batch: returns float32 [n, 7] = o'[3], d'[3], tmax'."""
    mi = np.ascontiguousarray(m_inv12, np.float32).reshape(-1, 12)
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tmax, np.float32).reshape(-1)
    out = np.zeros((len(o), 7), np.float32)
    lib().orc_apply_inverse_ray_batch(_p(mi), _p(o), _p(d), _p(t), len(o), _p(out))
    return out


def transform_bounds(m12, box6):
    out = np.zeros(6, np.float32)
    m = np.ascontiguousarray(m12, np.float32)
    b = np.ascontiguousarray(box6, np.float32)
    lib().orc_transform_bounds(_p(m), _p(b), _p(out))
    return out


def closest_inst(nodes, prims, verts, instances, rays, nthreads=1):
    from nn_bvh_amd._lib import HIT_DTYPE
    nodes, prims = np.ascontiguousarray(nodes), np.ascontiguousarray(prims)
    verts = np.ascontiguousarray(verts, np.float32)
    instances = np.ascontiguousarray(instances, INSTANCE_DTYPE)
    rays = np.ascontiguousarray(rays)
    hits = np.zeros(len(rays), HIT_DTYPE)
    lib().orc_intersect_closest_inst(_p(nodes), _p(prims), _p(verts), _p(instances), _p(rays),
                                     ctypes.c_int64(len(rays)), _p(hits), ctypes.c_int(nthreads))
    return hits


def any_hit_inst(nodes, prims, verts, instances, rays, nthreads=1):
    nodes, prims = np.ascontiguousarray(nodes), np.ascontiguousarray(prims)
    verts = np.ascontiguousarray(verts, np.float32)
    instances = np.ascontiguousarray(instances, INSTANCE_DTYPE)
    rays = np.ascontiguousarray(rays)
    occ = np.zeros(len(rays), np.uint8)
    vis = np.zeros(len(rays), np.int32)
    tst = np.zeros(len(rays), np.int32)
    lib().orc_intersect_any_inst(_p(nodes), _p(prims), _p(verts), _p(instances), _p(rays),
                                 ctypes.c_int64(len(rays)), _p(occ), _p(vis), _p(tst),
                                 ctypes.c_int(nthreads))
    return occ, vis, tst


def wavefront_enqueue_closest(hits, has_medium=None, prim_class=None):
    """orc_wavefront_enqueue_closest: returns the six index queues (escaped, hit_area_light,
    basic_eval, universal_eval, medium_sample, next_ray) as int32 arrays."""
    hits = np.ascontiguousarray(hits)
    n = len(hits)
    hm = np.ascontiguousarray(has_medium, np.uint8) if has_medium is not None else None
    pc = np.ascontiguousarray(prim_class, np.uint8) if prim_class is not None else None
    bufs = [np.zeros(max(n, 1), np.int32) for _ in range(6)]
    ptrs = (ctypes.c_void_p * 6)(*[b.ctypes.data for b in bufs])
    sizes = (ctypes.c_int32 * 6)()
    lib().orc_wavefront_enqueue_closest(_p(hits), ctypes.c_int(n), _p(hm), _p(pc),
                                        ctypes.c_int64(0 if pc is None else len(pc)), ptrs, sizes)
    return [b[:sizes[k]].copy() for k, b in enumerate(bufs)]


def record_shadow(occluded, Ld, r_u, r_l, pixel_index, L):
    """orc_record_shadow: returns the updated copy of L (float32 [n_pixels, 4])."""
    occ = np.ascontiguousarray(occluded, np.uint8)
    Ld, r_u, r_l = (np.ascontiguousarray(a, np.float32) for a in (Ld, r_u, r_l))
    px = np.ascontiguousarray(pixel_index, np.int32)
    out = np.array(L, np.float32, copy=True)
    lib().orc_record_shadow(_p(occ), ctypes.c_int(len(occ)), _p(Ld), _p(r_u), _p(r_l), _p(px), _p(out))
    return out


def triangle_interaction_batch(records45):
    """orc_triangle_interaction_batch on oracle/ref_interaction.cpp-style records -> float32 [n, 44]."""
    rec = np.ascontiguousarray(records45, np.float32).reshape(-1, 45)
    out = np.zeros((len(rec), 44), np.float32)
    lib().orc_triangle_interaction_batch(_p(rec), ctypes.c_int(len(rec)), _p(out))
    return out


def interaction_branches(reset=False):
    arr = (ctypes.c_long * 9).in_dll(lib(), "orc_interaction_branches")
    vals = list(arr)
    if reset:
        for k in range(9):
            arr[k] = 0
    return vals


def patch_interaction_batch(records40):
    """orc_patch_interaction_batch on oracle/ref_interaction.cpp "blp" records -> float32 [n, 50]."""
    rec = np.ascontiguousarray(records40, np.float32).reshape(-1, 40)
    out = np.zeros((len(rec), 50), np.float32)
    lib().orc_patch_interaction_batch(_p(rec), ctypes.c_int(len(rec)), _p(out))
    return out


def patch_branches(reset=False):
    arr = (ctypes.c_long * 8).in_dll(lib(), "orc_patch_branches")
    vals = list(arr)
    if reset:
        for k in range(8):
            arr[k] = 0
    return vals


def transform_interaction_batch(records72):
    """orc_transform_interaction_batch on oracle/ref_interaction.cpp "xf" records -> float32 [n, 40]."""
    rec = np.ascontiguousarray(records72, np.float32).reshape(-1, 72)
    out = np.zeros((len(rec), 40), np.float32)
    lib().orc_transform_interaction_batch(_p(rec), ctypes.c_int(len(rec)), _p(out))
    return out


def film_add_samples(pixels, bounds, max_component, px, py, rgb, weight, n_passes):
    """UpdateFilm + RGBFilm::AddSample on a float64 [n_pixels, 4] array, in place."""
    px, py = np.ascontiguousarray(px, np.int32), np.ascontiguousarray(py, np.int32)
    rgb = np.ascontiguousarray(rgb, np.float32)
    weight = None if weight is None else np.ascontiguousarray(weight, np.float32)
    b = np.asarray(bounds, np.int32)
    assert pixels.dtype == np.float64 and pixels.flags.c_contiguous
    assert rgb.shape[0] == len(px) * n_passes
    lib().orc_film_add_samples(_p(pixels), _p(b), ctypes.c_float(max_component), _p(px), _p(py), _p(rgb),
                               ctypes.c_int(rgb.shape[1]), _p(weight), ctypes.c_int(len(px)),
                               ctypes.c_int(n_passes))
    return pixels


def kd_closest(nodes, prim_indices, prims, verts, bounds, rays, nthreads=1):
    from nn_bvh_amd._lib import HIT_DTYPE
    nodes, prims = np.ascontiguousarray(nodes), np.ascontiguousarray(prims)
    idx = np.ascontiguousarray(prim_indices, np.int32)
    verts, bounds = np.ascontiguousarray(verts, np.float32), np.ascontiguousarray(bounds, np.float32)
    rays = np.ascontiguousarray(rays)
    hits = np.zeros(len(rays), HIT_DTYPE)
    lib().orc_kd_intersect_closest(_p(nodes), _p(idx), _p(prims), _p(verts), _p(bounds), _p(rays),
                                   ctypes.c_int64(len(rays)), _p(hits), ctypes.c_int(nthreads))
    return hits


def kd_any_hit(nodes, prim_indices, prims, verts, bounds, rays, nthreads=1):
    nodes, prims = np.ascontiguousarray(nodes), np.ascontiguousarray(prims)
    idx = np.ascontiguousarray(prim_indices, np.int32)
    verts, bounds = np.ascontiguousarray(verts, np.float32), np.ascontiguousarray(bounds, np.float32)
    rays = np.ascontiguousarray(rays)
    occ = np.zeros(len(rays), np.uint8)
    vis = np.zeros(len(rays), np.int32)
    tst = np.zeros(len(rays), np.int32)
    lib().orc_kd_intersect_any(_p(nodes), _p(idx), _p(prims), _p(verts), _p(bounds), _p(rays),
                               ctypes.c_int64(len(rays)), _p(occ), _p(vis), _p(tst), ctypes.c_int(nthreads))
    return occ, vis, tst


def bounds_t0t1(bounds6, o3, d3, tmax):
    n = len(tmax)
    hit = np.zeros(n, np.uint8)
    out = np.zeros((n, 2), np.float32)
    b, o, d, t = (np.ascontiguousarray(a, np.float32) for a in (bounds6, o3, d3, tmax))  # keep alive
    lib().orc_bounds_t0t1_batch(_p(b), _p(o), _p(d), _p(t), ctypes.c_int(n), _p(hit), _p(out))
    return hit, out


def hash_batch(in6):
    in6 = np.ascontiguousarray(in6, np.float32)
    n = len(in6)
    lo, hi, hf = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.float32)
    lib().orc_hash_batch(_p(in6), ctypes.c_int(n), _p(lo), _p(hi), _p(hf))
    return lo, hi, hf


def offset_batch(in12):
    in12 = np.ascontiguousarray(in12, np.float32)
    out = np.zeros((len(in12), 9), np.float32)
    lib().orc_offset_batch(_p(in12), ctypes.c_int(len(in12)), _p(out))
    return out


def wrs_batch(in7):
    in7 = np.ascontiguousarray(in7, np.float32)
    sel = np.zeros(len(in7), np.int32)
    out = np.zeros((len(in7), 2), np.float32)
    lib().orc_wrs_batch(_p(in7), ctypes.c_int(len(in7)), _p(sel), _p(out))
    return sel, out


def _inst(instances):
    return np.ascontiguousarray(instances, INSTANCE_DTYPE)


ANIM_DTYPE = np.dtype([("start_m", "<f4", 16), ("start_minv", "<f4", 16), ("end_m", "<f4", 16),
                       ("end_minv", "<f4", 16), ("T", "<f4", (2, 3)), ("R", "<f4", (2, 4)), ("S", "<f4", (2, 16)),
                       ("start_time", "<f4"), ("end_time", "<f4"), ("actually_animated", "<i4"), ("pad", "<i4")])


def anim_interpolate(anims, times):
    """AnimatedTransform::Interpolate per record: float32 [n, 32] = m (16), mInv (16)."""
    a = np.ascontiguousarray(anims, ANIM_DTYPE)
    t = np.ascontiguousarray(times, np.float32)
    out = np.zeros((len(a), 32), np.float32)
    lib().orc_anim_interpolate_batch(_p(a), _p(t), ctypes.c_int(len(a)), _p(out))
    return out


def closest_anim(nodes, prims, verts, instances, anims, rays, nthreads=1):
    from nn_bvh_amd._lib import HIT_DTYPE
    nodes, prims = np.ascontiguousarray(nodes), np.ascontiguousarray(prims)
    verts = np.ascontiguousarray(verts, np.float32)
    inst = _inst(instances)
    an = np.ascontiguousarray(anims, ANIM_DTYPE)
    rays = np.ascontiguousarray(rays)
    hits = np.zeros(len(rays), HIT_DTYPE)
    lib().orc_intersect_closest_anim(_p(nodes), _p(prims), _p(verts), _p(inst), _p(an), _p(rays),
                                     ctypes.c_int64(len(rays)), _p(hits), ctypes.c_int(nthreads))
    return hits


def any_hit_anim(nodes, prims, verts, instances, anims, rays, nthreads=1):
    nodes, prims = np.ascontiguousarray(nodes), np.ascontiguousarray(prims)
    verts = np.ascontiguousarray(verts, np.float32)
    inst = _inst(instances)
    an = np.ascontiguousarray(anims, ANIM_DTYPE)
    rays = np.ascontiguousarray(rays)
    occ = np.zeros(len(rays), np.uint8)
    vis = np.zeros(len(rays), np.int32)
    tst = np.zeros(len(rays), np.int32)
    lib().orc_intersect_any_anim(_p(nodes), _p(prims), _p(verts), _p(inst), _p(an), _p(rays),
                                 ctypes.c_int64(len(rays)), _p(occ), _p(vis), _p(tst), ctypes.c_int(nthreads))
    return occ, vis, tst


def set_sin_mode(mode):
    """0: Slerp's per-ray sines with libm's sinf (the reference); 1: fp64 sine rounded once (the device)."""
    lib().orc_set_sin_mode(ctypes.c_int(mode))


_normals_keepalive = None
_alpha_keepalive = None


_uvs_keepalive = None


def set_vertex_uvs(uvs):
    """(u, v) per vertex for prim kinds 12 .. 15 (alpha-tested patches of meshes with uv); None = none."""
    global _uvs_keepalive
    _uvs_keepalive = None if uvs is None else np.ascontiguousarray(uvs, np.float32)
    lib().orc_set_vertex_uvs(None if uvs is None else _p(_uvs_keepalive))


def set_prim_alpha(alpha):
    """Constant alpha per primitive for prim kinds 8 .. 11 (alpha-tested bilinear patches), indexed like the prims
    array given to closest() / any_hit(); None = none."""
    global _alpha_keepalive
    _alpha_keepalive = None if alpha is None else np.ascontiguousarray(alpha, np.float32)
    lib().orc_set_prim_alpha(None if alpha is None else _p(_alpha_keepalive))


def set_vertex_normals(normals):
    """Per-vertex shading normals for prim kinds 6 / 7 (alpha-tested triangles of smooth meshes); None = none."""
    global _normals_keepalive
    _normals_keepalive = None if normals is None else np.ascontiguousarray(normals, np.float32)
    lib().orc_set_vertex_normals(None if normals is None else _p(_normals_keepalive))
