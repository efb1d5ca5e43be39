"""ctypes binding of libnnbvh_hip.so (include/nnbvh.h).  Loading fails loudly: there is no
Python or CPU fallback for the traversal path."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NNBVH_LIB selects a diagnostic build (e.g. libnnbvh_hip_stats.so); default is the product
LIB_PATH = os.path.join(_HERE, os.environ.get("NNBVH_LIB", "libnnbvh_hip.so"))

# numpy views of the wire structs (byte-identical to include/nnbvh.h)
NODE_DTYPE = np.dtype([("pmin", "<f4", 3), ("pmax", "<f4", 3), ("offset", "<i4"),
                       ("nprims", "<u2"), ("axis", "u1"), ("pad", "u1")])
PRIM_DTYPE = np.dtype([("kind", "<i4"), ("id", "<i4"), ("v", "<i4", 4)])
RAY_DTYPE = np.dtype([("o", "<f4", 3), ("tmax", "<f4"), ("d", "<f4", 3), ("time", "<f4")])
INSTANCE_DTYPE = np.dtype([("render_from_prim", "<f4", 12), ("prim_from_render", "<f4", 12),
                           ("root", "<i4"), ("n_nodes", "<i4")])
ANIMATED_DTYPE = np.dtype([("start_from", "<f4", 16), ("start_inv", "<f4", 16), ("end_from", "<f4", 16),
                           ("end_inv", "<f4", 16), ("T", "<f4", (2, 3)), ("R", "<f4", (2, 4)), ("S", "<f4", (2, 16)),
                           ("start_time", "<f4"), ("end_time", "<f4"), ("actually_animated", "<i4"), ("pad", "<i4")])
BATCH_DTYPE = np.dtype([("kind", "<i4"), ("pad", "<i4"), ("d_rays", "<u8"), ("n", "<i8"),
                        ("d_out", "<u8"), ("d_nodes_visited", "<u8"), ("d_prim_tests", "<u8")])
HIT_DTYPE = np.dtype([("prim", "<i4"), ("t", "<f4"), ("b0", "<f4"), ("b1", "<f4"), ("b2", "<f4"),
                      ("nodes_visited", "<i4"), ("prim_tests", "<i4"), ("instance", "<i4")])
KD_NODE_DTYPE = np.dtype([("split_or_index", "<u4"), ("flags", "<u4")])
RAY_SOA_DTYPE = np.dtype([(k, "<u8") for k in ("ox", "oy", "oz", "dx", "dy", "dz", "time", "tmax",
                                                "has_medium")])
WORK_QUEUE_DTYPE = np.dtype([("items", "<u8"), ("size", "<u8"), ("capacity", "<i4"), ("pad", "<i4")])
CLOSEST_QUEUES = ("escaped", "hit_area_light", "basic_eval_material", "universal_eval_material",
                  "medium_sample", "next_ray")
CLOSEST_QUEUES_DTYPE = np.dtype([(k, WORK_QUEUE_DTYPE) for k in CLOSEST_QUEUES])
CLASS_BASIC, CLASS_UNIVERSAL, CLASS_INTERFACE, CLASS_AREA_LIGHT = 0, 1, 2, 4
INTERACTION_DTYPE = np.dtype([("pi_lo", "<f4", 3), ("pi_hi", "<f4", 3), ("uv", "<f4", 2), ("wo", "<f4", 3),
                              ("time", "<f4"), ("n", "<f4", 3), ("face_index", "<i4"), ("dpdu", "<f4", 3),
                              ("dpdv", "<f4", 3), ("ns", "<f4", 3), ("dpdus", "<f4", 3), ("dpdvs", "<f4", 3),
                              ("dndus", "<f4", 3), ("dndvs", "<f4", 3), ("dndu", "<f4", 3), ("dndv", "<f4", 3),
                              ("pad0", "<f4"), ("prim", "<i4"), ("status", "<i4"), ("pad1", "<i4", 2)])
TRI_FLIP_NORMAL, TRI_HAS_UV, TRI_HAS_N, TRI_HAS_S = 1, 2, 4, 8
assert INTERACTION_DTYPE.itemsize == 192
assert NODE_DTYPE.itemsize == 32 and PRIM_DTYPE.itemsize == 24
assert RAY_SOA_DTYPE.itemsize == 72 and CLOSEST_QUEUES_DTYPE.itemsize == 6 * 24
assert RAY_DTYPE.itemsize == 32 and HIT_DTYPE.itemsize == 32

EXPORTS = [
    "nnbvh_last_error", "nnbvh_device_count", "nnbvh_build_create", "nnbvh_build_nodes",
    "nnbvh_build_ordered_prims", "nnbvh_build_depth", "nnbvh_build_destroy",
    "nnbvh_scene_create", "nnbvh_scene_create_with_normals", "nnbvh_scene_create_with_attributes", "nnbvh_scene_create_gpu_build_with_attributes", "nnbvh_scene_create_instanced_with_attributes", "nnbvh_scene_destroy", "nnbvh_scene_bounds", "nnbvh_scene_info",
    "nnbvh_intersect_closest", "nnbvh_intersect_any", "nnbvh_intersect_closest_device",
    "nnbvh_intersect_any_device", "nnbvh_scene_set_option", "nnbvh_scene_sched_stats",
    "nnbvh_trace_batches_device", "nnbvh_scene_create_instanced", "nnbvh_transform_bounds",
    "nnbvh_build_create_with_bounds", "nnbvh_wavefront_intersect_closest",
    "nnbvh_wavefront_intersect_shadow", "nnbvh_wavefront_intersect_closest_and_shadow", "nnbvh_build_create_gpu", "nnbvh_build_gpu_timing",
    "nnbvh_shading_mesh_create", "nnbvh_shading_mesh_destroy", "nnbvh_triangle_interactions_device",
    "nnbvh_triangle_interactions", "nnbvh_scene_create_gpu_build",
    "nnbvh_shading_mesh_set_instances", "nnbvh_shading_mesh_set_instances_animated", "nnbvh_host_register",
    "nnbvh_host_unregister", "nnbvh_wavefront_record_shadow_device",
    "nnbvh_film_create", "nnbvh_film_destroy", "nnbvh_film_clear", "nnbvh_film_add_samples_device",
    "nnbvh_film_pixels_device", "nnbvh_film_read", "nnbvh_film_pack_pixels_device",
    "nnbvh_film_unpack_pixels_device", "nnbvh_kd_build_create", "nnbvh_kd_build_create_gpu",
    "nnbvh_kd_build_create_stable", "nnbvh_kd_build_timing", "nnbvh_kd_build_nodes",
    "nnbvh_kd_build_prim_indices", "nnbvh_kd_build_bounds", "nnbvh_kd_build_depth", "nnbvh_kd_build_destroy",
    "nnbvh_kd_scene_create", "nnbvh_kd_scene_create_with_attributes", "nnbvh_kd_scene_destroy", "nnbvh_kd_intersect_closest", "nnbvh_kd_intersect_any",
    "nnbvh_kd_intersect_closest_device", "nnbvh_kd_intersect_any_device",
    "nnbvh_wavefront_intersect_shadow_tr", "nnbvh_wavefront_intersect_one_random",
    "nnbvh_scene_create_instanced_animated",
]

_lib = None


class NNBVHError(RuntimeError):
    pass


def lib():
    """The loaded shared library (built by nn_bvh_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NNBVHError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the traversal path has no fallback)")
    # A process must hold ONE HIP runtime.  PyTorch (used for device buffers by the callers of the
    # *_device entry points) ships its own libamdhip64; if this library pulled in the system copy
    # first, torch.cuda would later find no device.  Loading torch first makes both share torch's.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    L.nnbvh_last_error.restype = ctypes.c_char_p
    L.nnbvh_device_count.restype = i32
    L.nnbvh_build_create.restype = vp
    L.nnbvh_build_create.argtypes = [vp, i32, vp, i32, i32, i32]
    L.nnbvh_build_nodes.restype = vp
    L.nnbvh_build_nodes.argtypes = [vp, ctypes.POINTER(i32)]
    L.nnbvh_build_ordered_prims.restype = vp
    L.nnbvh_build_ordered_prims.argtypes = [vp, ctypes.POINTER(i32)]
    L.nnbvh_build_depth.restype = i32
    L.nnbvh_build_depth.argtypes = [vp]
    L.nnbvh_build_destroy.restype = None
    L.nnbvh_build_destroy.argtypes = [vp]
    L.nnbvh_scene_create.restype = vp
    L.nnbvh_scene_create.argtypes = [vp, i32, vp, i32, vp, i32, i32]
    L.nnbvh_scene_create_with_normals.restype = vp
    L.nnbvh_scene_create_with_normals.argtypes = [vp, i32, vp, i32, vp, vp, i32, i32]
    L.nnbvh_scene_create_instanced_with_attributes.restype = vp
    L.nnbvh_scene_create_instanced_with_attributes.argtypes = [vp, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, i32]
    L.nnbvh_scene_create_with_attributes.restype = vp
    L.nnbvh_scene_create_with_attributes.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32]
    L.nnbvh_scene_destroy.restype = None
    L.nnbvh_scene_destroy.argtypes = [vp]
    L.nnbvh_scene_bounds.restype = i32
    L.nnbvh_scene_bounds.argtypes = [vp, vp]
    L.nnbvh_scene_info.restype = i32
    L.nnbvh_scene_info.argtypes = [vp, vp]
    L.nnbvh_intersect_closest.restype = i32
    L.nnbvh_intersect_closest.argtypes = [vp, vp, i64, vp]
    L.nnbvh_intersect_any.restype = i32
    L.nnbvh_intersect_any.argtypes = [vp, vp, i64, vp, vp, vp]
    L.nnbvh_intersect_closest_device.restype = i32
    L.nnbvh_intersect_closest_device.argtypes = [vp, vp, i64, vp, vp]
    L.nnbvh_intersect_any_device.restype = i32
    L.nnbvh_intersect_any_device.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    L.nnbvh_scene_set_option.restype = i32
    L.nnbvh_scene_set_option.argtypes = [vp, ctypes.c_char_p, i32]
    L.nnbvh_scene_create_instanced.restype = vp
    L.nnbvh_scene_create_instanced.argtypes = [vp, i32, i32, vp, i32, vp, i32, vp, i32, i32]
    L.nnbvh_scene_create_instanced_animated.restype = vp
    L.nnbvh_scene_create_instanced_animated.argtypes = [vp, i32, i32, vp, i32, vp, i32, vp, i32, vp, i32]
    L.nnbvh_transform_bounds.restype = None
    L.nnbvh_transform_bounds.argtypes = [vp, vp, vp]
    L.nnbvh_build_create_with_bounds.restype = vp
    L.nnbvh_build_create_with_bounds.argtypes = [vp, i32, vp, i32, vp, i32, i32]
    L.nnbvh_trace_batches_device.restype = i32
    L.nnbvh_trace_batches_device.argtypes = [vp, vp, i32, vp]
    L.nnbvh_scene_create_gpu_build.restype = vp
    L.nnbvh_scene_create_gpu_build.argtypes = [vp, i32, vp, i32, vp, i32, i32, i32]
    L.nnbvh_scene_create_gpu_build_with_attributes.restype = vp
    L.nnbvh_scene_create_gpu_build_with_attributes.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32]
    L.nnbvh_build_create_gpu.restype = vp
    L.nnbvh_build_create_gpu.argtypes = [vp, i32, vp, i32, vp, i32, i32, i32]
    L.nnbvh_build_gpu_timing.restype = i32
    L.nnbvh_build_gpu_timing.argtypes = [vp, vp]
    L.nnbvh_shading_mesh_create.restype = vp
    L.nnbvh_shading_mesh_create.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, i32]
    L.nnbvh_shading_mesh_set_instances.restype = i32
    L.nnbvh_shading_mesh_set_instances.argtypes = [vp, vp, i32]
    L.nnbvh_host_register.restype = i32
    L.nnbvh_host_register.argtypes = [vp, ctypes.c_size_t]
    L.nnbvh_host_unregister.restype = i32
    L.nnbvh_host_unregister.argtypes = [vp]
    L.nnbvh_shading_mesh_set_instances_animated.restype = i32
    L.nnbvh_shading_mesh_set_instances_animated.argtypes = [vp, vp, vp, i32]
    L.nnbvh_shading_mesh_destroy.restype = None
    L.nnbvh_shading_mesh_destroy.argtypes = [vp]
    L.nnbvh_triangle_interactions.restype = i32
    L.nnbvh_triangle_interactions.argtypes = [vp, vp, vp, i32, vp]
    L.nnbvh_triangle_interactions_device.restype = i32
    L.nnbvh_triangle_interactions_device.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
    L.nnbvh_wavefront_intersect_closest.restype = i32
    L.nnbvh_wavefront_intersect_closest.argtypes = [vp, i32, vp, vp, vp, i64, vp, vp, vp]
    L.nnbvh_wavefront_intersect_shadow.restype = i32
    L.nnbvh_wavefront_intersect_shadow.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp]
    L.nnbvh_wavefront_intersect_closest_and_shadow.restype = i32
    L.nnbvh_wavefront_intersect_closest_and_shadow.argtypes = [vp, i32, vp, vp, vp, i64, vp, vp,
                                                               i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp]
    L.nnbvh_wavefront_record_shadow_device.restype = i32
    L.nnbvh_wavefront_record_shadow_device.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i64, i32, vp]
    L.nnbvh_film_create.restype = vp
    L.nnbvh_film_create.argtypes = [i32, i32, i32, i32, ctypes.c_float, i32]
    L.nnbvh_film_destroy.restype = None
    L.nnbvh_film_destroy.argtypes = [vp]
    L.nnbvh_film_clear.restype = i32
    L.nnbvh_film_clear.argtypes = [vp, vp]
    L.nnbvh_film_add_samples_device.restype = i32
    L.nnbvh_film_add_samples_device.argtypes = [vp, vp, vp, vp, i32, vp, i32, i32, vp, vp]
    L.nnbvh_film_pixels_device.restype = i32
    L.nnbvh_film_pixels_device.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(i64)]
    L.nnbvh_film_read.restype = i32
    L.nnbvh_film_read.argtypes = [vp, vp]
    L.nnbvh_film_pack_pixels_device.restype = i32
    L.nnbvh_film_pack_pixels_device.argtypes = [vp, vp, i64, vp, vp]
    L.nnbvh_film_unpack_pixels_device.restype = i32
    L.nnbvh_film_unpack_pixels_device.argtypes = [vp, vp, i64, vp, vp]
    L.nnbvh_kd_build_create.restype = vp
    L.nnbvh_kd_build_create.argtypes = [vp, i32, vp, i32, vp, i32, i32, ctypes.c_float, i32, i32]
    L.nnbvh_kd_build_create_stable.restype = vp
    L.nnbvh_kd_build_create_stable.argtypes = [vp, i32, vp, i32, vp, i32, i32, ctypes.c_float, i32, i32]
    L.nnbvh_kd_build_create_gpu.restype = vp
    L.nnbvh_kd_build_create_gpu.argtypes = [vp, i32, vp, i32, vp, i32, i32, ctypes.c_float, i32, i32, i32]
    L.nnbvh_kd_build_timing.argtypes = [vp, vp]
    L.nnbvh_kd_build_nodes.restype = vp
    L.nnbvh_kd_build_nodes.argtypes = [vp, ctypes.POINTER(i32)]
    L.nnbvh_kd_build_prim_indices.restype = vp
    L.nnbvh_kd_build_prim_indices.argtypes = [vp, ctypes.POINTER(i32)]
    L.nnbvh_kd_build_bounds.restype = i32
    L.nnbvh_kd_build_bounds.argtypes = [vp, vp]
    L.nnbvh_kd_build_depth.restype = i32
    L.nnbvh_kd_build_depth.argtypes = [vp]
    L.nnbvh_kd_build_destroy.restype = None
    L.nnbvh_kd_build_destroy.argtypes = [vp]
    L.nnbvh_kd_scene_create.restype = vp
    L.nnbvh_kd_scene_create.argtypes = [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32]
    L.nnbvh_kd_scene_create_with_attributes.restype = vp
    L.nnbvh_kd_scene_create_with_attributes.argtypes = [vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, i32]
    L.nnbvh_kd_scene_destroy.restype = None
    L.nnbvh_kd_scene_destroy.argtypes = [vp]
    L.nnbvh_kd_intersect_closest.restype = i32
    L.nnbvh_kd_intersect_closest.argtypes = [vp, vp, i64, vp]
    L.nnbvh_kd_intersect_any.restype = i32
    L.nnbvh_kd_intersect_any.argtypes = [vp, vp, i64, vp, vp, vp]
    L.nnbvh_kd_intersect_closest_device.restype = i32
    L.nnbvh_kd_intersect_closest_device.argtypes = [vp, vp, i64, vp, vp]
    L.nnbvh_kd_intersect_any_device.restype = i32
    L.nnbvh_kd_intersect_any_device.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    L.nnbvh_wavefront_intersect_shadow_tr.restype = i32
    L.nnbvh_wavefront_intersect_shadow_tr.argtypes = [vp, vp, i32, vp, vp, vp, i64, vp, vp, vp, vp, vp, i64, vp, vp]
    L.nnbvh_wavefront_intersect_one_random.restype = i32
    L.nnbvh_wavefront_intersect_one_random.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp]
    L.nnbvh_scene_sched_stats.restype = i32
    L.nnbvh_scene_sched_stats.argtypes = [vp, vp, i32]
    _lib = L
    return L


def last_error():
    return lib().nnbvh_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise NNBVHError(f"{what} failed (status {rc}): {last_error()}")


def ptr(a):
    return ctypes.c_void_p(a.ctypes.data)
