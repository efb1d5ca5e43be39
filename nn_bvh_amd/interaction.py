"""Host-side mirror of the hit -> SurfaceInteraction post-pass (Triangle:: and
BilinearPatch::InteractionFromIntersection, /root/reference/src/pbrt/shapes.h:884-1010, 1396-1489) over
include/nnbvh.h's nnbvh_shading_mesh_* / nnbvh_triangle_interactions_device."""
import numpy as np

from . import _lib
from ._lib import INTERACTION_DTYPE, NNBVHError, check, ptr


class ShadingMesh:
    """TriangleMesh data (util/mesh.h:24-72) of a whole scene on the device: positions, per-triangle
    vertex indices (triangle k = the primitive with id k) and the optional n / uv / s / faceIndices
    arrays, in render space and as the TriangleMesh constructor stores them."""

    def __init__(self, verts, tri_vertices, normals=None, uvs=None, tangents=None, face_indices=None,
                 tri_flags=None, device=0, patch_vertices=None):
        f32 = lambda a, w: None if a is None else np.ascontiguousarray(a, np.float32).reshape(-1, w)  # noqa: E731
        self.verts = f32(verts, 3)
        self.tri_vertices = np.ascontiguousarray(tri_vertices, np.int32).reshape(-1, 3)
        self.patch_vertices = None
        if patch_vertices is not None:  # 4 per primitive (p00 p10 p01 p11), v[0] < 0: not a patch
            self.patch_vertices = np.ascontiguousarray(patch_vertices, np.int32).reshape(-1, 4)
            assert len(self.patch_vertices) == len(self.tri_vertices)
        self.normals, self.uvs, self.tangents = f32(normals, 3), f32(uvs, 2), f32(tangents, 3)
        self.face_indices = None if face_indices is None else np.ascontiguousarray(face_indices, np.int32)
        self.tri_flags = None if tri_flags is None else np.ascontiguousarray(tri_flags, np.uint8)
        self.device = int(device)
        opt = lambda a: ptr(a) if a is not None else None  # noqa: E731
        self._h = _lib.lib().nnbvh_shading_mesh_create(
            ptr(self.verts), len(self.verts), ptr(self.tri_vertices), opt(self.patch_vertices),
            len(self.tri_vertices),
            opt(self.normals), opt(self.uvs), opt(self.tangents), opt(self.face_indices),
            opt(self.tri_flags), self.device)
        if not self._h:
            raise NNBVHError("nnbvh_shading_mesh_create: " + _lib.last_error())

    def set_instances(self, instances, animated=None):
        """The scene's instance table (INSTANCE_DTYPE): hits inside instances are then finished on the
        device (TransformedPrimitive::Intersect's transform of the interaction).  animated: the
        ANIMATED_DTYPE table of the same scene — hits inside AnimatedPrimitives then use
        renderFromPrimitive.Interpolate(ray.time) (cpu/primitive.cpp:143-153)."""
        inst = np.ascontiguousarray(instances, _lib.INSTANCE_DTYPE)
        if animated is None:
            check(_lib.lib().nnbvh_shading_mesh_set_instances(self._h, ptr(inst), len(inst)),
                  "nnbvh_shading_mesh_set_instances")
            return
        anim = np.ascontiguousarray(animated, _lib.ANIMATED_DTYPE)
        assert len(anim) == len(inst)
        check(_lib.lib().nnbvh_shading_mesh_set_instances_animated(self._h, ptr(inst), ptr(anim), len(inst)),
              "nnbvh_shading_mesh_set_instances_animated")

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().nnbvh_shading_mesh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def interactions_device(self, d_hits, max_items, d_out, d_rays=None, ray_queue=None, d_size=None,
                            stream=0):
        """d_* are device pointers (ints); ray_queue is a nn_bvh_amd.wavefront.RayQueue (SOA)."""
        soa = ray_queue._wire() if ray_queue is not None else None
        check(_lib.lib().nnbvh_triangle_interactions_device(
            self._h, d_rays, ptr(soa) if soa is not None else None, d_hits, int(max_items), d_size,
            d_out, stream), "nnbvh_triangle_interactions_device")

    def interactions(self, rays, hits):
        """Host arrays in (RAY_DTYPE, HIT_DTYPE), INTERACTION_DTYPE records out."""
        rays = np.ascontiguousarray(rays, _lib.RAY_DTYPE)
        hits = np.ascontiguousarray(hits, _lib.HIT_DTYPE)
        out = np.zeros(len(hits), INTERACTION_DTYPE)
        check(_lib.lib().nnbvh_triangle_interactions(self._h, ptr(rays), ptr(hits), len(hits), ptr(out)),
              "nnbvh_triangle_interactions")
        return out
