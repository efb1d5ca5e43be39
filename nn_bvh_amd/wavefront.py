"""Host-side mirror of the reference's batched caller interface, WavefrontAggregate
(/root/reference/src/pbrt/wavefront/integrator.h:32-54; CPU implementation
wavefront/aggregate.cpp:34-68), over include/nnbvh.h's nnbvh_wavefront_* entry points.

Queues are device-resident (torch tensors are buffers + the current stream only).  A ray queue is
the reference's SOA<Ray> (workitems.soa:40-50): six float arrays plus the queue's device-side
size; the output queues hold indices into the input queue, pushed by the device with the
reference's rules (wavefront/intersect.h:16-156).  IntersectShadowTr / IntersectOneRandom come in
their media-free form (see include/nnbvh.h)."""
import numpy as np
import torch

from . import _lib
from ._lib import CLOSEST_QUEUES, check, ptr


class WorkQueue:
    """WorkQueue<T> (wavefront/workqueue.h:36-113) of int32 work-item indices on the device."""

    def __init__(self, capacity, device):
        self.capacity = int(capacity)
        self.items = torch.empty(max(self.capacity, 1), dtype=torch.int32, device=device)
        self.size = torch.zeros(1, dtype=torch.int32, device=device)

    def Reset(self):
        self.size.zero_()

    def Size(self):
        """Host read of the device counter (synchronises the current stream)."""
        return int(self.size.item())

    def indices(self):
        return self.items[:min(self.Size(), self.capacity)]

    def _wire(self, rec):
        rec["items"], rec["size"], rec["capacity"] = self.items.data_ptr(), self.size.data_ptr(), self.capacity


class RayQueue:
    """SOA ray queue: RayQueue (tmax None, has_medium optional) or ShadowRayQueue (tmax set)."""

    def __init__(self, o, d, tmax=None, time=None, has_medium=None, size=None):
        """o, d: float32 device tensors [3, capacity] (SOA: row k is the k-th coordinate array)."""
        self.o = o.contiguous()
        self.d = d.contiguous()
        assert self.o.dtype == torch.float32 and self.o.shape == self.d.shape and self.o.shape[0] == 3
        self.capacity = int(self.o.shape[1])
        self.tmax, self.time, self.has_medium = tmax, time, has_medium
        dev = self.o.device
        self.size = size if size is not None else torch.full((1,), self.capacity, dtype=torch.int32,
                                                             device=dev)

    @classmethod
    def from_records(cls, rays, device, shadow=False):
        """From a host RAY_DTYPE batch (the AoS wire format of the plain entry points)."""
        o = torch.from_numpy(np.ascontiguousarray(rays["o"].T)).to(device)
        d = torch.from_numpy(np.ascontiguousarray(rays["d"].T)).to(device)
        tmax = torch.from_numpy(np.ascontiguousarray(rays["tmax"])).to(device) if shadow else None
        return cls(o, d, tmax=tmax)

    def _wire(self):
        rec = np.zeros(1, _lib.RAY_SOA_DTYPE)
        for k, name in enumerate("xyz"):
            rec["o" + name] = self.o[k].data_ptr()
            rec["d" + name] = self.d[k].data_ptr()
        for key in ("time", "tmax", "has_medium"):
            t = getattr(self, key)
            rec[key] = t.data_ptr() if t is not None else 0
        return rec


class WavefrontAggregate:
    """IntersectClosest / IntersectShadow of wavefront/integrator.h:32-54 on one BVHAggregate.

    prim_class: optional uint8 per primitive id (nn_bvh_amd._lib.CLASS_*), what the reference
    reads off the hit's SurfaceInteraction when it decides the destination queues."""

    def __init__(self, aggregate, prim_class=None):
        self.aggregate = aggregate
        self.device = torch.device("cuda", aggregate.device)
        self.prim_class = None
        if prim_class is not None:
            self.prim_class = torch.from_numpy(np.ascontiguousarray(prim_class, np.uint8)).to(self.device)

    def Bounds(self):
        return self.aggregate.Bounds()

    def IntersectClosest(self, max_rays, ray_queue, escaped=None, hit_area_light=None,
                         basic_eval_material=None, universal_eval_material=None, medium_sample=None,
                         next_ray=None, hits=None):
        """Traces ray_queue[0 : min(max_rays, size)] and pushes each item's index to the queues
        the reference would (a None queue drops its pushes).  Returns the device hit records
        (HIT_DTYPE rows as a uint8 tensor [max_rays, 32]); asynchronous on the current stream."""
        if hits is None:
            hits = torch.empty((max(int(max_rays), 1), 32), dtype=torch.uint8, device=self.device)
        qrec = np.zeros(1, _lib.CLOSEST_QUEUES_DTYPE)
        given = dict(zip(CLOSEST_QUEUES, (escaped, hit_area_light, basic_eval_material,
                                          universal_eval_material, medium_sample, next_ray)))
        for name, q in given.items():
            if q is not None:
                q._wire(qrec[name][0:1])
        soa = ray_queue._wire()
        pc = self.prim_class
        check(_lib.lib().nnbvh_wavefront_intersect_closest(
            self.aggregate._h, int(max_rays), ptr(soa), ray_queue.size.data_ptr(),
            pc.data_ptr() if pc is not None else None, 0 if pc is None else pc.numel(),
            hits.data_ptr(), ptr(qrec), torch.cuda.current_stream(self.device).cuda_stream),
            "nnbvh_wavefront_intersect_closest")
        return hits

    def IntersectShadow(self, max_rays, shadow_queue, Ld, r_u, r_l, pixel_index, L, occluded=None):
        """Traces the shadow queue and adds Ld / (r_u + r_l).Average() to L[pixel_index] for the
        unoccluded rays (RecordShadowRayResult, wavefront/intersect.h:32-47).  Ld, r_u, r_l:
        float32 [capacity, 4]; L: float32 [n_pixels, 4]; all device tensors."""
        assert shadow_queue.tmax is not None, "a shadow queue carries tMax per item"
        for t in (Ld, r_u, r_l, L):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[-1] == 4
        assert pixel_index.dtype == torch.int32 and pixel_index.is_contiguous()
        soa = shadow_queue._wire()
        check(_lib.lib().nnbvh_wavefront_intersect_shadow(
            self.aggregate._h, int(max_rays), ptr(soa), shadow_queue.size.data_ptr(), Ld.data_ptr(),
            r_u.data_ptr(), r_l.data_ptr(), pixel_index.data_ptr(), L.data_ptr(), L.shape[0],
            occluded.data_ptr() if occluded is not None else None,
            torch.cuda.current_stream(self.device).cuda_stream),
            "nnbvh_wavefront_intersect_shadow")

    def IntersectClosestAndShadow(self, max_rays, ray_queue, max_shadow_rays, shadow_queue, Ld, r_u, r_l, pixel_index,
                                  L, escaped=None, hit_area_light=None, basic_eval_material=None,
                                  universal_eval_material=None, medium_sample=None, next_ray=None, hits=None,
                                  occluded=None):
        """IntersectShadow(max_shadow_rays, shadow_queue, ...) of one depth and IntersectClosest(max_rays, ray_queue,
        ...) of the next in ONE launch of the traversal kernel (both queues come out of the same shading pass and
        neither reads what the other writes: wavefront/integrator.cpp's render loop).  Same results as the two calls.
        Returns the hit records."""
        assert shadow_queue.tmax is not None, "a shadow queue carries tMax per item"
        for t in (Ld, r_u, r_l, L):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[-1] == 4
        assert pixel_index.dtype == torch.int32 and pixel_index.is_contiguous()
        if hits is None:
            hits = torch.empty((max(int(max_rays), 1), 32), dtype=torch.uint8, device=self.device)
        qrec = np.zeros(1, _lib.CLOSEST_QUEUES_DTYPE)
        given = dict(zip(CLOSEST_QUEUES, (escaped, hit_area_light, basic_eval_material,
                                          universal_eval_material, medium_sample, next_ray)))
        for name, q in given.items():
            if q is not None:
                q._wire(qrec[name][0:1])
        soa, ssoa = ray_queue._wire(), shadow_queue._wire()
        pc = self.prim_class
        check(_lib.lib().nnbvh_wavefront_intersect_closest_and_shadow(
            self.aggregate._h, int(max_rays), ptr(soa), ray_queue.size.data_ptr(),
            pc.data_ptr() if pc is not None else None, 0 if pc is None else pc.numel(), hits.data_ptr(), ptr(qrec),
            int(max_shadow_rays), ptr(ssoa), shadow_queue.size.data_ptr(), Ld.data_ptr(), r_u.data_ptr(),
            r_l.data_ptr(), pixel_index.data_ptr(), L.data_ptr(), L.shape[0],
            occluded.data_ptr() if occluded is not None else None,
            torch.cuda.current_stream(self.device).cuda_stream), "nnbvh_wavefront_intersect_closest_and_shadow")
        return hits

    def IntersectShadowTr(self, max_rays, shadow_queue, shading_mesh, Ld, r_u, r_l, pixel_index, L, state=None):
        """IntersectShadowTr (wavefront/aggregate.cpp:70-88, TraceTransmittance of wavefront/intersect.h:
        164-274) without media: shadow rays pass through interface surfaces (CLASS_INTERFACE) and are
        blocked by the first surface with a material; arriving rays add Ld * (1 / (r_u + r_l).Average())
        to L[pixel_index].  state: optional uint8 [capacity] out (0 arrived, 1 blocked, 2 host)."""
        assert shadow_queue.tmax is not None, "a shadow queue carries tMax per item"
        for t in (Ld, r_u, r_l, L):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[-1] == 4
        assert pixel_index.dtype == torch.int32 and pixel_index.is_contiguous()
        soa = shadow_queue._wire()
        pc = self.prim_class
        check(_lib.lib().nnbvh_wavefront_intersect_shadow_tr(
            self.aggregate._h, shading_mesh._h, int(max_rays), ptr(soa), shadow_queue.size.data_ptr(),
            pc.data_ptr() if pc is not None else None, 0 if pc is None else pc.numel(), Ld.data_ptr(),
            r_u.data_ptr(), r_l.data_ptr(), pixel_index.data_ptr(), L.data_ptr(), L.shape[0],
            state.data_ptr() if state is not None else None,
            torch.cuda.current_stream(self.device).cuda_stream), "nnbvh_wavefront_intersect_shadow_tr")

    def IntersectOneRandom(self, max_items, p0, p1, material, shading_mesh, prim_material=None, size=None):
        """IntersectOneRandom (wavefront/aggregate.cpp:90-116): p0, p1 float32 [n, 3], material int32 [n]
        device tensors; prim_material int32 per primitive id.  Returns (selected hit records uint8
        [n, 32], their segment rays uint8 [n, 32], reservoir pdf float32 [n], weight sum float32 [n])."""
        for t in (p0, p1):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[-1] == 3
        assert material.dtype == torch.int32 and material.is_contiguous()
        n = int(max_items)
        sel_hits = torch.empty((max(n, 1), 32), dtype=torch.uint8, device=self.device)
        sel_rays = torch.empty((max(n, 1), 32), dtype=torch.uint8, device=self.device)
        pdf = torch.zeros(max(n, 1), dtype=torch.float32, device=self.device)
        wsum = torch.zeros(max(n, 1), dtype=torch.float32, device=self.device)
        pm = prim_material
        check(_lib.lib().nnbvh_wavefront_intersect_one_random(
            self.aggregate._h, shading_mesh._h, n, p0.data_ptr(), p1.data_ptr(), material.data_ptr(),
            size.data_ptr() if size is not None else None, pm.data_ptr() if pm is not None else None,
            0 if pm is None else pm.numel(), sel_hits.data_ptr(), sel_rays.data_ptr(), pdf.data_ptr(),
            wsum.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream),
            "nnbvh_wavefront_intersect_one_random")
        return sel_hits, sel_rays, pdf, wsum
