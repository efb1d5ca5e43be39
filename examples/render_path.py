#!/usr/bin/env python3
"""A small wavefront PATH TRACER on the queue interface (diffuse surfaces, next-event estimation towards the crown
scene's six light quads, up to --depth bounces), every stage device-resident.  Per pixel sample:

  depth 0   IntersectClosest(camera rays)
  depth d   shading of depth d's hits (a few torch ops: light sample -> shadow ray, cosine-weighted bounce ray)
            -> IntersectClosestAndShadow: the SHADOW rays of depth d and the BOUNCE rays of depth d + 1 in ONE launch
               of the traversal kernel — both queues come out of the same shading pass and neither reads what the
               other writes (wavefront/integrator.cpp's render loop issues them back to back)
  last      IntersectShadow(shadow rays of the last depth)

The trace stages are the library (HIP kernels); the torch ops in between stand in for the reference's material and
light stages (wavefront/integrator.cpp:403-579) and are plumbing.
usage: examples/render_path.py [--scene crown] [--spp 16] [--depth 3] [--scale 2] [--out render.png]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="crown")
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--depth", type=int, default=3, help="bounces after the camera ray")
    ap.add_argument("--scale", type=int, default=2, help="render every k-th pixel per axis")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "render_path.png"))
    args = ap.parse_args()
    import torch
    from render_direct import write_png
    from nn_bvh_amd import BVHAggregate, make_prims, scene
    from nn_bvh_amd.interaction import ShadingMesh
    from nn_bvh_amd.wavefront import RayQueue, WavefrontAggregate, WorkQueue

    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    verts, tris, source = scene.load_scene(args.scene)
    agg = BVHAggregate.build_on_device(make_prims(tris), verts)
    wf = WavefrontAggregate(agg)
    smesh = ShadingMesh(verts, tris)
    cam = args.scene if args.scene in scene.CAMERAS else "crown"
    xres, yres = scene.CAMERAS[cam][4] // args.scale, scene.CAMERAS[cam][5] // args.scale
    quads = torch.from_numpy(scene.CROWN_LIGHT_QUADS.astype(np.float32)).to(dev)
    quad_n = torch.linalg.cross(quads[:, 1] - quads[:, 0], quads[:, 3] - quads[:, 0])
    quad_area = quad_n.norm(dim=1)
    quad_n = quad_n / quad_area[:, None]
    eps = 1e-4 * float(np.abs(verts).max())
    albedo = 0.7
    gen = torch.Generator(device=dev).manual_seed(11)
    film = torch.zeros((xres * yres, 4), dtype=torch.float32, device=dev)
    n_traced, t_trace, launches = 0, 0.0, 0

    def timed(fn):
        nonlocal t_trace, launches
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        t_trace += time.perf_counter() - t0
        launches += 1
        return out

    def shade(rq, hits, n, material, beta):
        """hits of one depth -> (shadow queue + its Ld / slot, bounce queue + its throughput / slot)"""
        idx = material.indices().long()
        intr = torch.empty((n, 48), dtype=torch.float32, device=dev)
        smesh.interactions_device(hits.data_ptr(), n, intr.data_ptr(), ray_queue=rq, stream=stream)
        rec = intr[idx]
        p = 0.5 * (rec[:, 0:3] + rec[:, 3:6])
        ng, wo = rec[:, 12:15], rec[:, 8:11]
        ng = torch.where((ng * wo).sum(1, keepdim=True) < 0, -ng, ng)
        po = p + ng * eps
        m = len(idx)
        # light sample -> shadow ray
        k = torch.randint(0, len(quads), (m,), device=dev, generator=gen)
        u = torch.rand((m, 2), device=dev, generator=gen)
        pl = quads[k, 0] + u[:, :1] * (quads[k, 1] - quads[k, 0]) + u[:, 1:] * (quads[k, 3] - quads[k, 0])
        wi = pl - po
        dist2 = (wi * wi).sum(1)
        wn = wi / dist2.sqrt()[:, None]
        Ld = (albedo / np.pi) * (ng * wn).sum(1).clamp_min(0) * (quad_n[k] * wn).sum(1).abs() / dist2 * quad_area[k] * \
            len(quads) * 12.0
        Ld = Ld * beta[idx]
        keep = Ld > 0
        sq = RayQueue(po[keep].T.contiguous(), wi[keep].T.contiguous(),
                      tmax=torch.full((int(keep.sum()),), 1 - 1e-4, device=dev))
        shadow = (sq, Ld[keep][:, None].expand(-1, 4).contiguous(), slot_of[idx][keep].contiguous())
        # cosine-weighted bounce (pdf cos / pi cancels the diffuse BRDF's cosine: throughput *= albedo)
        u2 = torch.rand((m, 2), device=dev, generator=gen)
        r, phi = u2[:, 0].sqrt(), 2 * np.pi * u2[:, 1]
        a = torch.zeros_like(ng)
        steep = ng[:, 0].abs() > 0.9
        a[:, 0], a[:, 1] = (~steep).float(), steep.float()
        t = torch.linalg.cross(ng, a)
        t = t / t.norm(dim=1, keepdim=True)
        b = torch.linalg.cross(ng, t)
        d = (r * phi.cos())[:, None] * t + (r * phi.sin())[:, None] * b + (1 - u2[:, 0]).clamp_min(0).sqrt()[:, None] * ng
        bounce = (RayQueue(po.T.contiguous(), d.T.contiguous()), beta[idx] * albedo, slot_of[idx].contiguous())
        return shadow, bounce

    for s in range(args.spp):
        rays, px, py = scene.camera_rays(cam, seed=1, sample=s, subsample=args.scale, return_pixels=True)
        n = len(rays)
        pixel = torch.from_numpy((py // args.scale) * xres + (px // args.scale)).to(dev).long()
        L = torch.zeros((n, 4), dtype=torch.float32, device=dev)   # one accumulator row per pixel sample
        slot_of = torch.arange(n, dtype=torch.int32, device=dev)    # the pixel-sample slot of a queue item
        beta = torch.ones(n, device=dev)
        rq = RayQueue.from_records(rays, dev)
        escaped, material = WorkQueue(n, dev), WorkQueue(n, dev)
        hits = timed(lambda: wf.IntersectClosest(n, rq, escaped=escaped, basic_eval_material=material))
        n_traced += n
        L[slot_of[escaped.indices().long()].long()] += 0.02
        for depth in range(args.depth + 1):
            (sq, Ld4, sslot), (bq, bbeta, bslot) = shade(rq, hits, rq.capacity, material, beta)
            half = torch.full((sq.capacity, 4), 0.5, device=dev)
            if depth == args.depth or bq.capacity == 0:
                if sq.capacity:
                    timed(lambda: wf.IntersectShadow(sq.capacity, sq, Ld4, half, half, sslot, L))
                    n_traced += sq.capacity
                break
            # the shadow rays of this depth and the rays of the next one: ONE launch
            rq, beta, slot_of = bq, bbeta, bslot
            escaped, material = WorkQueue(rq.capacity, dev), WorkQueue(rq.capacity, dev)
            if sq.capacity:
                hits = timed(lambda: wf.IntersectClosestAndShadow(rq.capacity, rq, sq.capacity, sq, Ld4, half, half,
                                                                  sslot, L, escaped=escaped,
                                                                  basic_eval_material=material))
            else:
                hits = timed(lambda: wf.IntersectClosest(rq.capacity, rq, escaped=escaped, basic_eval_material=material))
            n_traced += rq.capacity + sq.capacity
            esc = escaped.indices().long()
            L[slot_of[esc].long()] += 0.02 * beta[esc][:, None]
        film.index_add_(0, pixel, L)
    img = (film[:, :3] / args.spp).view(yres, xres, 3)
    img = (img / (1 + img)).clamp(0, 1) ** (1 / 2.2)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    write_png(args.out, (img.cpu().numpy() * 255 + 0.5).astype(np.uint8))
    print(f"{source}: {xres}x{yres}, {args.spp} spp, depth {args.depth}: {n_traced} rays in {launches} trace launches, "
          f"{t_trace * 1e3:.1f} ms of trace stages ({n_traced / t_trace / 1e6:.0f} Mray/s incl. queue kernels and "
          f"launch synchronisation) -> {args.out}")


if __name__ == "__main__":
    main()
