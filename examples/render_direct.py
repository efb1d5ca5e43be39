#!/usr/bin/env python3
"""A complete (tiny) wavefront renderer on top of the queue interface: direct lighting from the
crown scene's six area-light quads, every stage device-resident.

  triangles -> BVHAggregate.build_on_device    SAH tree built and baked on the GPU
  camera rays (SOA RayQueue)
    -> WavefrontAggregate.IntersectClosest       hit records + escaped / material index queues
    -> ShadingMesh.interactions_device           SurfaceInteraction records (p, n, ...) per hit
    -> "material stage" (a few torch ops over the material queue: light sample, unoccluded
       contribution Ld, shadow ray)              -> ShadowRayQueue
    -> WavefrontAggregate.IntersectShadow        L[pixel] += Ld / (r_u + r_l).Average() if visible
    -> film: mean over samples, tone map, PNG (zlib only)

The two trace stages are the library (HIP kernels); the torch ops in between stand in for the
reference's material / light stages (wavefront/integrator.cpp:403-579) and are plumbing.
usage: examples/render_direct.py [--scene crown] [--spp 16] [--scale 2] [--out render.png]"""
import argparse
import os
import struct
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_png(path, rgb8):
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="crown")
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--scale", type=int, default=2, help="render every k-th pixel per axis")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "render_direct.png"))
    args = ap.parse_args()
    import torch
    from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene
    from nn_bvh_amd.interaction import ShadingMesh
    from nn_bvh_amd.wavefront import RayQueue, WavefrontAggregate, WorkQueue

    dev = torch.device("cuda", 0)
    verts, tris, source = scene.load_scene(args.scene)
    t0 = time.perf_counter()
    agg = BVHAggregate.build_on_device(make_prims(tris), verts)  # SAH tree built and baked on the GPU
    t_build = time.perf_counter() - t0
    wf = WavefrontAggregate(agg)
    cam = args.scene if args.scene in scene.CAMERAS else "crown"
    xres, yres = scene.CAMERAS[cam][4] // args.scale, scene.CAMERAS[cam][5] // args.scale
    smesh = ShadingMesh(verts, tris)
    quads = torch.from_numpy(scene.CROWN_LIGHT_QUADS.astype(np.float32)).to(dev)
    quad_n = torch.linalg.cross(quads[:, 1] - quads[:, 0], quads[:, 3] - quads[:, 0])
    quad_area = quad_n.norm(dim=1)
    quad_n = quad_n / quad_area[:, None]
    eps = 1e-4 * float(np.abs(verts).max())

    film = torch.zeros((xres * yres, 4), dtype=torch.float32, device=dev)
    n_traced, t_trace = 0, 0.0
    gen = torch.Generator(device=dev).manual_seed(7)
    for s in range(args.spp):
        rays, px, py = scene.camera_rays(cam, seed=1, sample=s, subsample=args.scale, return_pixels=True)
        n = len(rays)
        pixel = torch.from_numpy((py // args.scale) * xres + (px // args.scale)).to(dev).int()
        rq = RayQueue.from_records(rays, dev)
        escaped, material = WorkQueue(n, dev), WorkQueue(n, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hits = wf.IntersectClosest(n, rq, escaped=escaped, basic_eval_material=material)
        intr = torch.empty((n, 48), dtype=torch.float32, device=dev)  # nnbvh_interaction = 192 B
        smesh.interactions_device(hits.data_ptr(), n, intr.data_ptr(), ray_queue=rq,
                                  stream=torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        t_trace += time.perf_counter() - t0
        # ---- material + light-sampling stage over the material queue (indices into rq) ----
        idx = material.indices().long()
        rec = intr[idx]                                   # nnbvh_interaction records of the hits
        p = 0.5 * (rec[:, 0:3] + rec[:, 3:6])             # Point3f(pi)
        ng, wo = rec[:, 12:15], rec[:, 8:11]
        ng = torch.where((ng * wo).sum(1, keepdim=True) < 0, -ng, ng)  # face the camera
        k = torch.randint(0, len(quads), (len(idx),), device=dev, generator=gen)
        u = torch.rand((len(idx), 2), device=dev, generator=gen)
        pl = quads[k, 0] + u[:, :1] * (quads[k, 1] - quads[k, 0]) + u[:, 1:] * (quads[k, 3] - quads[k, 0])
        po = p + ng * eps
        wi = pl - po
        dist2 = (wi * wi).sum(1)
        wn = wi / dist2.sqrt()[:, None]
        cos_s = (ng * wn).sum(1).clamp_min(0)
        cos_l = (quad_n[k] * wn).sum(1).abs()
        # diffuse albedo 0.7, unit emitted radiance, pdf = 1 / (6 * area) per light point
        Ld = (0.7 / np.pi) * cos_s * cos_l / dist2 * quad_area[k] * len(quads) * 40.0
        keep = Ld > 0
        m = int(keep.sum())
        sq = RayQueue(po[keep].T.contiguous(), wi[keep].T.contiguous(),
                      tmax=torch.full((m,), 1 - 1e-4, device=dev))
        Ld4 = Ld[keep][:, None].expand(-1, 4).contiguous()
        half = torch.full((m, 4), 0.5, device=dev)
        # one accumulator row per (pixel sample): unique within the stage, as in the reference
        L = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        slot = idx[keep].int().contiguous()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wf.IntersectShadow(m, sq, Ld4, half, half, slot, L)
        torch.cuda.synchronize()
        t_trace += time.perf_counter() - t0
        n_traced += n + m
        L[escaped.indices().long()] = 0.02  # faint background
        film.index_add_(0, pixel.long(), L)
    img = (film[:, :3] / args.spp).view(yres, xres, 3)
    img = (img / (1 + img)).clamp(0, 1) ** (1 / 2.2)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    write_png(args.out, (img.cpu().numpy() * 255 + 0.5).astype(np.uint8))
    print(f"scene built and baked on the device in {t_build * 1e3:.0f} ms")
    print(f"{source}: {xres}x{yres}, {args.spp} spp, {n_traced} rays in {t_trace * 1e3:.1f} ms of trace stages "
          f"({n_traced / t_trace / 1e6:.0f} Mray/s incl. queue kernels) -> {args.out}")


if __name__ == "__main__":
    main()
