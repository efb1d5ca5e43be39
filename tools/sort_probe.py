#!/usr/bin/env python3
"""Does re-ordering an incoherent ray batch pay?  Times the bounce (closest) and shadow (any)
batches of the bench workload in their natural order and in several sorted orders (host-side
sort, not timed): by direction octant, by Morton cell of the origin, combinations, and shuffled.
Results are order-independent per ray; this only measures speed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def morton3(q, bits):
    code = np.zeros(len(q), np.uint64)
    for b in range(bits):
        for k in range(3):
            code |= ((q[:, k] >> b) & 1).astype(np.uint64) << np.uint64(3 * b + k)
    return code


def main():
    import torch
    from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    verts, tris, _ = scene.load_scene("crown")
    tree = build_tree(make_prims(tris), verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    primary = np.concatenate([scene.camera_rays("crown", seed=1, sample=s) for s in range(spp)])
    hits = agg.Intersect(primary)
    bounce = scene.bounce_rays(primary, hits, verts, tris, seed=2)
    shadow = scene.shadow_rays_to_quads(primary, hits, verts, tris, scene.CROWN_LIGHT_QUADS, seed=3)
    lo, hi = verts.min(0), verts.max(0)
    stream = torch.cuda.current_stream().cuda_stream

    def orders(rays):
        n = len(rays)
        octant = ((rays["d"][:, 0] < 0) * 1 + (rays["d"][:, 1] < 0) * 2 + (rays["d"][:, 2] < 0) * 4).astype(np.uint64)
        out = {"natural": np.arange(n), "shuffled": np.random.default_rng(0).permutation(n),
               "octant (stable)": np.argsort(octant, kind="stable")}
        for bits in (4, 6, 8):
            q = np.clip(((rays["o"] - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
            m = morton3(q, bits)
            out[f"morton{bits}"] = np.argsort(m, kind="stable")
            out[f"morton{bits}+octant"] = np.argsort((m << np.uint64(3)) | octant, kind="stable")
            out[f"octant+morton{bits}"] = np.argsort((octant << np.uint64(3 * bits)) | m, kind="stable")
        return out

    for name, rays, closest in (("bounce", bounce, True), ("shadow", shadow, False)):
        out = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
        res = {}
        devs = {k: torch.from_numpy(np.ascontiguousarray(rays[p]).view(np.uint8).reshape(-1)).cuda()
                for k, p in orders(rays).items()}
        for rnd in range(4):
            for k, d in devs.items():
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                if closest:
                    agg.intersect_device(d.data_ptr(), out.data_ptr(), len(rays), stream)
                else:
                    agg.intersect_p_device(d.data_ptr(), out.data_ptr(), len(rays), stream=stream)
                b.record()
                torch.cuda.synchronize()
                if rnd:
                    res.setdefault(k, []).append(a.elapsed_time(b))
        print(f"{name}: {len(rays)} rays")
        for k, v in res.items():
            ms = float(np.median(v))
            print(f"  {k:22s} {ms:7.3f} ms  {len(rays) / ms / 1e3:8.1f} Mray/s")


if __name__ == "__main__":
    main()
