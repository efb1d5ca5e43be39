#!/usr/bin/env python3
"""What would tracing the bounce / shadow rays in a coherent order buy?  Sorts each class on the host by a
key (origin Morton code, direction octant + origin Morton code) and times the closest / any-hit launch
on the sorted copy against the caller's order (the sort itself is NOT timed: this is the ceiling)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def morton3(q):
    def spread(v):
        v = v.astype(np.uint64) & 0x3ff
        v = (v | (v << 16)) & 0x30000ff
        v = (v | (v << 8)) & 0x300f00f
        v = (v | (v << 4)) & 0x30c30c3
        v = (v | (v << 2)) & 0x9249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def main():
    import torch
    from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene
    verts, tris, source = scene.load_scene("crown")
    tree = build_tree(make_prims(tris), verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    primary = np.concatenate([scene.camera_rays("crown", seed=1, sample=s) for s in range(8)])
    hits = agg.Intersect(primary)
    bounce = scene.bounce_rays(primary, hits, verts, tris, seed=2)
    shadow = scene.shadow_rays_to_quads(primary, hits, verts, tris, scene.CROWN_LIGHT_QUADS, seed=3)
    lo, hi = verts.min(0), verts.max(0)
    stream = torch.cuda.current_stream().cuda_stream

    def orders(rays):
        q = np.clip((rays["o"] - lo) / (hi - lo) * 1023, 0, 1023).astype(np.uint32)
        m = morton3(q)
        octant = ((rays["d"][:, 0] < 0).astype(np.uint64) | ((rays["d"][:, 1] < 0).astype(np.uint64) << 1) |
                  ((rays["d"][:, 2] < 0).astype(np.uint64) << 2))
        dq = np.clip((rays["d"] / np.linalg.norm(rays["d"], axis=1, keepdims=True) * 0.5 + 0.5) * 7, 0, 7).astype(np.uint64)
        dkey = dq[:, 0] | (dq[:, 1] << 3) | (dq[:, 2] << 6)
        return {"caller's order": np.arange(len(rays)), "origin morton": np.argsort(m, kind="stable"),
                "octant, origin morton": np.argsort((octant << 30) | m, kind="stable"),
                "origin morton >> 12, direction 9 bits, rest": np.argsort(((m >> 12) << 21) | (dkey << 12) | (m & 0xfff), kind="stable"),
                "random": np.random.default_rng(1).permutation(len(rays))}

    for name, rays, anyhit in (("bounce closest", bounce, False), ("shadow any", shadow, True)):
        out = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
        for label, perm in orders(rays).items():
            d = torch.from_numpy(np.ascontiguousarray(rays[perm]).view(np.uint8).reshape(-1)).cuda()
            ts = []
            for rep in range(6):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                if anyhit:
                    agg.intersect_p_device(d.data_ptr(), out.data_ptr(), len(rays), stream=stream)
                else:
                    agg.intersect_device(d.data_ptr(), out.data_ptr(), len(rays), stream)
                b.record()
                torch.cuda.synchronize()
                if rep:
                    ts.append(a.elapsed_time(b))
            print(f"{name:15s} {label:45s} {len(rays) / np.median(ts) / 1e3:8.1f} Mray/s", flush=True)


if __name__ == "__main__":
    main()
