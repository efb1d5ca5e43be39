#!/usr/bin/env python3
"""Launch time as a function of batch size (strided subsets / repeats of the crown primary and
bounce batches): separates the steady-state rate from the per-launch ramp + drain."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene  # noqa: E402

verts, tris, source = scene.load_scene("crown")
tree = build_tree(make_prims(tris), verts)
agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
for k, v in [a.split("=") for a in sys.argv[1:]]:
    agg.set_option(k, int(v))
primary = scene.camera_rays("crown", seed=1, sample=0)
hits = agg.Intersect(primary)
bounce = scene.bounce_rays(primary, hits, verts, tris, seed=2)
stream = torch.cuda.current_stream().cuda_stream
for name, rays in (("primary", primary), ("bounce", bounce)):
    vmax = agg.Intersect(rays)["nodes_visited"]
    print(f"# {name}: V mean {vmax.mean():.1f} p99 {np.percentile(vmax, 99):.0f} max {vmax.max()}")
    for label, batch in [(f"1/{s}", np.ascontiguousarray(rays[::s])) for s in (16, 8, 4, 2, 1)] + \
                        [(f"x{m}", np.concatenate([rays] * m)) for m in (2, 4, 8)]:
        d = torch.from_numpy(batch.view(np.uint8).reshape(-1)).cuda()
        out = torch.empty(len(batch) * 32, dtype=torch.uint8, device="cuda")
        ts = []
        for it in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            agg.intersect_device(d.data_ptr(), out.data_ptr(), len(batch), stream)
            b.record()
            torch.cuda.synchronize()
            if it:
                ts.append(a.elapsed_time(b))
        t = float(np.median(ts))
        print(f"{name:8s} {label:5s} n={len(batch):9d}  {t * 1e3:9.1f} us  {len(batch) / t / 1e3:8.1f} Mray/s",
              flush=True)
