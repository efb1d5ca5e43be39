#!/usr/bin/env python3
"""How much do the per-launch drains cost?  Runs the bench's three batches (a) back to back on
one stream and (b) on three streams at once, and prints both times."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
verts, tris, _ = scene.load_scene("crown")
tree = build_tree(make_prims(tris), verts)
agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
primary = np.concatenate([scene.camera_rays("crown", seed=1, sample=s) for s in range(spp)])
hits = agg.Intersect(primary)
bounce = scene.bounce_rays(primary, hits, verts, tris, seed=2)
lo, hi = verts.min(0), verts.max(0)
shadow = scene.shadow_rays_to_quads(primary, hits, verts, tris, scene.CROWN_LIGHT_QUADS, seed=3)
dev = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()
d = [dev(primary), dev(bounce), dev(shadow)]
n = [len(primary), len(bounce), len(shadow)]
o = [torch.empty(len(primary) * 32, dtype=torch.uint8, device="cuda") for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]


def launch(i, st):
    if i == 2:
        agg.intersect_p_device(d[i].data_ptr(), o[i].data_ptr(), n[i], stream=st.cuda_stream)
    else:
        agg.intersect_device(d[i].data_ptr(), o[i].data_ptr(), n[i], st.cuda_stream)


import time
for mode in ("serial", "3 streams"):
    ts = []
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(5):
            for i in range(3):
                launch(i, streams[0] if mode == "serial" else streams[i])
        torch.cuda.synchronize()
        if it:
            ts.append((time.perf_counter() - t0) / 5)
    t = float(np.median(ts))
    print(f"spp {spp} {mode:10s}: {t * 1e3:7.3f} ms per step, {sum(n) / t / 1e6:8.1f} Mray/s", flush=True)
