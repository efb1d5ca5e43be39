#!/usr/bin/env python3
"""Two-level scene throughput: N placements of one loop-subdivided killeroo (33 264 triangles,
data/killeroos.npz) on a grid + a ground quad at the top level; closest-hit over a camera
batch with the INST kernels (parity of this path is tests/test_instancing.py's job)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from nn_bvh_amd import BVHAggregate, HIT_DTYPE, instancing, make_prims, scene  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 20
verts, tris = scene.load_blob("killeroos")
k1 = tris[4:4 + 33264]                      # the first killeroo's triangles
used, inv = np.unique(k1, return_inverse=True)
kv = verts[used]
ktris = inv.reshape(-1, 3).astype(np.int32)
ext = kv.max(0) - kv.min(0)
ground = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], np.float32) * side * ext.max() + [0, 0, kv[:, 2].min()]
allv = np.concatenate([kv, ground]).astype(np.float32)
obj = make_prims(ktris)
top = make_prims(np.array([[0, 1, 2], [2, 3, 0]], np.int32) + len(kv))
top["id"] += 10_000_000
rng = np.random.default_rng(1)
placements = []
for i in range(side):
    for j in range(side):
        a = rng.uniform(0, 2 * np.pi)
        s = rng.uniform(0.6, 1.2)
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) * s
        M = np.eye(4)
        M[:3, :3] = R
        M[:3, 3] = [(i - side / 2) * ext[0] * 1.3, (j - side / 2) * ext[1] * 1.3, 0]
        placements.append((0, M[:3].astype(np.float32).reshape(12), np.linalg.inv(M)[:3].astype(np.float32).reshape(12)))
t0 = time.time()
nodes, prims, instances, n_top = instancing.assemble_two_level(top, allv, [obj], placements)
agg = BVHAggregate.from_tree(nodes, prims, allv, instances=instances, n_top_nodes=n_top)
print(f"{len(placements)} instances x {len(ktris)} tris = {len(placements) * len(ktris) / 1e6:.1f} M instanced "
      f"triangles; {len(nodes)} nodes ({n_top} top-level); build {time.time() - t0:.1f}s")
c = np.array([0, 0, kv[:, 2].mean()])
eye = c + np.array([1.0, 0.8, 0.5]) * side * ext.max() * 0.9
cam = (tuple(eye), tuple(c), (0, 0, 1), 45.0, 1400, 1000)
rays = np.concatenate([scene.camera_rays(cam, seed=1, sample=s) for s in range(4)])
d_r = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
d_h = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ts = []
for it in range(6):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    agg.intersect_device(d_r.data_ptr(), d_h.data_ptr(), len(rays), st)
    b.record()
    torch.cuda.synchronize()
    if it:
        ts.append(a.elapsed_time(b))
hits = d_h.cpu().numpy().view(HIT_DTYPE)
print(f"closest hit: {len(rays) / np.median(ts) / 1e3:.1f} Mray/s ({np.median(ts):.2f} ms for {len(rays)} rays); "
      f"hit {np.mean(hits['prim'] >= 0):.2f}, in instances {np.mean(hits['instance'] > 0):.2f}, "
      f"V {hits['nodes_visited'].mean():.1f}, T {hits['prim_tests'].mean():.2f}")
