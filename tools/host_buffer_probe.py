import sys, time
sys.path.insert(0, '.')
import numpy as np
from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene
verts, tris, source = scene.load_scene("crown")
tree = build_tree(make_prims(tris), verts)
agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
primary = np.concatenate([scene.camera_rays("crown", seed=1, sample=s) for s in range(8)])
agg.Intersect(primary[:100000])
ts = []
for _ in range(4):
    t0 = time.perf_counter(); h = agg.Intersect(primary); ts.append(time.perf_counter() - t0)
print(f"host-buffer closest hit, {len(primary)} rays: {np.median(ts) * 1e3:.1f} ms = {len(primary) / np.median(ts) / 1e6:.0f} Mray/s (pageable numpy buffers)")
