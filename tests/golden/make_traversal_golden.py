#!/usr/bin/env python3
"""tests/golden/traversal_small.npz: a <= 4 K-triangle sub-mesh of the reference's coffee_maker
scene (data file), the tree our host builder makes for it, 8 192 rays
(random, edge-case and shadow-style) and the ORACLE's per-ray results.

Pinning note: the expected values come from oracle/nnbvh_oracle.c, not from the reference
binary (BVHAggregate cannot be built here); the fixture is a regression
anchor for the HIP path and for the oracle itself, not an independent pin."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402
import scenes_small as ss  # noqa: E402
from nn_bvh_amd import build_tree, make_prims, scene  # noqa: E402

verts, tris = scene.load_blob("coffee_maker")
tris = tris[100000:104000]
used, inv = np.unique(tris, return_inverse=True)
verts = verts[used]
tris = inv.reshape(-1, 3).astype(np.int32)
prims = make_prims(tris)
tree = build_tree(prims, verts)
lo, hi = verts.min(0), verts.max(0)
rays = np.concatenate([scene.random_rays(4096, lo - 0.02, hi + 0.02, 1),
                       ss.edge_case_rays(verts, prims, 2, 4096)])
hits = ob.closest(tree.nodes, tree.ordered_prims, verts, rays)
occ, vis, tst = ob.any_hit(tree.nodes, tree.ordered_prims, verts, rays)
out = os.path.join(ROOT, "tests", "golden", "traversal_small.npz")
np.savez_compressed(out, verts=verts, nodes=tree.nodes, ordered_prims=tree.ordered_prims,
                    rays=rays, hits=hits, occ=occ, occ_visited=vis, occ_tests=tst)
print(len(tris), "tris", len(tree.nodes), "nodes", len(rays), "rays", "hit frac",
      (hits["prim"] >= 0).mean(), "->", os.path.getsize(out) // 1024, "KiB")
