#!/usr/bin/env python3
"""Randomised check of the device kd-tree builder against the host builder with the same tie order
(nnbvh_kd_build_create_gpu vs nnbvh_kd_build_create_stable): random scenes (soups with patches, connected
meshes, coincident boxes, coordinates snapped to a coarse grid so that edges tie and +0 / -0 occur, duplicated
primitives, flat scenes) x random build parameters; node array and primitiveIndices must be byte-identical.
Usage (GPU box): python tools/fuzz_kd_build.py [--iterations 300] [--seed 1]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import scenes_small as ss
    from nn_bvh_amd.kdtree import build_kd_tree
    rng = np.random.default_rng(args.seed)
    bad = 0
    for it in range(args.iterations):
        kind = int(rng.integers(0, 5))
        seed = int(rng.integers(0, 1 << 30))
        if kind == 0:
            verts, prims = ss.random_soup(int(rng.integers(1, 3000)), int(rng.integers(0, 300)), seed, extent=float(rng.choice([1, 8, 100])))
        elif kind == 1:
            verts, prims = ss.grid_mesh(int(rng.integers(2, 40)), seed, bump=float(rng.choice([0.0, 0.3, 2.0])))
        elif kind == 2:
            verts, prims = ss.coincident_centroids(int(rng.integers(2, 200)), seed)
        elif kind == 3:  # duplicated primitives
            verts, prims = ss.random_soup(int(rng.integers(2, 400)), 0, seed)
            prims = np.concatenate([prims, prims[rng.integers(0, len(prims), len(prims) // 2)]])
            prims["id"] = np.arange(len(prims))
        else:            # flat: every vertex on one plane
            verts, prims = ss.random_soup(int(rng.integers(2, 800)), 0, seed)
            verts = verts.copy()
            verts[:, int(rng.integers(0, 3))] = np.float32(rng.choice([0.0, -0.0, 1.5]))
        if rng.random() < 0.5:  # snap to a grid: ties everywhere, and negative zeros
            q = np.float32(rng.choice([0.25, 0.5, 1.0]))
            verts = (np.round(verts / q) * q).astype(np.float32)
        kw = dict(max_prims=int(rng.choice([1, 1, 2, 4, 8])), max_depth=int(rng.choice([-1, -1, -1, 1, 3, 8])),
                  isect_cost=int(rng.choice([5, 5, 1, 80])), traversal_cost=int(rng.choice([1, 1, 4])),
                  empty_bonus=float(rng.choice([0.5, 0.5, 0.0, 1.0])))
        h = build_kd_tree(prims, verts, where="host_stable", **kw)
        g = build_kd_tree(prims, verts, where="gpu", **kw)
        same = g.nodes.tobytes() == h.nodes.tobytes() and g.prim_indices.tobytes() == h.prim_indices.tobytes() and g.depth == h.depth
        if not same:
            bad += 1
            print(f"MISMATCH iteration {it}: kind {kind} seed {seed} n {len(prims)} {kw}", flush=True)
        elif it % 25 == 0:
            print(f"iteration {it}: kind {kind}, {len(prims)} primitives, {len(h.nodes)} nodes, depth {h.depth}: equal", flush=True)
    print(f"{args.iterations} scenes: {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
