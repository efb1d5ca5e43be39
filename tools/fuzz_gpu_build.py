#!/usr/bin/env python3
"""Randomised check of the device builders against the host builder (byte identity) over many
scene shapes: uniform / clustered / lattice (ties everywhere) / duplicated primitives, sizes 1 ...
60 000, maxPrims 1 ... 8.  usage: tools/fuzz_gpu_build.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from nn_bvh_amd import build_tree, build_tree_gpu, make_prims  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    n = int(rng.choice([1, 2, 3, 7, 64, 65, 255, 256, 257, 1000, 5000, 20000, 60000]))
    kind = int(rng.integers(0, 5))
    if kind == 0:      # uniform soup
        c = rng.uniform(-10, 10, (n, 1, 3))
    elif kind == 1:    # clusters of very different scale
        k = max(1, n // 50)
        centers = rng.uniform(-100, 100, (k, 3))
        c = (centers[rng.integers(0, k, n)] + rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-4, 0, (n, 1)))[:, None]
    elif kind == 2:    # lattice: many equal centroid coordinates
        c = rng.integers(0, 6, (n, 1, 3)).astype(np.float64)
    elif kind == 3:    # many exact duplicates
        base = rng.uniform(-5, 5, (max(1, n // 7), 3))
        c = base[rng.integers(0, len(base), n)][:, None]
    else:              # flat (all in one plane) and elongated
        c = rng.uniform(-10, 10, (n, 1, 3)) * np.array([1, 1e-3, 0])
    size = 10.0 ** rng.uniform(-3, 0)
    off = rng.uniform(-size, size, (n, 3, 3))
    if kind in (2, 3):
        off = np.tile(rng.uniform(-size, size, (1, 3, 3)), (n, 1, 1))  # identical shapes -> identical centroids
    verts = (c + off).reshape(-1, 3).astype(np.float32)
    prims = make_prims(np.arange(3 * n, dtype=np.int32).reshape(n, 3))
    max_prims = int(rng.choice([1, 2, 4, 8]))
    for method in ("sah", "hlbvh"):
        try:
            host = build_tree(prims, verts, max_prims, method)
        except Exception as e:  # the reference aborts on some degenerate HLBVH inputs; both must refuse
            try:
                build_tree_gpu(prims, verts, max_prims, split_method=method)
                print(f"case {case} {method}: host refused ({e}) but the device built a tree")
                bad += 1
            except Exception:
                pass
            continue
        dev = build_tree_gpu(prims, verts, max_prims, split_method=method)
        same = dev.nodes.tobytes() == host.nodes.tobytes() and \
            dev.ordered_prims.tobytes() == host.ordered_prims.tobytes() and dev.depth == host.depth
        if not same:
            bad += 1
            print(f"case {case}: n {n} kind {kind} maxPrims {max_prims} {method}: MISMATCH "
                  f"({len(dev.nodes)} vs {len(host.nodes)} nodes, depth {dev.depth} vs {host.depth})")
print(f"{n_cases} cases x 2 methods: {bad} mismatches")
sys.exit(1 if bad else 0)
