#!/usr/bin/env python3
"""Build-time comparison on the GPU box: host SAH (threaded), host HLBVH and GPU HLBVH
(nnbvh_build_create_gpu) per scene blob; the GPU tree is checked byte-for-byte against the host
HLBVH.  Prints one JSON object.  usage: tools/bench_build.py [scene ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from nn_bvh_amd import BVHAggregate, build_tree, build_tree_gpu, make_prims, scene  # noqa: E402

names = sys.argv[1:] or ["killeroos", "coffee_maker", "bathroom", "crown"]
out = {}
for name in names:
    if not os.path.exists(os.path.join(ROOT, "data", name + ".npz")):
        continue
    verts, tris = scene.load_blob(name)
    prims = make_prims(tris)

    def timed(fn, reps):
        best, res = 1e30, None
        for _ in range(reps):
            t0 = time.perf_counter()
            res = fn()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3, res

    ms_hl, host = timed(lambda: build_tree(prims, verts, 4, "hlbvh"), 3)
    ms_sah, host_sah = timed(lambda: build_tree(prims, verts, 4, "sah"), 2)
    build_tree_gpu(prims, verts, 4)  # warm-up: module load, allocator
    ms_gpu, dev = timed(lambda: build_tree_gpu(prims, verts, 4), 5)
    build_tree_gpu(prims, verts, 4, split_method="sah")
    ms_gpu_sah, dev_sah = timed(lambda: build_tree_gpu(prims, verts, 4, split_method="sah"), 5)

    def same(a, b):
        return bool(a.nodes.tobytes() == b.nodes.tobytes() and
                    a.ordered_prims.tobytes() == b.ordered_prims.tobytes() and a.depth == b.depth)
    def scene_host():
        t = build_tree(prims, verts, 4, "sah")
        BVHAggregate.from_tree(t.nodes, t.ordered_prims, verts).close()

    def scene_device():
        BVHAggregate.build_on_device(prims, verts, 4, "sah").close()

    ms_scene_host, _ = timed(scene_host, 2)
    scene_device()
    ms_scene_dev, _ = timed(scene_device, 5)
    out[name] = {"triangles": int(len(tris)), "sah_nodes": int(len(host_sah.nodes)),
                 "scene_create_ms_host_build_and_bake_sah": round(ms_scene_host, 1),
                 "scene_create_ms_device_build_and_bake_sah": round(ms_scene_dev, 1),
                 "hlbvh_nodes": int(len(host.nodes)),
                 "host_sah_ms": round(ms_sah, 1), "host_hlbvh_ms": round(ms_hl, 1),
                 "gpu_sah_ms_end_to_end": round(ms_gpu_sah, 1),
                 "gpu_sah_phases_ms (upload, big nodes, subtrees, layout+bounds, download)":
                     [round(v, 2) for v in dev_sah.gpu_ms],
                 "gpu_sah_identical_to_host": same(dev_sah, host_sah),
                 "gpu_hlbvh_ms_end_to_end": round(ms_gpu, 1),
                 "gpu_hlbvh_phases_ms (upload, device tree, host upper, emit, download)":
                     [round(v, 2) for v in dev.gpu_ms],
                 "gpu_hlbvh_identical_to_host": same(dev, host)}
print(json.dumps(out))
