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
from nn_bvh_amd import build_tree, build_tree_gpu, make_prims, scene  # noqa: E402

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

    ms_sah, _ = timed(lambda: build_tree(prims, verts, 4, "sah"), 1)
    ms_hl, host = timed(lambda: build_tree(prims, verts, 4, "hlbvh"), 3)
    build_tree_gpu(prims, verts, 4)  # warm-up: module load, allocator
    ms_gpu, dev = timed(lambda: build_tree_gpu(prims, verts, 4), 5)
    same = dev.nodes.tobytes() == host.nodes.tobytes() and \
        dev.ordered_prims.tobytes() == host.ordered_prims.tobytes() and dev.depth == host.depth
    out[name] = {"triangles": int(len(tris)), "hlbvh_nodes": int(len(host.nodes)),
                 "host_sah_ms": round(ms_sah, 1), "host_hlbvh_ms": round(ms_hl, 1),
                 "gpu_hlbvh_ms_end_to_end": round(ms_gpu, 1),
                 "gpu_phases_ms": {k: round(v, 2) for k, v in dev.gpu_ms.items()},
                 "identical_to_host_hlbvh": bool(same)}
print(json.dumps(out))
