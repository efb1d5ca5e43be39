#!/usr/bin/env python3
"""KdTreeAggregate construction: the device builder (nnbvh_kd_build_create_gpu) against the host builder with
the same tie order (nnbvh_kd_build_create_stable) per scene: byte identity and build times.
Usage: python tools/bench_kd_build.py [crown bathroom killeroos coffee_maker] [--no-host]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from nn_bvh_amd import make_prims, scene
    from nn_bvh_amd.kdtree import build_kd_tree
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["killeroos", "bathroom", "crown"]
    out = {}
    for name in names:
        verts, tris, source = scene.load_scene(name)
        prims = make_prims(tris)
        build_kd_tree(prims[:1000], verts, where="gpu")  # first-call costs (module load) outside the timing
        t0 = time.perf_counter()
        g = build_kd_tree(prims, verts, where="gpu")
        t_gpu = time.perf_counter() - t0
        rec = {"source": source, "triangles": int(len(tris)), "nodes": int(len(g.nodes)), "indices": int(len(g.prim_indices)),
               "depth": int(g.depth), "gpu_call_s": round(t_gpu, 3), "gpu_device_ms": round(g.build_ms[0], 1),
               "gpu_incl_download_ms": round(g.build_ms[1], 1)}
        if "--no-host" not in sys.argv:
            t0 = time.perf_counter()
            h = build_kd_tree(prims, verts, where="host_stable")
            rec["host_s"] = round(time.perf_counter() - t0, 2)
            rec["identical"] = bool(g.nodes.tobytes() == h.nodes.tobytes() and g.prim_indices.tobytes() == h.prim_indices.tobytes())
        out[name] = rec
        print(name, json.dumps(rec), flush=True)
    return out


if __name__ == "__main__":
    main()
