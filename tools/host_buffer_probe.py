#!/usr/bin/env python3
"""Rate of the host-buffer entry point nnbvh_intersect_closest (H2D 32 B + trace + D2H 32 B per ray, chunked and
pipelined over three streams) on crown primary rays: pageable numpy buffers against buffers pinned with
nnbvh_host_register, and the chunk size.  Prints the PCIe link the box reports.
Usage: python tools/host_buffer_probe.py [--spp 8]"""
import ctypes
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pcie_link():
    out = []
    for d in glob.glob("/sys/class/drm/card*/device"):
        try:
            out.append(f"{open(d + '/current_link_speed').read().strip()} x{open(d + '/current_link_width').read().strip()}")
        except OSError:
            pass
    return sorted(set(out)) or ["unknown"]


def main():
    from nn_bvh_amd import BVHAggregate, HIT_DTYPE, _lib, build_tree, make_prims, scene
    spp = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else 8
    verts, tris, source = scene.load_scene("crown")
    tree = build_tree(make_prims(tris), verts)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts)
    primary = np.concatenate([scene.camera_rays("crown", seed=1, sample=s) for s in range(spp)])
    n = len(primary)
    print(f"# {source}; {n} primary rays; PCIe link: {', '.join(pcie_link())}")
    ref = agg.Intersect(primary)

    def rate(label, rays, out):
        agg.Intersect(rays[:200000], out[:200000])
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            agg.Intersect(rays, out)
            ts.append(time.perf_counter() - t0)
        same = out.tobytes() == ref.tobytes()
        print(f"{label}: {np.median(ts) * 1e3:.1f} ms = {n / np.median(ts) / 1e6:.0f} Mray/s; identical to the first call: {same}", flush=True)

    out = np.zeros(n, HIT_DTYPE)
    chunks = lambda c: -(-n // max(c, -(-n // 6)))  # noqa: E731  (at most 6 chunks of at least host_chunk rays)
    rate(f"pageable buffers, {chunks(1 << 20)} chunks", primary, out)
    L = _lib.lib()
    import mmap  # buffers to pin own their pages (include/nnbvh.h, nnbvh_host_register)
    maps = [mmap.mmap(-1, (n * 32 + 4095) // 4096 * 4096) for _ in range(2)]
    pr, po = np.frombuffer(maps[0], primary.dtype, n), np.frombuffer(maps[1], HIT_DTYPE, n)
    pr[:] = primary
    for a, m in zip((pr, po), maps):
        _lib.check(L.nnbvh_host_register(ctypes.c_void_p(a.ctypes.data), len(m)), "nnbvh_host_register")
    rate(f"pinned buffers (nnbvh_host_register), {chunks(1 << 20)} chunks", pr, po)
    for c in (1 << 22, n // 2 + 1):
        agg.set_option("host_chunk", c)
        rate(f"pinned buffers, {chunks(c)} chunks", pr, po)
    agg.set_option("host_chunk", 1 << 30)
    rate("pinned buffers, ONE chunk (no overlap)", pr, po)
    # torch-pinned (hipHostMalloc) buffers, for comparison with registered ones
    try:
        import torch
        agg.set_option("host_chunk", 1 << 20)
        tr = torch.from_numpy(primary.view(np.uint8).reshape(-1)).pin_memory()
        to = torch.empty(n * 32, dtype=torch.uint8).pin_memory()
        rate(f"hipHostMalloc'ed buffers, {chunks(1 << 20)} chunks", tr.numpy().view(primary.dtype), to.numpy().view(HIT_DTYPE))
    except Exception as e:  # torch is optional here
        print(f"(torch-pinned variant skipped: {e})")
    for a in (pr, po):
        _lib.check(L.nnbvh_host_unregister(ctypes.c_void_p(a.ctypes.data)), "nnbvh_host_unregister")


if __name__ == "__main__":
    main()
