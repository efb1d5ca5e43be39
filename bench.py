#!/usr/bin/env python3
"""bench.py — Mray/s of the HIP BVH traversal path on the crown workload (BASELINE.json).

One "step" = one wavefront pass of --spp samples per pixel (default 8, each sample its own
jitter stream) over the crown film through the hot path, all ray batches already resident in
HBM, one launch per ray class:
    closest-hit  over spp x 1000x1400 primary rays   (BVHAggregate::Intersect)
    closest-hit  over the diffuse-bounce rays of those hits
    any-hit      over the shadow rays of those hits  (BVHAggregate::IntersectP, tMax = 1-1e-4)
value = rays traced by all ranks / wall time of K steps (max over ranks).

Why 8 spp per launch: a launch's time is T(n) = ramp/drain + n / steady-state rate, and the
drain is the dependent-load chain of the longest ray in the batch (crown: V up to ~600 nodes,
about 0.55 ms) whatever n is (tools/scaling.py).  pbrt's wavefront integrator caps
its queues at ~1 M samples (wavefront/integrator.cpp:230-234) for the memory of other GPUs;
with 288 GB of HBM the natural MI355X design is fewer, larger launches (8 spp of the crown
film = 11.2 M rays = 0.7 GB of ray + hit records).  --spp 1 reproduces the 1 M-ray regime.

Multi-GPU: the BVH is replicated; each rank owns an interleaved set of 16x16
image tiles (Morton-ordered ray chunks), traces only its tiles' rays — no data-path
collective inside the timed steps — and the per-tile results are all-gathered once after the
timed region (reported separately as allgather_ms).  Per-GPU work is fixed as N grows
(weak scaling): with --gpus N the job is N samples per pixel and rank r traces its tiles of
every sample, i.e. --spp films' worth of rays per GPU.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--scene crown] [--no-cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md §HBM)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="crown")
    ap.add_argument("--spp", type=int, default=8, help="samples per pixel traced per step")
    ap.add_argument("--tree", default="sah", choices=["sah", "hlbvh", "middle", "equal", "nn", "sah_gpu", "hlbvh_gpu"],
                    help="tree builder: pbrt split methods (host), nn = greedy-SAH top levels of "
                         "machine_learning/nn_BVH.py finished by SAH and baked (BASELINE config 5), "
                         "*_gpu = the same sah / hlbvh tree built and baked on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--wavefront", action="store_true",
                    help="also time the step through the wavefront-queue entry points")
    ap.add_argument("--overlapped", action="store_true",
                    help="also time the step as one nnbvh_trace_batches_device call (concurrent "
                         "launches; off by default so that a rocprofv3 kernel average of this "
                         "command equals roofline.avg_launch_ms)")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000,
                    help="rays per class timed on the host cores for cpu_baseline")
    ap.add_argument("--cpu-passes", type=int, default=5)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the traversal path has no CPU fallback")
    # NNBVH_BENCH_BACKEND=gloo (+ all ranks on the visible GPUs round-robin) is the rehearsal
    # mode for boxes with fewer GPUs than ranks; the driver's runs use nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("NNBVH_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    coll_dev = "cuda" if backend == "nccl" else "cpu"

    from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene, shard
    from nn_bvh_amd._lib import HIT_DTYPE

    t0 = time.time()
    verts, tris, source = scene.load_scene(args.scene)
    prims = make_prims(tris)
    if args.tree == "nn":
        from nn_bvh_amd import nn_tree
        from nn_bvh_amd.aggregate import BuiltTree
        (nn_nodes, nn_ordered), _ = nn_tree.greedy_sah_tree(verts, tris, levels=4)
        tree = BuiltTree(nn_nodes, nn_ordered, -1)
    elif args.tree in ("sah_gpu", "hlbvh_gpu"):
        # built and baked on the device; the host copy of the same tree only serves the oracle leg
        tree = build_tree(prims, verts, 4, args.tree[:-4])
    else:
        tree = build_tree(prims, verts, 4, args.tree)
    if args.tree in ("sah_gpu", "hlbvh_gpu"):
        t_dev = time.time()
        agg = BVHAggregate.build_on_device(prims, verts, 4, args.tree[:-4], device=local_rank)
        if rank == 0:
            log(f"[bench] scene built and baked on the device in {(time.time() - t_dev) * 1e3:.0f} ms")
    else:
        agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, verts, device=local_rank)
    tree.depth = agg.info["depth"]
    if rank == 0:
        log(f"[bench] scene: {source}; {len(tris)} tris, {len(tree.nodes)} nodes, depth {tree.depth}; "
            f"build+upload {time.time() - t0:.1f}s; grid {agg.info['grid_blocks']} blocks, "
            f"window {agg.info['stack_window']}")

    # ---- ray batches (synthetic, seeded).  The job is `world * spp` samples per pixel of the
    # film; rank r traces its interleaved 16x16 tiles of every sample, i.e. spp films' worth of
    # rays per GPU whatever N is (weak scaling).
    cam_name = args.scene if args.scene in scene.CAMERAS else "crown"
    xres = scene.CAMERAS[cam_name][4]
    _, px, py = scene.camera_rays(cam_name, seed=1, sample=0, return_pixels=True)
    mine = shard.shard_indices(px, py, xres, world, rank)
    primary = np.concatenate([scene.camera_rays(cam_name, seed=1, sample=s_idx, subset=mine)
                              for s_idx in range(world * args.spp)])
    shard_bytes = world * args.spp * shard.shard_counts(px, py, xres, world) * 32  # hit bytes/rank
    n_primary = len(primary)

    def dev(a):
        return torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()

    stream = torch.cuda.current_stream().cuda_stream
    d_primary = dev(primary)
    d_hits = torch.empty(n_primary * 32, dtype=torch.uint8, device="cuda")
    agg.intersect_device(d_primary.data_ptr(), d_hits.data_ptr(), n_primary, stream)
    torch.cuda.synchronize()
    hits = d_hits.cpu().numpy().view(HIT_DTYPE)
    bounce = scene.bounce_rays(primary, hits, verts, tris, seed=2 + rank)
    lo, hi = verts.min(0), verts.max(0)
    if args.scene == "crown":  # towards the scene's six area-light quads (crown.pbrt:26-102)
        shadow = scene.shadow_rays_to_quads(primary, hits, verts, tris, scene.CROWN_LIGHT_QUADS,
                                            seed=3 + rank)
    else:
        shadow = scene.shadow_rays(primary, hits, verts, tris, lo + (hi - lo) * [0.3, 0.9, 0.3],
                                   lo + (hi - lo) * [0.7, 1.0, 0.7], seed=3 + rank)
    d_bounce, d_shadow = dev(bounce), dev(shadow)
    d_bhits = torch.empty(len(bounce) * 32, dtype=torch.uint8, device="cuda")
    d_occ = torch.empty(len(shadow), dtype=torch.uint8, device="cuda")
    rays_per_step = n_primary + len(bounce) + len(shadow)

    def step():
        agg.intersect_device(d_primary.data_ptr(), d_hits.data_ptr(), n_primary, stream)
        agg.intersect_device(d_bounce.data_ptr(), d_bhits.data_ptr(), len(bounce), stream)
        agg.intersect_p_device(d_shadow.data_ptr(), d_occ.data_ptr(), len(shadow), stream=stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([rays_per_step], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_rays_per_step = float(tot.item())
    else:
        total_rays_per_step = float(rays_per_step)

    # ---- per-kernel timing with events on the launch stream (dominant kernel = closest-hit) ----
    def time_kernel(fn, reps):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(reps)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in evs]))

    reps = max(3, min(args.steps, 10))
    ms_primary = time_kernel(lambda: agg.intersect_device(d_primary.data_ptr(), d_hits.data_ptr(),
                                                          n_primary, stream), reps)
    ms_bounce = time_kernel(lambda: agg.intersect_device(d_bounce.data_ptr(), d_bhits.data_ptr(),
                                                         len(bounce), stream), reps)
    ms_shadow = time_kernel(lambda: agg.intersect_p_device(d_shadow.data_ptr(), d_occ.data_ptr(),
                                                           len(shadow), stream=stream), reps)
    bhits = d_bhits.cpu().numpy().view(HIT_DTYPE)
    # algorithmic bytes: 32 in + 32*V + 48*T + 32 out per closest-hit ray
    def alg_bytes(h):
        return 64.0 * len(h) + 32.0 * h["nodes_visited"].sum(dtype=np.int64) + \
            48.0 * h["prim_tests"].sum(dtype=np.int64)
    bytes_closest = alg_bytes(hits) + alg_bytes(bhits)          # both launches of the kernel
    ms_closest = ms_primary + ms_bounce
    achieved = bytes_closest / (ms_closest * 1e-3) / 1e9         # GB/s over the kernel's launches

    # ---- the same step through nnbvh_trace_batches_device: the three batches run concurrently
    # on the library's internal streams, so each launch's drain overlaps the others' work.
    # Reported next to `value` (which stays the one-launch-at-a-time figure the per-kernel
    # roofline and the rocprofv3 kernel averages refer to).
    def step_overlapped():
        agg.trace_batches_device([("closest", d_primary.data_ptr(), n_primary, d_hits.data_ptr()),
                                  ("closest", d_bounce.data_ptr(), len(bounce), d_bhits.data_ptr()),
                                  ("any", d_shadow.data_ptr(), len(shadow), d_occ.data_ptr())], stream)

    overlapped_s = None
    if args.overlapped:
        step_overlapped()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_overlapped()
        barrier()
        overlapped_s = time.perf_counter() - t1
    if world > 1 and overlapped_s is not None:
        t = torch.tensor([overlapped_s], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        overlapped_s = float(t.item())

    # ---- the same step through the wavefront-queue entry points (SOA ray queues with
    # device-side sizes in, index queues and pixel radiance out): opt-in, reported next to `value`
    wavefront_s = None
    if args.wavefront:
        from nn_bvh_amd.wavefront import RayQueue, WavefrontAggregate, WorkQueue
        from nn_bvh_amd._lib import CLOSEST_QUEUES
        cdev = torch.device("cuda", local_rank)
        wf = WavefrontAggregate(agg, np.zeros(len(tris), np.uint8))
        q_primary, q_bounce = RayQueue.from_records(primary, cdev), RayQueue.from_records(bounce, cdev)
        q_shadow = RayQueue.from_records(shadow, cdev, shadow=True)
        outq = {k: WorkQueue(n_primary, cdev) for k in CLOSEST_QUEUES}
        rng = np.random.default_rng(11)
        spec = [torch.from_numpy(rng.random((len(shadow), 4), np.float32) + np.float32(0.5)).to(cdev)
                for _ in range(3)]
        pix = torch.arange(len(shadow), dtype=torch.int32, device=cdev)
        L = torch.zeros((len(shadow), 4), dtype=torch.float32, device=cdev)
        d_hits2 = d_hits.view(-1, 32)
        d_bhits2 = d_bhits.view(-1, 32)

        def step_wavefront():
            for q in outq.values():
                q.Reset()
            wf.IntersectClosest(n_primary, q_primary, hits=d_hits2, **outq)
            for q in outq.values():
                q.Reset()
            wf.IntersectClosest(len(bounce), q_bounce, hits=d_bhits2, **outq)
            wf.IntersectShadow(len(shadow), q_shadow, spec[0], spec[1], spec[2], pix, L)

        step_wavefront()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_wavefront()
        barrier()
        wavefront_s = time.perf_counter() - t1
        # ... and with the hit -> SurfaceInteraction post-pass of both closest-hit stages
        # (Triangle::InteractionFromIntersection, which the reference's Intersect runs per hit)
        from nn_bvh_amd.interaction import ShadingMesh
        smesh = ShadingMesh(verts, tris, device=local_rank)
        d_intr = torch.empty(n_primary * 192, dtype=torch.uint8, device=cdev)

        def step_wavefront_intr():
            step_wavefront()
            smesh.interactions_device(d_hits2.data_ptr(), n_primary, d_intr.data_ptr(), ray_queue=q_primary,
                                      stream=stream)
            smesh.interactions_device(d_bhits2.data_ptr(), len(bounce), d_intr.data_ptr(), ray_queue=q_bounce,
                                      stream=stream)

        step_wavefront_intr()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_wavefront_intr()
        barrier()
        wavefront_intr_s = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([wavefront_s], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wavefront_s = float(t.item())

    allgather_ms = None
    if world > 1:
        # film-sample hand-off after the pass: one RCCL all-gather of the per-tile hit records
        g_src = d_hits if backend == "nccl" else d_hits.cpu()
        shard.all_gather_records(g_src, shard_bytes)  # warm-up (communicator setup)
        barrier()
        t1 = time.perf_counter()
        gathered = shard.all_gather_records(g_src, shard_bytes)
        barrier()
        allgather_ms = (time.perf_counter() - t1) * 1e3
        assert sum(g.numel() for g in gathered) == int(shard_bytes.sum())

    # HBM-side traffic of the closest-hit kernel per launch, from the committed rocprofv3 PMC
    # passes of this same command (profiles/; collected and corrected as MI355X_MICROARCH.md
    # §HBM prescribes).  Only quoted when the profile was taken at the same --spp.
    traffic = None
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic_r01.json")))
        if tr.get("spp") == args.spp and args.scene == "crown" and world == 1:
            traffic = round(float(tr["bytes"]))
    except (OSError, ValueError, KeyError):
        pass

    result = None
    if rank == 0:
        value = total_rays_per_step * args.steps / elapsed / 1e6
        result = {
            "metric": "Mray/s (closest-hit + any-hit)",
            "value": round(value, 2),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.scene}: {args.spp} spp wavefront pass per step, primary+bounce "
                            f"closest-hit and shadow any-hit, {rays_per_step} rays/step/GPU",
                "spp_per_step": args.spp,
                "geometry": source,
                "tree": args.tree,
                "triangles": int(len(tris)),
                "nodes": int(len(tree.nodes)),
                "rays_primary": int(n_primary),
                "rays_bounce": int(len(bounce)),
                "rays_shadow": int(len(shadow)),
                "parallelism": f"tile-sharded x{world}, BVH replicated",
            },
            "per_class_mrays": {
                "primary_closest": round(n_primary / ms_primary / 1e3, 2),
                "bounce_closest": round(len(bounce) / ms_bounce / 1e3, 2),
                "shadow_any": round(len(shadow) / ms_shadow / 1e3, 2),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "trace_kernel<closest>",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "alg_bytes_per_launch": round(bytes_closest / 2),
                "alg_bytes_per_ray": round(bytes_closest / (len(hits) + len(bhits)), 1),
                "mean_nodes_visited": round(float(hits["nodes_visited"].mean()), 2),
                "mean_prim_tests": round(float(hits["prim_tests"].mean()), 2),
                "avg_launch_ms": round(ms_closest / 2, 4),
            },
        }
        if overlapped_s is not None:
            result["overlapped_batches"] = {
                "value": round(total_rays_per_step * args.steps / overlapped_s / 1e6, 2),
                "unit": "Mray/s",
                "ms_per_step": round(overlapped_s / args.steps * 1e3, 4),
                "how": "same step as one nnbvh_trace_batches_device call (3 batches concurrent)",
            }
        if wavefront_s is not None:
            result["wavefront_queues"] = {
                "value": round(total_rays_per_step * args.steps / wavefront_s / 1e6, 2),
                "unit": "Mray/s",
                "ms_per_step": round(wavefront_s / args.steps * 1e3, 4),
                "how": "same step through nnbvh_wavefront_intersect_closest/_shadow: SOA queues in, "
                       "6 index queues + pixel radiance out, queue resets included",
                "with_surface_interactions_ms_per_step": round(wavefront_intr_s / args.steps * 1e3, 4),
            }
        if allgather_ms is not None:
            result["allgather_ms"] = round(allgather_ms, 3)

    # ---- CPU baseline: the oracle on this box's host cores; rank 0, N=1 only.  Bounded: the
    # same three batches of one step, traced `passes` times after a warm-up pass (about 15
    # core-seconds of work), on every core this process may use.
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_binding as ob
        cores = host_cores()
        n = min(args.cpu_sample, n_primary, len(bounce), len(shadow))
        sel = np.sort(np.random.default_rng(0).choice(min(n_primary, len(bounce), len(shadow)), n,
                                                      replace=False))
        batches = (primary[sel], bounce[sel], shadow[sel])

        def cpu_pass():
            a = ob.closest(tree.nodes, tree.ordered_prims, verts, batches[0], nthreads=cores)
            b = ob.closest(tree.nodes, tree.ordered_prims, verts, batches[1], nthreads=cores)
            c, _, _ = ob.any_hit(tree.nodes, tree.ordered_prims, verts, batches[2], nthreads=cores)
            return a, b, c

        c1, c2, o3 = cpu_pass()  # warm-up pass; its output is also the cross-check below
        times = []
        for _ in range(args.cpu_passes):
            t1 = time.perf_counter()
            cpu_pass()
            times.append(time.perf_counter() - t1)
        cpu_s = float(np.median(times))
        # the checker also checks: the sample must agree with what the GPU produced
        same = (c1.tobytes() == hits[sel].tobytes() and c2.tobytes() == bhits[sel].tobytes()
                and (o3 == d_occ.cpu().numpy()[sel]).all())
        result["cpu_baseline"] = {
            "value": round(3 * n / cpu_s / 1e6, 3),
            "unit": "Mray/s",
            "cores": cores,
            "kind": "port",
            "sample": f"{n} rays of each class (primary, bounce, shadow) of the step's own batches, "
                      f"oracle/nnbvh_oracle.c on {cores} threads, median of {args.cpu_passes} "
                      f"passes of {cpu_s:.2f}s",
            "matches_gpu": bool(same),
        }
    if rank == 0:
        print(json.dumps(result), flush=True)
    agg.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
