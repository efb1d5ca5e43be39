#!/usr/bin/env python3
"""bench.py — Mray/s of the HIP BVH traversal path on the crown workload (BASELINE.json).

One "step" = one wavefront pass of --spp samples per pixel (default 8, each sample its own
jitter stream) over the crown film through the hot path, all ray batches already resident in
HBM, one launch per ray class:
    closest-hit  over spp x 1000x1400 primary rays   (BVHAggregate::Intersect)
    closest-hit  over the diffuse-bounce rays of those hits            (bounce 1)
    closest-hit  over the diffuse-bounce rays of the bounce-1 hits     (bounce 2)
    any-hit      over the shadow rays of the primary hits  (BVHAggregate::IntersectP, tMax = 1-1e-4)
    RecordShadowRayResult -> L per pixel sample, UpdateFilm / RGBFilm::AddSample -> the film's
    4 doubles per pixel (film.h:239-255, 302-307)
The ray batches are synthetic, seeded and generated on the device before the timed region (nn_bvh_amd/raygen.py).
value = rays traced by all ranks / wall time of K steps (max over ranks).  Consecutive steps
trace different samples: --sample-sets (default 4) distinct sets of spp samples are resident and
used round-robin.  BASELINE's "1024 spp" is extrapolated from these passes as SURVEY.md §8d
prescribes (>= 64 M rays per class are timed; 1.4 G x depth rays are not traced).

Why 8 spp per launch: a launch's time is T(n) = ramp/drain + n / steady-state rate, and the
drain is the dependent-load chain of the longest ray in the batch (crown: V up to ~600 nodes,
about 0.55 ms) whatever n is (tools/scaling.py).  pbrt's wavefront integrator caps
its queues at ~1 M samples (wavefront/integrator.cpp:230-234) for the memory of other GPUs;
with 288 GB of HBM the natural MI355X design is fewer, larger launches (8 spp of the crown
film = 11.2 M rays = 0.7 GB of ray + hit records).  --spp 1 reproduces the 1 M-ray regime.

Multi-GPU: the BVH is replicated; each rank owns an interleaved set of 16x16
image tiles (Morton-ordered ray chunks), traces only its tiles' rays and accumulates only its
tiles' film pixels — no data-path collective inside the timed steps — and when the pass is
complete the per-tile film accumulators are all-gathered once (RCCL over xGMI; reported as
film_allgather_ms; the all-gather of the 32-B hit records is kept as a second figure).
Per-GPU work is fixed as N grows (weak scaling): with --gpus N the job is N x spp samples per
pixel and rank r traces its tiles of every sample, i.e. spp films' worth of rays per GPU.

`python bench.py --gpus N` with N > 1 starts its own ranks (torch.distributed.run, one per
GPU) before anything touches the GPU; launched under torchrun it uses the ranks it was given.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--scene crown] [--no-cpu-baseline]
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md §HBM)
SHADER_GHZ = 2.4       # MI355X peak engine clock
N_SIMD = 1024          # 256 CUs x 4 SIMD-32
# the reference's own CPU path, 8 threads, crown — primary closest / bounce closest /
# bounce any-hit.  Measured in the survey container (Xeon 2.1 GHz, 8 vCPU), NOT on this box.
REFERENCE_8T_MRAYS = {"primary_closest": 5.2, "bounce_closest": 3.1, "bounce_any": 4.2}


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="crown")
    ap.add_argument("--spp", type=int, default=8, help="samples per pixel traced per step")
    ap.add_argument("--sample-sets", type=int, default=4,
                    help="distinct sets of --spp samples kept resident and traced round-robin")
    ap.add_argument("--tree", default="sah", choices=["sah", "hlbvh", "middle", "equal", "nn", "sah_gpu", "hlbvh_gpu", "kd"],
                    help="tree builder: pbrt split methods (host), nn = greedy-SAH top levels of "
                         "machine_learning/nn_BVH.py finished by SAH and baked (BASELINE config 5), "
                         "*_gpu = the same sah / hlbvh tree built and baked on the device, kd = the "
                         "reference's KdTreeAggregate (aggregates.cpp:746-1161) on its own traversal kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--wavefront", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--no-wavefront", action="store_true",
                    help="skip the dependent form of the step (wavefront-queue entry points, one launch per "
                         "queue), reported as dependent_step_ms")
    ap.add_argument("--no-order-probe", action="store_true",
                    help="skip timing the step's traces with the rays in the reference integrator's sample-major order")
    ap.add_argument("--serial", action="store_true",
                    help="trace the three batches of a step as three launches (round-1 behaviour) instead "
                         "of one nnbvh_trace_batches_device call = ONE launch over all three")
    ap.add_argument("--ray-order", default="tile", choices=["tile", "pixel", "sample"],
                    help="order of a step's rays: pixel = the spp samples of a pixel adjacent, pixels in scanline "
                         "order (each XCD's share of the batch is then a region of the image); tile = the same with "
                         "the pixels in 4x4 tiles (a wavefront's 64 rays = 8 pixels of a 4x2 block); sample = spp "
                         "whole-image passes concatenated")
    ap.add_argument("--overlapped", action="store_true",
                    help="also time the step with the three batches as concurrent launches on the "
                         "library's internal streams (fused_batches off)")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000,
                    help="rays per class timed on the host cores for cpu_baseline")
    ap.add_argument("--cpu-passes", type=int, default=5)
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n_gpus):
    """Start one rank per GPU as child processes and return their exit code.  Called before torch
    or HIP is imported: this parent never touches the GPU (a process that has must not exec)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log(f"[bench] --gpus {n_gpus} without a launcher: starting {n_gpus} ranks: {' '.join(cmd[1:8])} ...")
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, rank, world, out=sys.stdout):
    """NNBVH_BENCH_DRYRUN=1: rendezvous only (gloo, no GPU) — the CPU test of the launch path."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        total = 1.0
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rank_sum": total, "steps": args.steps,
                          "warmup": args.warmup}), file=out, flush=True)
    return 0


def profile_counters(prefix, pattern="r*_final*"):
    """PMC figures of the dominant kernel (name prefix) from the newest committed rocprofv3 summary that holds
    it (profiles/<pattern>/summary.json; collected as MI355X_MICROARCH.md prescribes: separate --pmc passes of
    this same command, tools/profile_bench.sh).  None if no summary is present."""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern, "summary.json")))
    for path in reversed(paths):
        try:
            summ = json.load(open(path))
            name = next(k for k in summ["counters_per_launch_mean"] if k.startswith(prefix))
            c = summ["counters_per_launch_mean"][name]
            ks = next(k for k in summ["kernel_stats"] if prefix in k["Name"])
            cycles = float(ks["AverageNs"]) * SHADER_GHZ
            return {
                "source": os.path.relpath(path, ROOT),
                "spp": summ.get("spp"),
                "kernel": name,
                "avg_launch_ms": round(float(ks["AverageNs"]) / 1e6, 4),
                "valu_insts_per_launch": c["SQ_INSTS_VALU"],
                # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (tools/halfwave_probe.hip)
                "frac": round(c["SQ_INSTS_VALU"] * 2.0 / (N_SIMD * cycles), 4),
                "lanes_per_valu": round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"], 2),
                "salu_per_valu": round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 3),
                "wait_frac_of_wave_cycles": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                "l2_hit": round(c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3),
                "hbm_bytes_per_launch": round(c["TCC_EA0_RDREQ_sum"] * 128 - c.get("TCC_EA0_RDREQ_32B_sum", 0) * 96
                                              - c.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + c["WRITE_SIZE"] * 1024),
            }
        except (OSError, ValueError, KeyError, StopIteration):
            continue
    return None


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
        sys.exit(2)
    # stdout carries ONE line, the result: whatever libraries print there (RCCL's and gloo's start-up banners)
    # goes to stderr instead — file descriptor 1 is pointed at stderr and the result is written to the real stdout
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    result_out = os.fdopen(result_fd, "w")
    if os.environ.get("NNBVH_BENCH_DRYRUN") == "1":
        rc = dry_run(args, rank, world, result_out)
        sys.exit(rc)

    import numpy as np
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
    cdev = torch.device("cuda", local_rank)

    from nn_bvh_amd import BVHAggregate, build_tree, make_prims, scene, shard
    from nn_bvh_amd._lib import HIT_DTYPE, check, lib
    from nn_bvh_amd.film import Film

    def max_over_ranks(x):
        if world == 1 or x is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    t0 = time.time()
    verts, tris, source = scene.load_scene(args.scene)
    prims = make_prims(tris)
    kd = None
    if args.tree == "kd":
        if args.overlapped or args.wavefront:
            sys.exit("bench.py: --tree kd supports the plain step only")
        from nn_bvh_amd.aggregate import BuiltTree
        from nn_bvh_amd.kdtree import KdTreeAggregate, build_kd_tree
        kd = build_kd_tree(prims, verts, where="gpu", device=local_rank)  # nnbvh_kd_build_create_gpu: crown in ~0.2 s
        tree = BuiltTree(kd.nodes, kd.prim_indices, kd.depth)
    elif args.tree == "nn":
        from nn_bvh_amd import nn_tree
        from nn_bvh_amd.aggregate import BuiltTree
        (nn_nodes, nn_ordered), _ = nn_tree.greedy_sah_tree(verts, tris, levels=4)
        tree = BuiltTree(nn_nodes, nn_ordered, -1)
    elif args.tree in ("sah_gpu", "hlbvh_gpu"):
        # built and baked on the device; the host copy of the same tree only serves the oracle leg
        tree = build_tree(prims, verts, 4, args.tree[:-4])
    else:
        tree = build_tree(prims, verts, 4, args.tree)
    if kd is not None:
        agg = KdTreeAggregate.from_tree(kd.nodes, kd.prim_indices, prims, verts, kd.bounds, device=local_rank)
        agg.info = {"depth": kd.depth, "grid_blocks": -1, "stack_window": 8}
    elif args.tree in ("sah_gpu", "hlbvh_gpu"):
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
    # film per step; rank r traces its interleaved 16x16 tiles of every sample, i.e. spp films'
    # worth of rays per GPU whatever N is (weak scaling).  --sample-sets distinct sets of samples
    # are resident; step i traces set i mod sets.
    cam_name = args.scene if args.scene in scene.CAMERAS else "crown"
    xres, yres = scene.CAMERAS[cam_name][4], scene.CAMERAS[cam_name][5]
    _, px, py = scene.camera_rays(cam_name, seed=1, sample=0, return_pixels=True)
    index_lists = [shard.shard_indices(px, py, xres, world, r) for r in range(world)]
    if args.ray_order == "tile":  # every rank's pixels in 4x4 tiles (row order of tiles, scanlines inside)
        index_lists = [ix[np.lexsort((px[ix], py[ix], px[ix] // 4, py[ix] // 4))] for ix in index_lists]
    mine = index_lists[rank]
    n_slots = len(mine)                     # pixel slots of this rank = film pixels it owns
    passes = world * args.spp               # samples per pixel and step
    shard_bytes = passes * shard.shard_counts(px, py, xres, world) * 32  # hit bytes/rank
    stream = torch.cuda.current_stream().cuda_stream

    def dev(a):
        return torch.from_numpy(a.view(np.uint8).reshape(-1)).to(cdev)

    lo, hi = verts.min(0), verts.max(0)
    t_gen = time.time()
    # the batches are generated on the device (nn_bvh_amd/raygen.py: scene.py's recipes as vectorised passes over
    # device arrays), so a rank does not spend tens of seconds of host time before its first step
    from nn_bvh_amd import raygen
    ds = raygen.DeviceScene(verts, tris, cdev)
    d_pxm = torch.from_numpy(px[mine].astype(np.float64)).to(cdev)
    d_pym = torch.from_numpy(py[mine].astype(np.float64)).to(cdev)
    sets = []
    for k in range(max(1, args.sample_sets)):
        per_pass = [ds.camera_rays(cam_name, d_pxm, d_pym, seed=1, sample=k * passes + s_idx) for s_idx in range(passes)]
        if args.ray_order != "sample":  # ray r = sample r % passes of slot r // passes
            primary_t = torch.stack(per_pass, 1).reshape(-1, 8).contiguous()
            ar = torch.arange(len(primary_t), device=cdev)
            sample_slot = (ar % passes) * n_slots + ar // passes
        else:                          # ray r = sample r // n_slots of slot r % n_slots
            primary_t = torch.cat(per_pass)
            sample_slot = torch.arange(len(primary_t), device=cdev)
        del per_pass
        n_primary = len(primary_t)
        d_primary = primary_t.view(torch.uint8).reshape(-1)
        d_hits = torch.empty(n_primary * 32, dtype=torch.uint8, device=cdev)
        agg.intersect_device(d_primary.data_ptr(), d_hits.data_ptr(), n_primary, stream)
        torch.cuda.synchronize()
        bounce_t, hit_idx = ds.bounce_rays(primary_t, d_hits, seed=[2, rank, k])
        # bounce 2: the diffuse bounce of the bounce-1 hits (the incoherent regime)
        d_bounce = bounce_t.view(torch.uint8).reshape(-1)
        d_bhits = torch.empty(len(bounce_t) * 32, dtype=torch.uint8, device=cdev)
        agg.intersect_device(d_bounce.data_ptr(), d_bhits.data_ptr(), len(bounce_t), stream)
        torch.cuda.synchronize()
        bounce2_t, _ = ds.bounce_rays(bounce_t, d_bhits, seed=[4, rank, k])
        if args.scene == "crown":  # towards the scene's six area-light quads (crown.pbrt:26-102)
            shadow_t, _ = ds.shadow_rays(primary_t, d_hits, seed=[3, rank, k], quads=scene.CROWN_LIGHT_QUADS)
        else:
            shadow_t, _ = ds.shadow_rays(primary_t, d_hits, seed=[3, rank, k],
                                         box=(lo + (hi - lo) * [0.3, 0.9, 0.3], lo + (hi - lo) * [0.7, 1.0, 0.7]))
        n_bounce, n_bounce2, n_shadow = len(bounce_t), len(bounce2_t), len(shadow_t)
        gen = torch.Generator(device=cdev).manual_seed(100 + 10 * rank + k)
        first = k == 0
        st = {
            "primary": raygen.as_records(primary_t) if first else None,
            "bounce": raygen.as_records(bounce_t) if first else None,
            "bounce2": raygen.as_records(bounce2_t) if first else None,
            "shadow": raygen.as_records(shadow_t) if first else None,
            "hits": d_hits.cpu().numpy().view(HIT_DTYPE) if first else None,
            "n_primary": n_primary, "n_bounce": n_bounce, "n_bounce2": n_bounce2, "n_shadow": n_shadow,
            "d_primary": d_primary, "d_hits": d_hits, "d_bounce": d_bounce,
            "d_shadow": shadow_t.view(torch.uint8).reshape(-1),
            "d_bhits": d_bhits, "d_bounce2": bounce2_t.view(torch.uint8).reshape(-1),
            "d_b2hits": torch.empty(n_bounce2 * 32, dtype=torch.uint8, device=cdev),
            "d_occ": torch.empty(n_shadow, dtype=torch.uint8, device=cdev),
            # ShadowRayWorkItem payload (workitems.soa:77-83): Ld, r_u, r_l per shadow ray, the pixel
            # sample it belongs to; PixelSampleState::L per pixel sample (SampledSpectrum = 4 floats)
            # (index into L, which stays [pass][slot] whatever the order of the rays)
            "d_pix": sample_slot[hit_idx].to(torch.int32),
            "d_Ld": torch.rand((n_shadow, 4), generator=gen, device=cdev) * 2.0,
            "d_ru": torch.rand((n_shadow, 4), generator=gen, device=cdev) + 0.5,
            "d_rl": torch.rand((n_shadow, 4), generator=gen, device=cdev) + 0.5,
            "d_L": torch.zeros((n_primary, 4), dtype=torch.float32, device=cdev),
            "d_w": torch.rand(n_primary, generator=gen, device=cdev) + 0.5,  # filterWeight per sample
        }
        del primary_t, bounce_t, bounce2_t, shadow_t, sample_slot, hit_idx
        sets.append(st)
    s0 = sets[0]
    rays_per_step = s0["n_primary"] + s0["n_bounce"] + s0["n_bounce2"] + s0["n_shadow"]
    if rank == 0:
        log(f"[bench] {len(sets)} sample sets x {passes} samples generated in {time.time() - t_gen:.1f}s; "
            f"{rays_per_step} rays/step/GPU")

    # the film of the whole image; this rank accumulates the pixels of its tiles
    film = Film(xres, yres, device=local_rank)
    d_px = torch.from_numpy(px[mine].astype(np.int32)).to(cdev)
    d_py = torch.from_numpy(py[mine].astype(np.int32)).to(cdev)
    L = lib()

    def vp(t):
        import ctypes
        return ctypes.c_void_p(t.data_ptr())

    def film_stage(st):
        import ctypes
        st["d_L"].zero_()
        check(L.nnbvh_wavefront_record_shadow_device(vp(st["d_occ"]), st["n_shadow"], None, vp(st["d_Ld"]),
                                                     vp(st["d_ru"]), vp(st["d_rl"]), vp(st["d_pix"]),
                                                     vp(st["d_L"]), st["n_primary"], local_rank,
                                                     ctypes.c_void_p(stream)), "record_shadow")
        # sensor RGB = the first three of L's four wavelength samples (the spectral sensor model,
        # film.h:95-100, is the caller's); slot i of every pass is pixel mine[i]
        film.add_samples_device(d_px, d_py, st["d_L"], st["d_w"], n_slots, passes, rgb_stride=4, stream=stream)

    fused = not args.serial and kd is None

    def batches_of(st):
        # the waves drain the batches in list order, so the launch's tail is the drain of the LAST batch's longest
        # rays: the coherent primary rays go last (tools/batch_order_probe.py: 9.01 ms against 9.27 ms primary-first)
        return [("any", st["d_shadow"].data_ptr(), st["n_shadow"], st["d_occ"].data_ptr()),
                ("closest", st["d_bounce2"].data_ptr(), st["n_bounce2"], st["d_b2hits"].data_ptr()),
                ("closest", st["d_bounce"].data_ptr(), st["n_bounce"], st["d_bhits"].data_ptr()),
                ("closest", st["d_primary"].data_ptr(), st["n_primary"], st["d_hits"].data_ptr())]

    def trace_stage(st):
        if fused:  # one launch: the wavefronts drain primary, bounce-1, bounce-2 and shadow rays one batch after the other
            agg.trace_batches_device(batches_of(st), stream)
            return
        agg.intersect_device(st["d_primary"].data_ptr(), st["d_hits"].data_ptr(), st["n_primary"], stream)
        agg.intersect_device(st["d_bounce"].data_ptr(), st["d_bhits"].data_ptr(), st["n_bounce"], stream)
        agg.intersect_device(st["d_bounce2"].data_ptr(), st["d_b2hits"].data_ptr(), st["n_bounce2"], stream)
        agg.intersect_p_device(st["d_shadow"].data_ptr(), st["d_occ"].data_ptr(), st["n_shadow"], stream=stream)

    def step(i):
        st = sets[i % len(sets)]
        trace_stage(st)
        film_stage(st)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    film.clear(stream)
    barrier()
    t_start = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t_start
    elapsed = max_over_ranks(elapsed)
    # every set has its own bounce / shadow counts: rays actually traced in the K steps
    def rays_of(st):
        return st["n_primary"] + st["n_bounce"] + st["n_bounce2"] + st["n_shadow"]
    rays_timed = float(sum(rays_of(sets[i % len(sets)]) for i in range(args.steps)))
    if world > 1:
        tot = torch.tensor([rays_timed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        rays_timed = float(tot.item())

    # ---- the film all-gather that ends the pass (RCCL over xGMI under nccl): every rank packs the
    # accumulators of its tiles, one all_gather_into_tensor, the others' tiles are unpacked
    film_allgather_ms = film_bytes = None
    allgather_ms = None
    if world > 1:
        if backend == "nccl":
            d_lists = [torch.from_numpy((py[ix].astype(np.int64) * xres + px[ix]).astype(np.int32)).to(cdev)
                       for ix in index_lists]
            film.all_gather_tiles(d_lists, rank, stream)  # warm-up (communicator setup), idempotent
            barrier()
            t1 = time.perf_counter()
            film_bytes = film.all_gather_tiles(d_lists, rank, stream)
            barrier()
            film_allgather_ms = max_over_ranks((time.perf_counter() - t1) * 1e3)
        else:  # rehearsal over gloo: the same protocol on a host copy of the accumulators
            pix = torch.from_numpy(film.read())
            lin = [torch.from_numpy(py[ix].astype(np.int64) * xres + px[ix]) for ix in index_lists]
            barrier()
            t1 = time.perf_counter()
            shard.all_gather_film(pix, lin, rank)
            barrier()
            film_allgather_ms = max_over_ranks((time.perf_counter() - t1) * 1e3)
            film_bytes = max(len(ix) for ix in index_lists) * 32
        # second figure: the 32-B hit records of one step
        g_src = s0["d_hits"] if backend == "nccl" else s0["d_hits"].cpu()
        shard.all_gather_records(g_src, shard_bytes)
        barrier()
        t1 = time.perf_counter()
        gathered = shard.all_gather_records(g_src, shard_bytes)
        barrier()
        allgather_ms = max_over_ranks((time.perf_counter() - t1) * 1e3)
        assert sum(g.numel() for g in gathered) == int(shard_bytes.sum())
    weight_sum = float(film.read()[:, 3].sum()) if rank == 0 else 0.0

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
    n_primary, n_bounce, n_bounce2, n_shadow = s0["n_primary"], s0["n_bounce"], s0["n_bounce2"], s0["n_shadow"]
    ms_primary = time_kernel(lambda: agg.intersect_device(s0["d_primary"].data_ptr(), s0["d_hits"].data_ptr(),
                                                          n_primary, stream), reps)
    ms_bounce = time_kernel(lambda: agg.intersect_device(s0["d_bounce"].data_ptr(), s0["d_bhits"].data_ptr(),
                                                         n_bounce, stream), reps)
    ms_bounce2 = time_kernel(lambda: agg.intersect_device(s0["d_bounce2"].data_ptr(), s0["d_b2hits"].data_ptr(),
                                                          n_bounce2, stream), reps)
    ms_shadow = time_kernel(lambda: agg.intersect_p_device(s0["d_shadow"].data_ptr(), s0["d_occ"].data_ptr(),
                                                           n_shadow, stream=stream), reps)
    ms_film = time_kernel(lambda: film_stage(s0), reps)
    hits = s0["hits"]
    primary, bounce, bounce2, shadow = s0["primary"], s0["bounce"], s0["bounce2"], s0["shadow"]
    bhits = s0["d_bhits"].cpu().numpy().view(HIT_DTYPE)
    b2hits = s0["d_b2hits"].cpu().numpy().view(HIT_DTYPE)

    # algorithmic bytes: 32 in + 32*V + 48*T + 32 out per closest-hit ray (a
    # KdTreeNode is 8 B, and a leaf primitive costs its 4-B index besides the 48 B of the triangle)
    node_bytes, prim_bytes = (8.0, 52.0) if kd is not None else (32.0, 48.0)

    def alg_bytes(h):
        return 64.0 * len(h) + node_bytes * h["nodes_visited"].sum(dtype=np.int64) + \
            prim_bytes * h["prim_tests"].sum(dtype=np.int64)
    bytes_closest = alg_bytes(hits) + alg_bytes(bhits) + alg_bytes(b2hits)   # the three launches of the kernel
    ms_closest = ms_primary + ms_bounce + ms_bounce2
    achieved = bytes_closest / (ms_closest * 1e-3) / 1e9         # GB/s over the kernel's launches
    roof_kernel = "kd_trace_kernel<closest>" if kd is not None else "trace_kernel<closest>"
    roof_bytes_per_launch, roof_ms = bytes_closest / 3, ms_closest / 3
    if fused:
        # the dominant kernel is the one launch of a step: trace_kernel<3> over all three batches.  Its
        # algorithmic bytes add the shadow rays' 36 + 32 V + 48 T (V, T from a counting any-hit launch)
        d_v = torch.empty(n_shadow, dtype=torch.int32, device=cdev)
        d_t = torch.empty(n_shadow, dtype=torch.int32, device=cdev)
        agg.intersect_p_device(s0["d_shadow"].data_ptr(), s0["d_occ"].data_ptr(), n_shadow, d_v.data_ptr(),
                               d_t.data_ptr(), stream)
        torch.cuda.synchronize()
        bytes_shadow = 36.0 * n_shadow + node_bytes * float(d_v.sum(dtype=torch.int64).item()) + \
            prim_bytes * float(d_t.sum(dtype=torch.int64).item())
        ms_fused = time_kernel(lambda: trace_stage(s0), reps)
        roof_kernel = "trace_kernel<3> (one launch: primary + bounce-1 + bounce-2 closest-hit, shadow any-hit)"
        roof_bytes_per_launch, roof_ms = bytes_closest + bytes_shadow, ms_fused
        achieved = roof_bytes_per_launch / (roof_ms * 1e-3) / 1e9

    # ---- the same step through nnbvh_trace_batches_device: the three batches run concurrently
    # on the library's internal streams, so each launch's drain overlaps the others' work.
    def step_overlapped():
        agg.trace_batches_device(batches_of(s0), stream)

    overlapped_s = None
    if args.overlapped:
        agg.set_option("fused_batches", 0)
        step_overlapped()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_overlapped()
        barrier()
        overlapped_s = max_over_ranks(time.perf_counter() - t1)
        agg.set_option("fused_batches", 1)

    # ---- the DEPENDENT form of the step: the same batches through the wavefront-queue entry points (SOA ray
    # queues with device-side sizes in, index queues and pixel radiance out), one launch per queue as an
    # integrator whose bounce rays depend on the previous hits must issue them.  Reported next to `value`
    # (dependent_step_ms): the headline's one launch per step needs batches that are independent.
    wavefront_s = wavefront_intr_s = wavefront_per_queue_s = None
    if not args.no_wavefront and kd is None:
        from nn_bvh_amd.wavefront import RayQueue, WavefrontAggregate, WorkQueue
        from nn_bvh_amd._lib import CLOSEST_QUEUES
        wf = WavefrontAggregate(agg, np.zeros(len(tris), np.uint8))
        q_primary, q_bounce = RayQueue.from_records(primary, cdev), RayQueue.from_records(bounce, cdev)
        q_bounce2 = RayQueue.from_records(bounce2, cdev)
        q_shadow = RayQueue.from_records(shadow, cdev, shadow=True)
        outq = {k: WorkQueue(n_primary, cdev) for k in CLOSEST_QUEUES}
        pix = torch.arange(n_shadow, dtype=torch.int32, device=cdev)
        Lw = torch.zeros((n_shadow, 4), dtype=torch.float32, device=cdev)
        d_hits2 = s0["d_hits"].view(-1, 32)
        d_bhits2 = s0["d_bhits"].view(-1, 32)
        d_b2hits2 = s0["d_b2hits"].view(-1, 32)

        def reset_queues():
            for q in outq.values():
                q.Reset()

        def step_wavefront():
            # what the wavefront render loop issues (wavefront/integrator.cpp): depth 0's closest-hit pass; then
            # the shadow rays of depth 0 WITH the closest-hit pass of depth 1 (both come out of depth 0's shading
            # and neither reads the other's results: one launch); then depth 2's closest-hit pass
            reset_queues()
            wf.IntersectClosest(n_primary, q_primary, hits=d_hits2, **outq)
            reset_queues()
            wf.IntersectClosestAndShadow(n_bounce, q_bounce, n_shadow, q_shadow, s0["d_Ld"], s0["d_ru"], s0["d_rl"], pix,
                                         Lw, hits=d_bhits2, **outq)
            reset_queues()
            wf.IntersectClosest(n_bounce2, q_bounce2, hits=d_b2hits2, **outq)

        def step_wavefront_one_launch_per_queue():
            for n_q, q_in, h_out in ((n_primary, q_primary, d_hits2), (n_bounce, q_bounce, d_bhits2),
                                     (n_bounce2, q_bounce2, d_b2hits2)):
                reset_queues()
                wf.IntersectClosest(n_q, q_in, hits=h_out, **outq)
            wf.IntersectShadow(n_shadow, q_shadow, s0["d_Ld"], s0["d_ru"], s0["d_rl"], pix, Lw)

        step_wavefront_one_launch_per_queue()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_wavefront_one_launch_per_queue()
        barrier()
        wavefront_per_queue_s = max_over_ranks(time.perf_counter() - t1)

        step_wavefront()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_wavefront()
        barrier()
        wavefront_s = max_over_ranks(time.perf_counter() - t1)
        # ... and with the hit -> SurfaceInteraction post-pass of the closest-hit stages
        # (Triangle::InteractionFromIntersection, which the reference's Intersect runs per hit)
        from nn_bvh_amd.interaction import ShadingMesh
        smesh = ShadingMesh(verts, tris, device=local_rank)
        d_intr = torch.empty(n_primary * 192, dtype=torch.uint8, device=cdev)

        def step_wavefront_intr():
            step_wavefront()
            for h_out, n_q, q_in in ((d_hits2, n_primary, q_primary), (d_bhits2, n_bounce, q_bounce),
                                     (d_b2hits2, n_bounce2, q_bounce2)):
                smesh.interactions_device(h_out.data_ptr(), n_q, d_intr.data_ptr(), ray_queue=q_in, stream=stream)

        step_wavefront_intr()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_wavefront_intr()
        barrier()
        wavefront_intr_s = max_over_ranks(time.perf_counter() - t1)
        del q_primary, q_bounce, q_bounce2, q_shadow, outq, d_intr, Lw, pix

    # ---- the step's traces with the SAME rays in the order the reference's wavefront integrator forms them:
    # one sample of every pixel per pass, the passes one after the other (wavefront/integrator.cpp:231-236,
    # 336-368), instead of this bench's pixel-major order.  The order decides nothing but which lane traces
    # which ray; it is a caller-side choice, so both figures are printed.
    order_probe = None
    if not args.no_order_probe and args.ray_order != "sample" and fused and rank == 0 and world == 1:
        def sample_major(idx_in_parent, n_parent_slots):
            """order of a compacted batch whose ray j descends from pixel-major ray idx_in_parent[j]"""
            key = (idx_in_parent % passes).astype(np.int64) * n_parent_slots + idx_in_parent // passes
            return np.argsort(key, kind="stable")
        root_of_primary = np.arange(n_primary)
        hit_idx = np.nonzero(hits["prim"] >= 0)[0]                     # bounce-1 / shadow ray j <- primary hit_idx[j]
        root_of_bounce = root_of_primary[hit_idx]
        root_of_bounce2 = root_of_bounce[np.nonzero(bhits["prim"] >= 0)[0]]
        perms = [sample_major(r, n_slots) for r in (root_of_primary, root_of_bounce, root_of_bounce2, root_of_bounce)]
        d_alt = [dev(a[pm]) for a, pm in zip((primary, bounce, bounce2, shadow), perms)]
        alt = [("any", d_alt[3].data_ptr(), n_shadow, s0["d_occ"].data_ptr()),
               ("closest", d_alt[2].data_ptr(), n_bounce2, s0["d_b2hits"].data_ptr()),
               ("closest", d_alt[1].data_ptr(), n_bounce, s0["d_bhits"].data_ptr()),
               ("closest", d_alt[0].data_ptr(), n_primary, s0["d_hits"].data_ptr())]
        ms_alt = time_kernel(lambda: agg.trace_batches_device(alt, stream), reps)
        ms_own = time_kernel(lambda: trace_stage(s0), reps)
        order_probe = {
            "this_order_trace_ms": round(ms_own, 4), "this_order_mrays": round(rays_per_step / ms_own / 1e3, 1),
            "reference_order_trace_ms": round(ms_alt, 4),
            "reference_order_mrays": round(rays_per_step / ms_alt / 1e3, 1),
            "how": "the same four batches of sample set 0 (traces only, one launch), rays permuted into the order of "
                   "the reference's wavefront integrator: one sample of every pixel per pass, passes concatenated "
                   "(wavefront/integrator.cpp:231-236, 336-368); `value` uses --ray-order " + args.ray_order +
                   ", a caller-side ordering the reference does not perform",
        }
        del d_alt
        agg.trace_batches_device(batches_of(s0), stream)  # restore set 0's own results for the checks below
        torch.cuda.synchronize()

    # What binds the kernel, from the committed rocprofv3 PMC passes of this same command
    # (profiles/; collected and corrected as MI355X_MICROARCH.md §HBM prescribes).  Only quoted when
    # the profile was taken at the same --spp on crown at N=1.
    if kd is not None:
        prof = profile_counters("kd_trace_kernel<0", "r*_kd_" + args.scene + "*")
    else:
        prof = profile_counters("trace_kernel<3" if fused else "trace_kernel<0")
    if prof is not None and not (prof.get("spp") in (None, args.spp) and args.scene == "crown" and world == 1):
        prof = None
    traffic = prof["hbm_bytes_per_launch"] if prof else None
    valu_peak = N_SIMD * SHADER_GHZ / 2.0  # G wave-instructions/s: one wave64 VALU instruction per SIMD-32 per 2 cycles

    result = None
    if rank == 0:
        value = rays_timed / elapsed / 1e6
        total_spp = world * args.spp
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
                "workload": f"{args.scene}, {xres}x{yres}: one wavefront pass of {total_spp} spp per step "
                            f"({args.spp} spp per GPU-film) — primary, bounce-1 and bounce-2 closest-hit, shadow "
                            f"any-hit towards the light quads (batches pre-generated, hence independent: ONE launch; the "
                            f"dependent form is dependent_step), rays in --ray-order {args.ray_order} (a caller-side "
                            f"order the reference's integrator does not form: ray_order_probe), "
                            f"RecordShadowRayResult + RGBFilm::AddSample into a "
                            f"4-doubles-per-pixel film; {args.steps} steps rotating over {len(sets)} distinct "
                            f"sample sets; BASELINE's 1024 spp = {1024 // max(1, total_spp)} such passes, "
                            f"extrapolated per SURVEY §8d (>= 64 M rays per class timed), not traced",
                "spp_per_step": args.spp,
                "ray_order": {"tile": "pixel-major, pixels in 4x4 tiles: the spp samples of a pixel are adjacent in every "
                                      "batch, a wavefront's 64 rays cover a 4x2 block of pixels",
                              "pixel": "pixel-major: the spp samples of a pixel are adjacent in every batch, pixels in "
                                       "scanline order",
                              "sample": "sample-major: spp whole-image passes concatenated"}[args.ray_order],
                "sample_sets": len(sets),
                "geometry": source,
                "tree": args.tree,
                "triangles": int(len(tris)),
                "nodes": int(len(tree.nodes)),
                "rays_primary": int(n_primary),
                "rays_bounce": int(n_bounce),
                "rays_bounce2": int(n_bounce2),
                "rays_shadow": int(n_shadow),
                "rays_per_step_per_gpu": int(rays_per_step),
                "parallelism": f"tile-sharded x{world}, BVH replicated, film all-gather after the pass",
                "launches_per_step": "1 (nnbvh_trace_batches_device, mode-3 kernel)" if fused else "4",
            },
            "per_class_mrays": {
                "primary_closest": round(n_primary / ms_primary / 1e3, 2),
                "bounce_closest": round(n_bounce / ms_bounce / 1e3, 2),
                "bounce2_closest": round(n_bounce2 / ms_bounce2 / 1e3, 2),
                "shadow_any": round(n_shadow / ms_shadow / 1e3, 2),
            },
            "film": {
                "pixels": xres * yres, "bytes_per_pixel": 32,
                "stage_ms_per_step": round(ms_film, 4),
                "weight_sum": weight_sum,
                "how": "RecordShadowRayResult -> L, then UpdateFilm/RGBFilm::AddSample of every pixel "
                       "sample in sample order (inside the timed step)",
            },
            "roofline": {
                # What binds the dominant kernel: the issue of vector instructions (wavefront-instructions
                # per second against one wave64 instruction per SIMD-32 per 2 cycles), from the committed
                # rocprofv3 PMC passes of this same command; `avg_launch_ms` is the live HIP-event time of this
                # run.  The SURVEY §8d "algorithmic bytes" figure is kept under `alg_hbm`: it prices every node
                # re-read that L1/L2 serve, so it is not a physical bound and may exceed 1.
                "bound": "valu_issue" if prof else "valu_issue (no committed PMC profile for this configuration)",
                "kernel": roof_kernel,
                "achieved": None if prof is None else round(prof["valu_insts_per_launch"] / (prof["avg_launch_ms"] * 1e-3) / 1e9, 1),
                "peak": round(valu_peak, 1),
                "unit": "G wave-instr/s",
                "frac": None if prof is None else prof["frac"],
                "traffic": traffic,
                "avg_launch_ms": round(roof_ms, 4),
                "lanes_per_valu": None if prof is None else prof["lanes_per_valu"],
                "salu_per_valu": None if prof is None else prof["salu_per_valu"],
                "wait_frac_of_wave_cycles": None if prof is None else prof["wait_frac_of_wave_cycles"],
                "l2_hit": None if prof is None else prof["l2_hit"],
                "profile": None if prof is None else {"source": prof["source"], "avg_launch_ms": prof["avg_launch_ms"],
                                                      "kernel": prof["kernel"]},
                "how": "frac = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x launch cycles at 2.4 GHz); lanes = "
                       "SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64); traffic = TCC_EA0_RDREQ by size + "
                       "WRITE_SIZE, bytes per launch",
                "hbm_physical": None if prof is None else {
                    "achieved": round(prof["hbm_bytes_per_launch"] / (prof["avg_launch_ms"] * 1e-3) / 1e9, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(prof["hbm_bytes_per_launch"] / (prof["avg_launch_ms"] * 1e-3) / 1e9
                                  / HBM_PEAK_GBS, 4),
                },
                "alg_hbm": {
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "alg_bytes_per_launch": round(roof_bytes_per_launch),
                    "alg_bytes_per_ray": round(bytes_closest / (len(hits) + len(bhits) + len(b2hits)), 1),
                    "mean_nodes_visited": round(float(hits["nodes_visited"].mean()), 2),
                    "mean_prim_tests": round(float(hits["prim_tests"].mean()), 2),
                    "note": "SURVEY §8d: (64 + 32 V + 48 T) bytes per closest-hit ray, (36 + 32 V + 48 T) per any-hit ray, "
                            "over the kernel's time; cache-served re-reads included, hence not a bound",
                },
            },
        }
        if order_probe is not None:
            result["ray_order_probe"] = order_probe
        if world > 1:
            result["rccl_ranks_seen"] = int(dist.get_world_size()) if backend == "nccl" else 0
        if overlapped_s is not None:
            result["overlapped_batches"] = {
                "value": round(rays_per_step * world * args.steps / overlapped_s / 1e6, 2),
                "unit": "Mray/s",
                "ms_per_step": round(overlapped_s / args.steps * 1e3, 4),
                "how": "the three traces of sample set 0 as concurrent launches on internal streams",
            }
        if wavefront_s is not None:
            result["dependent_step"] = {
                "value": round(rays_per_step * world * args.steps / wavefront_s / 1e6, 2),
                "unit": "Mray/s",
                "ms_per_step": round(wavefront_s / args.steps * 1e3, 4),
                "how": "the same four batches through the wavefront-queue entry points in the order an integrator "
                       "whose bounce rays depend on the previous hits can issue them: closest(depth 0); shadow(depth "
                       "0) + closest(depth 1) in one launch (nnbvh_wavefront_intersect_closest_and_shadow: both "
                       "queues come out of depth 0's shading); closest(depth 2).  SOA queues in, 6 index queues + "
                       "pixel radiance out, queue resets included; `value` is the one-launch form, which needs "
                       "independent batches",
                "with_surface_interactions_ms_per_step": round(wavefront_intr_s / args.steps * 1e3, 4),
                "one_launch_per_queue_ms_per_step": round(wavefront_per_queue_s / args.steps * 1e3, 4),
            }
        if film_allgather_ms is not None:
            result["film_allgather_ms"] = round(film_allgather_ms, 3)
            result["film_allgather_bytes_per_rank"] = int(film_bytes)
            result["film_allgather_backend"] = "rccl" if backend == "nccl" else backend + " (rehearsal, host copy)"
            result["ms_per_step_incl_film_allgather"] = round(
                (elapsed * 1e3 + film_allgather_ms) / args.steps, 4)
        if allgather_ms is not None:
            result["allgather_ms"] = round(allgather_ms, 3)

    # ---- CPU baseline: the oracle on this box's host cores; rank 0, N=1 only.  Bounded: the
    # same three batches of one step, traced `passes` times after a warm-up pass (about 15
    # core-seconds of work), on every core this process may use.
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_binding as ob
        cores = host_cores()
        n = min(args.cpu_sample, n_primary, n_bounce, n_shadow)
        sel = np.sort(np.random.default_rng(0).choice(min(n_primary, n_bounce, n_shadow), n, replace=False))
        batches = (primary[sel], bounce[sel], shadow[sel])

        def cpu_pass():
            if kd is not None:
                a = ob.kd_closest(kd.nodes, kd.prim_indices, prims, verts, kd.bounds, batches[0], nthreads=cores)
                b = ob.kd_closest(kd.nodes, kd.prim_indices, prims, verts, kd.bounds, batches[1], nthreads=cores)
                c, _, _ = ob.kd_any_hit(kd.nodes, kd.prim_indices, prims, verts, kd.bounds, batches[2], nthreads=cores)
                return a, b, c
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
                and (o3 == s0["d_occ"].cpu().numpy()[sel]).all())
        result["cpu_baseline"] = {
            "value": round(3 * n / cpu_s / 1e6, 3),
            "unit": "Mray/s",
            "cores": cores,
            "kind": "port",
            "sample": f"{n} rays of each class (primary, bounce, shadow) of the step's own batches, "
                      f"oracle/nnbvh_oracle.c on {cores} threads, median of {args.cpu_passes} "
                      f"passes of {cpu_s:.2f}s",
            "matches_gpu": bool(same),
            "reference_8t_mrays": dict(REFERENCE_8T_MRAYS,
                                       note="the reference's own compiled CPU path, 8 threads, crown "
                                            "(BASELINE.md §2): survey container, not this box"),
        }
    if rank == 0:
        print(json.dumps(result), file=result_out, flush=True)
    film.close()
    agg.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
