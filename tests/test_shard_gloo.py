"""The N>1 path on CPU: world_size-2 gloo processes shard a film by tile, each "traces" its
tiles (the oracle stands in for the GPU kernel here — this test is about the partition, the
all-gather and the reassembly, not about traversal), and every rank must end up with exactly
the single-process film."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
import scenes_small as ss
from nn_bvh_amd import HIT_DTYPE, build_tree, scene, shard

CAM = ((0, 12, 0.5), (0, 0, 0), (0, 1, 0), 50.0, 96, 80)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    verts, prims = ss.grid_mesh(24, 2)
    tree = build_tree(prims, verts)
    rays, px, py = scene.camera_rays(CAM, seed=4, return_pixels=True)
    idx = shard.shard_indices(px, py, CAM[4], world, rank)
    local = ob.closest(tree.nodes, tree.ordered_prims, verts, rays[idx])
    counts = shard.shard_counts(px, py, CAM[4], world) * HIT_DTYPE.itemsize
    parts = shard.all_gather_records(torch.from_numpy(local.view(np.uint8).reshape(-1).copy()), counts)
    lists = [shard.shard_indices(px, py, CAM[4], world, r) for r in range(world)]
    film = shard.assemble([p.numpy().view(HIT_DTYPE) for p in parts], lists, len(rays), HIT_DTYPE)
    np.save(os.path.join(out_dir, f"film{rank}.npy"), film)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharded_film_equals_single_process(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    verts, prims = ss.grid_mesh(24, 2)
    tree = build_tree(prims, verts)
    rays = scene.camera_rays(CAM, seed=4)
    full = ob.closest(tree.nodes, tree.ordered_prims, verts, rays)
    assert (full["prim"] >= 0).mean() > 0.3
    for r in range(world):
        film = np.load(os.path.join(str(tmp_path), f"film{r}.npy"))
        assert film.tobytes() == full.tobytes()


def test_tile_map_partitions_every_pixel_once_and_balances():
    _, px, py = scene.camera_rays("crown", return_pixels=True, jitter=False)
    for world in (1, 2, 4, 8):
        counts = shard.shard_counts(px, py, 1000, world)
        assert counts.sum() == len(px)
        assert counts.max() - counts.min() <= 16 * 16 * 2  # within a couple of tiles
        seen = np.zeros(len(px), int)
        for r in range(world):
            seen[shard.shard_indices(px, py, 1000, world, r)] += 1
        assert (seen == 1).all()
    # all rays of a tile stay on one rank
    t = shard.tile_of_pixel(px, py, 1000)
    owner = shard.rank_of_tile(t, 8)
    assert all(len(set(owner[t == k])) == 1 for k in np.unique(t)[:50])
