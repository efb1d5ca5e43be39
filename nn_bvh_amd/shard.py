"""Tile sharding of ray batches over ranks (one process per GPU) and the film all-gather.

The reference has exactly one parallelisation: shared-memory tiles of the image handed to
pool threads (ParallelFor2D, /root/reference/src/pbrt/util/parallel.cpp:307-324, tile edge
<= 32; cpu/integrators.cpp:164-187).  The multi-GPU form keeps the unit (the image tile) and
replaces the thread pool by ranks: the BVH is replicated, tile t belongs to rank t mod N
(interleaved, so every rank sees every image region and load balances), every ray of a
pixel stays on its tile's rank, and nothing is exchanged during traversal.  The only
collective is one all-gather of per-tile results when a pass is complete (RCCL over xGMI
when the backend is "nccl"; gloo in the CPU tests).
"""
import numpy as np

TILE = 16


def tile_grid(xres, yres, tile=TILE):
    return (xres + tile - 1) // tile, (yres + tile - 1) // tile


def tile_of_pixel(px, py, xres, tile=TILE):
    tx, _ = (xres + tile - 1) // tile, None
    return (np.asarray(py, np.int64) // tile) * tx + (np.asarray(px, np.int64) // tile)


def rank_of_tile(tile_id, world):
    return np.asarray(tile_id) % world


def shard_indices(px, py, xres, world, rank, tile=TILE):
    """Indices (into the pixel-ordered batch) of the rays whose tile belongs to `rank`."""
    return np.nonzero(rank_of_tile(tile_of_pixel(px, py, xres, tile), world) == rank)[0]


def shard_counts(px, py, xres, world, tile=TILE):
    owner = rank_of_tile(tile_of_pixel(px, py, xres, tile), world)
    return np.bincount(owner, minlength=world)


def all_gather_records(local, counts, group=None):
    """All-gather variable-length per-rank record arrays (numpy structured or torch uint8).

    local  : this rank's records as a torch uint8 tensor of count*itemsize bytes (CPU for
             gloo, CUDA for nccl/RCCL)
    counts : per-rank BYTE counts (known to every rank from the deterministic tile map)
    returns: list of per-rank torch uint8 tensors (views into one gathered buffer)
    One collective: ranks pad to the largest shard so a single all_gather_into_tensor moves
    everything (fixed-size ring all-gather)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    width = int(max(counts))
    send = torch.zeros(width, dtype=torch.uint8, device=local.device)
    send[: local.numel()] = local
    out = torch.empty(world * width, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, send, group=group)
    return [out[r * width: r * width + int(counts[r])] for r in range(world)]


def assemble(parts, index_lists, n_total, dtype):
    """Scatter per-rank record arrays back into pixel order."""
    full = np.zeros(n_total, dtype)
    for recs, idx in zip(parts, index_lists):
        full[idx] = np.frombuffer(bytes(recs), dtype) if not isinstance(recs, np.ndarray) else recs
    return full


def all_gather_film(pixels, index_lists, rank, group=None):
    """Film all-gather on a torch float64 [n_pixels, 4] tensor (RGBFilm::Pixel: rgbSum[3],
    weightSum): rank r owns the pixels index_lists[r] (torch int64 tensors on the same device).
    One all_gather_into_tensor of the padded per-rank pixel lists; the other ranks' pixels are
    written into `pixels` in place.  The device film does the same with its pack / unpack kernels
    (nn_bvh_amd.film.Film.all_gather_tiles); this form serves CPU tensors (gloo) and tests."""
    import torch
    import torch.distributed as dist
    world = len(index_lists)
    width = max(int(ix.numel()) for ix in index_lists)
    send = torch.zeros((width, 4), dtype=torch.float64, device=pixels.device)
    send[: index_lists[rank].numel()] = pixels[index_lists[rank]]
    out = torch.empty((world * width, 4), dtype=torch.float64, device=pixels.device)
    dist.all_gather_into_tensor(out, send, group=group)
    out = out.view(world, width, 4)
    for r in range(world):
        if r != rank:
            pixels[index_lists[r]] = out[r, : index_lists[r].numel()]
    return pixels
