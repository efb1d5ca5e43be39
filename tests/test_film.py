"""Film accumulators (RGBFilm::AddSample, /root/reference/src/pbrt/film.h:239-255, as UpdateFilm
calls it, wavefront/film.cpp:13-40): oracle known answers on the CPU, device kernels vs the oracle
bit for bit on the GPU, and the tile all-gather over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
from nn_bvh_amd import _lib, scene, shard


def _samples(seed, xres, yres, n_slots, n_passes, stride=3):
    rng = np.random.default_rng(seed)
    lin = rng.choice(xres * yres, n_slots, replace=False)
    px, py = (lin % xres).astype(np.int32), (lin // xres).astype(np.int32)
    px[::17] += xres  # outside the bounds: skipped (film.cpp:18-19)
    py[5::23] = -1
    rgb = (rng.random((n_slots * n_passes, stride), np.float32) * np.float32(3.0)).astype(np.float32)
    rgb[::11] *= np.float32(40.0)  # beyond the clamp
    w = (rng.random(n_slots * n_passes, np.float32) + np.float32(0.25)).astype(np.float32)
    return px, py, rgb, w


def test_oracle_film_known_answer():
    pix = np.zeros((4, 4), np.float64)
    px, py = np.array([1, 0], np.int32), np.array([0, 1], np.int32)
    rgb = np.array([[1, 2, 4], [0.5, 0.25, 0.125], [8, 0, 0], [1, 1, 1]], np.float32)
    w = np.array([2, 1, 0.5, 3], np.float32)
    ob.film_add_samples(pix, (0, 0, 2, 2), 2.0, px, py, rgb, w, 2)
    # pixel (1,0): pass 0 sample clamped 4 -> 2 (scale 0.5): (0.5, 1, 2) * 2; pass 1: (8,0,0) -> (2,0,0) * 0.5
    assert pix[1].tolist() == [1.0 + 1.0, 2.0, 4.0, 2.5]
    # pixel (0,1) = index 2: (0.5, .25, .125) * 1 + (1,1,1) * 3
    assert pix[2].tolist() == [3.5, 3.25, 3.125, 4.0]
    assert not pix[0].any() and not pix[3].any()


def test_oracle_film_sums_in_sample_order_as_doubles():
    # float product, double accumulation (film.h:252-254): compare with a literal restatement
    px, py, rgb, w = _samples(3, 8, 8, 40, 5)
    pix = ob.film_add_samples(np.zeros((64, 4)), (0, 0, 8, 8), 1e30, px, py, rgb, w, 5)
    exp = np.zeros((64, 4))
    for p in range(5):
        for i in range(40):
            if 0 <= px[i] < 8 and 0 <= py[i] < 8:
                k = p * 40 + i
                for c in range(3):
                    exp[py[i] * 8 + px[i], c] += float(np.float32(w[k] * rgb[k, c]))
                exp[py[i] * 8 + px[i], 3] += float(w[k])
    assert pix.tobytes() == exp.tobytes()


def test_film_create_rejects_bad_arguments_without_touching_a_device():
    L = _lib.lib()
    assert not L.nnbvh_film_create(0, 0, 0, 4, 1.0, 0)
    assert b"bounds" in L.nnbvh_last_error()
    assert not L.nnbvh_film_create(0, 0, 4, 4, 0.0, 0)
    assert L.nnbvh_film_add_samples_device(None, None, None, None, 3, None, 1, 1, None, None) == 1
    assert L.nnbvh_film_pack_pixels_device(None, None, 1, None, None) == 1


# ---- GPU ------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("stride,clamp,passes", [(3, 2.0, 1), (4, float("inf"), 4), (3, 0.75, 7)])
def test_device_film_equals_oracle(stride, clamp, passes):
    from nn_bvh_amd.film import Film
    xres, yres, n_slots = 200, 120, 9000
    px, py, rgb, w = _samples(7, xres, yres, n_slots, passes, stride)
    film = Film(xres, yres, clamp)
    d = [torch.from_numpy(a).cuda() for a in (px, py, rgb, w)]
    s = torch.cuda.current_stream().cuda_stream
    film.add_samples_device(d[0], d[1], d[2], d[3], n_slots, passes, rgb_stride=stride, stream=s)
    # a second call accumulates on top; device-side size clamps the slots
    n_dev = torch.tensor([n_slots // 2], dtype=torch.int32, device="cuda")
    film.add_samples_device(d[0], d[1], d[2], None, n_slots, 1, rgb_stride=stride, d_size=n_dev, stream=s)
    got = film.read()
    exp = ob.film_add_samples(np.zeros((xres * yres, 4)), (0, 0, xres, yres), clamp, px, py, rgb, w, passes)
    ob.film_add_samples(exp, (0, 0, xres, yres), clamp, px[: n_slots // 2], py[: n_slots // 2],
                        rgb[: n_slots // 2], None, 1)
    assert got.tobytes() == exp.tobytes()
    assert got[:, 3].sum() > 0
    # pack / unpack round trip through a second film
    idx = torch.from_numpy(np.random.default_rng(1).permutation(xres * yres)[:5000].astype(np.int32)).cuda()
    buf = torch.empty((5000, 4), dtype=torch.float64, device="cuda")
    film.pack(idx, 5000, buf, s)
    other = Film(xres, yres)
    other.unpack(idx, 5000, buf, s)
    back = other.read()
    ii = idx.cpu().numpy()
    assert back[ii].tobytes() == got[ii].tobytes()
    mask = np.ones(xres * yres, bool)
    mask[ii] = False
    assert not back[mask].any()
    film.clear(s)
    assert not film.read().any()
    film.close()
    other.close()


@pytest.mark.gpu
def test_film_all_gather_under_rccl_on_a_side_stream():
    """Film.gather_tiles / all_gather_tiles with CUDA tensors under the nccl (= RCCL) backend, world 1 on the
    one device, on a NON-default stream: pack -> all_gather_into_tensor -> consumer must be ordered on that
    stream (the collective orders itself against torch's current stream only)."""
    from nn_bvh_amd.film import Film
    xres, yres, n_slots, passes = 320, 200, 40000, 3
    px, py, rgb, w = _samples(11, xres, yres, n_slots, passes, 3)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        side = torch.cuda.Stream()
        film, other = Film(xres, yres), Film(xres, yres)
        d = [torch.from_numpy(a).cuda() for a in (px, py, rgb, w)]
        ok = (px >= 0) & (px < xres) & (py >= 0) & (py < yres)  # _samples puts some outside the film on purpose
        idx = torch.from_numpy(np.unique(py[ok].astype(np.int64) * xres + px[ok]).astype(np.int32)).cuda()
        torch.cuda.synchronize()
        raw = side.cuda_stream
        # the producer of the film's content runs on the side stream too, right before the gather
        film.add_samples_device(d[0], d[1], d[2], d[3], n_slots, passes, rgb_stride=3, stream=raw)
        out = film.gather_tiles([idx], 0, stream=raw)
        other.unpack(idx, int(idx.numel()), out[0], raw)
        sent = film.all_gather_tiles([idx], 0, stream=raw)  # world 1: nothing to unpack, must not disturb the film
        side.synchronize()
        assert sent == int(idx.numel()) * 32
        exp = ob.film_add_samples(np.zeros((xres * yres, 4)), (0, 0, xres, yres), float("inf"), px, py, rgb, w, passes)
        assert film.read().tobytes() == exp.tobytes()
        assert other.read().tobytes() == exp.tobytes()  # every touched pixel arrived, the rest is zero in both
        assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
        film.close()
        other.close()
    finally:
        dist.destroy_process_group()


# ---- N > 1 on CPU: tiles of the film are accumulated by their ranks and all-gathered ------------
CAM = ((0, 12, 0.5), (0, 0, 0), (0, 1, 0), 50.0, 96, 80)
SPP = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _film_inputs():
    _, px, py = scene.camera_rays(CAM, seed=4, return_pixels=True)
    rng = np.random.default_rng(9)
    rgb = rng.random((SPP * len(px), 3), np.float32) * np.float32(2.0)
    w = rng.random(SPP * len(px), np.float32) + np.float32(0.5)
    return px.astype(np.int32), py.astype(np.int32), rgb.astype(np.float32), w.astype(np.float32)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    px, py, rgb, w = _film_inputs()
    n = len(px)
    lists = [shard.shard_indices(px, py, CAM[4], world, r) for r in range(world)]
    mine = lists[rank]
    sel = np.concatenate([p * n + mine for p in range(SPP)])  # this rank's samples, pass-major
    pix = ob.film_add_samples(np.zeros((CAM[4] * CAM[5], 4)), (0, 0, CAM[4], CAM[5]), 1.5, px[mine], py[mine],
                              rgb[sel], w[sel], SPP)
    lin = [torch.from_numpy((py[ix].astype(np.int64) * CAM[4] + px[ix])) for ix in lists]
    full = shard.all_gather_film(torch.from_numpy(pix), lin, rank)
    np.save(os.path.join(out_dir, f"film{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharded_film_accumulators_equal_single_process(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    px, py, rgb, w = _film_inputs()
    exp = ob.film_add_samples(np.zeros((CAM[4] * CAM[5], 4)), (0, 0, CAM[4], CAM[5]), 1.5, px, py, rgb, w, SPP)
    assert exp[:, 3].min() > 0
    for r in range(world):
        assert np.load(os.path.join(str(tmp_path), f"film{r}.npy")).tobytes() == exp.tobytes()
