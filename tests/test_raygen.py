"""The bench's device-side batch generators (nn_bvh_amd/raygen.py, here on torch's CPU device) against the numpy
recipes of scene.py they restate: same geometry for the same random numbers is not the contract (they draw their
own streams) — the invariants are: camera rays through their pixels, bounce rays leaving the hit point into the
hemisphere of the ray-facing normal, shadow rays ending on the light."""
import numpy as np
import torch

import oracle_binding as ob
import scenes_small as ss
from nn_bvh_amd import build_tree, raygen, scene

CAM = ((0, 12, 0.5), (0, 0, 0), (0, 1, 0), 50.0, 96, 80)


def setup():
    verts, prims = ss.grid_mesh(20, 3, bump=0.3)
    tris = prims["v"][:, :3].copy()
    tree = build_tree(prims, verts)
    ds = raygen.DeviceScene(verts, tris, torch.device("cpu"))
    _, px, py = scene.camera_rays(CAM, seed=4, return_pixels=True)
    return verts, tris, tree, ds, px, py


def test_camera_rays_go_through_their_pixels_and_samples_differ():
    verts, tris, tree, ds, px, py = setup()
    tpx, tpy = torch.from_numpy(px.astype(np.float64)), torch.from_numpy(py.astype(np.float64))
    a = raygen.as_records(ds.camera_rays(CAM, tpx, tpy, seed=1, sample=0))
    b = raygen.as_records(ds.camera_rays(CAM, tpx, tpy, seed=1, sample=1))
    again = raygen.as_records(ds.camera_rays(CAM, tpx, tpy, seed=1, sample=0))
    assert a.tobytes() == again.tobytes() and a.tobytes() != b.tobytes()
    ref0 = scene.camera_rays(CAM, seed=1, sample=0, jitter=False)  # pixel centres
    # both are unit vectors from the same eye; a jittered ray stays within a pixel's angle of its centre ray
    cosang = (a["d"] * ref0["d"]).sum(1)
    assert np.allclose(np.linalg.norm(a["d"], axis=1), 1, atol=1e-6) and (a["o"] == ref0["o"]).all()
    pixel_angle = 2 * np.tan(np.radians(CAM[3]) / 2) / min(CAM[4], CAM[5])
    assert (np.arccos(np.clip(cosang, -1, 1)) < pixel_angle).all()


def test_bounce_and_shadow_rays_leave_the_hit_points():
    verts, tris, tree, ds, px, py = setup()
    tpx, tpy = torch.from_numpy(px.astype(np.float64)), torch.from_numpy(py.astype(np.float64))
    prim_t = ds.camera_rays(CAM, tpx, tpy, seed=1, sample=0)
    rays = raygen.as_records(prim_t)
    hits = ob.closest(tree.nodes, tree.ordered_prims, verts, rays)
    d_hits = torch.from_numpy(hits.view(np.uint8).reshape(-1).copy())
    p, n, m = scene.hit_points(rays, hits, verts, tris)
    bt, idx = ds.bounce_rays(prim_t, d_hits, seed=[2, 0, 0])
    bounce = raygen.as_records(bt)
    assert np.array_equal(idx.numpy(), np.nonzero(m)[0]) and len(bounce) == m.sum() > 1000
    off = bounce["o"].astype(np.float64) - p
    eps = 1e-4 * np.abs(verts).max()
    assert np.allclose(off, n * eps, atol=1e-6)                      # origin = hit point pushed along the normal
    assert ((bounce["d"] * n).sum(1) > -1e-6).all()                  # into the normal's hemisphere
    assert np.allclose(np.linalg.norm(bounce["d"], axis=1), 1, atol=1e-5)
    assert abs(((bounce["d"] * n).sum(1)).mean() - 2 / 3) < 0.02      # cosine-weighted: E[cos] = 2/3
    quads = np.array([[[-2, 9, -2], [2, 9, -2], [2, 9, 2], [-2, 9, 2]], [[5, 5, 0], [5, 7, 0], [5, 7, 2], [5, 5, 2]]], np.float64)
    st, sidx = ds.shadow_rays(prim_t, d_hits, seed=[3, 0, 0], quads=quads)
    shadow = raygen.as_records(st)
    end = shadow["o"].astype(np.float64) + shadow["d"].astype(np.float64)
    on0 = (np.abs(end[:, 1] - 9) < 1e-4) & (np.abs(end[:, 0]) <= 2.001) & (np.abs(end[:, 2]) <= 2.001)
    on1 = (np.abs(end[:, 0] - 5) < 1e-4) & (end[:, 1] >= 4.999) & (end[:, 1] <= 7.001) & (end[:, 2] >= -0.001) & (end[:, 2] <= 2.001)
    assert (on0 | on1).all() and 0.3 < on0.mean() < 0.7 and (shadow["tmax"] == np.float32(1 - 1e-4)).all()
    bx, _ = ds.shadow_rays(prim_t, d_hits, seed=[3, 0, 1], box=([0, 8, 0], [1, 9, 1]))
    e2 = raygen.as_records(bx)
    end2 = e2["o"].astype(np.float64) + e2["d"].astype(np.float64)
    assert (end2 >= [-1e-4, 8 - 1e-4, -1e-4]).all() and (end2 <= [1.0001, 9.0001, 1.0001]).all()
