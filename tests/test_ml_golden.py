"""The Python (machine_learning/) side against outputs of the reference's own numpy-only modules
(nn_parser, nn_AABB, nn_types, nn_mesh_list), recorded by tests/golden/make_ml_golden.py in the build
container: scale_scene and the scene bounds it uses, the AABB accessors' y-for-z slip, and the triangle
lists the two text readers produce.  Nothing Python from the reference is needed to run these."""
import hashlib
import os

import numpy as np
import pytest

from nn_bvh_amd import nn_tree, scene

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "ml_reference.npz"))
REF_SCENES = os.path.join(os.environ.get("NNBVH_REFERENCE", "/root/reference"), "machine_learning", "test_scenes")


@pytest.mark.parametrize("k", [0, 1, 2])
def test_scale_scene_equals_the_reference_bit_for_bit(k):
    P = G[f"scale_in{k}"]
    assert nn_tree.scale_scene(P).tobytes() == G[f"scale_out{k}"].tobytes()
    assert nn_tree.scale_scene(P, 2).tobytes() == G[f"scale_shift2_out{k}"].tobytes()
    assert P.tobytes() == G[f"scale_in{k}"].tobytes()  # the input is not modified (the reference scales in place)


def test_scene_bounds_keep_the_reference_max_initialisation():
    """Set 2 lies entirely below zero on x and z: the reference's maxima start at sys.float_info.min, so the
    recorded bounds keep 2.2e-308 there (nn_AABB.py:59-60) and scale_scene divides by -min."""
    import sys
    for k in (0, 2):
        lo, hi = nn_tree.scene_bounds(G[f"scale_in{k}"])
        assert np.concatenate([lo, hi]).tobytes() == G[f"aabb_from_prims{k}"].tobytes()
    assert G["aabb_from_prims2"][3] == sys.float_info.min and G["aabb_from_prims2"][5] == sys.float_info.min
    assert nn_tree.scene_bounds(np.zeros((0, 3, 3)))[1].tolist() == [0, 0, 0]


def test_aabb_accessors_return_y_for_z():
    boxes, acc = G["aabb_boxes"], G["aabb_min_max"]
    for b, a in zip(boxes, acc):
        lo, hi = b[:3], b[3:]
        for axis in range(3):
            assert nn_tree.aabb_get_min(lo, axis) == a[axis, 0] and nn_tree.aabb_get_max(hi, axis) == a[axis, 1]
        assert a[2, 0] == b[1] and a[2, 1] == b[4]  # the recorded reference values ARE the y bounds


def _prims_of(verts, tris):
    return verts[tris]  # (m, 3, 3) float32


def test_pbrt_trianglemesh_reader_equals_the_reference_parser():
    v, t, mesh_of = scene.read_pbrt_trianglemeshes(os.path.join(HERE, "golden", "ml_meshes.pbrt"))
    want = G["pbrt_prims"].astype(np.float32)
    assert _prims_of(v, t).tobytes() == want.tobytes()
    iv = G["pbrt_meshes"]
    assert [int((mesh_of == m).sum()) for m in range(len(iv))] == (iv[:, 1] - iv[:, 0]).tolist()
    assert np.all(np.diff(mesh_of) >= 0)


def test_obj_reader_equals_the_reference_parser():
    v, t, mesh_of = scene.read_obj(os.path.join(HERE, "golden", "ml_meshes.obj"))
    assert _prims_of(v, t).tobytes() == G["obj_prims"].astype(np.float32).tobytes()
    # the reference's mesh intervals (after it drops the empty mesh before the first `g`)
    iv = G["obj_meshes"]
    sizes = [int((mesh_of == m).sum()) for m in np.unique(mesh_of)]
    assert sizes == (iv[:, 1] - iv[:, 0]).tolist()


@pytest.mark.parametrize("i", [0, 1, 2])
def test_reference_test_scenes_read_identically(i):
    """machine_learning/test_scenes/*.obj, present in the build container only: triangle count, the float32
    corner array (sha256) and scale_scene on the first 512 triangles equal what the reference's parser +
    scale_scene gave."""
    path = os.path.join(REF_SCENES, str(G["test_scene_names"][i]) + ".obj")
    if not os.path.exists(path):
        pytest.skip("the reference's scene files are not on this machine")
    v, t, mesh_of = scene.read_obj(path)
    assert len(t) == int(G["test_scene_counts"][i, 0])
    P32 = np.ascontiguousarray(_prims_of(v, t))
    assert hashlib.sha256(P32.tobytes()).hexdigest() == str(G["test_scene_sha256_f32"][i])
    # scale_scene in float64 on the reference's own float64 corners: re-read the first 512 triangles at full
    # precision (read_obj keeps float32, which is what the device consumes)
    corners, faces = [], []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if tok and tok[0] == "v":
                corners.append([float(x) for x in tok[1:4]])
            elif tok and tok[0] == "f":
                faces.append([int(x.split("/")[0]) - 1 for x in tok[1:4]])
                if len(faces) == 512:
                    break
    head = np.array(corners, np.float64)[np.array(faces)]
    assert nn_tree.scale_scene(head).tobytes() == G["test_scene_scaled_head"][i].tobytes()
