"""Hit record -> SurfaceInteraction (Triangle::InteractionFromIntersection, shapes.h:884-1010).

CPU: the oracle's restatement against vectors produced by the REFERENCE's own compiled function
(tests/golden/tri_interaction.npz, made by tests/golden/make_interaction_golden.py with
oracle/_ref/ref_interaction) — every output word bit-equal — and, in the build container, against
the reference binary live on fresh random inputs.  GPU: the HIP post-pass against the oracle,
bit for bit, on the golden inputs and on hits traced through a small smooth-shaded mesh."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import oracle_binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from interaction_cases import cases, patch_cases, transform_cases  # noqa: E402

GOLDEN = os.path.join(HERE, "golden", "tri_interaction.npz")
REF = os.path.join(HERE, "..", "oracle", "_ref", "ref_interaction")
# record fields of the 44-float oracle / reference record
FIELDS = {"uv": slice(6, 8), "wo": slice(8, 11), "n": slice(11, 14), "dpdu": slice(14, 17),
          "dpdv": slice(17, 20), "ns": slice(20, 23), "dpdus": slice(23, 26), "dpdvs": slice(26, 29),
          "dndus": slice(29, 32), "dndvs": slice(32, 35), "pi_lo": slice(38, 41), "pi_hi": slice(41, 44)}


def test_oracle_matches_reference_vectors_bit_exact():
    g = np.load(GOLDEN)
    ob.interaction_branches(reset=True)
    out = ob.triangle_interaction_batch(g["inputs"])
    bad = np.nonzero(out.view(np.uint32) != g["outputs"])
    assert len(bad[0]) == 0, f"{len(bad[0])} words differ, first rows {bad[0][:5]} cols {bad[1][:5]}"
    # the vectors reach every rare branch of the function (see oracle/nnbvh_oracle.c)
    assert min(ob.interaction_branches()) > 0, ob.interaction_branches()


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.isdir("/root/reference")),
                    reason="compiled reference harness only exists in the build container")
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_equals_reference_live(seed):
    rec = cases(20000, seed)
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fi, "wb") as f:
            f.write(np.int32(len(rec)).tobytes())
            f.write(rec.tobytes())
        subprocess.run([REF, fi, fo], check=True)
        ref = np.fromfile(fo, dtype=np.uint32).reshape(len(rec), 44)
    out = ob.triangle_interaction_batch(rec)
    # faceIndex is 7 + record number in both
    assert np.array_equal(out.view(np.uint32), ref)


def test_oracle_patch_interaction_matches_reference_vectors_bit_exact():
    g = np.load(os.path.join(HERE, "golden", "blp_interaction.npz"))
    ob.patch_branches(reset=True)
    out = ob.patch_interaction_batch(g["inputs"])
    bad = np.nonzero(out.view(np.uint32) != g["outputs"])
    assert len(bad[0]) == 0, f"{len(bad[0])} words differ, first rows {bad[0][:5]} cols {bad[1][:5]}"
    assert min(ob.patch_branches()) > 0, ob.patch_branches()


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.isdir("/root/reference")),
                    reason="compiled reference harness only exists in the build container")
def test_oracle_patch_interaction_equals_reference_live():
    rec = patch_cases(20000, 5)
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fi, "wb") as f:
            f.write(np.int32(len(rec)).tobytes())
            f.write(rec.tobytes())
        subprocess.run([REF, "blp", fi, fo], check=True)
        ref = np.fromfile(fo, dtype=np.uint32).reshape(len(rec), 50)
    assert np.array_equal(ob.patch_interaction_batch(rec).view(np.uint32), ref)


@pytest.mark.gpu
def test_gpu_patch_interactions_match_oracle_on_reference_vectors():
    from nn_bvh_amd import HIT_DTYPE, RAY_DTYPE, _lib
    from nn_bvh_amd.interaction import ShadingMesh
    rec = np.load(os.path.join(HERE, "golden", "blp_interaction.npz"))["inputs"]
    n = len(rec)
    exp = ob.patch_interaction_batch(rec)
    flags_in = rec[:, 18].astype(np.int32)
    normals = rec[:, 27:39].reshape(-1, 3).copy()
    normals[np.repeat((flags_in & 8) != 0, 4)] *= -1  # util/mesh.cpp:216-223
    tri_flags = (((flags_in & 1) != 0) * _lib.TRI_HAS_UV + ((flags_in & 2) != 0) * _lib.TRI_HAS_N +
                 ((flags_in & 8) != 0) * _lib.TRI_FLIP_NORMAL).astype(np.uint8)
    mesh = ShadingMesh(rec[:, 0:12].reshape(-1, 3), np.full((n, 3), -1, np.int32), normals=normals,
                       uvs=rec[:, 19:27].reshape(-1, 2), face_indices=7 + np.arange(n, dtype=np.int32),
                       tri_flags=tri_flags, patch_vertices=np.arange(4 * n, dtype=np.int32).reshape(n, 4))
    hits = np.zeros(n, HIT_DTYPE)
    hits["prim"] = np.arange(n)
    hits["b0"], hits["b1"] = rec[:, 12], rec[:, 13]
    hits["t"] = 1.0
    rays = np.zeros(n, RAY_DTYPE)
    rays["d"] = -rec[:, 14:17]
    rays["time"] = rec[:, 17]
    got = mesh.interactions(rays, hits)
    assert (got["status"] == 3).all()
    assert_records_equal(got, exp, np.arange(n), "patch golden inputs")
    for name, sl in (("dndu", slice(44, 47)), ("dndv", slice(47, 50))):
        assert np.array_equal(got[name].view(np.uint32), exp[:, sl].view(np.uint32)), name
    mesh.close()


def test_oracle_interaction_transform_matches_reference_vectors_bit_exact():
    """Transform::operator()(const SurfaceInteraction &) (util/transform.cpp:229-261)."""
    g = np.load(os.path.join(HERE, "golden", "xf_interaction.npz"))
    out = ob.transform_interaction_batch(g["inputs"])
    assert np.array_equal(out.view(np.uint32), g["outputs"])
    # both branches of the error propagation and the FaceForward flip occur in the vectors
    rec = g["inputs"]
    assert ((rec[:, 35:38] - rec[:, 32:35]) == 0).all(1).any() and ((rec[:, 35:38] - rec[:, 32:35]) != 0).all(1).any()


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.isdir("/root/reference")),
                    reason="compiled reference harness only exists in the build container")
def test_oracle_interaction_transform_equals_reference_live():
    rec = transform_cases(20000, 6)
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fi, "wb") as f:
            f.write(np.int32(len(rec)).tobytes())
            f.write(rec.tobytes())
        subprocess.run([REF, "xf", fi, fo], check=True)
        ref = np.fromfile(fo, dtype=np.uint32).reshape(len(rec), 40)
    assert np.array_equal(ob.transform_interaction_batch(rec).view(np.uint32), ref)


def check_instance_interactions(got, rows, kind, rays, hits, verts, normals, tri_vertices, patch_vertices, m, mi):
    """Device records `got[rows]` of hits inside instances against the oracle composed of pinned pieces: the
    shape's InteractionFromIntersection in the instance's space (wo = -ApplyInverse(ray.d)), then
    Transform::operator()(SurfaceInteraction) with the per-row matrices m / mi ([n, 3, 4])."""
    d = rays["d"][rows]
    d_in = np.stack([(mi[:, i, 0] * d[:, 0] + mi[:, i, 1] * d[:, 1]) + mi[:, i, 2] * d[:, 2] for i in range(3)], 1)
    if kind == 0:
        rec = np.zeros((len(rows), 45), np.float32)
        tv = tri_vertices[hits["prim"][rows]]
        rec[:, 0:9] = verts[tv].reshape(-1, 9)
        rec[:, 9], rec[:, 10], rec[:, 11] = hits["b0"][rows], hits["b1"][rows], hits["b2"][rows]
        rec[:, 12:15] = -d_in
        rec[:, 18] = rays["time"][rows]
        rec[:, 19] = 2
        rec[:, 26:35] = normals[tv].reshape(-1, 9)
        local = ob.triangle_interaction_batch(rec)
        gdn = np.zeros((len(rows), 6), np.float32)
    else:
        rec = np.zeros((len(rows), 40), np.float32)
        pv = patch_vertices[hits["prim"][rows]]
        rec[:, 0:12] = verts[pv].reshape(-1, 12)
        rec[:, 12], rec[:, 13] = hits["b0"][rows], hits["b1"][rows]
        rec[:, 14:17] = -d_in
        rec[:, 17] = rays["time"][rows]
        rec[:, 18] = 2
        rec[:, 27:39] = normals[pv].reshape(-1, 12)
        local = ob.patch_interaction_batch(rec)
        gdn = local[:, 44:50]
    xf = np.zeros((len(rows), 72), np.float32)
    for half, mat in ((0, m), (16, mi)):
        m44 = np.zeros((len(rows), 4, 4), np.float32)
        m44[:, :3, :] = mat
        m44[:, 3, 3] = 1
        xf[:, half:half + 16] = m44.reshape(-1, 16)
    xf[:, 32:38] = local[:, 38:44]                       # pi low / high
    xf[:, 38:41], xf[:, 41:44] = local[:, 11:14], local[:, 8:11]   # n, wo
    xf[:, 44:50] = local[:, 14:20]                       # dpdu dpdv
    xf[:, 50:56] = gdn                                   # geometric dndu dndv
    xf[:, 56:71] = local[:, 20:35]                       # shading n dpdu dpdv dndu dndv
    exp = ob.transform_interaction_batch(xf)
    g = got[rows]
    for name, sl in (("pi_lo", slice(0, 3)), ("pi_hi", slice(3, 6)), ("n", slice(6, 9)), ("wo", slice(9, 12)),
                     ("dpdu", slice(12, 15)), ("dpdv", slice(15, 18)), ("dndu", slice(18, 21)),
                     ("dndv", slice(21, 24)), ("ns", slice(24, 27)), ("dpdus", slice(27, 30)),
                     ("dpdvs", slice(30, 33)), ("dndus", slice(33, 36)), ("dndvs", slice(36, 39))):
        a = np.ascontiguousarray(g[name]).view(np.uint32)
        b = np.ascontiguousarray(exp[:, sl]).view(np.uint32)
        bad = np.nonzero((a != b).any(1))[0]
        assert len(bad) == 0, f"kind {kind}: {name} differs on {len(bad)} of {len(rows)} records"
    # the (u, v) of the hit are not touched by the transform
    assert np.array_equal(g["uv"].view(np.uint32), np.ascontiguousarray(local[:, 6:8]).view(np.uint32))


@pytest.mark.gpu
def test_gpu_interactions_inside_animated_instances():
    """Hits inside AnimatedPrimitives (cpu/primitive.cpp:143-153): the interaction in the instance's space,
    then Interpolate(ray.time) applied to it — on the device, against the oracle composed of
    AnimatedTransform::Interpolate (pinned to the compiled reference; run with the device's sine, the
    path's documented exception), the shape's InteractionFromIntersection and Transform(SurfaceInteraction)."""
    import test_animated as ta
    from nn_bvh_amd import BVHAggregate, scene
    from nn_bvh_amd.interaction import ShadingMesh
    verts, prims, _, _, _, _, anims, oa, placements = ta.animated_scene(4, 36)
    nodes, aprims, instances, n_top = ta.rebuild_with_motion_bounds(verts, prims, placements, anims, oa)
    aprims = aprims.copy()
    obj = aprims["kind"] != 2  # the object's primitives; instance primitives keep their ids
    n_ids = int(aprims["id"][obj].max()) + 1
    tri_vertices = np.full((n_ids, 3), -1, np.int32)
    tri_vertices[aprims["id"][obj]] = aprims["v"][obj][:, :3]
    rng = np.random.default_rng(18)
    normals = rng.normal(size=(len(verts), 3)).astype(np.float32)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    mesh = ShadingMesh(verts, tri_vertices, normals=normals)
    agg = BVHAggregate.from_tree(nodes, aprims, verts, instances=instances, n_top_nodes=n_top, animated=anims)
    rays = scene.random_rays(60000, [-25, -25, -25], [25, 25, 25], 5)
    rays["time"] = np.random.default_rng(6).uniform(-0.2, 1.2, len(rays)).astype(np.float32)
    hits = agg.Intersect(rays)
    mesh.set_instances(instances, animated=anims)
    got = mesh.interactions(rays, hits)
    rows = np.nonzero((hits["prim"] >= 0) & (hits["instance"] > 0))[0]
    assert len(rows) > 3000 and (got["status"][rows] == 1).all()
    k = hits["instance"][rows] - 1
    try:
        ob.set_sin_mode(1)
        mm = ob.anim_interpolate(oa[k], rays["time"][rows])   # [n, 32]: m, mInv (4 x 4 each)
    finally:
        ob.set_sin_mode(0)
    m = mm[:, :16].reshape(-1, 4, 4)[:, :3, :]
    mi = mm[:, 16:].reshape(-1, 4, 4)[:, :3, :]
    moving = (anims["actually_animated"][k] != 0) & (rays["time"][rows] > 0) & (rays["time"][rows] < 1)
    assert moving.sum() > 1000  # the interpolated transform is what most of these records went through
    check_instance_interactions(got, rows, 0, rays, hits, verts, normals, tri_vertices, None, m, mi)
    agg.close()
    mesh.close()


@pytest.mark.gpu
def test_gpu_interactions_inside_instances():
    """Hits inside instances: interaction in the instance's space, then renderFromPrimitive applied
    (TransformedPrimitive::Intersect, cpu/primitive.cpp:112-125), all on the device."""
    from nn_bvh_amd import BVHAggregate, scene
    from nn_bvh_amd.interaction import ShadingMesh
    from test_instancing import two_level_scene
    verts, nodes, prims, instances, n_top, _ = two_level_scene(3, 50)
    prims = prims.copy()
    prims["id"] = np.arange(len(prims))  # ids index the shading mesh, so they must be unique
    n_ids = len(prims)
    tri_vertices = np.full((n_ids, 3), -1, np.int32)
    patch_vertices = np.full((n_ids, 4), -1, np.int32)
    tri_vertices[prims["id"][prims["kind"] == 0]] = prims["v"][prims["kind"] == 0][:, :3]
    patch_vertices[prims["id"][prims["kind"] == 1]] = prims["v"][prims["kind"] == 1]
    rng = np.random.default_rng(8)
    normals = rng.normal(size=(len(verts), 3)).astype(np.float32)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    mesh = ShadingMesh(verts, tri_vertices, normals=normals, patch_vertices=patch_vertices)
    agg = BVHAggregate.from_tree(nodes, prims, verts, instances=instances, n_top_nodes=n_top)
    lo = np.array([-30, -30, -30.0])
    rays = scene.random_rays(60000, lo, -lo, 9)
    rays["time"] = rng.random(len(rays)).astype(np.float32)
    hits = agg.Intersect(rays)
    inside = hits["instance"] > 0
    assert inside.sum() > 1000
    got = mesh.interactions(rays, hits)
    assert (got["status"][inside] == 2).all(), "without the instance table such hits are left to the host"
    mesh.set_instances(instances)
    got = mesh.interactions(rays, hits)
    for kind, status in ((0, 1), (1, 3)):
        is_kind = (tri_vertices if kind == 0 else patch_vertices)[np.maximum(hits["prim"], 0), 0] >= 0
        rows = np.nonzero(inside & (hits["prim"] >= 0) & is_kind)[0]
        assert len(rows) > (200 if kind == 0 else 20)
        assert (got["status"][rows] == status).all()
        inst = instances[hits["instance"][rows] - 1]
        check_instance_interactions(got, rows, kind, rays, hits, verts, normals, tri_vertices, patch_vertices,
                                    inst["render_from_prim"].reshape(-1, 3, 4), inst["prim_from_render"].reshape(-1, 3, 4))
    agg.close()
    mesh.close()


def mesh_from_records(rec):
    """One triangle with three private vertices per record; attributes as TriangleMesh stores them."""
    from nn_bvh_amd import _lib
    n = len(rec)
    flags_in = rec[:, 19].astype(np.int32)
    verts = rec[:, 0:9].reshape(-1, 3)
    uvs = rec[:, 20:26].reshape(-1, 2)
    normals = rec[:, 26:35].reshape(-1, 3).copy()
    normals[np.repeat((flags_in & 8) != 0, 3)] *= -1  # util/mesh.cpp:52-58
    tangents = rec[:, 36:45].reshape(-1, 3)
    tri_flags = (((flags_in & 1) != 0) * _lib.TRI_HAS_UV + ((flags_in & 2) != 0) * _lib.TRI_HAS_N +
                 ((flags_in & 4) != 0) * _lib.TRI_HAS_S + ((flags_in & 8) != 0) * _lib.TRI_FLIP_NORMAL)
    return dict(verts=verts, tri_vertices=np.arange(3 * n, dtype=np.int32).reshape(n, 3), normals=normals,
                uvs=uvs, tangents=tangents, face_indices=7 + np.arange(n, dtype=np.int32),
                tri_flags=tri_flags.astype(np.uint8))


def assert_records_equal(got, exp44, rows, what):
    for name, sl in FIELDS.items():
        a = np.ascontiguousarray(got[name][rows]).view(np.uint32).reshape(len(rows), -1)
        b = np.ascontiguousarray(exp44[rows][:, sl]).view(np.uint32)
        bad = np.nonzero((a != b).any(1))[0]
        assert len(bad) == 0, f"{what}: {name} differs on {len(bad)} records, first {rows[bad[:5]]}"
    assert np.array_equal(got["time"][rows].view(np.uint32), exp44[rows, 35].view(np.uint32)), what
    assert np.array_equal(got["face_index"][rows], exp44[rows, 36].astype(np.int32)), what


@pytest.mark.gpu
def test_gpu_interactions_match_oracle_on_reference_vectors():
    from nn_bvh_amd import HIT_DTYPE, RAY_DTYPE
    from nn_bvh_amd.interaction import ShadingMesh
    rec = np.load(GOLDEN)["inputs"]
    n = len(rec)
    exp = ob.triangle_interaction_batch(rec)
    mesh = ShadingMesh(**mesh_from_records(rec))
    hits = np.zeros(n, HIT_DTYPE)
    hits["prim"] = np.arange(n)
    hits["b0"], hits["b1"], hits["b2"] = rec[:, 9], rec[:, 10], rec[:, 11]
    hits["t"] = 1.0
    rays = np.zeros(n, RAY_DTYPE)
    rays["d"] = -rec[:, 12:15]
    rays["time"] = rec[:, 18]
    got = mesh.interactions(rays, hits)
    assert (got["status"] == 1).all() and np.array_equal(got["prim"], hits["prim"])
    assert_records_equal(got, exp, np.arange(n), "golden inputs")
    mesh.close()


@pytest.mark.gpu
def test_gpu_interactions_of_traced_hits_soa_queue_and_statuses():
    import torch
    import scenes_small as ss
    from nn_bvh_amd import BVHAggregate, HIT_DTYPE, _lib, build_tree, scene
    from nn_bvh_amd.interaction import ShadingMesh
    from nn_bvh_amd.wavefront import RayQueue, WavefrontAggregate
    verts, prims = ss.grid_mesh(40, 3)
    patch_v, patch_p = ss.random_soup(0, 60, 5)
    patch_p = patch_p.copy()
    patch_p["v"] += len(verts)
    patch_p["id"] += len(prims)
    allv, allp = np.concatenate([verts, patch_v]), np.concatenate([prims, patch_p])
    rng = np.random.default_rng(2)
    normals = rng.normal(size=(len(allv), 3)).astype(np.float32)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    uvs = rng.random((len(allv), 2)).astype(np.float32)
    tri_vertices = np.full((len(allp), 3), -1, np.int32)
    tri_vertices[allp["id"][allp["kind"] == 0]] = allp["v"][allp["kind"] == 0][:, :3]
    patch_vertices = np.full((len(allp), 4), -1, np.int32)
    patch_vertices[allp["id"][allp["kind"] == 1]] = allp["v"][allp["kind"] == 1]
    mesh = ShadingMesh(allv, tri_vertices, normals=normals, uvs=uvs, patch_vertices=patch_vertices)
    tree = build_tree(allp, allv)
    agg = BVHAggregate.from_tree(tree.nodes, tree.ordered_prims, allv)
    n = 20000
    rays = scene.random_rays(n, allv.min(0) - 2, allv.max(0) + 2, 7)
    rays["time"] = rng.random(n).astype(np.float32)
    dev = torch.device("cuda", 0)
    rq = RayQueue.from_records(rays, dev)
    rq.time = torch.from_numpy(np.ascontiguousarray(rays["time"])).to(dev)
    size = n - 123
    rq.size.fill_(size)
    hits_t = WavefrontAggregate(agg).IntersectClosest(n, rq)
    out = torch.full((n * 192,), 0x5A, dtype=torch.uint8, device=dev)
    mesh.interactions_device(hits_t.data_ptr(), n, out.data_ptr(), ray_queue=rq, d_size=rq.size.data_ptr(),
                             stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(_lib.INTERACTION_DTYPE)
    hits = hits_t.cpu().numpy().view(HIT_DTYPE).reshape(-1)[:size]
    assert (out.cpu().numpy()[size * 192:] == 0x5A).all(), "records beyond the queue size were written"
    is_tri = (hits["prim"] >= 0) & (hits["prim"] < len(prims))
    is_patch = hits["prim"] >= len(prims)
    assert is_tri.sum() > 1000 and is_patch.sum() > 10 and (hits["prim"] < 0).sum() > 1000
    assert (got["status"][:size][hits["prim"] < 0] == 0).all()
    assert (got["status"][:size][is_patch] == 3).all()
    assert (got["status"][:size][is_tri] == 1).all()
    # oracle records for the triangle hits
    rows = np.nonzero(is_tri)[0]
    rec = np.zeros((len(rows), 45), np.float32)
    tv = tri_vertices[hits["prim"][rows]]
    rec[:, 0:9] = allv[tv].reshape(-1, 9)
    rec[:, 9], rec[:, 10], rec[:, 11] = hits["b0"][rows], hits["b1"][rows], hits["b2"][rows]
    rec[:, 12:15] = -rays["d"][rows]
    rec[:, 18] = rays["time"][rows]
    rec[:, 19] = 3
    rec[:, 20:26] = uvs[tv].reshape(-1, 6)
    rec[:, 26:35] = normals[tv].reshape(-1, 9)
    exp = ob.triangle_interaction_batch(rec)
    exp[:, 36] = 0  # no faceIndices array: faceIndex 0
    full = np.zeros((size, 44), np.float32)
    full[rows] = exp
    assert_records_equal(got[:size], full, rows, "traced hits")
    assert not got["dndu"][rows].any() and not got["dndv"][rows].any()  # Triangle passes Normal3f()
    # ... and for the hits on bilinear patches (BilinearPatch::InteractionFromIntersection)
    prow = np.nonzero(is_patch)[0]
    prec = np.zeros((len(prow), 40), np.float32)
    pv = patch_vertices[hits["prim"][prow]]
    prec[:, 0:12] = allv[pv].reshape(-1, 12)
    prec[:, 12], prec[:, 13] = hits["b0"][prow], hits["b1"][prow]
    prec[:, 14:17] = -rays["d"][prow]
    prec[:, 17] = rays["time"][prow]
    prec[:, 18] = 3
    prec[:, 19:27] = uvs[pv].reshape(-1, 8)
    prec[:, 27:39] = normals[pv].reshape(-1, 12)
    pexp = ob.patch_interaction_batch(prec)
    pexp[:, 36] = 0
    pfull = np.zeros((size, 50), np.float32)
    pfull[prow] = pexp
    assert_records_equal(got[:size], pfull, prow, "traced patch hits")
    for name, sl in (("dndu", slice(44, 47)), ("dndv", slice(47, 50))):
        assert np.array_equal(got[name][prow].view(np.uint32), pexp[:, sl].view(np.uint32)), name
    agg.close()
    mesh.close()
