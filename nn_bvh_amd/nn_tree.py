"""NN / greedy-SAH split trees of the fork's machine_learning/ directory, baked into the
LinearBVHNode layout the traversal path consumes (BASELINE.json
config 5: "NN_BVH learned-split tree ... baked to LinearBVHNode").

The reference has no such baker: its Python side builds `BVHNode` objects and
scores them; nothing reaches pbrt.  This module restates, in numpy, the three pieces that
define a tree there —
  * BVHNode.split                   /root/reference/machine_learning/nn_BVH.py:32-101
    (primitive -> left if max <= pos; right if min > pos; a straddler goes left iff
     pos - min >= max - pos; children get tight AABBs)
  * build_tree_from_nn_prediction   /root/reference/machine_learning/nn_tree_bench.py:44-77
    (level-order [onehot_x, onehot_y, onehot_z, offset] rows -> splits)
  * build_greedy_SAH_tree_tf        /root/reference/machine_learning/nn_BVH.py:193-260 with
    SAH_single_node_tf              /root/reference/machine_learning/nn_loss.py:227-274
    (candidates = sorted unique float32 primitive mid-points per axis, padded to batches of 8
     with the parent box's max — including the reference's AABB.get_max(z) == y_max slip,
     nn_AABB.py:36-37; cost classifies by MID-POINT, not by the extent rule split() then
     applies; min/max reductions assume coordinates in [0, 1], i.e. a scene normalised by
     scale_scene, nn_parser.py:175-250)
— and defines the missing step: leaves of the top tree are finished by the host SAH builder
and everything is flattened depth-first into nnbvh_linear_node[] (first child = index + 1,
axis = the split dimension), so near/far ordering and the traversal kernels work unchanged.

Pinning: nn_BVH.py itself cannot be imported here (module-level `import tensorflow`, nn_BVH.py:191; no
trained weights ship), so the split / greedy-SAH restatement is "parity unpinned" except for the 3-triangle
known answer of nn_test.py:48-85 (tests/test_nn_tree.py).  What the reference's numpy-only modules DO define —
scale_scene, the AABB accessors and get_AABB_from_primitives — is pinned bit for bit to outputs of those
modules (tests/golden/ml_reference.npz, tests/test_ml_golden.py).  The candidate sweep
below evaluates the same float32 expression per candidate as the TensorFlow code, with
prefix/suffix reductions instead of O(n^2) masks.
"""
import numpy as np

from ._lib import NODE_DTYPE, PRIM_DTYPE
from .aggregate import build_tree, make_prims

F = np.float32
C_TRI = F(1.0)  # nn_loss.py:116
BATCH = 8       # batch_size_gpu default, nn_BVH.py:193


def scene_bounds(P):
    """get_AABB_from_primitives (nn_AABB.py:56-91) as the reference computes it: minima start at
    sys.float_info.max, MAXIMA at sys.float_info.min — the smallest positive double, not the most negative
    one — so an axis whose coordinates are all below 2.2e-308 keeps that value as its maximum; no primitives
    -> AABB(0, ..., 0).  Returns (lo[3], hi[3])."""
    import sys
    if len(P) == 0:
        return np.zeros(3), np.zeros(3)
    v = np.asarray(P, np.float64).reshape(-1, 3)
    return np.minimum(v.min(0), sys.float_info.max), np.maximum(v.max(0), sys.float_info.min)


def aabb_get_min(lo, axis):
    """AABB.get_min (nn_AABB.py:21-28): axis z returns y_min."""
    return lo[1] if axis == 2 else lo[axis]


def aabb_get_max(hi, axis):
    """AABB.get_max (nn_AABB.py:30-37): axis z returns y_max — the slip the greedy builder's candidate
    padding inherits (best_sah_split)."""
    return hi[1] if axis == 2 else hi[axis]


def scale_scene(P, shift=0):
    """nn_parser.py:175-250 — move the scene's minimum corner to the origin, divide by the largest extent
    of its bounds (scene_bounds: the reference's, quirk included), add `shift`; per coordinate the
    reference's three float64 operations in its order.  P: (n, 3, 3) triangle corners.  Returns a new array."""
    P = np.array(P, np.float64)
    lo, hi = scene_bounds(P)
    P = P + np.where(lo < 0, np.abs(lo), -lo)  # both branches move the minimum to 0 (a zero minimum: no move)
    P = P / (hi - lo).max()
    return P + shift


def split_mask(P, axis, pos):
    """BVHNode.split classification (nn_BVH.py:44-68): True = left."""
    mx = P[:, :, axis].max(1)
    mn = P[:, :, axis].min(1)
    return (mx <= pos) | ((mn <= pos) & (pos - mn >= mx - pos))


def _tight(P):
    if len(P) == 0:  # get_AABB_from_primitives([]) -> AABB(0,...,0), nn_AABB.py:79-82
        return np.zeros(3), np.zeros(3)
    v = P.reshape(-1, 3)
    return v.min(0), v.max(0)


def best_sah_split(P, box_min, box_max):
    """One node of build_greedy_SAH_tree_tf: (cost, axis, offset) minimising SAH_single_node_tf
    over all candidate offsets of all axes (first minimum wins, as `cost < best` does)."""
    ext = box_max - box_min
    p_surface = F(2.0 * (ext[0] * ext[1] + ext[0] * ext[2] + ext[1] * ext[2]))
    P32 = P.astype(F)
    best = (np.finfo(np.float64).max, None, F(-0.5))
    n = len(P32)
    vmin = P32.min(1)  # per primitive, per coordinate
    vmax = P32.max(1)
    for axis in range(3):
        mn, mx = vmin[:, axis], vmax[:, axis]
        mids = mn + (mx - mn) * F(0.5)
        order = np.argsort(mids, kind="stable")
        smids = mids[order]
        cand, first = np.unique(smids, return_index=True)
        # prims with mid <= cand[k] are order[: last[k] + 1]
        last = np.append(first[1:], n) - 1
        one, zero = F(1.0), F(0.0)
        pre_min = np.minimum(np.minimum.accumulate(vmin[order], 0), one)
        pre_max = np.maximum(np.maximum.accumulate(vmax[order], 0), zero)
        suf_min = np.minimum(np.minimum.accumulate(vmin[order][::-1], 0)[::-1], one)
        suf_max = np.maximum(np.maximum.accumulate(vmax[order][::-1], 0)[::-1], zero)
        lmin, lmax = pre_min[last], pre_max[last]
        has_right = last + 1 < n
        ridx = np.minimum(last + 1, n - 1)
        rmin = np.where(has_right[:, None], suf_min[ridx], one)   # empty set: min -> 1, max -> 0
        rmax = np.where(has_right[:, None], suf_max[ridx], zero)
        lcount = (last + 1).astype(F)
        rcount = (n - 1 - last).astype(F)
        # batches of 8 are padded with the parent's get_max(axis) (z returns y_max: nn_AABB.py:36-37)
        if len(cand) % BATCH:
            fill = F(aabb_get_max(box_max, axis))
            k = int(np.searchsorted(smids, fill, side="right"))  # prims with mid <= fill
            pad = BATCH - len(cand) % BATCH
            fl_min = pre_min[k - 1] if k > 0 else np.full(3, one)
            fl_max = pre_max[k - 1] if k > 0 else np.full(3, zero)
            fr_min = suf_min[k] if k < n else np.full(3, one)
            fr_max = suf_max[k] if k < n else np.full(3, zero)
            cand = np.concatenate([cand, np.full(pad, fill, F)])
            lmin = np.concatenate([lmin, np.tile(fl_min, (pad, 1))])
            lmax = np.concatenate([lmax, np.tile(fl_max, (pad, 1))])
            rmin = np.concatenate([rmin, np.tile(fr_min, (pad, 1))])
            rmax = np.concatenate([rmax, np.tile(fr_max, (pad, 1))])
            lcount = np.concatenate([lcount, np.full(pad, k, F)])
            rcount = np.concatenate([rcount, np.full(pad, n - k, F)])

        def surface(lo, hi):
            e = hi - lo
            return F(2.0) * (e[:, 0] * e[:, 1] + e[:, 0] * e[:, 2] + e[:, 1] * e[:, 2])

        with np.errstate(all="ignore"):
            cost = (((surface(lmin, lmax) / p_surface) * lcount) +
                    ((surface(rmin, rmax) / p_surface) * rcount)) * C_TRI
        # the reference takes reduce_min / argmin per batch of 8 and keeps a batch only if its
        # minimum is strictly below the best so far
        for b0 in range(0, len(cand), BATCH):
            c = cost[b0:b0 + BATCH]
            m = c.min()
            if m < best[0]:
                best = (float(m), axis, cand[b0 + int(np.argmin(c))])
    return best


class TopNode:
    __slots__ = ("prims", "lo", "hi", "axis", "offset", "left", "right", "is_leaf")

    def __init__(self, prims, lo, hi):
        self.prims, self.lo, self.hi = prims, lo, hi
        self.axis = self.offset = self.left = self.right = None
        self.is_leaf = False


def _split(node, P, axis, offset):
    m = split_mask(P[node.prims], axis, float(offset))
    lp, rp = node.prims[m], node.prims[~m]
    node.axis, node.offset = axis, float(offset)
    node.left = TopNode(lp, *_tight(P[lp]))
    node.right = TopNode(rp, *_tight(P[rp]))


def greedy_sah_top(P, levels):
    """build_greedy_SAH_tree_tf: breadth-first, `levels` levels.  P must be scale_scene'd."""
    root = TopNode(np.arange(len(P)), *_tight(P))
    level_nodes = [root]
    for _ in range(levels):
        nxt = []
        for node in level_nodes:
            if node.is_leaf:
                continue
            if len(node.prims) == 0:
                node.is_leaf = True
                continue
            cost, axis, offset = best_sah_split(P[node.prims], node.lo, node.hi)
            if axis is None:  # no candidate produced a finite cost: the reference would crash
                node.is_leaf = True
                continue
            _split(node, P, axis, offset)
            if len(node.left.prims) > 0 or len(node.right.prims) > 0:
                nxt += [node.left, node.right]
            else:
                node.is_leaf, node.left, node.right = True, None, None
        level_nodes = nxt
    for node in level_nodes:
        node.is_leaf = True
    return root


def top_from_prediction(P, tree_structure, max_prims_per_leaf=2):
    """build_tree_from_nn_prediction: rows [onehot_x, onehot_y, onehot_z, offset] in level order
    (2^levels - 1 rows).  A child with <= MAX_PRIMITIVES_PER_LEAF (2, nn_BVH.py:9) primitives is
    marked a leaf (only the LEFT child: the reference's right-child statement has no effect,
    nn_tree_bench.py:72-73); such a node stays a leaf for traversal although a later row splits it."""
    ts = np.asarray(tree_structure, np.float64)
    root = TopNode(np.arange(len(P)), *_tight(P))
    hierarchy = [root]
    for i, row in enumerate(ts):
        onehot = row[:3]
        if sorted(onehot.tolist()) != [0, 0, 1]:
            raise ValueError("Invalid split axis.")  # nn_tree_bench.py:57
        node = hierarchy[i]
        _split(node, P, int(np.argmax(onehot)), row[3])
        hierarchy += [node.left, node.right]
        if len(node.left.prims) <= max_prims_per_leaf:
            node.left.is_leaf = True
    levels = int(np.log2(len(ts) + 1))
    for node in hierarchy[2 ** levels - 1:]:
        node.is_leaf = True
    return root


def to_list(root):
    """BVHNode.to_list order (nn_BVH.py:104-121): inner nodes and leaves, left-first DFS."""
    inner, leaves, stack = [], [], [root]
    while stack:
        cur = stack.pop()
        while not cur.is_leaf:
            inner.append(cur)
            stack.append(cur.right)
            cur = cur.left
        leaves.append(cur)
    return inner, leaves


def bake(root, verts, tris, max_prims_in_node=4, split_method="sah"):
    """Flatten a top tree over triangles `tris` (rows index `verts`) into
    (nnbvh_linear_node[], leaf-ordered nnbvh_prim[]).  Top-tree leaves are finished by the host
    builder; a child without primitives is dropped (its sibling takes the parent's place) —
    LinearBVHNode cannot express an empty leaf (nprims == 0 means interior)."""
    verts = np.ascontiguousarray(verts, np.float32)
    tris = np.asarray(tris, np.int32)
    all_prims = make_prims(tris)
    nodes, ordered = [], []
    n_ordered = 0

    def emit(node):
        nonlocal n_ordered
        if not node.is_leaf:
            l_empty, r_empty = len(node.left.prims) == 0, len(node.right.prims) == 0
            if l_empty or r_empty:
                return emit(node.right if l_empty else node.left)
            me = len(nodes)
            nodes.append(None)
            first = emit(node.left)
            assert first == me + 1
            second = emit(node.right)
            rec = np.zeros((), NODE_DTYPE)
            rec["pmin"] = np.minimum(nodes[first]["pmin"], nodes[second]["pmin"])
            rec["pmax"] = np.maximum(nodes[first]["pmax"], nodes[second]["pmax"])
            rec["offset"], rec["nprims"], rec["axis"] = second, 0, node.axis
            nodes[me] = rec
            return me
        sub = build_tree(all_prims[node.prims], verts, max_prims_in_node, split_method)
        base = len(nodes)
        sn = sub.nodes.copy()
        interior = sn["nprims"] == 0
        sn["offset"][interior] += base
        sn["offset"][~interior] += n_ordered
        nodes.extend(sn)
        ordered.append(sub.ordered_prims)
        n_ordered += len(sub.ordered_prims)
        return base

    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 10000))
    try:
        emit(root)
    finally:
        sys.setrecursionlimit(old)
    return (np.array(nodes, NODE_DTYPE), np.concatenate(ordered).astype(PRIM_DTYPE))


def greedy_sah_tree(verts, tris, levels=4, max_prims_in_node=4):
    """Config-5 pipeline: normalise (scale_scene), greedy-SAH the top `levels` levels
    (nss_global_config.py:15: lvls = 4), finish with the SAH builder, bake."""
    P = scale_scene(np.asarray(verts, np.float64)[np.asarray(tris)])
    root = greedy_sah_top(P, levels)
    return bake(root, verts, tris, max_prims_in_node), root
