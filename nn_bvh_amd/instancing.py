"""Two-level scenes: object instances as pbrt builds them (scene.cpp:1521-1577 — each object
definition becomes a BVHAggregate, each ObjectInstance a TransformedPrimitive over it,
cpu/primitive.h:83-101) assembled into the single node / primitive arrays
nnbvh_scene_create_instanced takes.  Host-side array plumbing only."""
import numpy as np

from . import _lib
from ._lib import INSTANCE_DTYPE, NODE_DTYPE, PRIM_DTYPE, ptr
from .aggregate import build_tree


def transform_bounds(render_from_prim12, box6):
    """Transform::operator()(Bounds3f) — what TransformedPrimitive::Bounds() returns."""
    m = np.ascontiguousarray(render_from_prim12, np.float32).reshape(12)
    b = np.ascontiguousarray(box6, np.float32).reshape(6)
    out = np.zeros(6, np.float32)
    _lib.lib().nnbvh_transform_bounds(ptr(m), ptr(b), ptr(out))
    return out


def assemble_two_level(top_prims, verts, objects, placements, max_prims_in_node=4, split_method="sah"):
    """top_prims : PRIM_DTYPE triangles / patches that live directly in the top-level tree (may be empty)
    objects     : list of PRIM_DTYPE arrays, one per object definition (child BVHAggregate)
    placements  : list of (object index, render_from_prim 3x4, prim_from_render 3x4)
    returns (nodes, ordered_prims, instances, n_top_nodes)"""
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
    children = [build_tree(o, verts, max_prims_in_node, split_method) for o in objects]
    n_top = len(top_prims)
    inst_prims = np.zeros(len(placements), PRIM_DTYPE)
    inst_prims["kind"] = 2
    inst_prims["v"][:, 0] = np.arange(len(placements))
    inst_prims["id"] = n_top + np.arange(len(placements))
    all_top = np.concatenate([np.asarray(top_prims, PRIM_DTYPE), inst_prims])
    bounds = np.zeros((len(all_top), 6), np.float32)
    for j, (k, m, _) in enumerate(placements):
        root = children[k].nodes[0]
        bounds[n_top + j] = transform_bounds(m, np.concatenate([root["pmin"], root["pmax"]]))
    top = build_tree(all_top, verts, max_prims_in_node, split_method, prim_bounds=bounds)
    nodes, prims = [top.nodes], [top.ordered_prims]
    node_base, prim_base = [], []
    nb, pb = len(top.nodes), len(top.ordered_prims)
    for c in children:
        node_base.append(nb)
        prim_base.append(pb)
        cn = c.nodes.copy()
        interior = cn["nprims"] == 0
        cn["offset"][interior] += nb
        cn["offset"][~interior] += pb
        nodes.append(cn)
        prims.append(c.ordered_prims)
        nb += len(cn)
        pb += len(c.ordered_prims)
    instances = np.zeros(len(placements), INSTANCE_DTYPE)
    for j, (k, m, mi) in enumerate(placements):
        instances[j]["render_from_prim"] = np.asarray(m, np.float32).reshape(12)
        instances[j]["prim_from_render"] = np.asarray(mi, np.float32).reshape(12)
        instances[j]["root"] = node_base[k]
        instances[j]["n_nodes"] = len(children[k].nodes)
    return (np.concatenate(nodes).astype(NODE_DTYPE), np.concatenate(prims).astype(PRIM_DTYPE),
            instances, len(top.nodes))
