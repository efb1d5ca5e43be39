"""nn_bvh_amd — MI355X-native (gfx950) BVH traversal + ray-primitive intersection for pbrt's
BVHAggregate hot path.  The product is libnnbvh_hip.so behind include/nnbvh.h; this package
is its Python host-side mirror (buffers and launch only) plus input generators."""
from ._lib import (HIT_DTYPE, NODE_DTYPE, PRIM_DTYPE, RAY_DTYPE, NNBVHError, lib)  # noqa: F401
from .aggregate import BVHAggregate, build_tree, build_tree_gpu, make_prims, make_rays  # noqa: F401
