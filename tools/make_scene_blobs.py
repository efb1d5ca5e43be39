#!/usr/bin/env python3
"""Extract flat vertex/index arrays from the reference's scene PLY files into data/<scene>.npz
(git-ignored; they travel to the GPU box with the repo snapshot).  Build container only.

Meshes are every *.ply of the scene's geometry directory in sorted file-name order — the
same set the survey's reference probe loaded (crown 794 files, 3 540 310
triangles), so the V/T aggregates recorded there are reproducible from the blob.  (crown.pbrt
itself references 786 of the 794 files; the other 8 add 108 triangles.)  World space,
identity object transforms."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nn_bvh_amd import scene  # noqa: E402

REF = os.environ.get("NNBVH_REFERENCE", "/root/reference")
SCENES = {
    "crown": "scenes/crown/geometry",
    "bathroom": "scenes/bathroom/models",
    "coffee_maker": "scenes/coffee_maker/models",
}


def extract(name, rel):
    base = os.path.join(REF, rel)
    files = sorted(f for f in os.listdir(base) if f.endswith(".ply"))
    verts, tris, off = [], [], 0
    for f in files:
        v, t = scene.read_ply(os.path.join(base, f))
        verts.append(v)
        tris.append(t + off)
        off += len(v)
    verts, tris = np.concatenate(verts), np.concatenate(tris)
    scene.save_blob(name, verts, tris)
    print(f"{name}: {len(files)} meshes, {len(verts)} verts, {len(tris)} tris -> "
          f"{os.path.getsize(scene.blob_path(name)) / 1e6:.1f} MB")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit(f"{REF} not present: scene blobs can only be made in the build container")
    for n in (sys.argv[1:] or SCENES):
        extract(n, SCENES[n])
