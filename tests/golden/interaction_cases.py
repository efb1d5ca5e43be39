"""Input generator for the Triangle::InteractionFromIntersection vectors (shared by the golden
generator and the live reference comparison).  Record layout = oracle/ref_interaction.cpp's:
p0 p1 p2 [0:9], b0 b1 b2 [9:12], wo [12:15], (unused) [15:18], time [18], flags [19],
uv0 uv1 uv2 [20:26], n0 n1 n2 [26:35], (pad) [35], s0 s1 s2 [36:45].
flags: 1 = mesh has uv, 2 = has normals, 4 = has tangents, 8 = reverseOrientation."""
import numpy as np


def cases(n, seed):
    rng = np.random.default_rng(seed)
    rec = np.zeros((n, 45), np.float32)
    scale = (10.0 ** rng.uniform(-3, 3, size=(n, 1))).astype(np.float32)
    rec[:, 0:9] = rng.uniform(-1, 1, size=(n, 9)) * scale
    b = rng.dirichlet([1, 1, 1], size=n)
    kind = rng.integers(0, 20, size=n)
    b[kind == 1, 0] = 0                       # hit on an edge
    b[kind == 2] = [1, 0, 0]                  # hit on a vertex
    rec[:, 9:12] = b
    rec[:, 12:15] = rng.normal(size=(n, 3)) * (10.0 ** rng.uniform(-2, 2, size=(n, 1)))
    rec[:, 18] = rng.random(n)
    rec[:, 19] = rng.integers(0, 16, size=n)
    uv = rng.random((n, 6))
    k = kind == 3                             # degenerate parameterisation: identical uvs
    uv[k, 2:4] = uv[k, 0:2]
    uv[k, 4:6] = uv[k, 0:2]
    k = kind == 4                             # collinear uvs (determinant exactly or nearly 0)
    uv[k, 4:6] = 0.5 * (uv[k, 0:2] + uv[k, 2:4])
    k = kind == 5                             # tiny uv triangle: |det| around 1e-9, huge dpdu
    uv[k, 2:6] = np.tile(uv[k, 0:2], 2) + rng.uniform(-1, 1, size=(int(k.sum()), 4)) * 4.5e-5
    k = kind == 6                             # very stretched parameterisation
    uv[k] *= 1e-4
    rec[:, 20:26] = uv
    nrm = rng.normal(size=(n, 3, 3))
    nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
    k = kind == 7                             # all three normals equal (dn = 0)
    nrm[k, 1] = nrm[k, 0]
    nrm[k, 2] = nrm[k, 0]
    k = kind == 8                             # interpolated normal exactly zero
    nrm[k] = 0
    k = kind == 9                             # normals opposing the geometric one are included by chance;
    nrm[k] *= -1                              # force some sign flips too
    rec[:, 26:35] = nrm.reshape(n, 9)
    tan = rng.normal(size=(n, 3, 3))
    k = kind == 10                            # zero tangents -> falls back to dpdu
    tan[k] = 0
    k = kind == 11                            # tangent parallel to the normal -> CoordinateSystem
    tan[k] = nrm[k] * 2.0
    rec[:, 36:45] = tan.reshape(n, 9)
    k = kind == 12                            # nearly degenerate (sliver) triangle
    rec[k, 6:9] = rec[k, 0:3] + (rec[k, 3:6] - rec[k, 0:3]) * 0.5 + \
        rng.normal(size=(int(k.sum()), 3)).astype(np.float32) * 1e-6 * scale[k]
    k = kind == 13                            # shading tangent exactly parallel to the shading normal
    rec[k, 26:35] = np.tile(np.float32([0, 0, 1]), 3)
    rec[k, 36:45] = np.tile(np.float32([0, 0, 3]), 3)
    rec[k, 19] = (rec[k, 19].astype(np.int32) | 6).astype(np.float32)
    k = kind == 14                            # degenerate uv AND equal normals (dn == 0)
    rec[k, 22:24] = rec[k, 20:22]
    rec[k, 24:26] = rec[k, 20:22]
    rec[k, 29:32] = rec[k, 26:29]
    rec[k, 32:35] = rec[k, 26:29]
    rec[k, 19] = (rec[k, 19].astype(np.int32) | 3).astype(np.float32)
    k = kind == 15                            # |dpdu x dpdv|^2 underflows (small triangle, huge uv span)
    rec[k, 0:9] = (rng.uniform(-1, 1, size=(int(k.sum()), 9)) * 1e-4).astype(np.float32)
    rec[k, 20:26] = (rng.uniform(-1, 1, size=(int(k.sum()), 6)) * 1e8).astype(np.float32)
    rec[k, 19] = (rec[k, 19].astype(np.int32) | 1).astype(np.float32)
    return rec
