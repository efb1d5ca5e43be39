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


def patch_cases(n, seed):
    """BilinearPatch::InteractionFromIntersection inputs (oracle/ref_interaction.cpp "blp" layout):
    p00 p10 p01 p11 [0:12], hit (u, v) [12:14], wo [14:17], time [17], flags [18] (1 = uv, 2 = n,
    8 = reverseOrientation), uv00 uv10 uv01 uv11 [19:27], n00 n10 n01 n11 [27:39]."""
    rng = np.random.default_rng(seed)
    rec = np.zeros((n, 40), np.float32)
    scale = 10.0 ** rng.uniform(-2, 2, size=(n, 1))
    base = rng.uniform(-1, 1, size=(n, 3)) * scale
    eu = rng.normal(size=(n, 3)) * scale
    ev = rng.normal(size=(n, 3)) * scale
    kind = rng.integers(0, 12, size=n)
    tw = rng.normal(size=(n, 3)) * scale * 0.3
    tw[kind == 0] = 0                                   # planar (parallelogram): d2Pduv = 0
    rec[:, 0:3], rec[:, 3:6], rec[:, 6:9], rec[:, 9:12] = base, base + eu, base + ev, base + eu + ev + tw
    uvh = rng.random((n, 2))
    uvh[kind == 1] = np.round(uvh[kind == 1])            # hits on a corner
    rec[:, 12:14] = uvh
    rec[:, 14:17] = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-2, 2, size=(n, 1))
    rec[:, 17] = rng.random(n)
    rec[:, 18] = rng.choice([0, 1, 2, 3, 8, 9, 10, 11], size=n)
    # texture coordinates: the natural ones (+ noise), rotated / mirrored ones, degenerate ones
    nat = np.array([0, 0, 1, 0, 0, 1, 1, 1], np.float64)
    uv = nat[None] + rng.normal(size=(n, 8)) * 0.05
    k = kind == 2                                       # mirrored in s: cross(dpds, dpdt) flips
    uv[k, 0::2] = 1 - uv[k, 0::2]
    k = kind == 3                                       # s does not vary with u along one edge
    uv[k, 2] = uv[k, 0]
    uv[k, 6] = uv[k, 4]
    k = kind == 4                                       # all corners share one (s, t)
    uv[k] = np.tile(uv[k, 0:2], 4)
    k = kind == 5                                       # swapped axes
    uv[k] = uv[k][:, [0, 1, 4, 5, 2, 3, 6, 7]]
    rec[:, 19:27] = uv
    nrm = rng.normal(size=(n, 4, 3))
    nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
    k = kind == 6                                       # interpolated normal exactly zero
    nrm[k] = 0
    k = kind == 7                                       # normals near +-x / +-y: the other reflection axes
    nrm[k] = np.float64([1, 0.02, 0.01]) * rng.choice([-1, 1], size=(int(k.sum()), 1, 1))
    k = kind == 8
    nrm[k] = np.float64([0.9, 0.9, 0.05]) / np.linalg.norm([0.9, 0.9, 0.05])
    rec[k, 0:3], rec[k, 3:6] = base[k], base[k] + np.float64([0, 0, 1]) * scale[k]   # patch facing +-x / y
    rec[k, 6:9] = base[k] + np.float64([0.7, -0.7, 0]) * scale[k]
    rec[k, 9:12] = rec[k, 3:6] + np.float64([0.7, -0.7, 0]) * scale[k]
    rec[:, 27:39] = nrm.reshape(n, 12)
    return rec


def transform_cases(n, seed):
    """Transform::operator()(SurfaceInteraction) inputs (oracle/ref_interaction.cpp "xf" layout): m[16],
    mInv[16] (affine, row-major), pi low/high, and eleven vectors; rigid, scaled, mirrored transforms."""
    rng = np.random.default_rng(seed)
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)
    S = 10.0 ** rng.uniform(-1.5, 1.5, size=(n, 3)) * rng.choice([-1, 1], size=(n, 3), p=[0.15, 0.85])
    kind = rng.integers(0, 6, n)
    R[kind == 0] = np.eye(3)
    S[kind == 1] = 1.0
    M = np.zeros((n, 4, 4))
    M[:, :3, :3] = R * S[:, None, :]
    M[:, :3, 3] = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-1, 3, size=(n, 1))
    M[kind == 2, :3, 3] = 0
    M[:, 3, 3] = 1
    Mi = np.linalg.inv(M)
    rec = np.zeros((n, 72), np.float32)
    rec[:, 0:16] = M.reshape(n, 16)
    rec[:, 16:32] = Mi.reshape(n, 16)
    p = (rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-2, 3, size=(n, 1))).astype(np.float32)
    err = (np.abs(p) * 10.0 ** rng.uniform(-7, -4, size=(n, 3))).astype(np.float32)
    err[kind == 3] = 0                                 # exact points (IsExact)
    rec[:, 32:35] = p - err
    rec[:, 35:38] = p + err
    vecs = rng.normal(size=(n, 11, 3)) * 10.0 ** rng.uniform(-2, 2, size=(n, 11, 1))
    nrm = vecs[:, 0] / np.linalg.norm(vecs[:, 0], axis=1, keepdims=True)
    vecs[:, 0] = nrm
    sn = nrm + rng.normal(size=(n, 3)) * 0.3
    sn[kind == 4] *= -1                                # shading normal on the other side: FaceForward flips it
    vecs[:, 6] = sn / np.linalg.norm(sn, axis=1, keepdims=True)
    rec[:, 38:71] = vecs.reshape(n, 33)
    return rec
